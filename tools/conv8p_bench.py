"""A/B of the phase-interleaved conv tiles (conv8p.hip, configurations 30 / 31 / 32) against the 128x128x64 8-wave tile (21) and the
other big tiles on the decoder / head convolution shapes, interleaved rounds in one process (cdna_hip_programming.md rule 24), cold-ish:
a 512 MB buffer is rewritten between launches so operands come from HBM / MALL like in the network.
    python tools/conv8p_bench.py"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from soccdpt_amd.lib import op_igemm

dev = torch.device("cuda:0")
shapes = [("seg head / RCU 128^2 B=8", 8, 128, 256, 256), ("depth head d0 128^2 B=8", 8, 128, 256, 128), ("RCU 64^2 B=8", 8, 64, 256, 256),
          ("RCU 192^2 B=4 (384 models)", 4, 192, 256, 256), ("RCU 96^2 B=8 (384 models)", 8, 96, 256, 256)]
ABL = os.environ.get("CONV8P_ABLATE") == "1"   # ablations of the 256x256 kernel on the first shape: 33 no setprio, 34 wait after barrier,
if ABL:                                          # 35 no staging, 36 no fragment reads, 37 no MFMAs, 38 = 33+34, 39 = no staging + no reads
    shapes = shapes[:1]
flush = torch.empty(128 * 1024 * 1024, dtype=torch.float32, device=dev)
for name, B, H, Cin, Cout in shapes:
    g = torch.Generator().manual_seed(1)
    xh = torch.zeros(B, H + 2, H + 2, Cin, dtype=torch.bfloat16, device=dev)
    xh[:, 1:-1, 1:-1] = torch.randn(B, H, H, Cin, generator=g).to(torch.bfloat16).to(dev)
    wt = (torch.randn(Cout, 9 * Cin, generator=g) / math.sqrt(9 * Cin)).to(torch.bfloat16).to(dev)
    bias = torch.randn(Cout, generator=g).to(dev)
    outh = torch.zeros(B, H + 2, H + 2, Cout, dtype=torch.bfloat16, device=dev)
    M = B * H * H
    cands = [21, 16, 47, 51, 52] if Cout % 256 == 0 else [21, 1, 16, 49, 50]   # 47-49: the 16-wave tiles of round 5
    if os.environ.get("CONV_BENCH_M32") == "1":   # the v_mfma_f32_32x32x16 forms (40-45) against the 16x16x32 tiles of the same shape
        cands = [21, 41, 1, 40, 45, 42] + ([43, 44, 30] if Cout % 256 == 0 else [])
    if ABL:
        cands = [21, 30, 33, 34, 35, 36, 37, 38, 39]
    res = {c: [] for c in cands}
    for rnd in range(6):
        for c in cands:
            flush.fill_(float(rnd))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            op_igemm(xh, wt, M, Cout, Cin, taps=9, H=H, W=H, bias=bias, act=1, out_bf16=outh, out_halo=1, tune=c)
            e1.record()
            torch.cuda.synchronize()
            res[c].append(e0.elapsed_time(e1) * 1e3)
    fl = 2.0 * M * Cout * 9 * Cin
    print(name, f"M={M} N={Cout} K={9*Cin}:", "  ".join(f"cfg{c}: {sorted(v)[len(v)//2]:.1f} us ({fl / sorted(v)[len(v)//2] / 1e6:.0f} TF/s)" for c, v in res.items()), flush=True)
