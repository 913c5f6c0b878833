"""Where does a big 3x3 convolution launch spend its time?  Per-workgroup s_memrealtime stamps (entry, first k-tile landed, main loop done, epilogue done)
of the 8-wave 128 x 128 x 64 tile on the decoder / head shapes, fp16 operands, cold-ish (a 512 MB buffer is rewritten before each launch).
    python tools/conv_stamps.py"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from soccdpt_amd.lib import op_igemm

dev = torch.device("cuda:0")
flush = torch.empty(128 * 1024 * 1024, dtype=torch.float32, device=dev)
for name, B, H, Cin, Cout, full in [("RCU 64^2", 8, 64, 256, 256, False), ("RCU 64^2 + f32 out + residual", 8, 64, 256, 256, True), ("seg head 128^2", 8, 128, 256, 256, False),
                                    ("depth conv0 128^2", 8, 128, 256, 128, False), ("RCU 96^2", 8, 96, 256, 256, False)]:
    g = torch.Generator().manual_seed(1)
    xh = torch.zeros(B, H + 2, H + 2, Cin, dtype=torch.float16, device=dev)
    xh[:, 1:-1, 1:-1] = torch.randn(B, H, H, Cin, generator=g).to(torch.float16).to(dev)
    wt = (torch.randn(Cout, 9 * Cin, generator=g) / math.sqrt(9 * Cin)).to(torch.float16).to(dev)
    bias = torch.randn(Cout, generator=g).to(dev)
    outh = torch.zeros(B, H + 2, H + 2, Cout, dtype=torch.float16, device=dev)
    M = B * H * H
    res1 = torch.randn(M, Cout, generator=g).to(dev) if full else None
    outf = torch.empty(M, Cout, device=dev) if full else None
    stamps = torch.zeros(4 * 8192, dtype=torch.int64, device=dev)
    rows = []
    for rep in range(4):
        stamps.zero_()
        flush.fill_(float(rep))
        torch.cuda.synchronize()
        op_igemm(xh, wt, M, Cout, Cin, taps=9, H=H, W=H, bias=bias, act=1, out_bf16=outh, out_halo=1, res1=res1, out_f32=outf, tune=21, precision=2, stamps=stamps)
        torch.cuda.synchronize()
        s = stamps.cpu().numpy().reshape(-1, 4)
        s = s[s[:, 0] > 0].astype(np.float64) * 10.0
        s -= s[:, 0].min()
        rows.append(s)
    s = rows[-1]
    ts = []
    for rep in range(4):
        flush.fill_(float(rep))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        op_igemm(xh, wt, M, Cout, Cin, taps=9, H=H, W=H, bias=bias, act=1, out_bf16=outh, out_halo=1, res1=res1, out_f32=outf, tune=21, precision=2)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    fl = 2.0 * M * Cout * 9 * Cin
    q = lambda a, p: np.percentile(a, p) / 1e3
    life = s[:, 3] - s[:, 0]
    print(f"{name:30s} M={M} N={Cout}: {len(s)} workgroups, un-stamped launch {sorted(ts)[1]:.1f} us ({fl / sorted(ts)[1] / 1e6:.0f} TF/s) | entry p50 {q(s[:,0],50):.1f} p99 {q(s[:,0],99):.1f} max {s[:,0].max()/1e3:.1f} us "
          f"| entry->first tile p50 {q(s[:,1]-s[:,0],50):.2f} | main loop p50 {q(s[:,2]-s[:,1],50):.1f} p10 {q(s[:,2]-s[:,1],10):.1f} p90 {q(s[:,2]-s[:,1],90):.1f} "
          f"| epilogue p50 {q(s[:,3]-s[:,2],50):.2f} p90 {q(s[:,3]-s[:,2],90):.2f} | workgroup life p50 {q(life,50):.1f} | span {s[:,3].max()/1e3:.1f} us", flush=True)
