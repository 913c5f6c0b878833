"""Where do the hybrid's ResNetV2 convolution launches spend their time?  Per-workgroup s_memrealtime stamps (IgemmDesc::stamps) of the
bottleneck launches of dpt_hybrid_384 at B = 4, in the formats the mixed mode runs them in (x3 with the GroupNorm-statistics epilogue), next
to the same launch without the statistics and in fp16: dispatch spread, entry -> first landed k-tile, main loop, epilogue, launch span.
    python tools/rn_stamps.py > gpurun_out/rn_stamps.txt"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from soccdpt_amd.lib import PREC_F16, PREC_F16X3, op_igemm, x3_encode

dev = torch.device("cuda:0")
B = 4
# name, side (output pixels per side), Cin, Cout, taps
shapes = [("s0 conv1 1x1", 96, 256, 64, 1), ("s0 conv2 3x3", 96, 64, 64, 9), ("s0 conv3 1x1", 96, 64, 256, 1),
          ("s1 conv1 1x1", 48, 512, 128, 1), ("s1 conv2 3x3", 48, 128, 128, 9), ("s1 conv3 1x1", 48, 128, 512, 1),
          ("s2 conv1 1x1", 24, 1024, 256, 1), ("s2 conv2 3x3", 24, 256, 256, 9), ("s2 conv3 1x1", 24, 256, 1024, 1)]
only = sys.argv[1] if len(sys.argv) > 1 else ""


def run(name, side, Cin, Cout, taps, prec, stats_on, tune=-1):
    g = torch.Generator().manual_seed(1)
    M = B * side * side
    if taps == 9:
        xf = torch.zeros((B, side + 2, side + 2, Cin))
        xf[:, 1:-1, 1:-1] = torch.randn((B, side, side, Cin), generator=g)
        kw = dict(taps=9, conv=dict(stride=1, pad=1, in_halo=1, Hi=side, Wi=side))
    else:
        xf = torch.randn((M, Cin), generator=g)
        kw = dict(taps=1)
    wf = torch.randn((Cout, taps * Cin), generator=g) / math.sqrt(taps * Cin)
    if prec == PREC_F16X3:
        x, w = x3_encode(xf.to(dev)), x3_encode(wf.to(dev))
    else:
        x, w = xf.to(torch.float16).to(dev), wf.to(torch.float16).to(dev)
    out = torch.empty((M, Cout), device=dev)
    kws = {}
    if stats_on:
        kws = dict(gn_stats=torch.zeros((B, 32, 2), device=dev), gn_part=torch.zeros(((M // 32) * 32 * 2 * 4,), device=dev),
                   gn_count=torch.zeros((B + 8,), dtype=torch.int32, device=dev) if stats_on == 2 else None, gn_cpg=Cout // 32, gn_hw=side * side)
    stamps = torch.zeros(4 * 16384, dtype=torch.int64, device=dev)
    for rep in range(4):
        stamps.zero_()
        torch.cuda.synchronize()
        op_igemm(x, w, M, Cout, Cin, H=side, W=side, ldx=Cin, out_f32=out, precision=prec, stamps=stamps, tune=tune, **kw, **kws)
        torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(-1, 4)
    s = s[s[:, 0] > 0].astype(np.float64) * 10.0      # ns (100 MHz counter)
    s -= s[:, 0].min()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        op_igemm(x, w, M, Cout, Cin, H=side, W=side, ldx=Cin, out_f32=out, precision=prec, tune=tune, **kw, **kws)
    e1.record()
    torch.cuda.synchronize()
    per = e0.elapsed_time(e1) * 1e3 / 20
    fl = 2.0 * M * Cout * Cin * taps
    by = (M * Cin * (4 if prec == PREC_F16X3 else 2) + M * Cout * 4) / 1e6
    p = lambda a, q: np.percentile(a, q) / 1e3
    print(f"{name:13s} {'x3 ' if prec == PREC_F16X3 else 'f16'} {('ST', 'SD', 'ST')[stats_on] if stats_on else '  '} tune {tune:2d} M={M} N={Cout} K={taps * Cin}: {len(s):5d} wgs | launch {per:6.1f} us "
          f"({fl / per / 1e6:5.0f} TF, {by / per * 1e6 / 1e6:5.2f} TB/s of {by:5.1f} MB) | entry p50 {p(s[:,0],50):5.1f} max {p(s[:,0],100):5.1f} | first tile {p(s[:,1]-s[:,0],50):5.2f} | loop {p(s[:,2]-s[:,1],50):5.2f} "
          f"| epilogue p50 {p(s[:,3]-s[:,2],50):5.2f} max {p(s[:,3]-s[:,2],100):5.2f} | life p50 {p(s[:,3]-s[:,0],50):5.1f} | span {p(s[:,3],100):5.1f}", flush=True)


for sh in shapes:
    if only and only not in sh[0]:
        continue
    # stats_on: 1 = deferred partials (SD: what the forward launches since round 5), 2 = in-kernel last-arriver finish (ST), 0 = none
    run(*sh, PREC_F16X3, 2)
    for t in (-1, 4, 12):
        try:
            run(*sh, PREC_F16X3, 1, t)
        except RuntimeError as e:
            print(f"{sh[0]:13s} x3  SD tune {t}: {str(e)[-80:]}")
    run(*sh, PREC_F16X3, 0)
    run(*sh, PREC_F16, 1)
