"""Does the operand FORMAT or the operand DATA make the fp16 forward slower than the bf16 one in the same kernel?  One long-K 3x3 convolution
(M 32768, N 256, K 2304; the 8-wave 128x128x64 tile) timed with bf16 / fp16 operands holding (a) zeros, (b) N(0,1) data, interleaved in one process."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from soccdpt_amd.lib import op_igemm, PREC_BF16, PREC_F16

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, H, Cin, Cout = 8, 64, 256, 256
M, N, K = B * H * H, Cout, 9 * Cin
xr = torch.randn(B, H + 2, H + 2, Cin, generator=g)
wr = torch.randn(Cout, 9 * Cin, generator=g) / math.sqrt(9 * Cin)

def bench(fn, iters=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

cases = {}
for name, dt, prec in (("bf16", torch.bfloat16, PREC_BF16), ("f16", torch.float16, PREC_F16)):
    for data, (x, w) in (("zeros", (torch.zeros_like(xr), torch.zeros_like(wr))), ("randn", (xr, wr))):
        xd, wd = x.to(dt).to(dev), w.to(dt).to(dev)
        out = torch.empty(M, N, dtype=dt, device=dev)
        cases[f"{name}/{data}"] = (lambda xd=xd, wd=wd, out=out, prec=prec: op_igemm(xd, wd, M, N, Cin, taps=9, H=H, W=H, act=1, out_bf16=out, tune=21, precision=prec))
res = {k: [] for k in cases}
for rep in range(3):
    for k, fn in cases.items():
        res[k].append(bench(fn))
for k, v in res.items():
    us = sorted(v)[1]
    print(f"{k:12s} {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s   {['%.1f' % t for t in v]}")
