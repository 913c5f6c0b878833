"""igemm configurations on the 64x64 / 128x128 RCU convolutions WITH their real epilogues
(f32 raw + bf16 halo outputs, one or two f32 residual inputs)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from soccdpt_amd.lib import op_igemm
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (B, H, Cin, Cout) in [(8, 64, 256, 256), (8, 128, 256, 256), (8, 128, 256, 128)]:
    x = torch.randn(B, H + 2, H + 2, Cin, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(Cout, 9 * Cin, generator=g) / math.sqrt(9 * Cin)).to(torch.bfloat16).to(dev)
    M = B * H * H
    bias = torch.randn(Cout, device=dev)
    r1 = torch.randn(M, Cout, device=dev); r2 = torch.randn(M, Cout, device=dev)
    of = torch.empty(M, Cout, device=dev)
    oh = torch.zeros(B, H + 2, H + 2, Cout, dtype=torch.bfloat16, device=dev)
    ob = torch.empty(M, Cout, dtype=torch.bfloat16, device=dev)
    flops = 2.0 * M * Cout * 9 * Cin
    variants = {
        "bf16 plain out": lambda t: op_igemm(x, w, M, Cout, Cin, taps=9, H=H, W=H, bias=bias, act=1, out_bf16=ob, tune=t),
        "relu halo out": lambda t: op_igemm(x, w, M, Cout, Cin, taps=9, H=H, W=H, bias=bias, act=1, out_bf16=oh, out_halo=1, tune=t),
        "res1 + plain": lambda t: op_igemm(x, w, M, Cout, Cin, taps=9, H=H, W=H, bias=bias, res1=r1, out_bf16=ob, tune=t),
        "res1+res2 -> f32 + halo": lambda t: op_igemm(x, w, M, Cout, Cin, taps=9, H=H, W=H, bias=bias, res1=r1, res2=r2, act=1, out_f32=of, out_bf16=oh, out_halo=1, tune=t),
    }
    for name, fn in variants.items():
        res = []
        for t, cn in {1: "128x128s2", 10: "128x256s2", 8: "256x256s2", 6: "256x128s2"}.items():
            us = min(bench(lambda: fn(t)) for _ in range(2))
            res.append(f"{cn}: {us:6.1f}us {flops/us/1e6:6.0f}TF")
        print(f"conv{H} {Cin}->{Cout} [{name:24s}] " + " | ".join(res))
