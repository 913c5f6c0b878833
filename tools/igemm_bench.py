"""Micro-benchmark of the igemm kernel configurations on the shapes of the B=8 dpt_swin2_tiny_256 forward
(run on the GPU box; prints TFLOP/s per (shape, config)).  Interleaved rounds in ONE process."""
import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
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
    return e0.elapsed_time(e1) / iters * 1e3  # us

shapes = [
    ("conv128_256x256", "conv", 8, 128, 256, 256), ("conv128_256x128", "conv", 8, 128, 256, 128), ("conv64_256", "conv", 8, 64, 256, 256),
    ("conv256_128x32", "conv", 8, 256, 128, 32),
]
cfgs = {1: "128x128x64s2", 8: "256x256x64s2", 10: "128x256x64s2", 15: "128x256x32s3", 16: "256x128x32s3", 17: "128x256x32s4", 18: "256x256x32s3", 5: "128x32x64s4"}
for sh in shapes:
    name, kind = sh[0], sh[1]
    if kind == "conv":
        _, _, B, H, Cin, Cout = sh
        x = torch.randn(B, H + 2, H + 2, Cin, generator=g).to(torch.bfloat16).to(dev)
        w = (torch.randn(Cout, 9 * Cin, generator=g) / math.sqrt(9 * Cin)).to(torch.bfloat16).to(dev)
        M, N, K = B * H * H, Cout, 9 * Cin
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        def run(t): op_igemm(x, w, M, N, Cin, taps=9, H=H, W=H, act=1, out_bf16=out, tune=t)
    else:
        _, _, M, N, K = sh
        Cin = K
        x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
        w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).to(dev)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        def run(t): op_igemm(x, w, M, N, K, ldx=K, out_bf16=out, tune=t)
    flops = 2.0 * M * N * K
    res = []
    for t, cn in cfgs.items():
        if (N <= 32) != (t == 5): continue
        us = min(bench(lambda: run(t)) for _ in range(2))
        res.append(f"{cn}: {us:7.1f}us {flops / us / 1e6:7.1f}TF")
    us = bench(lambda: run(-1))
    print(f"{name:18s} M={M:6d} N={N:5d} K={K:5d} | " + " | ".join(res) + f" | auto {us:7.1f}us")
