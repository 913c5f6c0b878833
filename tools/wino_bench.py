"""Winograd F(2x2, 3x3) convolution (csrc/wino.hip) against the direct implicit-GEMM launch on the decoder's 3x3 shapes: error against float64 of the same 16-bit operands
and device time (events around 30 launches).   python tools/wino_bench.py"""
import math, os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soccdpt_amd.lib import PREC_F16, op_igemm, op_wino_conv, op_wino_weights
dev = torch.device("cuda:0")

def timed(fn, n=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

for name, B, H, C, N in [("RCU 64^2 B=8", 8, 64, 256, 256), ("RCU 32^2 B=8", 8, 32, 256, 256), ("RCU 16^2 B=8", 8, 16, 256, 256), ("seg head 128^2 B=8", 8, 128, 256, 256),
                         ("depth head d0 128^2 B=8", 8, 128, 256, 128), ("RCU 96^2 B=8 (384 models)", 8, 96, 256, 256)]:
    g = torch.Generator().manual_seed(1)
    x = torch.relu(torch.randn((B, H, H, C), generator=g)).half()
    w = torch.randn((N, C, 3, 3), generator=g) / math.sqrt(9 * C)
    bias = torch.randn(N, generator=g) * 0.1
    xh = torch.zeros((B, H + 2, H + 2, C), dtype=torch.float16)
    xh[:, 1:-1, 1:-1] = x
    xh = xh.to(dev)
    wd = w.to(dev)
    u = op_wino_weights(wd)
    wt = w.permute(0, 2, 3, 1).reshape(N, 9 * C).contiguous().half().to(dev)
    bd = bias.to(dev)
    o_w = torch.empty((B * H * H, N), device=dev)
    o_d = torch.empty((B * H * H, N), device=dev)
    oh = torch.zeros((B, H + 2, H + 2, N), dtype=torch.float16, device=dev)
    f_w = lambda: op_wino_conv(xh, u, B, H, H, C, N, bias=bd, relu=True, out_f32=o_w, out_op=oh, out_halo=True)
    f_d = lambda: op_igemm(xh, wt, B * H * H, N, C, taps=9, H=H, W=H, bias=bd, act=1, out_f32=o_d, out_bf16=oh, out_halo=1, precision=PREC_F16)
    t_w, t_d = timed(f_w), timed(f_d)
    nb = min(B, 2)
    ref = F.conv2d(x[:nb].permute(0, 3, 1, 2).double(), w.half().double(), bias.double(), padding=1).permute(0, 2, 3, 1).reshape(-1, N)
    rows = nb * H * H
    e_w = float((o_w[:rows].cpu().double() - ref).norm() / ref.norm())
    e_d = float((o_d[:rows].cpu().double() - ref).norm() / ref.norm())
    nwg = B * (H // 16) ** 2 * (N // 64)
    stamps = torch.zeros(8 * nwg, dtype=torch.int64, device=dev)
    op_wino_conv(xh, u, B, H, H, C, N, bias=bd, relu=True, out_f32=o_w, out_op=oh, out_halo=True, stamps=stamps)
    torch.cuda.synchronize()
    sm = stamps.cpu().numpy().reshape(-1, 8).astype("float64") / 100.0
    dd = sm - sm[:, :1]
    print(f"   per workgroup ({nwg}), mean us since entry: prologue done {dd[:, 1].mean():.2f}; chunk 4: start {dd[:, 2].mean():.2f}, V built {dd[:, 3].mean():.2f}, MFMAs issued {dd[:, 4].mean():.2f}, "
          f"barrier passed {dd[:, 5].mean():.2f}; k loop done {dd[:, 6].mean():.2f}; exit {dd[:, 7].mean():.2f}; entry spread {sm[:, 0].max() - sm[:, 0].min():.2f}, last exit {sm[:, 7].max() - sm[:, 0].min():.2f}")
    fl = 2.0 * B * H * H * N * 9 * C
    print(f"{name}: winograd {t_w:.1f} us ({fl / t_w / 1e6:.0f} TFLOP/s of direct-conv FLOPs), direct igemm {t_d:.1f} us ({fl / t_d / 1e6:.0f} TFLOP/s); "
          f"rel L2 vs float64 of the fp16 operands: winograd {e_w:.2e}, direct {e_d:.2e}", flush=True)
