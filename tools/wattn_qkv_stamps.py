"""Where does a workgroup of the fused qkv + window-attention kernel (csrc/attention_qkv.hip) spend its time?  Per-workgroup s_memrealtime stamps (100 MHz) at
entry / first weight panel staged / GEMM phase done / q-hat, k-hat, V^T in LDS / exit, for the four stage shapes of dpt_swin2_tiny_256 at B = 8, fp16 and x2w
weights, beside the device time of the launch and of the two-launch chain it replaces (igemm + window_attention).   python tools/wattn_qkv_stamps.py"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soccdpt_amd.lib import PREC_F16, PREC_F16X2W, op_igemm, op_window_attention, op_window_attention_qkv, x3_encode
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

for name, B, res, ws, shift, heads in [("tiny s0", 8, 64, 16, 8, 3), ("tiny s1", 8, 32, 16, 8, 6), ("tiny s2", 8, 16, 16, 0, 12), ("tiny s3", 8, 8, 8, 0, 24)]:
    C, M = heads * 32, B * res * res
    g = torch.Generator().manual_seed(1)
    x = torch.randn((M, C), generator=g).half().to(dev)
    w = torch.randn((3 * C, C), generator=g) / math.sqrt(C)
    bias = (torch.randn(3 * C, generator=g) * 0.3).to(dev)
    table = (torch.randn(((2 * ws - 1) ** 2, heads), generator=g) * 0.5).to(dev)
    scale = torch.full((heads,), 12.0, device=dev)
    out = torch.empty((M, C), dtype=torch.float16, device=dev)
    qkv = torch.empty((M, 3 * C), dtype=torch.float16, device=dev)
    nwg = B * (res // ws) ** 2 * heads
    for fmt, code, wd in (("fp16", PREC_F16, w.half().to(dev)), ("x2w", PREC_F16X2W, x3_encode(w.to(dev)))):
        stamps = torch.zeros(5 * nwg, dtype=torch.int64, device=dev)
        t_f = timed(lambda: op_window_attention_qkv(x, wd, bias, table, scale, out, B, res, ws, shift, heads, code))
        t_g = timed(lambda: op_igemm(x, wd, M, 3 * C, C, ldx=C, bias=bias, out_bf16=qkv, precision=code))
        t_a = timed(lambda: op_window_attention(qkv, table, scale, out, B, res, ws, shift, heads, PREC_F16))
        op_window_attention_qkv(x, wd, bias, table, scale, out, B, res, ws, shift, heads, code, stamps=stamps)
        torch.cuda.synchronize()
        s = stamps.cpu().numpy().reshape(-1, 5).astype("float64") / 100.0   # us
        t0 = s[:, 0].min()
        d = s - s[:, :1]
        print(f"{name} {fmt}: {nwg} workgroups, C {C}; fused launch {t_f:.1f} us (incl. the bias-table launch of the entry), chain igemm {t_g:.1f} + attention {t_a:.1f} us | "
              f"per workgroup, mean us since entry: first panel {d[:, 1].mean():.2f}, GEMM done {d[:, 2].mean():.2f}, q/k/v staged {d[:, 3].mean():.2f}, exit {d[:, 4].mean():.2f}; "
              f"entry spread {s[:, 0].max() - t0:.2f}, last exit {s[:, 4].max() - t0:.2f}", flush=True)
