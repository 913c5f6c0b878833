"""Device time of the Swin-V2 window attention launch shapes of dpt_swin2_base_384 / tiny_256 at B = 8 (kernel-level entry; events around 30 launches, the
bias-table launch of the entry included: run under rocprofv3 --kernel-trace --stats for the kernel alone)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soccdpt_amd.lib import op_window_attention, PREC_F16
dev = torch.device("cuda:0")
for name, B, res, ws, shift, heads in [("base s2", 8, 24, 24, 0, 16), ("base s1", 8, 48, 24, 12, 8), ("base s0", 8, 96, 24, 12, 4), ("tiny s0", 8, 64, 16, 8, 3), ("tiny s2", 8, 16, 16, 0, 12)]:
    C = heads * 32
    g = torch.Generator().manual_seed(1)
    qkv = torch.randn((B * res * res, 3 * C), generator=g).half().to(dev)
    table = (torch.randn(((2 * ws - 1) ** 2, heads), generator=g) * 0.5).to(dev)
    scale = torch.full((heads,), 2.3, device=dev)
    out = torch.empty((B * res * res, C), dtype=torch.float16, device=dev)
    for _ in range(3):
        op_window_attention(qkv, table, scale, out, B, res, ws, shift, heads, PREC_F16)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(30):
        op_window_attention(qkv, table, scale, out, B, res, ws, shift, heads, PREC_F16)
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: B {B} res {res} ws {ws} shift {shift} heads {heads}: {e0.elapsed_time(e1) * 1e3 / 30:.1f} us per call (attention + bias table)", flush=True)
