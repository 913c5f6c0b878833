"""Do back-to-back launches of one kernel on one stream overlap?  Event-timed average of 50 launches vs the rocprofv3 durations
of the same 50 launches (run under rocprofv3 --kernel-trace)."""
import sys, os, math, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from soccdpt_amd.lib import op_igemm
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for name, M, N, K in [("lin_s0_qkv", 32768, 288, 96), ("lin_s2_proj", 2048, 384, 384), ("lin_s2_qkv", 2048, 1152, 384)]:
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).to(dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for _ in range(5):
        op_igemm(x, w, M, N, K, ldx=K, out_bf16=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(50):
        op_igemm(x, w, M, N, K, ldx=K, out_bf16=out)
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: events {e0.elapsed_time(e1) / 50 * 1e3:.2f} us/launch; host enqueue {(t1 - t0) / 50 * 1e6:.2f} us/launch; wall to sync {(t2 - t0) / 50 * 1e6:.2f} us/launch", flush=True)
