"""Where does a small GEMM launch spend its time?  Per-workgroup s_memrealtime stamps (igemm.h: IgemmDesc::stamps) of the encoder's
latency-bound Linear layers: dispatch spread (first to last workgroup entry), time to the first landed k-tile, main loop, epilogue, and
the launch's wall span -- the numbers behind DESIGN.md's "every kernel costs >= 4.4 us" floor.  Stamps live in the generalised-addressing
instantiations only (configurations 2 = 64x64x64 and 20 = 32x64x128, 4 waves), which is what these launches then run on.
    python tools/igemm_stamps.py"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from soccdpt_amd.lib import op_igemm

dev = torch.device("cuda:0")
shapes = [("stage2 qkv", 2048, 1152, 384, 0), ("stage2 proj", 2048, 384, 384, 0), ("stage2 fc1+GELU", 2048, 1536, 384, 2), ("stage2 fc2", 2048, 384, 1536, 0),
          ("stage3 fc1+GELU", 512, 3072, 768, 2), ("stage1 qkv", 8192, 576, 192, 0)]
for name, M, N, K, act in shapes:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    stamps = torch.zeros(4 * 8192, dtype=torch.int64, device=dev)
    for rep in range(5):
        stamps.zero_()
        torch.cuda.synchronize()
        op_igemm(x, w, M, N, K, ldx=K, bias=bias, act=act, out_bf16=out, stamps=stamps)
        torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(-1, 4)
    s = s[s[:, 0] > 0].astype(np.float64) * 10.0      # ns (100 MHz counter)
    t0 = s[:, 0].min()
    s -= t0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        op_igemm(x, w, M, N, K, ldx=K, bias=bias, act=act, out_bf16=out)
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:18s} M={M} N={N} K={K}: {len(s)} workgroups | entry spread {s[:,0].max()/1e3:.2f} us | entry->first tile {np.median(s[:,1]-s[:,0])/1e3:.2f} "
          f"| main loop {np.median(s[:,2]-s[:,1])/1e3:.2f} | epilogue {np.median(s[:,3]-s[:,2])/1e3:.2f} | workgroup life {np.median(s[:,3]-s[:,0])/1e3:.2f} "
          f"| kernel span (first entry -> last exit) {s[:,3].max()/1e3:.2f} us | back-to-back launches {e0.elapsed_time(e1)*1e3/20:.2f} us each", flush=True)
