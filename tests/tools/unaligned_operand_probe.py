"""Feasibility probe: does the igemm (LDS-DMA 16-byte global -> LDS loads) accept a weight operand whose base address is only 4-byte
aligned?  out[n][c] = sum_k X[n][k] W[c][k + off].  Used to decide whether the 3x3 weight-gradient GEMM can read shifted views of ONE
transposed halo image (9 pointer offsets) instead of an explicit im2col^T.  Prints max abs error and the time per launch for a few offsets."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from soccdpt_amd.lib import op_igemm

dev = torch.device("cuda:0")
N, C, K, PAD = 256, 256, 8192, 128
g = torch.Generator().manual_seed(0)
X = torch.randn(N, K, generator=g).to(dev)
Wfull = torch.randn(C * K + 2 * PAD, generator=g).to(dev)
for off in (0, 4, 1, 2, 3, -67, 67):
    base = Wfull[PAD + off: PAD + off + C * K]            # a view whose data_ptr is shifted by `off` floats
    ref = X.double() @ base.view(C, K).double().t()
    out = torch.empty(N, C, device=dev)
    op_igemm(X, base, N, C, K, ldx=K, out_f32=out, f32=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        op_igemm(X, base, N, C, K, ldx=K, out_f32=out, f32=1)
    e1.record()
    torch.cuda.synchronize()
    print(f"offset {off:4d} floats (address % 16 = {base.data_ptr() % 16:2d}): max abs err {float((out.double() - ref).abs().max()):.2e}, {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", flush=True)
