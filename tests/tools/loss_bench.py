"""Times the fused training criterion (B = 3 and 8, 256^2 -> 1080 x 1920) and, beside it, the same criterion as the reference
forms it with torch ops + autograd on the GPU (oracle/loss_ref.py moved to the device: test infrastructure, timed as the baseline)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import loss_ref as LR
from soccdpt_amd.utils.loss import training_loss
from tests.golden_inputs import loss_inputs
dev = torch.device("cuda:0")
for B in (3, 8):
    ins = [t.to(dev) for t in loss_inputs(B=B, h=256, w=256, H=1080, W=1920, seed=1)]
    for _ in range(3):
        training_loss(*ins)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        training_loss(*ins)
    torch.cuda.synchronize(); t_hip = (time.perf_counter() - t0) / 20
    for _ in range(2):
        LR.loss_and_grads(*ins)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        LR.loss_and_grads(*ins)
    torch.cuda.synchronize(); t_torch = (time.perf_counter() - t0) / 5
    px = B * 1080 * 1920
    alg = px * (4 + 4 + 1) + px * 3 * (4 + 1)   # read targets + masks once (algorithmic minimum)
    print(f"B={B}: fused HIP {t_hip*1e3:.3f} ms ({alg/t_hip/1e9:.0f} GB/s of the algorithmic {alg/1e6:.0f} MB) | torch ops + autograd on the same GPU {t_torch*1e3:.2f} ms | x{t_torch/t_hip:.1f}")
