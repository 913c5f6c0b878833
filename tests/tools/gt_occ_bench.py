"""Times the GPU ground-truth occupancy generator at 1080 x 1920 (B = 8) and the C oracle / a numpy restatement per frame on the host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from oracle import cref
from soccdpt_amd.utils.gt_occupancy import OccupancyProcessor
from tests.golden_inputs import gt_occ_inputs
dev = torch.device("cuda:0")
B, H, W, C = 8, 1080, 1920, 3
frames = [gt_occ_inputs(H=H, W=W, seed=s) for s in range(B)]
K = frames[0][2]
disp = torch.from_numpy(np.stack([f[0] for f in frames])).to(dev)
seg = torch.from_numpy(np.stack([f[1] for f in frames])).to(dev)
proc = OccupancyProcessor(K, H, W, (256, 256, 32), (2.0, 2.0, 0.666), (0.0, 0.0, 0.0), (500.0, 2500.0, 200.0), (100.0, 40.0, 0.0), 10, num_classes=C)
for want_points in (True, False):
    for _ in range(3):
        proc.process(disp, seg, want_points=want_points)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        proc.process(disp, seg, want_points=want_points)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
    print(f"GPU B={B} 1080x1920 want_points={want_points}: {t*1e3:.3f} ms per batch = {t/B*1e3:.3f} ms per frame")
P = cref.gt_params(H, W, C, K[0, 0], K[1, 1], K[0, 2], K[1, 2])
t0 = time.perf_counter(); cref.gt_occupancy(frames[0][0], frames[0][1].astype(np.int32), P); t = time.perf_counter() - t0
print(f"C oracle, 1 host core: {t*1e3:.1f} ms per frame")
