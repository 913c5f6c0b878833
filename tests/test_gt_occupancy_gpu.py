"""GPU: soccdpt_gt_occupancy against the golden recorded from the reference's own OccupancyProcessor.process_frame and against the
C oracle on a second, larger frame pair -- occupancy grid, counts, depth and the float64 point cloud all bit for bit."""
import os

import numpy as np
import pytest
import torch

from oracle import cref
from tests.golden_inputs import gt_occ_inputs

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "gt_occupancy.npz"))


def _proc(K, H, W, C):
    from soccdpt_amd.utils.gt_occupancy import OccupancyProcessor
    return OccupancyProcessor(intrinsic_matrix=K, height=H, width=W, grid_size=(256, 256, 32), scale=(2.0, 2.0, 0.666), shift=(0.0, 0.0, 0.0),
                              pc_scale=(500.0, 2500.0, 200.0), pc_shift=(100.0, 40.0, 0.0), point_count_threshold=10, num_classes=C)


def test_gt_occupancy_matches_reference_golden(gpu_device):
    disp, seg, K, H, W, C = gt_occ_inputs()
    r = _proc(K, H, W, C).process(torch.from_numpy(disp).to(gpu_device), torch.from_numpy(seg).to(gpu_device))
    torch.cuda.synchronize()
    grid = np.unpackbits(G["grid_bits"])[: int(np.prod(G["grid_shape"]))].reshape(G["grid_shape"]).astype(bool)
    assert np.array_equal(r["occupancy_grid"][0].cpu().numpy(), grid)
    assert np.array_equal(r["depth"][0].cpu().numpy().view(np.uint32), G["depth"].view(np.uint32))
    assert np.array_equal(r["points"][0].cpu().numpy()[G["point_rows"]], G["points_sample"])


def test_gt_occupancy_batch_vs_c_oracle(gpu_device):
    frames = [gt_occ_inputs(H=540, W=960, seed=s) for s in (3, 4)]
    K, H, W, C = frames[0][2], 540, 960, 3
    disp = np.stack([f[0] for f in frames])
    seg = np.stack([f[1] for f in frames])
    r = _proc(K, H, W, C).process(torch.from_numpy(disp).to(gpu_device), torch.from_numpy(seg).to(gpu_device))
    torch.cuda.synchronize()
    P = cref.gt_params(H, W, C, K[0, 0], K[1, 1], K[0, 2], K[1, 2])
    for b in range(2):
        o = cref.gt_occupancy(disp[b], seg[b].astype(np.int32), P)
        assert o["grid"].sum() > 100
        assert np.array_equal(r["occupancy_grid"][b].cpu().numpy(), o["grid"])
        assert np.array_equal(r["counts"][b].cpu().numpy().astype(np.uint32), o["counts"])
        assert np.array_equal(r["depth"][b].cpu().numpy().view(np.uint32), o["depth"].view(np.uint32))
        assert np.array_equal(r["points"][b].cpu().numpy(), o["points"])
