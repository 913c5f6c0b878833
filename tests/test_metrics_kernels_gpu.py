"""GPU: the HIP metric reductions (through the C ABI) against the CPU oracle / the reference's golden values."""
import numpy as np
import pytest
import torch

from oracle import metrics_ref as MR
from tests.golden_inputs import metrics_inputs

pytestmark = pytest.mark.gpu


def test_depth_and_iou_metrics_match_oracle(gpu_device, golden_dir):
    from soccdpt_amd.utils.metrics import DEPTH_KEYS, depth_metrics, iou_metric
    g = np.load(f"{golden_dir}/metrics.npz")
    pred, gt, mask, seg_pred, seg_gt = metrics_inputs(int(g["seed"]))
    m = depth_metrics(pred.to(gpu_device), gt.to(gpu_device), mask.to(gpu_device))
    got = np.array([float(m[k]) for k in DEPTH_KEYS])
    np.testing.assert_allclose(got, g["depth"], rtol=2e-4)          # f32 sums on the CPU vs f64 block sums here
    np.testing.assert_allclose(m["scale"].cpu().numpy(), g["scale"], rtol=2e-4)
    np.testing.assert_allclose(m["shift"].cpu().numpy(), g["shift"], rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(iou_metric(seg_pred.to(gpu_device), seg_gt.to(gpu_device)).cpu().numpy(), g["iou"], rtol=1e-6)


def test_metrics_full_resolution_and_edge_cases(gpu_device):
    from soccdpt_amd.utils.metrics import DEPTH_KEYS, depth_metrics, iou_metric
    pred, gt, mask, seg_pred, seg_gt = metrics_inputs(seed=5, B=2, H=1080, W=1920)   # camera resolution
    m = depth_metrics(pred.to(gpu_device), gt.to(gpu_device), mask.to(gpu_device))
    ref = MR.depth_metrics_batch(pred, gt, mask)
    np.testing.assert_allclose(np.array([float(m[k]) for k in DEPTH_KEYS]), np.array(ref[:7], dtype=np.float64), rtol=5e-4)
    np.testing.assert_allclose(iou_metric(seg_pred.to(gpu_device), seg_gt.to(gpu_device)).cpu().numpy(), MR.iou_batch(seg_pred, seg_gt), rtol=1e-6)
    # empty mask -> zeros (the reference maps NaN/inf to 0); empty union -> IoU 0
    z = depth_metrics(pred.to(gpu_device), gt.to(gpu_device), torch.zeros_like(mask).to(gpu_device))
    assert all(float(z[k]) == 0.0 for k in DEPTH_KEYS) and float(z["scale"].abs().sum()) == 0.0
    assert float(iou_metric(torch.zeros(1, 3, 8, 8, device=gpu_device), torch.zeros(1, 3, 8, 8, device=gpu_device))[0]) == 0.0
