"""CPU: the metrics oracle against the golden values captured from the reference's own functions."""
import numpy as np

from oracle import metrics_ref as MR
from tests.golden_inputs import metrics_inputs


def test_metrics_oracle_matches_reference_golden(golden_dir):
    g = np.load(f"{golden_dir}/metrics.npz")
    pred, gt, mask, seg_pred, seg_gt = metrics_inputs(int(g["seed"]))
    m = MR.depth_metrics_batch(pred, gt, mask)
    np.testing.assert_allclose(np.array(m[:7], dtype=np.float64), g["depth"], rtol=1e-6)
    np.testing.assert_allclose(m[7], g["scale"], rtol=1e-6)
    np.testing.assert_allclose(m[8], g["shift"], rtol=1e-5)
    np.testing.assert_allclose(MR.iou_batch(seg_pred, seg_gt), g["iou"], rtol=1e-6)


def test_metrics_degenerate_cases():
    import torch
    pred, gt, mask, _, _ = metrics_inputs(B=2, H=16, W=16)
    empty = torch.zeros_like(mask)
    m = MR.depth_metrics_batch(pred, gt, empty)        # empty mask: det == 0 -> scale = shift = 0, all metrics 0
    assert all(v == 0 for v in m[:7]) and float(m[7].sum()) == 0.0
