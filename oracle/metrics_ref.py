"""CPU oracle for the evaluation metrics (TEST INFRASTRUCTURE ONLY) — the step after the hot path
(SURVEY.md §8f #2).  Restates, function by function:
  compute_scale_and_shift   /root/reference/SOccDPT/loss/ssi_loss.py:5-32
  compute_masked_errors     /root/reference/SOccDPT/utils/__init__.py:109-158
  depth metrics of a batch  /root/reference/SOccDPT/utils/__init__.py:201-234 (evaluate_depth loop body)
  IoU of a batch            /root/reference/SOccDPT/utils/__init__.py:298-332 (evaluate_seg loop body)
Pinned against the reference's own functions by oracle/make_golden.py (tests/golden/metrics.npz)."""
import numpy as np
import torch


def compute_scale_and_shift(prediction, target, mask):
    a_00 = torch.sum(mask * prediction * prediction, (1, 2))
    a_01 = torch.sum(mask * prediction, (1, 2))
    a_11 = torch.sum(mask, (1, 2))
    b_0 = torch.sum(mask * prediction * target, (1, 2))
    b_1 = torch.sum(mask * target, (1, 2))
    x_0 = torch.zeros_like(b_0)
    x_1 = torch.zeros_like(b_1)
    det = a_00 * a_11 - a_01 * a_01
    valid = det.nonzero()
    x_0[valid] = (a_11[valid] * b_0[valid] - a_01[valid] * b_1[valid]) / det[valid]
    x_1[valid] = (-a_01[valid] * b_0[valid] + a_00[valid] * b_1[valid]) / det[valid]
    return x_0, x_1


def compute_masked_errors(gt, pred, mask):
    g, p = gt[mask], pred[mask]
    with np.errstate(all="ignore"):
        thresh = np.maximum(g / p, p / g)
        a1, a2, a3 = (thresh < 1.25).mean(), (thresh < 1.25 ** 2).mean(), (thresh < 1.25 ** 3).mean()
        rmse = np.sqrt(((g - p) ** 2).mean())
        rmse_log = np.sqrt(((np.log(g) - np.log(p)) ** 2).mean())
        abs_rel = np.mean(np.abs(g - p) / g)
        sq_rel = np.mean(((g - p) ** 2) / g)
    z = lambda v: 0 if (np.isinf(v) or np.isnan(v)) else v  # noqa: E731
    n = lambda v: 0 if np.isnan(v) else v  # noqa: E731
    return z(abs_rel), z(sq_rel), z(rmse), z(rmse_log), n(a1), n(a2), n(a3)


def depth_metrics_batch(y_pred, y, mask):
    """y_pred, y [B,H,W] f32, mask [B,H,W] bool -> (abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3, scale[B], shift[B])."""
    scale, shift = compute_scale_and_shift(y_pred, y, mask)
    ssi = scale.view(-1, 1, 1) * y_pred + shift.view(-1, 1, 1)
    m = compute_masked_errors(y.numpy(), ssi.numpy(), mask.numpy())
    return m + (scale.numpy(), shift.numpy())


def iou_batch(y_pred, y):
    """y_pred, y [B,C,H,W] -> per-image IoU [B] (mean over classes of inter / (union + 1e-7), threshold 0.5)."""
    C = y_pred.shape[1]
    iou = 0.0
    for c in range(C):
        pm, ym = y_pred[:, c] > 0.5, y[:, c] > 0.5
        inter = torch.logical_and(pm, ym).sum(dim=(1, 2))
        union = torch.logical_or(pm, ym).sum(dim=(1, 2))
        iou = iou + inter / (union + 1e-7)
    return (iou / C).numpy()
