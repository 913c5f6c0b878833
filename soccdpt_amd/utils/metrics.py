"""Evaluation metrics on the GPU: same definitions and call shapes as the reference's
`evaluate_depth` / `evaluate_seg` loop bodies (/root/reference/SOccDPT/utils/__init__.py:161-332), computed by the
reduction kernels of libsoccdpt_hip.so without leaving HBM."""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from ..lib import _ptr, _stream_ptr, load_library

DEPTH_KEYS = ("abs_rel", "sq_rel", "rmse", "rmse_log", "a1", "a2", "a3")


def _scratch(B: int, C: int, device) -> torch.Tensor:
    n = load_library().soccdpt_metrics_scratch_bytes(B, C)
    return torch.empty(int(n), dtype=torch.uint8, device=device)


def depth_metrics(y_pred: torch.Tensor, y: torch.Tensor, mask: torch.Tensor) -> Dict[str, torch.Tensor]:
    """y_pred, y [B,H,W] f32, mask [B,H,W] bool (cuda) -> dict of 0-dim device tensors + 'scale'/'shift' [B].
    A prediction at another resolution is bicubic-resized to the ground truth first, like the reference
    (utils/__init__.py:207-212; that resize is torch plumbing, not part of the hot path)."""
    if y_pred.dim() == 2:
        y_pred = y_pred.unsqueeze(0)
    if y_pred.shape[-2:] != y.shape[-2:]:
        y_pred = F.interpolate(y_pred.unsqueeze(1), size=y.shape[-2:], mode="bicubic", align_corners=False)[:, 0]
    L = load_library()
    B = y.shape[0]
    npix = y.shape[1] * y.shape[2]
    p = y_pred.detach().to(torch.float32).contiguous()
    t = y.detach().to(torch.float32).contiguous()
    m = mask.detach().to(torch.uint8).contiguous()
    out = torch.empty(7 + 2 * B, dtype=torch.float32, device=y.device)
    sc = _scratch(B, 1, y.device)
    with torch.cuda.device(y.device):
        rc = L.soccdpt_metrics_depth(_ptr(p), _ptr(t), _ptr(m), B, npix, _ptr(out), _ptr(sc), _stream_ptr(y.device))
    if rc != 0:
        raise RuntimeError("soccdpt_metrics_depth failed: " + L.soccdpt_last_error(None).decode())
    res = {k: out[i] for i, k in enumerate(DEPTH_KEYS)}
    res["scale"] = out[7::2]
    res["shift"] = out[8::2]
    return res


def iou_metric(y_pred: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """y_pred, y [B,C,H,W] (cuda) -> per-image IoU [B] (threshold 0.5, mean over classes)."""
    if y_pred.dim() == 3:
        y_pred = y_pred.unsqueeze(0)
    if y_pred.shape[-2:] != y.shape[-2:]:
        y_pred = F.interpolate(y_pred, size=y.shape[-2:], mode="bicubic", align_corners=False)
    L = load_library()
    B, C = y.shape[0], y.shape[1]
    npix = y.shape[2] * y.shape[3]
    p = y_pred.detach().to(torch.float32).contiguous()
    t = y.detach().to(torch.float32).contiguous()
    out = torch.empty(B, dtype=torch.float32, device=y.device)
    sc = _scratch(B, C, y.device)
    with torch.cuda.device(y.device):
        rc = L.soccdpt_metrics_iou(_ptr(p), _ptr(t), B, C, npix, _ptr(out), _ptr(sc), _stream_ptr(y.device))
    if rc != 0:
        raise RuntimeError("soccdpt_metrics_iou failed: " + L.soccdpt_last_error(None).decode())
    return out


def evaluate_depth(net, dataloader, device, amp=False):
    """Same signature/return as the reference's evaluate_depth: mean over batches of the 7 depth metrics."""
    import numpy as np
    net.eval()
    acc = []
    for batch in dataloader:
        if len(batch) == 4:
            x, _, mask, y = batch
        else:
            x, _, mask, y, _, _ = batch
        x = x.to(device=device, dtype=torch.float32)
        y = y.to(device=device, dtype=torch.float32)
        mask = mask.to(device=device, dtype=torch.bool)
        y_pred = net(x)
        m = depth_metrics(y_pred, y, mask)
        acc.append(torch.stack([m[k] for k in DEPTH_KEYS]))
    vals = torch.stack(acc).cpu().numpy().astype(np.float64)          # [batches, 7]
    # per-metric filtering of the reference (utils/__init__.py:236-244): a non-finite abs_rel / sq_rel / rmse_log of ONE batch is
    # dropped from that metric's mean instead of poisoning it; rmse and a1..a3 are always kept
    out = []
    for i, k in enumerate(DEPTH_KEYS):
        col = vals[:, i]
        if k in ("abs_rel", "sq_rel", "rmse_log"):
            col = col[np.isfinite(col)]
        out.append(float(np.mean(col)))
    return tuple(out)


def evaluate_seg(net, dataloader, device, amp=False):
    """Same signature/return as the reference's evaluate_seg: mean IoU over images and batches."""
    net.eval()
    ious = []
    for batch in dataloader:
        if len(batch) == 4:
            x, _, _, y = batch
        else:
            x, _, _, _, _, y = batch
        x = x.to(device=device, dtype=torch.float32)
        y = y.to(device=device, dtype=torch.float32)
        ious.append(iou_metric(net(x), y))
    return float(torch.cat(ious).mean().item())
