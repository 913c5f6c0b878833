"""Training criterion of the reference's loop on the GPU (first component of the training step):
`loss_depth_w * ScaleAndShiftInvariantLoss + loss_seg_w * BCELoss` on the camera-resolution predictions
(/root/reference/SOccDPT/scripts/train_SOccDPT.py:323-338,368-386; loss/ssi_loss.py), value AND gradient w.r.t. the network
outputs, by the fused kernels of libsoccdpt_hip.so (csrc/loss.hip).  No autograd graph, no full-resolution temporaries."""
from __future__ import annotations

from typing import Dict

import torch

from ..lib import _ptr, _stream_ptr, load_library


def training_loss(inv: torch.Tensor, seg: torch.Tensor, y_disp: torch.Tensor, mask_disp: torch.Tensor, y_seg: torch.Tensor,
                  mask_seg: torch.Tensor, loss_depth_w: float = 0.5, loss_seg_w: float = 0.5, alpha: float = 0.5,
                  compute_scale_and_shift: bool = True) -> Dict[str, torch.Tensor]:
    """inv [B,h,w], seg [B,C,h,w]: the network outputs (SOccDPT_V3.network(x)); y_disp [B,H,W], y_seg [B,C,H,W] and bool masks of
    the same shapes at camera resolution (all cuda).  Returns loss / loss_disp / loss_seg (0-dim), scale / shift [B], and
    d_inv [B,h,w], d_seg [B,C,h,w] = d loss / d network outputs."""
    assert inv.is_cuda, "the HIP path needs cuda tensors (no CPU fallback)"
    L = load_library()
    B, h, w = inv.shape
    C = seg.shape[1]
    H, W = y_disp.shape[-2:]
    assert tuple(seg.shape) == (B, C, h, w) and tuple(y_disp.shape) == (B, H, W) and tuple(y_seg.shape) == (B, C, H, W)
    assert tuple(mask_disp.shape) == (B, H, W) and tuple(mask_seg.shape) == (B, C, H, W)
    f = lambda t: t.detach().to(torch.float32).contiguous()
    u = lambda t: t.detach().to(torch.uint8).contiguous()
    inv_, seg_, yd, ys, md, ms = f(inv), f(seg), f(y_disp), f(y_seg), u(mask_disp), u(mask_seg)
    dev = inv.device
    out = torch.empty(3 + 2 * B, dtype=torch.float32, device=dev)
    d_inv = torch.empty_like(inv_)
    d_seg = torch.empty_like(seg_)
    scratch = torch.empty(int(L.soccdpt_loss_scratch_bytes(B, H, W, h, w)), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = L.soccdpt_training_loss(B, H, W, h, w, C, int(bool(compute_scale_and_shift)), float(alpha), float(loss_depth_w),
                                     float(loss_seg_w), _ptr(inv_), _ptr(seg_), _ptr(yd), _ptr(md), _ptr(ys), _ptr(ms), _ptr(out),
                                     _ptr(d_inv), _ptr(d_seg), _ptr(scratch), _stream_ptr(dev))
    if rc != 0:
        raise RuntimeError("soccdpt_training_loss failed: " + L.soccdpt_last_error(None).decode())
    return dict(loss=out[0], loss_disp=out[1], loss_seg=out[2], scale=out[3::2], shift=out[4::2], d_inv=d_inv, d_seg=d_seg)
