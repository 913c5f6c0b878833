"""Ground-truth occupancy generator on the GPU with the constructor of the reference's
`OccupancyProcessor` (/root/reference/SOccDPT/datasets/bdd_helper.py:238-285).  The reference processes one frame at a time in
numpy on the host (2 M points, ~0.5 s per 1080p frame); here a batch of disparity frames and class maps that already sit in HBM
becomes depth, the rotated point cloud and the thresholded counting grid in two launches (csrc/gt_occ.hip).

What stays on the host, outside this boundary: decoding images, `rgb_seg_to_class` (colour -> class id LUT, :10-25) and the
colourised point cloud / `occupancy_points` list used for visualisation (:320-345, :491-521)."""
from __future__ import annotations

import ctypes
import math
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from ..lib import _ptr, _stream_ptr, load_library


class OccupancyProcessor:
    def __init__(self, intrinsic_matrix, height: int, width: int, grid_size: Sequence[int], scale: Sequence[float], shift: Sequence[float],
                 pc_scale: Sequence[float], pc_shift: Sequence[float], point_count_threshold: float, class_2_color=None, color_2_class=None,
                 num_classes: int = 3, correction_angle: Sequence[float] = (7.0, 0.0, 0.0)) -> None:
        K = np.asarray(intrinsic_matrix, dtype=np.float64)
        self.height, self.width, self.num_classes = int(height), int(width), int(num_classes)
        self.class_2_color, self.color_2_class = class_2_color, color_2_class
        self.grid_size, self.scale, self.shift = tuple(int(g) for g in grid_size), tuple(scale), tuple(shift)
        self.pc_scale, self.pc_shift = tuple(float(v) for v in pc_scale), tuple(float(v) for v in pc_shift)
        self.point_count_threshold = float(point_count_threshold)
        self.occupancy_shape = np.array([float(self.grid_size[i] / self.scale[i]) for i in range(3)], dtype=np.float32)   # :264-270
        self.baseline = 1.0 * 10**-2                                                                                      # :274
        self.fx, self.fy, self.cx, self.cy = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])
        a, b, c = [math.radians(v) for v in correction_angle]                                                             # rotate_points, :604-652
        Ra = np.array([[1, 0, 0], [0, math.cos(a), -math.sin(a)], [0, math.sin(a), math.cos(a)]])
        Rb = np.array([[math.cos(b), 0, math.sin(b)], [0, 1, 0], [-math.sin(b), 0, math.cos(b)]])
        Rc = np.array([[math.cos(c), -math.sin(c), 0], [math.sin(c), math.cos(c), 0], [0, 0, 1]])
        self._rot = np.concatenate([Ra.T.reshape(-1), Rb.T.reshape(-1), Rc.T.reshape(-1)]).astype(np.float64)

    def process(self, disparity: torch.Tensor, seg_class: torch.Tensor, want_points: bool = True, want_depth: bool = True) -> Dict[str, Optional[torch.Tensor]]:
        """disparity [B,H,W] (or [H,W]) float, seg_class same shape integer class ids, both cuda ->
        depth [B,H,W] f32, points [B,H*W,3] f64, occupancy_grid [B,g0,g1,g2,C] bool, counts [B,g0,g1,g2,C] int32."""
        assert disparity.is_cuda, "the HIP path needs cuda tensors (no CPU fallback)"
        if disparity.dim() == 2:
            disparity, seg_class = disparity.unsqueeze(0), seg_class.unsqueeze(0)
        B, H, W = disparity.shape
        assert (H, W) == (self.height, self.width) and tuple(seg_class.shape) == (B, H, W)
        dev = disparity.device
        d = disparity.detach().to(torch.float32).contiguous()
        sc = seg_class.detach().to(torch.int32).contiguous()
        g = self.grid_size
        depth = torch.empty((B, H, W), dtype=torch.float32, device=dev) if want_depth else None
        pts = torch.empty((B, H * W, 3), dtype=torch.float64, device=dev) if want_points else None
        counts = torch.empty((B, g[0], g[1], g[2], self.num_classes), dtype=torch.int32, device=dev)
        occ = torch.empty((B, g[0], g[1], g[2], self.num_classes), dtype=torch.uint8, device=dev)
        arr = lambda vals, t: (t * len(vals))(*vals)
        intr = arr([self.fx, self.fy, self.cx, self.cy, self.baseline], ctypes.c_double)
        L = load_library()
        with torch.cuda.device(dev):
            rc = L.soccdpt_gt_occupancy(B, H, W, self.num_classes, intr, arr(self.pc_scale, ctypes.c_double), arr(self.pc_shift, ctypes.c_double),
                                        arr([float(v) for v in self._rot], ctypes.c_double), arr([float(v) for v in self.occupancy_shape], ctypes.c_float),
                                        arr(list(g), ctypes.c_int), self.point_count_threshold, _ptr(d), _ptr(sc), _ptr(depth), _ptr(pts),
                                        _ptr(counts), _ptr(occ), _stream_ptr(dev))
        if rc != 0:
            raise RuntimeError("soccdpt_gt_occupancy failed: " + L.soccdpt_last_error(None).decode())
        return dict(depth=depth, points=pts, occupancy_grid=occ.bool(), counts=counts)
