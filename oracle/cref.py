"""ctypes binding of oracle/projection_ref.c (TEST INFRASTRUCTURE ONLY)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np
import torch

from . import soccdpt_ref as R

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libsoccdpt_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("projection_ref.c", "gt_occ_ref.c")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.soccdpt_ref_project.restype = ctypes.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def rot_matrices(angles) -> np.ndarray:
    """27 floats: Ra, Rb, Rc exactly as the reference builds them on the CPU."""
    return np.concatenate([m.numpy().reshape(-1) for m in R.rotation_matrices(angles)]).astype(np.float32)


def project(inv: torch.Tensor, seg: torch.Tensor, cam: R.Camera = R.Camera(), cfg: R.ProjConfig = R.ProjConfig(),
            want=("inv_up", "seg_up", "points", "occ_bits")):
    """Returns dict of numpy arrays (inv_up, seg_up, points, occ_bits)."""
    inv_n = np.ascontiguousarray(inv.numpy(), dtype=np.float32)
    seg_n = np.ascontiguousarray(seg.numpy(), dtype=np.float32)
    B, h, w = inv_n.shape
    C = seg_n.shape[1]
    Hc, Wc = cam.height, cam.width
    out = {}
    out["inv_up"] = np.empty((B, Hc, Wc), np.float32) if "inv_up" in want else None
    out["seg_up"] = np.empty((B, C, Hc, Wc), np.float32) if "seg_up" in want else None
    out["points"] = np.empty((B, Hc, Wc, 3), np.float32) if "points" in want else None
    g = cfg.grid_size
    nbits = g[0] * g[1] * g[2] * C
    out["occ_bits"] = np.zeros(((nbits + 31) // 32,), np.uint32) if "occ_bits" in want else None
    camv = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
    pcs = np.array(cfg.pc_scale, np.float32)
    pcb = np.array(cfg.pc_shift, np.float32)
    rot = rot_matrices(cfg.correction_angle)
    osh = cfg.occupancy_shape().astype(np.float32)
    grid = np.array(g, np.int32)
    rc = lib().soccdpt_ref_project(_p(inv_n), _p(seg_n), B, h, w, C, Hc, Wc, _p(camv), _p(pcs), _p(pcb), _p(rot),
                                   _p(osh), _p(grid), _p(out["inv_up"]), _p(out["seg_up"]), _p(out["points"]),
                                   _p(out["occ_bits"]))
    assert rc == 0
    return out


def pack_occ(occ_row: torch.Tensor) -> np.ndarray:
    """[gx,gy,gz,C] {0,1} float -> uint32 words, bit index == linear index."""
    flat = occ_row.reshape(-1).numpy() != 0
    pad = (-flat.size) % 32
    if pad:
        flat = np.concatenate([flat, np.zeros(pad, bool)])
    return np.packbits(flat, bitorder="little").view(np.uint32)


# ---------------------------------------------------------------------------------------------------------------------
# Ground-truth occupancy generator (oracle/gt_occ_ref.c; datasets/bdd_helper.py:238-530)
# ---------------------------------------------------------------------------------------------------------------------
class GtOccParams(ctypes.Structure):
    _fields_ = [("H", ctypes.c_int), ("W", ctypes.c_int), ("C", ctypes.c_int),
                ("fx", ctypes.c_double), ("fy", ctypes.c_double), ("cx", ctypes.c_double), ("cy", ctypes.c_double), ("base_focal", ctypes.c_double),
                ("pc_scale", ctypes.c_double * 3), ("pc_shift", ctypes.c_double * 3), ("rot", ctypes.c_double * 27),
                ("occ_shape", ctypes.c_float * 3), ("grid", ctypes.c_int * 3), ("threshold", ctypes.c_float)]


def gt_rot_matrices(angles=(7.0, 0.0, 0.0)) -> np.ndarray:
    """Ra^T, Rb^T, Rc^T (27 float64) built like rotate_points (datasets/bdd_helper.py:604-652): math.cos / math.sin of radians."""
    import math
    a, b, c = [math.radians(v) for v in angles]
    Ra = np.array([[1, 0, 0], [0, math.cos(a), -math.sin(a)], [0, math.sin(a), math.cos(a)]])
    Rb = np.array([[math.cos(b), 0, math.sin(b)], [0, 1, 0], [-math.sin(b), 0, math.cos(b)]])
    Rc = np.array([[math.cos(c), -math.sin(c), 0], [math.sin(c), math.cos(c), 0], [0, 0, 1]])
    return np.concatenate([Ra.T.reshape(-1), Rb.T.reshape(-1), Rc.T.reshape(-1)]).astype(np.float64)


def gt_params(H, W, C, fx, fy, cx, cy, grid_size=(256, 256, 32), scale=(2.0, 2.0, 0.666), pc_scale=(500.0, 2500.0, 200.0),
              pc_shift=(100.0, 40.0, 0.0), threshold=10, angles=(7.0, 0.0, 0.0), baseline=1.0 * 10**-2) -> GtOccParams:
    P = GtOccParams()
    P.H, P.W, P.C = H, W, C
    P.fx, P.fy, P.cx, P.cy = fx, fy, cx, cy
    P.base_focal = baseline * ((fx + fy) / 2.0)
    for k in range(3):
        P.pc_scale[k], P.pc_shift[k] = pc_scale[k], pc_shift[k]
        P.grid[k] = grid_size[k]
        P.occ_shape[k] = np.float32(float(grid_size[k] / scale[k]))
    for i, v in enumerate(gt_rot_matrices(angles)):
        P.rot[i] = v
    P.threshold = threshold
    return P


def gt_occupancy(disparity: np.ndarray, seg_class: np.ndarray, P: GtOccParams, want_points: bool = True):
    """disparity [H,W] f32, seg_class [H,W] i32 -> dict(depth f32 [H,W], points f64 [H*W,3], counts u32, grid bool [g0,g1,g2,C])."""
    L = lib()
    d = np.ascontiguousarray(disparity, dtype=np.float32)
    sc = np.ascontiguousarray(seg_class, dtype=np.int32)
    g = (P.grid[0], P.grid[1], P.grid[2], P.C)
    depth = np.empty((P.H, P.W), dtype=np.float32)
    pts = np.empty((P.H * P.W, 3), dtype=np.float64) if want_points else None
    counts = np.empty(g, dtype=np.uint32)
    grid = np.empty(g, dtype=np.uint8)
    L.gt_occupancy_ref(ctypes.byref(P), _p(d), _p(sc), _p(depth), _p(pts), _p(counts), _p(grid))
    return dict(depth=depth, points=pts, counts=counts, grid=grid.astype(bool))
