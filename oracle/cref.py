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
    src = os.path.join(_HERE, "projection_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
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
