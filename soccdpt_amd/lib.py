"""ctypes binding of libsoccdpt_hip.so (include/soccdpt_hip.h).

The HIP library is the product: there is no CPU or eager-PyTorch fallback.  If the
shared object is missing or a call fails, this module raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SOCCDPT_LIB_PATH") or os.path.join(_HERE, "libsoccdpt_hip.so")   # override: A/B of two builds in one GPU call (tools/ab_bench.sh)

ABI_VERSION = 5
BACKBONE_IDS = {"swin2t16_256": 0, "swin2b24_384": 1, "vitb_rn50_384": 2}
PREC_BF16 = 0
PREC_F32 = 1
PREC_F16 = 2
PREC_F16X3 = 3
PREC_MIXED = 4   # fp16 operands, x3 where the precision map says so (include/soccdpt_hip.h)
PREC_F16X2W = 5  # a precision-map value: fp16 activations, x3 weight pairs (two MFMAs per product)


class SoccdptConfig(ctypes.Structure):
    _fields_ = [
        ("abi_version", ctypes.c_int32),
        ("backbone", ctypes.c_int32),
        ("num_classes", ctypes.c_int32),
        ("features", ctypes.c_int32),
        ("sigmoid", ctypes.c_int32),
        ("compute_occ", ctypes.c_int32),
        ("precision", ctypes.c_int32),
        ("cam_width", ctypes.c_int32),
        ("cam_height", ctypes.c_int32),
        ("fx", ctypes.c_float), ("fy", ctypes.c_float), ("cx", ctypes.c_float), ("cy", ctypes.c_float),
        ("grid", ctypes.c_int32 * 3),
        ("occupancy_shape", ctypes.c_float * 3),
        ("pc_scale", ctypes.c_float * 3),
        ("pc_shift", ctypes.c_float * 3),
        ("rot", ctypes.c_float * 27),
    ]


class KernelStat(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 48), ("launches", ctypes.c_int32), ("ms", ctypes.c_double),
                ("flops", ctypes.c_double), ("bytes", ctypes.c_double)]


class CalibReport(ctypes.Structure):
    """soccdpt_calib_report (include/soccdpt_hip.h): what soccdpt_prec_calibrate measured."""
    _fields_ = [("n_groups", ctypes.c_int32), ("n_x3", ctypes.c_int32), ("n_x2w", ctypes.c_int32), ("n_x3_shipped", ctypes.c_int32), ("n_x2w_shipped", ctypes.c_int32), ("forwards", ctypes.c_int32),
                ("met_budget", ctypes.c_int32), ("shipped_met_budget", ctypes.c_int32), ("budget", ctypes.c_float),
                ("worst_calibrated", ctypes.c_float), ("worst_shipped", ctypes.c_float), ("worst_all_fp16", ctypes.c_float), ("worst_all_x3", ctypes.c_float),
                ("err_calibrated", ctypes.c_float * 7), ("err_shipped", ctypes.c_float * 7),
                ("cost_us_calibrated", ctypes.c_float), ("cost_us_shipped", ctypes.c_float),
                # ABI 5
                ("calib_frames", ctypes.c_int32), ("holdout_frames", ctypes.c_int32), ("met_headroom", ctypes.c_int32), ("met_holdout", ctypes.c_int32),
                ("headroom", ctypes.c_float), ("worst_holdout", ctypes.c_float), ("worst_holdout_shipped", ctypes.c_float), ("err_holdout", ctypes.c_float * 7),
                ("per_pixel_budget", ctypes.c_float), ("inv_p999_calibrated", ctypes.c_float), ("inv_max_calibrated", ctypes.c_float),
                ("inv_p999_holdout", ctypes.c_float), ("inv_max_holdout", ctypes.c_float), ("inv_p999_all_fp16", ctypes.c_float), ("inv_p999_all_x3", ctypes.c_float)]


class CalibOptions(ctypes.Structure):
    """soccdpt_calib_options (include/soccdpt_hip.h)."""
    _fields_ = [("struct_bytes", ctypes.c_int32), ("holdout", ctypes.c_int32), ("budget", ctypes.c_float), ("headroom", ctypes.c_float), ("per_pixel_p999", ctypes.c_float)]


CALIB_QUANTITIES = ("feat0", "feat1", "feat2", "feat3", "path1", "inv", "seg_logits")
PREC_SOURCE_NAMES = {0: "shipped", 1: "calibrated", 2: "edited", 3: "uncalibrated-all-x3", -1: "n/a"}


class IgemmArgs(ctypes.Structure):
    _fields_ = [
        ("x", ctypes.c_void_p), ("wt", ctypes.c_void_p),
        ("M", ctypes.c_int32), ("N", ctypes.c_int32), ("Cin", ctypes.c_int32), ("taps", ctypes.c_int32),
        ("ldx", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
        ("bias", ctypes.c_void_p), ("res1", ctypes.c_void_p), ("res2", ctypes.c_void_p),
        ("act", ctypes.c_int32), ("out_f32", ctypes.c_void_p), ("act_on_f32", ctypes.c_int32),
        ("out_bf16", ctypes.c_void_p), ("out_halo", ctypes.c_int32),
        ("dot_w", ctypes.c_void_p), ("dot_b", ctypes.c_float), ("out_dot", ctypes.c_void_p),
        ("tune", ctypes.c_int32), ("precision", ctypes.c_int32),
        ("splitk", ctypes.c_int32), ("sk_part", ctypes.c_void_p), ("sk_count", ctypes.c_void_p),
        ("sk_part_floats", ctypes.c_size_t), ("sk_count_words", ctypes.c_size_t),
        ("conv_general", ctypes.c_int32), ("stride", ctypes.c_int32), ("pad", ctypes.c_int32), ("in_halo", ctypes.c_int32),
        ("Hi", ctypes.c_int32), ("Wi", ctypes.c_int32), ("gather1", ctypes.c_int32),
        ("grp_rows", ctypes.c_int32), ("grp_off", ctypes.c_int32), ("seg2_k", ctypes.c_int32), ("seg2_off", ctypes.c_int32),
        ("grp_stride", ctypes.c_int64),
        ("gn_stats", ctypes.c_void_p), ("gn_part", ctypes.c_void_p), ("gn_count", ctypes.c_void_p),
        ("gn_cpg", ctypes.c_int32), ("gn_hw", ctypes.c_int32), ("gn_part_floats", ctypes.c_size_t), ("gn_count_words", ctypes.c_size_t),
        ("stamps", ctypes.c_void_p), ("sk_defer", ctypes.c_int32),
    ]


_lib = None


# Translation units (and the headers they share) that DEFINE OR SEQUENCE the forward path's kernels.  capi.cpp / internal.h (entry points,
# handle bookkeeping), the training step (train*), the criterion, optimiser, metrics, GT-occupancy and input-transform kernels launch
# nothing inside soccdpt_forward, so editing them does not invalidate PMC counters collected for the forward's kernels.
FORWARD_SOURCES = ("Makefile", "attention.hip", "attention_body.h", "attention_qkv.hip", "conv8p.hip", "depth_tail.hip", "elementwise.hip", "gelu.h", "half16.h", "hybrid.hip", "igemm.h",
                   "igemm.hip", "igemm_kernel.h", "kernels.h", "launch.h", "ln_body.h", "mlp_fused.hip", "model.cpp", "projection.hip", "resample.h",
                   "vit_attention.hip", "wino.hip")


def csrc_sha() -> str:
    """Short hash of the sources the FORWARD path's kernels are built from (FORWARD_SOURCES).  profiles/*_pmc_traffic.json carry it, so that
    bench.py -- and tests/test_profiles_current.py -- can tell whether the committed PMC counters still describe the current kernels."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(_HERE, "csrc")
    for name in FORWARD_SOURCES:
        h.update(name.encode())
        h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def load_library() -> ctypes.CDLL:
    """Load libsoccdpt_hip.so; fail loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C soccdpt_amd/csrc).  The SOccDPT MI355X path has no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    vp, ci, cs = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    L.soccdpt_abi_version.restype = ci
    L.soccdpt_create.argtypes = [ctypes.POINTER(SoccdptConfig), ctypes.POINTER(vp)]
    L.soccdpt_create.restype = ci
    L.soccdpt_destroy.argtypes = [vp]
    L.soccdpt_destroy.restype = None
    L.soccdpt_last_error.argtypes = [vp]
    L.soccdpt_last_error.restype = ctypes.c_char_p
    L.soccdpt_bind_weight.argtypes = [vp, ctypes.c_char_p, vp, ci, ctypes.POINTER(ctypes.c_int64), ci]
    L.soccdpt_bind_weight.restype = ci
    L.soccdpt_num_weights.argtypes = [vp]
    L.soccdpt_num_weights.restype = ci
    L.soccdpt_weight_key.argtypes = [vp, ci]
    L.soccdpt_weight_key.restype = ctypes.c_char_p
    L.soccdpt_prepared_bytes.argtypes = [vp]
    L.soccdpt_prepared_bytes.restype = cs
    L.soccdpt_workspace_bytes.argtypes = [vp, ci]
    L.soccdpt_workspace_bytes.restype = cs
    L.soccdpt_workspace_invalidate.argtypes = [vp]
    L.soccdpt_workspace_invalidate.restype = ci
    L.soccdpt_workspace_zero_fills.argtypes = [vp]
    L.soccdpt_workspace_zero_fills.restype = ci
    L.soccdpt_prepare.argtypes = [vp, vp, cs, vp]
    L.soccdpt_prepare.restype = ci
    L.soccdpt_forward.argtypes = [vp, vp, ci, vp, vp, vp, vp, vp, vp, cs, vp]
    L.soccdpt_forward.restype = ci
    L.soccdpt_network.argtypes = [vp, vp, ci, vp, vp, vp, cs, vp]
    L.soccdpt_network.restype = ci
    L.soccdpt_project.argtypes = [vp, vp, vp, ci, ci, ci, vp, vp, vp, vp, ci, vp]
    L.soccdpt_bind_grad.argtypes = [vp, ctypes.c_char_p, vp]
    L.soccdpt_train_set_amp.argtypes = [vp, ci]
    L.soccdpt_train_set_drop_path.argtypes = [vp, ctypes.c_float]
    L.soccdpt_train_unscale.argtypes = [vp, cs, ctypes.c_float, vp, vp]
    L.soccdpt_train_workspace_bytes.argtypes = [vp, ci]
    L.soccdpt_train_workspace_bytes.restype = cs
    L.soccdpt_train_forward.argtypes = [vp, vp, ci, vp, vp, vp, cs, ctypes.c_float, ctypes.c_uint32, vp]
    L.soccdpt_train_backward.argtypes = [vp, vp, ci, vp, vp, vp, cs, vp]
    L.soccdpt_train_backward_encoder.argtypes = [vp, ci, ctypes.POINTER(vp), vp, cs, vp]
    L.soccdpt_train_workspace_tensor.argtypes = [vp, ci, ctypes.c_char_p, ctypes.POINTER(cs), ctypes.POINTER(cs)]
    L.soccdpt_project.restype = ci
    L.soccdpt_project_backward_scratch_bytes.argtypes = [vp, ci, ci]
    L.soccdpt_project_backward_scratch_bytes.restype = cs
    L.soccdpt_project_backward.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, vp, vp, vp, cs, vp]
    L.soccdpt_project_backward.restype = ci
    L.soccdpt_occ_or.argtypes = [vp, vp, vp, ci, vp]
    L.soccdpt_occ_or.restype = ci
    L.soccdpt_occ_expand.argtypes = [vp, vp, ci, vp, vp]
    L.soccdpt_occ_expand.restype = ci
    L.soccdpt_occ_words.argtypes = [vp]
    L.soccdpt_occ_words.restype = cs
    L.soccdpt_last_launch_count.argtypes = [vp]
    L.soccdpt_last_launch_count.restype = ci
    L.soccdpt_launch_counter.argtypes = []
    L.soccdpt_launch_counter.restype = ctypes.c_ulonglong
    L.soccdpt_metrics_scratch_bytes.argtypes = [ci, ci]
    L.soccdpt_metrics_scratch_bytes.restype = cs
    L.soccdpt_metrics_depth.argtypes = [vp, vp, vp, ci, cs, vp, vp, vp]
    L.soccdpt_metrics_depth.restype = ci
    L.soccdpt_metrics_iou.argtypes = [vp, vp, ci, ci, cs, vp, vp, vp]
    L.soccdpt_metrics_iou.restype = ci
    cf = ctypes.c_float
    L.soccdpt_loss_scratch_bytes.argtypes = [ci, ci, ci, ci, ci]
    L.soccdpt_loss_scratch_bytes.restype = cs
    L.soccdpt_training_loss.argtypes = [ci, ci, ci, ci, ci, ci, ci, cf, cf, cf, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.soccdpt_training_loss.restype = ci
    cd = ctypes.c_double
    L.soccdpt_profile_sites.argtypes = [vp, ci]
    L.soccdpt_site_count.argtypes = [vp]
    ip = ctypes.POINTER(ctypes.c_int)
    L.soccdpt_site_get.argtypes = [vp, ci, ip, ip, ip, ip, ip, ip]
    L.soccdpt_tune_set.argtypes = [vp, ci, ci, ci, ci, ci]
    L.soccdpt_tune_clear.argtypes = [vp]
    L.soccdpt_gt_occupancy.argtypes = [ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, cf, vp, vp, vp, vp, vp, vp, vp]
    L.soccdpt_gt_occupancy.restype = ci
    L.soccdpt_input_transform_u8.argtypes = [vp, ci, ci, ci, ci, ci, ctypes.POINTER(cd), ctypes.POINTER(cd), vp, vp]
    L.soccdpt_input_transform_u8.restype = ci
    L.soccdpt_op_gn_finish.argtypes = [vp, vp, ci, ci, ci, ci, ci, ctypes.c_float, vp]
    L.soccdpt_op_gn_finish.restype = ci
    L.soccdpt_op_gn_apply.argtypes = [vp, vp, vp, ci, vp, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, ci, ci, ctypes.c_size_t, ci, ci, ci, ci, ctypes.c_float, vp]
    L.soccdpt_op_gn_apply.restype = ci
    L.soccdpt_op_mlp_ln.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]
    L.soccdpt_op_mlp_ln.restype = ci
    L.soccdpt_adam_step.argtypes = [ci, vp, vp, vp, vp, vp, cd, cd, cd, cd, cd, ci, vp]
    L.soccdpt_adam_step.restype = ci
    L.soccdpt_set_streams.argtypes = [vp, ci]
    L.soccdpt_set_streams.restype = ci
    L.soccdpt_set_graph.argtypes = [vp, ci]
    L.soccdpt_set_graph.restype = ci
    L.soccdpt_profile_enable.argtypes = [vp, ci]
    L.soccdpt_profile_enable.restype = ci
    L.soccdpt_profile_collect.argtypes = [vp, ctypes.POINTER(KernelStat), ci, ctypes.POINTER(ci)]
    L.soccdpt_profile_collect.restype = ci
    L.soccdpt_op_igemm.argtypes = [ctypes.POINTER(IgemmArgs), vp]
    L.soccdpt_op_igemm.restype = ci
    L.soccdpt_op_vit_attention.argtypes = [vp, vp, ci, ci, ci, ci, vp]
    L.soccdpt_op_vit_attention.restype = ci
    L.soccdpt_op_window_attention.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]
    L.soccdpt_op_window_attention.restype = ci
    L.soccdpt_op_window_attention_qkv.argtypes = [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp, vp]
    L.soccdpt_op_window_attention_qkv.restype = ci
    L.soccdpt_op_wino_weights.argtypes = [vp, vp, vp, ci, ci, ci, vp]
    L.soccdpt_op_wino_weights.restype = ci
    L.soccdpt_op_wino_conv.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp, vp, vp, ci, ci, ci, ci, vp, vp, ci, ci, ci, vp, vp]
    L.soccdpt_op_wino_conv.restype = ci
    L.soccdpt_op_wgrad_tn.argtypes = [vp, ctypes.c_long, vp, ctypes.c_long, ctypes.c_size_t, ci, ci, ci, ci, ci, vp, ctypes.c_size_t, vp, vp]
    L.soccdpt_op_wgrad_tn.restype = ci
    L.soccdpt_workspace_tensor.argtypes = [vp, ci, ctypes.c_char_p, ctypes.POINTER(cs), ctypes.POINTER(cs),
                                           ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.POINTER(ci)]
    L.soccdpt_workspace_tensor.restype = ci
    L.soccdpt_occ_zero.argtypes = [vp, ci, vp, vp]
    L.soccdpt_occ_zero.restype = ci
    L.soccdpt_occ_set.argtypes = [vp, vp, ci, vp, vp]
    L.soccdpt_occ_set.restype = ci
    L.soccdpt_sizeof.argtypes = [ci]
    L.soccdpt_sizeof.restype = cs
    L.soccdpt_prec_map_set.argtypes = [vp, ctypes.c_char_p, ci]
    L.soccdpt_prec_map_set.restype = ci
    L.soccdpt_prec_map_get.argtypes = [vp, ctypes.c_char_p, ci]
    L.soccdpt_prec_map_get.restype = ci
    L.soccdpt_prec_calibrate_scratch_bytes.argtypes = [vp, ci]
    L.soccdpt_prec_calibrate_scratch_bytes.restype = cs
    L.soccdpt_prec_calibrate.argtypes = [vp, vp, ci, ctypes.c_float, vp, cs, vp, cs, vp, cs, ctypes.POINTER(CalibReport), vp]
    L.soccdpt_prec_calibrate.restype = ci
    L.soccdpt_prec_calibrate_ex.argtypes = [vp, vp, ci, ctypes.POINTER(CalibOptions), vp, cs, vp, cs, vp, cs, ctypes.POINTER(CalibReport), vp]
    L.soccdpt_prec_calibrate_ex.restype = ci
    L.soccdpt_prec_map_source.argtypes = [vp]
    L.soccdpt_prec_map_source.restype = ci
    if L.soccdpt_abi_version() != ABI_VERSION:
        raise RuntimeError("libsoccdpt_hip.so ABI version mismatch; rebuild the library")
    # the ctypes mirrors of the public structs must have the layout the library was compiled with (include/soccdpt_hip.h)
    for which, cls in ((0, SoccdptConfig), (1, IgemmArgs), (2, KernelStat), (3, CalibReport), (4, CalibOptions)):
        if L.soccdpt_sizeof(which) != ctypes.sizeof(cls):
            raise RuntimeError(f"libsoccdpt_hip.so: sizeof mismatch for {cls.__name__}: library {L.soccdpt_sizeof(which)}, binding {ctypes.sizeof(cls)}")
    _lib = L
    return L


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    assert t.is_contiguous(), "device tensors handed to libsoccdpt_hip must be contiguous"
    return t.data_ptr()


def _stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def host_rotation_matrices(angles: Sequence[float]) -> np.ndarray:
    """Ra, Rb, Rc as 27 f32 values, computed on the host CPU with the same torch ops
    as rotate_points (/root/reference/SOccDPT/model/SOccDPT.py:74-111)."""
    a, b, c = [torch.deg2rad(torch.tensor(float(v), dtype=torch.float32)) for v in angles]
    ra = torch.tensor([[1, 0, 0], [0, torch.cos(a), -torch.sin(a)], [0, torch.sin(a), torch.cos(a)]], dtype=torch.float32)
    rb = torch.tensor([[torch.cos(b), 0, torch.sin(b)], [0, 1, 0], [-torch.sin(b), 0, torch.cos(b)]], dtype=torch.float32)
    rc = torch.tensor([[torch.cos(c), -torch.sin(c), 0], [torch.sin(c), torch.cos(c), 0], [0, 0, 1]], dtype=torch.float32)
    return torch.cat([ra.reshape(-1), rb.reshape(-1), rc.reshape(-1)]).numpy().astype(np.float32)


def make_config(backbone: str, num_classes: int, features: int, sigmoid: bool, compute_occ: bool,
                cam_width: int, cam_height: int, fx: float, fy: float, cx: float, cy: float,
                grid_size, occupancy_shape, pc_scale, pc_shift, correction_angle,
                precision: int = PREC_BF16) -> SoccdptConfig:
    if backbone not in BACKBONE_IDS:
        raise AssertionError(f"Backbone '{backbone}' not implemented on the MI355X path")
    cfg = SoccdptConfig()
    cfg.abi_version = ABI_VERSION
    cfg.backbone = BACKBONE_IDS[backbone]
    cfg.num_classes = int(num_classes)
    cfg.features = int(features)
    cfg.sigmoid = 1 if sigmoid else 0
    cfg.compute_occ = 1 if compute_occ else 0
    cfg.precision = int(precision)
    cfg.cam_width = int(cam_width)
    cfg.cam_height = int(cam_height)
    cfg.fx, cfg.fy, cfg.cx, cfg.cy = [float(np.float32(v)) for v in (fx, fy, cx, cy)]
    for i in range(3):
        cfg.grid[i] = int(grid_size[i])
        cfg.occupancy_shape[i] = float(np.float32(occupancy_shape[i]))
        cfg.pc_scale[i] = float(np.float32(pc_scale[i]))
        cfg.pc_shift[i] = float(np.float32(pc_shift[i]))
    rot = host_rotation_matrices(correction_angle)
    for i in range(27):
        cfg.rot[i] = float(rot[i])
    return cfg


def x3_encode(t: torch.Tensor) -> torch.Tensor:
    """float tensor -> the x3 split-fp16 operand bytes (csrc/half16.h; SOCCDPT_PREC_F16X3) as a flat fp16 tensor of 2 * numel on the same
    device: every aligned group of 8 elements is 8 hi values (RN fp16 of a) + 8 lo values (RN fp16 of (a - hi) * 2048), hi first in even
    groups, lo first in odd ones.  numel (and every row length) must be a multiple of 16."""
    f = t.detach().to(torch.float32).contiguous().reshape(-1, 8)
    assert f.shape[0] % 2 == 0, "x3 tensors are multiples of 16 elements"
    hi = f.clamp(-65504.0, 65504.0).to(torch.float16)
    lo = ((f - hi.float()) * 2048.0).clamp(-65504.0, 65504.0).to(torch.float16)
    odd = (torch.arange(f.shape[0], device=f.device) & 1).bool()[:, None]
    return torch.stack([torch.where(odd, lo, hi), torch.where(odd, hi, lo)], dim=1).reshape(-1)


def x3_decode(raw: torch.Tensor, shape) -> torch.Tensor:
    """Inverse of x3_encode: flat fp16 (or raw bytes) of an x3 tensor -> float64 tensor of `shape` (hi + lo / 2048, exact)."""
    r = raw.view(torch.float16) if raw.dtype != torch.float16 else raw
    r = r.reshape(-1, 2, 8).double()
    odd = (torch.arange(r.shape[0], device=r.device) & 1).bool()[:, None]
    hi = torch.where(odd, r[:, 1], r[:, 0])
    lo = torch.where(odd, r[:, 0], r[:, 1])
    return (hi + lo / 2048.0).reshape(shape)


class Engine:
    """One handle of libsoccdpt_hip.so bound to one GPU (one per rank)."""

    def __init__(self, cfg: SoccdptConfig, device: torch.device):
        if not torch.cuda.is_available():
            raise RuntimeError("soccdpt_amd needs a ROCm GPU (MI355X); there is no CPU fallback")
        self.L = load_library()
        self.cfg = cfg
        self.device = torch.device(device)
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            rc = self.L.soccdpt_create(ctypes.byref(cfg), ctypes.byref(self._h))
        if rc != 0:
            raise RuntimeError("soccdpt_create failed: " + self.L.soccdpt_last_error(None).decode())
        self._prepared: Optional[torch.Tensor] = None
        self._workspace: Optional[torch.Tensor] = None
        self._bound = {}

    def close(self):
        if getattr(self, "_h", None):
            self.L.soccdpt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise RuntimeError(f"{what} failed: {self.L.soccdpt_last_error(self._h).decode()}")

    # ---- weights ----
    def weight_keys(self):
        n = self.L.soccdpt_num_weights(self._h)
        return [self.L.soccdpt_weight_key(self._h, i).decode() for i in range(n)]

    def bind(self, key: str, t: torch.Tensor) -> bool:
        """Bind one device tensor under its state-dict key; False if the key is not consumed."""
        assert t.device.type == "cuda" and t.dtype == torch.float32 and t.is_contiguous(), key
        shape = (ctypes.c_int64 * t.dim())(*t.shape)
        rc = self.L.soccdpt_bind_weight(self._h, key.encode(), t.data_ptr(), 0, shape, t.dim())
        if rc == 2:
            return False
        self._check(rc, "soccdpt_bind_weight")
        self._bound[key] = t  # keep alive
        return True

    def prepare(self):
        nbytes = self.L.soccdpt_prepared_bytes(self._h)
        if self._prepared is None or self._prepared.numel() < nbytes:
            self._prepared = torch.zeros(max(nbytes, 16), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.L.soccdpt_prepare(self._h, self._prepared.data_ptr(), nbytes, _stream_ptr(self.device)),
                        "soccdpt_prepare")


    # ---- precision map (PREC_MIXED handles) ----
    def prec_map(self) -> dict:
        """{group: PREC_F16 | PREC_F16X3} in launch order (uniform-precision handles report their one format for every group)."""
        n = self.L.soccdpt_prec_map_get(self._h, None, 0)
        buf = ctypes.create_string_buffer(n + 1)
        self.L.soccdpt_prec_map_get(self._h, buf, n + 1)
        return {k: int(v) for k, v in (kv.split("=") for kv in buf.value.decode().split())}

    def prec_map_set(self, group: str, fmt: int) -> int:
        """Set one group ('s2.b0.attn'), a prefix ('s2.*') or everything ('*') to PREC_F16 / PREC_F16X3; re-prepares the weights."""
        n = self.L.soccdpt_prec_map_set(self._h, group.encode(), int(fmt))
        if n < 0:
            raise RuntimeError("soccdpt_prec_map_set failed: " + self.L.soccdpt_last_error(self._h).decode())
        if self._bound:
            self.prepare()
        return n

    def prec_map_source(self) -> int:
        """0 shipped map on the weights it was derived from, 1 calibrated on the bound weights, 2 edited, 3 other weights and no calibration yet (the
        library runs every group in x3 until one has run), -1 not a PREC_MIXED handle / not prepared (soccdpt_prec_map_source)."""
        return int(self.L.soccdpt_prec_map_source(self._h))

    def calibrate_precision(self, x: torch.Tensor, budget: float = 5e-4, holdout: int = None, headroom: float = 0.85, per_pixel_p999: float = None) -> dict:
        """soccdpt_prec_calibrate_ex: derive the precision map on the BOUND weights from the sample frames x [B,3,S,S] (against the library's exact-f32
        arithmetic on the same weights); the handle keeps the calibrated map, prepared.  The last `holdout` frames (default: a third of them when
        B >= 3) only verify the map (<= budget); the others select it and are held to headroom x budget.  per_pixel_p999: optional bound on the
        99.9th-percentile per-pixel relative error of the inverse depth.  Returns the report as a dict."""
        assert x.device == self.device and x.dtype == torch.float32 and x.is_contiguous()
        B = x.shape[0]
        if holdout is None:
            holdout = B // 3
        opt = CalibOptions(ctypes.sizeof(CalibOptions), int(holdout), float(budget), float(headroom), float(per_pixel_p999 or 0.0))
        nb = self.L.soccdpt_prec_calibrate_scratch_bytes(self._h, B)
        if nb == 0:
            raise RuntimeError("soccdpt_prec_calibrate: only SOCCDPT_PREC_MIXED handles have a precision map to calibrate")
        scratch = torch.empty(nb, dtype=torch.uint8, device=self.device)
        npre = self.L.soccdpt_prepared_bytes(self._h)
        if self._prepared is None or self._prepared.numel() < npre:
            self._prepared = torch.zeros(max(npre, 16), dtype=torch.uint8, device=self.device)
        ws = self.workspace(B)
        rep = CalibReport()
        with torch.cuda.device(self.device):
            self._check(self.L.soccdpt_prec_calibrate_ex(self._h, x.data_ptr(), B, ctypes.byref(opt), self._prepared.data_ptr(), self._prepared.numel(), ws.data_ptr(), ws.numel(),
                                                         scratch.data_ptr(), nb, ctypes.byref(rep), _stream_ptr(self.device)), "soccdpt_prec_calibrate_ex")
        out = {k: getattr(rep, k) for k, _ in CalibReport._fields_ if not k.startswith("err_")}
        out["err_holdout"] = dict(zip(CALIB_QUANTITIES, (float(v) for v in rep.err_holdout))) if rep.holdout_frames else None
        out["err_calibrated"] = dict(zip(CALIB_QUANTITIES, (float(v) for v in rep.err_calibrated)))
        out["err_shipped"] = dict(zip(CALIB_QUANTITIES, (float(v) for v in rep.err_shipped)))
        out["x3_groups"] = sorted(g for g, f in self.prec_map().items() if f == PREC_F16X3)
        out["x2w_groups"] = sorted(g for g, f in self.prec_map().items() if f == PREC_F16X2W)
        return out

    def workspace(self, B: int) -> torch.Tensor:
        """Per-forward scratch.  The library zero-fills it itself whenever (buffer, B, streams) changes (include/soccdpt_hip.h,
        workspace contract), so any allocation will do; the buffer only grows."""
        nbytes = self.L.soccdpt_workspace_bytes(self._h, B)
        if self._workspace is None or self._workspace.numel() < nbytes:
            self._workspace = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=self.device)
            self.L.soccdpt_workspace_invalidate(self._h)   # a freed buffer's address may be handed out again
        return self._workspace

    def workspace_zero_fills(self) -> int:
        return int(self.L.soccdpt_workspace_zero_fills(self._h))

    def occ_words(self) -> int:
        return int(self.L.soccdpt_occ_words(self._h))

    # ---- stage-level calls ----
    def project(self, inv: torch.Tensor, seg: torch.Tensor, inv_up, seg_up, points, occ_bits, clear_bits: bool = True):
        B, h, w = inv.shape
        with torch.cuda.device(self.device):
            self._check(self.L.soccdpt_project(self._h, _ptr(inv), _ptr(seg), B, h, w, _ptr(inv_up), _ptr(seg_up),
                                               _ptr(points), _ptr(occ_bits), 1 if clear_bits else 0,
                                               _stream_ptr(self.device)), "soccdpt_project")

    def project_backward(self, inv_up: torch.Tensor, d_inv_up, d_seg_up, d_points, in_h: int, in_w: int):
        """Gradients of the projection stage's differentiable outputs w.r.t. the network outputs (soccdpt_project_backward):
        -> (d_inv [B,in_h,in_w], d_seg [B,C,in_h,in_w])."""
        B = inv_up.shape[0]
        C = self.cfg.num_classes
        d_inv = torch.empty((B, in_h, in_w), device=self.device)
        d_seg = torch.empty((B, C, in_h, in_w), device=self.device)
        nbytes = self.L.soccdpt_project_backward_scratch_bytes(self._h, B, in_h)
        scratch = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=self.device)
        prep = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
        a, b, c = prep(d_inv_up), prep(d_seg_up), prep(d_points)
        with torch.cuda.device(self.device):
            self._check(self.L.soccdpt_project_backward(self._h, _ptr(inv_up.contiguous()), _ptr(a), _ptr(b), _ptr(c), B, in_h, in_w, _ptr(d_inv), _ptr(d_seg),
                                                        scratch.data_ptr(), scratch.numel(), _stream_ptr(self.device)), "soccdpt_project_backward")
        return d_inv, d_seg

    # ---- in-network tile tuning (tools/autotune_network.py) ----
    def profile_sites(self, on: bool):
        self._check(self.L.soccdpt_profile_sites(self._h, int(bool(on))), "soccdpt_profile_sites")

    def sites(self):
        out = []
        for i in range(self.L.soccdpt_site_count(self._h)):
            v = [ctypes.c_int() for _ in range(6)]
            self._check(self.L.soccdpt_site_get(self._h, i, *[ctypes.byref(x) for x in v]), "soccdpt_site_get")
            out.append(dict(site=f"site{i:03d}", M=v[0].value, N=v[1].value, K=v[2].value, taps=v[3].value, cfg=v[4].value, launches=v[5].value))
        return out

    def tune_set(self, M: int, N: int, K: int, taps: int, cfg: int):
        self._check(self.L.soccdpt_tune_set(self._h, M, N, K, taps, cfg), "soccdpt_tune_set")

    def tune_clear(self):
        self._check(self.L.soccdpt_tune_clear(self._h), "soccdpt_tune_clear")

    def occ_or(self, dst_bits: torch.Tensor, src_bits: torch.Tensor, n_sets: int):
        with torch.cuda.device(self.device):
            self._check(self.L.soccdpt_occ_or(self._h, _ptr(dst_bits), _ptr(src_bits), n_sets, _stream_ptr(self.device)),
                        "soccdpt_occ_or")

    def occ_expand(self, bits: torch.Tensor, B: int, occ: torch.Tensor):
        with torch.cuda.device(self.device):
            self._check(self.L.soccdpt_occ_expand(self._h, _ptr(bits), B, _ptr(occ), _stream_ptr(self.device)),
                        "soccdpt_occ_expand")

    def occ_zero(self, B: int, occ: torch.Tensor):
        with torch.cuda.device(self.device):
            self._check(self.L.soccdpt_occ_zero(self._h, B, _ptr(occ), _stream_ptr(self.device)), "soccdpt_occ_zero")

    def occ_set(self, bits: torch.Tensor, B: int, occ: torch.Tensor):
        with torch.cuda.device(self.device):
            self._check(self.L.soccdpt_occ_set(self._h, _ptr(bits), B, _ptr(occ), _stream_ptr(self.device)), "soccdpt_occ_set")

    def network(self, x: torch.Tensor, inv256: torch.Tensor, seg256: torch.Tensor):
        B = x.shape[0]
        ws = self.workspace(B)
        with torch.cuda.device(self.device):
            self._check(self.L.soccdpt_network(self._h, _ptr(x), B, _ptr(inv256), _ptr(seg256), ws.data_ptr(), ws.numel(),
                                               _stream_ptr(self.device)), "soccdpt_network")

    # ---- training step (include/soccdpt_hip.h: soccdpt_train_*) ----
    def bind_grad(self, key: str, g: Optional[torch.Tensor]):
        """Gradient destination of one weight (written by train_backward); None freezes the weight."""
        if g is not None:
            assert g.device.type == "cuda" and g.dtype == torch.float32 and g.is_contiguous(), key
        self._check(self.L.soccdpt_bind_grad(self._h, key.encode(), _ptr(g)), "soccdpt_bind_grad")
        self._grads = getattr(self, "_grads", {})
        self._grads[key] = g  # keep alive

    def train_set_amp(self, mode):
        """0 / False: f32; 1 / True / "bf16": bf16 operands for the gradient GEMMs; 2 / "f16": fp16 operands (use a GradScaler);
        3 / "x3" / "f16x3": split-fp16 operands (three fp16 MFMAs per product): f32-grade gradients, no loss scaling needed."""
        code = {False: 0, True: 1, 0: 0, 1: 1, 2: 2, 3: 3, "bf16": 1, "f16": 2, "fp16": 2, "x3": 3, "f16x3": 3, None: 0}[mode]
        self._check(self.L.soccdpt_train_set_amp(self._h, code), "soccdpt_train_set_amp")

    def train_set_drop_path(self, rate: float):
        """Stochastic-depth rate of the Swin-V2 encoder's train-mode forward (timm drop_path_rate; 0 = off)."""
        self._check(self.L.soccdpt_train_set_drop_path(self._h, float(rate)), "soccdpt_train_set_drop_path")

    def train_unscale(self, grads: torch.Tensor, inv_scale: float, found_inf: torch.Tensor):
        assert grads.is_contiguous() and grads.dtype == torch.float32 and found_inf.dtype == torch.int32
        with torch.cuda.device(self.device):
            rc = self.L.soccdpt_train_unscale(grads.data_ptr(), grads.numel(), float(inv_scale), found_inf.data_ptr(), _stream_ptr(self.device))
        if rc != 0:
            raise RuntimeError("soccdpt_train_unscale failed: " + self.L.soccdpt_last_error(None).decode())

    def train_workspace(self, B: int) -> torch.Tensor:
        nbytes = self.L.soccdpt_train_workspace_bytes(self._h, B)
        cur = getattr(self, "_train_ws", None)
        if cur is None or cur.numel() < nbytes:
            self._train_ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=self.device)
        return self._train_ws

    def train_backward_encoder(self, B: int, d_feats):
        """Encoder backward alone from the gradients of the four hooked feature maps ([B * r^2, C] f32 each, finest first)."""
        ws = self.train_workspace(B)
        keep = [t.detach().to(torch.float32).contiguous() for t in d_feats]
        arr = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in keep])
        with torch.cuda.device(self.device):
            self._check(self.L.soccdpt_train_backward_encoder(self._h, B, arr, ws.data_ptr(), ws.numel(), _stream_ptr(self.device)),
                        "soccdpt_train_backward_encoder")

    def train_tensor(self, B: int, name: str, channels: int) -> torch.Tensor:
        """A saved activation / gradient of the training workspace as an f32 [pixels, channels] view (see soccdpt_train_workspace_tensor)."""
        off, n = ctypes.c_size_t(), ctypes.c_size_t()
        if self.L.soccdpt_train_workspace_tensor(self._h, B, name.encode(), ctypes.byref(off), ctypes.byref(n)) != 0:
            raise KeyError(name)
        ws = self.train_workspace(B)
        return ws[off.value: off.value + 4 * n.value].view(torch.float32).view(-1, channels)

    def train_forward(self, x: torch.Tensor, inv: torch.Tensor, seg: torch.Tensor, dropout_p: float = 0.1, seed: int = 0):
        B = x.shape[0]
        ws = self.train_workspace(B)
        with torch.cuda.device(self.device):
            self._check(self.L.soccdpt_train_forward(self._h, _ptr(x), B, _ptr(inv), _ptr(seg), ws.data_ptr(), ws.numel(), float(dropout_p),
                                                     int(seed) & 0xFFFFFFFF, _stream_ptr(self.device)), "soccdpt_train_forward")

    def train_backward(self, x: torch.Tensor, d_inv: torch.Tensor, d_seg: torch.Tensor):
        B = x.shape[0]
        ws = self.train_workspace(B)
        assert d_inv.is_contiguous() and d_seg.is_contiguous() and d_inv.dtype == torch.float32 and d_seg.dtype == torch.float32
        with torch.cuda.device(self.device):
            self._check(self.L.soccdpt_train_backward(self._h, _ptr(x), B, _ptr(d_inv), _ptr(d_seg), ws.data_ptr(), ws.numel(),
                                                      _stream_ptr(self.device)), "soccdpt_train_backward")

    def forward(self, x: torch.Tensor, inv_up, seg_up, points, occ, occ_bits):
        B = x.shape[0]
        ws = self.workspace(B)
        with torch.cuda.device(self.device):
            self._check(self.L.soccdpt_forward(self._h, _ptr(x), B, _ptr(inv_up), _ptr(seg_up), _ptr(points), _ptr(occ),
                                               _ptr(occ_bits), ws.data_ptr(), ws.numel(), _stream_ptr(self.device)),
                        "soccdpt_forward")

    def set_streams(self, n: int):
        """Deal each batch to n concurrent sub-batches (see soccdpt_set_streams); invalidates the workspace."""
        with torch.cuda.device(self.device):
            self._check(self.L.soccdpt_set_streams(self._h, int(n)), "soccdpt_set_streams")
        self._workspace = None

    def set_graph(self, on: bool = True):
        self._check(self.L.soccdpt_set_graph(self._h, 1 if on else 0), "soccdpt_set_graph")

    def profile_enable(self, on: bool = True):
        self._check(self.L.soccdpt_profile_enable(self._h, 1 if on else 0), "soccdpt_profile_enable")

    def profile_collect(self):
        """-> {family: dict(launches, ms, flops, bytes)} summed over everything recorded since enable/collect."""
        buf = (KernelStat * 64)()
        n = ctypes.c_int(0)
        self._check(self.L.soccdpt_profile_collect(self._h, buf, 64, ctypes.byref(n)), "soccdpt_profile_collect")
        return {buf[i].name.decode(): dict(launches=buf[i].launches, ms=buf[i].ms, flops=buf[i].flops, bytes=buf[i].bytes)
                for i in range(n.value)}

    def launch_count(self) -> int:
        return int(self.L.soccdpt_last_launch_count(self._h))


    def workspace_tensor(self, B: int, name: str):
        """View of a named intermediate of the last soccdpt_network(B) call (diagnostics / parity tests).
        Returns an NHWC float32 tensor [B,H,W,C] (halo stripped, bf16 widened)."""
        off, n = ctypes.c_size_t(), ctypes.c_size_t()
        kind, H, W, C = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        rc = self.L.soccdpt_workspace_tensor(self._h, B, name.encode(), ctypes.byref(off), ctypes.byref(n), ctypes.byref(kind),
                                             ctypes.byref(H), ctypes.byref(W), ctypes.byref(C))
        if rc == 2:
            raise KeyError(f"{name}: not materialised in this mode (the seg head's classifier is part of the convolution's launch; SOCCDPT_SEG_DOT3_OFF=1 restores the map)")
        if rc != 0:
            raise KeyError(name)
        raw = self._workspace[off.value:]
        if "@" in name:  # one concurrent sub-batch of a multi-stream layout: its frame count follows from the element count
            halo = kind.value in (2, 3, 5, 7)
            B = n.value // ((H.value + 2 * halo) * (W.value + 2 * halo) * C.value)
        if kind.value == 0:
            return raw[: n.value * 4].view(torch.float32).reshape(B, H.value, W.value, C.value).clone()
        if kind.value == 3:
            t = raw[: n.value * 4].view(torch.float32).reshape(B, H.value + 2, W.value + 2, C.value)[:, 1:-1, 1:-1]
            return t.clone()
        if kind.value == 7:   # zero-halo NHWC in the x3 split-fp16 format (SOCCDPT_PREC_F16X3)
            t = x3_decode(raw[: n.value * 4], (B, H.value + 2, W.value + 2, C.value))[:, 1:-1, 1:-1]
            return t.float()
        t = raw[: n.value * 2].view(torch.float16 if kind.value in (4, 5) else torch.bfloat16)
        if kind.value in (2, 5):
            t = t.reshape(B, H.value + 2, W.value + 2, C.value)[:, 1:-1, 1:-1]
        else:
            t = t.reshape(B, H.value, W.value, C.value)
        return t.float()


def op_igemm(x, wt, M, N, Cin, taps=1, ldx=0, H=0, W=0, bias=None, res1=None, res2=None, act=0, out_f32=None,
             act_on_f32=0, out_bf16=None, out_halo=0, dot_w=None, dot_b=0.0, out_dot=None, tune=-1, f32=0, precision=None, splitk=1, sk_part=None, sk_count=None,
             conv=None, gather1=0, grp_rows=0, grp_off=0, grp_stride=0, seg2_k=0, seg2_off=0, gn_stats=None, gn_part=None, gn_count=None, gn_cpg=0, gn_hw=0,
             stamps=None, sk_defer=0):
    """Kernel-level entry (tests): one implicit-GEMM launch on the current stream.  conv = dict(stride, pad, in_halo, Hi, Wi) selects
    the generalised convolution addressing."""
    L = load_library()
    c = conv or {}
    a = IgemmArgs(_ptr(x), _ptr(wt), M, N, Cin, taps, ldx, H, W, _ptr(bias), _ptr(res1), _ptr(res2), act, _ptr(out_f32),
                  act_on_f32, _ptr(out_bf16), out_halo, _ptr(dot_w), float(dot_b), _ptr(out_dot), tune,
                  int(precision) if precision is not None else (PREC_F32 if f32 else PREC_BF16), int(splitk), _ptr(sk_part), _ptr(sk_count),
                  0 if sk_part is None else sk_part.numel(), 0 if sk_count is None else sk_count.numel(),
                  1 if conv else 0, c.get("stride", 1), c.get("pad", 1), c.get("in_halo", 1), c.get("Hi", 0), c.get("Wi", 0), int(gather1),
                  int(grp_rows), int(grp_off), int(seg2_k), int(seg2_off), int(grp_stride),
                  _ptr(gn_stats), _ptr(gn_part), _ptr(gn_count), int(gn_cpg), int(gn_hw),
                  0 if gn_part is None else gn_part.numel(), 0 if gn_count is None else gn_count.numel(), _ptr(stamps), int(sk_defer))
    rc = L.soccdpt_op_igemm(ctypes.byref(a), _stream_ptr(x.device))
    if rc != 0:
        raise RuntimeError("soccdpt_op_igemm failed: " + L.soccdpt_last_error(None).decode())


def op_gn_finish(part, stats, B, tps, groups, hw, cpg, eps=1e-5):
    """Kernel-level entry (tests): per-tile GroupNorm partials -> {mean, rstd} (soccdpt_op_gn_finish) on the current stream."""
    L = load_library()
    rc = L.soccdpt_op_gn_finish(_ptr(part), _ptr(stats), int(B), int(tps), int(groups), int(hw), int(cpg), float(eps), _stream_ptr(part.device))
    if rc != 0:
        raise RuntimeError("soccdpt_op_gn_finish failed: " + L.soccdpt_last_error(None).decode())


def op_gn_apply(raw, stats, gamma, beta, hw, w, cpg, part=None, tps=0, raw2=None, stats2=None, part2=None, tps2=0, gamma2=None, beta2=None, res=None,
                out_f32=None, out_op=None, out_halo=None, out_format=PREC_F32, relu=True, eps=1e-5):
    """Kernel-level entry (tests): GroupNorm apply (+ shortcut) (+ ReLU) of raw [M][C] (soccdpt_op_gn_apply) on the current stream."""
    L = load_library()
    M, C = raw.shape
    rc = L.soccdpt_op_gn_apply(_ptr(raw), _ptr(stats), _ptr(part), int(tps), _ptr(gamma), _ptr(beta), _ptr(raw2), _ptr(stats2), _ptr(part2), int(tps2),
                               _ptr(gamma2), _ptr(beta2), _ptr(res), _ptr(out_f32), _ptr(out_op), _ptr(out_halo), int(out_format), 1 if relu else 0, M, int(hw),
                               int(w), int(C), int(cpg), float(eps), _stream_ptr(raw.device))
    if rc != 0:
        raise RuntimeError("soccdpt_op_gn_apply failed: " + L.soccdpt_last_error(None).decode())


def op_vit_attention(qkv, out, B, N, heads, precision=PREC_BF16):
    """Kernel-level entry (tests): softmax(q k^T / 8) v of one ViT block on the current stream."""
    L = load_library()
    rc = L.soccdpt_op_vit_attention(_ptr(qkv), _ptr(out), int(precision), B, N, heads, _stream_ptr(qkv.device))
    if rc != 0:
        raise RuntimeError("soccdpt_op_vit_attention failed: " + L.soccdpt_last_error(None).decode())


def op_input_transform_u8(frames: torch.Tensor, Hd: int, Wd: int, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)) -> torch.Tensor:
    """uint8 [B,Hs,Ws,3] device frames -> float32 [B,3,Hd,Wd] network input (soccdpt_input_transform_u8) on the current stream."""
    L = load_library()
    assert frames.dtype == torch.uint8 and frames.dim() == 4 and frames.shape[3] == 3 and frames.is_cuda
    B, Hs, Ws, _ = frames.shape
    out = torch.empty((B, 3, Hd, Wd), dtype=torch.float32, device=frames.device)
    m, s = (ctypes.c_double * 3)(*mean), (ctypes.c_double * 3)(*std)
    with torch.cuda.device(frames.device):
        rc = L.soccdpt_input_transform_u8(_ptr(frames), B, Hs, Ws, Hd, Wd, m, s, _ptr(out), _stream_ptr(frames.device))
    if rc != 0:
        raise RuntimeError("soccdpt_input_transform_u8 failed: " + L.soccdpt_last_error(None).decode())
    return out


def op_mlp_ln(x_op, x_f32, w1, b1, w2, b2, ln_g, ln_b, x_op_out=None, halo=None, precision=PREC_BF16, H=0, W=0):
    """Kernel-level entry (tests): x_f32 += LN(fc2(GELU(fc1(x_op)))) in one launch (soccdpt_op_mlp_ln) on the current stream."""
    L = load_library()
    M, C = x_op.shape
    rc = L.soccdpt_op_mlp_ln(_ptr(x_op), _ptr(x_f32), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(ln_g), _ptr(ln_b), _ptr(x_op_out), _ptr(halo),
                             int(precision), M, C, H, W, _stream_ptr(x_op.device))
    if rc != 0:
        raise RuntimeError("soccdpt_op_mlp_ln failed: " + L.soccdpt_last_error(None).decode())


def op_wgrad_tn(a, lda, b, ldb, K, Nout, C, taps=1, rp=0, precision=PREC_BF16, b_row0=0):
    """Kernel-level entry (tests): weight gradient out [Nout][taps * C] from operands as stored (include/soccdpt_hip.h soccdpt_op_wgrad_tn).  a, b: flat device
    tensors (bf16 / fp16, or x3 bytes from x3_encode); b_row0: element offset of row 0 of b inside its tensor (margins in front for taps == 9)."""
    L = load_library()
    out = torch.empty((Nout, taps * C), dtype=torch.float32, device=a.device)
    scratch = torch.empty((64 * Nout * taps * C,), dtype=torch.float32, device=a.device)
    es = 4 if precision == PREC_F16X3 else 2
    rc = L.soccdpt_op_wgrad_tn(_ptr(a), lda, b.data_ptr() + b_row0 * es, ldb, K, Nout, C, taps, rp, int(precision), _ptr(scratch), scratch.numel(), _ptr(out),
                               _stream_ptr(a.device))
    if rc != 0:
        raise RuntimeError("soccdpt_op_wgrad_tn failed: " + L.soccdpt_last_error(None).decode())
    return out


def op_window_attention(qkv, cpb_table, scale, out, B, res, ws, shift, heads, precision=PREC_BF16):
    L = load_library()
    nt = (ws * ws + 31) // 32
    scratch = torch.empty((heads * nt * nt * 1024,), dtype=torch.float32, device=qkv.device)
    rc = L.soccdpt_op_window_attention(_ptr(qkv), _ptr(cpb_table), _ptr(scale), _ptr(out), _ptr(scratch), B, res, ws, shift,
                                       heads, int(precision), _stream_ptr(qkv.device))
    if rc != 0:
        raise RuntimeError("soccdpt_op_window_attention failed: " + L.soccdpt_last_error(None).decode())


def op_window_attention_qkv(x, wqkv, qkv_bias, cpb_table, scale, out, B, res, ws, shift, heads, precision=PREC_F16, out_x3=False, stamps=None):
    """soccdpt_op_window_attention_qkv: one Swin-V2 block's qkv projection + window attention as ONE launch (csrc/attention_qkv.hip)."""
    L = load_library()
    nt = (ws * ws + 31) // 32
    scratch = torch.empty((heads * nt * nt * 1024,), dtype=torch.float32, device=x.device)
    rc = L.soccdpt_op_window_attention_qkv(_ptr(x), _ptr(wqkv), _ptr(qkv_bias), _ptr(cpb_table), _ptr(scale), _ptr(out), _ptr(scratch), B, res, ws, shift,
                                           heads, int(precision), 1 if out_x3 else 0, _ptr(stamps), _stream_ptr(x.device))
    if rc != 0:
        raise RuntimeError("soccdpt_op_window_attention_qkv failed: " + L.soccdpt_last_error(None).decode())


def op_wino_weights(w, scale=None, precision=PREC_F16):
    """soccdpt_op_wino_weights: [N][C][3][3] f32 -> the Winograd-domain weights U = G g G^T, 16-bit [C/32][16][N][32] (csrc/wino.hip)."""
    L = load_library()
    N, C = w.shape[0], w.shape[1]
    u = torch.empty((16 * N * C,), dtype=torch.float16 if precision == PREC_F16 else torch.bfloat16, device=w.device)
    rc = L.soccdpt_op_wino_weights(_ptr(w), _ptr(scale), _ptr(u), N, C, int(precision), _stream_ptr(w.device))
    if rc != 0:
        raise RuntimeError("soccdpt_op_wino_weights failed: " + L.soccdpt_last_error(None).decode())
    return u


def op_wino_conv(x_halo, u, B, H, W, C, N, bias=None, res1=None, res2=None, res2_hw=(0, 0), relu=False, act_on_f32=False, out_f32=None, out_op=None, out_halo=False,
                 out_x3=False, precision=PREC_F16, stamps=None):
    """soccdpt_op_wino_conv: 3x3 stride-1 convolution over a zero-halo NHWC image in the Winograd F(2x2, 3x3) form, igemm's epilogue options."""
    L = load_library()
    rc = L.soccdpt_op_wino_conv(_ptr(x_halo), _ptr(u), B, H, W, C, N, _ptr(bias), _ptr(res1), _ptr(res2), int(res2_hw[0]), int(res2_hw[1]), 1 if relu else 0,
                                1 if act_on_f32 else 0, _ptr(out_f32), _ptr(out_op), 1 if out_halo else 0, 1 if out_x3 else 0, int(precision), _ptr(stamps), _stream_ptr(x_halo.device))
    if rc != 0:
        raise RuntimeError("soccdpt_op_wino_conv failed: " + L.soccdpt_last_error(None).decode())
