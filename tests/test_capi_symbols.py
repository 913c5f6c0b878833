"""CPU: libsoccdpt_hip.so loads and exports every symbol include/soccdpt_hip.h declares
(no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(REPO, "include", "soccdpt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(soccdpt_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported():
    so = os.path.join(REPO, "soccdpt_amd", "libsoccdpt_hip.so")
    if not os.path.exists(so):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(so)
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/soccdpt_hip.h but not exported"
    lib.soccdpt_abi_version.restype = ctypes.c_int
    assert lib.soccdpt_abi_version() == 1


def test_binding_struct_matches_header():
    from soccdpt_amd.lib import SoccdptConfig
    # 9 int32 + 4 f32 + 3 int32 + 9 f32 + 27 f32
    assert ctypes.sizeof(SoccdptConfig) == 4 * (9 + 4 + 3 + 9 + 27)


def test_no_cpu_fallback():
    """Without a GPU the product path must raise, not fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from soccdpt_amd.lib import Engine, make_config
    from oracle import soccdpt_ref as R
    cam, cfg = R.Camera(), R.ProjConfig()
    c = make_config("swin2t16_256", 3, 256, True, True, cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy,
                    cfg.grid_size, cfg.occupancy_shape(), cfg.pc_scale, cfg.pc_shift, cfg.correction_angle)
    with pytest.raises(RuntimeError):
        Engine(c, torch.device("cuda:0"))
