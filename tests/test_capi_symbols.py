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
    header = open(os.path.join(REPO, "include", "soccdpt_hip.h")).read()
    declared_version = int(re.search(r"#define\s+SOCCDPT_ABI_VERSION\s+(\d+)", header).group(1))
    from soccdpt_amd import lib as binding
    # one version number in three places: the header, the compiled library, the ctypes binding (it had never moved)
    assert lib.soccdpt_abi_version() == declared_version == binding.ABI_VERSION >= 3


def test_binding_struct_sizes_match_the_library():
    """The ctypes mirrors of the public structs have exactly the size the LIBRARY was compiled with (soccdpt_sizeof): a field added to
    soccdpt_igemm_args / soccdpt_config without the binding following is caught here and at load_library()."""
    from soccdpt_amd.lib import CalibOptions, CalibReport, IgemmArgs, KernelStat, SoccdptConfig, load_library
    L = load_library()
    assert L.soccdpt_sizeof(3) == ctypes.sizeof(CalibReport)
    assert L.soccdpt_sizeof(4) == ctypes.sizeof(CalibOptions) == 20
    assert L.soccdpt_sizeof(0) == ctypes.sizeof(SoccdptConfig)
    assert L.soccdpt_sizeof(1) == ctypes.sizeof(IgemmArgs)
    assert L.soccdpt_sizeof(2) == ctypes.sizeof(KernelStat)
    assert L.soccdpt_sizeof(99) == 0


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


def test_device_code_has_no_packed_f32_ops(tmp_path):
    """The shipped gfx950 code objects carry no v_pk_*_f32 instructions (soccdpt_amd/csrc/Makefile NOPK; DESIGN.md section 4:
    a kernel dense in packed-f32 math returned wrong sums when it shared CUs with another stream's MFMA kernel)."""
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    so = os.path.join(REPO, "soccdpt_amd", "libsoccdpt_hip.so")
    if not os.path.exists(objdump) or not os.path.exists(so):
        pytest.skip("llvm-objdump or the built library is not present")
    local = str(tmp_path / "lib.so")
    shutil.copy(so, local)
    subprocess.run([objdump, "--offloading", local], check=True, capture_output=True, cwd=str(tmp_path))
    bundles = [f for f in os.listdir(tmp_path) if f.endswith("gfx950")]
    assert bundles, "no gfx950 code object found in libsoccdpt_hip.so"
    packed = 0
    for b in bundles:
        asm = subprocess.run([objdump, "-d", str(tmp_path / b)], check=True, capture_output=True, text=True).stdout
        packed += len(re.findall(r"\bv_pk_(?:mul|add|fma)_f32\b", asm))
    assert packed == 0, f"{packed} packed-f32 instructions in the device code"


def test_gn_apply_issues_its_partial_loads_back_to_back(tmp_path):
    """gn_apply_kernel adds up to twelve per-tile GroupNorm partials per thread.  Written as `tt < tps ? part[tt] : 0` every one of them compiled to
    s_cbranch_execz / global_load / s_waitcnt vmcnt(0): nine dependent L2 round trips in a launch that is one round of workgroups long (DESIGN.md 11.5,
    4.44 -> 4.30 ms on the hybrid forward once the addresses were clamped and the select moved behind the loads).  Guard: in every gn_apply instantiation of the
    shipped code object there is a run of at least 8 global_load_dwordx2 with no branch and no s_waitcnt vmcnt(0) between them."""
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    so = os.path.join(REPO, "soccdpt_amd", "libsoccdpt_hip.so")
    if not os.path.exists(objdump) or not os.path.exists(so):
        pytest.skip("llvm-objdump or the built library is not present")
    local = str(tmp_path / "lib.so")
    shutil.copy(so, local)
    subprocess.run([objdump, "--offloading", local], check=True, capture_output=True, cwd=str(tmp_path))
    bundles = [f for f in os.listdir(tmp_path) if f.endswith("gfx950")]
    assert bundles
    seen = 0
    for b in bundles:
        asm = subprocess.run([objdump, "-d", "--no-show-raw-insn", str(tmp_path / b)], check=True, capture_output=True, text=True).stdout
        name, run, best = None, 0, 0
        for line in asm.splitlines() + ["0 <end>:"]:
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                if name and "gn_apply_kernel" in name:
                    seen += 1
                    assert best >= 8, (name, best)
                name, run, best = m.group(1), 0, 0
                continue
            t = line.strip()
            if t.startswith("global_load_dwordx2"):
                run += 1
                best = max(best, run)
            elif t.startswith("s_cbranch") or (t.startswith("s_waitcnt") and "vmcnt(0)" in t):
                run = 0
    assert seen >= 8, seen          # 4 output formats x pixels-per-thread variants


def test_igemm_kernels_use_no_scratch(tmp_path):
    """Every igemm_kernel instantiation must keep its descriptor and address arrays in registers: a select between two per-thread arrays
    once demoted the whole IgemmDesc to private memory in 24 instantiations (472 bytes of scratch per lane, 4-5x slower launches) without
    any functional symptom.  Checked on the shipped gfx950 code object's kernel descriptors."""
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    so = os.path.join(REPO, "soccdpt_amd", "libsoccdpt_hip.so")
    if not os.path.exists(objdump) or not os.path.exists(readelf) or not os.path.exists(so):
        pytest.skip("llvm tools or the built library are not present")
    local = str(tmp_path / "lib.so")
    shutil.copy(so, local)
    subprocess.run([objdump, "--offloading", local], check=True, capture_output=True, cwd=str(tmp_path))
    bundles = [f for f in os.listdir(tmp_path) if f.endswith("gfx950")]
    assert bundles
    n = 0
    for b in bundles:
        notes = subprocess.run([readelf, "--notes", str(tmp_path / b)], check=True, capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            scratch = re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk)
            if name and scratch and "igemm_kernel" in name.group(1):
                n += 1
                assert int(scratch.group(1)) == 0, f"{name.group(1)} uses {scratch.group(1)} bytes of scratch per lane"
    assert n >= 40, f"only {n} igemm kernels found in the code object metadata"
