"""CPU: the committed PMC counters describe the kernels in the tree.  bench.py fills roofline.traffic from the newest
profiles/r*_pmc_traffic.json and prints null when that file was collected from other forward-path sources; this test makes such a
commit fail instead of shipping a null: whoever edits a FORWARD_SOURCES file re-runs tools/collect_profiles.sh +
tools/install_profiles.py as the last step."""
import glob
import json
import os

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OTHER = ("_base384", "_hybrid384")


def _newest(suffix):
    c = sorted(glob.glob(os.path.join(REPO, "profiles", f"r*{suffix}_pmc_traffic.json")))
    if not suffix:
        c = [f for f in c if not any(t in os.path.basename(f) for t in OTHER)]
    return c[-1] if c else None


@pytest.mark.parametrize("suffix", ["", "_base384", "_hybrid384"])
def test_newest_pmc_traffic_matches_the_tree(suffix):
    from soccdpt_amd.lib import csrc_sha
    f = _newest(suffix)
    assert f is not None, f"no profiles/r*{suffix}_pmc_traffic.json"
    pj = json.load(open(f))
    assert pj.get("csrc_sha") == csrc_sha(), (f"{os.path.basename(f)} was collected from other forward-path kernel sources "
                                              f"({pj.get('csrc_sha')} != {csrc_sha()}): re-run tools/collect_profiles.sh / install_profiles.py")
    assert pj["kernels"], "empty PMC summary"


def test_newest_kernel_stats_is_beside_the_pmc_file():
    """The rocprofv3 --kernel-trace --stats summary the roofline's average launch time must agree with is from the same collection."""
    f = _newest("")
    tag = os.path.basename(f)[: -len("_pmc_traffic.json")]
    for name in ("_kernel_stats.csv", "_bench.json", "_bench_under_rocprof.json"):
        assert os.path.exists(os.path.join(REPO, "profiles", tag + name)), tag + name


def test_pmc_family_names_follow_the_kernel_templates():
    """tools/pmc_summary.py maps rocprofv3 kernel symbols to the family names bench.py's profiler reports (igemm.hip kCfgNames*); when the igemm template gained a
    seventh Cfg argument in round 3 the mapping silently stopped matching and roofline.traffic went null.  Pin the mapping."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("pmc_summary", os.path.join(REPO, "tools", "pmc_summary.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    sig = "void soccdpt::igemm_kernel<soccdpt::Cfg<%s>, %s, false, %s, false, false>(soccdpt::IgemmDesc, int, int, int)"
    assert m.family(sig % ("128, 128, 64, 2, 4, 2, 16", "unsigned short", "false")) == "igemm_bf16_128x128x64_s2_w8"
    assert m.family(sig % ("128, 128, 64, 2, 4, 2", "unsigned short", "false")) == "igemm_bf16_128x128x64_s2_w8"          # six-argument form (round 1 / 2 files)
    assert m.family(sig % ("32, 64, 128, 2, 2, 3, 16", "unsigned short", "true")) == "igemm_bf16_32x64x128_s3_splitk"
    assert m.family(sig % ("64, 64, 32, 2, 2, 4, 16", "soccdpt::f16_t", "false")) == "igemm_f16_64x64x32_s4"
    assert m.family(sig % ("128, 128, 64, 2, 2, 2, 16", "float", "false")) == "igemm_f32_128x128x32_s2"
    assert m.family(sig % ("128, 128, 128, 2, 4, 2, 16", "soccdpt::x3_t", "false")) == "igemm_x3_128x128x64_s2_w8"
    assert m.family(sig % ("128, 128, 64, 2, 4, 2, 32", "unsigned short", "false")) == "igemm_bf16_128x128x64_s2_m32"
    assert m.family("soccdpt::occ_expand_kernel(unsigned int const*, float*, unsigned long, int)") == "occ_expand"
    assert m.family("void soccdpt::project_rowsR_kernel<3, 256, 4, 7>(soccdpt::ProjParams, int, int)") == "project_voxelise"


@pytest.mark.parametrize("suffix", ["", "_base384", "_hybrid384"])
def test_newest_bench_line_carries_traffic(suffix):
    """The committed bench line of each model was produced AFTER its PMC summary was installed, so roofline.traffic (HBM bytes per launch of the dominant
    kernel family) and roofline_hbm.traffic are numbers, not null."""
    f = _newest(suffix)
    tag = os.path.basename(f)[: -len("_pmc_traffic.json")]
    line = open(os.path.join(REPO, "profiles", tag + "_bench.json")).read().strip().splitlines()[-1]
    d = json.loads(line)
    assert isinstance(d["roofline"]["traffic"], (int, float)) and d["roofline"]["traffic"] > 0, d["roofline"].get("traffic_note")
    assert d["roofline"].get("traffic_source") == os.path.basename(f)
    assert d["roofline_hbm"]["traffic"] > 0
