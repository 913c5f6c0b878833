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
    assert m.family(sig % ("64, 64, 64, 2, 4, 4, 16", "soccdpt::x3_t", "false")) == "igemm_x3_64x64x32_s4_w8"              # the x3 tiles of the mixed-precision forward
    assert m.family(sig % ("32, 64, 128, 2, 4, 3, 16", "soccdpt::x3_t", "false")) == "igemm_x3_32x64x64_s3_w8"
    assert m.family(sig % ("64, 64, 64, 2, 4, 4, 16", "soccdpt::f16_t", "false")) == "igemm_f16_64x64x64_s4_w8"
    sig7 = "void soccdpt::igemm_kernel<soccdpt::Cfg<128, 128, 64, 2, 4, 2, 16>, soccdpt::f16_t, false, false, false, false, %s>(soccdpt::IgemmDesc, int, int, int)"
    assert m.family(sig7 % "true") == "igemm_f16_128x128x64_s2_w8_dot3"           # the seg head's convolution with the classifier in its epilogue
    assert m.family(sig7 % "false") == "igemm_f16_128x128x64_s2_w8"
    sig16 = "void soccdpt::igemm_kernel<soccdpt::Cfg<256, 256, 64, 4, 4, 2, 16>, soccdpt::f16_t, false, false, false, false, true>(soccdpt::IgemmDesc, int, int, int)"
    assert m.family(sig16) == "igemm_f16_256x256x64_s2_w16_dot3"                    # round 5: the seg head on the 16-wave tile
    assert m.family(sig % ("64, 64, 64, 2, 4, 3, 16", "soccdpt::x2w_t", "false")) == "igemm_x2w_64x64x64_s3_w8"              # round 5: one-sided split launches
    assert m.family(sig % ("32, 64, 64, 2, 2, 4, 16", "soccdpt::x2w_t", "false")) == "igemm_x2w_32x64x64_s4"
    assert m.family("void soccdpt::igemm_kernel<soccdpt::Cfg<64, 128, 32, 2, 2, 3, 16>, soccdpt::x2w_t, true, false, false, false, false>(soccdpt::IgemmDesc, int, int, int)") == "igemm_x2w_64x128x32_s3_ln"
    assert m.family("void soccdpt::(anonymous namespace)::window_attention_qkv_kernel<16, true, 96, true, false>(unsigned short const*, void const*, float const*, float const*, float const*, unsigned short*, int, int, int, int, unsigned long long*)") == "window_attention_qkv"   # round 6
    assert m.family("void soccdpt::window_attention_flash_kernel<16, true, 1>(unsigned short const*, float const*, float const*, unsigned short*, int, int, int, int)") == "window_attention"
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


def _r04_lines():
    out = []
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "r04*_bench*.json"))):
        txt = open(f).read().strip()
        if txt.startswith("{") and '"metric"' in txt:
            out.append((os.path.basename(f), json.loads(txt.splitlines()[-1])))
    return out


def test_round4_bench_lines_carry_the_required_objects():
    """every committed bench line of this round has its roofline; forward lines at N = 1 name the arithmetic of `value`, carry the live
    tolerance measurement and (the full runs) the B = 1 latency the paper's 47 Hz is compared with; every training-step line -- the amp ones too --
    has roofline AND cpu_baseline."""
    lines = _r04_lines()
    assert lines, "no profiles/r04*_bench*.json"
    for name, d in lines:
        if "under_rocprof" in name:
            continue
        assert d.get("roofline") and d["roofline"].get("frac") is not None, name
        if "training step" in d["metric"]:
            assert d.get("cpu_baseline") and d["cpu_baseline"]["value"] > 0, name
            assert "drop_path_rate" in d["config"], name
        else:
            assert d["tolerance"]["dtype_of_value"] in d["dtype"] or d["dtype"].startswith(d["tolerance"]["dtype_of_value"]), name
            if d["n_gpus"] == 1 and "latency_b1" in d:
                assert d["latency_b1"]["forwards"] == 50 and abs(d["x_paper_hz"] - d["latency_b1"]["hz"] / 47.0) < 0.02, name
    main = dict(lines).get("r04_bench.json")
    assert main is not None
    assert main["dtype"].startswith("mixed") and main["tolerance"]["value_meets_tolerance"] is True
    assert main["tolerance"]["worst_measured"]["mixed"] <= main["tolerance"]["bar_for_value"] == 5e-4
    assert "latency_b1" in main and "f16_operands" in main and "bf16_operands" in main and main["f16_operands"]["roofline"]["frac"] > 0


def _lines(tag):
    out = []
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", f"{tag}*_bench*.json"))):
        txt = open(f).read().strip()
        if txt.startswith("{") and '"metric"' in txt:
            out.append((os.path.basename(f), json.loads(txt.splitlines()[-1])))
    return out


def test_round5_bench_lines_say_what_they_are():
    """VERDICT r4 #6 / #1d / #2: every committed forward line of round 5 prices its dominant kernel against the guide's 16-bit peak as well as against the blended
    one (`frac_of_16bit_peak`, also per tile configuration), carries the whole-forward fraction, measured its tolerance at the benchmark's batch with the per-pixel
    maximum, and says which precision map `value` ran; every training line has roofline and cpu_baseline."""
    lines = _lines("r05")
    assert lines, "no profiles/r05*_bench*.json"
    for name, d in lines:
        if "under_rocprof" in name:
            continue
        r = d["roofline"]
        assert r and r.get("frac") is not None, name
        if "training step" in d["metric"]:
            assert d.get("cpu_baseline") and d["cpu_baseline"]["value"] > 0, name
            continue
        assert "cpu_baseline" in d, name
        if not d["dtype"].startswith("f32"):
            assert 0 < r["frac_of_16bit_peak"] <= r["frac"] + 1e-9, name
            for row in r.get("by_config", []):
                assert "frac_of_16bit_peak" in row and "frac" in row, (name, row)
        assert 0 < r["whole_forward_frac"] < 1, name
        if d["dtype"].startswith("mixed"):
            assert d["config"]["precision_map_source"] in ("shipped", "calibrated"), name
            if "measured" in d["tolerance"] and d["tolerance"]["measured"]:
                assert "inv_per_pixel_max" in d["tolerance"]["measured"]["mixed"], name
                assert str(d["config"]["batch_per_gpu"]) in d["tolerance"]["bar_note"], name
    main = dict(lines).get("r05_bench.json")
    assert main is not None and main["dtype"].startswith("mixed") and main["tolerance"]["value_meets_tolerance"] is True
    assert main["tolerance"]["worst_measured"]["mixed"] <= main["tolerance"]["bar_for_value"] == 5e-4
    assert main["config"]["precision_map_source"] == "shipped" and main["config"]["precision_map_x3_groups"]
    assert abs(main["roofline"]["frac_of_16bit_peak"] - main["roofline"]["achieved"] / 2500.0) < 1e-3
    cal = dict(lines).get("r05_bench_calibrated.json")
    assert cal is not None and cal["config"]["precision_map_source"] == "calibrated" and cal["config"]["precision_map_calibration"]["met_budget"] == 1


def test_round6_bench_lines_say_what_other_weights_get():
    """the committed round-6 forward lines price the dominant kernel against the guide's 16-bit peak in `frac` itself (the blended figure under its
    own key), carry what checkpoints the shipped map was NOT derived on get (uncalibrated, calibrated, held-out error inside the budget), the forwards-in-flight side field with a
    bit-identical check, the per-pixel reading of the tolerance, and an SQ stall-counter summary from the same tree sits beside each model's PMC traffic file."""
    lines = dict(_lines("r06"))
    assert lines, "no profiles/r06*_bench*.json"
    main = lines.get("r06_bench.json")
    assert main is not None and main["dtype"].startswith("mixed") and main["tolerance"]["value_meets_tolerance"] is True
    r = main["roofline"]
    assert r["peak"] == 2500.0 and abs(r["frac"] - r["achieved"] / 2500.0) < 1e-3 and r["frac_vs_blended_peak"] >= r["frac"] and r["blended_peak"] <= 2500.0
    for row in r["by_config"]:
        assert "frac" in row and "frac_vs_format_peak" in row
    ow = main["other_weights"]
    assert set(ow["sets"]) >= {"salt1", "trained_like"} and ow["budget"] == 5e-4
    for name, v in ow["sets"].items():
        assert v["uncalibrated"]["precision_map_source"] == "uncalibrated-all-x3" and v["uncalibrated"]["worst_vs_f32_mode"] < 2e-4, name
        c = v["calibrated"]
        assert c["precision_map_source"] == "calibrated" and c["met_budget"] and c["heldout_within_budget"] and c["heldout_worst"] <= 5e-4, name
        assert c["holdout_frames"] >= 1 and c["library_worst_calibration_frames"] <= 0.85 * 5e-4 * 1.001 and c["library_worst_holdout_frames"] <= 5e-4, name
        assert v["uncalibrated"]["value"] < c["value"] <= main["value"] * 1.02, name
    p = main["pipelined"]
    assert p["in_flight"] == 2 and p["bit_identical_to_sequential"] is True and p["value"] > main["value"]
    assert main["tolerance"]["reading"].startswith("relative L2") and "inv_p999" in main["tolerance"]["per_pixel"]["mixed"]
    assert main["config"]["weights"] == "salt0" and main["config"]["precision_map_source"] == "shipped"
    for w in ("salt1", "trained_like"):
        d = lines.get(f"r06_bench_weights_{w}.json")
        assert d is not None and d["config"]["weights"] == w and d["config"]["precision_map_source"] == "calibrated", w
        assert d["tolerance"]["worst_measured"]["mixed"] <= 5e-4, w
    from soccdpt_amd.lib import csrc_sha
    for sfx in ("", "_base384", "_hybrid384"):
        st = json.load(open(os.path.join(REPO, "profiles", f"r06{sfx}_pmc_stall.json")))
        assert st["csrc_sha"] == csrc_sha() and st["kernels"], sfx
        k = next(iter(st["kernels"].values()))
        assert 0.9 < k["wait_any"] + k["wait_inst_any"] + k["active_inst_any"] < 1.1
    amp = lines.get("r06_bench_train_step_amp_B3_enc50_patch50.json")
    assert amp is not None and amp["config"]["batch_per_gpu"] == 3 and amp["config"]["encoder_percentage"] == 0.5 and "bf16" in amp["dtype"] and amp["roofline"] and amp["cpu_baseline"]
