"""GPU: soccdpt_prec_calibrate -- the precision map of the default arithmetic derived on the weights actually bound.

The reference computes in fp32 whatever checkpoint BaseModel.load_net binds (/root/reference/SOccDPT/model/base_model.py:5-37,
model/SOccDPT.py:29-57,634-636).  The shipped maps of SOCCDPT_PREC_MIXED were fitted to ONE synthetic weight draw (salt 0); these tests bind
OTHER weights -- two more synthetic draws and a draw with trained-like statistics (LayerNorm gains spread over U(0.2, 3), a few 10x outlier
channels) -- and check, against the fp32 CPU oracle on frames the calibration never saw:
  * the library notices: soccdpt_prec_map_source() == 3, every group runs x3 operands (f32-grade) until a calibration has run, and the Python
    mirror prints the notice once;
  * how the SHIPPED map would have generalised is measured and reported by the calibration (round 5: on the second synthetic draw it leaves the
    class logits at 1.4e-3 -- outside the north star -- which is why the uncalibrated default is all-x3 and not the shipped map);
  * the CALIBRATED map keeps all seven quantities within the budget (5e-4) -- measured by the library against its own f32 mode on the
    calibration frames (held to 0.85 x budget) and on its own hold-out frames (<= budget), and re-measured here against the CPU oracle on frames
    neither of them saw: round 6 holds these to THE BUDGET (round 5 allowed 1.1 - 1.15 x: a calibrated map outside its budget is a failed calibration)."""
import os
import tempfile

import pytest
import torch

from oracle import soccdpt_ref as R

pytestmark = pytest.mark.gpu

QUANT = ("feat0", "feat1", "feat2", "feat3", "path1", "inv", "seg_logits")


def _rel_l2(a, b):
    return float((a - b).norm() / b.norm())


def _trained_like(sd, seed=5):
    from soccdpt_amd.utils.synth import trained_like
    return trained_like(sd, seed)


def _build(sd, model_type="dpt_swin2_tiny_256", dev="cuda:0"):
    from soccdpt_amd.lib import PREC_MIXED
    from soccdpt_amd.model.SOccDPT import SOccDPT_V3
    from soccdpt_amd.utils.synth import write_synth_calib
    calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
    m = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True, precision=PREC_MIXED, model_type=model_type)
    m.load_state_dict(sd, strict=False)
    return m.eval().to(dev)


def _oracle(sd, x, backbone):
    torch.set_num_threads(16)
    with torch.no_grad():
        layers = R.hybrid_encoder(sd, x) if backbone == "vitb_rn50_384" else R.swin_encoder(sd, x, R.ARCHS[backbone])
        o_inv, o_p1 = R.dpt_decoder(sd, layers)
        return layers, o_inv, o_p1, R.seg_logits(sd, o_p1)


def _errors_vs_oracle(m, x, dev, ora):
    layers, o_inv, o_p1, o_logits = ora
    inv, _ = m.network(x.to(dev))
    torch.cuda.synchronize()
    eng, B = m._engine(dev), x.shape[0]
    e = {f"feat{s}": _rel_l2(eng.workspace_tensor(B, f"feat{s}").cpu().permute(0, 3, 1, 2), layers[s]) for s in range(4)}
    e["path1"] = _rel_l2(eng.workspace_tensor(B, "path1").cpu().permute(0, 3, 1, 2), o_p1)
    e["inv"] = _rel_l2(inv.cpu(), o_inv)
    e["seg_logits"] = _rel_l2(eng.workspace_tensor(B, "seg_logits").cpu().permute(0, 3, 1, 2), o_logits)
    return e


def test_shipped_weights_are_recognised_and_recalibration_keeps_the_budget(gpu_device):
    """salt 0 = the draw the shipped map was derived on: source 'shipped', no warning; calibrating anyway returns a map that meets the budget
    and costs no more than the shipped one by the compiled-in cost table."""
    from soccdpt_amd.utils.synth import synth_input, synth_state_dict
    m = _build(synth_state_dict(alias_pretrained=True))
    assert m.precision_map_source() == "shipped"
    rep = m.calibrate_precision(synth_input(6, seed0=4).to(gpu_device), budget=5e-4)
    print("calibration on the shipped map's own weights:", {k: (f"{v:.3g}" if isinstance(v, float) else v) for k, v in rep.items() if not isinstance(v, (dict, list))})
    assert m.precision_map_source() == "calibrated"
    assert rep["met_budget"] == 1 and rep["shipped_met_budget"] == 1 and rep["worst_calibrated"] <= 0.85 * 5e-4 + 1e-9
    assert rep["calib_frames"] == 4 and rep["holdout_frames"] == 2 and rep["met_headroom"] == 1 and rep["met_holdout"] == 1 and rep["worst_holdout"] <= 5e-4
    assert rep["inv_p999_calibrated"] > 0 and rep["inv_max_calibrated"] >= rep["inv_p999_calibrated"]
    assert rep["worst_all_x3"] < 1e-4 < rep["worst_all_fp16"]
    assert rep["forwards"] >= rep["n_groups"] + 4


@pytest.mark.parametrize("case", ["salt1", "salt2", "trained_like"])
def test_other_weights_shipped_map_reported_calibrated_map_within_budget(gpu_device, case, capsys):
    from soccdpt_amd.utils.synth import synth_input, synth_state_dict
    sd = synth_state_dict(salt={"salt1": 1, "salt2": 2, "trained_like": 3}[case], alias_pretrained=True)
    if case == "trained_like":
        sd = _trained_like(sd)
    m = _build(sd)
    x_cal = synth_input(6, seed0=4)          # what the calibration sees: 4 frames select the map, 2 verify it inside the library
    x_test = synth_input(2, seed0=90)        # what it never saw
    ora = _oracle(sd, x_test, "swin2t16_256")
    e_safe = _errors_vs_oracle(m, x_test, gpu_device, ora)
    assert m.precision_map_source() == "uncalibrated-all-x3"
    assert "not the weights the shipped precision map" in capsys.readouterr().out
    pm = m._engine(gpu_device).prec_map()
    assert all(v == 3 for v in pm.values())
    rep = m.calibrate_precision(x_cal.to(gpu_device), budget=5e-4)
    assert m.precision_map_source() == "calibrated"
    e_cal = _errors_vs_oracle(m, x_test, gpu_device, ora)
    print(f"[{case}] before calibration (all groups x3), held-out frames vs fp32 CPU oracle: worst {max(e_safe.values()):.2e}", {k: f"{v:.2e}" for k, v in e_safe.items()})
    print(f"[{case}] calibrated map ({rep['n_x3']} of {rep['n_groups']} groups x3, shipped {rep['n_x3_shipped']}; {rep['forwards']} forwards; library-measured worst "
          f"{rep['worst_calibrated']:.2e}, shipped {rep['worst_shipped']:.2e}, all-fp16 {rep['worst_all_fp16']:.2e}): held-out worst {max(e_cal.values()):.2e}",
          {k: f"{v:.2e}" for k, v in e_cal.items()})
    print(f"[{case}] library hold-out frames: worst {rep['worst_holdout']:.2e}; per-pixel inv p99.9 {rep['inv_p999_holdout']:.2e} max {rep['inv_max_holdout']:.2e}; {rep['n_x2w']} x2w groups")
    assert rep["met_budget"] == 1 and rep["met_headroom"] == 1 and rep["met_holdout"] == 1
    assert rep["worst_calibrated"] <= 0.85 * 5e-4 + 1e-9 and rep["worst_holdout"] <= 5e-4
    assert max(e_safe.values()) <= 2e-4          # the uncalibrated default is parity-grade
    assert max(e_cal.values()) <= 5e-4           # THE budget, on frames no part of the calibration saw, against the fp32 CPU oracle
    assert rep["worst_all_x3"] < 2e-4 < rep["worst_all_fp16"] and rep["worst_shipped"] > 0     # reported, not bounded: the shipped map is not claimed for these weights


def test_calibrate_hybrid_384(gpu_device):
    """dpt_hybrid_384 (budget: the north star's 1e-3) on a second weight draw, B = 1."""
    from soccdpt_amd.utils.synth import synth_input, synth_state_dict
    sd = synth_state_dict("vitb_rn50_384", salt=1, alias_pretrained=True)
    m = _build(sd, "dpt_hybrid_384")
    x_cal, x_test = synth_input(3, size=384, seed0=4), synth_input(1, size=384, seed0=90)   # 2 frames select, 1 verifies
    ora = _oracle(sd, x_test, "vitb_rn50_384")
    e_safe = _errors_vs_oracle(m, x_test, gpu_device, ora)
    print("[hybrid salt1] before calibration (all groups x3; the attention core stays fp16):", {k: f"{v:.2e}" for k, v in e_safe.items()})
    assert max(e_safe.values()) <= 4e-4
    rep = m.calibrate_precision(x_cal.to(gpu_device), budget=1e-3)
    e_cal = _errors_vs_oracle(m, x_test, gpu_device, ora)
    print(f"[hybrid salt1] shipped map would give {rep['worst_shipped']:.2e} (library-measured); calibrated {rep['n_x3']} of {rep['n_groups']} x3: library {rep['worst_calibrated']:.2e}, held-out vs oracle {max(e_cal.values()):.2e}",
          {k: f"{v:.2e}" for k, v in e_cal.items()})
    assert rep["met_budget"] == 1 and rep["met_holdout"] == 1 and max(e_cal.values()) <= 1e-3   # the north star itself (round 5: 1.15e-3)


def test_per_pixel_constraint_and_rebinding_voids_the_calibration(gpu_device, capsys):
    """(a) per_pixel_p999: the optional second constraint -- the 99.9th percentile over pixels of the inverse depth's relative error -- is met on the
    calibration frames and costs promotions (more x3 / x2w groups than the L2-only map).  (b) A calibrated map belongs to the weight VALUES it was
    derived on: loading other values into the same parameters sends the handle back to all-x3 (source 3) at the next forward, with a notice."""
    from soccdpt_amd.utils.synth import named_weights, synth_input
    sd = named_weights("salt1")
    m = _build(sd)
    x_cal = synth_input(6, seed0=4).to(gpu_device)
    rep0 = m.calibrate_precision(x_cal, budget=5e-4)
    rep1 = m.calibrate_precision(x_cal, budget=5e-4, per_pixel_p999=6e-4)
    print(f"L2-only map: {rep0['n_x3']} x3 + {rep0['n_x2w']} x2w, inv per-pixel p99.9 {rep0['inv_p999_calibrated']:.2e}; with per_pixel_p999 = 6e-4: "
          f"{rep1['n_x3']} x3 + {rep1['n_x2w']} x2w, p99.9 {rep1['inv_p999_calibrated']:.2e} (hold-out {rep1['inv_p999_holdout']:.2e}), cost {rep0['cost_us_calibrated']:.0f} -> {rep1['cost_us_calibrated']:.0f} us")
    assert rep1["met_budget"] == 1 and rep1["per_pixel_budget"] > 0
    assert rep1["inv_p999_calibrated"] <= 0.85 * 6e-4 * 1.001 and rep1["inv_p999_holdout"] <= 6e-4
    assert rep1["cost_us_calibrated"] >= rep0["cost_us_calibrated"] - 1e-3
    assert m.precision_map_source() == "calibrated"
    capsys.readouterr()
    m.load_state_dict(named_weights("salt2"), strict=False)     # same parameters, other values
    m.network(x_cal[:1])
    assert m.precision_map_source() == "uncalibrated-all-x3"
    assert "weights changed since" in capsys.readouterr().out
    assert all(v == 3 for v in m._engine(gpu_device).prec_map().values())


def test_failed_calibration_leaves_the_handle_as_it_was(gpu_device):
    """an error inside soccdpt_prec_calibrate_ex after its first measured forward used to leave a half-built map on an unprepared handle.
    Provoked here with a workspace that is too small for the calibration batch only AFTER entry checks pass is not possible from outside, so the
    argument checks are exercised (they must not touch the handle) and the map / source / outputs are compared before and after."""
    from soccdpt_amd.utils.synth import named_weights, synth_input
    m = _build(named_weights("salt1"))
    x = synth_input(3, seed0=4).to(gpu_device)
    inv0, _ = m.network(x)
    eng = m._engine(gpu_device)
    map0, src0 = dict(eng.prec_map()), eng.prec_map_source()
    for kw in (dict(holdout=3), dict(holdout=-1), dict(headroom=1.5)):
        with pytest.raises(RuntimeError):
            eng.calibrate_precision(x, 5e-4, **{"holdout": 0, "headroom": 0.85, **kw})
    assert dict(eng.prec_map()) == map0 and eng.prec_map_source() == src0
    inv1, _ = m.network(x)
    assert torch.equal(inv0, inv1)


def test_calibrate_errors(gpu_device):
    from soccdpt_amd.lib import PREC_F16
    from soccdpt_amd.model.SOccDPT import SOccDPT_V3
    from soccdpt_amd.utils.synth import synth_input, synth_state_dict, write_synth_calib
    calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
    m = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True, precision=PREC_F16)
    m.load_state_dict(synth_state_dict(alias_pretrained=True), strict=False)
    m = m.eval().to(gpu_device)
    with pytest.raises(AssertionError):
        m.calibrate_precision(synth_input(1).to(gpu_device))
    eng = m._engine(gpu_device)
    m.network(synth_input(1).to(gpu_device))
    assert eng.prec_map_source() == -1
    with pytest.raises(RuntimeError, match="SOCCDPT_PREC_MIXED"):
        eng.calibrate_precision(synth_input(1).to(gpu_device))
