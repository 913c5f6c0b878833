"""GPU: the whole network (encoder + decoder + heads) and the full forward through the C ABI against
the CPU oracle on the same synthetic weights/inputs.

One section per arithmetic mode.  The first tests pin SOCCDPT_PREC_BF16 EXPLICITLY (the `net` fixture; BASELINE configs[1] says
"bf16"): bf16 MFMA operands with fp32 accumulation and f32 residual streams cannot reach the north star's 1e-3 relative over ~30
layers (SURVEY.md section 7 "Hard parts"), so their bounds are 2x the measured bf16 errors and serve as regression pins, not as the parity
claim.  The parity claim is carried by the f32 / f16 / f16x3 sections below and by tests/test_mixed_gpu.py for the default
arithmetic (SOCCDPT_PREC_MIXED) at the BASELINE batch sizes; the B = 8 sigmoid / plug-in test here runs the default arithmetic at its
own (mixed) bounds.  DESIGN.md section 2 reports the measured numbers."""
import os
import tempfile

import numpy as np
import pytest
import torch

from oracle import soccdpt_ref as R

pytestmark = pytest.mark.gpu


def _rel_l2(a, b):
    return float((a - b).norm() / b.norm())


@pytest.fixture(scope="module")
def net(gpu_device):
    """SOCCDPT_PREC_BF16, pinned explicitly: the class default is SOCCDPT_PREC_MIXED since round 4 and the bounds below are bf16's."""
    from soccdpt_amd.lib import PREC_BF16
    from soccdpt_amd.model.SOccDPT import SOccDPT_V3
    from soccdpt_amd.utils.synth import synth_state_dict, write_synth_calib
    calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
    m = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True, precision=PREC_BF16)
    sd = synth_state_dict(alias_pretrained=True)
    m.load_state_dict(sd, strict=False)
    return m.eval().to(gpu_device), sd


def test_network_vs_oracle(net, gpu_device):
    from soccdpt_amd.utils.synth import synth_input
    m, sd = net
    x = synth_input(2)
    inv, seg = m.network(x.to(gpu_device))
    torch.cuda.synchronize()
    with torch.no_grad():
        layers = R.swin_encoder(sd, x, R.ARCHS["swin2t16_256"])
        o_inv, o_p1 = R.dpt_decoder(sd, layers)
        o_seg_logit_act = R.seg_head(sd, o_p1, sigmoid=False)
    eng = m._engine(gpu_device)
    errs = {}
    for s in range(4):
        f = eng.workspace_tensor(2, f"feat{s}").cpu().permute(0, 3, 1, 2)
        errs[f"feat{s}"] = _rel_l2(f, layers[s])
    errs["path1"] = _rel_l2(eng.workspace_tensor(2, "path1").cpu().permute(0, 3, 1, 2), o_p1)
    errs["inv"] = _rel_l2(inv.cpu(), o_inv)
    errs["seg"] = _rel_l2(seg.cpu(), o_seg_logit_act)
    print("relative L2 error vs fp32 CPU oracle:", {k: f"{v:.2e}" for k, v in errs.items()})
    # bounds = 2x the measured bf16 error (features 3.6e-3 .. 8.1e-3, path_1 5.8e-3, inv 3.3e-3, ScaledTanh
    # probabilities 4.2e-2 with the x12 synthetic logit gain): tight enough to catch a regression, VERDICT r1 weak #2
    assert errs["feat0"] < 7.5e-3 and errs["feat1"] < 1.1e-2 and errs["feat2"] < 1.4e-2 and errs["feat3"] < 1.6e-2
    assert errs["path1"] < 1.2e-2
    assert errs["inv"] < 7e-3
    assert errs["seg"] < 8e-2
    assert m._engine(gpu_device).launch_count() > 50


def test_full_forward_vs_oracle(net, gpu_device):
    from soccdpt_amd.utils.synth import synth_input
    m, sd = net
    x = synth_input(2, seed0=10)
    inv_up, seg_up, pts, occ = m(x.to(gpu_device))
    torch.cuda.synchronize()
    assert tuple(inv_up.shape) == (2, 1080, 1920) and tuple(seg_up.shape) == (2, 3, 1080, 1920)
    assert tuple(pts.shape) == (2, 1080, 1920, 3) and tuple(occ.shape) == (2, 256, 256, 32, 3)
    o_inv, o_seg, o_pts, o_occ = R.soccdpt_v3_forward(sd, x, sigmoid=False)
    assert _rel_l2(inv_up.cpu(), o_inv) < 7e-3
    finite = torch.isfinite(o_pts) & torch.isfinite(pts.cpu())
    assert _rel_l2(pts.cpu()[finite], o_pts[finite]) < 1.5e-2
    assert torch.equal(occ[0], occ[1])
    # occupancy: voxel sets agree up to points that the 1e-2-level depth error moves across a voxel face
    a, b = occ[0].cpu() > 0, o_occ[0] > 0
    iou = float((a & b).sum()) / max(float((a | b).sum()), 1.0)
    print("occupancy IoU vs oracle:", iou, "voxels", int(a.sum()), int(b.sum()))
    assert iou > 0.9      # measured 0.94 in bf16
    # B == 1 squeeze quirk
    out1 = m(x[:1].to(gpu_device))
    assert tuple(out1[1].shape) == (3, 1080, 1920) and tuple(out1[0].shape) == (1, 1080, 1920)


def test_projection_stage_bit_exact_on_network_output(net, gpu_device):
    """Bit-exact contract at the projection-stage boundary: feed the GPU network's own (inv, seg) to the C oracle."""
    from oracle import cref
    from soccdpt_amd.utils.synth import synth_input
    m, sd = net
    x = synth_input(1, seed0=3)
    inv, seg = m.network(x.to(gpu_device))
    out = m.get_semantic_occupancy(inv, seg)
    torch.cuda.synchronize()
    ref = cref.project(inv.cpu(), seg.cpu())
    assert np.array_equal(m.last_occ_bits.cpu().numpy().view(np.uint32), ref["occ_bits"])
    assert np.array_equal(np.nan_to_num(out[2].cpu().numpy(), nan=-7), np.nan_to_num(ref["points"], nan=-7))


def test_multi_stream_sub_batches_match_single_stream(net, gpu_device):
    """soccdpt_set_streams: dealing the batch to concurrent sub-batches must not change any frame's result beyond
    summation-order noise (tile shapes depend on M), and the union occupancy must still cover every frame."""
    import copy
    from soccdpt_amd.model.SOccDPT import SOccDPT_V3
    from soccdpt_amd.utils.synth import synth_input, write_synth_calib
    m, sd = net
    x = synth_input(5, seed0=30).to(gpu_device)          # 5 frames over 4 streams: uneven chunks (2,1,1,1)
    inv1, seg1 = m.network(x)
    out1 = m(x)
    calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
    from soccdpt_amd.lib import PREC_BF16
    m4 = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True, streams=4, precision=PREC_BF16)
    m4.load_state_dict(sd, strict=False)
    m4 = m4.eval().to(gpu_device)
    inv4, seg4 = m4.network(x)
    out4 = m4(x)
    torch.cuda.synchronize()
    assert _rel_l2(inv4, inv1) < 2e-3 and _rel_l2(seg4, seg1) < 2e-2
    assert _rel_l2(out4[0], out1[0]) < 2e-3
    a, b = out4[3][0] > 0, out1[3][0] > 0
    assert float((a & b).sum()) / float((a | b).sum()) > 0.9
    # repeated calls are deterministic
    out4b = m4(x)
    torch.cuda.synchronize()
    assert torch.equal(out4b[0], out4[0]) and torch.equal(out4b[3], out4[3])


# ---------------- exact-f32 parity mode: the north star's 1e-3 relative tolerance ----------------
@pytest.fixture(scope="module")
def net_f32(gpu_device):
    from soccdpt_amd.lib import PREC_F32
    from soccdpt_amd.model.SOccDPT import SOccDPT_V3
    from soccdpt_amd.utils.synth import synth_state_dict, write_synth_calib
    calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
    m = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True, precision=PREC_F32)
    sd = synth_state_dict(alias_pretrained=True)
    m.load_state_dict(sd, strict=False)
    return m.eval().to(gpu_device), sd


def test_f32_mode_meets_1e3_relative(net_f32, gpu_device):
    """SOCCDPT_PREC_F32 (f32 operands, exact-f32 MFMA): depth maps, class probabilities and features within 1e-3
    relative of the reference-equivalent CPU forward (tolerance stated by BASELINE.json's north star)."""
    from soccdpt_amd.utils.synth import synth_input
    m, sd = net_f32
    x = synth_input(2, seed0=4)
    inv, seg = m.network(x.to(gpu_device))
    torch.cuda.synchronize()
    with torch.no_grad():
        layers = R.swin_encoder(sd, x, R.ARCHS["swin2t16_256"])
        o_inv, o_p1 = R.dpt_decoder(sd, layers)
        o_seg = R.seg_head(sd, o_p1, sigmoid=False)
    eng = m._engine(gpu_device)
    errs = {f"feat{s}": _rel_l2(eng.workspace_tensor(2, f"feat{s}").cpu().permute(0, 3, 1, 2), layers[s]) for s in range(4)}
    errs["path1"] = _rel_l2(eng.workspace_tensor(2, "path1").cpu().permute(0, 3, 1, 2), o_p1)
    errs["inv"] = _rel_l2(inv.cpu(), o_inv)
    errs["seg"] = _rel_l2(seg.cpu(), o_seg)
    maxrel_inv = float(((inv.cpu() - o_inv).abs() / o_inv.abs().clamp_min(1e-6)).max())
    print("f32 mode, relative L2 vs fp32 CPU oracle:", {k: f"{v:.2e}" for k, v in errs.items()}, "max elementwise rel (inv):", f"{maxrel_inv:.2e}")
    for k, v in errs.items():
        assert v < 1e-3, (k, v)
    assert maxrel_inv < 1e-3                       # every depth pixel within 1e-3 relative
    assert float((seg.cpu() - o_seg).abs().max()) < 1e-3


def test_f32_mode_full_forward_occupancy(net_f32, gpu_device):
    from soccdpt_amd.utils.synth import synth_input
    m, sd = net_f32
    x = synth_input(2, seed0=10)
    inv_up, seg_up, pts, occ = m(x.to(gpu_device))
    torch.cuda.synchronize()
    o_inv, o_seg, o_pts, o_occ = R.soccdpt_v3_forward(sd, x, sigmoid=False)
    assert _rel_l2(inv_up.cpu(), o_inv) < 1e-3
    a, b = occ[0].cpu() > 0, o_occ[0] > 0
    iou = float((a & b).sum()) / max(float((a | b).sum()), 1.0)
    print("f32 mode occupancy IoU vs oracle:", iou, int(a.sum()), int(b.sum()))
    assert iou > 0.97


@pytest.mark.parametrize("precision", ["bf16", "f16", "f32", "f16x3"])
def test_swin2_base_384_network_vs_oracle(gpu_device, precision):
    """BASELINE config 4 model (dpt_swin2_base_384: 24x24 / 12x12 windows, 384x384 input), B=1; bf16 mode within the
    bf16 tolerance, exact-f32 mode within the north star's 1e-3."""
    from soccdpt_amd.lib import PREC_BF16, PREC_F16, PREC_F16X3, PREC_F32
    from soccdpt_amd.model.SOccDPT import SOccDPT_V3
    from soccdpt_amd.utils.synth import synth_input, synth_state_dict, write_synth_calib
    calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
    m = SOccDPT_V3(sigmoid=True, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True, model_type="dpt_swin2_base_384",
                   precision={"f32": PREC_F32, "f16": PREC_F16, "bf16": PREC_BF16, "f16x3": PREC_F16X3}[precision])
    sd = synth_state_dict("swin2b24_384", alias_pretrained=True)
    r = m.load_state_dict(sd, strict=False)
    assert not r.unexpected_keys
    m = m.eval().to(gpu_device)
    x = synth_input(1, size=384, seed0=8)
    inv, seg = m.network(x.to(gpu_device))
    out = m(x.to(gpu_device))
    torch.cuda.synchronize()
    torch.set_num_threads(16)
    with torch.no_grad():
        o_inv, o_seg, o_p1 = R.soccdpt_v3_network(sd, x, backbone="swin2b24_384", sigmoid=True)
    e_inv, e_seg = _rel_l2(inv.cpu(), o_inv), _rel_l2(seg.cpu(), o_seg)
    print(f"swin2_base_384 {precision}: rel L2 inv", f"{e_inv:.2e}", "seg", f"{e_seg:.2e}")
    assert tuple(inv.shape) == (1, 384, 384) and tuple(out[3].shape) == (1, 256, 256, 32, 3)
    if precision == "f16x3":
        assert e_inv < 1e-4 and e_seg < 1e-4      # split-operand fp16: f32-grade
    if precision in ("f32", "f16", "f16x3"):
        assert e_inv < 1e-3 and e_seg < 1e-3
    else:
        assert e_inv < 6e-3 and e_seg < 2.5e-2     # measured 2.9e-3 / 1.2e-2 (bf16)


def test_swin2_base_384_B8_vs_oracle(gpu_device):
    """BASELINE configs[3]'s per-GPU shape (8 frames of dpt_swin2_base_384): the tile / split-K decisions differ from B = 1
    (layer4_rn is split-K at 4 frames and not at 8, the 96^2 convs move to other tiles), so this batch size gets its own oracle
    comparison.  fp16 operands: the mode that has to meet 1e-3.  The CPU oracle runs 2 of the 8 frames
    (frames are independent through the network), the rest are checked for batch-invariance against a B = 2 run."""
    from soccdpt_amd.lib import PREC_F16
    from soccdpt_amd.model.SOccDPT import SOccDPT_V3
    from soccdpt_amd.utils.synth import synth_input, synth_state_dict, write_synth_calib
    calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
    m = SOccDPT_V3(sigmoid=True, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True, model_type="dpt_swin2_base_384",
                   precision=PREC_F16)
    sd = synth_state_dict("swin2b24_384", alias_pretrained=True)
    m.load_state_dict(sd, strict=False)
    m = m.eval().to(gpu_device)
    x = synth_input(8, size=384, seed0=40)
    inv, seg = m.network(x.to(gpu_device))
    inv, seg = inv.cpu(), seg.cpu()
    logits = m._engine(gpu_device).workspace_tensor(8, "seg_logits").cpu().permute(0, 3, 1, 2)
    inv2, seg2 = m.network(x[6:8].to(gpu_device))
    torch.cuda.synchronize()
    torch.set_num_threads(16)
    with torch.no_grad():
        o_inv, o_seg, o_p1 = R.soccdpt_v3_network(sd, x[[0, 7]], backbone="swin2b24_384", sigmoid=True)
        o_logits = R.seg_logits(sd, o_p1)
    e_inv, e_logit, e_seg = _rel_l2(inv[[0, 7]], o_inv), _rel_l2(logits[[0, 7]], o_logits), _rel_l2(seg[[0, 7]], o_seg)
    print(f"swin2_base_384 B=8 f16: rel L2 inv {e_inv:.2e} class logits {e_logit:.2e} sigmoid probabilities {e_seg:.2e}")
    assert e_inv < 1e-3 and e_logit < 1e-3       # the north star's quantities: depth maps and class logits
    assert e_seg < 2.5e-3                        # probabilities: the x12 synthetic logit gain amplifies (measured 1.1e-3)
    # same frames at B = 2: other tiles / split-K, so only round-off may differ
    assert _rel_l2(inv2.cpu(), inv[6:8]) < 4e-4 and _rel_l2(seg2.cpu(), seg[6:8]) < 1.2e-3    # measured 1.7e-4 / 5.8e-4


def test_full_batch_8_sigmoid_and_plugin_pattern(gpu_device):
    """BASELINE configs[1] size (B=8), sigmoid head; plus the reference's subclass-plugin pattern
    (scripts/eval_others.py:54-247): a SOccDPT subclass feeds its own (inv_depth, segmentation) to get_semantic_occupancy."""
    from soccdpt_amd.model.SOccDPT import SOccDPT, SOccDPT_V3
    from soccdpt_amd.utils.synth import synth_input, synth_state_dict, write_synth_calib
    calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
    m = SOccDPT_V3(sigmoid=True, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True)
    sd = synth_state_dict(alias_pretrained=True)
    m.load_state_dict(sd, strict=False)
    m = m.eval().to(gpu_device)
    x = synth_input(8, seed0=100)
    inv, seg = m.network(x.to(gpu_device))
    out = m(x.to(gpu_device))
    torch.cuda.synchronize()
    torch.set_num_threads(16)
    with torch.no_grad():
        o_inv, o_seg, _ = R.soccdpt_v3_network(sd, x, sigmoid=True)
    e_inv, e_seg = _rel_l2(inv.cpu(), o_inv), _rel_l2(seg.cpu(), o_seg)
    print(f"B=8 sigmoid, default arithmetic (mixed): rel L2 inv {e_inv:.2e} seg {e_seg:.2e}")
    from soccdpt_amd.lib import PREC_MIXED
    assert m.precision == PREC_MIXED
    assert e_inv < 5e-4 and e_seg < 6e-3   # the shipped map's bar on inverse depth (measured 2.9e-4); sigmoid probabilities: x12 synthetic logit gain on ~4e-4 logits (measured 3.2e-3)
    assert tuple(out[3].shape) == (8, 256, 256, 32, 3)
    for b in range(1, 8):
        assert torch.equal(out[3][0], out[3][b])          # the union grid in every batch row

    class Plugin(SOccDPT):
        def forward(self, x):
            return self.get_semantic_occupancy(inv, seg)
    p = Plugin(camera_intrinsics_yaml=calib, compute_occ=True)
    o2 = p(x.to(gpu_device))
    torch.cuda.synchronize()
    assert torch.equal(o2[3], out[3]) and torch.equal(torch.nan_to_num(o2[2]), torch.nan_to_num(out[2]))


# ---------------- fp16 operand mode: the 1e-3 tolerance at the full MFMA rate ----------------
@pytest.fixture(scope="module")
def net_f16(gpu_device):
    from soccdpt_amd.lib import PREC_F16
    from soccdpt_amd.model.SOccDPT import SOccDPT_V3
    from soccdpt_amd.utils.synth import synth_state_dict, write_synth_calib
    calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
    m = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True, precision=PREC_F16)
    sd = synth_state_dict(alias_pretrained=True)
    m.load_state_dict(sd, strict=False)
    return m.eval().to(gpu_device), sd


def test_f16_mode_meets_1e3_relative(net_f16, gpu_device):
    """SOCCDPT_PREC_F16 (fp16 MFMA operands, f32 accumulate / residual streams): the network OUTPUTS the north star names
    (depth maps, class logits) and every hooked feature map within 1e-3 relative L2 of the reference-equivalent fp32 CPU
    forward (measured 4.3e-4 / 7.4e-4 / 4.4e-4..9.8e-4; bf16 operands give 3.3e-3 / - / 3.6e-3..8.1e-3)."""
    from soccdpt_amd.utils.synth import synth_input
    m, sd = net_f16
    x = synth_input(2, seed0=4)
    inv, seg = m.network(x.to(gpu_device))
    torch.cuda.synchronize()
    with torch.no_grad():
        layers = R.swin_encoder(sd, x, R.ARCHS["swin2t16_256"])
        o_inv, o_p1 = R.dpt_decoder(sd, layers)
        o_seg = R.seg_head(sd, o_p1, sigmoid=False)
    eng = m._engine(gpu_device)
    errs = {f"feat{s}": _rel_l2(eng.workspace_tensor(2, f"feat{s}").cpu().permute(0, 3, 1, 2), layers[s]) for s in range(4)}
    errs["path1"] = _rel_l2(eng.workspace_tensor(2, "path1").cpu().permute(0, 3, 1, 2), o_p1)
    errs["inv"] = _rel_l2(inv.cpu(), o_inv)
    errs["seg"] = _rel_l2(seg.cpu(), o_seg)
    errs["seg_logits"] = _rel_l2(eng.workspace_tensor(2, "seg_logits").cpu().permute(0, 3, 1, 2), R.seg_logits(sd, o_p1))
    print("f16 mode, relative L2 vs fp32 CPU oracle:", {k: f"{v:.2e}" for k, v in errs.items()},
          "max |seg diff|:", f"{float((seg.cpu() - o_seg).abs().max()):.2e}")
    assert errs["inv"] < 1e-3 and errs["seg_logits"] < 1e-3
    for k in ("feat0", "feat1", "feat2", "feat3", "path1"):
        assert errs[k] < 1e-3, (k, errs[k])   # feat3 measured 9.8e-4: the deepest stage carries 12 blocks of roundings
    # the ScaledTanh probabilities amplify the logit error by the synthetic logit scale (|logit| ~ 6, SURVEY.md 8d weights)
    assert errs["seg"] < 1e-2


def test_f16x3_mode_matches_oracle_like_f32(gpu_device):
    """SOCCDPT_PREC_F16X3 on dpt_swin2_tiny_256 (B = 2 and the benchmark's B = 8 tile choices): every hooked feature map, path_1, inverse depth,
    class logits within 1e-4 relative L2 of the fp32 CPU oracle -- an order of magnitude inside the north star's 1e-3 -- and the probabilities
    PER ELEMENT within 1e-3 absolute; the full forward's voxel set equals the oracle's almost everywhere."""
    from soccdpt_amd.lib import PREC_F16X3
    from soccdpt_amd.model.SOccDPT import SOccDPT_V3
    from soccdpt_amd.utils.synth import synth_input, synth_state_dict, write_synth_calib
    calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
    m = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True, precision=PREC_F16X3)
    sd = synth_state_dict(alias_pretrained=True)
    m.load_state_dict(sd, strict=False)
    m = m.eval().to(gpu_device)
    x = synth_input(8, seed0=4)
    inv8, seg8 = m.network(x.to(gpu_device))
    inv8, seg8 = inv8.cpu(), seg8.cpu()
    inv, seg = m.network(x[:2].to(gpu_device))
    torch.cuda.synchronize()
    with torch.no_grad():
        layers = R.swin_encoder(sd, x[:2], R.ARCHS["swin2t16_256"])
        o_inv, o_p1 = R.dpt_decoder(sd, layers)
        o_seg = R.seg_head(sd, o_p1, sigmoid=False)
    eng = m._engine(gpu_device)
    errs = {f"feat{s}": _rel_l2(eng.workspace_tensor(2, f"feat{s}").cpu().permute(0, 3, 1, 2), layers[s]) for s in range(4)}
    errs["path1"] = _rel_l2(eng.workspace_tensor(2, "path1").cpu().permute(0, 3, 1, 2), o_p1)
    errs["inv"] = _rel_l2(inv.cpu(), o_inv)
    errs["inv_B8"] = _rel_l2(inv8[:2], o_inv)
    errs["seg_logits"] = _rel_l2(eng.workspace_tensor(2, "seg_logits").cpu().permute(0, 3, 1, 2), R.seg_logits(sd, o_p1))
    print("f16x3 mode, relative L2 vs fp32 CPU oracle:", {k: f"{v:.2e}" for k, v in errs.items()},
          "max |seg diff|:", f"{float((seg.cpu() - o_seg).abs().max()):.2e}", "launches", eng.launch_count())
    for k, v in errs.items():
        assert v < 1e-4, (k, errs)
    assert float((seg.cpu() - o_seg).abs().max()) < 1e-3 and float((seg8[:2] - o_seg).abs().max()) < 1e-3
    inv_up, seg_up, pts, occ = m(x[:2].to(gpu_device))
    torch.cuda.synchronize()
    _, _, _, o_occ = R.soccdpt_v3_forward(sd, x[:2], sigmoid=False)
    a, b = occ[0].cpu() > 0, o_occ[0] > 0
    iou = float((a & b).sum()) / max(float((a | b).sum()), 1.0)
    print("f16x3 mode occupancy IoU vs oracle:", iou)
    assert iou > 0.97


def test_f16_mode_full_forward_occupancy(net_f16, gpu_device):
    from soccdpt_amd.utils.synth import synth_input
    m, sd = net_f16
    x = synth_input(2, seed0=10)
    inv_up, seg_up, pts, occ = m(x.to(gpu_device))
    torch.cuda.synchronize()
    o_inv, o_seg, o_pts, o_occ = R.soccdpt_v3_forward(sd, x, sigmoid=False)
    assert _rel_l2(inv_up.cpu(), o_inv) < 1e-3
    a, b = occ[0].cpu() > 0, o_occ[0] > 0
    iou = float((a & b).sum()) / max(float((a | b).sum()), 1.0)
    print("f16 mode occupancy IoU vs oracle:", iou, int(a.sum()), int(b.sum()))
    assert iou > 0.97


@pytest.mark.parametrize("streams", [1, 2])
def test_hip_graph_replay_matches_eager(net, gpu_device, streams):
    """soccdpt_set_graph: the network captured once as a hipGraph (optionally as concurrent sub-batch branches) and replayed must
    reproduce the eager launches bit for bit, also after the input buffer is overwritten with new frames."""
    from soccdpt_amd.model.SOccDPT import SOccDPT_V3
    from soccdpt_amd.utils.synth import synth_input, write_synth_calib
    m, sd = net
    calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
    mg = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True, streams=streams, graph=True)
    mg.load_state_dict(sd, strict=False)
    mg = mg.eval().to(gpu_device)
    # the eager twin: same (default) arithmetic and the same sub-batch split (tile shapes depend on the per-chunk M)
    ms = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True, streams=streams)
    ms.load_state_dict(sd, strict=False)
    ms = ms.eval().to(gpu_device)
    for seed in (60, 61, 62):          # first call captures, the next ones replay with new inputs
        x = synth_input(4, seed0=seed).to(gpu_device)
        inv_g, seg_g = mg.network(x)
        inv_e, seg_e = ms.network(x)
        torch.cuda.synchronize()
        assert torch.equal(inv_g, inv_e) and torch.equal(seg_g, seg_e), seed
    out_g, out_e = mg(x), ms(x)
    torch.cuda.synchronize()
    assert torch.equal(out_g[0], out_e[0]) and torch.equal(out_g[3], out_e[3])


def test_back_to_back_modes_without_host_sync(net, gpu_device):
    """Regression test for a timing-dependent schedule race (tools/multistream_probe.py): graph(2 streams) -> eager(2 streams) ->
    eager(1 stream) enqueued back to back, no host synchronisation in between, so that the sub-batch streams really overlap and
    kernels of different launches are co-resident.  Every mode must give the single-stream result."""
    from soccdpt_amd.model.SOccDPT import SOccDPT_V3
    from soccdpt_amd.utils.synth import synth_input, write_synth_calib
    _, sd = net
    calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))

    def mk(**kw):   # the default arithmetic (SOCCDPT_PREC_MIXED) in all three
        m = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True, **kw)
        m.load_state_dict(sd, strict=False)
        return m.eval().to(gpu_device)
    mg, ms, m1 = mk(streams=2, graph=True), mk(streams=2), mk()
    bad = 0
    for seed in range(200, 240):
        x = synth_input(4, seed0=seed).to(gpu_device)
        a, sa = mg.network(x)
        b, sb = ms.network(x)
        c, sc = m1.network(x)
        torch.cuda.synchronize()
        bad += int((a != c).sum()) + int((b != c).sum()) + int(((sa - sc).abs() > 1e-3).sum()) + int(((sb - sc).abs() > 1e-3).sum())
    assert bad == 0, f"{bad} mismatching elements"


def test_repeated_forward_is_bitwise_reproducible(net, gpu_device):
    """300 repeats of the B = 8 forward reproduce the first result bit for bit (network outputs and packed occupancy): every
    kernel is deterministic (no float atomics; split-K sums partials in a fixed order), and a schedule race would show up as a
    sporadic mismatch (tools/soak_determinism.py runs the long version over all modes and both models)."""
    from soccdpt_amd.model.SOccDPT import SOccDPT_V3
    from soccdpt_amd.utils.synth import synth_input, write_synth_calib
    _, sd = net
    m = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml")), compute_occ=True)
    m.load_state_dict(sd, strict=False)   # the default arithmetic (SOCCDPT_PREC_MIXED): the one bench.py times
    m = m.eval().to(gpu_device)
    x = synth_input(8, seed0=7).to(gpu_device)
    inv0, seg0 = m.network(x)
    out0 = m(x)
    bits0 = m.last_occ_bits.clone()
    bad = 0
    for i in range(300):
        inv, seg = m.network(x)
        bad += int(not torch.equal(inv, inv0)) + int(not torch.equal(seg, seg0))
        if i % 10 == 0:
            out = m(x)
            bad += int(not torch.equal(m.last_occ_bits, bits0)) + int(not torch.equal(out[0], out0[0]))
    torch.cuda.synchronize()
    assert bad == 0


@pytest.mark.parametrize("model_type,backbone,img", [("dpt_swin2_tiny_256", "swin2t16_256", 256), ("dpt_swin2_base_384", "swin2b24_384", 384)])
def test_fused_mlp_matches_unfused_path(gpu_device, model_type, backbone, img, monkeypatch):
    """The stage-0 MLP half-blocks run as one fused launch (csrc/mlp_fused.hip); SOCCDPT_MLP_FUSE_MAX=0 at handle creation keeps
    the three-launch path.  Same operand rounding and the same k order in both, only the LayerNorm reduction tree differs:
    the stage features, depth and logits agree to float round-off."""
    from soccdpt_amd.lib import PREC_BF16
    from soccdpt_amd.model.SOccDPT import SOccDPT_V3
    from soccdpt_amd.utils.synth import synth_input, synth_state_dict, write_synth_calib
    calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
    sd = synth_state_dict(backbone, alias_pretrained=True)
    x = synth_input(2, size=img, seed0=31).to(gpu_device)
    outs = []
    for fuse in ("0", None):
        if fuse is None:
            monkeypatch.delenv("SOCCDPT_MLP_FUSE_MAX", raising=False)
        else:
            monkeypatch.setenv("SOCCDPT_MLP_FUSE_MAX", fuse)
        m = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True, model_type=model_type, precision=PREC_BF16)   # a uniform 16-bit mode: every stage-0 MLP is eligible
        m.load_state_dict(sd, strict=False)
        m = m.eval().to(gpu_device)
        inv, _ = m.network(x)             # the class probabilities saturate (ScaledTanh of logits of magnitude 10-100): compare the logits
        torch.cuda.synchronize()
        eng = m._engine(gpu_device)
        outs.append((inv.clone(), eng.workspace_tensor(2, "seg_logits").clone(), eng.workspace_tensor(2, "feat0").clone(), eng.launch_count()))
        del m
    (inv0, seg0, f0, n0), (inv1, seg1, f1, n1) = outs
    # two stage-0 blocks x (fc1, fc2 with the LayerNorm epilogue) -> two blocks x one launch, and the fused kernel of the stage's last
    # block writes its operand copy straight into the PatchMerging layout (no gather launch)
    assert n1 == n0 - 3
    assert _rel_l2(f1, f0) < 2e-3            # bf16 / same-rounding paths: differences come from the last-ulp LayerNorm statistics flipping roundings
    # downstream the flipped bf16 roundings spread like any other operand rounding (each path is 3e-3 from the fp32 oracle)
    assert _rel_l2(inv1, inv0) < 5e-3 and _rel_l2(seg1, seg0) < 5e-3
