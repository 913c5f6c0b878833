"""GPU: dpt_hybrid_384 (BASELINE configs[2]) -- the ViT-hybrid encoder's new kernels one by one against torch
on identical operands, then the whole network and the full forward through the C ABI against the CPU oracle
(oracle/soccdpt_ref.py hybrid_encoder: timm ResNetV2 + ViT restated, parity unpinned at the timm boundary; the adapters around it and the
decoder are pinned to the reference's own code by tests/golden/hybrid_B1_tanh.npz)."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import soccdpt_ref as R

pytestmark = pytest.mark.gpu

PREC = {"bf16": 0, "f32": 1, "f16": 2, "f16x3": 3}
DT = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}


def _rel_l2(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm())


@pytest.mark.parametrize("precision", ["bf16", "f16", "f32"])
# (2, 20): one key tile -- the second key half of the 8-wave form is empty.  B = 22 (B x heads = 264 >= 256 workgroups: no query split, one workgroup
# owns every query block) with N in 33..128: the LDS the key halves meet in is larger than the K / V^T image (the launch sized it for K / V^T alone)
@pytest.mark.parametrize("B,N", [(1, 577), (3, 577), (8, 577), (2, 64), (1, 33), (2, 20), (22, 33), (22, 64), (22, 100), (22, 128)])
def test_vit_attention_vs_torch(gpu_device, precision, B, N):
    """softmax(q k^T / 8) v over a prime-length sequence (ragged last key tile, padded query block) and short sequences."""
    from soccdpt_amd.lib import op_vit_attention
    heads, d = 12, 64
    g = torch.Generator().manual_seed(B * 1000 + N)
    qkv = (torch.randn((B * N, 3 * heads * d), generator=g) * 1.5).to(DT[precision])
    # one query row with a large-magnitude key match: the running max jumps at a late key tile (online-softmax rescale path)
    qkv[N // 2, :d] = 6.0
    qkv[N - 1, heads * d: heads * d + d] = 6.0
    out = torch.empty((B * N, heads * d), dtype=DT[precision], device=gpu_device)
    op_vit_attention(qkv.to(gpu_device), out, B, N, heads, PREC[precision])
    torch.cuda.synchronize()
    x = qkv.float().reshape(B, N, 3, heads, d).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax((x[0] @ x[1].transpose(-2, -1)) * d ** -0.5, dim=-1) @ x[2]).transpose(1, 2).reshape(B * N, heads * d)
    err = _rel_l2(out.cpu(), ref)
    assert err < {"bf16": 6e-3, "f16": 8e-4, "f32": 2e-6}[precision], err
    assert float((out.cpu().float() - ref).abs().max()) < {"bf16": 6e-2, "f16": 8e-3, "f32": 2e-5}[precision]


def _halo(x_nhwc, dtype):
    B, H, W, C = x_nhwc.shape
    out = torch.zeros((B, H + 2, W + 2, C), dtype=dtype)
    out[:, 1:-1, 1:-1] = x_nhwc.to(dtype)
    return out


@pytest.mark.parametrize("precision", ["bf16", "f32"])
@pytest.mark.parametrize("case", ["s2_same", "s2_pad1", "s1", "ds_1x1_s2"])
def test_igemm_strided_conv_and_group_norm_stats(gpu_device, precision, case):
    """The generalised convolution addressing of igemm (stride 2 with timm's dynamic 'SAME' padding = pad 0 + the halo's right / bottom
    pixel; stride 2 / pad 1 = nn.Conv2d(3, 2, 1); strided 1x1 gather) and the GroupNorm statistics its epilogue produces, against
    F.conv2d / F.group_norm statistics on the operand-rounded inputs."""
    from soccdpt_amd.lib import op_igemm
    dt = DT[precision]
    g = torch.Generator().manual_seed(11)
    B, Hi, Cin, Cout = 2, 48, 128, 256
    x = torch.randn((B, Cin, Hi, Hi), generator=g) + 0.3
    xq = x.to(dt).float()
    if case == "ds_1x1_s2":
        w = torch.randn((Cout, Cin, 1, 1), generator=g) / Cin ** 0.5
        wq = w.to(dt).float()
        ref = F.conv2d(xq, wq, stride=2)
        Ho = Hi // 2
        xin = xq.permute(0, 2, 3, 1).contiguous().to(dt)
        wt = wq.reshape(Cout, Cin).to(dt)
        kw = dict(taps=1, gather1=1, conv=dict(stride=2, pad=0, in_halo=0, Hi=Hi, Wi=Hi))
    else:
        w = torch.randn((Cout, Cin, 3, 3), generator=g) / (9 * Cin) ** 0.5
        wq = w.to(dt).float()
        if case == "s2_same":
            ref = F.conv2d(R.pad_same(xq, 3, 2), wq, stride=2)
            conv = dict(stride=2, pad=0, in_halo=1, Hi=Hi, Wi=Hi)
        elif case == "s2_pad1":
            ref = F.conv2d(xq, wq, stride=2, padding=1)
            conv = dict(stride=2, pad=1, in_halo=1, Hi=Hi, Wi=Hi)
        else:
            ref = F.conv2d(xq, wq, padding=1)
            conv = dict(stride=1, pad=1, in_halo=1, Hi=Hi, Wi=Hi)
        Ho = ref.shape[-1]
        xin = _halo(xq.permute(0, 2, 3, 1), dt)
        wt = wq.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().to(dt)
        kw = dict(taps=9, conv=conv)
    M = B * Ho * Ho
    out = torch.empty((M, Cout), device=gpu_device)
    cpg = Cout // 32
    stats = torch.full((B, 32, 2), -7.0, device=gpu_device)
    part = torch.zeros(((M // 64) * 32 * 2,), device=gpu_device)
    count = torch.zeros((B,), dtype=torch.int32, device=gpu_device)
    for rep in range(2):      # the counters must be back at zero for the second launch
        op_igemm(xin.to(gpu_device), wt.to(gpu_device), M, Cout, Cin, H=Ho, W=Ho, ldx=Cin, out_f32=out, precision=PREC[precision],
                 gn_stats=stats, gn_part=part, gn_count=count, gn_cpg=cpg, gn_hw=Ho * Ho, **kw)
        torch.cuda.synchronize()
        got = out.cpu().reshape(B, Ho, Ho, Cout).permute(0, 3, 1, 2)
        torch.testing.assert_close(got, ref, rtol=2e-5 if precision == "f32" else 1e-4, atol=2e-5 if precision == "f32" else 2e-4)
        gr = got.reshape(B, 32, -1).double()
        mean, var = gr.mean(-1), gr.var(-1, unbiased=False)
        torch.testing.assert_close(stats[..., 0].cpu().double(), mean, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(stats[..., 1].cpu().double(), 1.0 / torch.sqrt(var + 1e-5), rtol=1e-5, atol=1e-6)
        assert int(count.abs().sum()) == 0
    # deferred form (round 5; what the eval forward launches): no counter -> the launch stops at the per-tile partials [M tiles][32][2] and the READER adds
    # them (gn_apply / gn_finish in csrc/hybrid.hip; covered through the network tests).  Here: the partials, added in float64, are the same statistics.
    stats.fill_(-7.0)
    part.zero_()
    op_igemm(xin.to(gpu_device), wt.to(gpu_device), M, Cout, Cin, H=Ho, W=Ho, ldx=Cin, out_f32=out, precision=PREC[precision],
             gn_stats=stats, gn_part=part, gn_count=None, gn_cpg=cpg, gn_hw=Ho * Ho, **kw)
    torch.cuda.synchronize()
    assert float(stats.min()) == -7.0 and float(stats.max()) == -7.0          # untouched
    p = part.cpu().double()
    used = int(torch.nonzero(p).max()) + 1                                    # sums of squares are positive: the last used float is non-zero
    assert used % (B * 32 * 2) == 0 and (Ho * Ho) % (used // (B * 32 * 2)) == 0
    sums = p[:used].reshape(B, -1, 32, 2).sum(1)
    cnt = Ho * Ho * cpg
    pmean = sums[..., 0] / cnt
    pvar = sums[..., 1] / cnt - pmean ** 2
    torch.testing.assert_close(pmean, mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(1.0 / torch.sqrt(pvar + 1e-5), 1.0 / torch.sqrt(var + 1e-5), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("precision", ["bf16", "f32"])
def test_igemm_readout_projection_two_segment_rows(gpu_device, precision):
    """ProjectReadout (/root/reference/SOccDPT/model/backbones/utils.py:27-40) as ONE GEMM: rows = the 576 patch tokens of each sample
    (class-token row skipped: row groups), K = [token | that sample's class token] (second A segment), + bias, GELU."""
    from soccdpt_amd.lib import op_igemm
    dt = DT[precision]
    g = torch.Generator().manual_seed(5)
    B, NT, E = 3, 577, 768
    tok = torch.randn((B, NT, E), generator=g).to(dt)
    w = (torch.randn((E, 2 * E), generator=g) / (2 * E) ** 0.5).to(dt)
    bias = torch.randn((E,), generator=g) * 0.1
    out = torch.empty((B * (NT - 1), E), dtype=dt, device=gpu_device)
    op_igemm(tok.to(gpu_device), w.to(gpu_device), B * (NT - 1), E, 2 * E, ldx=E, bias=bias.to(gpu_device), act=2, out_bf16=out, precision=PREC[precision],
             grp_rows=NT - 1, grp_stride=NT * E, grp_off=E, seg2_k=E, seg2_off=0)
    torch.cuda.synchronize()
    x = tok.float()
    feats = torch.cat((x[:, 1:], x[:, :1].expand(-1, NT - 1, -1)), -1)
    ref = F.gelu(F.linear(feats, w.float(), bias)).reshape(B * (NT - 1), E)
    tol = dict(rtol=2e-5, atol=2e-5) if precision == "f32" else dict(rtol=2e-2, atol=2e-2)   # 16-bit: output rounding + the fast GELU of the 16-bit epilogue
    torch.testing.assert_close(out.cpu().float(), ref, **tol)
    assert _rel_l2(out.cpu(), ref) < (1e-5 if precision == "f32" else 5e-3)


def _model(gpu_device, precision, sigmoid=False):
    from soccdpt_amd.model.SOccDPT import SOccDPT_V3
    from soccdpt_amd.utils.synth import synth_state_dict, write_synth_calib
    calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
    m = SOccDPT_V3(sigmoid=sigmoid, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True, model_type="dpt_hybrid_384", precision=PREC[precision])
    sd = synth_state_dict("vitb_rn50_384", alias_pretrained=True)
    r = m.load_state_dict(sd, strict=False)
    assert not r.unexpected_keys
    return m.eval().to(gpu_device), sd


# Measured on MI355X (rel. L2 vs the fp32 CPU oracle, synthetic random weights); the asserts sit at <= 2x the measurement.
#   f32   feat 2.2e-6 / 8.0e-6 / 4.1e-5 / 3.7e-5, path_1 2.0e-5, inv 1.0e-5, logits 3.2e-5     -> meets the north star's 1e-3 by 25x
#   fp16  feat 1.4e-3 / 5.1e-3 / 2.5e-2 / 2.2e-2, path_1 1.2e-2, inv 5.7e-3, logits 1.8e-2     -> does NOT meet 1e-3
#   bf16  feat 1.1e-2 / 4.0e-2 / 1.5e-1 / 1.4e-1, path_1 7.2e-2, inv 3.4e-2, logits 1.1e-1     -> does NOT meet 1e-3
# The random-weight hybrid network amplifies any perturbation ~17x between its first and its last hooked map (visible in f32 too:
# 2.2e-6 -> 4.1e-5): 16 weight-standardised (zero-mean filter) bottlenecks + 12 ViT blocks.  16-bit operand rounding is therefore not
# parity-grade on this model.  The parity claim for configs[2] is carried by SOCCDPT_PREC_MIXED (x3 on the ResNetV2 stages, the 1x1 out_convs and
# one read-out projection, fp16 elsewhere: all seven quantities <= 1e-3 at 820 frames/s, tests/test_mixed_gpu.py), by SOCCDPT_PREC_F16X3 (1e-5, 520
# frames/s) and by the exact-f32 mode (DESIGN.md sections 2 and 10).
TOL = {
    "f32": dict(feat0=1e-4, feat1=1e-4, feat2=1e-4, feat3=1e-4, path1=1e-4, inv=1e-4, logits=1e-4),
    # split-operand fp16 (three fp16 MFMAs per product, ~22 significand bits): the fast parity-grade mode, same bounds as f32
    "f16x3": dict(feat0=1e-4, feat1=1e-4, feat2=1e-4, feat3=1e-4, path1=1e-4, inv=1e-4, logits=1e-4),
    "f16": dict(feat0=3e-3, feat1=1.1e-2, feat2=5e-2, feat3=4.5e-2, path1=2.4e-2, inv=1.2e-2, logits=3.7e-2),
    "bf16": dict(feat0=2.3e-2, feat1=8e-2, feat2=0.31, feat3=0.28, path1=0.15, inv=7e-2, logits=0.22),
}


@pytest.mark.parametrize("B", [1, 4])
@pytest.mark.parametrize("precision", ["bf16", "f16", "f32", "f16x3"])
def test_hybrid_384_network_vs_oracle(gpu_device, precision, B):
    """BASELINE configs[2]: SOccDPT_V3 dpt_hybrid_384 at 384 x 384, B = 1 and the configuration's B = 4: the four reassembled feature maps,
    path_1, inverse depth and class logits against the CPU oracle; f32 / fp16 modes within the north star's 1e-3."""
    from soccdpt_amd.utils.synth import synth_input
    m, sd = _model(gpu_device, precision)
    x = synth_input(B, size=384, seed0=21)
    inv, seg = m.network(x.to(gpu_device))
    torch.cuda.synchronize()
    eng = m._engine(gpu_device)
    nref = min(B, 2)                         # the CPU oracle runs the first frames; frames are independent through the network
    torch.set_num_threads(16)
    with torch.no_grad():
        layers = R.hybrid_encoder(sd, x[:nref])
        o_inv, o_p1 = R.dpt_decoder(sd, layers)
        o_logits = R.seg_logits(sd, o_p1)
    errs = {f"feat{s}": _rel_l2(eng.workspace_tensor(B, f"feat{s}")[:nref].cpu().permute(0, 3, 1, 2), layers[s]) for s in range(4)}
    errs["path1"] = _rel_l2(eng.workspace_tensor(B, "path1")[:nref].cpu().permute(0, 3, 1, 2), o_p1)
    errs["inv"] = _rel_l2(inv[:nref].cpu(), o_inv)
    errs["logits"] = _rel_l2(eng.workspace_tensor(B, "seg_logits")[:nref].cpu().permute(0, 3, 1, 2), o_logits)
    print(f"hybrid_384 {precision} B={B}: rel L2 vs fp32 CPU oracle:", {k: f"{v:.2e}" for k, v in errs.items()}, "launches", eng.launch_count())
    for k, bound in TOL[precision].items():
        assert errs[k] < bound, (k, errs)
    if precision in ("f32", "f16x3"):
        assert max(errs.values()) < 1e-3          # the north star's tolerance
    assert tuple(inv.shape) == (B, 384, 384) and tuple(seg.shape) == (B, 3, 384, 384)


def test_hybrid_384_is_bitwise_reproducible_and_batch_invariant(gpu_device):
    """The GroupNorm statistics reach their reader as per-tile partials that it adds in a fixed order in f64 (round 5: no counters, no atomics): two fresh
    models give the same bits, a frame gives the same bits alone and inside a batch of 4 of the same tiles (the tile configurations of the ResNetV2 launches
    depend on M: compared within a tolerance there), and a second forward on the same engine repeats the first."""
    from soccdpt_amd.lib import PREC_MIXED
    from soccdpt_amd.model.SOccDPT import SOccDPT_V3
    from soccdpt_amd.utils.synth import synth_input, synth_state_dict, write_synth_calib

    def build():
        calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
        m = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True, model_type="dpt_hybrid_384", precision=PREC_MIXED)
        m.load_state_dict(synth_state_dict("vitb_rn50_384", alias_pretrained=True), strict=False)
        return m.eval().to(gpu_device)

    x = synth_input(4, size=384, seed0=5).to(gpu_device)
    a = build()
    inv_a, seg_a = (t.clone() for t in a.network(x))
    inv_a2, seg_a2 = (t.clone() for t in a.network(x))
    b = build()
    inv_b, seg_b = (t.clone() for t in b.network(x))
    torch.cuda.synchronize()
    assert torch.equal(inv_a, inv_a2) and torch.equal(seg_a, seg_a2)          # same engine, repeated
    assert torch.equal(inv_a, inv_b) and torch.equal(seg_a, seg_b)            # fresh engine
    inv_1, seg_1 = (t.clone() for t in b.network(x[2:3]))
    torch.cuda.synchronize()
    # B = 1 picks other tiles (other summation orders, other fp16 roundings of intermediate sums): the same function within the mode's own noise -- measured
    # 1.5e-4 on the inverse depth, 1.4e-3 on the activated class maps (this network amplifies a perturbation 17 x, see TOL above)
    assert _rel_l2(inv_1.cpu(), inv_a[2:3].cpu()) < 5e-4 and _rel_l2(seg_1.cpu(), seg_a[2:3].cpu()) < 4e-3


def test_hybrid_384_full_forward_golden_and_projection(gpu_device, golden_dir):
    """Full forward (depth + seg + points + occupancy) of the hybrid model in the exact-f32 mode against the fixture recorded from the
    REFERENCE's own forward_vit / DPT.forward / seg head / projection (tests/golden/hybrid_B1_tanh.npz), and the projection stage
    bit-exact against the C oracle on the network's own outputs."""
    from oracle import cref
    from soccdpt_amd.utils.synth import synth_input
    g = np.load(os.path.join(golden_dir, "hybrid_B1_tanh.npz"))
    m, sd = _model(gpu_device, "f32")
    x = synth_input(1, size=384, seed0=int(g["seed"]))
    inv, seg = m.network(x.to(gpu_device))
    out = m(x.to(gpu_device))
    torch.cuda.synchronize()
    assert _rel_l2(inv.cpu(), torch.from_numpy(g["inv384"])) < 2e-5
    assert float((seg.cpu()[0, :, ::4, ::4] - torch.from_numpy(g["seg384_sample"])).abs().max()) < 5e-3   # ScaledTanh of logits of magnitude ~70
    assert tuple(out[0].shape) == (1, 1080, 1920) and tuple(out[1].shape) == (3, 1080, 1920) and tuple(out[3].shape) == (1, 256, 256, 32, 3)
    ref = cref.project(inv.cpu(), seg.cpu(), want=("points", "occ_bits"))
    assert np.array_equal(m.last_occ_bits.cpu().numpy().view(np.uint32), ref["occ_bits"])
    assert np.array_equal(np.nan_to_num(out[2].cpu().numpy(), nan=-7), np.nan_to_num(ref["points"], nan=-7))
    a = np.unpackbits(m.last_occ_bits.cpu().numpy().view(np.uint8), bitorder="little")
    b = np.unpackbits(g["occ_bits"].view(np.uint8), bitorder="little")
    iou = float((a & b).sum()) / max(float((a | b).sum()), 1.0)
    print("hybrid f32 occupancy IoU vs the reference-generated golden:", iou, int(a.sum()), int(b.sum()))
    assert iou > 0.97
