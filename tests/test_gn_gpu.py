"""GPU: the reader side of the hybrid's GroupNorm (csrc/hybrid.hip gn_apply_kernel / gn_finish_kernel, round 5) at kernel level.

timm's ResNetV2 puts GroupNormAct(32 groups) after every StdConv2dSame of the stem and the stages (the backbone of dpt_hybrid_384,
/root/reference/SOccDPT/model/backbones/vit.py:147-201; oracle/soccdpt_ref.py rn_bottleneck).  Since round 5 the producing convolution only leaves per-tile
partial sums; the GroupNorm-apply kernel adds them up itself -- per thread for short tile lists, per workgroup for mid ones, by a separate launch for long ones.
These tests feed the kernel partials computed here in float64 from a random raw tensor (tiles of 64 rows, as a 64-row convolution tile would leave them) and
compare every output with torch.nn.functional.group_norm on the same raw tensor: all three finish modes, both shortcut kinds, the two-group float4 case
(C = 64), every output format, and the statistics the kernel writes back."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _partials(raw, B, hw, C, rows):
    """[B * hw / rows][32][2] = {sum, sum of squares} of each tile of `rows` pixel rows and each group, float32 as the convolution's epilogue leaves them."""
    cpg = C // 32
    t = raw.double().reshape(B, hw // rows, rows, 32, cpg)
    return torch.stack((t.sum((2, 4)), (t * t).sum((2, 4))), -1).reshape(B * (hw // rows), 32, 2).float()


def _gn(raw, B, side, C, gamma, beta):
    x = raw.reshape(B, side, side, C).permute(0, 3, 1, 2)
    return F.group_norm(x.double(), 32, gamma.double(), beta.double(), 1e-5).permute(0, 2, 3, 1).reshape(B * side * side, C)


# side, C, tile rows -> tiles per sample: 9 (per-thread walk), 36 (workgroup walk through LDS), 144 (separate finish launch; C = 64: a float4 spans two groups)
CASES = [(24, 256, 64), (24, 1024, 64), (48, 128, 64), (48, 512, 128), (96, 64, 64), (96, 256, 128)]


@pytest.mark.parametrize("shortcut", ["none", "identity", "projection"])
@pytest.mark.parametrize("side,C,rows", CASES)
def test_gn_apply_from_partials_matches_group_norm(gpu_device, side, C, rows, shortcut):
    from soccdpt_amd.lib import PREC_F16, PREC_F16X3, PREC_F32, op_gn_apply, x3_decode
    B, hw, cpg = 2, side * side, C // 32
    g = torch.Generator().manual_seed(side * 1000 + C)
    raw = torch.randn((B * hw, C), generator=g) * 2.0 + 0.7
    raw[:, : C // 2] += 3.0                                   # groups with a large mean: the variance is a difference of large numbers
    gamma, beta = torch.rand((C,), generator=g) + 0.5, torch.randn((C,), generator=g) * 0.3
    ref = _gn(raw, B, side, C, gamma, beta)
    dev = lambda t: t.to(gpu_device)
    kw = {}
    if shortcut == "identity":
        res = torch.randn((B * hw, C), generator=g)
        ref = ref + res.double()
        kw["res"] = dev(res)
    elif shortcut == "projection":
        raw2 = torch.randn((B * hw, C), generator=g) * 0.5 - 1.0
        gamma2, beta2 = torch.rand((C,), generator=g) + 0.5, torch.randn((C,), generator=g) * 0.3
        ref = ref + _gn(raw2, B, side, C, gamma2, beta2)
        kw.update(raw2=dev(raw2), stats2=torch.full((B, 32, 2), -7.0, device=gpu_device), part2=dev(_partials(raw2, B, hw, C, rows)), tps2=hw // rows,
                  gamma2=dev(gamma2), beta2=dev(beta2))
    ref = ref.clamp_min(0.0)
    stats = torch.full((B, 32, 2), -7.0, device=gpu_device)
    out_f32 = torch.empty((B * hw, C), device=gpu_device)
    out_op = torch.empty((B * hw, C), dtype=torch.float16, device=gpu_device)
    op_gn_apply(dev(raw), stats, dev(gamma), dev(beta), hw, side, cpg, part=dev(_partials(raw, B, hw, C, rows)), tps=hw // rows, out_f32=out_f32, out_op=out_op,
                out_format=PREC_F16, **kw)
    torch.cuda.synchronize()
    # the statistics the kernel (or its finish launch) leaves behind
    x = raw.double().reshape(B, hw, 32, cpg)
    mean, var = x.mean((1, 3)), x.var((1, 3), unbiased=False)
    torch.testing.assert_close(stats[..., 0].cpu().double(), mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(stats[..., 1].cpu().double(), 1.0 / torch.sqrt(var + 1e-5), rtol=1e-5, atol=1e-6)
    if shortcut == "projection":
        x2 = raw2.double().reshape(B, hw, 32, cpg)
        torch.testing.assert_close(kw["stats2"][..., 0].cpu().double(), x2.mean((1, 3)), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(out_f32.cpu().double(), ref, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(out_op.cpu().double(), ref, rtol=2e-3, atol=2e-3)
    # the same launch reading FINISHED statistics (the training step's form) gives the same bits as the one that added the partials up
    out_b = torch.empty_like(out_f32)
    kw_b = {k: v for k, v in kw.items() if k not in ("part2", "tps2")}
    op_gn_apply(dev(raw), stats, dev(gamma), dev(beta), hw, side, cpg, out_f32=out_b, **kw_b)
    torch.cuda.synchronize()
    assert torch.equal(out_b, out_f32)
    # zero-halo image in the x3 format, f32 plain operand copy
    halo = torch.zeros((B * (side + 2) * (side + 2) * C * 2,), dtype=torch.float16, device=gpu_device)
    op32 = torch.empty((B * hw, C), device=gpu_device)
    op_gn_apply(dev(raw), stats, dev(gamma), dev(beta), hw, side, cpg, out_op=op32, out_format=PREC_F32, **kw_b)
    op_gn_apply(dev(raw), stats, dev(gamma), dev(beta), hw, side, cpg, out_halo=halo, out_format=PREC_F16X3, **kw_b)
    torch.cuda.synchronize()
    assert torch.equal(op32, out_f32)
    h = x3_decode(halo, (B, side + 2, side + 2, C)).cpu()
    torch.testing.assert_close(h[:, 1:-1, 1:-1].reshape(B * hw, C), out_f32.cpu().double(), rtol=3e-7, atol=1e-6)
    assert float(h[:, 0].abs().max()) == 0.0 and float(h[:, :, -1].abs().max()) == 0.0


def test_gn_finish_matches_float64(gpu_device):
    from soccdpt_amd.lib import op_gn_finish
    B, hw, C, rows = 3, 192 * 192, 64, 64                      # the stem: 576 tiles per sample
    g = torch.Generator().manual_seed(3)
    raw = torch.randn((B * hw, C), generator=g) * 1.5 + 40.0   # un-normalised pixel levels: mean >> spread
    part = _partials(raw, B, hw, C, rows).to(gpu_device)
    stats = torch.empty((B, 32, 2), device=gpu_device)
    op_gn_finish(part, stats, B, hw // rows, 32, hw, C // 32)
    torch.cuda.synchronize()
    x = raw.double().reshape(B, hw, 32, C // 32)
    torch.testing.assert_close(stats[..., 0].cpu().double(), x.mean((1, 3)), rtol=1e-6, atol=1e-6)
    # the partials are float32: what the sum of squares can resolve of a variance 700 times smaller than mean^2 is ~1e-4 relative
    torch.testing.assert_close(stats[..., 1].cpu().double(), 1.0 / torch.sqrt(x.var((1, 3), unbiased=False) + 1e-5), rtol=5e-4, atol=0)


def test_gn_apply_refuses_what_it_cannot_do(gpu_device):
    from soccdpt_amd.lib import op_gn_apply
    raw = torch.zeros((2 * 16, 96), device=gpu_device)        # C / 4 = 24 does not divide 256
    with pytest.raises(RuntimeError, match="gn_apply"):
        op_gn_apply(raw, torch.zeros((2, 32, 2), device=gpu_device), torch.ones((96,), device=gpu_device), torch.zeros((96,), device=gpu_device), 16, 4, 3)
