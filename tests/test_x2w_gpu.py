"""GPU: the one-sided split "x2w" (round 5): fp16 activations, the launch's WEIGHTS as x3 pairs (hi, lo * 2^11), two fp16 MFMAs
per product (csrc/igemm_kernel.h x2w_t).  Kernel level: every x2w tile against float64 products of the operands it was given (activations
exactly representable in fp16, weights decoded from their pairs): what is left is f32 accumulation.  Network level: a precision map with x2w
groups through the C ABI -- the all-x2w map sits between the all-fp16 and all-x3 maps, and mixing the three formats keeps the zero halos."""
import os
import tempfile

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

X2W = 5


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def _x3(t, dev):
    from soccdpt_amd.lib import x3_decode, x3_encode
    enc = x3_encode(t).to(dev)
    return enc, x3_decode(enc.cpu(), t.shape)


@pytest.mark.parametrize("M,N,K,tune", [(500, 192, 384, -1), (4096, 96, 96, -1), (2048, 768, 3072, -1), (256, 64, 1536, 1), (16384, 384, 384, 2), (2048, 1152, 384, 0)])
def test_x2w_linear_matches_float64(gpu_device, M, N, K, tune):
    from soccdpt_amd.lib import op_igemm
    g = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn((M, K), generator=g) * 1.7 + 0.3).to(torch.float16)      # the activation operand IS fp16
    w = torch.randn((N, K), generator=g) / K ** 0.5
    bias = torch.randn((N,), generator=g)
    we, wd = _x3(w, gpu_device)
    out = torch.full((M, N), float("nan"), device=gpu_device)
    op_igemm(x.to(gpu_device), we, M, N, K, ldx=K, bias=bias.to(gpu_device), out_f32=out, precision=X2W, tune=tune)
    torch.cuda.synchronize()
    ref = x.double() @ wd.T + bias.double()
    err = _rel(out.cpu(), ref)
    # the same launch with fp16-rounded weights (plain fp16 tiles): what x2w removes
    out16 = torch.empty_like(out)
    op_igemm(x.to(gpu_device), w.to(torch.float16).to(gpu_device), M, N, K, ldx=K, bias=bias.to(gpu_device), out_f32=out16, precision=2)
    torch.cuda.synchronize()
    e16 = _rel(out16.cpu(), ref)
    print(f"x2w linear {M}x{N}x{K} (tile {tune}): rel L2 vs float64 {err:.2e}; fp16 weights {e16:.2e}")
    assert err < 1e-6 and e16 > 30 * err


@pytest.mark.parametrize("Cin,Cout,H", [(256, 256, 32), (96, 256, 64), (128, 64, 64)])
def test_x2w_conv3x3_matches_float64(gpu_device, Cin, Cout, H):
    """3x3 convolution over a zero-halo NHWC fp16 image with tap-major x3 weight pairs, bias + ReLU, fp16 halo output + f32 output."""
    from soccdpt_amd.lib import op_igemm
    g = torch.Generator().manual_seed(Cin + Cout)
    B = 2
    x = torch.randn((B, Cin, H, H), generator=g).to(torch.float16)
    w = torch.randn((Cout, Cin, 3, 3), generator=g) / (9 * Cin) ** 0.5
    bias = torch.randn((Cout,), generator=g) * 0.1
    halo = torch.zeros((B, H + 2, H + 2, Cin), dtype=torch.float16)
    halo[:, 1:-1, 1:-1] = x.permute(0, 2, 3, 1)
    wt = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
    we, wd = _x3(wt, gpu_device)
    M = B * H * H
    out = torch.empty((M, Cout), device=gpu_device)
    outh = torch.zeros((B, H + 2, H + 2, Cout), dtype=torch.float16, device=gpu_device)
    op_igemm(halo.to(gpu_device), we, M, Cout, Cin, taps=9, H=H, W=H, bias=bias.to(gpu_device), act=1, out_f32=out, act_on_f32=1, out_bf16=outh, out_halo=1, precision=X2W)
    torch.cuda.synchronize()
    wd4 = wd.reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
    ref = F.relu(F.conv2d(x.double(), wd4, bias.double(), padding=1)).permute(0, 2, 3, 1).reshape(M, Cout)
    err = _rel(out.cpu(), ref)
    print(f"x2w conv3x3 {Cin}->{Cout} @ {H}: rel L2 vs float64 {err:.2e}")
    assert err < 1e-6
    assert _rel(outh.cpu()[:, 1:-1, 1:-1].reshape(M, Cout), ref) < 5e-4            # fp16 operand copy
    assert float(outh.cpu()[:, 0].abs().max()) == 0.0 and float(outh.cpu()[:, :, 0].abs().max()) == 0.0   # the halo stays zero


def _build():
    from soccdpt_amd.lib import PREC_MIXED
    from soccdpt_amd.model.SOccDPT import SOccDPT_V3
    from soccdpt_amd.utils.synth import synth_state_dict, write_synth_calib
    calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
    m = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True, precision=PREC_MIXED)
    sd = synth_state_dict(alias_pretrained=True)
    m.load_state_dict(sd, strict=False)
    return m.eval().to("cuda:0"), sd


def test_all_x2w_map_sits_between_fp16_and_x3(gpu_device):
    """Every group x2w: the weight rounding of every launch is gone, the activation rounding stays -- the errors against the fp32 CPU oracle fall
    to roughly half the all-fp16 ones (weights carry a median 72 % of the variance), still above the all-x3 map's; a map mixing all three formats
    runs and reproduces itself after a detour (zero halos move with the operand size of the activations only)."""
    from oracle import soccdpt_ref as R
    from soccdpt_amd.lib import PREC_F16, PREC_F16X3
    from soccdpt_amd.utils.synth import synth_input
    m, sd = _build()
    eng = m._engine(gpu_device)
    x = synth_input(2, seed0=4)
    with torch.no_grad():
        layers = R.swin_encoder(sd, x, R.ARCHS["swin2t16_256"])
        o_inv, o_p1 = R.dpt_decoder(sd, layers)
        o_logits = R.seg_logits(sd, o_p1)

    def errs():
        inv, _ = m.network(x.to(gpu_device))
        torch.cuda.synchronize()
        e = {f"feat{s}": _rel(eng.workspace_tensor(2, f"feat{s}").cpu().permute(0, 3, 1, 2), layers[s]) for s in range(4)}
        e["path1"] = _rel(eng.workspace_tensor(2, "path1").cpu().permute(0, 3, 1, 2), o_p1)
        e["inv"] = _rel(inv.cpu(), o_inv)
        e["seg_logits"] = _rel(eng.workspace_tensor(2, "seg_logits").cpu().permute(0, 3, 1, 2), o_logits)
        return e, inv
    eng.prec_map_set("*", PREC_F16)
    e16, _ = errs()
    eng.prec_map_set("*", X2W)
    pm = eng.prec_map()
    assert {g for g, v in pm.items() if v != X2W} == {"head.d2", "head.s1"} and all(pm[g] == PREC_F16 for g in ("head.d2", "head.s1"))   # "head" takes x2w since round 6
    with pytest.raises(RuntimeError, match="cannot run x2w"):
        eng.prec_map_set("head.d2", X2W)   # the fused depth tail reads its filter as plain fp16
    e2w, inv_a = errs()
    eng.prec_map_set("*", PREC_F16X3)
    e3, _ = errs()
    print("all fp16:", {k: f"{v:.2e}" for k, v in e16.items()})
    print("all x2w: ", {k: f"{v:.2e}" for k, v in e2w.items()})
    print("all x3:  ", {k: f"{v:.2e}" for k, v in e3.items()})
    for k in e16:
        assert e3[k] < e2w[k] < 0.8 * e16[k], (k, e16[k], e2w[k], e3[k])
    # a mixed map: encoder x2w, reassemble convolutions x3, the rest fp16; there and back again
    eng.prec_map_set("*", PREC_F16)
    eng.prec_map_set("s*", X2W)
    eng.prec_map_set("lrn*", PREC_F16X3)
    a = m.network(x.to(gpu_device))[0].clone()
    eng.prec_map_set("s2*", PREC_F16X3)
    m.network(x.to(gpu_device))
    eng.prec_map_set("s2*", X2W)
    b = m.network(x.to(gpu_device))[0]
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert _rel(a.cpu(), o_inv) < 1e-3
