"""GPU: the split-operand fp16 arithmetic (SOCCDPT_PREC_F16X3; csrc/half16.h "x3"): every GEMM / convolution operand is an fp16 pair
(hi, lo * 2^11) and every product three fp16 MFMAs with f32 accumulation.  Kernel level: igemm instantiations (Linear, 3x3 convolution,
fused LayerNorm epilogue, split-K, GELU + x3 output) against float64 products of the SAME decoded operands; the f32 attentions with x3
output.  Network level: tests/test_hybrid_gpu.py and tests/test_network_gpu.py run the mode against the fp32 CPU oracle (<= 1e-3)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

X3 = 3


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def _x3(t, dev):
    from soccdpt_amd.lib import x3_decode, x3_encode
    enc = x3_encode(t).to(dev)
    return enc, x3_decode(enc.cpu(), t.shape)   # device operand, the float64 value it stands for


@pytest.mark.parametrize("M,N,K", [(500, 192, 384), (4096, 96, 96), (2048, 768, 3072), (64, 32, 128)])
def test_x3_linear_matches_float64(gpu_device, M, N, K):
    """out = X W^T + bias through the x3 igemm (ragged M, every tile family) vs float64 on the decoded operands: ~2^-22 per product."""
    from soccdpt_amd.lib import op_igemm
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn((M, K), generator=g) * 1.7 + 0.3
    w = torch.randn((N, K), generator=g) / K ** 0.5
    x[0, :8] = torch.tensor([1e-7, -3e-6, 6e-5, 0.0, 1000.0, -2047.5, 3.1e-5, 7.0])   # subnormal-hi, zero, large values
    bias = torch.randn((N,), generator=g)
    xe, xd = _x3(x, gpu_device)
    we, wd = _x3(w, gpu_device)
    assert _rel(xd, x) < 2e-7 and _rel(wd, w) < 2e-7          # the pair carries the f32 value to ~2^-24
    out = torch.full((M, N), float("nan"), device=gpu_device)
    op_igemm(xe, we, M, N, K, ldx=K, bias=bias.to(gpu_device), out_f32=out, precision=X3)
    torch.cuda.synchronize()
    ref = xd @ wd.T + bias.double()
    err = _rel(out.cpu(), ref)
    assert err < 1e-6, err
    # the same product in exact f32 (SOCCDPT_PREC_F32) is not closer to float64 by more than a small factor
    out32 = torch.empty_like(out)
    op_igemm(x.to(gpu_device), w.to(gpu_device), M, N, K, ldx=K, bias=bias.to(gpu_device), out_f32=out32, precision=1)
    torch.cuda.synchronize()
    e32 = _rel(out32.cpu(), x.double() @ w.double().T + bias.double())
    print(f"x3 linear {M}x{N}x{K}: rel L2 vs float64 {err:.2e} (exact-f32 MFMA: {e32:.2e})")
    assert err < 8 * max(e32, 5e-8)


def test_x3_gelu_and_x3_output_roundtrip(gpu_device):
    """fc1-style launch: bias + GELU(erf), output written in the x3 operand format; decoded it equals the f32 output to 2^-23."""
    from soccdpt_amd.lib import op_igemm, x3_decode
    g = torch.Generator().manual_seed(3)
    M, N, K = 1000, 384, 96
    x, w = torch.randn((M, K), generator=g), torch.randn((N, K), generator=g) / K ** 0.5
    bias = torch.randn((N,), generator=g) * 0.2
    xe, xd = _x3(x, gpu_device)
    we, wd = _x3(w, gpu_device)
    out_op = torch.zeros((M * N * 2,), dtype=torch.float16, device=gpu_device)
    out_f = torch.empty((M, N), device=gpu_device)
    op_igemm(xe, we, M, N, K, ldx=K, bias=bias.to(gpu_device), act=2, out_bf16=out_op, out_f32=out_f, act_on_f32=1, precision=X3)
    torch.cuda.synchronize()
    ref = F.gelu(xd @ wd.T + bias.double())
    assert _rel(out_f.cpu(), ref) < 1e-6
    dec = x3_decode(out_op.cpu(), (M, N))
    assert _rel(dec, out_f.cpu()) < 1.5e-7
    assert float((dec - out_f.cpu().double()).abs().max()) < 1e-6


@pytest.mark.parametrize("Cin,Cout,H", [(256, 256, 32), (96, 256, 64), (128, 32, 64)])
def test_x3_conv3x3_matches_float64(gpu_device, Cin, Cout, H):
    """3x3 convolution over a zero-halo NHWC x3 image (tap-major x3 weights), bias + ReLU epilogue, f32 and x3 halo outputs."""
    from soccdpt_amd.lib import op_igemm, x3_decode
    g = torch.Generator().manual_seed(Cin + Cout)
    B = 2
    x = torch.randn((B, Cin, H, H), generator=g)
    w = torch.randn((Cout, Cin, 3, 3), generator=g) / (9 * Cin) ** 0.5
    bias = torch.randn((Cout,), generator=g) * 0.1
    halo = torch.zeros((B, H + 2, H + 2, Cin))
    halo[:, 1:-1, 1:-1] = x.permute(0, 2, 3, 1)
    xe, xd = _x3(halo, gpu_device)
    we, wd = _x3(w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous(), gpu_device)
    M = B * H * H
    out = torch.empty((M, Cout), device=gpu_device)
    kw = {}
    if Cout % 16 == 0:
        kw = dict(out_bf16=torch.zeros((B * (H + 2) * (H + 2) * Cout * 2,), dtype=torch.float16, device=gpu_device), out_halo=1)
    op_igemm(xe, we, M, Cout, Cin, taps=9, H=H, W=H, bias=bias.to(gpu_device), act=1, out_f32=out, act_on_f32=1, precision=X3, **kw)
    torch.cuda.synchronize()
    xin = xd[:, 1:-1, 1:-1].permute(0, 3, 1, 2)
    wk = wd.reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
    ref = F.relu(F.conv2d(xin, wk, bias.double(), padding=1)).permute(0, 2, 3, 1).reshape(M, Cout)
    err = _rel(out.cpu(), ref)
    print(f"x3 conv3x3 {Cin}->{Cout} @{H}: rel L2 vs float64 {err:.2e}")
    assert err < 1e-6, err
    if kw:
        dec = x3_decode(kw["out_bf16"].cpu(), (B, H + 2, H + 2, Cout))
        assert float(dec[:, 0].abs().max()) == 0.0 and float(dec[:, :, 0].abs().max()) == 0.0       # the halo stays zero
        assert _rel(dec[:, 1:-1, 1:-1].reshape(M, Cout), out.cpu()) < 1.5e-7


def test_x3_splitk_and_layernorm_epilogue(gpu_device):
    """The small-grid long-K launch (layer4_rn: split-K, bitwise reproducible) and the fused Swin-V2 post-norm epilogue (x += LN(acc + bias),
    x3 operand copy) in the x3 arithmetic."""
    from soccdpt_amd.lib import op_igemm
    g = torch.Generator().manual_seed(9)
    M, N, K = 512, 256, 6912
    x, w = torch.randn((M, K), generator=g), torch.randn((N, K), generator=g) / K ** 0.5
    xe, xd = _x3(x, gpu_device)
    we, wd = _x3(w, gpu_device)
    part = torch.empty((4 * M * N,), device=gpu_device)
    count = torch.zeros((4096,), dtype=torch.int32, device=gpu_device)
    outs = []
    for _ in range(2):
        out = torch.empty((M, N), device=gpu_device)
        op_igemm(xe, we, M, N, K, ldx=K, out_f32=out, precision=X3, splitk=4, sk_part=part, sk_count=count)
        torch.cuda.synchronize()
        outs.append(out.cpu())
    assert torch.equal(outs[0], outs[1]) and int(count.abs().sum()) == 0
    assert _rel(outs[0], xd @ wd.T) < 1e-6


def test_x3_vit_attention_output(gpu_device):
    """softmax(q k^T / 8) v in exact f32 (MFMA f32) with the output written as the x3 operand of the proj GEMM."""
    from soccdpt_amd.lib import op_vit_attention, x3_decode
    B, N, heads, d = 2, 577, 12, 64
    g = torch.Generator().manual_seed(1)
    qkv = torch.randn((B * N, 3 * heads * d), generator=g) * 1.5
    out = torch.zeros((B * N * heads * d * 2,), dtype=torch.float16, device=gpu_device)
    op_vit_attention(qkv.to(gpu_device), out, B, N, heads, X3)
    torch.cuda.synchronize()
    x = qkv.double().reshape(B, N, 3, heads, d).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax((x[0] @ x[1].transpose(-2, -1)) * d ** -0.5, dim=-1) @ x[2]).transpose(1, 2).reshape(B * N, heads * d)
    assert _rel(x3_decode(out.cpu(), (B * N, heads * d)), ref) < 2e-6
