"""GPU: the individual HIP kernels (through the kernel-level C-ABI entry points) against plain
PyTorch fp32 references of the same op on the same bf16-rounded operands.

Tolerances: operands are identical bf16 values on both sides, accumulation is fp32 on both sides,
so GEMM/conv results agree to fp32 summation-order noise (rtol 2e-3 on the bf16-rounded outputs,
1e-4 on f32 outputs).  The attention kernel additionally rounds q-hat, k-hat and the softmax
numerators to bf16: atol 3e-2 on outputs of magnitude O(1)."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

from oracle import soccdpt_ref as R

pytestmark = pytest.mark.gpu


def _bf(t):
    return t.to(torch.bfloat16)


def _halo(x_nhwc):  # [B,H,W,C] -> zero-haloed [B,H+2,W+2,C]
    return F.pad(x_nhwc, (0, 0, 1, 1, 1, 1)).contiguous()


@pytest.mark.parametrize("M,N,K", [(300, 288, 96), (1024, 96, 384), (512, 2304, 768), (64, 768, 3072), (4096, 192, 384)])
def test_igemm_linear(gpu_device, M, N, K):
    from soccdpt_amd.lib import op_igemm
    g = torch.Generator().manual_seed(M + N + K)
    x = _bf(torch.randn(M, K, generator=g)).to(gpu_device)
    w = _bf(torch.randn(N, K, generator=g) / math.sqrt(K)).to(gpu_device)
    bias = torch.randn(N, generator=g).to(gpu_device)
    res = torch.randn(M, N, generator=g).to(gpu_device)
    ref = x.float() @ w.float().t() + bias
    out = torch.empty(M, N, device=gpu_device)
    op_igemm(x, w, M, N, K, ldx=K, bias=bias, out_f32=out)
    torch.cuda.synchronize()
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)
    # gelu -> bf16, and residual -> f32
    outb = torch.empty(M, N, dtype=torch.bfloat16, device=gpu_device)
    out2 = torch.empty(M, N, device=gpu_device)
    op_igemm(x, w, M, N, K, ldx=K, bias=bias, act=2, out_bf16=outb)
    op_igemm(x, w, M, N, K, ldx=K, bias=bias, res1=res, out_f32=out2)
    torch.cuda.synchronize()
    torch.testing.assert_close(outb.float(), F.gelu(ref), rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(out2, ref + res, rtol=1e-4, atol=1e-4)
    # every kernel configuration that accepts this K must agree
    for tune in ((0, 1, 2, 3, 4) if K % 64 == 0 else (3, 4)):
        o = torch.empty(M, N, device=gpu_device)
        op_igemm(x, w, M, N, K, ldx=K, bias=bias, out_f32=o, tune=tune)
        torch.cuda.synchronize()
        torch.testing.assert_close(o, ref, rtol=1e-4, atol=1e-4, msg=f"tune={tune}")


@pytest.mark.parametrize("B,H,Cin,Cout", [(2, 16, 256, 256), (3, 8, 96, 256), (1, 32, 192, 256), (2, 16, 256, 128), (1, 8, 768, 256)])
def test_igemm_conv3x3(gpu_device, B, H, Cin, Cout):
    from soccdpt_amd.lib import op_igemm
    g = torch.Generator().manual_seed(B * 1000 + H + Cin + Cout)
    x = _bf(torch.randn(B, Cin, H, H, generator=g)).to(gpu_device)            # NCHW like the reference
    w = _bf(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to(gpu_device)
    bias = torch.randn(Cout, generator=g).to(gpu_device)
    res1 = torch.randn(B, H, H, Cout, generator=g).to(gpu_device)
    res2 = torch.randn(B, H, H, Cout, generator=g).to(gpu_device)
    ref = F.conv2d(x.float(), w.float(), bias, padding=1).permute(0, 2, 3, 1)  # NHWC
    xh = _halo(x.permute(0, 2, 3, 1).contiguous())
    wt = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()             # [Cout][tap][Cin]
    M = B * H * H
    out = torch.empty(B, H, H, Cout, device=gpu_device)
    outh = torch.zeros(B, H + 2, H + 2, Cout, dtype=torch.bfloat16, device=gpu_device)
    op_igemm(xh, wt, M, Cout, Cin, taps=9, H=H, W=H, bias=bias, res1=res1, res2=res2, act=1, out_f32=out, out_bf16=outh,
             out_halo=1)
    torch.cuda.synchronize()
    full = ref + res1 + res2
    torch.testing.assert_close(out, full, rtol=1e-4, atol=2e-4)                 # raw f32 (pre-activation)
    torch.testing.assert_close(outh[:, 1:-1, 1:-1].float(), F.relu(full), rtol=1e-2, atol=1e-2)
    assert float(outh[:, 0].abs().max()) == 0 and float(outh[:, :, 0].abs().max()) == 0  # halo untouched
    for tune in ((0, 1, 2, 3, 4) if Cin % 64 == 0 else (3, 4)):
        o = torch.empty(B, H, H, Cout, device=gpu_device)
        op_igemm(xh, wt, M, Cout, Cin, taps=9, H=H, W=H, bias=bias, out_f32=o, tune=tune)
        torch.cuda.synchronize()
        torch.testing.assert_close(o, ref, rtol=1e-4, atol=2e-4, msg=f"tune={tune}")


@pytest.mark.parametrize("tune", [6, 7, 8, 10, 13, 14, 15, 16, 17, 18, 20, 21, 22, 23, 24, 40, 41, 42, 43, 44, 45])   # 40-45: the v_mfma_f32_32x32x16 forms
@pytest.mark.parametrize("prec", ["bf16", "f16"])
def test_igemm_big_tiles_ragged(gpu_device, tune, prec):
    """The 8-wave tiles (incl. the skewed wave-row schedule of the 2 x 4 layouts) and the other heuristic-only configurations on a
    problem whose M is not a multiple of any tile (1600 rows) and whose K is odd in k-tiles (9 x 128 / 64 = 18, 9 x 96 / 32 = 27)."""
    from soccdpt_amd.lib import PREC_BF16, PREC_F16, op_igemm
    dt = torch.bfloat16 if prec == "bf16" else torch.float16
    B, H, Cout = 1, 40, 256
    Cin = 128 if tune in (6, 7, 8, 10, 13, 14, 20, 21, 22, 23, 40, 41, 42, 43, 44, 45) else 96  # the x32 configurations also take Cin % 64 != 0
    g = torch.Generator().manual_seed(tune)
    x = torch.randn(B, Cin, H, H, generator=g).to(dt).to(gpu_device)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to(dt).to(gpu_device)
    bias = torch.randn(Cout, generator=g).to(gpu_device)
    ref = F.conv2d(x.float(), w.float(), bias, padding=1).permute(0, 2, 3, 1)
    xh = _halo(x.permute(0, 2, 3, 1).contiguous())
    wt = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
    out = torch.empty(B, H, H, Cout, device=gpu_device)
    for _ in range(3):   # repeated launches: a schedule race would show as a sporadic wrong tile
        op_igemm(xh, wt, B * H * H, Cout, Cin, taps=9, H=H, W=H, bias=bias, out_f32=out, tune=tune,
                 precision=PREC_BF16 if prec == "bf16" else PREC_F16)
        torch.cuda.synchronize()
        torch.testing.assert_close(out, ref, rtol=1e-4, atol=2e-4, msg=f"tune={tune}")


def test_igemm_depth_tail(gpu_device):
    """conv3x3 128->32 + bias, ReLU, 1x1 32->1 + bias, ReLU fused (model/dpt.py:209-216)."""
    from soccdpt_amd.lib import op_igemm
    g = torch.Generator().manual_seed(5)
    B, H, Cin = 2, 16, 128
    x = _bf(torch.randn(B, Cin, H, H, generator=g)).to(gpu_device)
    w = _bf(torch.randn(32, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to(gpu_device)
    bias = torch.randn(32, generator=g).to(gpu_device)
    w2 = torch.randn(32, generator=g).to(gpu_device)
    b2 = 0.1
    ref = F.relu(F.conv2d(x.float(), w.float(), bias, padding=1))
    ref = F.relu((ref * w2.view(1, 32, 1, 1)).sum(1) + b2)
    out = torch.empty(B, H, H, device=gpu_device)
    xh = _halo(x.permute(0, 2, 3, 1).contiguous())
    wt = w.permute(0, 2, 3, 1).reshape(32, 9 * Cin).contiguous()
    op_igemm(xh, wt, B * H * H, 32, Cin, taps=9, H=H, W=H, bias=bias, act=1, dot_w=w2, dot_b=b2, out_dot=out)
    torch.cuda.synchronize()
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)


def _halo_t(x_nhwc, dtype):
    B, H, W, C = x_nhwc.shape
    out = torch.zeros((B, H + 2, W + 2, C), dtype=dtype)
    out[:, 1:-1, 1:-1] = x_nhwc.to(dtype)
    return out


def _attention_ref(qkv, table, scale, B, res, ws, shift, heads):
    C = heads * 32
    N = ws * ws
    nw = res // ws
    x = qkv.float().reshape(B, res, res, 3 * C)
    if shift:
        x = torch.roll(x, (-shift, -shift), (1, 2))
    win = x.reshape(B, nw, ws, nw, ws, 3 * C).permute(0, 1, 3, 2, 4, 5).reshape(B * nw * nw, N, 3, heads, 32)
    q, k, v = [win[:, :, i].permute(0, 2, 1, 3) for i in range(3)]
    attn = F.normalize(q, dim=-1) @ F.normalize(k, dim=-1).transpose(-2, -1) * scale.view(1, heads, 1, 1)
    idx = R.relative_position_index(ws).reshape(-1).to(table.device)
    attn = attn + table[idx].reshape(N, N, heads).permute(2, 0, 1).unsqueeze(0)
    mask = R.shift_attn_mask(res, ws, shift)
    if mask is not None:
        attn = (attn.reshape(B, nw * nw, heads, N, N) + mask.to(attn.device)[None, :, None]).reshape(-1, heads, N, N)
    out = (torch.softmax(attn, -1) @ v).transpose(1, 2).reshape(B, nw, nw, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, res, res, C)
    if shift:
        out = torch.roll(out, (shift, shift), (1, 2))
    return out.reshape(B * res * res, C)


@pytest.mark.parametrize("B,res,ws,shift,heads", [(2, 64, 16, 0, 3), (2, 64, 16, 8, 3), (1, 32, 16, 8, 6), (3, 16, 16, 0, 12), (2, 8, 8, 0, 24),
                                                   (1, 96, 24, 12, 4), (2, 48, 24, 12, 8), (2, 24, 24, 0, 16), (3, 12, 12, 0, 32)])
def test_window_attention(gpu_device, B, res, ws, shift, heads):
    from soccdpt_amd.lib import op_window_attention
    g = torch.Generator().manual_seed(res * 100 + shift + heads)
    C = heads * 32
    qkv = _bf(torch.randn(B * res * res, 3 * C, generator=g)).to(gpu_device)
    table = (16 * torch.sigmoid(torch.randn((2 * ws - 1) ** 2, heads, generator=g))).to(gpu_device)
    scale = (10.0 + 5 * torch.rand(heads, generator=g)).to(gpu_device)
    out = torch.empty(B * res * res, C, dtype=torch.bfloat16, device=gpu_device)
    op_window_attention(qkv, table, scale, out, B, res, ws, shift, heads)
    torch.cuda.synchronize()
    ref = _attention_ref(qkv, table, scale, B, res, ws, shift, heads)
    err = (out.float() - ref).abs()
    assert float(err.max()) < 6e-2 and float(err.mean()) < 6e-3, (float(err.max()), float(err.mean()))


@pytest.mark.parametrize("fmt", ["f16", "x2w", "bf16"])
@pytest.mark.parametrize("B,res,ws,shift,heads", [(2, 64, 16, 0, 3), (2, 64, 16, 8, 3), (1, 32, 16, 8, 6), (3, 16, 16, 0, 12), (2, 8, 8, 0, 24), (1, 32, 16, 0, 12), (1, 16, 8, 0, 12)])
def test_window_attention_with_qkv_projection_inside(gpu_device, fmt, B, res, ws, shift, heads):
    """Round 6: qkv = x Wqkv^T + cat(q_bias, 0, v_bias) and the window attention as ONE launch (csrc/attention_qkv.hip), against float64
    of the same 16-bit operands through the reference maths, and against the two-launch chain it replaces (igemm -> fp16 qkv tensor -> window_attention):
    the fused form skips the fp16 rounding of q, k, v in between, so it must sit at least as close to float64 as the chain does (small slack for
    the different summation order)."""
    from soccdpt_amd.lib import PREC_BF16, PREC_F16, PREC_F16X2W, op_igemm, op_window_attention, op_window_attention_qkv, x3_decode, x3_encode
    g = torch.Generator().manual_seed(res * 100 + shift + heads)
    C = heads * 32
    M = B * res * res
    dt = torch.bfloat16 if fmt == "bf16" else torch.float16
    x = (torch.randn(M, C, generator=g) * 0.8).to(dt)
    w = torch.randn(3 * C, C, generator=g) / math.sqrt(C)
    q_bias, v_bias = torch.randn(C, generator=g) * 0.3, torch.randn(C, generator=g) * 0.3
    bias = torch.cat([q_bias, torch.zeros(C), v_bias])
    table = (16 * torch.sigmoid(torch.randn((2 * ws - 1) ** 2, heads, generator=g)))
    scale = (10.0 + 5 * torch.rand(heads, generator=g))
    if fmt == "x2w":
        w_dev = x3_encode(w.to(gpu_device))
        w_eff = x3_decode(w_dev.cpu(), w.shape)        # what the pair represents (22+ bits)
        code = PREC_F16X2W
    else:
        w_dev = w.to(dt).to(gpu_device)
        w_eff = w.to(dt).float()
        code = PREC_BF16 if fmt == "bf16" else PREC_F16
    xd, bd, td, sd = x.to(gpu_device), bias.to(gpu_device), table.to(gpu_device), scale.to(gpu_device)
    out = torch.empty(M, C, dtype=dt, device=gpu_device)
    op_window_attention_qkv(xd, w_dev, bd, td, sd, out, B, res, ws, shift, heads, code)
    torch.cuda.synchronize()
    qkv64 = x.double() @ w_eff.double().t() + bias.double()
    ref = _attention_ref(qkv64.float(), table, scale, B, res, ws, shift, heads)
    err = (out.float().cpu() - ref).abs()
    tol_max, tol_mean = (6e-2, 6e-3) if fmt == "bf16" else (1.2e-2, 8e-4)
    assert float(err.max()) < tol_max and float(err.mean()) < tol_mean, (float(err.max()), float(err.mean()))
    # the chain it replaces, same operands
    qkv16 = torch.empty(M, 3 * C, dtype=dt, device=gpu_device)
    op_igemm(xd, w_dev, M, 3 * C, C, ldx=C, bias=bd, out_bf16=qkv16, precision=code)
    out2 = torch.empty(M, C, dtype=dt, device=gpu_device)
    op_window_attention(qkv16, td, sd, out2, B, res, ws, shift, heads, PREC_BF16 if fmt == "bf16" else PREC_F16)
    torch.cuda.synchronize()
    err2 = (out2.float().cpu() - ref).abs()
    rel = float((out.float() - out2.float()).norm() / out2.float().norm())
    print(f"fused qkv+attention {fmt} B{B} res{res} ws{ws} shift{shift} heads{heads}: mean |err| vs float64 fused {float(err.mean()):.2e} / chain {float(err2.mean()):.2e}; fused vs chain rel L2 {rel:.2e}")
    assert float(err.mean()) <= float(err2.mean()) * 1.05 + 1e-6
    assert rel < (2e-2 if fmt == "bf16" else 3e-3)


def test_window_attention_qkv_x3_output_and_errors(gpu_device):
    """out_x3: the attention output in the x3 operand format (the proj GEMM of an x3 group reads it) decodes to the fp16-output run's values within fp16
    rounding; unsupported window sizes / precisions are refused with an error, not launched."""
    from soccdpt_amd.lib import PREC_F16, PREC_F32, op_window_attention_qkv, x3_decode
    g = torch.Generator().manual_seed(5)
    B, res, ws, heads = 1, 32, 16, 6
    C, M = heads * 32, B * res * res
    x = (torch.randn(M, C, generator=g) * 0.8).half().to(gpu_device)
    w = (torch.randn(3 * C, C, generator=g) / math.sqrt(C)).half().to(gpu_device)
    bias = (torch.randn(3 * C, generator=g) * 0.3).to(gpu_device)
    table = (16 * torch.sigmoid(torch.randn((2 * ws - 1) ** 2, heads, generator=g))).to(gpu_device)
    scale = (10.0 + 5 * torch.rand(heads, generator=g)).to(gpu_device)
    o16 = torch.empty(M, C, dtype=torch.float16, device=gpu_device)
    o3 = torch.empty(M, C, dtype=torch.float32, device=gpu_device)
    op_window_attention_qkv(x, w, bias, table, scale, o16, B, res, ws, 8, heads, PREC_F16)
    op_window_attention_qkv(x, w, bias, table, scale, o3, B, res, ws, 8, heads, PREC_F16, out_x3=True)
    torch.cuda.synchronize()
    dec = x3_decode(o3.cpu(), (M, C)).float()
    torch.testing.assert_close(dec, o16.cpu().float(), rtol=1e-3, atol=1e-6)      # the pair carries the f32 value: within one fp16 rounding of the fp16-output run
    assert float((dec.half() != o16.cpu()).float().mean()) < 1e-3                # (hi + lo re-rounded lands on the neighbour only at ties)
    with pytest.raises(RuntimeError):
        op_window_attention_qkv(x, w, bias, table, scale, o16, B, 48, 24, 0, heads, PREC_F16)     # 24 x 24 windows: the two-launch chain
    with pytest.raises(RuntimeError):
        op_window_attention_qkv(x, w, bias, table, scale, o16, B, res, ws, 0, heads, PREC_F32)


@pytest.mark.parametrize("case", ["relu_halo", "res_sampled_f32_relu_halo", "res_plain_x3", "f32_only_n128"])
def test_winograd_conv3x3_matches_float64_and_the_direct_launch(gpu_device, case):
    """Round 6: the Winograd F(2x2, 3x3) form of the decoder's 3x3 convolutions (csrc/wino.hip; measured slower than the direct launch and NOT routed in
    the forward -- DESIGN.md 9.4 -- but correct, with igemm's epilogue variants of the RCU convolutions).  Against float64 of the same fp16 activations and f32 weights,
    and against the direct implicit-GEMM launch with fp16 weights: the Winograd result must sit within 2.5x the direct launch's error (transformed weights and transformed
    activations each round once more)."""
    from soccdpt_amd.lib import PREC_F16, op_igemm, op_wino_conv, op_wino_weights, x3_decode
    g = torch.Generator().manual_seed(7)
    B, H, C = 2, 32, 256
    N = 128 if case == "f32_only_n128" else 256
    M = B * H * H
    x = torch.relu(torch.randn(B, H, H, C, generator=g)).half()
    w = torch.randn(N, C, 3, 3, generator=g) / math.sqrt(9 * C)
    bias = torch.randn(N, generator=g) * 0.1
    res1 = torch.randn(M, N, generator=g) if case.startswith("res") else None
    res2 = torch.randn(B, H // 2, H // 2, N, generator=g) if case == "res_sampled_f32_relu_halo" else None
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), bias.double(), padding=1).permute(0, 2, 3, 1)
    if res1 is not None:
        ref = ref + res1.reshape(B, H, H, N).double()
    if res2 is not None:
        ref = ref + F.interpolate(res2.permute(0, 3, 1, 2).double(), size=(H, H), mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    xh = _halo_t(x, torch.float16).to(gpu_device)
    u = op_wino_weights(w.to(gpu_device))
    wt = w.permute(0, 2, 3, 1).reshape(N, 9 * C).contiguous().half().to(gpu_device)
    dev = lambda t: None if t is None else t.to(gpu_device)
    relu = case in ("relu_halo", "res_sampled_f32_relu_halo")
    halo = case in ("relu_halo", "res_sampled_f32_relu_halo")
    x3 = case == "res_plain_x3"
    want_f32 = case != "relu_halo"
    want_op = case != "f32_only_n128"
    o32 = torch.empty(M, N, device=gpu_device) if want_f32 else None
    d32 = torch.empty(M, N, device=gpu_device)
    if want_op:
        shape = (B, H + 2, H + 2, N) if halo else (M, N)
        oop = torch.zeros(shape, dtype=torch.float32 if x3 else torch.float16, device=gpu_device)
    else:
        oop = None
    op_wino_conv(xh, u, B, H, H, C, N, bias=dev(bias), res1=dev(res1), res2=dev(res2), res2_hw=(H // 2, H // 2) if res2 is not None else (0, 0), relu=relu,
                 out_f32=o32, out_op=oop, out_halo=halo, out_x3=x3)
    op_igemm(xh, wt, M, N, C, taps=9, H=H, W=H, bias=dev(bias), res1=dev(res1), out_f32=d32, precision=PREC_F16)   # the direct launch (no sampled residual: added below)
    torch.cuda.synchronize()
    refm = ref.reshape(M, N)
    d_ref = refm if res2 is None else refm - F.interpolate(res2.permute(0, 3, 1, 2).double(), size=(H, H), mode="bilinear", align_corners=True).permute(0, 2, 3, 1).reshape(M, N)
    e_direct = float((d32.cpu().double() - d_ref).norm() / d_ref.norm())
    if want_f32:
        e_w = float((o32.cpu().double() - refm).norm() / refm.norm())      # the f32 output is pre-activation (act_on_f32 = 0), like igemm's
        print(f"winograd {case}: rel L2 vs float64 {e_w:.2e}, direct launch {e_direct:.2e}")
        assert e_w < 2.5 * e_direct + 1e-5 and e_w < 8e-4   # in isolation ~2.2x (transformed weights + twice-rounded transformed activations); in the network +4 % of variance (tools/winograd_numerics.py)
    if want_op:
        act = torch.relu(refm) if relu else refm
        got = x3_decode(oop.cpu(), (M, N)).double() if x3 else (oop[:, 1:-1, 1:-1] if halo else oop).cpu().double().reshape(M, N)
        e_op = float((got - act).norm() / act.norm())
        assert e_op < (8e-4 if x3 else 1.2e-3), e_op
        if halo:   # the zero border is never written
            o = oop.cpu()
            assert float(o[:, 0].abs().max()) == 0 and float(o[:, -1].abs().max()) == 0 and float(o[:, :, 0].abs().max()) == 0 and float(o[:, :, -1].abs().max()) == 0
    with pytest.raises(RuntimeError):
        op_wino_conv(xh, u, B, 24, 24, C, N, out_f32=d32)      # 24 is not a multiple of 16: refused, not launched


# ---------------- split-K (deterministic partial-tile exchange, last arriver runs the epilogue) ----------------
@pytest.mark.parametrize("splitk", [2, 3, 8])
def test_igemm_splitk(gpu_device, splitk):
    from soccdpt_amd.lib import op_igemm
    g = torch.Generator().manual_seed(splitk)
    part = torch.full((2 << 20,), float("nan"), device=gpu_device)       # stale contents must not matter
    count = torch.zeros(4096, dtype=torch.int32, device=gpu_device)
    # Linear: M not a multiple of the 32-row tile, N not a multiple of 64
    M, N, K = 300, 736, 3072
    x = _bf(torch.randn(M, K, generator=g)).to(gpu_device)
    w = _bf(torch.randn(N, K, generator=g) / math.sqrt(K)).to(gpu_device)
    bias = torch.randn(N, generator=g).to(gpu_device)
    res = torch.randn(M, N, generator=g).to(gpu_device)
    ref = x.float() @ w.float().t() + bias + res
    outs = []
    for _ in range(3):
        o = torch.empty(M, N, device=gpu_device)
        op_igemm(x, w, M, N, K, ldx=K, bias=bias, res1=res, out_f32=o, splitk=splitk, sk_part=part, sk_count=count)
        outs.append(o)
    torch.cuda.synchronize()
    torch.testing.assert_close(outs[0], ref, rtol=1e-4, atol=1e-4)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])   # fixed summation order: bitwise reproducible
    assert int(count.abs().sum()) == 0                                       # counters are left at zero
    with pytest.raises(RuntimeError):                                        # scratch too small is refused, not overrun
        op_igemm(x, w, M, N, K, ldx=K, out_f32=outs[0], splitk=splitk, sk_part=part[: M * N], sk_count=count)
    # 3x3 convolution with the full decoder epilogue
    B, H, Cin, Cout = 2, 8, 256, 256
    xc = _bf(torch.randn(B, Cin, H, H, generator=g)).to(gpu_device)
    wc = _bf(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to(gpu_device)
    bc = torch.randn(Cout, generator=g).to(gpu_device)
    r1 = torch.randn(B, H, H, Cout, generator=g).to(gpu_device)
    refc = F.conv2d(xc.float(), wc.float(), bc, padding=1).permute(0, 2, 3, 1) + r1
    xh = _halo(xc.permute(0, 2, 3, 1).contiguous())
    wt = wc.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
    oc = torch.empty(B, H, H, Cout, device=gpu_device)
    oh = torch.zeros(B, H + 2, H + 2, Cout, dtype=torch.bfloat16, device=gpu_device)
    op_igemm(xh, wt, B * H * H, Cout, Cin, taps=9, H=H, W=H, bias=bc, res1=r1, act=1, out_f32=oc, out_bf16=oh, out_halo=1,
             splitk=splitk, sk_part=part, sk_count=count)
    torch.cuda.synchronize()
    torch.testing.assert_close(oc, refc, rtol=1e-4, atol=2e-4)
    torch.testing.assert_close(oh[:, 1:-1, 1:-1].float(), F.relu(refc), rtol=1e-2, atol=1e-2)
    assert int(count.abs().sum()) == 0


@pytest.mark.parametrize("prec", ["bf16", "f16", "f32", "f16x3"])
def test_igemm_splitk_deferred_big_tiles(gpu_device, prec):
    """The weight-gradient form of the training step: long K, a handful of 128 x 128 output tiles, many splits, every split storing its partial
    tile and a second launch summing them in order (sk_defer).  Equal to the un-split product, bitwise reproducible, counters untouched,
    epilogue options refused."""
    from soccdpt_amd.lib import PREC_BF16, PREC_F16, PREC_F16X3, PREC_F32, op_igemm, x3_encode
    g = torch.Generator().manual_seed(11)
    M, N, K, S = 256, 384, 4096, 14
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    code, tune, tol = {"bf16": (PREC_BF16, 46, 1e-4), "f16": (PREC_F16, 46, 1e-4), "f32": (PREC_F32, 3, 2e-5), "f16x3": (PREC_F16X3, 3, 2e-5)}[prec]
    if prec == "bf16":
        x, w = _bf(x), _bf(w)
        xd, wd = x.to(gpu_device), w.to(gpu_device)
    elif prec == "f16":
        x, w = x.half(), w.half()
        xd, wd = x.to(gpu_device), w.to(gpu_device)
    elif prec == "f32":
        xd, wd = x.to(gpu_device), w.to(gpu_device)
    else:
        xd, wd = x3_encode(x.to(gpu_device)), x3_encode(w.to(gpu_device))
    ref = (x.double() @ w.double().t()).float().to(gpu_device)
    part = torch.full((S * M * N,), float("nan"), device=gpu_device)
    count = torch.zeros(4096, dtype=torch.int32, device=gpu_device)
    outs = []
    for _ in range(2):
        o = torch.empty(M, N, device=gpu_device)
        op_igemm(xd, wd, M, N, K, ldx=K, out_f32=o, splitk=S, sk_part=part, sk_count=count, tune=tune, precision=code, sk_defer=1)
        outs.append(o)
    torch.cuda.synchronize()
    torch.testing.assert_close(outs[0], ref, rtol=tol, atol=tol)
    assert torch.equal(outs[0], outs[1])
    assert int(count.abs().sum()) == 0
    with pytest.raises(RuntimeError):   # the deferred form writes out_f32 only
        op_igemm(xd, wd, M, N, K, ldx=K, bias=torch.zeros(N, device=gpu_device), out_f32=outs[0], splitk=S, sk_part=part, sk_count=count, tune=tune,
                 precision=code, sk_defer=1)


# ---------------- fp16 operand mode (SOCCDPT_PREC_F16): same kernels, v_mfma_*_f16 ----------------
def _hf(t):
    return t.to(torch.float16)


@pytest.mark.parametrize("M,N,K", [(300, 288, 96), (4096, 192, 384), (64, 768, 3072)])
def test_igemm_linear_f16(gpu_device, M, N, K):
    from soccdpt_amd.lib import PREC_F16, op_igemm
    g = torch.Generator().manual_seed(M + N + K)
    x = _hf(torch.randn(M, K, generator=g)).to(gpu_device)
    w = _hf(torch.randn(N, K, generator=g) / math.sqrt(K)).to(gpu_device)
    bias = torch.randn(N, generator=g).to(gpu_device)
    ref = x.float() @ w.float().t() + bias
    out = torch.empty(M, N, device=gpu_device)
    outh = torch.empty(M, N, dtype=torch.float16, device=gpu_device)
    op_igemm(x, w, M, N, K, ldx=K, bias=bias, out_f32=out, precision=PREC_F16)
    op_igemm(x, w, M, N, K, ldx=K, bias=bias, act=2, out_bf16=outh, precision=PREC_F16)
    torch.cuda.synchronize()
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)          # identical fp16 operands, f32 accumulation on both sides
    torch.testing.assert_close(outh.float(), F.gelu(ref), rtol=1.5e-3, atol=1.5e-3)   # one fp16 rounding of the output
    big = torch.full((M, K), 300.0, dtype=torch.float16, device=gpu_device)           # 300*300*K overflows fp16: stores saturate, never inf
    wbig = torch.full((N, K), 300.0, dtype=torch.float16, device=gpu_device)
    op_igemm(big, wbig, M, N, K, ldx=K, out_bf16=outh, precision=PREC_F16)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(outh).all()) and float(outh.max()) == 65504.0


def test_igemm_conv3x3_f16(gpu_device):
    from soccdpt_amd.lib import PREC_F16, op_igemm
    B, H, Cin, Cout = 2, 16, 256, 256
    g = torch.Generator().manual_seed(77)
    x = _hf(torch.randn(B, Cin, H, H, generator=g)).to(gpu_device)
    w = _hf(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to(gpu_device)
    bias = torch.randn(Cout, generator=g).to(gpu_device)
    ref = F.conv2d(x.float(), w.float(), bias, padding=1).permute(0, 2, 3, 1)
    xh = _halo(x.permute(0, 2, 3, 1).contiguous())
    wt = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
    out = torch.empty(B, H, H, Cout, device=gpu_device)
    outh = torch.zeros(B, H + 2, H + 2, Cout, dtype=torch.float16, device=gpu_device)
    op_igemm(xh, wt, B * H * H, Cout, Cin, taps=9, H=H, W=H, bias=bias, act=1, out_f32=out, out_bf16=outh, out_halo=1, precision=PREC_F16)
    torch.cuda.synchronize()
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=2e-4)
    torch.testing.assert_close(outh[:, 1:-1, 1:-1].float(), F.relu(ref), rtol=1.5e-3, atol=1.5e-3)


@pytest.mark.parametrize("B,res,ws,shift,heads", [(2, 64, 16, 8, 3), (3, 16, 16, 0, 12), (2, 8, 8, 0, 24), (1, 96, 24, 12, 4), (3, 12, 12, 0, 32)])
def test_window_attention_f16(gpu_device, B, res, ws, shift, heads):
    from soccdpt_amd.lib import PREC_F16, op_window_attention
    g = torch.Generator().manual_seed(res * 100 + shift + heads)
    C = heads * 32
    qkv = _hf(torch.randn(B * res * res, 3 * C, generator=g)).to(gpu_device)
    table = (16 * torch.sigmoid(torch.randn((2 * ws - 1) ** 2, heads, generator=g))).to(gpu_device)
    scale = (10.0 + 5 * torch.rand(heads, generator=g)).to(gpu_device)
    out = torch.empty(B * res * res, C, dtype=torch.float16, device=gpu_device)
    op_window_attention(qkv, table, scale, out, B, res, ws, shift, heads, precision=PREC_F16)
    torch.cuda.synchronize()
    ref = _attention_ref(qkv, table, scale, B, res, ws, shift, heads)
    err = (out.float() - ref).abs()
    # q-hat, k-hat and the softmax numerators are rounded to fp16 (2^-12 relative): 8x tighter than the bf16 kernel's bounds
    assert float(err.max()) < 8e-3 and float(err.mean()) < 8e-4, (float(err.max()), float(err.mean()))


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("B,res,ws,shift,heads", [(2, 64, 16, 8, 3), (2, 8, 8, 0, 24), (1, 96, 24, 12, 4), (2, 48, 24, 12, 8), (2, 24, 24, 0, 16), (3, 12, 12, 0, 32),
                                                   (1, 32, 8, 4, 3)])
def test_window_attention_f32(gpu_device, B, res, ws, shift, heads, precision):
    """Exact-f32 cosine window attention (MFMA f32): the 16 x 16 / 8 x 8 kernel, the streaming kernel of the 24 x 24 / 12 x 12 windows (ragged last
    key tile at 144 tokens, shifted and unshifted) and the generic fallback (shifted 8 x 8), f32 output and the x3 operand output of F16X3."""
    from soccdpt_amd.lib import PREC_F16X3, PREC_F32, op_window_attention, x3_decode
    g = torch.Generator().manual_seed(res * 100 + shift + heads)
    C = heads * 32
    qkv = torch.randn(B * res * res, 3 * C, generator=g).to(gpu_device)
    table = (16 * torch.sigmoid(torch.randn((2 * ws - 1) ** 2, heads, generator=g))).to(gpu_device)
    scale = (10.0 + 5 * torch.rand(heads, generator=g)).to(gpu_device)
    if precision == "f32":
        out = torch.empty(B * res * res, C, device=gpu_device)
        op_window_attention(qkv, table, scale, out, B, res, ws, shift, heads, precision=PREC_F32)
        got = out
    else:
        raw = torch.zeros(B * res * res * C * 2, dtype=torch.float16, device=gpu_device)
        op_window_attention(qkv, table, scale, raw, B, res, ws, shift, heads, precision=PREC_F16X3)
        got = x3_decode(raw.cpu(), (B * res * res, C)).float().to(gpu_device)
    torch.cuda.synchronize()
    ref = _attention_ref(qkv, table, scale, B, res, ws, shift, heads)
    err = (got - ref).abs()
    assert float(err.max()) < 2e-5 and float((got - ref).norm() / ref.norm()) < 3e-6, (float(err.max()), float((got - ref).norm() / ref.norm()))


# ---------------- exact-f32 parity mode (SOCCDPT_PREC_F32) ----------------
@pytest.mark.parametrize("M,N,K", [(300, 288, 96), (2048, 384, 1536), (512, 2304, 768)])
def test_igemm_linear_f32(gpu_device, M, N, K):
    from soccdpt_amd.lib import op_igemm
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(gpu_device)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(gpu_device)
    bias = torch.randn(N, generator=g).to(gpu_device)
    ref = (x.double() @ w.double().t() + bias.double()).float()
    out = torch.empty(M, N, device=gpu_device)
    outg = torch.empty(M, N, device=gpu_device)
    op_igemm(x, w, M, N, K, ldx=K, bias=bias, out_f32=out, act=2, out_bf16=outg, f32=1)
    torch.cuda.synchronize()
    torch.testing.assert_close(out, ref, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(outg, F.gelu(ref), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("B,H,Cin,Cout", [(2, 16, 256, 256), (3, 8, 96, 256), (1, 64, 128, 32)])
def test_igemm_conv3x3_f32(gpu_device, B, H, Cin, Cout):
    from soccdpt_amd.lib import op_igemm
    g = torch.Generator().manual_seed(B * 1000 + H + Cin + Cout)
    x = torch.randn(B, Cin, H, H, generator=g).to(gpu_device)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to(gpu_device)
    bias = torch.randn(Cout, generator=g).to(gpu_device)
    ref = F.conv2d(x.double(), w.double(), bias.double(), padding=1).permute(0, 2, 3, 1).float()
    xh = _halo(x.permute(0, 2, 3, 1).contiguous())
    wt = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
    out = torch.empty(B, H, H, Cout, device=gpu_device)
    outh = torch.zeros(B, H + 2, H + 2, Cout, device=gpu_device)
    op_igemm(xh, wt, B * H * H, Cout, Cin, taps=9, H=H, W=H, bias=bias, act=1, out_f32=out, out_bf16=outh, out_halo=1, f32=1)
    torch.cuda.synchronize()
    torch.testing.assert_close(out, ref, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(outh[:, 1:-1, 1:-1], F.relu(ref), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("M,C,prec,hook", [(640, 96, "bf16", False), (2 * 16 * 16, 96, "bf16", True), (300, 128, "f16", False), (1000, 192, "bf16", False),
                                           (2 * 8 * 8, 256, "f16", True), (64, 128, "bf16", False)])
def test_mlp_ln_fused(gpu_device, M, C, prec, hook):
    """soccdpt_op_mlp_ln: x += LayerNorm(fc2(GELU(fc1(x_op)))) in one launch against the same chain in torch fp32 on the rounded
    operands (hidden activation rounded to the operand format, as the unfused path stores it); ragged M, both formats, halo copy."""
    from soccdpt_amd.lib import PREC_BF16, PREC_F16, op_mlp_ln
    dt, pr = (torch.bfloat16, PREC_BF16) if prec == "bf16" else (torch.float16, PREC_F16)
    g = torch.Generator().manual_seed(M + C)
    x = torch.randn(M, C, generator=g)
    xop = x.to(dt).to(gpu_device)
    xf = x.clone().to(gpu_device)
    w1 = (torch.randn(4 * C, C, generator=g) / math.sqrt(C)).to(dt).to(gpu_device)
    b1 = (0.2 * torch.randn(4 * C, generator=g)).to(gpu_device)
    w2 = (torch.randn(C, 4 * C, generator=g) / math.sqrt(4 * C)).to(dt).to(gpu_device)
    b2 = (0.2 * torch.randn(C, generator=g)).to(gpu_device)
    lg = (1 + 0.3 * torch.randn(C, generator=g)).to(gpu_device)
    lb = (0.3 * torch.randn(C, generator=g)).to(gpu_device)
    hid = F.gelu(xop.float() @ w1.float().t() + b1).to(dt).float()
    ref = xf + F.layer_norm(hid @ w2.float().t() + b2, (C,), lg, lb, eps=1e-5)
    out_op = torch.zeros(M, C, dtype=dt, device=gpu_device)
    halo = None
    Hh = 0
    if hook:
        Hh = int(round(math.sqrt(M / 2)))
        halo = torch.zeros(2, Hh + 2, Hh + 2, C, dtype=dt, device=gpu_device)
    got = xf.clone()
    op_mlp_ln(xop, got, w1, b1, w2, b2, lg, lb, x_op_out=out_op, halo=halo, precision=pr, H=Hh, W=Hh)
    torch.cuda.synchronize()
    tol = 2e-2 if prec == "bf16" else 3e-3     # LayerNorm output is O(1); the hidden tile is rounded to 8 / 11 significand bits
    torch.testing.assert_close(got, ref, rtol=tol, atol=tol)
    assert float((got - ref).pow(2).sum().sqrt() / ref.pow(2).sum().sqrt()) < (4e-3 if prec == "bf16" else 5e-4)
    torch.testing.assert_close(out_op.float(), got.to(dt).float(), rtol=0, atol=0)
    if hook:
        torch.testing.assert_close(halo[:, 1:-1, 1:-1].reshape(M, C).float(), got.to(dt).float(), rtol=0, atol=0)
        assert float(halo[:, 0].abs().max()) == 0 and float(halo[:, :, -1].abs().max()) == 0
    # in place on the operand copy (how the network calls it)
    xop2, got2 = xop.clone(), xf.clone()
    op_mlp_ln(xop2, got2, w1, b1, w2, b2, lg, lb, x_op_out=xop2, precision=pr)
    torch.cuda.synchronize()
    assert torch.equal(got2, got) and torch.equal(xop2, out_op)


# ---------------- conv8p.hip: phase-interleaved big-tile 3x3 convolution (configuration ids 30 / 31 / 32) ----------------
@pytest.mark.parametrize("tune,Cout", [(30, 256), (31, 256), (32, 128), (30, 512), (32, 256)])
@pytest.mark.parametrize("prec", ["bf16", "f16"])
def test_conv8p_matches_torch(gpu_device, tune, Cout, prec):
    """256x256 / 128x256 / 256x128 tiles with the staggered read / matrix intervals: ragged M (1600 and 2 x 50 x 50 pixels are no tile
    multiples), 18 k-tiles (Cin 128) and 36 (Cin 256), every epilogue path (bias, two residuals, ReLU, f32 + halo operand stores), and
    repeated launches (a staging / fragment-read race would show as a sporadic wrong tile)."""
    from soccdpt_amd.lib import PREC_BF16, PREC_F16, op_igemm
    dt = torch.bfloat16 if prec == "bf16" else torch.float16
    P = PREC_BF16 if prec == "bf16" else PREC_F16
    for B, H, Cin in ((1, 40, 128), (2, 50, 256)):
        g = torch.Generator().manual_seed(tune * 100 + H)
        x = torch.randn(B, Cin, H, H, generator=g).to(dt).to(gpu_device)
        w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to(dt).to(gpu_device)
        bias = torch.randn(Cout, generator=g).to(gpu_device)
        res1 = torch.randn(B, H, H, Cout, generator=g).to(gpu_device)
        res2 = torch.randn(B, H, H, Cout, generator=g).to(gpu_device)
        ref = F.conv2d(x.float(), w.float(), bias, padding=1).permute(0, 2, 3, 1)
        full = ref + res1 + res2
        xh = torch.zeros(B, H + 2, H + 2, Cin, dtype=dt, device=gpu_device)
        xh[:, 1:-1, 1:-1] = x.permute(0, 2, 3, 1)
        wt = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
        out = torch.empty(B, H, H, Cout, device=gpu_device)
        outh = torch.zeros(B, H + 2, H + 2, Cout, dtype=dt, device=gpu_device)
        for rep in range(4):
            out.fill_(-7.0)
            op_igemm(xh, wt, B * H * H, Cout, Cin, taps=9, H=H, W=H, bias=bias, res1=res1, res2=res2, act=1, out_f32=out, out_bf16=outh, out_halo=1,
                     tune=tune, precision=P)
            torch.cuda.synchronize()
            torch.testing.assert_close(out, full, rtol=1e-4, atol=2e-4, msg=f"tune={tune} rep={rep}")
        torch.testing.assert_close(outh[:, 1:-1, 1:-1].float(), F.relu(full), rtol=1e-2, atol=1e-2)
        assert float(outh[:, 0].abs().max()) == 0 and float(outh[:, :, 0].abs().max()) == 0 and float(outh[:, -1].abs().max()) == 0
        # bit-identical to the 128 x 128 tile: same k order per output element (k-tiles of 64, two MFMA k-steps each)
        o2 = torch.empty_like(out)
        op_igemm(xh, wt, B * H * H, Cout, Cin, taps=9, H=H, W=H, bias=bias, res1=res1, res2=res2, act=1, out_f32=o2, tune=21, precision=P)
        torch.cuda.synchronize()
        assert torch.equal(o2, out), f"tune={tune}: differs from configuration 21"


# ---------------- weight gradient from operands as stored (csrc/train_wgrad_tn.hip: LDS transpose reads) ----------------
@pytest.mark.parametrize("prec", ["bf16", "f16", "f16x3"])
def test_wgrad_tn_linear(gpu_device, prec):
    """dW[n][c] = sum_k dY[k][n] X[k][c] with both operands row-major over k (no transposes).  Nout = 192 and C = 96 exercise the masked edge tiles
    (operands over-read past their rows), K = 1600 * ... a multiple of 64 that is not a multiple of the split count."""
    from soccdpt_amd.lib import PREC_BF16, PREC_F16, PREC_F16X3, op_wgrad_tn, x3_encode
    g = torch.Generator().manual_seed(5)
    for K, Nout, C in ((4096, 256, 384), (1600 * 4, 192, 96), (576, 128, 128)):
        dy = torch.randn(K, Nout, generator=g)
        x = torch.randn(K, C, generator=g)
        if prec == "bf16":
            dy, x = _bf(dy), _bf(x)
            a, b, code, tol = dy.to(gpu_device), x.to(gpu_device), PREC_BF16, 2e-5
        elif prec == "f16":
            dy, x = dy.half(), x.half()
            a, b, code, tol = dy.to(gpu_device), x.to(gpu_device), PREC_F16, 2e-5
        else:
            a, b, code, tol = x3_encode(dy.to(gpu_device)), x3_encode(x.to(gpu_device)), PREC_F16X3, 2e-6
        # the edge tiles over-read up to 127 columns past a row: give the flat tensors room behind the last row
        pad = torch.zeros(4096, dtype=a.dtype, device=gpu_device)
        a, b = torch.cat([a.reshape(-1), pad]), torch.cat([b.reshape(-1), pad])
        ref = (dy.double().t() @ x.double()).float().to(gpu_device)
        out = op_wgrad_tn(a, Nout, b, C, K, Nout, C, precision=code)
        out2 = op_wgrad_tn(a, Nout, b, C, K, Nout, C, precision=code)
        torch.cuda.synchronize()
        err = float((out - ref).norm() / ref.norm())
        assert err < tol, (prec, K, Nout, C, err)
        assert torch.equal(out, out2)          # deterministic split-K


@pytest.mark.parametrize("prec", ["bf16", "f16x3"])
def test_wgrad_tn_conv3x3(gpu_device, prec):
    """The 3x3 convolution's weight gradient in halo pixel order: tap (ky, kx) = the same GEMM over the input's halo image read (ky - 1)(r + 2) + (kx - 1) rows
    further on.  Against torch autograd of conv2d."""
    from soccdpt_amd.lib import PREC_BF16, PREC_F16X3, op_wgrad_tn, x3_encode
    g = torch.Generator().manual_seed(7)
    B, r, C, N = 2, 12, 128, 128
    x = torch.randn(B, C, r, r, generator=g)
    dy = torch.randn(B, N, r, r, generator=g)
    if prec == "bf16":
        x, dy = _bf(x).float(), _bf(dy).float()
    w = torch.zeros(N, C, 3, 3, requires_grad=True)
    F.conv2d(x, w, padding=1).backward(dy)
    ref = w.grad.permute(0, 2, 3, 1).reshape(N, 9 * C).contiguous().to(gpu_device)     # tap-major [N][9][C]
    rp = r + 2
    Kh = B * rp * rp
    Kp = (Kh + 63) // 64 * 64
    mrg = rp + 1
    xh = torch.zeros(Kp + 2 * mrg, C)
    xh[mrg:mrg + Kh] = _halo(x.permute(0, 2, 3, 1).contiguous().to(torch.float32)).reshape(Kh, C)
    dyh = torch.zeros(Kp, N)
    dyh[:Kh] = _halo(dy.permute(0, 2, 3, 1).contiguous().to(torch.float32)).reshape(Kh, N)
    if prec == "bf16":
        a, b, code, tol = dyh.to(torch.bfloat16).to(gpu_device).reshape(-1), xh.to(torch.bfloat16).to(gpu_device).reshape(-1), PREC_BF16, 2e-5
    else:
        a, b, code, tol = x3_encode(dyh.to(gpu_device)), x3_encode(xh.to(gpu_device)), PREC_F16X3, 2e-6
    out = op_wgrad_tn(a, N, b, C, Kp, N, C, taps=9, rp=rp, precision=code, b_row0=mrg * C)
    torch.cuda.synchronize()
    err = float((out - ref).norm() / ref.norm())
    assert err < tol, (prec, err)


def test_lds_transpose_read_semantics(gpu_device, tmp_path):
    """ds_read_b64_tr_b16 as train_wgrad_tn.hip uses it: per 16-lane group a 4-row x 16-column block of 16-bit elements, lane 4q + p addresses row q columns
    4p..4p+3, lane i receives column i of the four rows.  tests/tools/tr_read_probe.hip is compiled with the box's hipcc and run as a child process."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "tr_read_probe.hip")
    exe = str(tmp_path / "tr_read_probe")
    subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", src, "-o", exe], check=True, capture_output=True, timeout=300)
    out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=120).stdout
    assert "mismatches vs 'lane i gets column i of rows 4g..4g+3': 0" in out, out
