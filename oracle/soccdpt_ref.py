"""CPU oracle for the SOccDPT_V3 forward path (TEST INFRASTRUCTURE ONLY).

This module is a plain fp32 PyTorch-CPU restatement of the reference algorithm.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it; the product path (``soccdpt_amd``) never does.

Pinning status
--------------
* decoder + heads + projection: pinned bit-for-bit against the reference's own
  code run in the build container (``oracle/make_golden.py``; fixtures in
  ``tests/golden/``).
* Swin-V2 and ViT-hybrid encoders: **parity unpinned** at the timm boundary (the hybrid's adapters -- pos-embed resize, readout
  projection, hooks / unflatten / reassemble convs -- ARE the reference's code and are pinned by fixtures).  The arithmetic
  lives in ``timm==0.6.12`` (``/root/reference/requirements.txt:12``), which is
  neither vendored nor installed; the reference has no tests or golden vectors.
  The restatement below follows the published Swin-V2 algorithm as called from
  ``SOccDPT/model/backbones/swin2.py:24-30`` and is cross-checked against the
  independent HF ``transformers`` ports (``Swinv2Model``, DPT hybrid; tests/test_oracle_encoder.py, oracle/hf_crosscheck.py).

Every function cites the reference file:line it follows (paths relative to
``/root/reference/SOccDPT``).  The state-dict key layout is the reference's
(SURVEY.md §8b): ``depth_net.pretrained.model.*``, ``depth_net.scratch.*``,
``seg_head.*``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------
# Architecture table (model/dpt.py:51-89 hooks, model/blocks.py:59-78 channel
# lists, backbones/swin2.py:15-30 timm model names)
# ----------------------------------------------------------------------------
@dataclass(frozen=True)
class SwinArch:
    img: int
    patch: int
    embed: int
    depths: Tuple[int, ...]
    heads: Tuple[int, ...]
    window: int
    pretrained_window: Tuple[int, ...]
    hooks: Tuple[int, ...]

    @property
    def grid(self) -> int:
        return self.img // self.patch

    def dims(self) -> List[int]:
        return [self.embed << i for i in range(len(self.depths))]


ARCHS: Dict[str, SwinArch] = {
    # swinv2_tiny_window16_256
    "swin2t16_256": SwinArch(256, 4, 96, (2, 2, 6, 2), (3, 6, 12, 24), 16, (0, 0, 0, 0), (1, 1, 5, 1)),
    # swinv2_base_window12to24_192to384_22kft1k
    "swin2b24_384": SwinArch(384, 4, 128, (2, 2, 18, 2), (4, 8, 16, 32), 24, (12, 12, 12, 6), (1, 1, 17, 1)),
}

MODEL_TYPE_TO_BACKBONE = {  # model/loader.py:65-77,115-120
    "dpt_swin2_tiny_256": "swin2t16_256",
    "dpt_swin2_base_384": "swin2b24_384",
    "dpt_hybrid_384": "vitb_rn50_384",
}


@dataclass(frozen=True)
class HybridArch:
    """timm 0.6.12 `vit_base_resnet50_384` (= vit_base_r50_s16_384) as created by backbones/vit.py:244-258: ResNetV2 (3, 4, 9)
    stem/stages with weight-standardised SAME-padded convolutions + GroupNorm(32), 1x1 projection to 768, 12 pre-norm ViT-B blocks
    over 24 x 24 + 1 tokens; hooks [0, 1, 8, 11] (model/dpt.py:86), reassemble channels [256, 512, 768, 768] (model/blocks.py:103-112)."""
    img: int = 384
    embed: int = 768
    depth: int = 12
    heads: int = 12
    stem: int = 64
    layers: Tuple[int, ...] = (3, 4, 9)
    hooks: Tuple[int, ...] = (0, 1, 8, 11)
    features: Tuple[int, ...] = (256, 512, 768, 768)
    patch: int = 16

    @property
    def grid(self) -> int:
        return self.img // self.patch


HYBRID = HybridArch()


# ----------------------------------------------------------------------------
# Swin-V2 encoder  (timm 0.6.12 SwinTransformerV2)
# ----------------------------------------------------------------------------
def window_geometry(res: int, window: int, block_index: int) -> Tuple[int, int]:
    """Clamp window to the resolution; odd blocks shift by window//2 only when
    the resolution is larger than the window."""
    ws = min(res, window)
    shift = 0 if (block_index % 2 == 0 or res <= window) else window // 2
    return ws, shift


def cpb_coords_table(ws: int, pretrained_ws: int) -> Tensor:
    """log-spaced relative coordinates, [(2ws-1)^2, 2]."""
    r = torch.arange(-(ws - 1), ws, dtype=torch.float32)
    tab = torch.stack(torch.meshgrid(r, r, indexing="ij"), dim=-1)  # [2ws-1, 2ws-1, 2]
    denom = (pretrained_ws - 1) if pretrained_ws > 0 else (ws - 1)
    tab = tab / denom
    tab = tab * 8
    tab = torch.sign(tab) * torch.log2(torch.abs(tab) + 1.0) / math.log2(8)
    return tab.reshape(-1, 2)


def relative_position_index(ws: int) -> Tensor:
    """[ws*ws, ws*ws] index into the (2ws-1)^2 CPB table."""
    c = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)  # [2, N]
    rel = c[:, :, None] - c[:, None, :]  # [2, N, N]
    return (rel[0] + ws - 1) * (2 * ws - 1) + (rel[1] + ws - 1)


def cpb_bias(sd: Dict[str, Tensor], pfx: str, ws: int, pretrained_ws: int, heads: int) -> Tensor:
    """16*sigmoid(cpb_mlp(coords))[rel_index] -> [heads, N, N]."""
    tab = cpb_coords_table(ws, pretrained_ws).to(sd[pfx + "cpb_mlp.0.weight"].dtype)   # (f64 runs: the gradient-noise reference of the training tests)
    h = F.relu(F.linear(tab, sd[pfx + "cpb_mlp.0.weight"], sd[pfx + "cpb_mlp.0.bias"]))
    t = F.linear(h, sd[pfx + "cpb_mlp.2.weight"])  # [(2ws-1)^2, heads]
    idx = relative_position_index(ws).reshape(-1)
    n = ws * ws
    bias = t[idx].reshape(n, n, heads).permute(2, 0, 1).contiguous()
    return 16.0 * torch.sigmoid(bias)


def shift_attn_mask(res: int, ws: int, shift: int) -> Optional[Tensor]:
    """[nW, N, N] 0 / -100 mask of the cyclic-shift regions."""
    if shift == 0:
        return None
    idx = torch.arange(res)
    reg = (idx >= res - ws).long() + (idx >= res - shift).long()
    img = (reg[:, None] * 3 + reg[None, :]).float()  # [res, res]
    nw = res // ws
    win = img.reshape(nw, ws, nw, ws).permute(0, 2, 1, 3).reshape(nw * nw, ws * ws)
    diff = win[:, None, :] - win[:, :, None]
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


def window_attention(sd, pfx, x: Tensor, res: int, ws: int, shift: int, heads: int, pretrained_ws: int) -> Tensor:
    """x [B, L, C] -> attention output (after proj) [B, L, C]."""
    B, L, C = x.shape
    d = C // heads
    nw = res // ws
    N = ws * ws
    h = x.reshape(B, res, res, C)
    if shift > 0:
        h = torch.roll(h, shifts=(-shift, -shift), dims=(1, 2))
    win = h.reshape(B, nw, ws, nw, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B * nw * nw, N, C)

    qkv_bias = torch.cat([sd[pfx + "q_bias"], torch.zeros_like(sd[pfx + "v_bias"]), sd[pfx + "v_bias"]])
    qkv = F.linear(win, sd[pfx + "qkv.weight"], qkv_bias)
    qkv = qkv.reshape(-1, N, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]  # [Bw, heads, N, d]

    attn = F.normalize(q, dim=-1) @ F.normalize(k, dim=-1).transpose(-2, -1)
    scale = torch.clamp(sd[pfx + "logit_scale"], max=math.log(1.0 / 0.01)).exp()
    attn = attn * scale.reshape(1, heads, 1, 1)
    attn = attn + cpb_bias(sd, pfx, ws, pretrained_ws, heads).unsqueeze(0)
    mask = shift_attn_mask(res, ws, shift)
    if mask is not None:
        attn = attn.reshape(B, nw * nw, heads, N, N) + mask.to(attn.dtype)[None, :, None]
        attn = attn.reshape(-1, heads, N, N)
    attn = torch.softmax(attn, dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(-1, N, C)
    out = F.linear(out, sd[pfx + "proj.weight"], sd[pfx + "proj.bias"])

    out = out.reshape(B, nw, nw, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, res, res, C)
    if shift > 0:
        out = torch.roll(out, shifts=(shift, shift), dims=(1, 2))
    return out.reshape(B, L, C)


def swin_block(sd, pfx, x, res, ws, shift, heads, pretrained_ws, drop_path=None) -> Tensor:
    """Residual post-norm block: x + LN(attn(x)); then + LN(mlp(.)).  drop_path = (s1 [B], s2 [B]): the per-sample scales timm's DropPath applies
    to the two normalised branches in train mode (0 or 1 / keep_prob; SwinTransformerV2Block.forward: x + drop_path1(norm1(attn(x))), ...)."""
    C = x.shape[-1]
    a = window_attention(sd, pfx + "attn.", x, res, ws, shift, heads, pretrained_ws)
    a = F.layer_norm(a, (C,), sd[pfx + "norm1.weight"], sd[pfx + "norm1.bias"], 1e-5)
    if drop_path is not None:
        a = a * drop_path[0].view(-1, 1, 1)
    x = x + a
    m = F.linear(x, sd[pfx + "mlp.fc1.weight"], sd[pfx + "mlp.fc1.bias"])
    m = F.gelu(m)
    m = F.linear(m, sd[pfx + "mlp.fc2.weight"], sd[pfx + "mlp.fc2.bias"])
    m = F.layer_norm(m, (C,), sd[pfx + "norm2.weight"], sd[pfx + "norm2.bias"], 1e-5)
    if drop_path is not None:
        m = m * drop_path[1].view(-1, 1, 1)
    return x + m


def patch_merging(sd, pfx, x, res) -> Tensor:
    B, L, C = x.shape
    h = x.reshape(B, res, res, C)
    h = torch.cat([h[:, 0::2, 0::2], h[:, 1::2, 0::2], h[:, 0::2, 1::2], h[:, 1::2, 1::2]], dim=-1)
    h = h.reshape(B, L // 4, 4 * C)
    h = F.linear(h, sd[pfx + "reduction.weight"])
    return F.layer_norm(h, (2 * C,), sd[pfx + "norm.weight"], sd[pfx + "norm.bias"], 1e-5)


def swin_encoder(sd: Dict[str, Tensor], x: Tensor, arch: SwinArch, pfx: str = "depth_net.pretrained.model.", drop_path=None) -> List[Tensor]:
    """forward_features with hooks on layers[i].blocks[hooks[i]]
    (backbones/swin_common.py:12-54, backbones/utils.py:54-81).
    Returns the four hooked tensors as NCHW maps."""
    B = x.shape[0]
    t = F.conv2d(x, sd[pfx + "patch_embed.proj.weight"], sd[pfx + "patch_embed.proj.bias"], stride=arch.patch)
    t = t.flatten(2).transpose(1, 2)  # [B, L, C]
    C0 = arch.embed
    t = F.layer_norm(t, (C0,), sd[pfx + "patch_embed.norm.weight"], sd[pfx + "patch_embed.norm.bias"], 1e-5)
    res = arch.grid
    outs = []
    for s, depth in enumerate(arch.depths):
        for j in range(depth):
            ws, shift = window_geometry(res, arch.window, j)
            t = swin_block(sd, f"{pfx}layers.{s}.blocks.{j}.", t, res, ws, shift, arch.heads[s], arch.pretrained_window[s],
                           drop_path=None if drop_path is None else drop_path[(s, j)])   # {(stage, block): (s1 [B], s2 [B])}
            if j == arch.hooks[s]:
                C = t.shape[-1]
                outs.append(t.transpose(1, 2).reshape(B, C, res, res))
        if s < len(arch.depths) - 1:
            t = patch_merging(sd, f"{pfx}layers.{s}.downsample.", t, res)
            res //= 2
    return outs


# ----------------------------------------------------------------------------
# ViT-hybrid encoder.  PARITY UNPINNED at the timm boundary, like the Swin-V2 encoder: the ResNetV2
# and VisionTransformer arithmetic lives in timm==0.6.12 (requirements.txt:12; call site backbones/vit.py:244-258), not vendored,
# not installable; restated from the published architectures and cross-checked against HF transformers' independent port
# (DPT hybrid = BiT backbone + ViT; oracle/hf_crosscheck.py, tests/test_oracle_encoder.py).  The ADAPTERS around it are the
# reference's own code and are pinned by fixtures generated from it (oracle/make_golden.py): forward_flex / _resize_pos_embed
# (backbones/vit.py:23-85), forward_adapted_unflatten + ProjectReadout + act_postprocess (backbones/utils.py:27-40,84-133,
# backbones/vit.py:147-231).  backbones/vit.py:181-182,222-223 are broken in the snapshot (`_ = nn.Sequential(...)` followed by
# exec("...=value")): the evident upstream-MiDaS intent `value = nn.Sequential(...)` is followed.
# ----------------------------------------------------------------------------
def pad_same(x: Tensor, k: int, s: int, value: float = 0.0) -> Tensor:
    """TF-style SAME padding (timm padding.pad_same): total = max((ceil(i/s)-1)*s + k - i, 0), the extra pixel goes right/bottom."""
    ih, iw = x.shape[-2:]
    ph = max((math.ceil(ih / s) - 1) * s + k - ih, 0)
    pw = max((math.ceil(iw / s) - 1) * s + k - iw, 0)
    if ph > 0 or pw > 0:
        x = F.pad(x, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2], value=value)
    return x


def std_conv_same(x: Tensor, w: Tensor, stride: int = 1, eps: float = 1e-8) -> Tensor:
    """timm StdConv2dSame(bias=False, eps=1e-8): per-output-channel weight standardisation (biased variance, evaluated through
    F.batch_norm exactly like timm) and 'SAME' padding -- static symmetric padding when stride == 1, dynamic TF padding otherwise."""
    k = w.shape[-1]
    ws = F.batch_norm(w.reshape(1, w.shape[0], -1), None, None, training=True, momentum=0.0, eps=eps).reshape_as(w)
    if stride == 1:
        return F.conv2d(x, ws, None, 1, (k - 1) // 2)
    return F.conv2d(pad_same(x, k, stride), ws, None, stride, 0)


def group_norm_act(x: Tensor, w: Tensor, b: Tensor, relu: bool = True) -> Tensor:
    """timm GroupNormAct(num_groups=32, eps=1e-5) (+ ReLU)."""
    y = F.group_norm(x, 32, w, b, 1e-5)
    return F.relu(y) if relu else y


def resnetv2_bottleneck(sd, p: str, x: Tensor, stride: int) -> Tensor:
    """timm resnetv2.Bottleneck (non-pre-activation, the ViT-hybrid form): 1x1 -> GN+ReLU -> 3x3 (stride) -> GN+ReLU -> 1x1 -> GN,
    shortcut = GN(1x1 stride conv) on the first block of a stage, ReLU after the sum."""
    sc = x
    if p + "downsample.conv.weight" in sd:
        sc = std_conv_same(x, sd[p + "downsample.conv.weight"], stride)
        sc = group_norm_act(sc, sd[p + "downsample.norm.weight"], sd[p + "downsample.norm.bias"], relu=False)
    y = std_conv_same(x, sd[p + "conv1.weight"])
    y = group_norm_act(y, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
    y = std_conv_same(y, sd[p + "conv2.weight"], stride)
    y = group_norm_act(y, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    y = std_conv_same(y, sd[p + "conv3.weight"])
    y = group_norm_act(y, sd[p + "norm3.weight"], sd[p + "norm3.bias"], relu=False)
    return F.relu(y + sc)


def resnetv2_backbone(sd, p: str, x: Tensor, arch: HybridArch = HYBRID) -> List[Tensor]:
    """timm ResNetV2(layers=(3,4,9), preact=False, stem_type='same', conv_layer=StdConv2dSame(eps=1e-8)): outputs of the three stages
    ([B,256,96,96], [B,512,48,48], [B,1024,24,24] at 384 x 384)."""
    y = std_conv_same(x, sd[p + "stem.conv.weight"], 2)
    y = group_norm_act(y, sd[p + "stem.norm.weight"], sd[p + "stem.norm.bias"])
    y = F.max_pool2d(pad_same(y, 3, 2, value=float("-inf")), 3, 2)     # MaxPool2dSame
    outs = []
    for s, depth in enumerate(arch.layers):
        for j in range(depth):
            y = resnetv2_bottleneck(sd, f"{p}stages.{s}.blocks.{j}.", y, stride=(2 if (s > 0 and j == 0) else 1))
        outs.append(y)
    return outs


def resize_pos_embed(posemb: Tensor, gs_h: int, gs_w: int, start_index: int = 1) -> Tensor:
    """backbones/vit.py:23-41 (_resize_pos_embed)."""
    tok, grid = posemb[:, :start_index], posemb[0, start_index:]
    gs_old = int(math.sqrt(len(grid)))
    grid = grid.reshape(1, gs_old, gs_old, -1).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, size=(gs_h, gs_w), mode="bilinear")
    grid = grid.permute(0, 2, 3, 1).reshape(1, gs_h * gs_w, -1)
    return torch.cat([tok, grid], dim=1)


def vit_block(sd, p: str, x: Tensor, heads: int) -> Tensor:
    """timm vision_transformer.Block (pre-norm, LayerNorm eps 1e-6, no layer-scale, erf GELU)."""
    B, N, C = x.shape
    h = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)
    qkv = F.linear(h, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]).reshape(B, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    a = (q @ k.transpose(-2, -1)) * ((C // heads) ** -0.5)
    a = a.softmax(dim=-1)
    h = (a @ v).transpose(1, 2).reshape(B, N, C)
    x = x + F.linear(h, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
    h = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)
    h = F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
    return x + F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])


def project_readout(sd, p: str, x: Tensor, start_index: int = 1) -> Tensor:
    """backbones/utils.py:27-40 ProjectReadout: Linear(2C, C) + GELU on cat(token, cls)."""
    readout = x[:, 0].unsqueeze(1).expand_as(x[:, start_index:])
    feats = torch.cat((x[:, start_index:], readout), -1)
    return F.gelu(F.linear(feats, sd[p + "project.0.weight"], sd[p + "project.0.bias"]))


def hybrid_encoder(sd: Dict[str, Tensor], x: Tensor, arch: HybridArch = HYBRID, pfx: str = "depth_net.pretrained.") -> List[Tensor]:
    """forward_vit -> forward_adapted_unflatten(pretrained, x, "forward_flex") (backbones/vit.py:19-85, backbones/utils.py:84-133)
    with hooks on patch_embed.backbone.stages[0,1] and blocks[8,11] (vit.py:164-171).  Returns the four NCHW maps handed to
    scratch.layerN_rn: [B,256,96,96], [B,512,48,48], [B,768,24,24], [B,768,12,12]."""
    B, _, H, W = x.shape
    m = pfx + "model."
    stages = resnetv2_backbone(sd, m + "patch_embed.backbone.", x, arch)
    gh, gw = H // arch.patch, W // arch.patch
    pos = resize_pos_embed(sd[m + "pos_embed"], gh, gw)
    t = F.conv2d(stages[-1], sd[m + "patch_embed.proj.weight"], sd[m + "patch_embed.proj.bias"]).flatten(2).transpose(1, 2)
    t = torch.cat((sd[m + "cls_token"].expand(B, -1, -1), t), dim=1) + pos            # no_embed_class=False: cat first, then add
    hooked = {}
    for i in range(arch.depth):
        t = vit_block(sd, f"{m}blocks.{i}.", t, arch.heads)
        if i in arch.hooks[2:]:
            hooked[i] = t
    # (the final model.norm is dead on this path: both hooks fire before it)
    l1, l2 = stages[0], stages[1]                                   # act_postprocess1/2 = Identity x 3 (vit.py:179-182)
    outs = [l1, l2]
    for n, hk in ((3, arch.hooks[2]), (4, arch.hooks[3])):
        ap = f"{pfx}act_postprocess{n}."
        y = project_readout(sd, ap + "0.", hooked[hk]).transpose(1, 2).reshape(B, arch.embed, gh, gw)   # [0:2] + unflatten
        y = F.conv2d(y, sd[ap + "3.weight"], sd[ap + "3.bias"])                                           # Conv2d(768, features[s], 1)
        if n == 4:
            y = F.conv2d(y, sd[ap + "4.weight"], sd[ap + "4.bias"], stride=2, padding=1)                  # Conv2d(768, 768, 3, 2, 1)
        outs.append(y)
    return outs


# ----------------------------------------------------------------------------
# DPT decoder + heads (model/dpt.py:142-232, model/blocks.py:391-497,
# model/SOccDPT.py:655-685)
# ----------------------------------------------------------------------------
def rcu(sd, pfx, x: Tensor) -> Tensor:
    """ResidualConvUnit_custom (model/blocks.py:391-414), bn=False."""
    out = F.relu(x)
    out = F.conv2d(out, sd[pfx + "conv1.weight"], sd[pfx + "conv1.bias"], padding=1)
    out = F.relu(out)
    out = F.conv2d(out, sd[pfx + "conv2.weight"], sd[pfx + "conv2.bias"], padding=1)
    return out + x


def fusion_block(sd, pfx, xs: Sequence[Tensor], size=None) -> Tensor:
    """FeatureFusionBlock_custom.forward (model/blocks.py:466-497)."""
    out = xs[0]
    if len(xs) == 2:
        out = out + rcu(sd, pfx + "resConfUnit1.", xs[1])
    out = rcu(sd, pfx + "resConfUnit2.", out)
    if size is None:
        out = F.interpolate(out, scale_factor=2, mode="bilinear", align_corners=True)
    else:
        out = F.interpolate(out, size=size, mode="bilinear", align_corners=True)
    return F.conv2d(out, sd[pfx + "out_conv.weight"], sd[pfx + "out_conv.bias"])


def dpt_decoder(sd, layers: Sequence[Tensor], pfx: str = "depth_net.scratch.") -> Tuple[Tensor, Tensor]:
    """DPT.forward after the encoder + depth head (model/dpt.py:152-182,199-232).
    Returns (inv_depth [B,H,W], path_1 [B,256,H/2,W/2])."""
    l1, l2, l3, l4 = layers
    l1r = F.conv2d(l1, sd[pfx + "layer1_rn.weight"], padding=1)
    l2r = F.conv2d(l2, sd[pfx + "layer2_rn.weight"], padding=1)
    l3r = F.conv2d(l3, sd[pfx + "layer3_rn.weight"], padding=1)
    l4r = F.conv2d(l4, sd[pfx + "layer4_rn.weight"], padding=1)
    p4 = fusion_block(sd, pfx + "refinenet4.", [l4r], size=l3r.shape[2:])
    p3 = fusion_block(sd, pfx + "refinenet3.", [p4, l3r], size=l2r.shape[2:])
    p2 = fusion_block(sd, pfx + "refinenet2.", [p3, l2r], size=l1r.shape[2:])
    p1 = fusion_block(sd, pfx + "refinenet1.", [p2, l1r])
    h = F.conv2d(p1, sd[pfx + "output_conv.0.weight"], sd[pfx + "output_conv.0.bias"], padding=1)
    h = F.interpolate(h, scale_factor=2, mode="bilinear", align_corners=True)
    h = F.conv2d(h, sd[pfx + "output_conv.2.weight"], sd[pfx + "output_conv.2.bias"], padding=1)
    h = F.relu(h)
    h = F.conv2d(h, sd[pfx + "output_conv.4.weight"], sd[pfx + "output_conv.4.bias"])
    h = F.relu(h)
    return h.squeeze(1), p1


def seg_logits(sd, feats: Tensor, pfx: str = "seg_head.", training: bool = False, dropout_p: float = 0.0) -> Tensor:
    """Class logits of the seg head at half resolution: Conv3x3 -> BN -> ReLU -> Dropout -> Conv1x1
    (model/SOccDPT.py:660-671), i.e. seg_head before Interpolate and the activation.  training=True is nn.Module.train():
    BatchNorm2d normalises with batch statistics and updates running_mean / running_var in place (momentum 0.1), Dropout(dropout_p) is live."""
    h = F.conv2d(feats, sd[pfx + "0.weight"], padding=1)
    h = F.batch_norm(h, sd[pfx + "1.running_mean"], sd[pfx + "1.running_var"],
                     sd[pfx + "1.weight"], sd[pfx + "1.bias"], training, 0.1, 1e-5)
    h = F.relu(h)
    if training and dropout_p > 0.0:
        h = F.dropout(h, dropout_p, True)
    return F.conv2d(h, sd[pfx + "4.weight"], sd[pfx + "4.bias"])


def seg_head(sd, feats: Tensor, sigmoid: bool, pfx: str = "seg_head.", training: bool = False, dropout_p: float = 0.0) -> Tensor:
    """Seg head (model/SOccDPT.py:660-674; ScaledTanh model/scaled_tanh.py:4-10); eval mode unless training=True."""
    h = seg_logits(sd, feats, pfx, training, dropout_p)
    h = F.interpolate(h, scale_factor=2, mode="bilinear", align_corners=True)
    if sigmoid:
        return torch.sigmoid(h)
    return 0.5 * torch.tanh(h) + 0.5


# ----------------------------------------------------------------------------
# Projection (model/SOccDPT.py:60-130, 264-463)
# ----------------------------------------------------------------------------
@dataclass(frozen=True)
class Camera:
    """Synthetic calibration (media/manydepth/intrinsics.json)."""
    fx: float = 1250.6
    fy: float = 1254.8
    cx: float = 978.4
    cy: float = 562.1
    width: int = 1920
    height: int = 1080


@dataclass(frozen=True)
class ProjConfig:
    grid_size: Tuple[int, int, int] = (256, 256, 32)
    scale: Tuple[float, float, float] = (2.0, 2.0, 0.666)
    pc_scale: Tuple[float, float, float] = (10000.0, 50000.0, 800.0)
    pc_shift: Tuple[float, float, float] = (55.0, -20.0, 15.0)
    correction_angle: Tuple[float, float, float] = (7.0, 0.0, 0.0)
    num_classes: int = 3

    def occupancy_shape(self) -> np.ndarray:
        # model/SOccDPT.py:175-181
        return np.array([float(self.grid_size[i] / self.scale[i]) for i in range(3)], dtype=np.float32)


def rotation_matrices(angles_deg: Sequence[float]) -> Tuple[Tensor, Tensor, Tensor]:
    """model/SOccDPT.py:74-111: f32 rotation matrices about x, y, z."""
    a, b, c = [torch.deg2rad(torch.tensor(float(v), dtype=torch.float32)) for v in angles_deg]
    ra = torch.tensor([[1, 0, 0], [0, torch.cos(a), -torch.sin(a)], [0, torch.sin(a), torch.cos(a)]], dtype=torch.float32)
    rb = torch.tensor([[torch.cos(b), 0, torch.sin(b)], [0, 1, 0], [-torch.sin(b), 0, torch.cos(b)]], dtype=torch.float32)
    rc = torch.tensor([[torch.cos(c), -torch.sin(c), 0], [torch.sin(c), torch.cos(c), 0], [0, 0, 1]], dtype=torch.float32)
    return ra, rb, rc


def project(inv_depth: Tensor, seg: Tensor, cam: Camera = Camera(), cfg: ProjConfig = ProjConfig(),
            compute_occ: bool = True):
    """get_semantic_occupancy + rotate_points + points_to_occupancy_grid
    (model/SOccDPT.py:264-463), same ATen op order as the reference so the CPU
    result is bit-identical to it.  Inputs are not modified."""
    Hc, Wc = cam.height, cam.width
    if inv_depth.dim() == 3:
        inv_depth = inv_depth.unsqueeze(1)
    inv_up = F.interpolate(inv_depth, size=(Hc, Wc), mode="bicubic", align_corners=False).squeeze()
    seg_up = F.interpolate(seg, size=(Hc, Wc), mode="nearest").squeeze()
    if inv_up.dim() == 2:
        inv_up = inv_up.unsqueeze(0)
    inv_up = inv_up.clone()
    inv_up[inv_up < 1e-8] = 1e-8
    depth = 1.0 / inv_up
    depth[torch.isinf(depth)] = float("inf")
    depth[torch.isnan(depth)] = float("inf")
    B = inv_up.shape[0]
    U, V = torch.meshgrid(torch.arange(Hc, dtype=torch.float32), torch.arange(Wc, dtype=torch.float32), indexing="ij")
    U = U.unsqueeze(0).repeat(B, 1, 1)
    V = V.unsqueeze(0).repeat(B, 1, 1)
    X = (V - cam.cx) * depth / cam.fx
    Y = (U - cam.cy) * depth / cam.fy
    Z = depth
    points = torch.stack([X, Y, Z], dim=3)
    sem = seg_up.reshape(-1, cfg.num_classes, Hc * Wc).permute(0, 2, 1)
    p3 = points.reshape(-1, Hc * Wc, cfg.num_classes)  # view: the quirk writes through to `points`
    for n in range(3):  # model/SOccDPT.py:351-353 (indexes the POINT axis)
        p3[:, n] = p3[:, n] * cfg.pc_scale[n] + cfg.pc_shift[n]
    ra, rb, rc = rotation_matrices(cfg.correction_angle)
    rot = torch.einsum("bnm,mj->bnj", p3, ra)
    rot = torch.einsum("bnm,mj->bnj", rot, rb)
    rot = torch.einsum("bnm,mj->bnj", rot, rc)
    occ = None
    if compute_occ:
        occ = points_to_occupancy(rot, sem, cfg)
    return inv_up, seg_up, points, occ


def points_to_occupancy(points: Tensor, sem: Tensor, cfg: ProjConfig = ProjConfig()) -> Tensor:
    """model/SOccDPT.py:374-463."""
    B = sem.shape[0]
    g = cfg.grid_size
    occ = torch.zeros((B, g[0], g[1], g[2], cfg.num_classes), dtype=torch.float32)
    ok = (~torch.isinf(points).any(dim=-1)) & (~torch.isnan(points).any(dim=-1))
    pts = torch.masked_select(points, ok.unsqueeze(-1)).reshape(-1, 3)
    sm = torch.masked_select(sem, ok.unsqueeze(-1)).reshape(-1, cfg.num_classes)
    oshape = torch.tensor(cfg.occupancy_shape()).to(dtype=torch.float32)
    gsz = torch.tensor(g).to(dtype=torch.float32)
    ijk = (pts / oshape * gsz).type(torch.int64)
    inb = ((0 < ijk[..., 0]) & (ijk[..., 0] < g[0]) & (0 < ijk[..., 1]) & (ijk[..., 1] < g[1])
           & (0 < ijk[..., 2]) & (ijk[..., 2] < g[2]))
    ijk = torch.masked_select(ijk, inb.unsqueeze(-1)).reshape(-1, 3)
    sm = torch.masked_select(sm, inb.unsqueeze(-1)).reshape(-1, cfg.num_classes)
    nz = sm.nonzero(as_tuple=False)
    idx = torch.cat([ijk[nz[:, 0]], nz[:, 1].view(-1, 1)], dim=1)
    occ[:, idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]] += 1
    return occ


# ----------------------------------------------------------------------------
# Whole forward (model/SOccDPT.py:681-685)
# ----------------------------------------------------------------------------
def soccdpt_v3_network(sd, x: Tensor, backbone: str = "swin2t16_256", sigmoid: bool = True, training: bool = False,
                       dropout_p: float = 0.0):
    """Encoder + decoder + heads only: (inv_depth [B,H,W], seg [B,C,H,W], path_1).  training=True: the train-mode forward the
    reference's training loop differentiates (scripts/train_SOccDPT.py:360-393) -- torch autograd over this function is the
    gradient oracle of tests/test_train_step_gpu.py."""
    layers = hybrid_encoder(sd, x) if backbone == "vitb_rn50_384" else swin_encoder(sd, x, ARCHS[backbone])
    inv, p1 = dpt_decoder(sd, layers)
    seg = seg_head(sd, p1, sigmoid, training=training, dropout_p=dropout_p)
    return inv, seg, p1


def soccdpt_v3_forward(sd, x: Tensor, backbone: str = "swin2t16_256", sigmoid: bool = True,
                       cam: Camera = Camera(), cfg: ProjConfig = ProjConfig(), compute_occ: bool = True):
    """Full SOccDPT_V3.forward: (inv_depth_up, seg_up, points, occupancy|None)."""
    with torch.no_grad():
        inv, seg, _ = soccdpt_v3_network(sd, x, backbone, sigmoid)
        return project(inv, seg, cam, cfg, compute_occ)
