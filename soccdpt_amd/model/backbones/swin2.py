"""Swin-V2 parameter tree with timm-0.6.12 names (what the reference creates through
timm.create_model in /root/reference/SOccDPT/model/backbones/swin2.py:15-30 and hooks in
backbones/swin_common.py:12-54).  These modules only HOLD parameters so that
state_dict()/load_state_dict()/parameters() behave like the reference's; the arithmetic runs
in libsoccdpt_hip.so (patch-embed, window-attention, igemm and LayerNorm kernels)."""
import math

import torch
import torch.nn as nn

from ..spec import SWIN_ARCHS, SwinV2Arch


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder: the Swin-V2 encoder runs inside libsoccdpt_hip.so")


class WindowAttentionParams(_Holder):
    def __init__(self, dim, heads):
        super().__init__()
        self.logit_scale = nn.Parameter(torch.log(10 * torch.ones((heads, 1, 1))))
        self.q_bias = nn.Parameter(torch.zeros(dim))
        self.v_bias = nn.Parameter(torch.zeros(dim))
        self.cpb_mlp = nn.Sequential(nn.Linear(2, 512, bias=True), nn.ReLU(inplace=True), nn.Linear(512, heads, bias=False))
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        self.proj = nn.Linear(dim, dim)


class MlpParams(_Holder):
    def __init__(self, dim):
        super().__init__()
        self.fc1 = nn.Linear(dim, 4 * dim)
        self.fc2 = nn.Linear(4 * dim, dim)


class BlockParams(_Holder):
    def __init__(self, dim, heads):
        super().__init__()
        self.attn = WindowAttentionParams(dim, heads)
        self.norm1 = nn.LayerNorm(dim)
        self.mlp = MlpParams(dim)
        self.norm2 = nn.LayerNorm(dim)


class PatchMergingParams(_Holder):
    def __init__(self, dim):
        super().__init__()
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(2 * dim)


class StageParams(_Holder):
    def __init__(self, dim, depth, heads, downsample):
        super().__init__()
        self.blocks = nn.ModuleList([BlockParams(dim, heads) for _ in range(depth)])
        if downsample:
            self.downsample = PatchMergingParams(dim)


class PatchEmbedParams(_Holder):
    def __init__(self, arch: SwinV2Arch):
        super().__init__()
        self.proj = nn.Conv2d(3, arch.embed, kernel_size=arch.patch, stride=arch.patch)
        self.norm = nn.LayerNorm(arch.embed)


class SwinTransformerV2Params(_Holder):
    """Same registration order as timm 0.6.12 SwinTransformerV2: patch_embed, layers, norm, head."""

    def __init__(self, arch: SwinV2Arch):
        super().__init__()
        self.arch = arch
        self.patch_embed = PatchEmbedParams(arch)
        n = len(arch.depths)
        self.layers = nn.ModuleList([
            StageParams(arch.embed << s, arch.depths[s], arch.heads[s], downsample=(s < n - 1)) for s in range(n)])
        self.norm = nn.LayerNorm(arch.embed << (n - 1))   # dead on the DPT path (hooks fire earlier)
        self.head = nn.Linear(arch.embed << (n - 1), 1000)  # dead on the DPT path


class SwinBackbone(_Holder):
    """`pretrained` of the reference: `.model` is the timm network (backbones/swin_common.py:13-15)."""

    def __init__(self, model: SwinTransformerV2Params, hooks):
        super().__init__()
        self.model = model
        self.hooks = list(hooks)


def _make_pretrained_swin2t16_256(pretrained, hooks=None):
    assert not pretrained, "no network access: load weights through load_net / load_state_dict"
    arch = SWIN_ARCHS["swin2t16_256"]
    return SwinBackbone(SwinTransformerV2Params(arch), arch.hooks if hooks is None else hooks)


def _make_pretrained_swin2b24_384(pretrained, hooks=None):
    assert not pretrained, "no network access: load weights through load_net / load_state_dict"
    arch = SWIN_ARCHS["swin2b24_384"]
    return SwinBackbone(SwinTransformerV2Params(arch), arch.hooks if hooks is None else hooks)
