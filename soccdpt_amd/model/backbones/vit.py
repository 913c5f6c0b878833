"""ViT-hybrid (vitb_rn50_384) parameter tree with the reference's names: what
/root/reference/SOccDPT/model/backbones/vit.py:147-258 (_make_vit_b_rn50_backbone / _make_pretrained_vitb_rn50_384) builds around
timm 0.6.12's `vit_base_resnet50_384` -- `pretrained.model` (VisionTransformer with a HybridEmbed: ResNetV2 (3, 4, 9) backbone of
weight-standardised convolutions + GroupNorm, 1x1 projection, 12 pre-norm blocks), parameter-free `act_postprocess1/2` and the
readout-projection + reassemble convs `act_postprocess3/4` (backbones/utils.py:27-40 ProjectReadout).  These modules only HOLD
parameters so that state_dict() / load_state_dict() / parameters() behave like the reference's; the arithmetic runs in
libsoccdpt_hip.so (csrc/hybrid.hip, igemm.hip, vit_attention.hip).  The snapshot's constructor is broken (vit.py:181-182,222-223:
`_ = nn.Sequential(...)` then exec("...=value")); the evident upstream-MiDaS intent is followed."""
import torch
import torch.nn as nn

from ..spec import HYBRID_ARCHS, HybridArch


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder: the ViT-hybrid encoder runs inside libsoccdpt_hip.so")


class StdConv2dSameParams(_Holder):
    def __init__(self, cin, cout, k):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k).normal_(std=(1.0 / (cin * k * k)) ** 0.5))


class GroupNormActParams(_Holder):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))


class DownsampleConvParams(_Holder):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = StdConv2dSameParams(cin, cout, 1)
        self.norm = GroupNormActParams(cout)


class BottleneckParams(_Holder):
    """timm resnetv2.Bottleneck registration order: downsample (first block of a stage), conv1, norm1, conv2, norm2, conv3, norm3."""

    def __init__(self, cin, cout, proj):
        super().__init__()
        mid = cout // 4
        if proj:
            self.downsample = DownsampleConvParams(cin, cout)
        self.conv1, self.norm1 = StdConv2dSameParams(cin, mid, 1), GroupNormActParams(mid)
        self.conv2, self.norm2 = StdConv2dSameParams(mid, mid, 3), GroupNormActParams(mid)
        self.conv3, self.norm3 = StdConv2dSameParams(mid, cout, 1), GroupNormActParams(cout)


class ResNetStageParams(_Holder):
    def __init__(self, cin, cout, depth):
        super().__init__()
        self.blocks = nn.Sequential(*[BottleneckParams(cin if j == 0 else cout, cout, j == 0) for j in range(depth)])


class ResNetV2Params(_Holder):
    def __init__(self, arch: HybridArch):
        super().__init__()
        self.stem = _Holder()
        self.stem.conv = StdConv2dSameParams(3, arch.stem, 7)
        self.stem.norm = GroupNormActParams(arch.stem)
        stages, prev = [], arch.stem
        for s, depth in enumerate(arch.layers):
            stages.append(ResNetStageParams(prev, 256 << s, depth))
            prev = 256 << s
        self.stages = nn.Sequential(*stages)
        self.num_features = prev


class HybridEmbedParams(_Holder):
    def __init__(self, arch: HybridArch):
        super().__init__()
        self.backbone = ResNetV2Params(arch)
        self.proj = nn.Conv2d(self.backbone.num_features, arch.embed, kernel_size=1, stride=1)


class VitAttentionParams(_Holder):
    def __init__(self, dim):
        super().__init__()
        self.qkv = nn.Linear(dim, 3 * dim, bias=True)
        self.proj = nn.Linear(dim, dim)


class VitMlpParams(_Holder):
    def __init__(self, dim):
        super().__init__()
        self.fc1 = nn.Linear(dim, 4 * dim)
        self.fc2 = nn.Linear(4 * dim, dim)


class VitBlockParams(_Holder):
    def __init__(self, dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = VitAttentionParams(dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = VitMlpParams(dim)


class VisionTransformerHybridParams(_Holder):
    """timm 0.6.12 VisionTransformer: own parameters cls_token, pos_embed; children patch_embed, blocks, norm, head."""

    def __init__(self, arch: HybridArch):
        super().__init__()
        self.arch = arch
        self.patch_embed = HybridEmbedParams(arch)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, arch.embed))
        self.pos_embed = nn.Parameter(torch.randn(1, arch.grid * arch.grid + 1, arch.embed) * 0.02)
        self.blocks = nn.Sequential(*[VitBlockParams(arch.embed) for _ in range(arch.depth)])
        self.norm = nn.LayerNorm(arch.embed, eps=1e-6)    # dead on the DPT path (both hooks fire before it)
        self.head = nn.Linear(arch.embed, 1000)            # dead on the DPT path
        self.start_index = 1
        self.patch_size = [arch.patch, arch.patch]


class ProjectReadout(_Holder):
    """backbones/utils.py:27-40."""

    def __init__(self, in_features, start_index=1):
        super().__init__()
        self.start_index = start_index
        self.project = nn.Sequential(nn.Linear(2 * in_features, in_features), nn.GELU())


class Transpose(_Holder):
    def __init__(self, dim0, dim1):
        super().__init__()
        self.dim0, self.dim1 = dim0, dim1


class HybridBackbone(_Holder):
    """`pretrained` of the reference for vitb_rn50_384: .model + act_postprocess1..4 (backbones/vit.py:159-229)."""

    def __init__(self, arch: HybridArch, hooks, use_readout="project"):
        super().__init__()
        assert use_readout == "project", "DPT builds its encoders with readout='project' (model/dpt.py:35)"
        self.model = VisionTransformerHybridParams(arch)
        self.hooks = list(hooks)
        f, E, g = arch.features, arch.embed, arch.grid
        self.act_postprocess1 = nn.Sequential(nn.Identity(), nn.Identity(), nn.Identity())
        self.act_postprocess2 = nn.Sequential(nn.Identity(), nn.Identity(), nn.Identity())
        self.act_postprocess3 = nn.Sequential(ProjectReadout(E), Transpose(1, 2), nn.Unflatten(2, torch.Size([g, g])),
                                              nn.Conv2d(E, f[2], kernel_size=1, stride=1, padding=0))
        self.act_postprocess4 = nn.Sequential(ProjectReadout(E), Transpose(1, 2), nn.Unflatten(2, torch.Size([g, g])),
                                              nn.Conv2d(E, f[3], kernel_size=1, stride=1, padding=0),
                                              nn.Conv2d(f[3], f[3], kernel_size=3, stride=2, padding=1))


def _make_pretrained_vitb_rn50_384(pretrained, use_readout="ignore", hooks=None, use_vit_only=False):
    assert not pretrained, "no network access: load weights through load_net / load_state_dict"
    assert not use_vit_only, "use_vit_only=False is what DPT passes (model/blocks.py:103-109)"
    arch = HYBRID_ARCHS["vitb_rn50_384"]
    return HybridBackbone(arch, arch.hooks if hooks is None else hooks, use_readout=use_readout)
