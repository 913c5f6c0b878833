"""DPT decoder module tree (parameter holders) with the reference's names:
_make_encoder / _make_scratch / Interpolate / ResidualConvUnit_custom / FeatureFusionBlock_custom
(/root/reference/SOccDPT/model/blocks.py:31-193,239-273,345-497).  The arithmetic (3x3 implicit-GEMM
convolutions with fused ReLU / bias / residual epilogues, bilinear resize) runs in libsoccdpt_hip.so."""
import torch.nn as nn

from .backbones.swin2 import _make_pretrained_swin2b24_384, _make_pretrained_swin2t16_256
from .backbones.vit import _make_pretrained_vitb_rn50_384
from .spec import HYBRID_ARCHS, SWIN_ARCHS


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder: the DPT decoder runs inside libsoccdpt_hip.so")


def _make_encoder(backbone, features, use_pretrained, groups=1, expand=False, exportable=True, hooks=None,
                  use_vit_only=False, use_readout="ignore", in_features=None):
    if backbone == "swin2t16_256":
        pretrained = _make_pretrained_swin2t16_256(use_pretrained, hooks=hooks)
    elif backbone == "swin2b24_384":
        pretrained = _make_pretrained_swin2b24_384(use_pretrained, hooks=hooks)
    elif backbone == "vitb_rn50_384":
        pretrained = _make_pretrained_vitb_rn50_384(use_pretrained, hooks=hooks, use_vit_only=use_vit_only, use_readout=use_readout)
    else:
        print(f"Backbone '{backbone}' not implemented")
        assert False, f"Backbone '{backbone}' not implemented on the MI355X path"
    arch = HYBRID_ARCHS[backbone] if backbone in HYBRID_ARCHS else SWIN_ARCHS[backbone]
    scratch = _make_scratch(arch.dims(), features, groups=groups, expand=expand)
    return pretrained, scratch


def _make_scratch(in_shape, out_shape, groups=1, expand=False):
    assert groups == 1 and not expand, "the MI355X path implements groups=1, expand=False (what SOccDPT_V3 uses)"
    scratch = _Holder()
    for i, c in enumerate(in_shape):
        setattr(scratch, f"layer{i + 1}_rn", nn.Conv2d(c, out_shape, kernel_size=3, stride=1, padding=1, bias=False))
    return scratch


class Interpolate(_Holder):
    def __init__(self, scale_factor, mode, align_corners=False):
        super().__init__()
        self.scale_factor, self.mode, self.align_corners = scale_factor, mode, align_corners


class ResidualConvUnit_custom(_Holder):
    def __init__(self, features, activation, bn):
        super().__init__()
        assert not bn, "use_bn=False on the SOccDPT_V3 path (model/dpt.py:18-27)"
        self.bn = bn
        self.groups = 1
        self.conv1 = nn.Conv2d(features, features, kernel_size=3, stride=1, padding=1, bias=True)
        self.conv2 = nn.Conv2d(features, features, kernel_size=3, stride=1, padding=1, bias=True)
        self.activation = activation


class FeatureFusionBlock_custom(_Holder):
    def __init__(self, features, activation, deconv=False, bn=False, expand=False, align_corners=True, size=None):
        super().__init__()
        assert not deconv and not expand and align_corners
        self.deconv, self.align_corners, self.groups, self.expand, self.size = deconv, align_corners, 1, expand, size
        self.out_conv = nn.Conv2d(features, features, kernel_size=1, stride=1, padding=0, bias=True)
        self.resConfUnit1 = ResidualConvUnit_custom(features, activation, bn)
        self.resConfUnit2 = ResidualConvUnit_custom(features, activation, bn)
