"""DPT / DPTDepthModel module tree (parameter holders) — /root/reference/SOccDPT/model/dpt.py:30-232."""
import torch.nn as nn

from .base_model import BaseModel
from .blocks import FeatureFusionBlock_custom, Interpolate, _make_encoder
from .spec import HYBRID_ARCHS, SWIN_ARCHS


def _make_fusion_block(features, use_bn, size=None):
    return FeatureFusionBlock_custom(features, nn.ReLU(False), deconv=False, bn=use_bn, expand=False,
                                     align_corners=True, size=size)


class DPT(BaseModel):
    def __init__(self, head, features=256, backbone="swin2t16_256", readout="project", channels_last=False,
                 use_bn=False, return_features=False, **kwargs):
        super().__init__()
        assert backbone in SWIN_ARCHS or backbone in HYBRID_ARCHS, f"Backbone '{backbone}' not implemented on the MI355X path"
        assert features == 256 and not use_bn
        self.channels_last = channels_last
        self.return_features = return_features
        self.backbone = backbone
        hooks = list((HYBRID_ARCHS[backbone] if backbone in HYBRID_ARCHS else SWIN_ARCHS[backbone]).hooks)   # model/dpt.py:51-89
        self.pretrained, self.scratch = _make_encoder(backbone, features, False, groups=1, expand=False,
                                                      exportable=False, hooks=hooks, use_readout=readout)
        self.number_layers = 4
        self.scratch.stem_transpose = None
        self.scratch.refinenet1 = _make_fusion_block(features, use_bn)
        self.scratch.refinenet2 = _make_fusion_block(features, use_bn)
        self.scratch.refinenet3 = _make_fusion_block(features, use_bn)
        self.scratch.refinenet4 = _make_fusion_block(features, use_bn)
        self.scratch.output_conv = head

    def forward(self, x):  # pragma: no cover
        raise RuntimeError("DPT runs as part of SOccDPT_V3.forward inside libsoccdpt_hip.so (soccdpt_network)")


class DPTDepthModel(DPT):
    def __init__(self, path=None, non_negative=True, **kwargs):
        features = kwargs["features"] if "features" in kwargs else 256
        head_features_1 = kwargs.pop("head_features_1", features)
        head_features_2 = kwargs.pop("head_features_2", 32)
        assert non_negative and head_features_1 == 256 and head_features_2 == 32
        head = nn.Sequential(
            nn.Conv2d(head_features_1, head_features_1 // 2, kernel_size=3, stride=1, padding=1),
            Interpolate(scale_factor=2, mode="bilinear", align_corners=True),
            nn.Conv2d(head_features_1 // 2, head_features_2, kernel_size=3, stride=1, padding=1),
            nn.ReLU(True),
            nn.Conv2d(head_features_2, 1, kernel_size=1, stride=1, padding=0),
            nn.ReLU(True),
            nn.Identity(),
        )
        super().__init__(head, **kwargs)
        if path is not None:
            self.load_net(path)
