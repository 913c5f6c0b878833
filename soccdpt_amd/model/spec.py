"""Architecture tables and the state-dict key layout of the SOccDPT_V3 hot path.

Mirrors what the reference derives at construction time:
  * model_type -> backbone id: /root/reference/SOccDPT/model/loader.py:65-77
  * hooks per backbone:        /root/reference/SOccDPT/model/dpt.py:51-89
  * reassemble channel lists:  /root/reference/SOccDPT/model/blocks.py:64-78
  * timm model ids:            /root/reference/SOccDPT/model/backbones/swin2.py:15-30
  * default weight files:      /root/reference/SOccDPT/model/SOccDPT.py:29-57
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Tuple


@dataclass(frozen=True)
class SwinV2Arch:
    name: str
    timm_name: str
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

    def stage_res(self, s: int) -> int:
        return self.grid >> s

    def window_shift(self, s: int, j: int) -> Tuple[int, int]:
        res = self.stage_res(s)
        ws = min(res, self.window)
        shift = 0 if (j % 2 == 0 or res <= self.window) else self.window // 2
        return ws, shift


SWIN_ARCHS: Dict[str, SwinV2Arch] = {
    "swin2t16_256": SwinV2Arch("swin2t16_256", "swinv2_tiny_window16_256", 256, 4, 96,
                               (2, 2, 6, 2), (3, 6, 12, 24), 16, (0, 0, 0, 0), (1, 1, 5, 1)),
    "swin2b24_384": SwinV2Arch("swin2b24_384", "swinv2_base_window12to24_192to384_22kft1k", 384, 4, 128,
                               (2, 2, 18, 2), (4, 8, 16, 32), 24, (12, 12, 12, 6), (1, 1, 17, 1)),
}

# model_type -> backbone (only the backbones this build implements on the HIP path
# are constructible; the other ids are kept so the switch reports them by name).
MODEL_TYPE_TO_BACKBONE: Dict[str, str] = {
    "dpt_beit_large_512": "beitl16_512",
    "dpt_beit_large_384": "beitl16_384",
    "dpt_beit_base_384": "beitb16_384",
    "dpt_swin2_large_384": "swin2l24_384",
    "dpt_swin2_base_384": "swin2b24_384",
    "dpt_swin2_tiny_256": "swin2t16_256",
    "dpt_swin_large_384": "swinl12_384",
    "dpt_next_vit_large_384": "next_vit_large_6m",
    "dpt_levit_224": "levit_384",
    "dpt_large_384": "vitl16_384",
    "dpt_hybrid_384": "vitb_rn50_384",
}

def backbone_image_size(backbone: str) -> int:
    """Network input size of a backbone this build implements (model/loader.py:141-272 net_w / net_h)."""
    if backbone in SWIN_ARCHS:
        return SWIN_ARCHS[backbone].img
    if backbone in HYBRID_ARCHS:
        return HYBRID_ARCHS[backbone].img
    raise AssertionError(f"Backbone '{backbone}' not implemented on the MI355X path")


DEFAULT_DEPTH_WEIGHTS: Dict[str, str] = {k: f"weights/{k}.pt" for k in MODEL_TYPE_TO_BACKBONE}

model_types = DEFAULT_DEPTH_WEIGHTS.keys()


def encoder_param_shapes(arch: SwinV2Arch) -> "OrderedDict[str, Tuple[int, ...]]":
    """timm-0.6.12 SwinTransformerV2 parameter names/shapes in registration order."""
    p: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    C0 = arch.embed
    p["patch_embed.proj.weight"] = (C0, 3, arch.patch, arch.patch)
    p["patch_embed.proj.bias"] = (C0,)
    p["patch_embed.norm.weight"] = (C0,)
    p["patch_embed.norm.bias"] = (C0,)
    for s, depth in enumerate(arch.depths):
        C = C0 << s
        H = arch.heads[s]
        for j in range(depth):
            b = f"layers.{s}.blocks.{j}."
            p[b + "attn.logit_scale"] = (H, 1, 1)
            p[b + "attn.q_bias"] = (C,)
            p[b + "attn.v_bias"] = (C,)
            p[b + "attn.cpb_mlp.0.weight"] = (512, 2)
            p[b + "attn.cpb_mlp.0.bias"] = (512,)
            p[b + "attn.cpb_mlp.2.weight"] = (H, 512)
            p[b + "attn.qkv.weight"] = (3 * C, C)
            p[b + "attn.proj.weight"] = (C, C)
            p[b + "attn.proj.bias"] = (C,)
            p[b + "norm1.weight"] = (C,)
            p[b + "norm1.bias"] = (C,)
            p[b + "mlp.fc1.weight"] = (4 * C, C)
            p[b + "mlp.fc1.bias"] = (4 * C,)
            p[b + "mlp.fc2.weight"] = (C, 4 * C)
            p[b + "mlp.fc2.bias"] = (C,)
            p[b + "norm2.weight"] = (C,)
            p[b + "norm2.bias"] = (C,)
        if s < len(arch.depths) - 1:
            d = f"layers.{s}.downsample."
            p[d + "reduction.weight"] = (2 * C, 4 * C)
            p[d + "norm.weight"] = (2 * C,)
            p[d + "norm.bias"] = (2 * C,)
    Cl = C0 << (len(arch.depths) - 1)
    p["norm.weight"] = (Cl,)
    p["norm.bias"] = (Cl,)
    p["head.weight"] = (1000, Cl)
    p["head.bias"] = (1000,)
    return p


@dataclass(frozen=True)
class HybridArch:
    """timm 0.6.12 vit_base_resnet50_384 as the reference creates it (/root/reference/SOccDPT/model/backbones/vit.py:244-258);
    hooks /root/reference/SOccDPT/model/dpt.py:86, reassemble channels /root/reference/SOccDPT/model/blocks.py:103-112."""
    name: str = "vitb_rn50_384"
    timm_name: str = "vit_base_resnet50_384"
    img: int = 384
    patch: int = 16
    embed: int = 768
    depth: int = 12
    heads: int = 12
    stem: int = 64
    layers: Tuple[int, ...] = (3, 4, 9)
    hooks: Tuple[int, ...] = (0, 1, 8, 11)
    features: Tuple[int, ...] = (256, 512, 768, 768)

    @property
    def grid(self) -> int:
        return self.img // self.patch

    def dims(self) -> List[int]:
        return list(self.features)


HYBRID_ARCHS: Dict[str, HybridArch] = {"vitb_rn50_384": HybridArch()}


def hybrid_param_shapes(arch: HybridArch) -> "OrderedDict[str, Tuple[int, ...]]":
    """Parameters of `pretrained` for vitb_rn50_384 in registration order: `model.*` (timm VisionTransformer with a HybridEmbed:
    own parameters cls_token / pos_embed first, then patch_embed.backbone (ResNetV2: stem, stages), patch_embed.proj, blocks, norm,
    head), then act_postprocess3 / act_postprocess4 (backbones/vit.py:183-229; 1 and 2 are parameter-free Identity stacks)."""
    p: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    E = arch.embed
    n_tok = arch.grid * arch.grid + 1
    p["model.cls_token"] = (1, 1, E)
    p["model.pos_embed"] = (1, n_tok, E)
    bb = "model.patch_embed.backbone."
    p[bb + "stem.conv.weight"] = (arch.stem, 3, 7, 7)
    p[bb + "stem.norm.weight"] = (arch.stem,)
    p[bb + "stem.norm.bias"] = (arch.stem,)
    prev = arch.stem
    for s, depth in enumerate(arch.layers):
        out = 256 << s
        mid = out // 4
        for j in range(depth):
            b = f"{bb}stages.{s}.blocks.{j}."
            if j == 0:
                p[b + "downsample.conv.weight"] = (out, prev, 1, 1)
                p[b + "downsample.norm.weight"] = (out,)
                p[b + "downsample.norm.bias"] = (out,)
            p[b + "conv1.weight"] = (mid, prev, 1, 1)
            p[b + "norm1.weight"] = (mid,)
            p[b + "norm1.bias"] = (mid,)
            p[b + "conv2.weight"] = (mid, mid, 3, 3)
            p[b + "norm2.weight"] = (mid,)
            p[b + "norm2.bias"] = (mid,)
            p[b + "conv3.weight"] = (out, mid, 1, 1)
            p[b + "norm3.weight"] = (out,)
            p[b + "norm3.bias"] = (out,)
            prev = out
    p["model.patch_embed.proj.weight"] = (E, prev, 1, 1)
    p["model.patch_embed.proj.bias"] = (E,)
    for i in range(arch.depth):
        b = f"model.blocks.{i}."
        p[b + "norm1.weight"] = (E,)
        p[b + "norm1.bias"] = (E,)
        p[b + "attn.qkv.weight"] = (3 * E, E)
        p[b + "attn.qkv.bias"] = (3 * E,)
        p[b + "attn.proj.weight"] = (E, E)
        p[b + "attn.proj.bias"] = (E,)
        p[b + "norm2.weight"] = (E,)
        p[b + "norm2.bias"] = (E,)
        p[b + "mlp.fc1.weight"] = (4 * E, E)
        p[b + "mlp.fc1.bias"] = (4 * E,)
        p[b + "mlp.fc2.weight"] = (E, 4 * E)
        p[b + "mlp.fc2.bias"] = (E,)
    p["model.norm.weight"] = (E,)
    p["model.norm.bias"] = (E,)
    p["model.head.weight"] = (1000, E)
    p["model.head.bias"] = (1000,)
    for n in (3, 4):
        a = f"act_postprocess{n}."
        p[a + "0.project.0.weight"] = (E, 2 * E)
        p[a + "0.project.0.bias"] = (E,)
        p[a + "3.weight"] = (arch.features[n - 1], E, 1, 1)
        p[a + "3.bias"] = (arch.features[n - 1],)
    p["act_postprocess4.4.weight"] = (arch.features[3], arch.features[3], 3, 3)
    p["act_postprocess4.4.bias"] = (arch.features[3],)
    return p


def scratch_param_shapes(arch, features: int = 256) -> "OrderedDict[str, Tuple[int, ...]]":
    """depth_net.scratch.* (blocks.py:139-193, dpt.py:133-140,199-219)."""
    p: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    for i, c in enumerate(arch.dims()):
        p[f"layer{i + 1}_rn.weight"] = (features, c, 3, 3)
    for r in (1, 2, 3, 4):
        b = f"refinenet{r}."
        p[b + "out_conv.weight"] = (features, features, 1, 1)
        p[b + "out_conv.bias"] = (features,)
        for u in (1, 2):
            for c in (1, 2):
                p[b + f"resConfUnit{u}.conv{c}.weight"] = (features, features, 3, 3)
                p[b + f"resConfUnit{u}.conv{c}.bias"] = (features,)
    p["output_conv.0.weight"] = (features // 2, features, 3, 3)
    p["output_conv.0.bias"] = (features // 2,)
    p["output_conv.2.weight"] = (32, features // 2, 3, 3)
    p["output_conv.2.bias"] = (32,)
    p["output_conv.4.weight"] = (1, 32, 1, 1)
    p["output_conv.4.bias"] = (1,)
    return p


def seg_head_param_shapes(features: int = 256, num_classes: int = 3) -> "OrderedDict[str, Tuple[int, ...]]":
    """seg_head.* (model/SOccDPT.py:660-674); includes BN buffers."""
    p: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    p["0.weight"] = (features, features, 3, 3)
    p["1.weight"] = (features,)
    p["1.bias"] = (features,)
    p["1.running_mean"] = (features,)
    p["1.running_var"] = (features,)
    p["4.weight"] = (num_classes, features, 1, 1)
    p["4.bias"] = (num_classes,)
    return p


def v3_state_shapes(backbone: str, features: int = 256, num_classes: int = 3) -> "OrderedDict[str, Tuple[int, ...]]":
    """All V3 tensors under the reference's canonical prefixes (without the
    duplicate `pretrained.model.*` aliases)."""
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    if backbone in HYBRID_ARCHS:
        arch = HYBRID_ARCHS[backbone]
        for k, v in hybrid_param_shapes(arch).items():
            out["depth_net.pretrained." + k] = v
    else:
        arch = SWIN_ARCHS[backbone]
        for k, v in encoder_param_shapes(arch).items():
            out["depth_net.pretrained.model." + k] = v
    for k, v in scratch_param_shapes(arch, features).items():
        out["depth_net.scratch." + k] = v
    for k, v in seg_head_param_shapes(features, num_classes).items():
        out["seg_head." + k] = v
    return out
