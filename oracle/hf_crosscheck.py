"""Independent cross-checks of the encoder restatements in oracle/soccdpt_ref.py (TEST INFRASTRUCTURE ONLY).

The encoder arithmetic of the reference lives in timm==0.6.12 (/root/reference/requirements.txt:12), which is neither vendored nor
installable here, and the reference holds no fixture for it: the encoders are "parity unpinned" at the timm boundary.  HF
`transformers` ships independent ports of the same upstream models (Swinv2Model; DPT's hybrid BiT + ViT embeddings); this module
loads the SAME synthetic weights into them and returns their hooked feature maps, so that tests/test_oracle_encoder.py and
oracle/make_golden.py can compare.  Agreement does not pin the oracle to timm -- it shows two independent readings of the published
architecture agree.  Only tests/ and oracle/make_golden.py import this file."""
from __future__ import annotations

from typing import Dict, List

import torch

Tensor = torch.Tensor

HF_SWIN_CFG = {
    "swin2t16_256": dict(image_size=256, patch_size=4, embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=16,
                         pretrained_window_sizes=[0, 0, 0, 0]),
    "swin2b24_384": dict(image_size=384, patch_size=4, embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], window_size=24,
                         pretrained_window_sizes=[12, 12, 12, 6]),
}


def swinv2_hf_features(sd: Dict[str, Tensor], x: Tensor, backbone: str, pfx: str = "depth_net.pretrained.model.") -> List[Tensor]:
    """The four pre-downsample stage outputs [B,C,H,W] of HF Swinv2Model carrying the timm-keyed weights `sd`
    (what the reference hooks: backbones/swin_common.py:12-54)."""
    from transformers import Swinv2Config, Swinv2Model
    c = HF_SWIN_CFG[backbone]
    hf = Swinv2Model(Swinv2Config(drop_path_rate=0.0, **c), add_pooling_layer=False).eval()
    hsd = {}
    hsd["embeddings.patch_embeddings.projection.weight"] = sd[pfx + "patch_embed.proj.weight"]
    hsd["embeddings.patch_embeddings.projection.bias"] = sd[pfx + "patch_embed.proj.bias"]
    hsd["embeddings.norm.weight"] = sd[pfx + "patch_embed.norm.weight"]
    hsd["embeddings.norm.bias"] = sd[pfx + "patch_embed.norm.bias"]
    nst = len(c["depths"])
    for s, depth in enumerate(c["depths"]):
        C = c["embed_dim"] << s
        for j in range(depth):
            t = f"{pfx}layers.{s}.blocks.{j}."
            h = f"encoder.layers.{s}.blocks.{j}."
            hsd[h + "attention.self.logit_scale"] = sd[t + "attn.logit_scale"]
            for m in ("0.weight", "0.bias", "2.weight"):
                hsd[h + "attention.self.continuous_position_bias_mlp." + m] = sd[t + "attn.cpb_mlp." + m]
            w = sd[t + "attn.qkv.weight"]
            hsd[h + "attention.self.query.weight"] = w[:C]
            hsd[h + "attention.self.key.weight"] = w[C:2 * C]
            hsd[h + "attention.self.value.weight"] = w[2 * C:]
            hsd[h + "attention.self.query.bias"] = sd[t + "attn.q_bias"]
            hsd[h + "attention.self.value.bias"] = sd[t + "attn.v_bias"]
            hsd[h + "attention.output.dense.weight"] = sd[t + "attn.proj.weight"]
            hsd[h + "attention.output.dense.bias"] = sd[t + "attn.proj.bias"]
            hsd[h + "layernorm_before.weight"] = sd[t + "norm1.weight"]
            hsd[h + "layernorm_before.bias"] = sd[t + "norm1.bias"]
            hsd[h + "intermediate.dense.weight"] = sd[t + "mlp.fc1.weight"]
            hsd[h + "intermediate.dense.bias"] = sd[t + "mlp.fc1.bias"]
            hsd[h + "output.dense.weight"] = sd[t + "mlp.fc2.weight"]
            hsd[h + "output.dense.bias"] = sd[t + "mlp.fc2.bias"]
            hsd[h + "layernorm_after.weight"] = sd[t + "norm2.weight"]
            hsd[h + "layernorm_after.bias"] = sd[t + "norm2.bias"]
        if s < nst - 1:
            for m in ("reduction.weight", "norm.weight", "norm.bias"):
                hsd[f"encoder.layers.{s}.downsample.{m}"] = sd[f"{pfx}layers.{s}.downsample.{m}"]
    hsd["layernorm.weight"] = sd[pfx + "norm.weight"]
    hsd["layernorm.bias"] = sd[pfx + "norm.bias"]
    res = hf.load_state_dict(hsd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert not [k for k in res.missing_keys if "relative" not in k], res.missing_keys   # buffers HF builds itself
    with torch.no_grad():
        emb, dims = hf.embeddings(x)
        eo = hf.encoder(emb, dims, output_hidden_states=True, output_hidden_states_before_downsampling=True)
    return list(eo.reshaped_hidden_states[1:])
