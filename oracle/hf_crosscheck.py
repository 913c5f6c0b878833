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


def hybrid_hf_features(sd: Dict[str, Tensor], x: Tensor, pfx: str = "depth_net.pretrained.") -> List[Tensor]:
    """The four maps HF's DPT-hybrid (BiT ResNetV2 backbone + ViT-B + 'project' readout + reassemble stage) hands to the fusion
    convs, carrying the timm / reference-keyed weights `sd`: what hybrid_encoder() in oracle/soccdpt_ref.py restates from
    backbones/vit.py:19-85,147-258 and backbones/utils.py:27-40,84-133."""
    from transformers import BitConfig, DPTConfig, DPTModel
    from transformers.models.dpt.modeling_dpt import DPTReassembleStage
    bc = BitConfig(global_padding="same", layer_type="bottleneck", depths=[3, 4, 9], out_features=["stage1", "stage2", "stage3"],
                   embedding_dynamic_padding=True, hidden_sizes=[256, 512, 1024, 2048], embedding_size=64, num_groups=32, drop_path_rate=0.0)
    cfg = DPTConfig(is_hybrid=True, backbone_config=bc, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                    image_size=x.shape[-1], patch_size=16, backbone_out_indices=[0, 1, 8, 11], neck_hidden_sizes=[256, 512, 768, 768],
                    reassemble_factors=[1, 1, 1, 0.5], readout_type="project", backbone_featmap_shape=[1, 1024, 24, 24], neck_ignore_stages=[0, 1],
                    qkv_bias=True, layer_norm_eps=1e-6, hidden_act="gelu", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    hf = DPTModel(cfg, add_pooling_layer=False).eval()
    rs = DPTReassembleStage(cfg).eval()
    m = pfx + "model."
    h = {"embeddings.cls_token": sd[m + "cls_token"], "embeddings.position_embeddings": sd[m + "pos_embed"],
         "embeddings.projection.weight": sd[m + "patch_embed.proj.weight"], "embeddings.projection.bias": sd[m + "patch_embed.proj.bias"]}
    bb, hb = m + "patch_embed.backbone.", "embeddings.backbone.bit."
    h[hb + "embedder.convolution.weight"] = sd[bb + "stem.conv.weight"]
    h[hb + "embedder.norm.weight"] = sd[bb + "stem.norm.weight"]
    h[hb + "embedder.norm.bias"] = sd[bb + "stem.norm.bias"]
    for k, v in sd.items():
        if k.startswith(bb + "stages."):
            s, _, j, rest = k[len(bb + "stages."):].split(".", 3)
            h[f"{hb}encoder.stages.{s}.layers.{j}.{rest}"] = v
    for i in range(12):
        t, e = f"{m}blocks.{i}.", f"encoder.layer.{i}."
        w, b = sd[t + "attn.qkv.weight"], sd[t + "attn.qkv.bias"]
        for n, name in enumerate(("query", "key", "value")):
            h[f"{e}attention.attention.{name}.weight"] = w[768 * n:768 * (n + 1)]
            h[f"{e}attention.attention.{name}.bias"] = b[768 * n:768 * (n + 1)]
        h[e + "attention.output.dense.weight"] = sd[t + "attn.proj.weight"]
        h[e + "attention.output.dense.bias"] = sd[t + "attn.proj.bias"]
        h[e + "intermediate.dense.weight"] = sd[t + "mlp.fc1.weight"]
        h[e + "intermediate.dense.bias"] = sd[t + "mlp.fc1.bias"]
        h[e + "output.dense.weight"] = sd[t + "mlp.fc2.weight"]
        h[e + "output.dense.bias"] = sd[t + "mlp.fc2.bias"]
        for a, c in (("layernorm_before", "norm1"), ("layernorm_after", "norm2")):
            h[f"{e}{a}.weight"] = sd[f"{t}{c}.weight"]
            h[f"{e}{a}.bias"] = sd[f"{t}{c}.bias"]
    h["layernorm.weight"] = sd[m + "norm.weight"]
    h["layernorm.bias"] = sd[m + "norm.bias"]
    res = hf.load_state_dict(h, strict=False)
    assert not res.unexpected_keys and not res.missing_keys, (res.unexpected_keys, res.missing_keys)
    r = {}
    for n in (3, 4):
        a = f"{pfx}act_postprocess{n}."
        r[f"readout_projects.{n - 1}.0.weight"] = sd[a + "0.project.0.weight"]
        r[f"readout_projects.{n - 1}.0.bias"] = sd[a + "0.project.0.bias"]
        r[f"layers.{n - 1}.projection.weight"] = sd[a + "3.weight"]
        r[f"layers.{n - 1}.projection.bias"] = sd[a + "3.bias"]
    r["layers.3.resize.weight"] = sd[pfx + "act_postprocess4.4.weight"]
    r["layers.3.resize.bias"] = sd[pfx + "act_postprocess4.4.bias"]
    res = rs.load_state_dict(r, strict=False)
    assert not res.unexpected_keys and not res.missing_keys, (res.unexpected_keys, res.missing_keys)
    with torch.no_grad():
        out = hf(x, output_hidden_states=True)
        hs = list(out.intermediate_activations) + [out.hidden_states[1:][i] for i in (8, 11)]
        g = x.shape[-1] // 16
        return rs(hs, g, g)
