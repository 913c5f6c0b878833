"""Deterministic synthetic weights, camera calibration and inputs.

No checkpoints, datasets or calibration files exist offline, so
benchmarks and tests run on tensors generated from the state-dict key name:
the same code produces the same values here and on the GPU box.
"""
from __future__ import annotations

import math
import os
import zlib
from collections import OrderedDict
from typing import Dict, Optional

import torch

from ..model.spec import v3_state_shapes

# The only intrinsics in the reference tree (media/manydepth/intrinsics.json:1-5);
# image size from datasets/bengaluru_driving_dataset.py:118-121.
SYNTH_CALIB = {
    "Camera.fx": 1250.6, "Camera.fy": 1254.8, "Camera.cx": 978.4, "Camera.cy": 562.1,
    "Camera.k1": 0.0, "Camera.k2": 0.0, "Camera.p1": 0.0, "Camera.p2": 0.0,
    "Camera.width": 1920, "Camera.height": 1080,
}


def write_synth_calib(path: str, **overrides) -> str:
    """Write a calib YAML with the keys model/SOccDPT.py:198-228 reads."""
    vals = dict(SYNTH_CALIB)
    vals.update(overrides)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        for k, v in vals.items():
            f.write(f"{k}: {v}\n")
    return path


def _gen(key: str, salt: int = 0) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) + 7919 * salt) & 0x7FFFFFFF)
    return g


def _tensor_for(key: str, shape, salt: int) -> torch.Tensor:
    g = _gen(key, salt)
    leaf = key.split(".")[-1]
    n = lambda std: torch.randn(shape, generator=g, dtype=torch.float32) * std  # noqa: E731
    if key.endswith("logit_scale"):
        return math.log(10.0) + n(0.2)
    if "running_var" in key:
        return 1.0 + 0.2 * torch.rand(shape, generator=g, dtype=torch.float32)
    if "running_mean" in key:
        return n(0.05)
    if ".norm" in key or key.startswith("norm") or key.endswith("seg_head.1.weight") or key.endswith("seg_head.1.bias"):
        # LayerNorm / BatchNorm affine: ones/zeros plus jitter so a missing gamma/beta is caught
        return (1.0 + n(0.05)) if leaf == "weight" else n(0.05)
    if leaf in ("q_bias", "v_bias", "bias"):
        return n(0.05)
    if "cpb_mlp.0.weight" in key:
        return n(0.5)
    if "cpb_mlp.2.weight" in key:
        return n(0.05)
    if len(shape) == 4:  # conv: variance-preserving-ish
        fan_in = shape[1] * shape[2] * shape[3]
        return n(1.0 / math.sqrt(fan_in))
    if len(shape) == 2:  # linear
        return n(1.0 / math.sqrt(shape[1]))
    return n(0.02)


def synth_state_dict(backbone: str = "swin2t16_256", features: int = 256, num_classes: int = 3,
                     salt: int = 0, alias_pretrained: bool = False) -> "OrderedDict[str, torch.Tensor]":
    """Synthetic V3 state dict keyed like the reference's checkpoints
    (SURVEY.md §8b).  A few tensors are biased so the synthetic network produces
    a non-degenerate scene: inverse depth in roughly [0.02, 0.3] (points land
    inside the 128 m x 128 m x 48 m grid) and seg logits wide enough that
    ScaledTanh produces exact zeros (model/SOccDPT.py:440 depends on them)."""
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, shp in v3_state_shapes(backbone, features, num_classes).items():
        sd[k] = _tensor_for(k, tuple(shp), salt)
    k4 = "depth_net.scratch.output_conv.4."
    sd[k4 + "weight"] = sd[k4 + "weight"].abs() * 0.01
    sd[k4 + "bias"] = torch.full((1,), 0.015)
    sd["seg_head.4.weight"] = sd["seg_head.4.weight"] * 12.0
    if alias_pretrained:  # the reference registers the encoder twice (model/SOccDPT.py:650)
        for k in list(sd.keys()):
            if k.startswith("depth_net.pretrained."):
                sd[k[len("depth_net."):]] = sd[k]
    return sd


def synth_input(batch: int, size: int = 256, seed0: int = 0, all_zero: bool = False) -> torch.Tensor:
    """x[B,3,H,W] f32: uint8 image through NormalizeImage(0.5,0.5) WITHOUT /255
    (faithful to datasets/bengaluru_driving_dataset.py:128 + model/loader.py:201-203;
    SURVEY.md §3.4).  One seed per frame so shards are reproducible."""
    frames = []
    for b in range(batch):
        if all_zero:
            img = torch.zeros(3, size, size)
        else:
            g = torch.Generator(device="cpu")
            g.manual_seed(seed0 + b)
            # smooth-ish image: low-res noise upsampled + pixel noise
            lo = torch.randint(0, 256, (1, 3, size // 16, size // 16), generator=g).float()
            img = torch.nn.functional.interpolate(lo, size=(size, size), mode="bilinear", align_corners=False)[0]
            img = (img + torch.randint(-8, 9, (3, size, size), generator=g).float()).clamp(0, 255).round()
        frames.append((img - 0.5) / 0.5)
    return torch.stack(frames).contiguous()


def trained_like(sd: dict, seed: int = 5) -> dict:
    """Statistics a trained checkpoint has and the synthetic draws lack: LayerNorm / BatchNorm gains far from 1 (U(0.2, 3)), and a few output rows
    whose weights are an order of magnitude larger than the rest (the outlier features of trained transformers).  Used by tests/test_calibrate_gpu.py
    and bench.py's `other_weights` as a stand-in for a real checkpoint (there is no network for one)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in sd.items():
        t = v.clone()
        leaf = k.split(".")[-1]
        if leaf == "weight" and (".norm" in k or k.endswith("seg_head.1.weight")) and t.dim() == 1:
            t = 0.2 + 2.8 * torch.rand(t.shape, generator=g)
        elif leaf == "weight" and t.dim() == 2 and "attn.qkv" not in k and "cpb_mlp" not in k and t.shape[0] >= 96:
            rows = torch.randperm(t.shape[0], generator=g)[:2]
            t[rows] *= 10.0
        out[k] = t
    for k in list(out):   # the reference registers the encoder twice (model/SOccDPT.py:650): keep the aliases identical
        if k.startswith("depth_net.pretrained.") and k[len("depth_net."):] in out:
            out[k[len("depth_net."):]] = out[k]
    return out


WEIGHT_SETS = ("salt0", "salt1", "salt2", "trained_like")


def named_weights(name: str, backbone: str = "swin2t16_256") -> dict:
    """'salt0' = the draw the shipped precision maps were derived on (tests, bench default); 'salt1' / 'salt2' = other draws of the same generator;
    'trained_like' = draw 3 with trained-like statistics."""
    if name == "trained_like":
        return trained_like(synth_state_dict(backbone, salt=3, alias_pretrained=True))
    assert name in WEIGHT_SETS, name
    return synth_state_dict(backbone, salt=int(name[4:]), alias_pretrained=True)
