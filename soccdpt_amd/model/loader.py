"""load_model / load_transforms with the reference's signatures
(/root/reference/SOccDPT/model/loader.py:13-138,141-272)."""
from typing import Type, Union

import numpy as np
import torch

from .base_model import BaseModel
from .spec import HYBRID_ARCHS, MODEL_TYPE_TO_BACKBONE, SWIN_ARCHS
from .transforms import InputTransform


def load_model(arch, model_kwargs: dict, device: torch.device, model_path: str, model_type: str = "dpt_large_384",
               optimize: bool = False) -> Type[BaseModel]:
    """model_type -> backbone id -> arch(path=..., backbone=..., **model_kwargs), then .to(device).
    Unknown model types assert like the reference (loader.py:122-124).  model_types whose backbone has
    no HIP implementation are constructible only by classes that merely store the backbone string (the
    SOccDPT outer class, exactly as in the reference's scripts — SURVEY.md §3.2)."""
    assert issubclass(arch, BaseModel), f"arch '{arch}' not implemented, must be an instance of soccdpt_amd BaseModel"
    if model_type not in MODEL_TYPE_TO_BACKBONE:
        print(f"model_type '{model_type}' not implemented")
        assert False, f"model_type '{model_type}' not implemented"
    kwargs = dict(model_kwargs)
    if model_type == "dpt_levit_224":
        kwargs.update(head_features_1=64, head_features_2=8)
    model = arch(path=model_path, backbone=MODEL_TYPE_TO_BACKBONE[model_type], **kwargs)
    print("Model loaded, number of parameters = {:.0f}M".format(sum(p.numel() for p in model.parameters()) / 1e6))
    if optimize and torch.device(device).type == "cuda" and hasattr(model, "precision"):
        # the reference switches to channels_last + fp16 here (loader.py:132-134, model.half()); the MI355X path is already
        # NHWC, so "optimize" selects the fp16 operand mode of the same kernels (SOCCDPT_PREC_F16)
        from ..lib import PREC_F16
        model.precision = PREC_F16
    model.to(device)
    return model


def load_transforms(model_type: str = "dpt_large_384", height: int = 0, square: bool = False):
    """Returns (transform, net_w, net_h) (loader.py:141-272): Resize(cubic, ensure_multiple_of=32, "minimal") + NormalizeImage(0.5, 0.5)
    + PrepareForNet as one GPU transform (model/transforms.py here).  Sizes follow the model (the reference returns 256 for
    dpt_swin2_base_384, an inconsistency noted in SURVEY.md section 3.4); the Swin-V2 types do not keep the aspect ratio
    (loader.py:183-204), the others do unless `square`."""
    if model_type not in MODEL_TYPE_TO_BACKBONE:
        print(f"model_type '{model_type}' not implemented")
        assert False, f"model_type '{model_type}' not implemented"
    backbone = MODEL_TYPE_TO_BACKBONE[model_type]
    size = SWIN_ARCHS[backbone].img if backbone in SWIN_ARCHS else (HYBRID_ARCHS[backbone].img if backbone in HYBRID_ARCHS else 384)
    net_w = net_h = size
    keep_aspect_ratio = False if backbone in SWIN_ARCHS else (not square)
    if height != 0:
        net_w = net_h = height
    return InputTransform(net_w, net_h, keep_aspect_ratio=keep_aspect_ratio, ensure_multiple_of=32, resize_method="minimal"), net_w, net_h
