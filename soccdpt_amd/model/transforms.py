"""Input transform of the network on the GPU (/root/reference/SOccDPT/model/transforms.py:53-251 Resize / NormalizeImage /
PrepareForNet as composed by model/loader.py:256-270).

The reference resizes each uint8 camera frame on the host with cv2 (INTER_CUBIC), normalises in numpy and transposes to CHW.
Here the three steps are one HIP kernel (soccdpt_input_transform_u8, csrc/input_transform.hip); only the output-size rule
stays on the host.  There is no CPU fallback: a uint8 frame needs the HIP library and a GPU.
"""
from typing import Sequence, Tuple

import numpy as np
import torch


def constrain_to_multiple_of(x: float, multiple_of: int, min_val: int = 0, max_val=None) -> int:
    """transforms.py:107-122: round to the nearest multiple; floor / ceil instead when that leaves [min_val, max_val]."""
    y = int(np.round(x / multiple_of) * multiple_of)
    if max_val is not None and y > max_val:
        y = int(np.floor(x / multiple_of) * multiple_of)
    if y < min_val:
        y = int(np.ceil(x / multiple_of) * multiple_of)
    return y


def get_size(width: int, height: int, net_w: int, net_h: int, keep_aspect_ratio: bool = False, multiple_of: int = 1,
             resize_method: str = "lower_bound") -> Tuple[int, int]:
    """(new_width, new_height) for an input of (width, height): transforms.py:124-176."""
    sh, sw = net_h / height, net_w / width
    if keep_aspect_ratio:
        if resize_method == "lower_bound":
            fit_width = sw > sh
        elif resize_method == "upper_bound":
            fit_width = sw < sh
        elif resize_method == "minimal":
            fit_width = abs(1 - sw) < abs(1 - sh)
        else:
            raise ValueError(f"resize_method {resize_method} not implemented")
        if fit_width:
            sh = sw
        else:
            sw = sh
    if resize_method == "lower_bound":
        return (constrain_to_multiple_of(sw * width, multiple_of, min_val=net_w),
                constrain_to_multiple_of(sh * height, multiple_of, min_val=net_h))
    if resize_method == "upper_bound":
        return (constrain_to_multiple_of(sw * width, multiple_of, max_val=net_w),
                constrain_to_multiple_of(sh * height, multiple_of, max_val=net_h))
    if resize_method == "minimal":
        return constrain_to_multiple_of(sw * width, multiple_of), constrain_to_multiple_of(sh * height, multiple_of)
    raise ValueError(f"resize_method {resize_method} not implemented")


class InputTransform:
    """What load_transforms returns: callable on {"image": frame} like the reference's Compose.

    * uint8 HWC frame (numpy or torch, any size): resized (bicubic), normalised and transposed on the GPU; a numpy frame comes
      back as a numpy float32 CHW array (drop-in for datasets/*.py), a device tensor stays on the device.
    * `batch(frames)`: uint8 [B,H,W,3] device tensor -> float32 [B,3,h,w] device tensor, one launch (the serving path).
    * float frames already at network resolution only take the normalisation (no resampling is defined for them here).
    """

    def __init__(self, net_w: int, net_h: int, keep_aspect_ratio: bool = False, ensure_multiple_of: int = 32,
                 resize_method: str = "minimal", mean: Sequence[float] = (0.5, 0.5, 0.5), std: Sequence[float] = (0.5, 0.5, 0.5),
                 device="cuda:0"):
        self.net_w, self.net_h = net_w, net_h
        self.keep_aspect_ratio = keep_aspect_ratio
        self.multiple_of = ensure_multiple_of
        self.resize_method = resize_method
        self.mean, self.std = tuple(float(m) for m in mean), tuple(float(s) for s in std)
        self.device = torch.device(device)

    def get_size(self, width: int, height: int) -> Tuple[int, int]:
        return get_size(width, height, self.net_w, self.net_h, self.keep_aspect_ratio, self.multiple_of, self.resize_method)

    def batch(self, frames: torch.Tensor) -> torch.Tensor:
        from ..lib import op_input_transform_u8
        assert frames.dtype == torch.uint8 and frames.dim() == 4 and frames.shape[-1] == 3, "frames must be uint8 [B,H,W,3]"
        if frames.device.type != "cuda":
            raise RuntimeError("soccdpt_amd input transform runs on MI355X only: move the frames to a cuda device (there is no CPU fallback)")
        w, h = self.get_size(frames.shape[2], frames.shape[1])
        return op_input_transform_u8(frames.contiguous(), h, w, self.mean, self.std)

    def __call__(self, sample: dict) -> dict:
        img = sample["image"]
        is_np = not torch.is_tensor(img)
        t = torch.from_numpy(np.ascontiguousarray(img)) if is_np else img
        if t.dtype == torch.uint8:
            out = self.batch(t.to(self.device if t.device.type != "cuda" else t.device).unsqueeze(0))[0]
            sample["image"] = out.cpu().numpy() if is_np else out
            return sample
        a = np.asarray(img, dtype=np.float32)
        assert a.shape[0] == self.net_h and a.shape[1] == self.net_w, \
            "float frames must already be at network resolution; hand over the uint8 frame to have it resized"
        a = (a - np.asarray(self.mean)) / np.asarray(self.std)
        sample["image"] = np.ascontiguousarray(np.transpose(a, (2, 0, 1))).astype(np.float32)
        return sample
