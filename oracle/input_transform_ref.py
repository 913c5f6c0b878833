"""CPU restatement of the reference's input transform (TEST INFRASTRUCTURE ONLY: imported by tests/ and nothing else).

PARITY UNPINNED.  The reference's transform is `Resize(net_w, net_h, keep_aspect_ratio, ensure_multiple_of=32, "minimal",
cv2.INTER_CUBIC)` -> `NormalizeImage(0.5, 0.5)` -> `PrepareForNet` (/root/reference/SOccDPT/model/loader.py:256-270,
model/transforms.py:53-251) applied to the uint8 RGB frame of the datasets (datasets/bengaluru_driving_dataset.py:118-130).
The resampling itself lives in OpenCV (`cv2.resize`, opencv-python 4.x, requirements.txt), which is absent from this image and
has no fixture in the reference tree, so nothing here can be checked against it.  The restatement follows OpenCV's published
8-bit bicubic algorithm (modules/imgproc/src/resize.cpp: `resize` coordinate set-up, `interpolateCubic`, `HResizeCubic`,
`VResizeCubic` with `FixedPtCast<int, uchar, INTER_RESIZE_COEF_BITS*2>`), i.e. its scalar fixed-point path:

  scale = 1.0 / (double(dst) / double(src))                          (double)
  f     = float((d + 0.5) * scale - 0.5);  s = floor(f);  f -= s     (float after the cast)
  c0 = ((A*(f+1) - 5A)*(f+1) + 8A)*(f+1) - 4A;  c1 = ((A+2)*f - (A+3))*f*f + 1;
  c2 = ((A+2)*(1-f) - (A+3))*(1-f)*(1-f) + 1;   c3 = 1 - c0 - c1 - c2            (float, A = -0.75, no contraction)
  ic = saturate_short(round_half_even(c * 2048))
  horizontal: int32 sum of 4 taps (columns clamped to [0, W-1]) * ic;  vertical: int32 sum of 4 rows (clamped) * ic
  out = saturate_u8((v + 2^21) >> 22)

(OpenCV's SIMD builds evaluate the vertical pass in float for the bulk of each row, which can differ from this by one
grey level on exact ties; that, too, cannot be settled without the library.)  Down-scaling does not pre-filter.

`get_size` restates transforms.py:107-176 (the only branch load_transforms uses is resize_method="minimal").
"""
import numpy as np

A = np.float32(-0.75)


def constrain_to_multiple_of(x, multiple_of, min_val=0, max_val=None):
    y = int(np.round(x / multiple_of) * multiple_of)
    if max_val is not None and y > max_val:
        y = int(np.floor(x / multiple_of) * multiple_of)
    if y < min_val:
        y = int(np.ceil(x / multiple_of) * multiple_of)
    return y


def get_size(width, height, net_w, net_h, keep_aspect_ratio, multiple_of=32):
    """(new_width, new_height) of Resize(..., resize_method='minimal')."""
    scale_height = net_h / height
    scale_width = net_w / width
    if keep_aspect_ratio:
        if abs(1 - scale_width) < abs(1 - scale_height):
            scale_height = scale_width
        else:
            scale_width = scale_height
    return (constrain_to_multiple_of(scale_width * width, multiple_of), constrain_to_multiple_of(scale_height * height, multiple_of))


def _axis(dst, src):
    """Per destination index: first tap index (s - 1) and the four int16 coefficients."""
    scale = 1.0 / (float(dst) / float(src))
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    one = np.float32(1.0)
    f1 = f + one
    c0 = ((A * f1 - np.float32(5) * A) * f1 + np.float32(8) * A) * f1 - np.float32(4) * A
    c1 = ((A + np.float32(2)) * f - (A + np.float32(3))) * f * f + one
    g = one - f
    c2 = ((A + np.float32(2)) * g - (A + np.float32(3))) * g * g + one
    c3 = one - c0 - c1 - c2
    c = np.stack([c0, c1, c2, c3], axis=1).astype(np.float32)
    ic = np.clip(np.rint(c * np.float32(2048)), -32768, 32767).astype(np.int64)
    return s - 1, ic


def resize_cubic_u8(img, dst_w, dst_h):
    """img uint8 [H, W, C] -> uint8 [dst_h, dst_w, C]."""
    assert img.dtype == np.uint8 and img.ndim == 3
    H, W, _ = img.shape
    x0, ax = _axis(dst_w, W)
    y0, ay = _axis(dst_h, H)
    src = img.astype(np.int64)
    hor = np.zeros((H, dst_w, img.shape[2]), dtype=np.int64)
    for j in range(4):
        cols = np.clip(x0 + j, 0, W - 1)
        hor += src[:, cols, :] * ax[:, j][None, :, None]
    out = np.zeros((dst_h, dst_w, img.shape[2]), dtype=np.int64)
    for k in range(4):
        rows = np.clip(y0 + k, 0, H - 1)
        out += hor[rows] * ay[:, k][:, None, None]
    out = (out + (1 << 21)) >> 22
    return np.clip(out, 0, 255).astype(np.uint8)


def input_transform(img, net_w, net_h, keep_aspect_ratio=False, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)):
    """uint8 HWC frame -> float32 CHW network input, as Compose([Resize, NormalizeImage, PrepareForNet]) does."""
    w, h = get_size(img.shape[1], img.shape[0], net_w, net_h, keep_aspect_ratio)
    r = resize_cubic_u8(img, w, h)
    x = (r - np.asarray(mean, dtype=np.float64)) / np.asarray(std, dtype=np.float64)   # uint8 - list -> float64 (transforms.py:213)
    return np.ascontiguousarray(np.transpose(x, (2, 0, 1))).astype(np.float32)
