"""CPU: the input-transform oracle (oracle/input_transform_ref.py, parity UNPINNED: cv2 is absent) against independent
statements of the same mathematics, and the host-side size rule of soccdpt_amd.model.transforms against it."""
import numpy as np
import pytest
import torch

from oracle import input_transform_ref as R


def _frame(h, w, seed):
    g = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = 127 + 90 * np.sin(xx / 37.0 + seed)[..., None] * np.cos(yy / 23.0)[..., None] * np.ones(3)
    return np.clip(base + g.normal(0, 25, (h, w, 3)), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("src,dst", [((1080, 1920), (256, 256)), ((270, 480), (384, 384)), ((64, 96), (64, 96)), ((33, 47), (96, 160))])
def test_fixed_point_resize_tracks_float_bicubic(src, dst):
    """Same taps, same A = -0.75, same half-pixel mapping as torch's bicubic (align_corners=False, no antialias):
    the 11-bit fixed-point result stays within one grey level of the float evaluation (rounded, saturated)."""
    img = _frame(*src, seed=3)
    got = R.resize_cubic_u8(img, dst[1], dst[0]).astype(np.int64)
    t = torch.from_numpy(img).permute(2, 0, 1)[None].double()
    ref = torch.nn.functional.interpolate(t, size=dst, mode="bicubic", align_corners=False)[0].permute(1, 2, 0).numpy()
    ref = np.clip(np.rint(ref), 0, 255)
    assert np.abs(got - ref).max() <= 1
    assert (got != ref).mean() < 0.05


def test_identity_size_is_identity():
    img = _frame(40, 56, seed=5)
    assert np.array_equal(R.resize_cubic_u8(img, 56, 40), img)


def test_constant_image_stays_constant_and_saturation():
    for v in (0, 1, 128, 254, 255):
        img = np.full((50, 70, 3), v, dtype=np.uint8)
        assert np.array_equal(R.resize_cubic_u8(img, 32, 32), np.full((32, 32, 3), v, dtype=np.uint8))
    # overshoot of the cubic kernel at a hard edge must saturate, not wrap
    img = np.zeros((32, 64, 3), dtype=np.uint8)
    img[:, 32:] = 255
    out = R.resize_cubic_u8(img, 200, 32)
    assert out.min() == 0 and out.max() == 255


def test_normalisation_is_the_reference_quirk():
    """(u8 - 0.5) / 0.5 without a /255 (transforms.py:213 on the uint8 frame): values span [-1, 509]."""
    img = _frame(1080, 1920, seed=1)
    x = R.input_transform(img, 256, 256)
    assert x.shape == (3, 256, 256) and x.dtype == np.float32
    r = R.resize_cubic_u8(img, 256, 256)
    assert np.array_equal(x, (2.0 * r.astype(np.float32) - 1.0).transpose(2, 0, 1))


@pytest.mark.parametrize("w,h,net,keep", [(1920, 1080, 256, False), (1920, 1080, 384, False), (1920, 1080, 384, True), (640, 480, 384, True),
                                           (100, 1000, 256, True), (257, 255, 256, False), (4000, 3000, 512, True)])
def test_host_size_rule_matches_oracle(w, h, net, keep):
    from soccdpt_amd.model.transforms import get_size
    assert get_size(w, h, net, net, keep_aspect_ratio=keep, multiple_of=32, resize_method="minimal") == R.get_size(w, h, net, net, keep)
    nw, nh = get_size(w, h, net, net, keep_aspect_ratio=keep, multiple_of=32, resize_method="minimal")
    assert nw % 32 == 0 and nh % 32 == 0


def test_size_rule_bounds():
    from soccdpt_amd.model.transforms import get_size
    # lower_bound never goes below the network size, upper_bound never above (transforms.py:150-166)
    for w, h in ((1920, 1080), (333, 777), (64, 48)):
        lw, lh = get_size(w, h, 384, 384, keep_aspect_ratio=True, multiple_of=32, resize_method="lower_bound")
        uw, uh = get_size(w, h, 384, 384, keep_aspect_ratio=True, multiple_of=32, resize_method="upper_bound")
        assert lw >= 384 and lh >= 384 and uw <= 384 and uh <= 384
    with pytest.raises(ValueError):
        get_size(10, 10, 32, 32, resize_method="nearest")


def test_transform_refuses_cpu_frames_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from soccdpt_amd.model.loader import load_transforms
    t, w, h = load_transforms("dpt_swin2_tiny_256")
    assert (w, h) == (256, 256)
    with pytest.raises(RuntimeError):
        t.batch(torch.zeros((1, 8, 8, 3), dtype=torch.uint8))
    # float frames at network resolution keep the host normalisation path
    s = t({"image": np.full((256, 256, 3), 0.75, dtype=np.float32)})
    assert s["image"].shape == (3, 256, 256) and np.allclose(s["image"], 0.5)
