"""GPU: soccdpt_input_transform_u8 through the C ABI against the oracle, bit for bit (integer resampling, float64 normalisation)."""
import numpy as np
import pytest
import torch

from oracle import input_transform_ref as R
from tests.test_oracle_input_transform import _frame

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("src,net,keep", [((1080, 1920), 256, False), ((1080, 1920), 384, False), ((480, 640), 384, True), ((33, 47), 96, False),
                                           ((256, 256), 256, False), ((7, 5), 32, False)])
def test_bit_exact_against_oracle(src, net, keep):
    from soccdpt_amd.model.transforms import InputTransform
    t = InputTransform(net, net, keep_aspect_ratio=keep)
    frames = np.stack([_frame(src[0], src[1], seed=s) for s in range(3)])
    got = t.batch(torch.from_numpy(frames).cuda()).cpu().numpy()
    for b in range(3):
        ref = R.input_transform(frames[b], net, net, keep_aspect_ratio=keep)
        assert got[b].shape == ref.shape
        assert np.array_equal(got[b], ref), f"frame {b}: {np.abs(got[b] - ref).max()} max difference"


def test_extreme_frames_and_general_mean_std():
    from soccdpt_amd.lib import op_input_transform_u8
    g = np.random.default_rng(0)
    frames = np.stack([np.zeros((90, 120, 3), np.uint8), np.full((90, 120, 3), 255, np.uint8),
                       (g.integers(0, 2, (90, 120, 3)) * 255).astype(np.uint8)])   # hardest case for overshoot / saturation
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    got = op_input_transform_u8(torch.from_numpy(frames).cuda(), 64, 96, mean, std).cpu().numpy()
    for b in range(3):
        r = R.resize_cubic_u8(frames[b], 96, 64)
        ref = ((r - np.asarray(mean)) / np.asarray(std)).transpose(2, 0, 1).astype(np.float32)
        assert np.array_equal(got[b], ref)


def test_numpy_sample_round_trip_and_network_input():
    """The reference's call shape: transform({"image": frame})["image"] -> CHW float32 numpy, fed to the network."""
    from soccdpt_amd.model.loader import load_transforms
    t, w, h = load_transforms("dpt_swin2_tiny_256")
    frame = _frame(1080, 1920, seed=9)
    x = t({"image": frame})["image"]
    assert isinstance(x, np.ndarray) and x.dtype == np.float32 and x.shape == (3, h, w)
    assert np.array_equal(x, R.input_transform(frame, w, h))
    dev = t({"image": torch.from_numpy(frame).cuda()})["image"]
    assert dev.is_cuda and torch.equal(dev.cpu(), torch.from_numpy(x))


def test_bad_arguments_are_refused():
    from soccdpt_amd.lib import op_input_transform_u8
    f = torch.zeros((1, 8, 8, 3), dtype=torch.uint8, device="cuda")
    with pytest.raises(RuntimeError):
        op_input_transform_u8(f, 8, 8, (0.5, 0.5, 0.5), (0.5, 0.0, 0.5))
