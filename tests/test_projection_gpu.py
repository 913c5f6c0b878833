"""GPU: the fused HIP projection kernel (through the C ABI) against the C oracle, bit-exact."""
import numpy as np
import pytest
import torch

from oracle import cref, soccdpt_ref as R
from tests.golden_inputs import proj_inputs

pytestmark = pytest.mark.gpu


def _engine(dev, compute_occ=True):
    from soccdpt_amd.lib import Engine, make_config
    cam, cfg = R.Camera(), R.ProjConfig()
    c = make_config("swin2t16_256", 3, 256, False, compute_occ, cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy,
                    cfg.grid_size, cfg.occupancy_shape(), cfg.pc_scale, cfg.pc_shift, cfg.correction_angle)
    return Engine(c, dev)


def _run(eng, inv, seg, dev, occ=True):
    B = inv.shape[0]
    inv_d, seg_d = inv.to(dev), seg.to(dev)
    inv_up = torch.empty((B, 1080, 1920), device=dev)
    seg_up = torch.empty((B, 3, 1080, 1920), device=dev)
    pts = torch.empty((B, 1080, 1920, 3), device=dev)
    bits = torch.full((eng.occ_words(),), -1, dtype=torch.int32, device=dev) if occ else None
    eng.project(inv_d, seg_d, inv_up, seg_up, pts, bits, clear_bits=True)
    torch.cuda.synchronize()
    return inv_up, seg_up, pts, bits


def _bits_np(bits):
    return bits.cpu().numpy().view(np.uint32)


def _same(a, b):
    return np.array_equal(np.nan_to_num(a, nan=-7.0), np.nan_to_num(b, nan=-7.0))


def test_projection_bit_exact_vs_oracle_and_golden(gpu_device, golden_dir):
    eng = _engine(gpu_device)
    g = np.load(f"{golden_dir}/projection_B2.npz")
    inv, seg = proj_inputs(int(g["seed"]))
    inv_up, seg_up, pts, bits = _run(eng, inv, seg, gpu_device)
    ref = cref.project(inv, seg)
    assert _same(inv_up.cpu().numpy(), ref["inv_up"])       # bicubic + clamp: bit-exact
    assert _same(seg_up.cpu().numpy(), ref["seg_up"])       # nearest: exact data movement
    assert _same(pts.cpu().numpy(), ref["points"])          # back-projection incl. the 3-pixel quirk
    assert np.array_equal(_bits_np(bits), ref["occ_bits"])  # voxel indices: bit-exact
    assert np.array_equal(_bits_np(bits), g["occ_bits"])    # ... and equal to the reference's own output
    rows = g["rows"]
    assert _same(inv_up[:, rows].cpu().numpy(), g["inv_up_rows"])
    assert _same(pts[:, rows].cpu().numpy(), g["points_rows"])


def test_projection_b1_and_expand(gpu_device):
    eng = _engine(gpu_device)
    inv, seg = proj_inputs(seed=5, B=1)
    inv_up, seg_up, pts, bits = _run(eng, inv, seg, gpu_device)
    ref = cref.project(inv, seg)
    assert np.array_equal(_bits_np(bits), ref["occ_bits"])
    occ = torch.empty((3, 256, 256, 32, 3), device=gpu_device)
    eng.occ_expand(bits, 3, occ)
    torch.cuda.synchronize()
    dense = np.unpackbits(ref["occ_bits"].view(np.uint8), bitorder="little").astype(np.float32).reshape(256, 256, 32, 3)
    for b in range(3):
        assert np.array_equal(occ[b].cpu().numpy(), dense)


def test_projection_degenerate_inputs(gpu_device):
    """all-zero / all-NaN inverse depth: every point is clamped far away or non-finite -> empty grid."""
    eng = _engine(gpu_device)
    seg = torch.rand(1, 3, 256, 256)
    for fill in (0.0, float("nan"), float("inf"), -1.0):
        inv = torch.full((1, 256, 256), fill)
        inv_up, seg_up, pts, bits = _run(eng, inv, seg, gpu_device)
        ref = cref.project(inv, seg)
        assert np.array_equal(_bits_np(bits), ref["occ_bits"])
        assert _same(inv_up.cpu().numpy(), ref["inv_up"])
        assert _same(pts.cpu().numpy(), ref["points"])
    # all-zero class probabilities: nothing is marked
    inv, _ = proj_inputs(seed=9, B=1)
    inv_up, seg_up, pts, bits = _run(eng, inv, torch.zeros(1, 3, 256, 256), gpu_device)
    assert int(_bits_np(bits).sum()) == 0


def test_projection_full_batch_properties(gpu_device):
    """BASELINE config 2 size (B=8): union property, idempotence, batch-order invariance."""
    eng = _engine(gpu_device)
    inv, seg = proj_inputs(seed=21, B=8)
    _, _, pts8, bits8 = _run(eng, inv, seg, gpu_device)
    b8 = _bits_np(bits8).copy()
    # union over frames == OR of the per-frame grids
    acc = np.zeros_like(b8)
    for b in range(8):
        _, _, ptsb, bitsb = _run(eng, inv[b:b + 1], seg[b:b + 1], gpu_device)
        acc |= _bits_np(bitsb)
        assert torch.equal(torch.nan_to_num(ptsb[0], nan=-7.0), torch.nan_to_num(pts8[b], nan=-7.0))
    assert np.array_equal(acc, b8)
    # idempotent: projecting again without clearing changes nothing
    eng.project(inv.to(gpu_device), seg.to(gpu_device), None, None, None, bits8, clear_bits=False)
    torch.cuda.synchronize()
    assert np.array_equal(_bits_np(bits8), b8)
    # order of frames does not matter
    perm = torch.randperm(8, generator=torch.Generator().manual_seed(0))
    _, _, _, bitsp = _run(eng, inv[perm].contiguous(), seg[perm].contiguous(), gpu_device)
    assert np.array_equal(_bits_np(bitsp), b8)
    # occ_or of split halves == whole
    _, _, _, lo = _run(eng, inv[:4].contiguous(), seg[:4].contiguous(), gpu_device)
    _, _, _, hi = _run(eng, inv[4:].contiguous(), seg[4:].contiguous(), gpu_device)
    both = torch.stack([lo, hi]).contiguous()
    dst = torch.zeros_like(lo)
    eng.occ_or(dst, both, 2)
    torch.cuda.synchronize()
    assert np.array_equal(_bits_np(dst), b8)
    # spot-check two frames of the big batch against the C oracle
    ref = cref.project(inv[:2], seg[:2], want=("points",))
    assert _same(pts8[:2].cpu().numpy(), ref["points"])


def test_projection_skips_voxelisation_without_bits(gpu_device):
    eng = _engine(gpu_device, compute_occ=False)
    inv, seg = proj_inputs(seed=3, B=1)
    inv_up, seg_up, pts, _ = _run(eng, inv, seg, gpu_device, occ=False)
    ref = cref.project(inv, seg, want=("inv_up", "points"))
    assert _same(inv_up.cpu().numpy(), ref["inv_up"]) and _same(pts.cpu().numpy(), ref["points"])
