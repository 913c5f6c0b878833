"""CPU: the projection oracles (torch restatement and C restatement) against the golden
vectors captured from the reference's own code (oracle/make_golden.py)."""
import hashlib

import numpy as np
import torch

from oracle import cref, soccdpt_ref as R
from tests.golden_inputs import proj_inputs


def _sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _same(a, b):
    return np.array_equal(np.nan_to_num(a, nan=-7.0), np.nan_to_num(b, nan=-7.0))


def test_c_oracle_matches_golden(golden_dir):
    g = np.load(f"{golden_dir}/projection_B2.npz")
    inv, seg = proj_inputs(int(g["seed"]))
    out = cref.project(inv, seg)
    assert np.array_equal(out["occ_bits"], g["occ_bits"])           # voxel indices: bit-exact
    assert int(np.unpackbits(out["occ_bits"].view(np.uint8)).sum()) == int(g["occ_count"])
    rows = g["rows"]
    assert _same(out["inv_up"][:, rows], g["inv_up_rows"])
    assert _same(out["points"][:, rows], g["points_rows"])
    assert _same(out["seg_up"][:, :, rows], g["seg_up_rows"])
    assert _sha(out["inv_up"]) == str(g["inv_up_sha"])
    assert _sha(out["points"]) == str(g["points_sha"])
    assert _sha(out["seg_up"]) == str(g["seg_up_sha"])


def test_torch_oracle_matches_golden(golden_dir):
    g = np.load(f"{golden_dir}/projection_B2.npz")
    inv, seg = proj_inputs(int(g["seed"]))
    inv_up, seg_up, pts, occ = R.project(inv, seg)
    assert torch.equal(occ[0], occ[1])                              # union over the batch in every row
    assert np.array_equal(cref.pack_occ(occ[0]), g["occ_bits"])
    rows = g["rows"]
    assert _same(inv_up[:, rows].numpy(), g["inv_up_rows"])
    assert _same(pts[:, rows].numpy(), g["points_rows"])


def test_quirks_preserved():
    """3-pixel scale/shift, strict lower bound, B==1 squeeze, clamp aliasing."""
    inv, seg = proj_inputs(B=1)
    inv_up, seg_up, pts, occ = R.project(inv, seg)
    assert tuple(seg_up.shape) == (3, 1080, 1920) and tuple(inv_up.shape) == (1, 1080, 1920)
    c = cref.project(inv, seg)
    # pixels 0,1,2 carry (p*scale+shift); pixel 3 is metric
    d = 1.0 / c["inv_up"][0, 0, :4]
    z = c["points"][0, 0, :4, 2]
    cfg = R.ProjConfig()
    for n in range(3):
        assert z[n] == np.float32(np.float32(d[n]) * np.float32(cfg.pc_scale[n]) + np.float32(cfg.pc_shift[n]))
    assert z[3] == d[3]
    assert float(c["inv_up"].min()) >= 1e-8 or np.isnan(c["inv_up"]).any()
    bits = np.unpackbits(c["occ_bits"].view(np.uint8), bitorder="little").reshape(256, 256, 32, 3)
    assert bits[0].sum() == 0 and bits[:, 0].sum() == 0 and bits[:, :, 0].sum() == 0   # index 0 never set


def test_expand_matches_dense():
    inv, seg = proj_inputs(B=2)
    out = cref.project(inv, seg, want=("occ_bits",))
    occ = R.project(inv, seg)[3]
    dense = np.unpackbits(out["occ_bits"].view(np.uint8), bitorder="little").astype(np.float32)
    assert np.array_equal(dense.reshape(256, 256, 32, 3), occ[1].numpy())
