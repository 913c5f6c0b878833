"""CPU: the criterion restatement (oracle/loss_ref.py) against the golden produced by the reference's own ssi_loss module +
torch autograd (oracle/make_golden_loss.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import loss_ref as LR
from tests.golden_inputs import loss_inputs

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss.npz"))


@pytest.mark.parametrize("tag,compute_ss", [("ss", True), ("noss", False)])
def test_loss_oracle_matches_reference_golden(tag, compute_ss):
    torch.set_num_threads(1)
    r = LR.loss_and_grads(*loss_inputs(), compute_ss=compute_ss)
    got = np.array([float(r["loss"]), float(r["loss_disp"]), float(r["loss_seg"])])
    np.testing.assert_allclose(got, G[f"{tag}_loss"], rtol=1e-6)
    np.testing.assert_allclose(r["d_inv"].numpy(), G[f"{tag}_d_inv"], rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(r["d_seg"].numpy(), G[f"{tag}_d_seg"], rtol=1e-5, atol=1e-10)
    assert float(np.abs(G[f"{tag}_d_inv"]).max()) > 0 and int((G[f"{tag}_d_inv"] == 0).sum()) > 0   # clamp hit somewhere


def test_degenerate_masks():
    inv, seg, y_disp, mask_disp, y_seg, mask_seg = loss_inputs()
    mask_disp = torch.zeros_like(mask_disp)            # no valid depth pixel: SSI loss is 0 and carries no gradient
    r = LR.loss_and_grads(inv, seg, y_disp, mask_disp, y_seg, mask_seg)
    assert float(r["loss_disp"]) == 0.0 and float(r["d_inv"].abs().max()) == 0.0
