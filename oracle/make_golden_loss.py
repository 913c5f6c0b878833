"""Generates tests/golden/loss.npz by running THE REFERENCE'S OWN criterion code in this container:
`SOccDPT.loss.ssi_loss.ScaleAndShiftInvariantLoss` (imports as-is: torch only) + `torch.nn.BCELoss` composed exactly as
`scripts/train_SOccDPT.py:323-338,368-386` does, on the prediction side of `model/SOccDPT.py:264-290` (bicubic / nearest
up-sampling, in-place 1e-8 clamp), with torch autograd for the gradients w.r.t. the network outputs.
Runs only where /root/reference exists; the .npz it writes is the committed fixture.

    python oracle/make_golden_loss.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from tests.golden_inputs import loss_inputs  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_ssi_loss", "/root/reference/SOccDPT/loss/ssi_loss.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)   # the module imports torch / torch.nn only


def reference_step(inv, seg, y_disp, mask_disp, y_seg, mask_seg, compute_ss, w_d=0.5, w_s=0.5):
    inv = inv.clone().requires_grad_(True)
    seg = seg.clone().requires_grad_(True)
    H, W = y_disp.shape[-2:]
    # model/SOccDPT.py:270-290
    inv_depth = torch.nn.functional.interpolate(inv.unsqueeze(1), size=(H, W), mode="bicubic", align_corners=False).squeeze()
    segmentation = torch.nn.functional.interpolate(seg, size=(H, W), mode="nearest").squeeze()
    if len(inv_depth.shape) == 2:
        inv_depth = inv_depth.unsqueeze(0)
    depth = inv_depth
    depth[depth < 1e-8] = 1e-8
    y_disp_pred, y_seg_pred = inv_depth, segmentation
    # scripts/train_SOccDPT.py:323-338,368-386
    crit = ref.ScaleAndShiftInvariantLoss(compute_scale_and_shift=compute_ss)
    bce = torch.nn.BCELoss(reduction="mean")
    loss_disp = crit(y_disp_pred, y_disp, mask_disp)
    loss_seg = bce(torch.masked_select(y_seg_pred, mask_seg), torch.masked_select(y_seg, mask_seg))
    loss = w_d * loss_disp + w_s * loss_seg
    loss.backward()
    return loss.detach(), loss_disp.detach(), loss_seg.detach(), inv.grad, seg.grad


def patchwise_schedule():
    """requires_grad patterns produced by THE REFERENCE'S PatchWiseInplace (patchwise_training/__init__.py:148-252) on a small
    module with some frozen parameters -> tests/golden/patchwise.json."""
    import json
    spec2 = importlib.util.spec_from_file_location("ref_patchwise", "/root/reference/SOccDPT/patchwise_training/__init__.py")
    pw = importlib.util.module_from_spec(spec2)
    spec2.loader.exec_module(pw)
    out = {}
    for pct in (1.0, 0.5, 0.3, 0.1, 0.01):
        net = torch.nn.Sequential(*[torch.nn.Linear(3, 3) for _ in range(7)])      # 14 parameters
        frozen = (0, 5, 6)
        for i, p in enumerate(net.parameters()):
            p.requires_grad = i not in frozen
        it = pw.PatchWiseInplace(net, pct)
        pats = []
        for net_patch in it:
            pats.append([int(p.requires_grad) for p in net_patch.parameters()])
        out[str(pct)] = dict(len=len(it), patterns=pats, after=[int(p.requires_grad) for p in net.parameters()])
    path = os.path.join(REPO, "tests", "golden", "patchwise.json")
    json.dump(dict(n_params=14, frozen=[0, 5, 6], schedules=out), open(path, "w"), indent=1)
    print("wrote", path)


def main():
    patchwise_schedule()
    torch.manual_seed(0)
    torch.set_num_threads(1)
    out = {}
    for tag, compute_ss in (("ss", True), ("noss", False)):
        ins = loss_inputs()
        loss, ld, ls, gi, gs = reference_step(*ins, compute_ss=compute_ss)
        out[f"{tag}_loss"] = np.array([float(loss), float(ld), float(ls)], dtype=np.float64)
        out[f"{tag}_d_inv"] = gi.numpy()
        out[f"{tag}_d_seg"] = gs.numpy()
    path = os.path.join(REPO, "tests", "golden", "loss.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: (v.shape, float(np.abs(v).max())) for k, v in out.items()})


if __name__ == "__main__":
    main()
