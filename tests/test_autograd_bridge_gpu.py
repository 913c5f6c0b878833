"""GPU: SOccDPT_V3.forward in train mode is an autograd node.  The reference's loop body
(/root/reference/SOccDPT/scripts/train_SOccDPT.py:365-393) -- `net_patch(x)`, a criterion written in torch ops on the returned camera-resolution
tensors, `grad_scaler.scale(loss).backward()`, `grad_scaler.step(optimizer)` -- runs unchanged on the HIP model: backward =
soccdpt_project_backward (bicubic + clamp / nearest / back-projected points) + soccdpt_train_backward."""
import os
import tempfile

import pytest
import torch
import torch.nn.functional as F

from oracle import loss_ref
from oracle import soccdpt_ref as R

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def _make(gpu_device, compute_occ=False):
    from soccdpt_amd.lib import PREC_F32
    from soccdpt_amd.model.SOccDPT import SOccDPT_V3
    from soccdpt_amd.utils.synth import synth_state_dict, write_synth_calib
    calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
    m = SOccDPT_V3(sigmoid=True, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=compute_occ, precision=PREC_F32)
    sd = synth_state_dict(alias_pretrained=True)
    m.load_state_dict(sd, strict=False)
    m.drop_path_rate = 0.0      # deterministic parity runs (the model's default follows timm: 0.1 for the Swin-V2 encoders)
    return m.to(gpu_device), sd


@pytest.mark.parametrize("clamped", [False, True])
def test_project_backward_matches_autograd(gpu_device, clamped):
    """soccdpt_project_backward against float64 torch autograd through the same tail: bicubic (align_corners=False) + clamp at 1e-8 (zero gradient
    where clamped), nearest, points = ((v - cx) d / fx, (u - cy) d / fy, d) with the 3-pixel pc_scale quirk.  clamped=False: random upstream
    gradients on all three outputs (inverse depth bounded away from zero: the points' 1 / inv^2 factor is then well conditioned);
    clamped=True: a negative patch of the inverse depth (the forward clamps it to 1e-8) with gradients on inv_up / seg_up."""
    m, _ = _make(gpu_device)
    eng = m._engine(gpu_device)
    g = torch.Generator().manual_seed(0)
    B, h, w, Hc, Wc = 2, 64, 64, m.height, m.width
    inv = torch.rand((B, h, w), generator=g) * 0.1 + 0.2
    if clamped:
        inv[0, 10:14, 20:30] = -0.5
    seg = torch.rand((B, 3, h, w), generator=g)
    inv_up = torch.empty((B, Hc, Wc), device=gpu_device); seg_up = torch.empty((B, 3, Hc, Wc), device=gpu_device)
    pts = torch.empty((B, Hc, Wc, 3), device=gpu_device)
    eng.project(inv.to(gpu_device), seg.to(gpu_device), inv_up, seg_up, pts, None)
    w1 = torch.randn((B, Hc, Wc), generator=g); w2 = torch.randn((B, 3, Hc, Wc), generator=g)
    w3 = None if clamped else torch.randn((B, Hc, Wc, 3), generator=g) * 1e-2
    d_inv, d_seg = eng.project_backward(inv_up, w1.to(gpu_device), w2.to(gpu_device), None if w3 is None else w3.to(gpu_device), h, w)
    d_inv1, d_seg1 = eng.project_backward(inv_up, w1.to(gpu_device), None, None, h, w)          # absent gradients are zeros
    torch.cuda.synchronize()
    # float64 reference
    a = inv.double().requires_grad_(True); s = seg.double().requires_grad_(True)
    up = F.interpolate(a.unsqueeze(1), size=(Hc, Wc), mode="bicubic", align_corners=False)[:, 0]
    up = torch.where(up < 1e-8, torch.full_like(up, 1e-8), up)
    su = F.interpolate(s, size=(Hc, Wc), mode="nearest")
    loss = (up * w1.double()).sum() + (su * w2.double()).sum()
    if w3 is not None:
        d = 1.0 / up
        vv = torch.arange(Wc, dtype=torch.float64)[None, None, :]; uu = torch.arange(Hc, dtype=torch.float64)[None, :, None]
        P = torch.stack([(vv - float(m.cx)) * d / float(m.fx), (uu - float(m.cy)) * d / float(m.fy), d.expand(B, Hc, Wc)], dim=-1)
        scale = torch.ones((Hc * Wc, 1), dtype=torch.float64); scale[:3, 0] = torch.tensor(m.pc_scale, dtype=torch.float64)
        loss = loss + (P * scale.reshape(1, Hc, Wc, 1) * w3.double()).sum()
    loss.backward()
    e_inv, e_seg = _rel(d_inv.cpu(), a.grad), _rel(d_seg.cpu(), s.grad)
    print(f"project_backward (clamped={clamped}): rel L2 vs float64 autograd d_inv {e_inv:.2e}, d_seg {e_seg:.2e}")
    # clamped=True: pixels whose raw bicubic value is within f32 rounding of the 1e-8 threshold may fall on the other side in float64
    assert e_inv < (2e-3 if clamped else 2e-5), e_inv
    assert e_seg < 2e-6
    if clamped:   # the clamp really cut gradient paths: without the mask the result differs grossly
        up2 = F.interpolate(a.detach().unsqueeze(1), size=(Hc, Wc), mode="bicubic", align_corners=False)[:, 0]
        assert int((up2 < 1e-8).sum()) > 1000
    assert float(d_seg1.abs().max()) == 0.0 and float(d_inv1.abs().max()) > 0.0


@pytest.mark.parametrize("patchwise_percentage", [1.0, 0.5])
def test_reference_loop_body_runs_unchanged(gpu_device, patchwise_percentage):
    """The reference's inner loop, statement for statement, on the HIP model: the criterion is plain torch ops (the pinned restatement of
    ssi_loss + masked BCE, run on the GPU tensors forward() returned), torch.cuda.amp.GradScaler and torch.optim.Adam are torch's own.  The
    parameter gradients equal those of torch autograd over the oracle network + the same criterion within 1e-3 (the bound of
    test_training_step_gradients_with_criterion), and the optimizer step moves exactly the trainable tensors."""
    from soccdpt_amd.scripts.train_SOccDPT import SyntheticDepthSegDataset, get_batch
    from soccdpt_amd.utils.optim import PatchWiseInplace
    from soccdpt_amd.utils.synth import synth_input
    net, sd = _make(gpu_device)
    net.train()
    net.seg_head[3].p = 0.0
    for p in net.parameters():
        p.requires_grad_(True)
    B = 3
    x = synth_input(B, seed0=3)
    _, _, mask_disp, y_disp, mask_seg, y_seg = get_batch(SyntheticDepthSegDataset(B, 256), B, B)
    y_disp, y_seg = y_disp.float(), y_seg.float()
    sd_o = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v.clone()) for k, v in sd.items()}
    o_inv, o_seg, _ = R.soccdpt_v3_network(sd_o, x, sigmoid=True, training=True)
    o_loss = loss_ref.training_loss(o_inv, o_seg, y_disp, mask_disp, y_seg, mask_seg, 0.5, 0.5, True)[0]
    o_loss.backward()

    device = gpu_device
    criterion_disp = loss_ref.ssi_loss
    criterion_seg = loss_ref.bce_masked
    optimizer = torch.optim.Adam(net.parameters(), lr=1e-5, betas=(0.9, 0.999), eps=1e-08, weight_decay=0.0, amsgrad=False)
    grad_scaler = torch.cuda.amp.GradScaler(enabled=False)
    loss_depth_w, loss_seg_w = 0.5, 0.5
    x = x.to(device=device, dtype=torch.float32)
    y_disp, y_seg = y_disp.to(device), y_seg.to(device)
    mask_disp, mask_seg = mask_disp.to(device=device, dtype=torch.bool), mask_seg.to(device=device, dtype=torch.bool)
    before = {k: p.detach().clone() for k, p in net.named_parameters()}
    got, n_patches = {}, 0
    for net_patch in PatchWiseInplace(net, patchwise_percentage):            # ---- scripts/train_SOccDPT.py:362-393 ----
        y_disp_pred, y_seg_pred, points, y_occupancy_grid = net_patch(x)
        if len(y_seg_pred.shape) == 3:
            y_seg_pred = y_seg_pred.unsqueeze(0)
        if len(y_disp_pred.shape) == 2:
            y_disp_pred = y_disp_pred.unsqueeze(0)
        loss_disp = criterion_disp(y_disp_pred, y_disp, mask_disp)
        loss_seg = criterion_seg(y_seg_pred, y_seg, mask_seg)
        loss = loss_depth_w * loss_disp + loss_seg_w * loss_seg
        optimizer.zero_grad(set_to_none=True)
        grad_scaler.scale(loss).backward()
        for k, p in net.named_parameters():                                    # (inspection, not part of the loop body)
            if p.grad is not None:
                assert k not in got, f"{k} is in two patches"
                got[k] = p.grad.detach().cpu().clone()
        grad_scaler.step(optimizer)
        grad_scaler.update()
        n_patches += 1
        if n_patches == 1:
            assert abs(float(loss) - float(o_loss)) < 2e-4 * abs(float(o_loss)), (float(loss), float(o_loss))
            assert tuple(y_disp_pred.shape) == (B, net.height, net.width) and tuple(points.shape) == (B, net.height, net.width, 3) and y_occupancy_grid is None
    torch.cuda.synchronize()
    assert n_patches == (1 if patchwise_percentage == 1.0 else 2)
    if patchwise_percentage == 1.0:       # (with two patches the second forward sees the first patch's update: only the first is comparable)
        errs = []
        for k, p in net.named_parameters():
            ref = sd_o[k].grad
            if ref is None:
                continue
            assert k in got, f"no gradient for {k}"
            if float(ref.norm()) < 1e-5:
                assert float((got[k] - ref).norm()) < 1e-5, k
                continue
            errs.append((_rel(got[k], ref), k))
        print(f"reference loop body: {len(errs)} parameter gradients vs torch autograd over the oracle: median {sorted(e for e, _ in errs)[len(errs) // 2]:.2e}, "
              f"worst {max(errs)[0]:.2e} ({max(errs)[1]})")
        assert not [(e, k) for e, k in errs if e > 1e-3]
    moved = [k for k, p in net.named_parameters() if not torch.equal(p.detach(), before[k])]
    assert len(moved) >= len(got) - 2 and len(got) > 200        # Adam moved what received a gradient (a zero gradient may leave a tensor in place)


def test_train_mode_forward_guards(gpu_device):
    """A second train-mode forward invalidates the first one's tape (one tape per handle): its backward raises instead of returning a wrong
    gradient; under torch.no_grad() the train-mode forward returns plain tensors; B == 1 keeps the reference's squeezed segmentation."""
    from soccdpt_amd.utils.synth import synth_input
    net, _ = _make(gpu_device, compute_occ=True)
    net.train()
    x = synth_input(1, seed0=1).to(gpu_device)
    a = net(x)
    assert tuple(a[1].shape) == (3, net.height, net.width) and a[0].requires_grad and tuple(a[3].shape) == (1, 256, 256, 32, 3) and not a[3].requires_grad
    b = net(x)
    with pytest.raises(RuntimeError, match="most recent"):
        a[0].sum().backward()
    b[1].sum().backward()
    assert net.seg_head[4].weight.grad is not None
    with torch.no_grad():
        c = net(x)
    assert not c[0].requires_grad
