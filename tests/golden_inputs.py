"""Seeded inputs of the golden fixtures (shared by oracle/make_golden.py and the tests).
Only outputs are stored in tests/golden/*.npz; inputs are regenerated from these seeds."""
import torch


def proj_inputs(seed: int = 1234, B: int = 2, S: int = 256):
    """inv[B,S,S] in ~[0.005,0.3] with zeros / negatives / NaN / inf / tiny values, and
    ScaledTanh class probabilities with exact zeros (SURVEY.md §8c fixture 1)."""
    g = torch.Generator().manual_seed(seed)
    lo = torch.rand((B, 1, 16, 16), generator=g) * 0.29 + 0.005
    inv = torch.nn.functional.interpolate(lo, size=(S, S), mode="bilinear", align_corners=False)[:, 0]
    inv = inv + torch.randn((B, S, S), generator=g) * 0.002
    inv[:, 5, 7] = 0.0           # clamp path (-> 1e-8 -> depth 1e8)
    inv[:, 9, 11:14] = -0.5      # negative -> clamp
    inv[0, min(100, S - 1), 50] = float("nan")
    inv[B - 1, 33, min(200, S - 1)] = float("inf")
    inv[B - 1, 34, min(200, S - 1)] = 1e-12
    logits = torch.randn((B, 3, S, S), generator=g) * 6.0
    seg = 0.5 * torch.tanh(logits) + 0.5   # ScaledTanh -> exact zeros for logits << 0
    return inv.contiguous(), seg.contiguous()


def decoder_features(seed: int = 77, B: int = 1, dims=(96, 192, 384, 768), res=(64, 32, 16, 8)):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn((B, c, r, r), generator=g) for c, r in zip(dims, res)]


def metrics_inputs(seed: int = 321, B: int = 3, H: int = 135, W: int = 240):
    """Synthetic (prediction, ground truth, mask) triples for the evaluation metrics."""
    g = torch.Generator().manual_seed(seed)
    lo = torch.rand((B, 1, 9, 16), generator=g) * 0.2 + 0.02
    gt = torch.nn.functional.interpolate(lo, size=(H, W), mode="bilinear", align_corners=False)[:, 0]
    pred = 0.7 * gt + 0.013 + 0.004 * torch.randn((B, H, W), generator=g)
    mask = torch.rand((B, H, W), generator=g) > 0.2
    seg_gt = (torch.rand((B, 3, H, W), generator=g) > 0.6).float()
    seg_pred = (0.45 * seg_gt + 0.6 * torch.rand((B, 3, H, W), generator=g)).contiguous()
    return pred.contiguous(), gt.contiguous(), mask.contiguous(), seg_pred, seg_gt


def loss_inputs(B=2, h=64, w=64, H=135, W=240, C=3, seed=7):
    """Seeded inputs of the training criterion (oracle/loss_ref.py): network-resolution predictions (a few non-positive inverse
    depths so the 1e-8 clamp is hit), camera-resolution targets, ~90 % true masks with one fully masked-out row block."""
    g = torch.Generator().manual_seed(seed)
    lo = torch.rand((B, 1, 6, 8), generator=g)
    inv = torch.nn.functional.interpolate(lo, size=(h, w), mode="bilinear", align_corners=False)[:, 0] * 0.3 - 0.02
    inv = inv + 0.01 * torch.randn((B, h, w), generator=g)
    seg = torch.sigmoid(2.5 * torch.randn((B, C, h, w), generator=g))
    y_disp = torch.nn.functional.interpolate(torch.rand((B, 1, 9, 16), generator=g), size=(H, W), mode="bilinear", align_corners=False)[:, 0] * 0.4 + 0.01
    y_seg = (torch.nn.functional.interpolate(torch.rand((B, C, 7, 11), generator=g), size=(H, W), mode="bilinear", align_corners=False) > 0.55).float()
    mask_disp = torch.rand((B, H, W), generator=g) > 0.1
    mask_disp[:, : H // 8] = False
    mask_seg = torch.rand((B, C, H, W), generator=g) > 0.1
    return inv.float().contiguous(), seg.float().contiguous(), y_disp.contiguous(), mask_disp, y_seg.contiguous(), mask_seg


def gt_occ_inputs(H=270, W=480, C=3, seed=11):
    """Seeded frame for the ground-truth occupancy generator (datasets/bdd_helper.py:433-530): a smooth positive disparity with a
    few zeros / NaN / negatives (-> inf / nan / negative depth), a blocky class map, intrinsics scaled to the frame."""
    import numpy as np
    g = torch.Generator().manual_seed(seed)
    lo = torch.rand((1, 1, 9, 16), generator=g)
    # depth = 0.01 * focal / disparity must land in the grid after pc_scale = (500, 2500, 200): depth ~ 0.02 .. 0.1 -> disparity 30 .. 180
    disp = (torch.nn.functional.interpolate(lo, size=(H, W), mode="bilinear", align_corners=False)[0, 0] * 150.0 + 30.0).numpy().astype(np.float32)
    disp[H // 2 + 5, 10:20] = 0.0
    disp[H // 2 + 9, 30:35] = np.nan
    disp[H // 2 + 12, 50:60] = -0.3
    seg = (torch.nn.functional.interpolate(torch.rand((1, 1, 6, 10), generator=g), size=(H, W), mode="nearest")[0, 0] * C).long().clamp(0, C - 1).numpy().astype(np.int64)
    K = np.array([[1250.6 * W / 1920.0, 0.0, 978.4 * W / 1920.0], [0.0, 1254.8 * H / 1080.0, 562.1 * H / 1080.0], [0.0, 0.0, 1.0]])
    return disp, seg, K, H, W, C
