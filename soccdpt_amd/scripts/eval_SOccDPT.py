"""Counterpart of the reference's evaluation entry point (`/root/reference/SOccDPT/scripts/eval_SOccDPT.py`,
invoked by `scripts/eval.sh:3-13` as `python -m SOccDPT.scripts.eval_SOccDPT -v 3 -dt bdd -t dpt_swin2_tiny_256 ...`):
same flags, same protocol — load the model through `load_model`, run the FPS loop (50 forwards,
eval_SOccDPT.py:246-259), then `evaluate_seg` / `evaluate_depth` on the validation subset and print the same lines.

Differences, all forced by what exists on a GPU box: the datasets (and cv2) are not there, so when `--base_path`
does not exist the 10-image validation subset is synthetic (seeded frames, ground truth = a smooth perturbation of
the CPU-free model output, so the numbers are meaningful only as a smoke/regression signal); the PNG visual dumps of
eval_SOccDPT.py:136-243 are skipped; the FPS loop synchronises the stream before stopping the clock (the reference does
not); metrics run on the GPU (soccdpt_amd.utils.metrics).  `-l/--load` may be omitted for random weights.

    python -m soccdpt_amd.scripts.eval_SOccDPT -v 3 -dt bdd -t dpt_swin2_tiny_256 -d cuda:0 [-l ckpt.pth] [-o]
"""
import argparse
import os
import random
import tempfile
import time

import numpy as np
import torch

from ..model.loader import load_model, load_transforms
from ..model.SOccDPT import DepthNet, SegNet, SOccDPT_versions, model_types
from ..utils.metrics import evaluate_depth, evaluate_seg
from ..utils.synth import synth_input, synth_state_dict, write_synth_calib


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(description="Evaluate SOccDPT")
    parser.add_argument("-v", "--version", choices=[1, 2, 3], required=True, type=int, help="SOccDPT version")
    parser.add_argument("-dt", "--dataset", choices=["bdd", "idd"], required=True, help="Dataset to evaluate on")
    parser.add_argument("-t", "--model_type", choices=model_types, required=True, help="Model architecture to use")
    parser.add_argument("-d", "--device", default="cpu", help="Device (the HIP path needs cuda:N; cpu raises like any missing-GPU use)")
    parser.add_argument("-l", "--load", default=None, help="Checkpoint path (omit: deterministic synthetic weights)")
    parser.add_argument("-cm", "--compile", action="store_true", help="accepted for compatibility; the HIP path has no tracing compiler")
    parser.add_argument("-o", "--optimize", action="store_true", help="fp16 operands (the reference's net.half())")
    parser.add_argument("-b", "--base_path", default=os.path.expanduser("~/Datasets/Depth_Dataset_Bengaluru"), help="Base path to dataset")
    parser.add_argument("-ld", "--load_depth", default=None, help="Which depth checkpoint to load")
    parser.add_argument("-ls", "--load_seg", default=None, help="Which seg checkpoint to load")
    parser.add_argument("--camera_intrinsics_yaml", default=None, help="calibration file (default: the synthetic 1920x1080 camera of SURVEY.md 8d)")
    return parser


def synthetic_val_set(net, device, img: int, n: int = 10):
    """n batches (x, x_raw, mask_disp, y_disp, mask_seg, y_seg) in the datasets' layout (bengaluru_driving_dataset.py:118-140):
    GT at camera resolution, masks all-true.  GT = smooth perturbation of the model's own output."""
    H, W = net.height, net.width
    g = torch.Generator().manual_seed(0)
    out = []
    for i in range(n):
        x = synth_input(1, size=img, seed0=100 + i)
        with torch.no_grad():
            inv, seg, _, _ = net(x.to(device))
        inv, seg = inv.float().cpu().reshape(1, H, W), seg.float().cpu().reshape(1, -1, H, W)
        pert = torch.nn.functional.interpolate(torch.rand((1, 1, 9, 16), generator=g), size=(H, W), mode="bilinear", align_corners=False)[:, 0]
        blobs = torch.nn.functional.interpolate(torch.rand((1, seg.shape[1], 12, 20), generator=g), size=(H, W), mode="bilinear", align_corners=False)
        y_disp = inv * (0.7 + 0.6 * pert) + 0.01 * pert
        y_seg = ((seg > 0.5) ^ (blobs > 0.8)).float()
        out.append((x, None, torch.ones_like(y_disp, dtype=torch.bool), y_disp, torch.ones_like(y_seg, dtype=torch.bool), y_seg))
    return out


@torch.no_grad()
def main(args) -> dict:
    print(f"Model: SOccDPT_V{str(args.version)}_{args.model_type}")
    SOccDPT = SOccDPT_versions[args.version]
    device = torch.device(args.device)
    _, net_w, net_h = load_transforms(model_type=args.model_type)
    if "idd" in args.dataset:   # the class count comes from the IDD label tables (datasets/anue_labels.py), dataset code outside the hot path
        raise NotImplementedError("-dt idd needs the IDD label tables; only the 3-class bdd layout (eval_SOccDPT.py:72-77) is built")
    num_classes = 3
    calib = args.camera_intrinsics_yaml or write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
    model_kwargs = dict(num_classes=num_classes, camera_intrinsics_yaml=calib)
    if args.version == 1:
        model_kwargs["load_depth"] = args.load_depth
        model_kwargs["load_seg"] = args.load_seg
    elif args.version == 2:
        assert args.load_depth is None or args.load_depth is False, "V2 does not support loading depth"
        assert args.load_seg is None or args.load_seg is False, "V2 does not support loading seg"
    elif args.version == 3:
        model_kwargs["load_depth"] = args.load_depth if args.load_depth is not None else False
        assert args.load_seg is None or args.load_seg is False, "V3 does not support loading seg"
    net = load_model(arch=SOccDPT, model_kwargs=model_kwargs, device=torch.device("cpu"), model_path=args.load,
                     model_type=args.model_type)
    if args.load is None:
        from ..model.spec import MODEL_TYPE_TO_BACKBONE
        net.load_state_dict(synth_state_dict(MODEL_TYPE_TO_BACKBONE[args.model_type], num_classes=num_classes, alias_pretrained=True), strict=False)
    if args.optimize and hasattr(net, "precision"):
        from ..lib import PREC_F16
        net.precision = PREC_F16
    net = net.to(device=device).eval()
    print("Model Parameters: {:.2f}M".format(sum(p.numel() for p in net.parameters()) / 1e6))

    random.seed(0)
    np.random.seed(0)
    torch.manual_seed(0)
    if os.path.isdir(args.base_path):
        raise NotImplementedError("dataset readers (cv2, csv) are outside the hot path and not part of this build; "
                                  "pass a --base_path that does not exist to evaluate on the synthetic subset")
    dataset = synthetic_val_set(net, device, net_w, n=10)
    x = dataset[-1][0].to(device=device, dtype=torch.float32)

    frame_count = 50          # eval_SOccDPT.py:246-259
    for _ in range(5):
        _ = net(x)
    torch.cuda.synchronize(device)
    start_time = time.time()
    for _ in range(frame_count):
        _ = net(x)
    torch.cuda.synchronize(device)
    end_time = time.time()
    fps = frame_count / (end_time - start_time)
    print(f"FPS: {fps:.2f} ({frame_count} frames in {(end_time - start_time):.4f} seconds)")

    iou = evaluate_seg(SegNet(net), dataset, device, amp=False)
    abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3 = evaluate_depth(DepthNet(net), dataset, device, amp=False)
    print(f"IOU: {iou:.4f}")
    print(f"ABS_REL: {abs_rel:.4f}")
    print(f"SQ_REL: {sq_rel:.4f}")
    print(f"RMSE: {rmse:.4f}")
    print(f"RMSE_LOG: {rmse_log:.4f}")
    print(f"A1: {a1:.4f}")
    print(f"A2: {a2:.4f}")
    print(f"A3: {a3:.4f}")
    print("=" * 20)
    return dict(fps=fps, iou=iou, abs_rel=abs_rel, sq_rel=sq_rel, rmse=rmse, rmse_log=rmse_log, a1=a1, a2=a2, a3=a3)


if __name__ == "__main__":
    main(build_parser().parse_args())
