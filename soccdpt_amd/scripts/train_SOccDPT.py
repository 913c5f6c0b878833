"""Counterpart of the reference's training entry point (/root/reference/SOccDPT/scripts/train_SOccDPT.py): the same command line
(`-v/--version`, `-n/--count`, `-dt/--dataset`, `-t/--model_type`, `-d/--device`, `-c/--checkpoint_dir`, `-b/--base_path`, `--sweep_json`,
:485-546), the same wandb-sweep JSON schema (`config/*.json`: `parameters.<name>.values`, :452-479), the same run set-up -- seeding
(:150-154), dataset split (:199-224), `load_model` (:248-253), `freeze_pretrained_encoder` + `unfreeze_pretrained_encoder_by_percentage`
(:262-263), Adam (:311-318), ReduceLROnPlateau(min, patience 2) (:320-322), the `PatchWiseInplace` inner loop (:362-393), the criterion
`loss_depth_w * SSI + loss_seg_w * BCE` (:323-338,380-386), the evaluation round every n_train // (3 * batch_size) steps (:406-430: the seven depth
metrics and the IoU over the validation split, here on the GPU through csrc/metrics.hip; wandb histograms / images are not produced), one
checkpoint per epoch (:437-449).

What runs on MI355X: all of it.  The train-mode forward and the network backward (csrc/train_step.cpp, csrc/train_hybrid_step.cpp, csrc/train*.hip: exact f32, all
three models) sit behind `net.train_forward(x)` / `net.backward(d_inv, d_seg)`; the criterion and its gradient w.r.t. the network outputs are
csrc/loss.hip, the optimizer is the fused Adam (csrc/adam.hip).  A frozen parameter (freeze / unfreeze helpers, PatchWiseInplace) has no
gradient bound and its weight-gradient GEMM is skipped.  `--forward_only` walks the schedule with the eval-mode forward + criterion only
(any operand precision).  There is no autograd / eager-PyTorch fallback: the HIP library is
the product.

Differences forced by the GPU box: wandb is absent -> the sweep is sampled locally (`method: random` with random.seed(0), `count` runs)
and logging goes to stdout; the datasets are absent -> when `--base_path` does not exist a seeded synthetic set in the datasets' layout
(bengaluru_driving_dataset.py:104-137: `[x, x_raw, mask_disp, y_disp, mask_seg, y_seg]`, ground truth at 1920 x 1080) is used; `--model_type`
is forwarded into the model (the reference only uses it for the transform and the project name).

    python -m soccdpt_amd.scripts.train_SOccDPT -v 3 -dt bdd -t dpt_swin2_tiny_256 -d cuda:0 \\
        --sweep_json config/SOccDPT_V3_dpt_swin2_tiny_256_Aug_22.json [--forward_only] [--max_steps N]
"""
import argparse
import json
import os
import random
import tempfile
from pathlib import Path

import numpy as np
import torch

from ..loss import freeze_pretrained_encoder, unfreeze_pretrained_encoder_by_percentage
from ..model.loader import load_model, load_transforms
from ..model.SOccDPT import DepthNet, SegNet, SOccDPT_versions, model_types
from ..utils.loss import training_loss
from ..utils.metrics import evaluate_depth, evaluate_seg
from ..utils.optim import Adam, GradScaler, PatchWiseInplace
from ..utils.synth import synth_input, write_synth_calib

# keys of the reference's sweep files (config/*.json) and the defaults train_net() gives them (train_SOccDPT.py:96-121)
SWEEP_DEFAULTS = dict(amp=False, epochs=5, batch_size=1, val_percent=0.1, weight_decay=1e-8, learning_rate=1e-5, save_checkpoint=True,
                      encoder_percentage=1.0, patchwise_percentage=1.0, dataset_percentage=1.0, loss_weights=[0.5, 0.5], load=False,
                      load_depth=False, load_seg=False, compute_scale_and_shift=True, sigmoid=True)


def read_sweep(path: str):
    """-> (method, {name: [values]}) of a wandb sweep specification (train_SOccDPT.py:452-455)."""
    with open(path, "r") as f:
        cfg = json.load(f)
    assert "parameters" in cfg, f"{path}: not a sweep specification (no 'parameters')"
    params = {}
    for name, spec in cfg["parameters"].items():
        if "values" in spec:
            params[name] = list(spec["values"])
        elif "value" in spec:
            params[name] = [spec["value"]]
        else:
            raise AssertionError(f"{path}: parameter '{name}' has neither 'values' nor 'value'")
    return cfg.get("method", "random"), params


def sample_runs(method: str, params: dict, count: int, seed: int = 0):
    """`count` run configurations: 'grid' walks the cartesian product in file order, anything else ('random', 'bayes') draws every parameter
    uniformly from its values with a seeded generator (what the wandb agent does for method 'random')."""
    names = list(params)
    runs = []
    if method == "grid":
        import itertools
        for combo in itertools.islice(itertools.product(*[params[n] for n in names]), count):
            runs.append(dict(zip(names, combo)))
    else:
        rng = random.Random(seed)
        for _ in range(count):
            runs.append({n: rng.choice(params[n]) for n in names})
    return runs


class SyntheticDepthSegDataset(torch.utils.data.Dataset):
    """Items in the datasets' layout (bengaluru_driving_dataset.py:104-137), seeded per index: network input through NormalizeImage WITHOUT /255
    (SURVEY.md 3.4), a smooth positive disparity and three blob masks at camera resolution, masks all true."""

    def __init__(self, n: int, net_size: int, H: int = 1080, W: int = 1920, num_classes: int = 3):
        self.n, self.net_size, self.H, self.W, self.C = n, net_size, H, W, num_classes

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(1000 + i)
        x = synth_input(1, size=self.net_size, seed0=5000 + i)
        lo = torch.rand((1, 1, 9, 16), generator=g) * 0.25 + 0.02
        y_disp = torch.nn.functional.interpolate(lo, size=(self.H, self.W), mode="bilinear", align_corners=False)[:, 0]
        blobs = torch.nn.functional.interpolate(torch.rand((1, self.C, 7, 11), generator=g), size=(self.H, self.W), mode="bilinear", align_corners=False)
        y_seg = (blobs > 0.55).float()
        return [x, None, torch.ones_like(y_disp, dtype=torch.bool), y_disp, torch.ones_like(y_seg, dtype=torch.bool), y_seg]


def get_batch(dataset, index: int, batch_size: int):
    """utils/__init__.py:768-780: items [index - batch_size, index) concatenated along dim 0."""
    items = [dataset[i] for i in range(index - batch_size, index)]
    cat = lambda k: None if items[0][k] is None else torch.cat([it[k] for it in items], dim=0)
    return [cat(k) for k in range(6)]


class ReduceLROnPlateau:
    """torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, 'min', patience=2) for the fused Adam (factor 0.1, rel. threshold 1e-4)."""

    def __init__(self, optimizer, patience: int = 2, factor: float = 0.1, threshold: float = 1e-4, min_lr: float = 0.0):
        self.opt, self.patience, self.factor, self.threshold, self.min_lr = optimizer, patience, factor, threshold, min_lr
        self.best, self.bad = float("inf"), 0

    def step(self, metric: float):
        if metric < self.best * (1.0 - self.threshold):
            self.best, self.bad = metric, 0
        else:
            self.bad += 1
        if self.bad > self.patience:
            self.opt.lr = max(self.opt.lr * self.factor, self.min_lr)
            self.bad = 0


def train_net(SOccDPT_version=3, device="cuda:0", model_type="dpt_swin2_tiny_256", checkpoint_dir="checkpoints", dataset="bdd", base_path="",
              forward_only=False, max_steps=0, run_id="dummy_run", n_synthetic=12, camera_intrinsics_yaml=None, **cfg):
    p = dict(SWEEP_DEFAULTS)
    unknown = [k for k in cfg if k not in p]
    assert not unknown, f"unknown sweep parameters: {unknown}"
    p.update(cfg)
    loss_weights = p["loss_weights"]
    assert type(loss_weights) == list, "loss_weights must be a list"
    loss_depth_w, loss_seg_w = [float(w) for w in loss_weights]
    assert loss_depth_w >= 0.0 and loss_seg_w >= 0.0, "loss_weights must be >= 0.0"
    device = torch.device(device)
    # Data parallel (not in the reference, which trains on one device): under torch.distributed.run every rank takes a contiguous shard of each
    # batch, gradients are averaged by one bucketed all-reduce per step (soccdpt_amd.dist.attach_training), rank 0 logs and writes checkpoints.
    from .. import dist as sdist
    rank, local_rank, world = 0, 0, 1
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        rank, local_rank, world = sdist.init_from_env("nccl")
        device = torch.device("cuda", local_rank)
        torch.cuda.set_device(device)
    # REPRODUCIBILITY (train_SOccDPT.py:150-154)
    random.seed(0)
    np.random.seed(0)
    torch.manual_seed(0)
    torch.use_deterministic_algorithms(True, warn_only=True)

    transforms, net_w, net_h = load_transforms(model_type=model_type)
    if base_path and os.path.isdir(os.path.expanduser(base_path)):
        raise RuntimeError("dataset readers (cv2 / pandas pipelines of SOccDPT/datasets) are outside the hot-path scope (SURVEY.md 2); "
                           "point --base_path at a non-existing directory to train on the synthetic set")
    full = SyntheticDepthSegDataset(n_synthetic, net_w)
    num_classes = 3
    total_use = int(round(len(full) * p["dataset_percentage"]))
    used, _ = torch.utils.data.random_split(full, [total_use, len(full) - total_use], generator=torch.Generator().manual_seed(0))
    assert len(used) > 0, "Dataset is empty"
    n_val = int(len(used) * p["val_percent"])
    n_val = max(n_val, 1)        # the reference asserts n_val > 0; a 12-item synthetic set at val_percent 0.005 would trip that
    n_train = len(used) - n_val
    assert n_train > 0, "Train count is 0"
    train_set, val_set = torch.utils.data.random_split(used, [n_train, n_val], generator=torch.Generator().manual_seed(0))

    arch = SOccDPT_versions[SOccDPT_version]
    model_kwargs = dict(num_classes=num_classes, model_type=model_type)
    if SOccDPT_version == 3:
        model_kwargs["load_depth"] = p["load_depth"]
        assert p["load_seg"] is False, "V3 does not support loading seg"
        model_kwargs["sigmoid"] = p["sigmoid"]
    calib = camera_intrinsics_yaml or write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
    model_kwargs["camera_intrinsics_yaml"] = calib
    if not forward_only:
        from ..lib import PREC_F32
        model_kwargs["precision"] = PREC_F32     # the training step (csrc/train_step.cpp) computes in exact f32, like the reference's amp=False runs
    net = load_model(arch=arch, model_kwargs=model_kwargs, device=torch.device("cpu"), model_path=p["load"] or None, model_type=model_type)
    net = net.to(device=device)
    # amp=True is the reference's fp16 autocast + GradScaler (:340,360-366,390-393): fp16 MFMA operands for the gradient GEMMs, the criterion's output
    # gradients scaled, parameter gradients unscaled and inf-checked before the step.  SOCCDPT_AMP=bf16 selects bf16 operands instead (no scaling needed).
    amp_mode = (os.environ.get("SOCCDPT_AMP", "f16") if p["amp"] else False)
    net.train_amp = amp_mode
    grad_scaler = GradScaler(enabled=(amp_mode in ("f16", "x3")))   # x3 operands are fp16 pairs: the same exponent range, so the same power-of-two loss scale (unscaled in f32)
    if world > 1:
        # equal shards: the gradient exchange averages with a flat 1 / world, which is the global-batch gradient only when every rank holds
        # batch_size / world frames
        assert p["batch_size"] % world == 0, f"data parallel: batch_size ({p['batch_size']}) must be a multiple of the number of ranks ({world})"
        sdist.attach_training(net)
    freeze_pretrained_encoder(net)
    unfreeze_pretrained_encoder_by_percentage(net, p["encoder_percentage"])
    n_train_t = sum(1 for q in net.parameters() if q.requires_grad)
    print(f"net {type(net).__name__}: {sum(q.numel() for q in net.parameters()) / 1e6:.1f} M parameters, {n_train_t} trainable tensors "
          f"(encoder_percentage {p['encoder_percentage']})")

    optimizer = Adam(net.parameters(), lr=p["learning_rate"], betas=(0.9, 0.999), eps=1e-08, weight_decay=p["weight_decay"], amsgrad=False)
    scheduler = ReduceLROnPlateau(optimizer, patience=2)
    batch_size = p["batch_size"]
    global_step, history, evals = 0, [], []
    for epoch in range(1, p["epochs"] + 1):
        net.train()
        epoch_loss = 0.0
        for batch_index in range(batch_size, len(train_set), batch_size):   # the reference's range (scripts/train_SOccDPT.py:350): a trailing full batch is NOT run
            if world > 1:      # this rank's contiguous shard of the batch
                lo, hi = sdist.shard_range(batch_size, rank, world)
                x, _, mask_disp, y_disp, mask_seg, y_seg = get_batch(train_set, batch_index - batch_size + hi, hi - lo)
            else:
                x, _, mask_disp, y_disp, mask_seg, y_seg = get_batch(train_set, batch_index, batch_size)
            x = x.to(device=device, dtype=torch.float32)
            y_disp, y_seg = y_disp.to(device=device, dtype=torch.float32), y_seg.to(device=device, dtype=torch.float32)
            mask_disp, mask_seg = mask_disp.to(device=device, dtype=torch.bool), mask_seg.to(device=device, dtype=torch.bool)
            for net_patch in PatchWiseInplace(net, p["patchwise_percentage"]):
                if forward_only:
                    net_patch.eval()
                    inv, seg = net_patch.network(x)
                    net_patch.train()
                else:
                    inv, seg = net_patch.train_forward(x)
                out = training_loss(inv, seg, y_disp, mask_disp, y_seg, mask_seg, loss_depth_w, loss_seg_w,
                                    compute_scale_and_shift=p["compute_scale_and_shift"])
                optimizer.zero_grad(set_to_none=True)
                if not forward_only:
                    net_patch.backward(*grad_scaler.scale(out["d_inv"], out["d_seg"]))
                    grad_scaler.step(optimizer, net_patch)
                    grad_scaler.update()
            # the GLOBAL-batch loss on every rank (ReduceLROnPlateau must see the same number everywhere, or the ranks cut the learning rate at different
            # steps and the replicas diverge silently: ADVICE r2): the device scalar is all-reduced in place and read back once -- one collective and
            # one host sync per step instead of a host round trip + a second blocking collective
            loss = sdist.all_reduce_mean_scalar(out["loss"]) if world > 1 else float(out["loss"].item())
            epoch_loss += loss
            history.append(loss)
            if rank == 0:
                print(f"epoch {epoch} step {global_step}: train_loss {loss:.6f} (disp {float(out['loss_disp']):.6f}, seg {float(out['loss_seg']):.6f}) lr {optimizer.lr:g}")
            division_step = max(n_train // (3 * batch_size), 1)
            if global_step % division_step == 0:
                # evaluation round (train_SOccDPT.py:406-430 -> utils/__init__.py:598-768): the 7 depth metrics and the IoU over the validation
                # split, computed on the GPU (csrc/metrics.hip); the wandb histograms / images of the reference are not produced
                if rank == 0:      # the replicas are identical: one rank builds the validation batches, evaluates and prints
                    val_batches = [val_set[i] for i in range(len(val_set))]      # items carry their batch dimension (datasets' layout)
                    abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3 = evaluate_depth(DepthNet(net), val_batches, device, amp=p["amp"])
                    print("abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3", abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3)
                    iou = evaluate_seg(SegNet(net), val_batches, device, amp=p["amp"])
                    print("iou", iou)
                    evals.append(dict(step=global_step, abs_rel=abs_rel, rmse=rmse, a1=a1, iou=iou))
                    net.train()
                if world > 1:
                    # the other ranks wait HERE, not inside the next step's gradient all-reduce, where a long validation pass would run into the
                    # collective watchdog (init_from_env gives the process group a generous timeout as well): ADVICE r3
                    sdist.barrier()
                scheduler.step(loss)   # `loss` is the all-reduced global-batch loss: every rank takes the same decision
            global_step += 1
            if max_steps and global_step >= max_steps:
                break
        if p["save_checkpoint"] and rank == 0:
            d = os.path.join(checkpoint_dir, run_id)
            Path(d).mkdir(parents=True, exist_ok=True)
            torch.save(net.state_dict(), os.path.join(d, "checkpoint_epoch_{}.pth".format(epoch)))
            print(f"Checkpoint {epoch} saved!")
        if max_steps and global_step >= max_steps:
            break
    return history


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(description="Train SOccDPT")
    parser.add_argument("-v", "--version", choices=[1, 2, 3], required=True, type=int, help="SOccDPT version")
    parser.add_argument("-n", "--count", default=1, type=int, help="Number of times to run the sweep")
    parser.add_argument("-dt", "--dataset", choices=["bdd", "idd", "idd+bdd"], required=True, help="Dataset to train using")
    parser.add_argument("-t", "--model_type", choices=model_types, required=True, help="Model architecture to use")
    parser.add_argument("-d", "--device", default="cpu", help="Device to use for training (the HIP path needs cuda:N)")
    parser.add_argument("-c", "--checkpoint_dir", default=os.path.join(os.getcwd(), "checkpoints"), help="Directory to save checkpoints in")
    parser.add_argument("-b", "--base_path", default=os.path.expanduser("~/Datasets/Depth_Dataset_Bengaluru"), help="Base path to dataset")
    parser.add_argument("--sweep_json", required=True, help="Path to checkpoint to sweep json")
    parser.add_argument("--forward_only", action="store_true", help="walk the schedule with the eval-mode forward + criterion only (no backward / optimizer)")
    parser.add_argument("--max_steps", default=0, type=int, help="stop every run after this many batches (0 = all)")
    return parser


def main(args):
    method, params = read_sweep(args.sweep_json)
    os.makedirs(args.checkpoint_dir, exist_ok=True)
    project_name = "SOccDPT_V{version}_{model_type}_{dataset}".format(version=str(args.version), model_type=args.model_type, dataset=args.dataset)
    print("project", project_name, "sweep method", method)
    out = []
    for i, run in enumerate(sample_runs(method, params, args.count)):
        print(f"run {i}: {run}")
        out.append(train_net(SOccDPT_version=args.version, device=args.device, model_type=args.model_type, checkpoint_dir=args.checkpoint_dir,
                             dataset=args.dataset, base_path=args.base_path, forward_only=args.forward_only, max_steps=args.max_steps,
                             run_id=f"local_run_{i}", **run))
    return out


if __name__ == "__main__":
    main(build_parser().parse_args())
