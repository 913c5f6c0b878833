"""The training entry point's config / CLI surface (`config/*.json` sweep files, the reference's flags, freeze / unfreeze order)
on CPU, and on the GPU a forward-only walk of the schedule of BASELINE configs[4] (patch-wise step shape: B = 3, patchwise 0.5)."""
import json
import os

import pytest
import torch

SWEEP = {   # the keys / values of /root/reference/config/SOccDPT_V3_dpt_swin2_tiny_256_Aug_22.json (a data file; inlined because the
            # reference tree does not exist on the GPU box)
    "method": "random", "metric": {"goal": "minimize", "name": "train_loss"},
    "parameters": {"amp": {"values": [False]}, "epochs": {"values": [15]}, "batch_size": {"values": [3]}, "val_percent": {"values": [0.005]},
                   "weight_decay": {"values": [0]}, "learning_rate": {"values": [0.00001]}, "save_checkpoint": {"values": [True]},
                   "encoder_percentage": {"values": [0.5]}, "patchwise_percentage": {"values": [1.0]}, "dataset_percentage": {"values": [1.0]},
                   "loss_weights": {"values": [[0.75, 0.25], [0.25, 0.75], [0.5, 0.5]]}, "load": {"values": [False]},
                   "load_depth": {"values": [False]}, "load_seg": {"values": [False]}, "compute_scale_and_shift": {"values": [True]},
                   "sigmoid": {"values": [False]}}}


def _sweep_file(tmp_path, **over):
    cfg = json.loads(json.dumps(SWEEP))
    for k, v in over.items():
        cfg["parameters"][k] = {"values": [v]}
    p = tmp_path / "sweep.json"
    p.write_text(json.dumps(cfg))
    return str(p)


def test_sweep_json_and_cli_surface(tmp_path):
    from soccdpt_amd.scripts.train_SOccDPT import SWEEP_DEFAULTS, build_parser, read_sweep, sample_runs
    method, params = read_sweep(_sweep_file(tmp_path))
    assert method == "random" and set(params) == set(SWEEP_DEFAULTS)          # every key of the reference's sweep files is consumed
    runs = sample_runs(method, params, 4)
    assert len(runs) == 4 and all(r["batch_size"] == 3 and r["encoder_percentage"] == 0.5 for r in runs)
    assert runs == sample_runs(method, params, 4)                              # seeded
    assert {tuple(r["loss_weights"]) for r in sample_runs(method, params, 40)} == {(0.75, 0.25), (0.25, 0.75), (0.5, 0.5)}
    args = build_parser().parse_args(["-v", "3", "-dt", "bdd", "-t", "dpt_swin2_tiny_256", "--sweep_json", "x.json"])
    assert args.device == "cpu" and args.count == 1 and args.version == 3     # the reference's defaults (train_SOccDPT.py:485-546)
    with pytest.raises(SystemExit):
        build_parser().parse_args(["-v", "4", "-dt", "bdd", "-t", "dpt_swin2_tiny_256", "--sweep_json", "x.json"])
    if os.path.isdir("/root/reference/config"):     # build container only: every sweep file of the reference parses
        for f in sorted(os.listdir("/root/reference/config")):
            if f.endswith(".json"):
                m, p = read_sweep(os.path.join("/root/reference/config", f))
                assert p and all(isinstance(v, list) and v for v in p.values()), f


def test_freeze_unfreeze_is_index_based_like_the_reference(tmp_path):
    """unfreeze_pretrained_encoder_by_percentage(net, 0.5): the FIRST half of net.pretrained.parameters() (patch embedding first) trains,
    the second half is frozen (/root/reference/SOccDPT/loss/__init__.py:20-31); decoder / head parameters are untouched."""
    import contextlib
    import io
    from soccdpt_amd.loss import freeze_pretrained_encoder, unfreeze_pretrained_encoder_by_percentage
    from soccdpt_amd.model.SOccDPT import SOccDPT_V3
    from soccdpt_amd.utils.optim import PatchWiseInplace
    from soccdpt_amd.utils.synth import write_synth_calib
    calib = write_synth_calib(str(tmp_path / "calib.yaml"))
    with contextlib.redirect_stdout(io.StringIO()):
        net = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib)
    freeze_pretrained_encoder(net)
    assert not any(p.requires_grad for p in net.pretrained.parameters())
    unfreeze_pretrained_encoder_by_percentage(net, 0.5)
    enc = list(net.pretrained.parameters())
    m = round(len(enc) * 0.5)
    assert all(p.requires_grad for p in enc[:m]) and not any(p.requires_grad for p in enc[m:])
    names = [n for n, p in net.named_parameters() if p.requires_grad]
    assert names[0] == "depth_net.pretrained.model.patch_embed.proj.weight"
    assert any(n.startswith("depth_net.scratch.") for n in names) and any(n.startswith("seg_head.") for n in names)
    n_tr = len(names)
    seen = []
    for patch in PatchWiseInplace(net, 0.5):                       # two patches, disjoint, covering all trainable tensors
        seen.append([n for n, p in patch.named_parameters() if p.requires_grad])
    assert len(seen) == 2 and len(seen[0]) + len(seen[1]) == n_tr and not set(seen[0]) & set(seen[1])
    assert [n for n, p in net.named_parameters() if p.requires_grad] == names      # flags restored


@pytest.mark.gpu
def test_forward_only_schedule_walk_gpu(tmp_path, gpu_device):
    """BASELINE configs[4] shape: B = 3, encoder_percentage 0.5, patchwise_percentage 0.5 -> 2 patches per batch; forward + criterion (+ its
    output gradients) run on the GPU for every patch and one checkpoint per epoch is written; without --forward_only the same schedule
    trains (train-mode forward, backward, fused Adam) and moves only parameters that are unfrozen."""
    from soccdpt_amd.scripts.train_SOccDPT import build_parser, main
    sweep = _sweep_file(tmp_path, epochs=2, patchwise_percentage=0.5, val_percent=0.1)
    ck = tmp_path / "ck"
    argv = ["-v", "3", "-dt", "bdd", "-t", "dpt_swin2_tiny_256", "-d", "cuda:0", "-c", str(ck), "-b", "/nonexistent", "--sweep_json", sweep, "--max_steps", "2"]
    hist = main(build_parser().parse_args(argv + ["--forward_only"]))
    assert len(hist) == 1 and len(hist[0]) == 2 and all(h > 0 and h == h for h in hist[0])
    assert (ck / "local_run_0" / "checkpoint_epoch_1.pth").exists()
    sd = torch.load(ck / "local_run_0" / "checkpoint_epoch_1.pth", map_location="cpu")
    assert "depth_net.scratch.refinenet1.out_conv.weight" in sd and "seg_head.4.bias" in sd
    ck2 = tmp_path / "ck_train"
    argv[argv.index(str(ck))] = str(ck2)
    hist = main(build_parser().parse_args(argv))
    assert len(hist) == 1 and len(hist[0]) == 2 and all(h > 0 and h == h for h in hist[0])
    sd2 = torch.load(ck2 / "local_run_0" / "checkpoint_epoch_1.pth", map_location="cpu")
    moved = [k for k in sd if sd[k].is_floating_point() and not torch.equal(sd[k], sd2[k])]
    assert "depth_net.scratch.refinenet1.out_conv.weight" in moved and "seg_head.4.bias" in moved and "seg_head.1.running_mean" in moved
    # encoder_percentage 0.5 unfreezes the FIRST half of the encoder's parameter list (model/loss.py:124-152): the last stage stays frozen
    assert "depth_net.pretrained.model.patch_embed.proj.weight" in moved
    assert not any(k.startswith("depth_net.pretrained.model.layers.3.") for k in moved)


@pytest.mark.gpu
def test_training_reduces_the_loss_gpu(gpu_device):
    """Eight optimisation steps on one fixed synthetic batch (train-mode forward -> SSI + BCE criterion -> backward -> fused Adam): the
    loss falls.  The end-to-end check that the gradients point downhill through the whole stack."""
    import os
    import tempfile
    from soccdpt_amd.lib import PREC_F32
    from soccdpt_amd.model.SOccDPT import SOccDPT_V3
    from soccdpt_amd.scripts.train_SOccDPT import SyntheticDepthSegDataset, get_batch
    from soccdpt_amd.utils.loss import training_loss
    from soccdpt_amd.utils.optim import Adam
    from soccdpt_amd.utils.synth import synth_state_dict, write_synth_calib
    calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
    net = SOccDPT_V3(sigmoid=True, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=False, precision=PREC_F32)
    net.load_state_dict(synth_state_dict(alias_pretrained=True), strict=False)
    net = net.to(gpu_device).train()
    ds = SyntheticDepthSegDataset(2, 256)
    x, _, mask_disp, y_disp, mask_seg, y_seg = get_batch(ds, 2, 2)
    x = x.to(gpu_device, torch.float32)
    y_disp, y_seg = y_disp.to(gpu_device, torch.float32), y_seg.to(gpu_device, torch.float32)
    mask_disp, mask_seg = mask_disp.to(gpu_device, torch.bool), mask_seg.to(gpu_device, torch.bool)
    opt = Adam(net.parameters(), lr=1e-4)
    losses = []
    for step in range(8):
        inv, seg = net.train_forward(x, seed=step)
        out = training_loss(inv, seg, y_disp, mask_disp, y_seg, mask_seg, 0.5, 0.5, compute_scale_and_shift=True)
        opt.zero_grad(set_to_none=True)
        net.backward(out["d_inv"], out["d_seg"])
        opt.step()
        losses.append(float(out["loss"]))
    print("losses:", [f"{v:.4f}" for v in losses])
    assert all(v == v for v in losses)
    assert losses[-1] < 0.9 * losses[0], losses
