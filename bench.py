#!/usr/bin/env python3
"""bench.py — frames/s of the full SOccDPT_V3 forward (depth + seg + points + occupancy) on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches it
under torch.distributed.run, one rank per GPU (RCCL).  One "step" = one forward of one batch of
synthetic 256x256 frames per rank (weak scaling: per-GPU batch fixed).  Rank 0 prints ONE JSON line.

Workload at N = 1: BASELINE.json configs[1] — SOccDPT_V3 dpt_swin2_tiny_256, batch 8, compute_occ=True,
eval mode, synthetic weights/inputs/camera.  `value` is timed in the default arithmetic
`--precision mixed` (SOCCDPT_PREC_MIXED: fp16 MFMA operands, x3 split where the shipped precision map asks
for it), the fastest mode whose outputs are measured -- in this same run, against the library's own
exact-f32 mode -- inside HALF the north star's 1e-3 (`tolerance`); the bf16 figure BASELINE configs[1]
names and the plain-fp16 figure are side fields (`bf16_operands`, `f16_operands`), each with its own
measured errors and roofline fraction.

Extra objects:
  roofline     dominant kernel family of the forward (by summed device time), timed live with HIP
               events on the launch stream (soccdpt_profile_*): achieved = algorithmic FLOPs / time.
  cpu_baseline the CPU oracle (oracle/soccdpt_ref.py, fp32 PyTorch-CPU restatement, kind "port")
               timed on this host's cores on a bounded sample (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
PEAK_F32_TFLOPS = 157.3     # f32-input MFMA (same table)
PEAK_HBM_GBS = 8000.0       # HBM3E spec (same table)


FWD_GFLOP_PER_FRAME = {"dpt_swin2_tiny_256": 78.82, "dpt_swin2_base_384": 259.6, "dpt_hybrid_384": 293.0}   # SURVEY.md 8d (algorithmic, forward)


def train_step_bench(args):
    """One JSON line for the training step (BASELINE configs[4]).  A step = one PatchWiseInplace patch: train-mode forward, SSI + BCE criterion
    at 1080 x 1920 with its output gradients, network backward, fused Adam; inputs and targets resident in HBM.  value = samples/s.
    roofline: the step is MFMA-bound in f32 (three GEMM passes: forward, dgrad, wgrad); achieved = 3 x the forward's algorithmic FLOPs per
    sample x samples/s against the 157.3 TFLOP/s f32 MFMA peak -- only quoted when everything is trainable (frozen tensors skip their wgrad)."""
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    from soccdpt_amd.lib import PREC_F32
    from soccdpt_amd.loss import freeze_pretrained_encoder, unfreeze_pretrained_encoder_by_percentage
    from soccdpt_amd.model.SOccDPT import SOccDPT_V3
    from soccdpt_amd.model.spec import MODEL_TYPE_TO_BACKBONE, backbone_image_size
    from soccdpt_amd.scripts.train_SOccDPT import SyntheticDepthSegDataset, get_batch
    from soccdpt_amd.utils.loss import training_loss
    from soccdpt_amd.utils.optim import Adam, PatchWiseInplace
    from soccdpt_amd.utils.synth import synth_state_dict, write_synth_calib
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    backbone = MODEL_TYPE_TO_BACKBONE[args.model_type]
    S = backbone_image_size(backbone)
    B = args.batch
    calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
    net = SOccDPT_V3(sigmoid=True, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=False, precision=PREC_F32, model_type=args.model_type)
    sd = synth_state_dict(backbone, alias_pretrained=True)
    net.load_state_dict(sd, strict=False)
    net = net.to(dev).train()
    net.train_amp = args.amp or False
    freeze_pretrained_encoder(net)
    unfreeze_pretrained_encoder_by_percentage(net, args.encoder_percentage)
    x, _, mask_disp, y_disp, mask_seg, y_seg = get_batch(SyntheticDepthSegDataset(B, S), B, B)
    x = x.to(dev, torch.float32)
    y_disp, y_seg = y_disp.to(dev, torch.float32), y_seg.to(dev, torch.float32)
    mask_disp, mask_seg = mask_disp.to(dev, torch.bool), mask_seg.to(dev, torch.bool)
    opt = Adam(net.parameters(), lr=1e-5)

    def one_batch():
        n = 0
        for net_patch in PatchWiseInplace(net, args.patchwise_percentage):
            inv, seg = net_patch.train_forward(x, seed=n)
            out = training_loss(inv, seg, y_disp, mask_disp, y_seg, mask_seg, 0.5, 0.5, compute_scale_and_shift=True)
            opt.zero_grad(set_to_none=True)
            net_patch.backward(out["d_inv"], out["d_seg"])
            opt.step()
            n += 1
        return n

    from soccdpt_amd.lib import load_library
    steps = max(1, min(args.steps, 50))
    for _ in range(max(1, min(args.warmup, 5))):
        one_batch()
    torch.cuda.synchronize()
    l0 = int(load_library().soccdpt_launch_counter())
    t0 = time.perf_counter()
    n = 0
    for _ in range(steps):
        n += one_batch()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    launches = (int(load_library().soccdpt_launch_counter()) - l0) / n   # kernels of libsoccdpt_hip.so per step (forward + criterion + backward + Adam)
    sps = B * n / dt
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record(); inv, seg = net.train_forward(x, seed=0); ev[1].record()
    out = training_loss(inv, seg, y_disp, mask_disp, y_seg, mask_seg, 0.5, 0.5, compute_scale_and_shift=True); ev[2].record()
    net.backward(out["d_inv"], out["d_seg"]); ev[3].record()
    torch.cuda.synchronize()
    all_trainable = args.encoder_percentage >= 1.0 and args.patchwise_percentage >= 1.0
    roof = None
    if all_trainable:
        # f32: the exact-f32 MFMA peak; x3: every GEMM of the step (forward, dgrad, wgrad) runs three fp16 MFMAs per product -> a third of the fp16 peak;
        # bf16 / f16 amp: the train forward runs x3 operands (its outputs feed the f32 tape), dgrad and wgrad run 16-bit operands -> one third of the
        # algorithmic FLOPs against 833.3, two thirds against 2500: the FLOP-weighted harmonic mean 3 / (1 / 833.3 + 2 / 2500) = 1500
        x3_peak = PEAK_BF16_TFLOPS / 3.0
        peak = PEAK_F32_TFLOPS if not args.amp else (round(x3_peak, 1) if args.amp == "x3" else round(3.0 / (1.0 / x3_peak + 2.0 / PEAK_BF16_TFLOPS), 1))
        ach = 3.0 * FWD_GFLOP_PER_FRAME[args.model_type] * sps / 1e3
        roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
                "kernel": "whole step (igemm forward + dgrad + split-K wgrad; profiles/r04_train_*kernel_stats.csv split it per kernel)",
                "peak_note": ("exact-f32 MFMA" if not args.amp else "x3: a third of the 16-bit MFMA peak" if args.amp == "x3" else
                              "x3 forward (1/3 of the FLOPs at 833.3) + 16-bit dgrad / wgrad (2/3 at 2500): FLOP-weighted harmonic mean"),
                "flops_per_sample_gflop": round(3.0 * FWD_GFLOP_PER_FRAME[args.model_type], 1)}
    else:
        # partly frozen schedule (encoder_percentage / patchwise_percentage < 1): frozen tensors skip their wgrad GEMM and the dgrad stops where nothing
        # upstream trains, so the executed FLOPs vary per patch.  What always runs in full is the train-mode forward: a LOWER BOUND of the achieved rate.
        peak = PEAK_F32_TFLOPS if not args.amp else round(PEAK_BF16_TFLOPS / 3.0, 1)
        ach = FWD_GFLOP_PER_FRAME[args.model_type] * sps / 1e3
        roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None, "lower_bound": True,
                "kernel": "whole step; only the train-mode forward's FLOPs are counted (the backward's share depends on the frozen set of each patch)",
                "flops_per_sample_gflop": round(FWD_GFLOP_PER_FRAME[args.model_type], 1)}
    cpu = None
    if not args.no_cpu_baseline and not args.headline_only:
        from oracle import soccdpt_ref as R
        sd_o = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v.clone()) for k, v in sd.items()}
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        xc = x[:2].cpu()
        t0 = time.perf_counter()
        o_inv, o_seg, _ = R.soccdpt_v3_network(sd_o, xc, backbone=backbone, sigmoid=True, training=True, dropout_p=0.1)
        (o_inv.sum() + o_seg.sum()).backward()
        cpu = {"value": round(xc.shape[0] / (time.perf_counter() - t0), 3), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"oracle forward + torch autograd backward on {xc.shape[0]} samples (no criterion / optimizer)"}
    result = {"metric": f"samples/sec SOccDPT_V3 {args.model_type.replace('dpt_', '')} training step (train forward + criterion + backward + Adam)",
              "value": round(sps, 2), "unit": "samples/s", "n_gpus": 1, "steps": n, "warmup": max(1, min(args.warmup, 5)), "ms_per_step": round(1e3 * dt / n, 3),
              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": ("f16x3 (split-fp16 operands, f32 accumulate) in every GEMM, f32 tape" if args.amp == "x3" else f"f32 forward, {args.amp}-operand gradient GEMMs" if args.amp else "f32"), "data": "synthetic",
              "config": {"workload": f"SOccDPT_V3 {args.model_type} patch-wise training step, synthetic 1080x1920 targets", "batch_per_gpu": B, "image": S,
                         "encoder_percentage": args.encoder_percentage, "patchwise_percentage": args.patchwise_percentage,
                         "trainable_tensors": sum(1 for q in net.parameters() if q.requires_grad),
                         "drop_path_rate": float(getattr(net, "drop_path_rate", 0.0)), "seg_head_dropout_p": float(net.seg_head[3].p)},
              "split_ms": {"train_forward": round(ev[0].elapsed_time(ev[1]), 2), "criterion": round(ev[1].elapsed_time(ev[2]), 2),
                           "backward_all_unfrozen": round(ev[2].elapsed_time(ev[3]), 2)},
              "launches_per_step": round(launches, 1),
              "roofline": roof, "cpu_baseline": cpu, "loss": round(float(out["loss"]), 6),
              "train_workspace_gib": round(net._engine(dev).train_workspace(B).numel() / 2 ** 30, 2)}
    os.dup2(real_stdout, 1)
    print(json.dumps(result), flush=True)
    return 0


PREC_CODE = {"bf16": 0, "f32": 1, "f16": 2, "f16x3": 3, "mixed": 4}
QUANTITIES = ("feat0", "feat1", "feat2", "feat3", "path1", "inv", "seg_logits")


def igemm_roofline(stats, prof_steps, precision, pmc=None):
    """The igemm_kernel template as a whole: algorithmic FLOPs / summed launch durations against the MFMA peak.  x3 launches spend three
    fp16 MFMAs per algorithmic product, so their peak for ALGORITHMIC FLOPs is a third of the dense 16-bit peak; a mixed-precision forward is
    priced against the FLOP-weighted harmonic mean of its members' peaks (what the same launches would take at their own peaks)."""
    mem = {n: v for n, v in stats.items() if n.startswith("igemm_") and v["flops"] > 0}
    if not mem:
        return None
    def peak_of(name):
        if name.startswith("igemm_f32"):
            return PEAK_F32_TFLOPS
        return PEAK_BF16_TFLOPS / 3.0 if name.startswith("igemm_x3") else PEAK_BF16_TFLOPS
    flops, ms, launches = sum(v["flops"] for v in mem.values()), sum(v["ms"] for v in mem.values()), sum(v["launches"] for v in mem.values())
    peak = flops / sum(v["flops"] / peak_of(n) for n, v in mem.items())
    ach = flops / (ms * 1e-3) / 1e12
    total_ms = sum(v["ms"] for v in stats.values())
    # `peak` / `frac`: the guide's dense 16-bit MFMA peak (2500 TFLOP/s) for every 16-bit arithmetic, whatever operand splitting the launches chose -- three
    # MFMAs per product (x3) or two (x2w) are an implementation cost, not algorithmic work; the exact-f32 mode is priced
    # against the f32 MFMA peak.  The FLOP-weighted harmonic mean of the members' own peaks (what rounds 3-5 printed as `frac`) moves to
    # blended_peak / frac_vs_blended_peak.
    top = PEAK_F32_TFLOPS if precision == "f32" else PEAK_BF16_TFLOPS
    roof = dict(bound="mfma", kernel="igemm_kernel (all tile configurations)", achieved=round(ach, 2), peak=top, unit="TFLOP/s", frac=round(ach / top, 4),
                traffic=None, avg_launch_us=round(ms * 1e3 / launches, 2), share_of_device_time=round(ms / total_ms, 4),
                flops_per_step=flops / prof_steps, launches_per_step=launches / prof_steps)
    roof["frac_of_16bit_peak"] = round(ach / PEAK_BF16_TFLOPS, 4) if not precision == "f32" else None   # = frac for the 16-bit modes (kept: rounds 4-5 consumers read it)
    roof["blended_peak"] = round(peak, 1)
    roof["frac_vs_blended_peak"] = round(ach / peak, 4)
    if precision == "mixed":
        roof["peak_note"] = "frac = achieved / 2500 (dense 16-bit MFMA); blended_peak = FLOP-weighted harmonic mean of 2500 (fp16 launches) and 833.3 (x3 launches: three MFMAs per algorithmic product)"
    by = []
    for name in sorted(mem, key=lambda n: -mem[n]["ms"]):
        v = mem[name]
        e = dict(config=name, launches_per_step=v["launches"] / prof_steps, avg_launch_us=round(v["ms"] * 1e3 / v["launches"], 2),
                 achieved=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                 frac=round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / (PEAK_F32_TFLOPS if name.startswith("igemm_f32") else PEAK_BF16_TFLOPS), 4),
                 frac_vs_format_peak=round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / peak_of(name), 4),
                 frac_of_16bit_peak=(None if name.startswith("igemm_f32") else round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4)))
        if pmc and name in pmc:
            e["traffic"] = round(pmc[name]["hbm_bytes_per_launch"])
            if "mfma_util" in pmc[name]:
                e["mfma_util_pmc"] = round(pmc[name]["mfma_util"], 4)
        by.append(e)
    if len(by) > 1:
        roof["by_config"] = by
    if pmc and all(n in pmc for n in mem):
        roof["traffic"] = round(sum(pmc[n]["hbm_bytes_per_launch"] * mem[n]["launches"] for n in mem) / launches)   # HBM bytes per launch, launch-weighted
    return roof


def launch_ranks(args):
    """`python bench.py --gpus N` started plainly (no torchrun around it): this parent -- which never initialises the GPU -- starts
    `python -m torch.distributed.run --nproc-per-node N bench.py <same flags>` as a CHILD process (one rank per GPU over RCCL), relays rank 0's
    single JSON line and exits non-zero when the ranks did not all connect (`rccl_ranks != N`) or the child failed."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    line = None
    for ln in proc.stdout.decode(errors="replace").splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if line is None:
        print(f"[bench] the {args.gpus}-rank child produced no result line (exit code {proc.returncode})", file=sys.stderr)
        return proc.returncode or 1
    print(line, flush=True)
    try:
        ranks = json.loads(line).get("rccl_ranks")
    except ValueError:
        ranks = None
    if proc.returncode != 0 or ranks != args.gpus:
        print(f"[bench] expected {args.gpus} connected ranks, the result line says {ranks} (child exit code {proc.returncode})", file=sys.stderr)
        return proc.returncode or 3
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=None, help="frames per GPU per step (default: 8; 4 for dpt_hybrid_384 under --config 2)")
    ap.add_argument("--config", type=int, default=None, choices=[1, 2, 3, 4],
                    help="BASELINE.json configs[k] preset: 1 = dpt_swin2_tiny_256 bf16 batch 8 (the default workload); 2 = dpt_hybrid_384 batch 4; "
                         "3 = dpt_swin2_base_384, 8 frames per GPU (64 frames over --gpus 8); 4 = the patch-wise training step (tiny_256)")
    ap.add_argument("--streams", type=int, default=1, help="concurrent sub-batches inside one forward (soccdpt_set_streams)")
    ap.add_argument("--model-type", default="dpt_swin2_tiny_256", choices=["dpt_swin2_tiny_256", "dpt_swin2_base_384", "dpt_hybrid_384"],
                    help="dpt_swin2_tiny_256 = BASELINE metric config (configs[1]); dpt_swin2_base_384 --batch 8 = configs[3]'s per-GPU shape "
                         "(64 frames over 8 GPUs); dpt_hybrid_384 --batch 4 = configs[2]")
    ap.add_argument("--precision", choices=["mixed", "bf16", "f16", "f32", "f16x3"], default="mixed",
                    help="mixed (default): fp16 MFMA operands, x3 split where the shipped precision map asks for it -- the fastest mode inside half the "
                         "north star's 1e-3; bf16: bf16 operands (what BASELINE configs[1] names; 3e-3, fails the tolerance); f16: IEEE fp16 operands, "
                         "same kernels and MFMA rate, meets 1e-3 on the Swin-V2 models without margin; f32: exact-f32 parity mode (1/16 MFMA rate); "
                         "f16x3: split-operand fp16 everywhere (three fp16 MFMAs per product, ~22 significand bits at 1/3 of the 16-bit rate)")
    ap.add_argument("--calibrate", action="store_true",
                    help="--precision mixed: derive the precision map on the bound weights with soccdpt_prec_calibrate (sample = the first two frames of the "
                         "benchmark batch, budget = the bar of `tolerance`) before timing; config.precision_map_source says which map `value` ran")
    ap.add_argument("--weights", default="salt0", choices=["salt0", "salt1", "salt2", "trained_like"],
                    help="which synthetic checkpoint is bound (soccdpt_amd/utils/synth.py named_weights): salt0 = the draw the shipped precision maps were derived on "
                         "(default); any other set is what a user's own checkpoint looks like to the library: --precision mixed then calibrates on six other frames "
                         "(4 select + 2 verify) before timing unless --no-calibrate is given (all-x3 operands then)")
    ap.add_argument("--no-calibrate", action="store_true", help="--weights <other>: time the uncalibrated default (every group x3) instead of calibrating")
    ap.add_argument("--no-other-weights", action="store_true", help="skip the `other_weights` side object (salt1 and trained_like: all-x3 and calibrated frames/s)")
    ap.add_argument("--no-side-modes", action="store_true", help="skip the bf16 / fp16 side legs and the live error measurement")
    ap.add_argument("--graph", action="store_true", help="replay the network as a captured hipGraph (measured: slower than eager)")
    ap.add_argument("--prewarm", type=int, default=100, help="untimed clock/allocator pre-warm forwards before the W warm-up steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the timed headline leg + its per-kernel event pass: no fp16 / two-stream / CPU-baseline legs "
                         "(what tools/collect_profiles.sh runs under rocprofv3, so profiles/*_kernel_stats.csv describe the headline launches alone)")
    ap.add_argument("--with-two-streams", action="store_true",
                    help="also time the EXPERIMENTAL two-concurrent-sub-batches mode (soccdpt_set_streams(2); opt-in, DESIGN.md section 4)")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="side object `pipelined`: this many whole forwards in flight -- independent engines / workspaces, step i + 1 issued on another stream while "
                         "step i drains, no synchronisation in between (the reference's timing loop never synchronises between forwards either: "
                         "scripts/eval_SOccDPT.py:246-259); 1 = off.  `value` and the per-kernel rooflines stay single-stream")
    ap.add_argument("--cpu-sample-frames", type=int, default=2)
    ap.add_argument("--train-step", action="store_true",
                    help="BASELINE configs[4] instead of the forward: one optimisation step (train-mode forward + criterion + backward + fused Adam, exact f32) "
                         "per PatchWiseInplace patch; N = 1")
    ap.add_argument("--amp", nargs="?", const="bf16", default=None, choices=["bf16", "f16", "x3"],
                    help="--train-step: operand format of the gradient GEMMs (the reference's amp sweep parameter): bf16 (default when given), f16, or "
                         "x3 = split-operand fp16 (three fp16 MFMAs per product: f32-grade gradients, no loss scaling)")
    ap.add_argument("--encoder-percentage", type=float, default=1.0, help="--train-step: unfreeze_pretrained_encoder_by_percentage")
    ap.add_argument("--patchwise-percentage", type=float, default=1.0, help="--train-step: PatchWiseInplace")
    args = ap.parse_args()
    if args.config == 2:
        args.model_type = "dpt_hybrid_384"
        args.batch = args.batch or 4
    elif args.config == 3:
        args.model_type = "dpt_swin2_base_384"
    elif args.config == 4:
        args.train_step = True
    args.batch = args.batch or 8
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.train_step:
        return launch_ranks(args)   # before anything touches the GPU
    if args.train_step:
        return train_step_bench(args)
    if args.headline_only:
        args.no_cpu_baseline = True
    if args.with_two_streams or args.streams > 1:
        os.environ["SOCCDPT_ALLOW_MULTISTREAM"] = "1"

    # stdout must carry exactly ONE JSON line: RCCL prints a version banner to stdout when a communicator is created, and
    # libraries may print too.  Keep the real stdout aside and point fd 1 at stderr for everything else.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from soccdpt_amd import dist as sdist
    from soccdpt_amd.model.SOccDPT import SOccDPT_V3
    from soccdpt_amd.utils.synth import named_weights, synth_input, synth_state_dict, write_synth_calib

    rank, local, world = sdist.init_from_env("nccl")
    if world != max(args.gpus, 1):
        if rank == 0:
            print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    import contextlib
    import io
    calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
    with contextlib.redirect_stdout(io.StringIO()):
        net = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True, streams=args.streams,
                         model_type=args.model_type,
                         graph=args.graph, precision=PREC_CODE[args.precision])
    from soccdpt_amd.model.spec import MODEL_TYPE_TO_BACKBONE, backbone_image_size
    backbone = MODEL_TYPE_TO_BACKBONE[args.model_type]
    img = backbone_image_size(backbone)
    sd = named_weights(args.weights, backbone)
    net.load_state_dict(sd, strict=False)
    net = net.eval().to(dev)
    sdist.attach(net)

    B = args.batch
    x = synth_input(B, size=img, seed0=rank * B).to(dev)   # different frames per rank
    budget = 1e-3 if args.model_type == "dpt_hybrid_384" else 5e-4
    n_cal = 6 if img <= 256 else 3                       # calibration sample: 4 + 2 frames (2 + 1 at 384 px), none of them in the benchmark batch
    x_cal = synth_input(n_cal, size=img, seed0=5000 + rank * n_cal).to(dev)
    calib_report = None
    if args.precision == "mixed" and (args.calibrate or (args.weights != "salt0" and not args.no_calibrate)):
        with contextlib.redirect_stdout(io.StringIO()):
            calib_report = net.calibrate_precision(x_cal, budget=budget)

    def barrier():
        if dist.is_initialized():
            if dist.get_backend() == "nccl":
                dist.barrier(device_ids=[local])
            else:
                dist.barrier()      # SOCCDPT_DIST_REHEARSAL=1 (gloo; see soccdpt_amd/dist.py)
        torch.cuda.synchronize()

    # clock / allocator pre-warm (untimed, in addition to the W warm-up steps): a GPU that has just been handed over from another
    # process can sit in a low power state for the first ~100 ms (observed once: 4.4 instead of 2.2 ms per step)
    for _ in range(args.prewarm):   # a FIXED count: with N > 1 every forward holds a collective, so all ranks must run the same number
        out = net(x)
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        out = net(x)
    # R repeats of EXACTLY K timed steps, each bracketed by barrier + synchronize on both sides; R is chosen so that at least ~50 steps are
    # timed in all (20 steps of this forward are 37 ms: one repeat alone moves +-2 % with the clock state the previous process left).
    # `value` comes from the MEDIAN repeat; every repeat, the minimum and the median are printed.
    repeats = 1 if args.steps >= 50 else min(5, -(-50 // max(args.steps, 1)))
    rep_elapsed = []
    for _ in range(repeats):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = net(x)
        barrier()
        rep_elapsed.append(time.perf_counter() - t0)
    elapsed = sorted(rep_elapsed)[(len(rep_elapsed) - 1) // 2]
    per_rank_ms = [round(elapsed / args.steps * 1e3, 3)]
    rccl_ranks = 1
    if world > 1:
        cdev = dev if dist.get_backend() == "nccl" else torch.device("cpu")
        t = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
        allt = torch.empty((world,), device=cdev, dtype=torch.float64)
        dist.all_gather_into_tensor(allt, t)                # evidence of how many ranks RCCL really connected
        per_rank_ms = [round(float(v) / args.steps * 1e3, 3) for v in allt.cpu()]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        rccl_ranks = dist.get_world_size()
    frames = world * B * args.steps
    fps = frames / elapsed

    # ---- per-kernel timing (separate, equally sized region): every launch carries a HIP start / stop event pair bound to its own dispatch
    # (hipExtLaunchKernelGGL, csrc/launch.h), so a kernel's time is its begin -> end on the device -- what rocprofv3 --kernel-trace reports --
    # and device_ms_per_step, their sum, excludes the gaps between dependent launches (it is <= ms_per_step) ----
    eng = net._engine(dev)
    prof_steps = max(3, min(args.steps, 10))
    eng.profile_enable(True)
    barrier()
    t1 = time.perf_counter()
    for _ in range(prof_steps):
        out = net(x)
    torch.cuda.synchronize()
    elapsed_prof = time.perf_counter() - t1
    stats = eng.profile_collect()
    eng.profile_enable(False)

    def load_pmc():
        """HBM bytes / MFMA-pipe utilisation per launch: PMC counters cannot be read inside this process; they come from the committed rocprofv3
        --pmc passes over this same command (profiles/, tools/collect_profiles.sh, tools/pmc_summary.py).  A file is only used when it was
        collected from the CURRENT kernel sources (csrc_sha); otherwise traffic is null and traffic_note says why."""
        try:
            import glob
            from soccdpt_amd.lib import csrc_sha
            tagsfx = {"dpt_swin2_tiny_256": "", "dpt_swin2_base_384": "_base384", "dpt_hybrid_384": "_hybrid384"}[args.model_type]
            cands = sorted(glob.glob(os.path.join(REPO, "profiles", f"r*{tagsfx}_pmc_traffic.json")))
            cands = [c for c in cands if tagsfx or not any(t in os.path.basename(c) for t in ("_base384", "_hybrid384"))]
            pj = json.load(open(cands[-1]))
            if pj.get("csrc_sha") == csrc_sha() and pj.get("precision", "bf16") == args.precision:
                return pj["kernels"], os.path.basename(cands[-1]), None
            return {}, None, (f"{os.path.basename(cands[-1])} was collected from other kernel sources or another precision (csrc_sha {pj.get('csrc_sha')} vs "
                              f"{csrc_sha()}, precision {pj.get('precision', 'bf16')} vs {args.precision})")
        except Exception:
            return {}, None, None

    def kernel_table(stats, steps):
        total = sum(v["ms"] for v in stats.values())
        rows = []
        for name, v in sorted(stats.items(), key=lambda kv: -kv[1]["ms"]):
            k = dict(name=name, launches_per_step=v["launches"] / steps, ms_per_step=round(v["ms"] / steps, 4), share=round(v["ms"] / total, 4))
            if v["flops"] > 0:
                k["tflops"] = round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)
            if v["bytes"] > 0:
                k["gbs"] = round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)
            rows.append(k)
        return rows, total

    result = None
    if rank == 0:
        kernels, total_ms = kernel_table(stats, prof_steps)
        # The dominant KERNEL is the igemm_kernel template (igemm.hip): every Linear layer and convolution of the network is one of its tile
        # instantiations ("igemm_<dtype>_<BM>x<BN>x<BK>_s<stages>" families).  The roofline object prices the template as a whole (sum of
        # algorithmic FLOPs / sum of launch durations) and lists every instantiation under by_config.
        pmc, pmc_file, pmc_stale = load_pmc()
        roofline = igemm_roofline(stats, prof_steps, args.precision, pmc)
        if roofline is not None:   # every algorithmic FLOP of the forward over the wall time of a step, against the 16-bit MFMA peak
            roofline["whole_forward_frac"] = round(world * B * FWD_GFLOP_PER_FRAME[args.model_type] / (elapsed / args.steps) / 1e3 / PEAK_BF16_TFLOPS, 4)
            roofline["whole_forward_note"] = f"B x {FWD_GFLOP_PER_FRAME[args.model_type]} GFLOP per frame / ms_per_step / 2500 TFLOP/s per GPU (projection and every non-GEMM launch included in the time)"
            if world > 1:
                roofline["whole_forward_frac"] = round(roofline["whole_forward_frac"] / world, 4)
        if pmc_file and roofline and roofline.get("traffic") is not None:
            roofline["traffic_source"] = pmc_file
        elif roofline is not None and pmc_stale:
            roofline["traffic_note"] = "null: " + pmc_stale
        # second regime (SURVEY.md 8d "two regimes, report both"): the HBM-bound projection + occupancy expansion
        hb = {n: stats[n] for n in ("project_voxelise", "occ_expand") if n in stats and stats[n]["bytes"] > 0}
        roofline_hbm = None
        if hb:
            hb_bytes, hb_ms, hb_l = sum(v["bytes"] for v in hb.values()), sum(v["ms"] for v in hb.values()), sum(v["launches"] for v in hb.values())
            ach = hb_bytes / (hb_ms * 1e-3) / 1e9
            roofline_hbm = dict(bound="hbm", kernel="project_rows_kernel + occ_expand_kernel", achieved=round(ach, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                                frac=round(ach / PEAK_HBM_GBS, 4), traffic=None, bytes_per_step=hb_bytes / prof_steps, us_per_step=round(hb_ms * 1e3 / prof_steps, 2),
                                by_kernel=[dict(kernel=n, avg_launch_us=round(v["ms"] * 1e3 / v["launches"], 2), achieved=round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1),
                                                frac=round(v["bytes"] / (v["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                                                **({"traffic": round(pmc[n]["hbm_bytes_per_launch"])} if n in pmc else {})) for n, v in hb.items()])
            if all(n in pmc for n in hb):
                roofline_hbm["traffic"] = round(sum(pmc[n]["hbm_bytes_per_launch"] * hb[n]["launches"] for n in hb) / hb_l)
        dtype_name = {"mixed": "mixed fp16 + f16x3 (SOCCDPT_PREC_MIXED: fp16 MFMA operands, x3 split per the shipped precision map; f32 accumulate)"}.get(args.precision, args.precision)
        result = {
            "metric": f"frames/sec SOccDPT_V3 {args.model_type.replace('dpt_', '')} @{img}px (depth+seg+points+occupancy forward)",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "repeats": {"count": repeats, "steps_each": args.steps, "ms_per_step": [round(e / args.steps * 1e3, 4) for e in rep_elapsed],
                        "min_ms_per_step": round(min(rep_elapsed) / args.steps * 1e3, 4), "median_ms_per_step": round(elapsed / args.steps * 1e3, 4),
                        "note": "value / ms_per_step are the median repeat (rank-0 clock; with N > 1 the maximum over ranks of each rank's median)"},
            "vs_baseline": None, "dtype": dtype_name, "data": "synthetic",
            "config": {"workload": f"SOccDPT_V3 {args.model_type} full forward, compute_occ=True, camera 1920x1080",
                       "batch_per_gpu": B, "global_batch": B * world, "image": img, "weights": args.weights, "streams_per_gpu": args.streams, "hip_graph": args.graph,
                       "parallelism": f"dp{world}" if world > 1 else "single", "dist_backend": (dist.get_backend() if dist.is_initialized() else None),
                       "exchange": "RCCL all-gather of bit-packed occupancy grids (786432 B/rank)" if world > 1 else "none",
                       "precision_map_x3_groups": (sorted(g for g, f in eng.prec_map().items() if f == 3) if args.precision == "mixed" else None),
                       # shipped = the compiled-in map on the synthetic weights it was derived from; calibrated = soccdpt_prec_calibrate ran (--calibrate)
                       "precision_map_source": (net.precision_map_source(dev) if args.precision == "mixed" else None),
                       "precision_map_calibration": ({k: v for k, v in calib_report.items() if k != "x3_groups"} if calib_report else None)},
            "roofline": roofline,
            "roofline_hbm": roofline_hbm,
            "rccl_ranks": rccl_ranks, "per_rank_ms_per_step": per_rank_ms,
            "kernels": kernels,
            "device_ms_per_step": round(total_ms / prof_steps, 3),
            "ms_per_step_with_events": round(elapsed_prof / prof_steps * 1e3, 3),
            "launches_per_step": eng.launch_count() + 2,
        }
        if world > 1 and getattr(net, "occ_exchange", None) is not None and hasattr(net.occ_exchange, "window_ms"):
            wms = net.occ_exchange.window_ms()   # issue of the all-gather -> end of the OR kernel, device time, with the rows' zero-fill overlapped (soccdpt_amd/dist.py)
            result["exchange_window_ms"] = None if wms is None else round(wms, 4)

    single = rank == 0 and world == 1 and not args.graph and args.streams == 1
    side_ok = single and not args.headline_only and not args.no_side_modes

    def build(prec_name, streams=1, weights=None):
        with contextlib.redirect_stdout(io.StringIO()):
            m = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True, streams=streams,
                           model_type=args.model_type, precision=PREC_CODE[prec_name])
            m.load_state_dict(sd if weights is None else weights, strict=False)
            m = m.eval().to(dev)
            m.network(x[:1])   # binds and prepares (the uncalibrated-weights notice goes to the discarded stream)
        return m

    def time_steps(m, steps):
        for _ in range(args.warmup):
            m(x)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            m(x)
        torch.cuda.synchronize()
        return time.perf_counter() - t

    # ---- B = 1 latency: the protocol behind the paper's 47 Hz (/root/reference/SOccDPT/scripts/eval_SOccDPT.py:246-259: 50 forwards of ONE frame,
    # fps = 50 / elapsed), here WITH a device synchronisation on both sides of the 50 forwards.  x_paper_hz compares like with like.
    if single and not args.headline_only:
        x1 = x[:1].contiguous()
        for _ in range(10):
            net(x1)
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        for _ in range(50):
            out = net(x1)
        torch.cuda.synchronize()
        e1 = time.perf_counter() - t5
        result["latency_b1"] = {"ms_per_frame": round(e1 / 50 * 1e3, 4), "hz": round(50 / e1, 1), "forwards": 50, "dtype": args.precision,
                                "protocol": "50 forwards of one frame, compute_occ=True, device-synchronised on both sides (scripts/eval_SOccDPT.py:246-259 without its missing sync)"}
        result["paper_hz"] = 47.0
        result["x_paper_hz"] = round(50 / e1 / 47.0, 2)
        result["x_paper_hz_note"] = "B = 1 latency protocol against the paper's B = 1 figure; the B = 8 throughput `value` is not divided by 47 any more"
        for _ in range(5):
            net(x)   # back to the benchmark's batch (workspace layout, clocks)

    # ---- live error measurement against the library's own exact-f32 mode (product path only; the CPU oracle checks the same quantities in
    # tests/test_mixed_gpu.py and tests/test_network_gpu.py), and the side modes: the bf16 figure BASELINE configs[1] names and plain fp16 ----
    measured = {}
    if side_ok:
        xe = x   # at the BENCHMARK's batch: tile and split-K choices depend on M (VERDICT r4 #1d)

        def quantities(m):
            inv, _ = m.network(xe)
            e = m._engine(dev)
            q = {k: e.workspace_tensor(B, k).double() for k in QUANTITIES if k != "inv"}
            q["inv"] = inv.double()
            return q

        ref_net = build("f32")
        ref = quantities(ref_net)
        del ref_net

        def errors(m, ref=ref):
            q = quantities(m)
            e = {k: float((q[k] - ref[k]).norm() / ref[k].norm()) for k in QUANTITIES}
            pix = ((q["inv"] - ref["inv"]).abs() / ref["inv"].abs().clamp_min(1e-6)).flatten()
            e["inv_per_pixel_p999"] = float(pix.kthvalue(max(1, int(0.999 * pix.numel()))).values)
            e["inv_per_pixel_max"] = float(pix.max())
            m(x)   # the full forward again (projection + expansion buffers warm)
            return {k: float(f"{v:.3e}") for k, v in e.items()}

        measured[args.precision] = errors(net)
        for side in [p for p in ("mixed", "f16", "bf16") if p != args.precision]:
            m = build(side)
            err_side = errors(m)
            for _ in range(args.warmup):
                out = m(x)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            for _ in range(args.steps):
                out = m(x)
            torch.cuda.synchronize()
            es = time.perf_counter() - t3
            es_eng = m._engine(dev)
            es_eng.profile_enable(True)
            for _ in range(prof_steps):
                out = m(x)
            torch.cuda.synchronize()
            st_side = es_eng.profile_collect()
            es_eng.profile_enable(False)
            rs = igemm_roofline(st_side, prof_steps, side)
            measured[side] = err_side
            result[f"{side}_operands"] = {"value": round(B * args.steps / es, 2), "unit": "frames/s", "ms_per_step": round(es / args.steps * 1e3, 3),
                                          "device_ms_per_step": round(sum(v["ms"] for v in st_side.values()) / prof_steps, 3),
                                          "roofline": {k: rs[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_vs_blended_peak", "avg_launch_us")} if rs else None,
                                          "errors_vs_f32_mode": err_side}
            del m
    # ---- what OTHER weights get: the headline's precision map is valid for the benchmark's own synthetic draw only.  For two more
    # checkpoints -- another draw of the generator and one with trained-like statistics -- bind them, time the uncalibrated default (the library runs every
    # group in x3 operands), calibrate on six frames that are not in the benchmark batch (4 select the map at 0.85 x budget, 2 verify it at the budget
    # inside the library), time the calibrated map, and measure it on the BENCHMARK batch -- frames no part of the calibration saw -- against the
    # library's exact-f32 mode on the same weights.
    if side_ok and args.precision == "mixed" and not args.no_other_weights:
        ow = {}
        for wname in [w for w in ("salt1", "trained_like") if w != args.weights]:
            wsd = named_weights(wname, backbone)
            rf = build("f32", weights=wsd)
            ref_w = quantities(rf)
            del rf
            m = build("mixed", weights=wsd)
            src0 = m.precision_map_source(dev)
            err_x3 = errors(m, ref_w)
            e_x3 = time_steps(m, args.steps)
            with contextlib.redirect_stdout(io.StringIO()):
                rep = m.calibrate_precision(x_cal, budget=budget)
            err_cal = errors(m, ref_w)
            e_cal = time_steps(m, args.steps)
            ow[wname] = {"uncalibrated": {"precision_map_source": src0, "value": round(B * args.steps / e_x3, 2), "ms_per_step": round(e_x3 / args.steps * 1e3, 3),
                                          "worst_vs_f32_mode": max(v for k, v in err_x3.items() if k in QUANTITIES)},
                         "calibrated": {"precision_map_source": m.precision_map_source(dev), "value": round(B * args.steps / e_cal, 2), "ms_per_step": round(e_cal / args.steps * 1e3, 3),
                                        "n_groups": rep["n_groups"], "n_x3": rep["n_x3"], "n_x2w": rep["n_x2w"], "forwards": rep["forwards"],
                                        "calib_frames": rep["calib_frames"], "holdout_frames": rep["holdout_frames"], "headroom": round(rep["headroom"], 3),
                                        "library_worst_calibration_frames": float(f"{rep['worst_calibrated']:.3e}"), "library_worst_holdout_frames": float(f"{rep['worst_holdout']:.3e}"),
                                        "met_budget": bool(rep["met_budget"] and rep["met_holdout"] != 0),
                                        "heldout_benchmark_batch_vs_f32_mode": err_cal,
                                        "heldout_worst": max(v for k, v in err_cal.items() if k in QUANTITIES),
                                        "heldout_within_budget": max(v for k, v in err_cal.items() if k in QUANTITIES) <= budget},
                         "vs_headline": round((B * args.steps / e_cal) / result["value"], 3)}
            del m
        result["other_weights"] = {"budget": budget, "unit": "frames/s", "sets": ow,
                                   "note": "value (top level) runs the shipped precision map on the synthetic draw it was derived on; these are the same forward on "
                                           "checkpoints the map was NOT derived on: uncalibrated = the library's safe default for unknown weights (every group x3), "
                                           "calibrated = after net.calibrate_precision(6 frames: 4 select at headroom x budget, 2 verify at budget); heldout_* = relative L2 "
                                           "(and per-pixel p99.9 / max of the inverse depth) on the benchmark batch, which the calibration never saw"}

    if rank == 0 and result is not None:
        # Which arithmetic meets the north star's tolerance (1e-3 relative on depth maps / class logits), stated as top-level fields (VERDICT r1 #5e)
        # and, since round 4, MEASURED in this run: relative L2 of every hooked feature map, path_1, inverse depth and the class logits against the
        # library's exact-f32 mode on the same weights and frames (that mode is pinned to the fp32 CPU oracle at 1e-5 by tests/).  The bar for
        # `value` is HALF the north star (margin: VERDICT r3 #1); dpt_hybrid_384, whose fp16 error is 5.7e-3, is held to the north star itself.
        bar = 1e-3 if args.model_type == "dpt_hybrid_384" else 5e-4
        static = {"bf16": False, "f16": args.model_type != "dpt_hybrid_384", "f32": True, "f16x3": True, "mixed": True}   # tests/test_network_gpu.py, test_hybrid_gpu.py, test_mixed_gpu.py
        def worst(p):
            return max(v for k, v in measured[p].items() if k in QUANTITIES) if p in measured else None
        meets = {p: (worst(p) <= bar if p in measured else static[p]) for p in set(list(measured) + [args.precision])}
        cands = {args.precision: result["value"]}
        for p in ("mixed", "f16", "bf16"):
            if f"{p}_operands" in result:
                cands[p] = result[f"{p}_operands"]["value"]
        ok = {p: v for p, v in cands.items() if meets.get(p)}
        result["tolerance"] = {"north_star": "1e-3 relative (depth, logits), voxel indices bit-exact at the projection boundary",
                               "bar_for_value": bar, "bar_note": f"relative L2 of feat0-3, path_1, inverse depth, class logits vs the exact-f32 mode, measured in this run at the benchmark's batch ({B} frames)"
                               if measured else "not measured in this run (side modes off): the static table of tests/ applies",
                               "reading": "relative L2 per quantity (||a - ref|| / ||ref||); per_pixel = the same inverse depth element-wise, |a - ref| / max(|ref|, 1e-6): its 99.9th percentile and maximum",
                               "per_pixel": ({p: {"inv_p999": measured[p]["inv_per_pixel_p999"], "inv_max": measured[p]["inv_per_pixel_max"],
                                                  "p999_within_1e-3": measured[p]["inv_per_pixel_p999"] <= 1e-3} for p in measured} if measured else None),
                               "dtype_of_value": args.precision, "value_meets_tolerance": bool(meets[args.precision]),
                               "worst_measured": {p: worst(p) for p in measured}, "measured": measured,
                               "meets_north_star_1e-3": {p: (worst(p) <= 1e-3) for p in measured},
                               "fastest_dtype_meeting_tolerance": (max(ok, key=ok.get) if ok else None),
                               "value_meeting_tolerance": (max(ok.values()) if ok else None)}

    # ---- the same workload dealt to two concurrent sub-batches on internal streams (soccdpt_set_streams(2), eager): bit for bit the
    # result of running the two sub-batches one after the other (tools/multistream_split_check.py; equal to the whole-batch result too
    # unless the sub-batch size flips a split-K decision, which moves last bits), faster because the latency-bound launches of one half overlap the other half's.  Reported
    # beside `value`, which stays on one stream so that the per-kernel durations behind `roofline` are those of kernels running alone.
    if single and B >= 2 and args.with_two_streams and not args.headline_only:
        net2 = build(args.precision, streams=2)
        for _ in range(args.warmup):
            out = net2(x)
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        for _ in range(args.steps):
            out = net2(x)
        torch.cuda.synchronize()
        e2 = time.perf_counter() - t4
        result["two_streams"] = {"value": round(B * args.steps / e2, 2), "unit": "frames/s", "ms_per_step": round(e2 / args.steps * 1e3, 3),
                                 "experimental": True,
                                 "note": "EXPERIMENTAL opt-in mode (soccdpt_set_streams(2)): same forward, batch dealt to 2 concurrent sub-batches on internal streams"}
        del net2

    # ---- whole forwards in flight: N engines with their own workspaces and outputs, forward i on stream i mod N, nothing between them but
    # the stream order of each engine's own launches.  The HBM-bound tail of one step (depth tail -> projection -> expansion: little MFMA / LDS use) and the
    # latency-bound encoder launches of the next overlap.  Outputs are bit-compared with the sequential run first.  Not the headline: the kernels of
    # `roofline` are timed running alone.
    if single and args.in_flight > 1 and not args.headline_only:
        nets = [net] + [build(args.precision) for _ in range(args.in_flight - 1)]
        if args.precision == "mixed" and calib_report is not None:
            for m in nets[1:]:
                with contextlib.redirect_stdout(io.StringIO()):
                    m.calibrate_precision(x_cal, budget=budget)
        streams = [torch.cuda.Stream(device=dev) for _ in nets]
        ref_out = [t.clone() for t in net(x)]
        torch.cuda.synchronize()
        same = True
        outs = [None] * len(nets)
        for r in range(2):
            for i, (m, s) in enumerate(zip(nets, streams)):
                with torch.cuda.stream(s):
                    outs[i] = m(x)
        torch.cuda.synchronize()
        for o in outs:
            same = same and all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(o, ref_out))
        def run(n):
            for k in range(n):
                i = k % len(nets)
                with torch.cuda.stream(streams[i]):
                    outs[i] = nets[i](x)
        run(args.warmup)
        torch.cuda.synchronize()
        t6 = time.perf_counter()
        run(args.steps)
        torch.cuda.synchronize()
        e6 = time.perf_counter() - t6
        result["pipelined"] = {"in_flight": len(nets), "value": round(B * args.steps / e6, 2), "unit": "frames/s", "ms_per_step": round(e6 / args.steps * 1e3, 3),
                               "vs_value": round(B * args.steps / e6 / result["value"], 4), "bit_identical_to_sequential": bool(same),
                               "note": f"{len(nets)} independent engines (own workspace, own outputs), forward i on stream i mod {len(nets)}, no synchronisation between forwards; "
                                       "throughput of the same step when the caller keeps more than one frame batch in flight -- a side field, `value` stays one forward at a time"}
        del nets, outs

    # ---- CPU baseline: the oracle on this host's cores, bounded sample, rank 0 at N = 1 only ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import soccdpt_ref as R
        # the GPU box gives one GPU's share of the host: 16 cores (more threads only oversubscribe)
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        cores = max(1, min(avail, 16))
        torch.set_num_threads(cores)
        nb = max(1, min(args.cpu_sample_frames, B))
        xs = x[:nb].cpu()
        sd_cpu = {k: v.cpu() for k, v in sd.items()}
        R.soccdpt_v3_forward(sd_cpu, xs[:1], backbone=backbone, sigmoid=False)   # warm-up (allocator, oneDNN primitives)
        reps, tcpu = 0, 0.0
        while tcpu < 12.0 and reps < 40:   # about 12 s of CPU work (a bounded sample; the oracle does 3-4 frames/s here)
            t2 = time.perf_counter()
            R.soccdpt_v3_forward(sd_cpu, xs, backbone=backbone, sigmoid=False)
            tcpu += time.perf_counter() - t2
            reps += 1
        result["cpu_baseline"] = {"value": round(nb * reps / tcpu, 3), "unit": "frames/s", "cores": cores, "kind": "port",
                                  "sample": f"{reps} x batch {nb} of the same synthetic workload, fp32 PyTorch-CPU oracle "
                                            f"(oracle/soccdpt_ref.py), {tcpu:.1f} s"}
    if rank == 0 and "cpu_baseline" not in result:   # the key is always there: a consumer of the N > 1 line need not special-case it
        result["cpu_baseline"] = None
        result["cpu_baseline_note"] = "timed on rank 0 at N = 1 only (bench contract)" if world > 1 else "switched off (--no-cpu-baseline / --headline-only)"
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(result) + "\n").encode())
    if dist.is_initialized():
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
