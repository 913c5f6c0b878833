"""GPU: the RCCL exchange path (all-gather of packed grids + occ_or + expand) under torchrun with one rank
(SOCCDPT_FORCE_DIST=1; more ranks need more GPUs than a test box has): the union of one rank's grid with itself must equal
the single-process result bit for bit.  The N-rank logic is covered on CPU by tests/test_dist_cpu.py (gloo, world_size 2)."""
import hashlib
import os
import subprocess
import sys
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_rccl_exchange_single_rank_matches_plain_forward(gpu_device):
    from soccdpt_amd.model.SOccDPT import SOccDPT_V3
    from soccdpt_amd.utils.synth import synth_input, synth_state_dict, write_synth_calib
    calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
    net = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True)
    net.load_state_dict(synth_state_dict(alias_pretrained=True), strict=False)
    net = net.eval().to(gpu_device)
    inv_up, _, _, occ = net(synth_input(2, seed0=40).to(gpu_device))
    torch.cuda.synchronize()
    h = lambda t: hashlib.sha1(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()
    want = (h(net.last_occ_bits), h(inv_up), str(int((occ[0] > 0).sum())))
    env = dict(os.environ, SOCCDPT_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    worker = os.path.join(os.path.dirname(__file__), "dist_worker_gpu.py")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29547", worker], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1].split()
    assert tuple(line[1:4]) == want


def test_bench_two_rank_rehearsal_on_one_gpu(gpu_device, tmp_path):
    """The N > 1 path of bench.py (sharded frames, barrier + max-over-ranks timing, the packed-grid exchange, rank-0 JSON line) under
    torch.distributed.run with TWO ranks on the one GPU of a test box: SOCCDPT_DIST_REHEARSAL=1 swaps RCCL for gloo (RCCL refuses two ranks
    on one device), everything else is the code the 8-GPU scaling run executes.  The line must report both ranks and whole-job throughput."""
    import json
    env = dict(os.environ, SOCCDPT_DIST_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    # bench.py is called the way the driver may call it: plainly, with --gpus 2 and no launcher around it.  The parent starts the two ranks itself
    # (python -m torch.distributed.run as a child process, before anything touches the GPU), relays rank 0's line and checks rccl_ranks == 2.
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "5", "--warmup", "2", "--prewarm", "3", "--batch", "2"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # exactly ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and len(d["per_rank_ms_per_step"]) == 2
    assert d["config"]["global_batch"] == 4 and d["scaling"] == "weak" and d["config"]["dist_backend"] == "gloo"
    assert abs(d["value"] - 4 * 5 / (d["ms_per_step"] * 5e-3)) / d["value"] < 1e-3      # whole-job frames / max-over-ranks time
    assert d["repeats"]["count"] * d["repeats"]["steps_each"] >= 25 and len(d["repeats"]["ms_per_step"]) == d["repeats"]["count"]
    assert d["cpu_baseline"] is None and "N = 1" in d["cpu_baseline_note"]                  # the key is always there; the CPU leg runs at N = 1 only


def test_bench_config3_four_rank_rehearsal_on_one_gpu(gpu_device):
    """`python bench.py --gpus N --config 3` (dpt_swin2_base_384, 8 frames per rank: BASELINE configs[3] is N = 8, 64 frames) rehearsed with FOUR gloo
    ranks on the one GPU of a test box -- the pool allows at most 6 processes on a card, so the 8-rank job itself cannot be rehearsed here; the
    8 x 8 sharding and the 8-rank exchange are covered on CPU (tests/test_dist_cpu.py::test_eight_ranks_split_exchange_64_frames).  The line must
    name the model, report four ranks, 32 frames, and the exchange window (all-gather + OR, with the rows' zero-fill overlapped)."""
    import json
    env = dict(os.environ, SOCCDPT_DIST_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", "4", "--config", "3", "--steps", "3", "--warmup", "1", "--prewarm", "2"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["rccl_ranks"] == 4 and len(d["per_rank_ms_per_step"]) == 4
    assert "swin2_base_384" in d["metric"] and d["config"]["global_batch"] == 32 and d["config"]["batch_per_gpu"] == 8
    assert d["exchange_window_ms"] is not None and d["exchange_window_ms"] > 0
    assert d["dtype"].startswith("mixed") and d["tolerance"]["value_meets_tolerance"] is True


def test_split_exchange_and_shared_rows_match_the_plain_forward(gpu_device):
    """One forced rank over RCCL: the split exchange (async all-gather, zero-fill of the dense rows meanwhile, OR, set bits) returns the occupancy
    grid of the plain forward bit for bit; share_occupancy_rows=True returns the same values as a stride-0 expand of one row."""
    from soccdpt_amd.model.SOccDPT import SOccDPT_V3
    from soccdpt_amd.utils.synth import synth_input, synth_state_dict, write_synth_calib
    calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
    sd = synth_state_dict(alias_pretrained=True)
    x = synth_input(3, seed0=41).to(gpu_device)

    def build(**kw):
        net = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True, **kw)
        net.load_state_dict(sd, strict=False)
        return net.eval().to(gpu_device)
    occ = build()(x)[3]
    shared = build(share_occupancy_rows=True)(x)[3]
    torch.cuda.synchronize()
    assert shared.shape == occ.shape and shared.stride(0) == 0 and torch.equal(shared, occ)
    # zero + set == expand on the same bits (the two halves the multi-GPU path runs around the collective)
    net = build()
    out = net(x)
    eng = net._engine(gpu_device)
    occ2 = torch.full_like(out[3], 7.0)
    eng.occ_zero(3, occ2)
    eng.occ_set(net.last_occ_bits, 3, occ2)
    torch.cuda.synchronize()
    assert torch.equal(occ2, out[3])


def test_data_parallel_training_rehearsal(gpu_device):
    """Data-parallel training (soccdpt_amd.dist.attach_training) with TWO ranks on the one GPU of a test box (gloo rehearsal): the exchanged
    gradient equals the mean of the ranks' local gradients, and after two Adam steps both replicas hold bit-identical weights and BatchNorm
    buffers.  On an 8-GPU node the same code runs with backend nccl (RCCL all-reduce over xGMI), one rank per GPU."""
    env = dict(os.environ, SOCCDPT_DIST_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    worker = os.path.join(os.path.dirname(__file__), "dist_train_worker_gpu.py")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29561", worker], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = sorted(l.split() for l in r.stdout.splitlines() if l.startswith("RESULT"))
    assert len(lines) == 2
    assert all(float(l[2]) < 1e-6 for l in lines), lines          # averaged gradient == mean of the local gradients
    assert lines[0][3] == lines[1][3], lines                       # identical replicas after two optimizer steps
    assert all(int(l[4]) >= 3 for l in lines)
