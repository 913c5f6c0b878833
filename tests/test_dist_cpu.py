"""CPU (gloo, world_size 2): the data-parallel sharding + packed-occupancy exchange logic
(soccdpt_amd/dist.py).  The per-rank compute is stood in for by the CPU oracle; what is under test is
the shard arithmetic, the all-gather of packed grids and the union semantics:
N-rank result == 1-rank result on the same global batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from soccdpt_amd.dist import OccExchange, attach, gather_occ_bits, shard_range


def test_shard_range_partitions_batch():
    for gb in (1, 7, 8, 64):
        for world in (1, 2, 3, 8):
            spans = [shard_range(gb, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import cref
    from tests.golden_inputs import proj_inputs
    inv, seg = proj_inputs(seed=77, B=4, S=64)     # small maps keep the CPU oracle fast
    lo, hi = shard_range(4, rank, world)
    local = cref.project(inv[lo:hi], seg[lo:hi], want=("occ_bits",))["occ_bits"]
    bits = torch.from_numpy(local.view(np.int32).copy())
    gathered = gather_occ_bits(bits)
    assert tuple(gathered.shape) == (world, bits.numel())
    # the exchange object with an injected CPU reducer (the GPU path uses the HIP occ_or kernel)
    def or_reduce(g):
        out = g[0].clone()
        for i in range(1, g.shape[0]):
            out |= g[i]
        return out
    union = OccExchange(or_reduce=or_reduce)(None, bits)
    np.save(os.path.join(outdir, f"union_{rank}.npy"), union.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_union_equals_single_rank(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    from oracle import cref
    from tests.golden_inputs import proj_inputs
    inv, seg = proj_inputs(seed=77, B=4, S=64)
    whole = cref.project(inv, seg, want=("occ_bits",))["occ_bits"].view(np.int32)
    u0 = np.load(tmp_path / "union_0.npy")
    u1 = np.load(tmp_path / "union_1.npy")
    assert np.array_equal(u0, u1)          # every rank ends with the same union grid
    assert np.array_equal(u0, whole)       # ... equal to the single-rank grid of the global batch
    assert int(np.unpackbits(whole.view(np.uint8)).sum()) > 0


def _worker_attach(rank, world, port, outdir, gb):
    """The path bench.py takes: init -> attach(net) -> net.occ_exchange(eng, bits), with an uneven batch split."""
    import types
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from oracle import cref
    from tests.golden_inputs import proj_inputs
    inv, seg = proj_inputs(seed=91, B=gb, S=64)
    lo, hi = shard_range(gb, rank, world)
    assert hi > lo
    bits = torch.from_numpy(cref.project(inv[lo:hi], seg[lo:hi], want=("occ_bits",))["occ_bits"].view(np.int32).copy())

    def or_reduce(g):
        out = g[0].clone()
        for i in range(1, g.shape[0]):
            out |= g[i]
        return out
    net = attach(types.SimpleNamespace(occ_exchange=None), or_reduce=or_reduce)
    assert net.occ_exchange is not None and dist.get_world_size() == world
    for _ in range(2):                       # the receive buffer is reused on the second call
        union = net.occ_exchange(None, bits)
    np.save(os.path.join(outdir, f"union_{rank}.npy"), union.numpy())
    np.save(os.path.join(outdir, f"shard_{rank}.npy"), np.array([lo, hi]))
    dist.barrier()
    dist.destroy_process_group()


def test_four_ranks_uneven_shards_through_attach(tmp_path):
    """world_size 4, global batch 7 -> shards of 2, 2, 2, 1 frames: every rank ends with the single-rank union grid."""
    gb, world = 7, 4
    mp.spawn(_worker_attach, args=(world, _free_port(), str(tmp_path), gb), nprocs=world, join=True)
    from oracle import cref
    from tests.golden_inputs import proj_inputs
    inv, seg = proj_inputs(seed=91, B=gb, S=64)
    whole = cref.project(inv, seg, want=("occ_bits",))["occ_bits"].view(np.int32)
    sizes = []
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"union_{r}.npy"), whole), r
        lo, hi = np.load(tmp_path / f"shard_{r}.npy")
        sizes.append(int(hi - lo))
    assert sizes == [2, 2, 2, 1]
    assert int(np.unpackbits(whole.view(np.uint8)).sum()) > 0


def test_init_from_env_refuses_multi_rank_without_port(monkeypatch):
    from soccdpt_amd.dist import init_from_env
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.delenv("MASTER_PORT", raising=False)
    with pytest.raises(RuntimeError):
        init_from_env("gloo")


def _grad_worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from soccdpt_amd.dist import GradExchange
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(5000, generator=g)
    before = flat.clone()
    ex = GradExchange(bucket_elems=1000)
    runs = [[0, 1536], [2048, 4800]]            # two runs of trainable tensors; [1536, 2048) and the tail belong to frozen ones
    ex(flat, runs)
    assert ex.calls == 2 + 3                    # ceil(1536 / 1000) + ceil(2752 / 1000) collectives
    buf = torch.full((4,), float(rank))
    ex.average_buffers([buf])
    np.save(os.path.join(outdir, f"grad_{rank}.npy"), np.stack([before.numpy(), flat.numpy()]))
    np.save(os.path.join(outdir, f"buf_{rank}.npy"), buf.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_grad_exchange_averages_trainable_runs(tmp_path, world):
    """Data-parallel training exchange (soccdpt_amd.dist.GradExchange): inside the runs of trainable tensors every rank ends with the mean of
    the ranks' gradients, bucketed; outside them (frozen tensors) the buffer is untouched; BatchNorm buffers are averaged."""
    port = _free_port()
    mp.spawn(_grad_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    data = [np.load(tmp_path / f"grad_{r}.npy") for r in range(world)]
    mean = np.mean([d[0] for d in data], axis=0, dtype=np.float64)
    for r in range(world):
        before, after = data[r]
        for lo, hi in ([0, 1536], [2048, 4800]):
            assert np.allclose(after[lo:hi], mean[lo:hi], rtol=1e-6, atol=1e-7)
        assert np.array_equal(after[1536:2048], before[1536:2048]) and np.array_equal(after[4800:], before[4800:])
        assert np.allclose(np.load(tmp_path / f"buf_{r}.npy"), (world - 1) / 2.0)
    for r in range(1, world):
        assert np.array_equal(data[r][1][:1536], data[0][1][:1536])        # replicas hold bit-identical averaged gradients


def _loss_worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from soccdpt_amd.dist import all_reduce_mean_scalar
    from soccdpt_amd.scripts.train_SOccDPT import ReduceLROnPlateau
    from soccdpt_amd.utils.optim import Adam
    opt = Adam([torch.nn.Parameter(torch.zeros(4))], lr=1e-3)
    sched = ReduceLROnPlateau(opt, patience=2)
    lrs = []
    for step in range(12):
        local = 1.0 + (0.5 if rank == 0 else -0.5) * ((-1) ** step) + (0.0 if step < 2 else 0.1)   # rank-local losses disagree about "plateau"
        g = all_reduce_mean_scalar(local)
        sched.step(g)
        lrs.append((g, opt.lr))
    np.save(os.path.join(outdir, f"lrs_{rank}.npy"), np.array(lrs))
    dist.barrier()
    dist.destroy_process_group()


def test_training_loss_is_all_reduced_before_the_scheduler(tmp_path):
    """Data-parallel training: ReduceLROnPlateau steps on the mean loss over the ranks, so every rank cuts the learning rate at
    the same step although the rank-local shard losses differ."""
    world, port = 2, _free_port()
    mp.spawn(_loss_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(tmp_path / "lrs_0.npy"), np.load(tmp_path / "lrs_1.npy")
    assert np.array_equal(a, b)
    assert a[-1, 1] < a[0, 1]           # the plateau was detected (identically)


def _worker_split(rank, world, port, outdir, gb):
    """The split exchange SOccDPT._finish_occupancy drives at N > 1: start() (all-gather issued), the caller's zero-fill, finish() (wait + OR)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from oracle import cref
    from tests.golden_inputs import proj_inputs
    inv, seg = proj_inputs(seed=13, B=gb, S=64)
    lo, hi = shard_range(gb, rank, world)
    bits = torch.from_numpy(cref.project(inv[lo:hi], seg[lo:hi], want=("occ_bits",))["occ_bits"].view(np.int32).copy())

    def or_reduce(g):
        out = g[0].clone()
        for i in range(1, g.shape[0]):
            out |= g[i]
        return out
    ex = OccExchange(or_reduce=or_reduce)
    for _ in range(2):
        ticket = ex.start(bits)
        union = ex.finish(None, bits, ticket)
    assert ex.window_ms() is None            # CPU tensors carry no device events
    np.save(os.path.join(outdir, f"union_{rank}.npy"), union.numpy())
    np.save(os.path.join(outdir, f"shard_{rank}.npy"), np.array([lo, hi]))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_split_exchange_64_frames(tmp_path):
    """BASELINE configs[3]'s shape of the job: a 64-frame global batch over world_size 8 shards 8 x 8 (contiguous, in rank order), and the split
    exchange (start / finish) leaves every rank with the single-rank union grid.  (The frames here are small synthetic maps; the grid is the real one.)"""
    gb, world = 64, 8
    assert [shard_range(gb, r, world) for r in range(world)] == [(8 * r, 8 * r + 8) for r in range(world)]
    mp.spawn(_worker_split, args=(world, _free_port(), str(tmp_path), gb), nprocs=world, join=True)
    from oracle import cref
    from tests.golden_inputs import proj_inputs
    inv, seg = proj_inputs(seed=13, B=gb, S=64)
    whole = cref.project(inv, seg, want=("occ_bits",))["occ_bits"].view(np.int32)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"union_{r}.npy"), whole), r
        assert list(np.load(tmp_path / f"shard_{r}.npy")) == [8 * r, 8 * r + 8]


def test_multi_rank_bench_line_is_parseable():
    """the first real `python bench.py --gpus 8 --config 3` cannot come back unparseable.  tests/golden/bench_line_4rank_rehearsal_config3.json
    is the line a 4-rank rehearsal of exactly that command printed on a one-GPU box (tests/test_dist_gpu.py runs the rehearsal itself on the GPU
    side; recorded with gpurun in round 5).  The N > 1 result object must carry everything the driver and the judge read."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bench_line_4rank_rehearsal_config3.json")
    lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "bench.py prints exactly one JSON line"
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "roofline_hbm", "cpu_baseline", "rccl_ranks", "per_rank_ms_per_step", "exchange_window_ms", "tolerance"):
        assert k in d, k
    assert d["n_gpus"] == d["rccl_ranks"] == len(d["per_rank_ms_per_step"]) == 4
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["config"]["global_batch"] == 4 * d["config"]["batch_per_gpu"] and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["global_batch"] / d["ms_per_step"] * 1e3) / d["value"] < 1e-3     # whole-job frames/s from the max-over-ranks time
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_of_16bit_peak", "whole_forward_frac"):
        assert k in r, k
    assert r["bound"] == "mfma" and 0 < r["frac_of_16bit_peak"] <= r["frac"] < 1 and 0 < r["whole_forward_frac"] < 1
    assert d["cpu_baseline"] is None and "N = 1" in d["cpu_baseline_note"]      # the contract times the CPU leg on rank 0 at N = 1 only
    assert d["exchange_window_ms"] > 0 and d["tolerance"]["dtype_of_value"] == "mixed"
