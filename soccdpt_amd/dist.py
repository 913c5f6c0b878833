"""Data-parallel sharding of the SOccDPT_V3 forward over the GPUs of one node.

One process per GPU.  Frames are independent through encoder, decoder, heads and back-projection; the
only cross-frame coupling of the reference is the occupancy grid, which is the UNION over the whole batch
written to every batch row (/root/reference/SOccDPT/model/SOccDPT.py:449-455).  Each rank therefore ORs
its own frames into a bit-packed grid (786,432 B for 256x256x32x3) and the ranks exchange only those
packed grids: one RCCL all-gather (backend "nccl" is RCCL on ROCm) followed by a local OR-reduce kernel and
the bits->f32 expansion.  The dense 25 MB f32 grids never cross xGMI.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous batch shard [lo, hi) of `rank`; the first (global_batch % world) ranks get one extra frame."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def rehearsal() -> bool:
    """SOCCDPT_DIST_REHEARSAL=1: run the N-rank code path of bench.py / attach() on a box with FEWER GPUs than ranks -- backend gloo, rank r on
    device r % device_count -- so that the sharding, barrier, timing and exchange plumbing can be exercised before an 8-GPU run.  RCCL itself is
    not involved (it refuses two ranks on one device); the product configuration is backend "nccl", one rank per GPU."""
    import os
    return os.environ.get("SOCCDPT_DIST_REHEARSAL", "0") == "1"


def gather_occ_bits(bits: torch.Tensor, group=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """All-gather the packed local grids: [words] int32 -> [world, words] int32 (same device).  `out`: reusable flat buffer."""
    world = dist.get_world_size(group)
    n = world * bits.numel()
    flat = out if (out is not None and out.numel() == n and out.device == bits.device and out.dtype == bits.dtype) else \
        torch.empty((n,), dtype=bits.dtype, device=bits.device)
    if bits.is_cuda and dist.get_backend(group) == "gloo":   # rehearsal only: gloo moves host memory
        host = torch.empty((n,), dtype=bits.dtype)
        dist.all_gather_into_tensor(host, bits.detach().cpu().contiguous().reshape(-1), group=group)
        flat.copy_(host)
    else:
        dist.all_gather_into_tensor(flat, bits.contiguous().reshape(-1), group=group)
    return flat.reshape(world, bits.numel())


def all_reduce_mean_scalar(value, group=None) -> float:
    """Mean of a scalar over the ranks (the training loop's loss for ReduceLROnPlateau / logging): identical on every rank.  `value` may be a
    device tensor (the step's loss as the library left it): it is reduced where it lives -- no host-to-device copy, one .item() -- or a host float."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return float(value.item()) if torch.is_tensor(value) else float(value)
    world = dist.get_world_size(group)
    if torch.is_tensor(value):
        t = value.detach().reshape(1).to(torch.float64)
        if t.is_cuda and dist.get_backend(group) == "gloo":   # rehearsal only: gloo moves host memory
            t = t.cpu()
    else:
        t = torch.tensor([float(value)], dtype=torch.float64)
        if dist.get_backend(group) == "nccl":
            t = t.to(torch.device("cuda", torch.cuda.current_device()))
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t.item()) / world


def barrier(group=None):
    """dist.barrier on the device this rank owns (RCCL wants the device named; gloo takes none)."""
    if not dist.is_initialized():
        return
    if dist.get_backend(group) == "nccl":
        dist.barrier(group=group, device_ids=[torch.cuda.current_device()])
    else:
        dist.barrier(group=group)


class OccExchange:
    """Callable installed as `net.occ_exchange`: local packed grid -> union over all ranks."""

    def __init__(self, group=None, or_reduce: Optional[Callable] = None):
        self.group = group
        self._or_reduce = or_reduce  # tests on CPU (gloo) inject a reducer; on GPU the HIP kernel is used
        self._flat: Optional[torch.Tensor] = None   # reused receive buffer (world x 786,432 B)
        self.windows = []                           # (start, end) event pairs of the last exchanges (window_ms)

    def __call__(self, eng, bits: torch.Tensor) -> torch.Tensor:
        gathered = gather_occ_bits(bits, self.group, out=self._flat)
        self._flat = gathered.reshape(-1)
        if self._or_reduce is not None:
            return self._or_reduce(gathered)
        eng.occ_or(bits, gathered, gathered.shape[0])   # in place: bits |= every rank's grid (its own is among them)
        return bits

    # ---- split form: the collective in flight while the caller zero-fills the dense rows (SOccDPT._finish_occupancy) ----
    def start(self, bits: torch.Tensor):
        """Issue the all-gather of the packed grids WITHOUT blocking the caller's stream: RCCL runs it on the process group's own stream (ordered
        after the projection kernel that wrote `bits`), so whatever the caller launches next -- the zero-fill of the B dense rows, which depends
        on no rank's voxels -- overlaps the exchange.  Returns a ticket for finish().  gloo (CPU tests, the one-GPU rehearsal) gathers synchronously."""
        world = dist.get_world_size(self.group)
        n = world * bits.numel()
        ev = None
        if bits.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if bits.is_cuda and dist.get_backend(self.group) == "nccl":
            if self._flat is None or self._flat.numel() != n or self._flat.device != bits.device:
                self._flat = torch.empty((n,), dtype=bits.dtype, device=bits.device)
            work = dist.all_gather_into_tensor(self._flat, bits.contiguous().reshape(-1), group=self.group, async_op=True)
            return (work, self._flat.reshape(world, bits.numel()), ev)
        gathered = gather_occ_bits(bits, self.group, out=self._flat)
        self._flat = gathered.reshape(-1)
        return (None, gathered, ev)

    def finish(self, eng, bits: torch.Tensor, ticket) -> torch.Tensor:
        work, gathered, ev = ticket
        if work is not None:
            work.wait()            # stream-level: the caller's stream waits for the collective, the host does not
        if self._or_reduce is not None:
            out = self._or_reduce(gathered)
        else:
            eng.occ_or(bits, gathered, gathered.shape[0])
            out = bits
        if ev is not None:
            ev[1].record()
            self.windows.append(ev)
            del self.windows[:-32]
        return out

    def window_ms(self):
        """Mean device time from the issue of the all-gather to the end of the OR kernel over the recorded steps (the exchange window
        bench.py reports at N > 1); includes whatever the caller overlapped with it.  Synchronises."""
        if not self.windows:
            return None
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self.windows) / len(self.windows)


class GradExchange:
    """Callable installed as `net.grad_exchange` by attach_training: averages the flat f32 gradient buffer of the training step over the ranks.
    `runs` are the [lo, hi) element ranges that hold this step's trainable tensors (PatchWiseInplace patches and the unfrozen encoder prefix
    are contiguous index ranges of the parameter list, so a step needs one to three collectives); each is one all-reduce over RCCL / xGMI.
    Buckets are capped at `bucket_elems` so that a 166 MB all-trainable step is a few ring-friendly messages rather than one."""

    def __init__(self, group=None, bucket_elems: int = 16 << 20):
        self.group = group
        self.bucket = int(bucket_elems)
        self.calls = 0      # collectives issued (tests)

    def _all_reduce_mean(self, t: torch.Tensor):
        world = dist.get_world_size(self.group)
        if t.is_cuda and dist.get_backend(self.group) == "gloo":   # rehearsal only: gloo moves host memory
            host = t.detach().cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(host)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        t.mul_(1.0 / world)
        self.calls += 1

    def __call__(self, flat: torch.Tensor, runs):
        for lo, hi in runs:
            for a in range(lo, hi, self.bucket):
                self._all_reduce_mean(flat[a:min(a + self.bucket, hi)])

    def average_buffers(self, tensors):
        for t in tensors:
            self._all_reduce_mean(t)


def attach_training(net, group=None, bucket_elems: int = 16 << 20):
    """Data-parallel training of `net` (SOccDPT_V3): every rank runs train_forward / backward on its own batch shard; backward() then averages
    the gradients over the ranks (GradExchange), so identical optimizer steps keep the replicas identical.  The seg head's BatchNorm normalises
    with the LOCAL batch statistics (torch DDP's default, no SyncBN); its running buffers are averaged every step."""
    import os
    force = os.environ.get("SOCCDPT_FORCE_DIST", "0") == "1"
    if dist.is_initialized() and (dist.get_world_size(group) > 1 or force):
        net.grad_exchange = GradExchange(group, bucket_elems)
    return net


def init_from_env(backend: str = "nccl"):
    """Initialise torch.distributed from torchrun's env (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*)."""
    import os
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    force = os.environ.get("SOCCDPT_FORCE_DIST", "0") == "1"  # exercise the RCCL path with a single rank (tests)
    if rehearsal():
        backend = "gloo"
        local = local % max(torch.cuda.device_count(), 1)
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            if world > 1:   # ranks cannot agree on a port by themselves: the launcher (torchrun --master-port) must name it
                raise RuntimeError("soccdpt_amd.dist: WORLD_SIZE > 1 but MASTER_PORT is not set (launch with torch.distributed.run)")
            import socket
            with socket.socket() as s:   # single forced rank: any free port (a fixed default collides between concurrent jobs)
                s.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s.getsockname()[1])
        import datetime
        timeout = datetime.timedelta(minutes=60)   # a validation pass on rank 0 may hold the other ranks at a barrier for a long time (train_SOccDPT.py)
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=torch.device("cuda", local), timeout=timeout)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timeout)
    return rank, local, world


def attach(net, group=None, or_reduce: Optional[Callable] = None):
    """Make `net` (SOccDPT / SOccDPT_V3) produce the union-over-all-ranks occupancy grid.  `or_reduce`: CPU stand-in for the
    HIP OR-reduce kernel (gloo tests only; the product path leaves it None)."""
    import os
    force = os.environ.get("SOCCDPT_FORCE_DIST", "0") == "1"
    if dist.is_initialized() and (dist.get_world_size(group) > 1 or force):
        net.occ_exchange = OccExchange(group, or_reduce=or_reduce)
    return net
