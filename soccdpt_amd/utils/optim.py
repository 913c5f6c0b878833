"""Training-loop plumbing of the reference on the MI355X path:

* `Adam` — drop-in for the `torch.optim.Adam(net.parameters(), lr, betas, eps, weight_decay, amsgrad=False)` the reference builds
  (/root/reference/SOccDPT/scripts/train_SOccDPT.py:311-318): `zero_grad(set_to_none=True)` / `step()`, parameters without a
  gradient are skipped, per-parameter step counts — the update itself is ONE fused multi-tensor HIP launch per 48 tensors
  (csrc/adam.hip) instead of ~8 ATen kernels per tensor.
* `patches` / `PatchWiseInplace` — the patch-wise schedule of /root/reference/SOccDPT/patchwise_training/__init__.py:148-252:
  the parameters that are trainable when the iterator is created are visited in consecutive groups of
  M = ceil(N * percentage); inside a group only its members have requires_grad, afterwards the original flags return.
"""
from __future__ import annotations

import ctypes
import math
from typing import Iterable, Iterator, List, Tuple

import torch

from ..lib import _stream_ptr, load_library


class Adam:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, amsgrad: bool = False):
        assert not amsgrad, "amsgrad=False is what the reference uses; the fused kernel has no max-tracking variant"
        self.params: List[torch.nn.Parameter] = [p for p in params]
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.state = {}

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            if p.grad is not None:
                p.grad = None if set_to_none else p.grad.zero_()

    @torch.no_grad()
    def step(self):
        groups = {}   # parameters that share a step count go into one fused call
        for p in self.params:
            if p.grad is None:
                continue
            assert p.is_cuda and p.dtype == torch.float32 and p.is_contiguous(), "fused Adam: f32 contiguous cuda parameters (no CPU fallback)"
            st = self.state.get(p)
            if st is None:
                st = self.state[p] = dict(step=0, exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p))
            st["step"] += 1
            groups.setdefault((st["step"], p.device), []).append((p, st))
        L = load_library()
        for (step, dev), items in groups.items():
            n = len(items)
            arr = lambda ptrs: (ctypes.c_void_p * n)(*ptrs)
            ps = arr([p.data_ptr() for p, _ in items])
            grads = [p.grad.contiguous() for p, _ in items]   # kept alive until after the launch (a non-contiguous grad makes a temporary)
            gs = arr([g.data_ptr() for g in grads])
            ms = arr([s["exp_avg"].data_ptr() for _, s in items])
            vs = arr([s["exp_avg_sq"].data_ptr() for _, s in items])
            sizes = (ctypes.c_size_t * n)(*[p.numel() for p, _ in items])
            with torch.cuda.device(dev):
                rc = L.soccdpt_adam_step(n, ps, gs, ms, vs, sizes, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, step,
                                         _stream_ptr(dev))
            if rc != 0:
                raise RuntimeError("soccdpt_adam_step failed: " + L.soccdpt_last_error(None).decode())
            # the kernel wrote through raw pointers: bump the autograd version counters so that everything keyed on
            # tensor._version (SOccDPT_V3._sync_weights re-runs soccdpt_prepare) sees the update
            torch._C._increment_version([p for p, _ in items])
            del grads


def patches(n_trainable: int, percentage: float) -> List[range]:
    """Index ranges (into the list of initially trainable parameters) of the successive patches."""
    assert n_trainable > 0, "The number of parameters is 0, check the network"
    m = min(math.ceil(n_trainable * percentage), n_trainable)
    assert m > 0, f"The number of parameters to unfreeze is 0, choose a higher training percentage N={n_trainable}, M={m}"
    return [range(lo, min(lo + m, n_trainable)) for lo in range(0, n_trainable, m)]


class PatchWiseInplace:
    """`for net_patch in PatchWiseInplace(net, percentage): ...` — yields `net` once per patch with only that patch trainable."""

    def __init__(self, net: torch.nn.Module, train_percentage: float):
        assert isinstance(net, torch.nn.Module), "The network must be a torch.nn.Module, got {}".format(type(net))
        self.net = net
        self._all = list(net.parameters())
        self._flags = [p.requires_grad for p in self._all]
        self._trainable = [p for p in self._all if p.requires_grad]
        self._patches = patches(len(self._trainable), train_percentage)

    def __len__(self) -> int:
        return len(self._patches)

    def __iter__(self) -> Iterator[torch.nn.Module]:
        try:
            for patch in self._patches:
                for i, p in enumerate(self._trainable):
                    p.requires_grad = i in patch
                yield self.net
        finally:   # the reference restores the flags when the iteration is exhausted; also do so if the loop is left early
            for p, f in zip(self._all, self._flags):
                p.requires_grad = f


class GradScaler:
    """torch.cuda.amp.GradScaler for the HIP training step's fp16 amp mode (the reference: scripts/train_SOccDPT.py:340,390-393).
    `scale(d_inv, d_seg)` multiplies the criterion's output gradients (= scaling the loss), `step(optimizer, net)` unscales the parameter
    gradients in the flat buffer (one launch per run, csrc/train.hip) and skips the optimizer step when one of them is not finite,
    `update()` backs the scale off after a skipped step and grows it after `growth_interval` clean ones."""

    def __init__(self, init_scale: float = 65536.0, growth_factor: float = 2.0, backoff_factor: float = 0.5, growth_interval: int = 2000, enabled: bool = True):
        self._scale, self.growth_factor, self.backoff_factor, self.growth_interval, self.enabled = float(init_scale), growth_factor, backoff_factor, growth_interval, enabled
        self._good_steps = 0
        self._found_inf = False
        self._flag = None
        self.skipped_steps = 0

    def get_scale(self) -> float:
        return self._scale if self.enabled else 1.0

    def scale(self, *tensors):
        if not self.enabled:
            return tensors if len(tensors) > 1 else tensors[0]
        out = tuple(t * self._scale for t in tensors)
        return out if len(out) > 1 else out[0]

    def step(self, optimizer, net):
        if not self.enabled:
            optimizer.step()
            return
        eng, flat, runs = net._last_grad_runs
        if self._flag is None or self._flag.device != flat.device:
            self._flag = torch.zeros(1, dtype=torch.int32, device=flat.device)
        self._flag.zero_()
        for lo, hi in runs:
            eng.train_unscale(flat[lo:hi], 1.0 / self._scale, self._flag)
        self._found_inf = bool(self._flag.item())
        if self._found_inf:
            self.skipped_steps += 1
        else:
            optimizer.step()

    def update(self):
        if not self.enabled:
            return
        if self._found_inf:
            self._scale *= self.backoff_factor
            self._good_steps = 0
        else:
            self._good_steps += 1
            if self._good_steps >= self.growth_interval:
                self._scale *= self.growth_factor
                self._good_steps = 0
