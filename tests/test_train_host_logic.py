"""CPU: the host side of the training step (soccdpt_amd/model/SOccDPT.py: _bind_for_training / train_forward / backward) against a stand-in engine
that records what it is asked to do -- no kernels involved.  Covers: one flat gradient buffer with an aligned view per consumed tensor; only
trainable tensors get a gradient bound, and re-binding follows requires_grad changes (PatchWiseInplace, freeze helpers); .grad assignment and
autograd-style accumulation; the data-parallel exchange sees contiguous runs of this step's trainable tensors."""
import os
import tempfile

import pytest
import torch

from soccdpt_amd.lib import PREC_BF16, PREC_F32


class FakeEngine:
    def __init__(self, keys, device):
        self._keys, self.device = list(keys), torch.device(device)
        self.bound, self.grads, self.amp, self.calls = {}, {}, None, []

    def weight_keys(self):
        return self._keys

    def bind(self, key, t):
        self.bound[key] = t
        return True

    def bind_grad(self, key, g):
        self.grads[key] = g

    def train_set_amp(self, on):
        self.amp = bool(on)

    def train_set_drop_path(self, rate):
        self.drop_path = float(rate)

    def train_forward(self, x, inv, seg, dropout_p=0.1, seed=0):
        self.calls.append(("fwd", float(dropout_p), int(seed)))
        inv.fill_(1.0)
        seg.fill_(0.5)

    def train_backward(self, x, d_inv, d_seg):
        self.calls.append(("bwd",))
        for i, (k, g) in enumerate(self.grads.items()):
            if g is not None:
                g.fill_(float(i + 1))          # the library WRITES (does not accumulate)


@pytest.fixture()
def net_and_engine(tmp_path):
    from soccdpt_amd.model.SOccDPT import SOccDPT_V3
    from soccdpt_amd.utils.synth import synth_state_dict, write_synth_calib
    calib = write_synth_calib(os.path.join(str(tmp_path), "calib.yaml"))
    net = SOccDPT_V3(sigmoid=True, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=False, precision=PREC_F32)
    sd = synth_state_dict(alias_pretrained=True)
    net.load_state_dict(sd, strict=False)
    net.train()
    keys = [k for k in sd if not k.startswith("pretrained.") and "num_batches_tracked" not in k and "relative_position" not in k
            and "attn_mask" not in k and k in dict(net.named_parameters(remove_duplicate=False)) | dict(net.named_buffers(remove_duplicate=False))]
    keys = [k for k in keys if "model.norm." not in k and "model.head." not in k and "refinenet4.resConfUnit1" not in k]
    eng = FakeEngine(keys, "cpu")
    net._engine = lambda device: eng
    return net, eng


def test_flat_buffer_views_and_rebinding(net_and_engine):
    net, eng = net_and_engine
    x = torch.zeros(1, 3, 256, 256)
    for p in net.parameters():
        p.requires_grad_(True)
    inv, seg = net.train_forward(x, seed=7)
    assert tuple(inv.shape) == (1, 256, 256) and tuple(seg.shape) == (1, 3, 256, 256) and eng.calls[-1] == ("fwd", 0.1, 7) and eng.amp is False
    assert int(net.seg_head[1].num_batches_tracked) == 1
    st = net._train_state[id(eng)]
    flat = st["flat"]
    params = dict(net.named_parameters())
    prev_hi = 0
    for k in eng.weight_keys():
        lo, hi = st["span"][k]
        live = dict(net.named_parameters(remove_duplicate=False)) | dict(net.named_buffers(remove_duplicate=False))
        assert lo % 64 == 0 and lo >= prev_hi and hi - lo == live[k].numel()
        prev_hi = hi
        g = eng.grads[k]
        if k in params:      # parameters get a view of the flat buffer, buffers (running statistics) nothing
            assert g is not None and g.data_ptr() == flat.data_ptr() + 4 * lo and g.shape == live[k].shape
        else:
            assert g is None
    # freeze the encoder: its gradients are unbound on the next step, the others keep their views
    for k, p in net.named_parameters():
        p.requires_grad_("pretrained" not in k)
    net.train_forward(x)
    assert all((eng.grads[k] is None) == ("pretrained" in k or k not in params) for k in eng.weight_keys())
    net.train_amp = True
    net.train_forward(x)
    assert eng.amp is True


def test_grad_assignment_accumulation_and_exchange_runs(net_and_engine):
    net, eng = net_and_engine
    x = torch.zeros(1, 3, 256, 256)
    names = [k for k, _ in net.named_parameters()]
    trainable = set(k for k in names if k.startswith("seg_head.") or "refinenet1." in k)
    for k, p in net.named_parameters():
        p.requires_grad_(k in trainable)
    seen_runs = []

    class Exchange:
        def __call__(self, flat, runs):
            seen_runs.append([list(r) for r in runs])

        def average_buffers(self, tensors):
            pass

    net.grad_exchange = Exchange()
    net.train_forward(x)
    net.backward(torch.zeros(1, 256, 256), torch.zeros(1, 3, 256, 256))
    first = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    assert set(first) == {k for k in trainable if k in eng.weight_keys()}
    # a second backward without zero_grad accumulates like autograd, although the engine overwrote its buffers
    net.train_forward(x)
    net.backward(torch.zeros(1, 256, 256), torch.zeros(1, 3, 256, 256))
    for k, p in net.named_parameters():
        if p.grad is not None:
            assert torch.equal(p.grad, 2 * first[k]), k
    # the exchange saw contiguous runs covering exactly the trainable tensors: refinenet1.* is one run, seg_head.* two (the BatchNorm running
    # buffers sit between seg_head.1.bias and seg_head.4.weight in the key order and are not gradients)
    st = net._train_state[id(eng)]
    runs = seen_runs[-1]
    assert len(runs) == 3
    covered = lambda k: any(lo <= st["span"][k][0] and st["span"][k][1] <= hi for lo, hi in runs)
    for k in eng.weight_keys():
        if k in dict(net.named_parameters()):
            assert covered(k) == (k in trainable), k
    with pytest.raises(RuntimeError, match="train_forward"):
        net.backward(torch.zeros(1, 256, 256), torch.zeros(1, 3, 256, 256))      # one backward per forward


def test_train_forward_refuses_other_precisions(tmp_path):
    from soccdpt_amd.model.SOccDPT import SOccDPT_V3
    from soccdpt_amd.utils.synth import write_synth_calib
    calib = write_synth_calib(os.path.join(str(tmp_path), "calib.yaml"))
    net = SOccDPT_V3(sigmoid=True, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=False, precision=PREC_BF16)
    with pytest.raises(RuntimeError, match="PREC_F32"):
        net.train_forward(torch.zeros(1, 3, 256, 256))
