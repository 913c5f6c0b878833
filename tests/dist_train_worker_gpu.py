"""Worker of tests/test_dist_gpu.py::test_data_parallel_training_rehearsal: one torchrun rank (gloo rehearsal, ranks share the GPU).  Two
optimisation steps of data-parallel training on a per-rank batch shard; prints the SHA-1 of a few weights after the steps and the gradient
check against the mean of the ranks' local gradients."""
import hashlib
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from soccdpt_amd import dist as sdist  # noqa: E402
from soccdpt_amd.lib import PREC_F32  # noqa: E402
from soccdpt_amd.model.SOccDPT import SOccDPT_V3  # noqa: E402
from soccdpt_amd.scripts.train_SOccDPT import SyntheticDepthSegDataset, get_batch  # noqa: E402
from soccdpt_amd.utils.loss import training_loss  # noqa: E402
from soccdpt_amd.utils.optim import Adam  # noqa: E402
from soccdpt_amd.utils.synth import synth_state_dict, write_synth_calib  # noqa: E402

rank, local, world = sdist.init_from_env("nccl")
dev = torch.device("cuda", local)
torch.cuda.set_device(dev)
calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
net = SOccDPT_V3(sigmoid=True, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=False, precision=PREC_F32)
net.drop_path_rate = 0.0   # deterministic comparisons: no stochastic depth
net.load_state_dict(synth_state_dict(alias_pretrained=True), strict=False)
net = net.to(dev).train()
net.seg_head[3].p = 0.0
ds = SyntheticDepthSegDataset(world, 256)
lo, hi = sdist.shard_range(world, rank, world)          # one sample per rank
x, _, mask_disp, y_disp, mask_seg, y_seg = get_batch(ds, hi, hi - lo)
x = x.to(dev, torch.float32)
y_disp, y_seg = y_disp.to(dev, torch.float32), y_seg.to(dev, torch.float32)
mask_disp, mask_seg = mask_disp.to(dev, torch.bool), mask_seg.to(dev, torch.bool)

def step_grads():
    inv, seg = net.train_forward(x, seed=0)
    r = training_loss(inv, seg, y_disp, mask_disp, y_seg, mask_seg, 0.5, 0.5, compute_scale_and_shift=True)
    for p in net.parameters():
        p.grad = None
    net.backward(r["d_inv"], r["d_seg"])
    return float(r["loss"])

# 1. local gradients (no exchange), then the same step with the exchange attached: must equal the mean over the ranks
step_grads()
key = "depth_net.scratch.refinenet1.out_conv.weight"
params = dict(net.named_parameters())
local_g = params[key].grad.detach().cpu().clone()
rm0 = net.seg_head[1].running_mean.detach().cpu().clone()
sdist.attach_training(net)
assert net.grad_exchange is not None
# undo the running-buffer update of the first forward so that both ranks start the exchanged step from the same state
net.load_state_dict(synth_state_dict(alias_pretrained=True), strict=False)
step_grads()
torch.cuda.synchronize()
avg_g = params[key].grad.detach().cpu().clone()
gathered = [torch.empty_like(local_g) for _ in range(world)]
dist.all_gather(gathered, local_g)
mean_g = torch.stack(gathered).mean(0)
err = float((avg_g - mean_g).norm() / mean_g.norm())
# 2. two optimizer steps: replicas stay bit-identical
opt = Adam(net.parameters(), lr=1e-4)
for _ in range(2):
    step_grads()
    opt.step()
torch.cuda.synchronize()
h = hashlib.sha1()
for k in (key, "depth_net.pretrained.model.layers.0.blocks.0.attn.qkv.weight", "seg_head.4.bias", "seg_head.1.running_mean"):
    t = dict(net.state_dict())[k]
    h.update(t.detach().cpu().contiguous().numpy().tobytes())
print("RESULT", rank, f"{err:.3e}", h.hexdigest(), net.grad_exchange.calls, flush=True)
dist.barrier()
dist.destroy_process_group()
