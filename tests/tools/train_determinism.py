"""Six optimisation steps on one synthetic batch; prints the SHA-1 of the resulting weights / buffers (bit-reproducibility across processes).
AMP=bf16|f16 selects the 16-bit operand mode (no GradScaler here: the hash, not the trajectory, is the point)."""
import os, sys, tempfile, hashlib, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from soccdpt_amd.lib import PREC_F32
from soccdpt_amd.model.SOccDPT import SOccDPT_V3
from soccdpt_amd.scripts.train_SOccDPT import SyntheticDepthSegDataset, get_batch
from soccdpt_amd.utils.loss import training_loss
from soccdpt_amd.utils.optim import Adam
from soccdpt_amd.utils.synth import synth_state_dict, write_synth_calib
dev = torch.device("cuda:0")
calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "c.yaml"))
net = SOccDPT_V3(sigmoid=True, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=False, precision=PREC_F32)
net.drop_path_rate = 0.0   # deterministic comparisons: no stochastic depth
net.load_state_dict(synth_state_dict(alias_pretrained=True), strict=False)
net = net.to(dev).train(); net.train_amp = os.environ.get("AMP", "") or False
x, _, md, yd, ms, ys = get_batch(SyntheticDepthSegDataset(4, 256), 4, 4)
x = x.to(dev, torch.float32); yd, ys = yd.to(dev, torch.float32), ys.to(dev, torch.float32); md, ms = md.to(dev, torch.bool), ms.to(dev, torch.bool)
opt = Adam(net.parameters(), lr=3e-5)
for step in range(6):
    inv, seg = net.train_forward(x, seed=step)
    out = training_loss(inv, seg, yd, md, ys, ms, 0.5, 0.5, compute_scale_and_shift=True)
    opt.zero_grad(set_to_none=True)
    net.backward(out["d_inv"], out["d_seg"])
    opt.step()
torch.cuda.synchronize()
h = hashlib.sha1()
for k, v in net.state_dict().items():
    if k.startswith("pretrained."): continue
    h.update(v.detach().cpu().contiguous().numpy().tobytes())
print("HASH", os.environ.get("AMP", "f32"), h.hexdigest(), float(out["loss"]))
