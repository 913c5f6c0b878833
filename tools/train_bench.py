"""Training-step timing (BASELINE configs[4]): train-mode forward + criterion + backward + fused Adam on synthetic data, exact f32.

    python tools/train_bench.py [B] [steps] [encoder_percentage] [patchwise_percentage] [model_type]
Prints ms per optimisation step (one PatchWiseInplace patch = one forward + backward + Adam step, as scripts/train_SOccDPT.py runs them).
"""
import os, sys, tempfile, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from soccdpt_amd.lib import PREC_F32
from soccdpt_amd.loss import freeze_pretrained_encoder, unfreeze_pretrained_encoder_by_percentage
from soccdpt_amd.model.SOccDPT import SOccDPT_V3
from soccdpt_amd.scripts.train_SOccDPT import SyntheticDepthSegDataset, get_batch
from soccdpt_amd.utils.loss import training_loss
from soccdpt_amd.utils.optim import Adam, PatchWiseInplace
from soccdpt_amd.utils.synth import synth_state_dict, write_synth_calib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
enc_pct = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
patch_pct = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
model_type = sys.argv[5] if len(sys.argv) > 5 else "dpt_swin2_tiny_256"
from soccdpt_amd.model.spec import MODEL_TYPE_TO_BACKBONE, backbone_image_size
backbone = MODEL_TYPE_TO_BACKBONE[model_type]
S = backbone_image_size(backbone)
dev = torch.device("cuda:0")
calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
net = SOccDPT_V3(sigmoid=True, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=False, precision=PREC_F32, model_type=model_type)
net.load_state_dict(synth_state_dict(backbone, alias_pretrained=True), strict=False)
net = net.to(dev).train()
_amp = os.environ.get("TRAIN_AMP", "0")
net.train_amp = {"0": False, "1": True}.get(_amp, _amp)   # "bf16" | "f16" | "x3"
freeze_pretrained_encoder(net)
unfreeze_pretrained_encoder_by_percentage(net, enc_pct)
ds = SyntheticDepthSegDataset(B, S)
x, _, mask_disp, y_disp, mask_seg, y_seg = get_batch(ds, B, B)
x = x.to(dev, torch.float32)
y_disp, y_seg = y_disp.to(dev, torch.float32), y_seg.to(dev, torch.float32)
mask_disp, mask_seg = mask_disp.to(dev, torch.bool), mask_seg.to(dev, torch.bool)
opt = Adam(net.parameters(), lr=1e-5)

def one_batch():
    n = 0
    for net_patch in PatchWiseInplace(net, patch_pct):
        inv, seg = net_patch.train_forward(x, seed=n)
        out = training_loss(inv, seg, y_disp, mask_disp, y_seg, mask_seg, 0.5, 0.5, compute_scale_and_shift=True)
        opt.zero_grad(set_to_none=True)
        net_patch.backward(out["d_inv"], out["d_seg"])
        opt.step()
        n += 1
    return n, out

for _ in range(2):
    one_batch()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
t0 = time.time()
n = 0
for _ in range(steps):
    k, out = one_batch()
    n += k
torch.cuda.synchronize()
dt = time.time() - t0
# split of one step
ev[0].record(); inv, seg = net.train_forward(x, seed=0); ev[1].record()
out = training_loss(inv, seg, y_disp, mask_disp, y_seg, mask_seg, 0.5, 0.5, compute_scale_and_shift=True); ev[2].record()
net.backward(out["d_inv"], out["d_seg"]); ev[3].record()
torch.cuda.synchronize()
cpu = None   # the CPU baseline of the step is bench.py --train-step's cpu_baseline leg (only tests/, smoke() and bench.py may use oracle/)
print(json.dumps({"cpu_baseline": cpu, "model_type": model_type, "amp": net.train_amp, "B": B, "encoder_percentage": enc_pct, "patchwise_percentage": patch_pct, "optimisation_steps": n,
                  "ms_per_step": round(1e3 * dt / n, 2), "samples_per_s": round(B * n / dt, 1),
                  "train_forward_ms": round(ev[0].elapsed_time(ev[1]), 2), "criterion_ms": round(ev[1].elapsed_time(ev[2]), 2),
                  "backward_all_trainable_ms": round(ev[2].elapsed_time(ev[3]), 2), "loss": float(out["loss"]),
                  "train_workspace_gib": round(net._engine(dev).train_workspace(B).numel() / 2 ** 30, 2)}))
