"""Per-parameter gradient report of the HIP training step against torch autograd over the CPU oracle (the same comparison as
tests/test_train_step_gpu.py, printed in network order so that the first wrong gradient walking backwards locates a bug).

    python tests/tools/train_grad_check.py [B] [sigmoid 0|1] [model_type] [random|criterion]
"criterion": the upstream gradients come from the training criterion (HIP soccdpt_training_loss on the GPU side, oracle/loss_ref.py +
autograd on the oracle side) on the synthetic camera-resolution targets, i.e. the whole optimisation step's gradient.
"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import soccdpt_ref as R
from soccdpt_amd.lib import PREC_F32
from soccdpt_amd.model.SOccDPT import SOccDPT_V3
from soccdpt_amd.model.spec import MODEL_TYPE_TO_BACKBONE, backbone_image_size
from soccdpt_amd.utils.synth import synth_input, synth_state_dict, write_synth_calib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sigmoid = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
model_type = sys.argv[3] if len(sys.argv) > 3 else "dpt_swin2_tiny_256"
backbone = MODEL_TYPE_TO_BACKBONE[model_type]
S = backbone_image_size(backbone)
mode = sys.argv[4] if len(sys.argv) > 4 else "random"
dev = torch.device("cuda:0")
calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
m = SOccDPT_V3(sigmoid=sigmoid, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=False, precision=PREC_F32, model_type=model_type)
m.drop_path_rate = 0.0   # deterministic comparisons: no stochastic depth
sd = synth_state_dict(backbone, alias_pretrained=True)
m.load_state_dict(sd, strict=False)
m = m.to(dev).train()
m.seg_head[3].p = 0.0
m.train_amp = os.environ.get("TRAIN_AMP", "0") == "1"     # bf16 operands for the gradient GEMMs
x = synth_input(B, size=S, seed0=3)
g = torch.Generator().manual_seed(11)
a = torch.randn((B, S, S), generator=g)
b = torch.randn((B, 3, S, S), generator=g)
if mode == "criterion":
    from oracle import loss_ref
    from soccdpt_amd.scripts.train_SOccDPT import SyntheticDepthSegDataset, get_batch
    from soccdpt_amd.utils.loss import training_loss
    _, _, mask_disp, y_disp, mask_seg, y_seg = get_batch(SyntheticDepthSegDataset(B, S), B, B)
    y_disp, y_seg = y_disp.float(), y_seg.float()

def oracle(dt):
    sd_o = {k: (v.clone().to(dt).requires_grad_(True) if v.is_floating_point() and "running_" not in k else (v.clone().to(dt) if v.is_floating_point() else v.clone()))
            for k, v in sd.items()}
    o_inv, o_seg, _ = R.soccdpt_v3_network(sd_o, x.to(dt), backbone=backbone, sigmoid=sigmoid, training=True)
    if mode == "criterion":
        loss_ref.training_loss(o_inv, o_seg, y_disp.to(dt), mask_disp, y_seg.to(dt), mask_seg, 0.5, 0.5, True)[0].backward()
    else:
        ((o_inv * a.to(dt)).sum() + (o_seg * b.to(dt)).sum()).backward()
    return sd_o, o_inv, o_seg
t0 = time.time()
sd_o, o_inv, o_seg = oracle(torch.float32)
print(f"oracle forward + autograd backward on the CPU: {time.time() - t0:.1f} s", flush=True)
sd_64, _, _ = oracle(torch.float64)     # the exact gradient: how far torch's own f32 autograd is from it is the noise floor of the comparison
inv, seg = m.train_forward(x.to(dev))
torch.cuda.synchronize()
rel = lambda p, q: float((p - q).norm() / q.norm().clamp_min(1e-30))
print("forward: inv", rel(inv.cpu(), o_inv.detach()), "seg", rel(seg.cpu(), o_seg.detach()), flush=True)
def upstream(inv, seg):
    if mode == "criterion":
        r = training_loss(inv, seg, y_disp.to(dev), mask_disp.to(dev), y_seg.to(dev), mask_seg.to(dev), 0.5, 0.5, compute_scale_and_shift=True)
        return r["d_inv"], r["d_seg"]
    return a.to(dev), b.to(dev)
m.backward(*upstream(inv, seg))
torch.cuda.synchronize()
for _ in range(2):
    for p in m.parameters():
        p.grad = None
    t0 = time.time()
    inv, seg = m.train_forward(x.to(dev))
    torch.cuda.synchronize()
    t1 = time.time()
    m.backward(*upstream(inv, seg))
    torch.cuda.synchronize()
    print(f"GPU: train_forward {1e3 * (t1 - t0):.1f} ms, backward {1e3 * (time.time() - t1):.1f} ms (B={B})", flush=True)
bad = 0
for k, p in m.named_parameters():
    ref = sd_o[k].grad
    got = p.grad.cpu() if p.grad is not None else None
    if ref is None:
        print("unused   ", k, "" if got is None else "  <-- has a gradient"); continue
    if got is None:
        print("MISSING", k); bad += 1; continue
    e = rel(got.double(), sd_64[k].grad)
    e32 = rel(ref.double(), sd_64[k].grad)
    flag = "" if e < 3 * e32 + 1e-5 else "   <-- BAD"
    bad += not (e < 3 * e32 + 1e-5)
    print(f"{e:9.2e} (torch f32 autograd {e32:9.2e}) vs_f32 {rel(got, ref):9.2e}  |g| {float(ref.norm()):9.2e}  {k}{flag}")
print("bad:", bad)
