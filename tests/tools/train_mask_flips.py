"""How many ReLU masks of the train-mode forward differ from the float64 oracle's -- for the HIP f32 forward and for torch's own f32
forward.  A flipped mask (a pre-activation within rounding of zero) is an O(1) local difference in the gradient: it, not kernel
arithmetic, sets the ~1e-3 floor of the gradient comparison in tests/test_train_step_gpu.py.

    python tests/tools/train_mask_flips.py [B]
"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from oracle import soccdpt_ref as R
from soccdpt_amd.lib import PREC_F32
from soccdpt_amd.model.SOccDPT import SOccDPT_V3
from soccdpt_amd.utils.synth import synth_input, synth_state_dict, write_synth_calib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
m = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=False, precision=PREC_F32)
m.drop_path_rate = 0.0   # deterministic comparisons: no stochastic depth
sd = synth_state_dict(alias_pretrained=True)
m.load_state_dict(sd, strict=False)
m = m.to(dev).train()
m.seg_head[3].p = 0.0
x = synth_input(B, seed0=3)
m.train_forward(x.to(dev))
torch.cuda.synchronize()
eng = m._engine(dev)

def oracle(dt):
    s = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in sd.items()}
    with torch.no_grad():
        layers = R.swin_encoder(s, x.to(dt), R.ARCHS["swin2t16_256"])
        inv, p1 = R.dpt_decoder(s, layers)
        c = F.conv2d(p1, s["seg_head.0.weight"], padding=1)
        mu, var = c.mean((0, 2, 3), keepdim=True), c.var((0, 2, 3), unbiased=False, keepdim=True)
        z = (c - mu) / torch.sqrt(var + 1e-5) * s["seg_head.1.weight"].view(1, -1, 1, 1) + s["seg_head.1.bias"].view(1, -1, 1, 1)
        h = F.conv2d(p1, s["depth_net.scratch.output_conv.0.weight"], s["depth_net.scratch.output_conv.0.bias"], padding=1)
        h = F.interpolate(h, scale_factor=2, mode="bilinear", align_corners=True)
        e = F.conv2d(h, s["depth_net.scratch.output_conv.2.weight"], s["depth_net.scratch.output_conv.2.bias"], padding=1)
    return {"seg BatchNorm output (ReLU input)": z.permute(0, 2, 3, 1).reshape(-1, 256), "depth output_conv.2 (ReLU input)": e.permute(0, 2, 3, 1).reshape(-1, 32)}

o64, o32 = oracle(torch.float64), oracle(torch.float32)
hip = {"seg BatchNorm output (ReLU input)": None, "depth output_conv.2 (ReLU input)": eng.train_tensor(B, "depth_conv2", 32).cpu()}
# the HIP path stores the seg activation after the ReLU: its mask is (value > 0)
seg_act = eng.train_tensor(B, "seg_act", 256).cpu()
for k in o64:
    t64 = o64[k] > 0
    t32 = o32[k] > 0
    th = (seg_act > 0) if hip[k] is None else (hip[k] > 0)
    n = t64.numel()
    print(f"{k}: {n} elements; masks differing from float64: torch f32 {int((t32 != t64).sum())}, HIP f32 {int((th != t64).sum())}; "
          f"|pre-activation| < 1e-5 * rms: {int((o64[k].abs() < 1e-5 * o64[k].pow(2).mean().sqrt()).sum())}")
    if hip[k] is not None:
        print(f"   forward rel L2 vs float64: torch f32 {float((o32[k].double() - o64[k]).norm() / o64[k].norm()):.2e}, HIP {float((hip[k].double() - o64[k]).norm() / o64[k].norm()):.2e}")
c64 = None
