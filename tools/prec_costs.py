"""Per-group promotion costs for soccdpt_prec_calibrate's compiled-in table (csrc/prec_cost_table.h, tools/gen_prec_cost_table.py): the device time
(sum of the kernels' own durations per forward, dispatch-bound event pairs) of the forward with ONE group in x3 / in x2w and every other group in
fp16, minus the all-fp16 forward, at the model's BASELINE batch.
    python tools/prec_costs.py [model_type] [batch] > gpurun_out/r05_prec_costs_<model>.json"""
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from soccdpt_amd.lib import PREC_F16, PREC_F16X2W, PREC_F16X3, PREC_MIXED
from soccdpt_amd.model.SOccDPT import SOccDPT_V3
from soccdpt_amd.model.spec import MODEL_TYPE_TO_BACKBONE, backbone_image_size
from soccdpt_amd.utils.synth import synth_input, synth_state_dict, write_synth_calib

model_type = sys.argv[1] if len(sys.argv) > 1 else "dpt_swin2_tiny_256"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
STEPS = int(os.environ.get("COST_STEPS", "30"))
dev = torch.device("cuda:0")
backbone = MODEL_TYPE_TO_BACKBONE[model_type]
calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
net = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True, model_type=model_type, precision=PREC_MIXED)
net.load_state_dict(synth_state_dict(backbone, alias_pretrained=True), strict=False)
net = net.eval().to(dev)
x = synth_input(B, size=backbone_image_size(backbone), seed0=0).to(dev)
eng = net._engine(dev)
net(x)
groups = list(eng.prec_map())


def device_us():
    for _ in range(3):
        net(x)
    eng.profile_enable(True)
    for _ in range(STEPS):
        net(x)
    torch.cuda.synchronize()
    st = eng.profile_collect()
    eng.profile_enable(False)
    return sum(v["ms"] for v in st.values()) / STEPS * 1e3


eng.prec_map_set("*", PREC_F16)
base = [device_us() for _ in range(3)]
base_us = sorted(base)[1]
out = {"model": model_type, "B": B, "all_fp16_us": base_us, "all_fp16_repeats": base, "groups": {}}
for g in groups:
    row = {}
    for name, fmt in (("x3", PREC_F16X3), ("x2w", PREC_F16X2W)):
        eng.prec_map_set("*", PREC_F16)
        try:
            eng.prec_map_set(g, fmt)
        except RuntimeError:
            row[name] = None
            continue
        row[name] = device_us() - base_us
    out["groups"][g] = row
    print(f"{g:14s} x3 {row['x3']:+7.1f} us   x2w " + (f"{row['x2w']:+7.1f} us" if row["x2w"] is not None else "   n/a"), file=sys.stderr, flush=True)
eng.prec_map_set("*", PREC_F16)
out["all_fp16_us_after"] = device_us()
print(json.dumps(out, indent=1))
