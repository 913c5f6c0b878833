"""Per-site device time of every GEMM / conv launch shape of one forward, for one precision mode, plus the non-GEMM families:
the cost side of the precision map (tools/precision_map.py).

    python tools/site_times.py [model_type] [batch] [precision] [x3 groups: "empty" | "a,b,c"] > gpurun_out/site_times_<model>_<precision>.json
"""
import os, sys, tempfile, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from soccdpt_amd.model.SOccDPT import SOccDPT_V3
from soccdpt_amd.model.spec import MODEL_TYPE_TO_BACKBONE, backbone_image_size
from soccdpt_amd.utils.synth import synth_input, synth_state_dict, write_synth_calib

model_type = sys.argv[1] if len(sys.argv) > 1 else "dpt_swin2_tiny_256"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
pname = sys.argv[3] if len(sys.argv) > 3 else "f16"
prec = {"bf16": 0, "f32": 1, "f16": 2, "f16x3": 3, "mixed": 4}[pname]
dev = torch.device("cuda:0")
backbone = MODEL_TYPE_TO_BACKBONE[model_type]
img = backbone_image_size(backbone)
calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
net = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True, model_type=model_type, precision=prec)
net.load_state_dict(synth_state_dict(backbone, alias_pretrained=True), strict=False)
net = net.eval().to(dev)
x = synth_input(B, size=img, seed0=0).to(dev)
eng = net._engine(dev)
if len(sys.argv) > 4 and prec == 4:   # override the shipped precision map
    eng.prec_map_set("*", 2)
    for g in ([] if sys.argv[4] == "empty" else sys.argv[4].split(",")):
        eng.prec_map_set(g, 3)
REPS = 20
for _ in range(5):
    net(x)
eng.profile_sites(True)
eng.profile_enable(True)
for _ in range(REPS):
    net(x)
torch.cuda.synchronize()
st = eng.profile_collect()
eng.profile_enable(False)
sites = {}
for s in eng.sites():   # the x3 launches of a shape in the mixed mode are their own site, named "siteNNNx"
    sites[s["site"]] = sites[s["site"] + "x"] = s
rows = []
for k, v in st.items():
    r = {"name": k, "us_per_forward": v["ms"] / REPS * 1e3, "launches": v.get("launches", 0) / REPS}
    if k in sites:
        s = sites[k]
        r.update(M=s["M"], N=s["N"], K=s["K"], taps=s["taps"], cfg=s["cfg"])
    rows.append(r)
print(json.dumps({"model": model_type, "B": B, "precision": pname, "rows": rows}, indent=1))
