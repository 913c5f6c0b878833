"""In-network A/B of tile configurations for the big fp16 3x3 convolutions of the default (mixed) forward: per-site device time with each candidate
forced through soccdpt_tune_set, alternated rounds in one process.    python tools/bigconv_ab.py [model_type] [batch]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from soccdpt_amd.model.SOccDPT import SOccDPT_V3
from soccdpt_amd.model.spec import MODEL_TYPE_TO_BACKBONE, backbone_image_size
from soccdpt_amd.utils.synth import synth_input, synth_state_dict, write_synth_calib

model_type = sys.argv[1] if len(sys.argv) > 1 else "dpt_swin2_tiny_256"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
backbone = MODEL_TYPE_TO_BACKBONE[model_type]
img = backbone_image_size(backbone)
calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
net = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True, model_type=model_type)
net.load_state_dict(synth_state_dict(backbone, alias_pretrained=True), strict=False)
net = net.eval().to(dev)
x = synth_input(B, size=img, seed0=0).to(dev)
eng = net._engine(dev)
REPS = 20


def measure():
    for _ in range(3):
        net(x)
    eng.profile_enable(True)
    for _ in range(REPS):
        net(x)
    torch.cuda.synchronize()
    st = eng.profile_collect()
    eng.profile_enable(False)
    return {k: v["ms"] / REPS * 1e3 for k, v in st.items()}, sum(v["ms"] for v in st.values()) / REPS * 1e3


eng.profile_sites(True)
base, tot = measure()
sites = [s for s in eng.sites() if s["taps"] == 9 and s["M"] * s["N"] >= 4096 * 256 and s["site"] in base and not s["site"].endswith(("x", "w"))]
print(f"{model_type} B={B}: forward {tot:.0f} us of kernels; big fp16 3x3 sites:")
for s in sites:
    print(f"  {s['site']}: M={s['M']} N={s['N']} K={s['K']} launches/forward {s['launches'] // (REPS + 3)}  cfg {s['cfg']}  {base[s['site']]:.1f} us")
for s in sites:
    cands = [21, 1, 16, 6] + ([47, 51, 52] if s["N"] % 256 == 0 else [49, 50])
    res = {c: [] for c in cands}
    tots = {c: [] for c in cands}
    for rnd in range(3):
        for c in cands:
            eng.tune_clear()
            eng.tune_set(s["M"], s["N"], s["K"], s["taps"], c)
            try:
                t, tt = measure()
            except RuntimeError as e:
                res[c].append(float("nan")); tots[c].append(float("nan"))
                continue
            res[c].append(t.get(s["site"], float("nan")))
            tots[c].append(tt)
    eng.tune_clear()
    med = lambda v: sorted(v)[len(v) // 2]
    print(f"{s['site']} M={s['M']} N={s['N']}: " + "  ".join(f"cfg{c}: {med(res[c]):.1f} us (forward {med(tots[c]):.0f})" for c in cands), flush=True)
