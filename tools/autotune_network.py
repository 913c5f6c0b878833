"""In-network tile tuning: for every distinct GEMM / conv shape of the forward, time each candidate tile configuration INSIDE the
launch sequence (weights and activations arrive as cold as they do in production) with the library's HIP-event profiler, and
print the per-shape ranking against the heuristic's choice.  A warm repeated-launch benchmark (tools/igemm_tune.py) mis-ranks
configurations that re-read weights more often.

    python tools/autotune_network.py [model_type] [batch] [precision]
"""
import os, sys, tempfile, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from soccdpt_amd.model.SOccDPT import SOccDPT_V3
from soccdpt_amd.model.spec import MODEL_TYPE_TO_BACKBONE, backbone_image_size
from soccdpt_amd.utils.synth import synth_input, synth_state_dict, write_synth_calib

model_type = sys.argv[1] if len(sys.argv) > 1 else "dpt_swin2_tiny_256"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
pname = sys.argv[3] if len(sys.argv) > 3 else "bf16"
prec = {"bf16": 0, "f16": 2, "f16x3": 3}[pname]
dev = torch.device("cuda:0")
backbone = MODEL_TYPE_TO_BACKBONE[model_type]
img = backbone_image_size(backbone)
calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
net = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True, model_type=model_type, precision=prec)
net.load_state_dict(synth_state_dict(backbone, alias_pretrained=True), strict=False)
net = net.eval().to(dev)
x = synth_input(B, size=img, seed0=0).to(dev)
eng = net._engine(dev)
REPS = 12

def measure():
    """ms per forward of every site, averaged over REPS forwards."""
    for _ in range(3):
        net(x)
    eng.profile_enable(True)
    for _ in range(REPS):
        net(x)
    torch.cuda.synchronize()
    st = eng.profile_collect()
    eng.profile_enable(False)
    return {k: v["ms"] / REPS * 1e3 for k, v in st.items() if k.startswith("site")}   # us per forward

eng.profile_sites(True)
base = measure()
sites = eng.sites()
total_base = sum(base.values())
print(f"{model_type} B={B}: {len(sites)} distinct igemm shapes, {sum(s['launches'] for s in sites) // (REPS + 0)} launches profiled, "
      f"{total_base:.0f} us of igemm per forward with the heuristic")
K64 = [2, 1, 13, 10, 14, 8, 6, 20, 21, 22, 23] + ([40, 41, 42, 43, 45] if os.environ.get('AUTOTUNE_M32') == '1' else [])
K32 = [4, 9, 19, 15, 16, 24] + [int(c) for c in os.environ.get('AUTOTUNE_EXTRA', '').split(',') if c]
results = []
for s in sites:
    Cin = s["K"] // s["taps"]
    cands = [c for c in (K64 if Cin % 64 == 0 else []) + K32 if not (c in (20, 22) and Cin % 128)]
    if pname == "f16x3":   # the x3 tile set (igemm.hip kCfgNamesX3)
        x3c = [int(c) for c in os.environ.get('AUTOTUNE_X3_CANDS', '0,1,3,4,5,6,7,8,9,10,11,12').split(',')]
        cands = [c for c in x3c if not (c in (8, 9, 10) and Cin % 64)]
    if s["N"] <= 32 or s["M"] < int(os.environ.get('AUTOTUNE_MIN_M', '0')):
        continue
    row = {"auto": (s["cfg"], base[s["site"]])}
    for c in cands:
        if c == s["cfg"]:
            continue
        eng.tune_set(s["M"], s["N"], s["K"], s["taps"], c)
        try:
            t = measure()[s["site"]]
        except RuntimeError as e:
            t = float("inf")
        eng.tune_set(s["M"], s["N"], s["K"], s["taps"], -1)
        row[c] = t
    best = min(((k, v) for k, v in row.items() if k != "auto"), key=lambda kv: kv[1], default=("auto", row["auto"][1]))
    gain = row["auto"][1] - best[1]
    results.append(dict(shape=s, auto_cfg=s["cfg"], auto_us=row["auto"][1], best_cfg=best[0], best_us=best[1], gain_us=gain))
    alts = " ".join(f"{k}:{v:.1f}" for k, v in row.items() if k != "auto")
    print(f"{s['site']} M={s['M']:6d} N={s['N']:5d} K={s['K']:5d} taps={s['taps']} x{s['launches'] // (REPS * (len(cands) + 1)) or 1}: auto cfg {s['cfg']:2d} {row['auto'][1]:7.1f} us | best {best[0]} {best[1]:7.1f} | {alts}", flush=True)
gain = sum(r["gain_us"] for r in results if r["gain_us"] > 0)
print(f"sum of per-shape gains over the heuristic: {gain:.0f} us per forward")
json.dump(results, open(f"gpurun_out/autotune_{model_type}_B{B}_{pname}.json", "w"), indent=1)
