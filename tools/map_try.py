"""Try variations of the shipped precision map of SOCCDPT_PREC_MIXED: errors of the seven quantities against the library's exact-f32 mode and the
step time, variants interleaved in one process.

    python tools/map_try.py <model_type> <batch> "<variant>" "<variant>" ...     variant: "+group,-group,..." relative to the shipped map ("" = shipped)
"""
import os, sys, tempfile, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from soccdpt_amd.lib import PREC_F16, PREC_F16X3, PREC_F32, PREC_MIXED
from soccdpt_amd.model.SOccDPT import SOccDPT_V3
from soccdpt_amd.model.spec import MODEL_TYPE_TO_BACKBONE, backbone_image_size
from soccdpt_amd.utils.synth import synth_input, synth_state_dict, write_synth_calib

model_type, B = sys.argv[1], int(sys.argv[2])
variants = sys.argv[3:] or [""]
QUANT = ["feat0", "feat1", "feat2", "feat3", "path1", "inv", "seg_logits"]
dev = torch.device("cuda:0")
backbone = MODEL_TYPE_TO_BACKBONE[model_type]
img = backbone_image_size(backbone)
calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
sd = synth_state_dict(backbone, alias_pretrained=True)

def build(prec):
    net = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True, model_type=model_type, precision=prec)
    net.load_state_dict(sd, strict=False)
    return net.eval().to(dev)

def quantities(net, x):
    inv, seg = net.network(x)
    eng = net._engine(dev)
    out = {q: eng.workspace_tensor(x.shape[0], q).double() for q in QUANT if q != "inv"}
    out["inv"] = inv.double()
    return out

xs = [synth_input(B, size=img, seed0=s).to(dev) for s in (0, 4, 100)]   # bench.py's batch, the tests' seeds, one more
f32 = build(PREC_F32)
refs = [quantities(f32, x) for x in xs]
del f32
net = build(PREC_MIXED)
eng = net._engine(dev)
shipped = sorted(g for g, f in eng.prec_map().items() if f == 3)

def apply(variant):
    eng.prec_map_set("*", PREC_F16)
    s = set(shipped)
    for t in filter(None, variant.split(",")):
        (s.add if t[0] == "+" else s.discard)(t[1:])
    for g in s:
        eng.prec_map_set(g, PREC_F16X3)
    return sorted(s)

res = {}
for v in variants:
    apply(v)
    worst = {}
    for x, r in zip(xs, refs):
        q = quantities(net, x)
        for k in QUANT:
            worst[k] = max(worst.get(k, 0.0), float((q[k] - r[k]).norm() / r[k].norm()))
    res[v] = dict(errors=worst, worst=max(worst.values()), ms=[])
for rep in range(4):
    for v in variants:
        apply(v)
        for _ in range(20):
            net(xs[0])
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(200):
            net(xs[0])
        torch.cuda.synchronize()
        res[v]["ms"].append((time.perf_counter() - t) / 200 * 1e3)
for v in variants:
    r = res[v]
    ms = sorted(r["ms"])[len(r["ms"]) // 2]
    print(f"{v or '(shipped)':60s} worst {r['worst']:.3e} ({max(r['errors'], key=r['errors'].get)})  {ms:.4f} ms  {B / ms * 1e3:.0f} frames/s   " +
          " ".join(f"{k}={e:.2e}" for k, e in r["errors"].items()), flush=True)
print(json.dumps(dict(model=model_type, B=B, shipped=shipped, results=res)))
