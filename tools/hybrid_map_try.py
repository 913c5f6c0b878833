"""dpt_hybrid_384: candidate precision maps around round 4's (ResNetV2 stages in x3) with the round-5 formats and groups: errors against the library's f32
mode (B = 4, the seven quantities) and frames/s, one process.   python tools/hybrid_map_try.py > gpurun_out/hybrid_map_try.txt"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from soccdpt_amd.lib import PREC_F16, PREC_F16X2W, PREC_F16X3, PREC_F32, PREC_MIXED
from soccdpt_amd.model.SOccDPT import SOccDPT_V3
from soccdpt_amd.utils.synth import synth_input, synth_state_dict, write_synth_calib

dev = torch.device("cuda:0")
calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
sd = synth_state_dict("vitb_rn50_384", alias_pretrained=True)
QUANT = ["feat0", "feat1", "feat2", "feat3", "path1", "inv", "seg_logits"]


def build(prec):
    n = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True, model_type="dpt_hybrid_384", precision=prec)
    n.load_state_dict(sd, strict=False)
    return n.eval().to(dev)


def quantities(net, x):
    inv, _ = net.network(x)
    e = net._engine(dev)
    q = {k: e.workspace_tensor(x.shape[0], k).double() for k in QUANT if k != "inv"}
    q["inv"] = inv.double()
    return q


x = synth_input(4, size=384, seed0=0).to(dev)
ref = quantities(build(PREC_F32), x)
net = build(PREC_MIXED)
eng = net._engine(dev)
X3, X2, F16 = PREC_F16X3, PREC_F16X2W, PREC_F16
R4 = [("rn.s0.*", X3), ("rn.s1.*", X3), ("rn.s2.*", X3), ("ro1", X3), ("oc0", X3), ("oc1", X3), ("oc2", X3), ("oc3", X3), ("head.s1", X3)]
cands = {
    "round-4 map": R4,
    "c2 of every stage x2w": R4 + [("rn.s0.c2", X2), ("rn.s1.c2", X2), ("rn.s2.c2", X2)],
    "c1 + c2 x2w": R4 + [(f"rn.s{s}.c{c}", X2) for s in range(3) for c in (1, 2)],
    "all rn x2w": R4 + [(f"rn.s{s}.*", X2) for s in range(3)],
    "all rn x2w, oc x2w": R4 + [(f"rn.s{s}.*", X2) for s in range(3)] + [(f"oc{l}", X2) for l in range(4)],
    "s2 x2w, s0 s1 x3": R4 + [("rn.s2.*", X2)],
    "s2.c2 x2w only": R4 + [("rn.s2.c2", X2)],
    "s2.c1 s2.c2 x2w": R4 + [("rn.s2.c1", X2), ("rn.s2.c2", X2)],
    "s0.c2 s2.c1 s2.c2 s1.c1 x2w": R4 + [("rn.s0.c2", X2), ("rn.s2.c1", X2), ("rn.s2.c2", X2), ("rn.s1.c1", X2)],
}
for name, m in cands.items():
    eng.prec_map_set("*", F16)
    for g, f in m:
        eng.prec_map_set(g, f)
    q = quantities(net, x)
    e = {k: float((q[k] - ref[k]).norm() / ref[k].norm()) for k in QUANT}
    ts = []
    for _ in range(3):
        for _ in range(10):
            net(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            net(x)
        torch.cuda.synchronize()
        ts.append(4 * 100 / (time.perf_counter() - t0))
    print(f"{name:34s} worst {max(e.values()):.2e}  frames/s {sorted(ts)[1]:.1f}   " + " ".join(f"{k} {v:.1e}" for k, v in e.items()), flush=True)
