import os, sys, tempfile
sys.path.insert(0, os.getcwd())
import torch
from soccdpt_amd.model.SOccDPT import SOccDPT_V3
from soccdpt_amd.utils.synth import synth_input, synth_state_dict, write_synth_calib
dev = torch.device("cuda:0")
calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
sd = synth_state_dict(alias_pretrained=True)
def mk(**kw):
    m = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True, **kw)
    m.load_state_dict(sd, strict=False)
    return m.eval().to(dev)
m1 = mk()
cases = {"graph2": dict(streams=2, graph=True), "eager2": dict(streams=2), "graph1": dict(graph=True), "eager1": dict()}
for name, kw in cases.items():
    first = mk(**kw)
    ms = mk(streams=2)
    bad = []
    for seed in range(60, 70):
        x = synth_input(4, seed0=seed).to(dev)
        a, sa = first.network(x)
        b, sb = ms.network(x)
        torch.cuda.synchronize()
        c, sc = m1.network(x)
        torch.cuda.synchronize()
        bad.append((int(((sa - sc).abs() > 1e-3).sum()), int(((sb - sc).abs() > 1e-3).sum())))
    print(name, "-> then eager2; (first bad, eager2 bad) per seed:", bad, flush=True)
