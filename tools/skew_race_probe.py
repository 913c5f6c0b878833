"""Reproducer of the timing-dependent mismatch that the (removed) skewed igemm schedule produced -- kept as a stress probe for
any future schedule change of igemm_kernel: graph(2 streams) -> eager(2 streams) ->
eager(1 stream) launched back to back without host synchronisation; counts seg elements of the eager 2-stream result that differ
from the single-stream one."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
mg, ms, m1 = mk(streams=2, graph=True), mk(streams=2), mk()
tot = [0, 0, 0]
for seed in range(60, 160):
    x = synth_input(4, seed0=seed).to(dev)
    a, sa = mg.network(x); b, sb = ms.network(x); c, sc = m1.network(x)
    torch.cuda.synchronize()
    tot[0] += int(((sa - sc).abs() > 1e-3).sum()); tot[1] += int(((sb - sc).abs() > 1e-3).sum()); tot[2] += int((b != c).sum()) + int((a != c).sum())
print("MISMATCH graph-vs-1stream", tot[0], " eager2-vs-1stream", tot[1], " inv", tot[2])
