"""Reproducer of the packed-f32 problem recorded in DESIGN.md section 4: with the library built WITH packed-f32 VALU ops
(`make -C soccdpt_amd/csrc clean all NOPK=`), the eager two-stream seg output deviates from the single-stream one as soon as the
seg-head conv of the other sub-batch runs on 64x64 / 128x128 tiles next to conv1x1_c3_kernel; with the shipped build (no packed ops)
every variant reports 0.  Forces tiles per shape with tune_set and counts mismatching elements per variant.
usage: python tools/race_bisect.py [M,N,K,taps,cfg ...]"""
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
eng = ms._engine(dev)
SEG = (32768, 256, 2304, 9)
variants = [("heuristic", []), ("seg->10", [SEG + (10,)]), ("seg->2", [SEG + (2,)]), ("seg->16", [SEG + (16,)]), ("seg->1", [SEG + (1,)])]
if len(sys.argv) > 1:
    variants = [(v, [tuple(int(t) for t in v.split(","))]) for v in sys.argv[1:]]
for name, sets in variants:
    eng.tune_clear()
    for s in sets:
        eng.tune_set(*s)
    nbad = nseed = 0
    for seed in range(60, 160):
        x = synth_input(4, seed0=seed).to(dev)
        a, sa = mg.network(x); b, sb = ms.network(x); c, sc = m1.network(x)
        torch.cuda.synchronize()
        n = int(((sb - sc).abs() > 1e-3).sum())
        nbad += n; nseed += n > 0
    print(f"variant {name}: {nbad} bad elements in {nseed} / 100 seeds", flush=True)
