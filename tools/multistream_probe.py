"""Stress probe of the multi-stream paths: hipGraph(n streams) -> eager(n streams) -> eager(1 stream) launched back to back without
host synchronisation, over many inputs; counts output elements of the multi-stream results that differ from the single-stream one.
Co-residency of different kernels only happens in these modes; this probe found the packed-f32 problem recorded in DESIGN.md
section 4 (seg output of the eager two-stream mode) and is the check to run after any kernel or schedule change.
Note: the comparison is against the WHOLE batch on one stream, so it only reads 0 when the sub-batches take the same tile / split-K
decisions as the whole batch (tiny_256 at B = 4 / 8, base_384 at B = 4); where they do not (base_384 at B = 8) use
tools/multistream_split_check.py, which compares against the same sub-batches run one after the other.
usage: python tools/multistream_probe.py [model_type] [bf16|f16|f32] [streams] [B] [n_inputs]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from soccdpt_amd.model.SOccDPT import SOccDPT_V3
from soccdpt_amd.model.spec import MODEL_TYPE_TO_BACKBONE, SWIN_ARCHS
from soccdpt_amd.utils.synth import synth_input, synth_state_dict, write_synth_calib
model_type = sys.argv[1] if len(sys.argv) > 1 else "dpt_swin2_tiny_256"
prec = {"bf16": 0, "f32": 1, "f16": 2}[sys.argv[2] if len(sys.argv) > 2 else "bf16"]
streams = int(sys.argv[3]) if len(sys.argv) > 3 else 2
B = int(sys.argv[4]) if len(sys.argv) > 4 else 4
n = int(sys.argv[5]) if len(sys.argv) > 5 else 100
dev = torch.device("cuda:0")
backbone = MODEL_TYPE_TO_BACKBONE[model_type]
img = SWIN_ARCHS[backbone].img
calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
sd = synth_state_dict(backbone, alias_pretrained=True)
def mk(**kw):
    m = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True, model_type=model_type, precision=prec, **kw)
    m.load_state_dict(sd, strict=False)
    return m.eval().to(dev)
mg, ms, m1 = mk(streams=streams, graph=True), mk(streams=streams), mk()
tot = dict(graph_seg=0, eager_seg=0, graph_inv=0, eager_inv=0)
for seed in range(60, 60 + n):
    x = synth_input(B, size=img, seed0=seed).to(dev)
    a, sa = mg.network(x); b, sb = ms.network(x); c, sc = m1.network(x)
    torch.cuda.synchronize()
    tot["graph_seg"] += int(((sa - sc).abs() > 1e-3).sum()); tot["eager_seg"] += int(((sb - sc).abs() > 1e-3).sum())
    tot["graph_inv"] += int((a != c).sum()); tot["eager_inv"] += int((b != c).sum())
print(f"{model_type} prec={prec} streams={streams} B={B}: {n} inputs, differing elements vs 1 stream:", tot, flush=True)
