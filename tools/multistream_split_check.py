"""Companion of multistream_probe.py for batch sizes at which the sub-batches take different tile / split-K decisions than the whole
batch (base_384, B = 8: layer4_rn is split-K at 4 frames and not at 8, so sub-batch results differ from the whole-batch ones in the last
bits by construction).  Here the single-stream model runs the SAME sub-batches one after the other: the concurrent result must equal
that bit for bit.  usage: python tools/multistream_split_check.py [model_type] [bf16|f16] [streams] [B] [n_inputs]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from soccdpt_amd.model.SOccDPT import SOccDPT_V3
from soccdpt_amd.model.spec import MODEL_TYPE_TO_BACKBONE, SWIN_ARCHS
from soccdpt_amd.utils.synth import synth_input, synth_state_dict, write_synth_calib
model_type = sys.argv[1] if len(sys.argv) > 1 else "dpt_swin2_base_384"
prec = {"bf16": 0, "f32": 1, "f16": 2}[sys.argv[2] if len(sys.argv) > 2 else "bf16"]
streams = int(sys.argv[3]) if len(sys.argv) > 3 else 2
B = int(sys.argv[4]) if len(sys.argv) > 4 else 8
n = int(sys.argv[5]) if len(sys.argv) > 5 else 40
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
per = B // streams
tot = dict(graph_seg=0, eager_seg=0, graph_inv=0, eager_inv=0)
for seed in range(60, 60 + n):
    x = synth_input(B, size=img, seed0=seed).to(dev)
    a, sa = mg.network(x); b, sb = ms.network(x)
    parts = [m1.network(x[i * per:(i + 1) * per].contiguous()) for i in range(streams)]
    c = torch.cat([p[0] for p in parts]); sc = torch.cat([p[1] for p in parts])
    torch.cuda.synchronize()
    tot["graph_seg"] += int((sa != sc).sum()); tot["eager_seg"] += int((sb != sc).sum())
    tot["graph_inv"] += int((a != c).sum()); tot["eager_inv"] += int((b != c).sum())
print(f"{model_type} prec={prec} streams={streams} B={B}: {n} inputs, elements differing from the same sub-batches run one by one:", tot, flush=True)
