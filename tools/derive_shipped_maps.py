"""The shipped precision maps (csrc/model.cpp: model_prec_default) = soccdpt_prec_calibrate on the synthetic weights of the tests and the benchmark, with a
little head-room under the bar the tests hold them to (5e-4; dpt_hybrid_384: 1e-3).  Prints the C++ initialiser lists and writes the reports.
    python tools/derive_shipped_maps.py [tiny256|base384|hybrid384] > gpurun_out/r06_shipped_maps.txt"""
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from soccdpt_amd.lib import PREC_MIXED
from soccdpt_amd.model.SOccDPT import SOccDPT_V3
from soccdpt_amd.model.spec import MODEL_TYPE_TO_BACKBONE, backbone_image_size
from soccdpt_amd.utils.synth import synth_input, synth_state_dict, write_synth_calib

dev = torch.device("cuda:0")
calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
only = sys.argv[1] if len(sys.argv) > 1 else None


def fps(net, x, steps=150):
    import time
    for _ in range(20):
        net(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        net(x)
    torch.cuda.synchronize()
    return x.shape[0] * steps / (time.perf_counter() - t0)


for model_type, budget, tag in (("dpt_swin2_tiny_256", 5e-4, "tiny256"), ("dpt_swin2_base_384", 5e-4, "base384"), ("dpt_hybrid_384", 1e-3, "hybrid384")):   # round 6: the budget itself; the head-room is the calibration's (0.85 on the calibration frames, the budget on its hold-out frames)
    if only and only != tag:
        continue
    backbone = MODEL_TYPE_TO_BACKBONE[model_type]
    net = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True, model_type=model_type, precision=PREC_MIXED)
    net.load_state_dict(synth_state_dict(backbone, alias_pretrained=True), strict=False)
    net = net.eval().to(dev)
    x = synth_input(6 if tag == "tiny256" else 3, size=backbone_image_size(backbone), seed0=4).to(dev)   # 4 + 2 frames (2 + 1 at 384 px)
    rep = net.calibrate_precision(x, budget=budget)
    q = lambda gs: ", ".join('"%s"' % g for g in gs)
    print(f"// {model_type}: budget {budget:g}, worst of the seven quantities {rep['worst_calibrated']:.2e} (fp16 everywhere: {rep['worst_all_fp16']:.2e}); "
          f"{rep['n_x3']} groups x3, {rep['n_x2w']} x2w of {rep['n_groups']}; {rep['forwards']} forwards")
    print("            x3({%s});" % q(rep["x3_groups"]))
    print("            x2w({%s});" % q(rep["x2w_groups"]))
    xb = synth_input(4 if tag == "hybrid384" else 8, size=backbone_image_size(backbone), seed0=0).to(dev)
    f_new = fps(net, xb)
    eng = net._engine(dev)
    if tag == "hybrid384":   # the round-4 map in the round-5 group names, for comparison
        eng.prec_map_set("*", 2)
        for g in ("rn.s0.*", "rn.s1.*", "rn.s2.*", "ro1", "oc0", "oc1", "oc2", "oc3", "head.s1"):
            eng.prec_map_set(g, 3)
        print(f"// frames/s: calibrated map {f_new:.1f}, round-4 map {fps(net, xb):.1f}")
    else:
        print(f"// frames/s: calibrated map {f_new:.1f}")
    json.dump(rep, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", f"r06_precision_map_{tag}.json"), "w"), indent=1)
    sys.stdout.flush()
    del net
