"""Soak test: the same B = 8 forward repeated N times must reproduce its first result bit for bit (network outputs and the packed
occupancy grid), in every precision mode -- a rare schedule race shows up here as a sporadic mismatch."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from soccdpt_amd.model.SOccDPT import SOccDPT_V3
from soccdpt_amd.utils.synth import synth_input, synth_state_dict, write_synth_calib
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
for model_type, backbone, img, n in (("dpt_swin2_tiny_256", "swin2t16_256", 256, N), ("dpt_swin2_base_384", "swin2b24_384", 384, N // 4)):
    sd = synth_state_dict(backbone, alias_pretrained=True)
    for prec, name in ((4, "mixed"), (0, "bf16"), (2, "f16"), (3, "f16x3"), (1, "f32")):
        reps = n if prec not in (1, 3) else n // 4
        m = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True, precision=prec, model_type=model_type)
        m.load_state_dict(sd, strict=False)
        m = m.eval().to(dev)
        x = synth_input(8, size=img, seed0=7).to(dev)
        inv0, seg0 = m.network(x)
        out0 = m(x)
        bits0 = m.last_occ_bits.clone()
        torch.cuda.synchronize()
        bad = 0
        for i in range(reps):
            inv, seg = m.network(x)
            if i % 8 == 0:
                out = m(x)
                bad += int(not torch.equal(m.last_occ_bits, bits0)) + int(not torch.equal(out[0], out0[0]))
            bad += int(not torch.equal(inv, inv0)) + int(not torch.equal(seg, seg0))
        torch.cuda.synchronize()
        print(f"{model_type} {name}: {reps} repeats, mismatching results: {bad}", flush=True)
        del m
