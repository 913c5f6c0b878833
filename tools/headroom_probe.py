"""How much head-room does a calibrated precision map need?  For three non-shipped weight sets and head-room factors 0.85 / 0.90 / 0.95: calibrate on six frames (4 + 2),
then measure the seven quantities against the library's exact-f32 mode on three batches the calibration never saw, and time the forward.   python tools/headroom_probe.py [model_type]"""
import contextlib, io, json, os, sys, tempfile, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soccdpt_amd.lib import PREC_F32, PREC_MIXED
from soccdpt_amd.model.SOccDPT import SOccDPT_V3
from soccdpt_amd.model.spec import MODEL_TYPE_TO_BACKBONE, backbone_image_size
from soccdpt_amd.utils.synth import named_weights, synth_input, write_synth_calib
Q = ("feat0", "feat1", "feat2", "feat3", "path1", "inv", "seg_logits")
mt = sys.argv[1] if len(sys.argv) > 1 else "dpt_swin2_tiny_256"
bb = MODEL_TYPE_TO_BACKBONE[mt]
S = backbone_image_size(bb)
B = 8 if S == 256 else 4
budget = 1e-3 if mt == "dpt_hybrid_384" else 5e-4
dev = torch.device("cuda:0")
calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))

def build(prec, sd):
    with contextlib.redirect_stdout(io.StringIO()):
        m = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True, precision=prec, model_type=mt)
        m.load_state_dict(sd, strict=False)
        m = m.eval().to(dev)
        m.network(synth_input(1, size=S).to(dev))
    return m

def quant(m, x):
    inv, _ = m.network(x)
    e = m._engine(dev)
    q = {k: e.workspace_tensor(x.shape[0], k).double() for k in Q if k != "inv"}
    q["inv"] = inv.double()
    return q

x_cal = synth_input(6 if S == 256 else 3, size=S, seed0=5000).to(dev)
tests = [synth_input(B, size=S, seed0=s).to(dev) for s in (0, 300, 700)]
for w in ("salt1", "salt2", "trained_like"):
    sd = named_weights(w, bb)
    rf = build(PREC_F32, sd)
    refs = [quant(rf, x) for x in tests]
    del rf
    for hr in (0.85, 0.90, 0.95):
        m = build(PREC_MIXED, sd)
        with contextlib.redirect_stdout(io.StringIO()):
            rep = m.calibrate_precision(x_cal, budget=budget, headroom=hr)
        worst = []
        for x, r in zip(tests, refs):
            q = quant(m, x)
            worst.append(max(float((q[k] - r[k]).norm() / r[k].norm()) for k in Q))
        for _ in range(20):
            m(tests[0])
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(200):
            m(tests[0])
        torch.cuda.synchronize(); fps = B * 200 / (time.perf_counter() - t)
        print(f"{mt} {w} headroom {hr}: {rep['n_x3']} x3 + {rep['n_x2w']} x2w of {rep['n_groups']}; library calib {rep['worst_calibrated']:.3e} holdout {rep['worst_holdout']:.3e}; "
              f"three unseen batches worst {', '.join(f'{v:.3e}' for v in worst)} (budget {budget}); {fps:.0f} frames/s", flush=True)
        del m
