"""A/B of the XCD-local persistent stage kernel against the launch chain inside ONE GPU call (old / new alternated), per precision:
ms per forward with and without it, and the device time of the `stage_xcd` launch against the chain's stage-2/3 launches.

    python tools/stage_xcd_ab.py [precision ...] > gpurun_out/stage_xcd_ab.json
"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from soccdpt_amd.model.SOccDPT import SOccDPT_V3
from soccdpt_amd.utils.synth import synth_input, synth_state_dict, write_synth_calib

PREC = {"bf16": 0, "f16": 2, "mixed": 4}
precs = sys.argv[1:] or ["mixed", "f16", "bf16"]
dev = torch.device("cuda:0")
calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
sd = synth_state_dict(alias_pretrained=True)
x = synth_input(8, seed0=0).to(dev)


def build(p, xcd):
    net = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True, precision=PREC[p])
    net.load_state_dict(sd, strict=False)
    net = net.eval().to(dev)
    if xcd:
        net._engine(dev).set_stage_xcd(True)
    return net


def ms(net, steps=200):
    for _ in range(20):
        net(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        net(x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def families(net, steps=20):
    eng = net._engine(dev)
    eng.profile_enable(True)
    for _ in range(steps):
        net(x)
    torch.cuda.synchronize()
    st = eng.profile_collect()
    eng.profile_enable(False)
    return {k: round(v["ms"] / steps * 1e3, 1) for k, v in st.items()}, round(sum(v["ms"] for v in st.values()) / steps * 1e3, 1)


out = {}
for p in precs:
    chain, pers = build(p, False), build(p, True)
    rounds = [(ms(chain), ms(pers)) for _ in range(3)]
    fc, tc = families(chain)
    fp, tp = families(pers)
    out[p] = {"ms_chain": [round(a, 4) for a, _ in rounds], "ms_persistent": [round(b, 4) for _, b in rounds],
              "device_us_chain": tc, "device_us_persistent": tp, "stage_xcd_us": fp.get("stage_xcd"),
              "status": pers._engine(dev).stage_xcd_status(), "launches": [chain._engine(dev).launch_count(), pers._engine(dev).launch_count()],
              "families_chain": fc, "families_persistent": fp}
    print(p, out[p]["ms_chain"], out[p]["ms_persistent"], "device us", tc, tp, "stage_xcd", fp.get("stage_xcd"), file=sys.stderr, flush=True)
    # phase timeline of the persistent launch (XCD 0's rank-0 workgroup; 100 MHz stamps)
    eng = pers._engine(dev)
    nph = 58
    eng.stage_xcd_timeline(1, 0)
    pers(x)
    t = eng.stage_xcd_timeline(0, 3 * nph)
    names = []
    for s_, nb in ((2, 6), (3, 2)):
        for j in range(nb):
            names += [f"s{s_}.b{j}.{k}" for k in ("qkv", "attn", "proj", "ln1", "fc1", "fc2", "ln2")]
        if s_ == 2:
            names += ["merge2", "merge2.ln"]
    tl = [{"phase": names[i], "work_us": round((t[3 * i + 1] - t[3 * i]) / 100.0, 2), "barrier_us": round((t[3 * i + 2] - t[3 * i + 1]) / 100.0, 2)} for i in range(nph)]
    out[p]["timeline"] = tl
    out[p]["timeline_total_us"] = round((t[3 * nph - 1] - t[0]) / 100.0, 1)
    print(p, "timeline total", out[p]["timeline_total_us"], "us; work", round(sum(e["work_us"] for e in tl), 1), "barriers", round(sum(e["barrier_us"] for e in tl), 1), file=sys.stderr)
    for e in tl[:9] + tl[42:51]:
        print("   ", e, file=sys.stderr)
print(json.dumps(out, indent=1))
