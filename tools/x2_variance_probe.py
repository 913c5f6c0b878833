"""Feasibility probe for one-sided operand splits: how the rounding variance of an fp16 launch site divides between its two operands.

x3 (hi + lo on both operands, three MFMAs) removes a site's whole fp16 rounding error at twice its operand bytes.  Two cheaper formats would
keep ONE operand at 2 bytes: x2a (activations hi + lo, weights fp16: for the weight-heavy small-M sites of Swin stages 2-3) and x2w (weights
hi + lo, activations fp16: for the activation-heavy 1x1 sites).  Each removes only the other operand's share of the variance.  This probe
measures the shares through the product path, without any new kernel:

    T_g  = variance group g adds in fp16 (all groups x3 except g in fp16, minus the all-x3 floor)            -- as soccdpt_prec_calibrate does
    W_g  = variance of g's WEIGHT rounding alone: all groups x3, g's weights rounded to fp16 on the host (an fp16-representable weight has lo = 0,
           so the x3 launch computes exactly what x2a would: activations hi + lo, weights fp16)
    A_g  = T_g - W_g = what x2w would leave in

then prices a three-format greedy selection (fp16 / x2 / x3) with the byte model of these fill-bound launches: the extra cost of a format over
fp16 scales with its extra operand bytes (x3: A + W extra; x2a: A; x2w: W, with A = M K, W = N K elements of the group's launches).

    python tools/x2_variance_probe.py [model_type] [budget] > gpurun_out/x2_variance_<model>.json
"""
import json
import math
import os
import re
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from soccdpt_amd.lib import PREC_F16, PREC_F16X3, PREC_F32, PREC_MIXED
from soccdpt_amd.model.SOccDPT import SOccDPT_V3
from soccdpt_amd.model.spec import MODEL_TYPE_TO_BACKBONE, backbone_image_size
from soccdpt_amd.utils.synth import synth_input, synth_state_dict, write_synth_calib

model_type = sys.argv[1] if len(sys.argv) > 1 else "dpt_swin2_tiny_256"
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 5e-4
QUANT = ["feat0", "feat1", "feat2", "feat3", "path1", "inv", "seg_logits"]
dev = torch.device("cuda:0")
backbone = MODEL_TYPE_TO_BACKBONE[model_type]
assert backbone.startswith("swin"), "the probe maps Swin-V2 group names to weight keys"
img = backbone_image_size(backbone)
calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
sd0 = synth_state_dict(backbone, alias_pretrained=True)
ENC, SCR = "depth_net.pretrained.model.", "depth_net.scratch."


def group_keys(g):
    m = re.fullmatch(r"s(\d)\.b(\d+)\.(qkv|proj|fc1|fc2)", g)
    if m:
        s, j, part = m.groups()
        leaf = {"qkv": "attn.qkv", "proj": "attn.proj", "fc1": "mlp.fc1", "fc2": "mlp.fc2"}[part]
        return [f"{ENC}layers.{s}.blocks.{j}.{leaf}.weight"]
    m = re.fullmatch(r"merge(\d)", g)
    if m:
        return [f"{ENC}layers.{m.group(1)}.downsample.reduction.weight"]
    m = re.fullmatch(r"(lrn|ref|oc)(\d)", g)
    if m:
        kind, l = m.group(1), int(m.group(2)) + 1
        if kind == "lrn":
            return [f"{SCR}layer{l}_rn.weight"]
        if kind == "oc":
            return [f"{SCR}refinenet{l}.out_conv.weight"]
        return [f"{SCR}refinenet{l}.resConfUnit{u}.conv{c}.weight" for u in (1, 2) for c in (1, 2) if not (l == 4 and u == 1)]
    if g == "head":
        return [f"{SCR}output_conv.0.weight", "seg_head.0.weight"]
    if g == "head.d2":
        return [f"{SCR}output_conv.2.weight"]
    return []


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build(prec, sd):
    net = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=True, model_type=model_type, precision=prec)
    net.load_state_dict(sd, strict=False)
    return net.eval().to(dev)


def quantities(net, x):
    inv, _ = net.network(x)
    eng = net._engine(dev)
    out = {q: eng.workspace_tensor(x.shape[0], q).double() for q in QUANT if q != "inv"}
    out["inv"] = inv.double()
    return out


def rel(a, b):
    return float((a - b).norm() / b.norm())


x = synth_input(2, size=img, seed0=4).to(dev)
ref = quantities(build(PREC_F32, sd0), x)
net = build(PREC_MIXED, sd0)
eng = net._engine(dev)
groups = list(eng.prec_map())
live = dict(net.named_parameters(remove_duplicate=False))


def errors():
    q = quantities(net, x)
    return {k: rel(q[k], ref[k]) for k in QUANT}


eng.prec_map_set("*", PREC_F16X3)
base = errors()
log("all x3:", {k: f"{v:.2e}" for k, v in base.items()})
table = {}
for g in groups:
    eng.prec_map_set("*", PREC_F16X3)
    eng.prec_map_set(g, PREC_F16)
    e = errors()
    T = {k: max(e[k] ** 2 - base[k] ** 2, 0.0) for k in QUANT}
    eng.prec_map_set("*", PREC_F16X3)
    keys = [k for k in group_keys(g) if k in live]
    saved = {k: live[k].detach().clone() for k in keys}
    with torch.no_grad():
        for k in keys:
            live[k].copy_(live[k].to(torch.float16).float())      # in place: bumps _version, the engine re-binds and re-prepares
    e = errors()
    with torch.no_grad():
        for k in keys:
            live[k].copy_(saved[k])
    W = {k: max(e[k] ** 2 - base[k] ** 2, 0.0) for k in QUANT}
    elems_w = sum(sd0[k].numel() for k in keys)
    table[g] = {"T": T, "W": W, "weight_elems": elems_w, "keys": len(keys)}
    tw, tt = sum(W.values()), sum(T.values())
    log(f"{g:12s} weight share of the fp16 variance {tw / tt if tt > 0 else float('nan'):.2f}   sqrt(T) worst {math.sqrt(max(T.values())):.2e}")

shares = [sum(v["W"].values()) / sum(v["T"].values()) for v in table.values() if sum(v["T"].values()) > 0 and v["keys"]]
shares.sort()
summary = {"model": model_type, "budget": budget, "all_x3": base, "groups": table,
           "weight_share_of_variance": {"median": shares[len(shares) // 2], "p10": shares[len(shares) // 10], "p90": shares[(9 * len(shares)) // 10], "n": len(shares)}}
log("weight share of a site's fp16 rounding variance: median %.2f, p10 %.2f, p90 %.2f over %d groups" % (
    summary["weight_share_of_variance"]["median"], summary["weight_share_of_variance"]["p10"], summary["weight_share_of_variance"]["p90"], len(shares)))
print(json.dumps(summary, indent=1))
