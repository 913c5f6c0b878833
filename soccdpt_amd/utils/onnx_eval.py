"""A small ONNX interpreter (opset 13 subset) on torch CPU ops: runs the graphs `soccdpt_amd/scripts/export_SOccDPT.py` writes, so that an exported
file can be checked without onnxruntime (absent from the image).  Every op follows the ONNX operator specification for opset 13; the op set is the
one the exporter emits plus what torch's own exporter produces for the small modules tests/test_onnx_export.py pins the semantics with.

Not a product path: nothing in the forward (libsoccdpt_hip.so) depends on it."""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import onnx_proto as P

_TORCH_OF = {P.FLOAT: torch.float32, P.INT64: torch.int64, P.INT32: torch.int32, P.BOOL: torch.bool, P.DOUBLE: torch.float64, P.FLOAT16: torch.float16,
             P.UINT8: torch.uint8, P.INT8: torch.int8}


def _ints(t: torch.Tensor) -> List[int]:
    return [int(v) for v in t.reshape(-1).tolist()]


def _resize(x, scales, sizes, a):
    mode = a.get("mode", "nearest")
    ctm = a.get("coordinate_transformation_mode", "half_pixel")
    nd = x.dim() - 2
    if sizes is not None and sizes.numel():
        out = _ints(sizes)[2:]
        kw = dict(size=out)
    else:
        sc = [float(v) for v in scales.reshape(-1).tolist()][2:]
        out = [int(np.floor(x.shape[2 + i] * sc[i])) for i in range(nd)]
        kw = dict(size=out)
    if mode == "nearest":
        if ctm != "asymmetric" or a.get("nearest_mode", "round_prefer_floor") != "floor":
            raise NotImplementedError(f"Resize nearest with {ctm} / {a.get('nearest_mode')}")
        return F.interpolate(x, mode="nearest", **kw)          # src = floor(dst * in / out): torch's 'nearest'
    if mode == "linear":
        if ctm == "align_corners":
            return F.interpolate(x, mode="bilinear", align_corners=True, **kw)
        if ctm in ("half_pixel", "pytorch_half_pixel"):
            return F.interpolate(x, mode="bilinear", align_corners=False, **kw)
        raise NotImplementedError(f"Resize linear with {ctm}")
    if mode == "cubic":
        if abs(float(a.get("cubic_coeff_a", -0.75)) + 0.75) > 1e-6 or int(a.get("exclude_outside", 0)) != 0:
            raise NotImplementedError("Resize cubic: only cubic_coeff_a = -0.75, exclude_outside = 0 (torch's bicubic)")
        if ctm == "align_corners":
            return F.interpolate(x, mode="bicubic", align_corners=True, **kw)
        if ctm in ("half_pixel", "pytorch_half_pixel"):
            return F.interpolate(x, mode="bicubic", align_corners=False, **kw)
        raise NotImplementedError(f"Resize cubic with {ctm}")
    raise NotImplementedError(f"Resize mode {mode}")


def run(model: P.Model, feeds: Dict[str, torch.Tensor], outputs: Optional[List[str]] = None) -> List[torch.Tensor]:
    g = model.graph
    env: Dict[str, Optional[torch.Tensor]] = {"": None}
    for t in g.initializers:
        env[t.name] = torch.from_numpy(t.array.copy())
    for k, v in feeds.items():
        env[k] = v
    # only what the requested outputs need, stopping at fed tensors: an intermediate tensor may be fed (and the nodes before it never run)
    want = list(outputs or [v.name for v in g.outputs])
    producer = {o: n for n in g.nodes for o in n.outputs}
    needed, stack = set(), [w for w in want if w not in env]
    while stack:
        name = stack.pop()
        n = producer.get(name)
        if n is None:
            raise KeyError(f"ONNX tensor {name!r} is neither fed, an initializer nor produced by a node")
        if id(n) in needed:
            continue
        needed.add(id(n))
        stack += [x for x in n.inputs if x not in env]
    for n in g.nodes:
        if id(n) not in needed:
            continue
        i = [env[x] for x in n.inputs]
        a, op = n.attrs, n.op_type
        get = lambda k: i[k] if k < len(i) else None
        if op == "Constant":
            o = torch.from_numpy(np.array(a["value"]).copy()) if "value" in a else torch.tensor(a.get("value_float", a.get("value_int")))
        elif op == "Identity":
            o = i[0]
        elif op == "Conv":
            pads = a.get("pads", [0] * (2 * (i[0].dim() - 2)))
            nd = i[0].dim() - 2
            if list(pads[:nd]) != list(pads[nd:]):
                x = F.pad(i[0], [p for d in reversed(range(nd)) for p in (pads[d], pads[nd + d])])
                pad = [0] * nd
            else:
                x, pad = i[0], list(pads[:nd])
            o = F.conv2d(x, i[1], get(2), stride=a.get("strides", [1] * nd), padding=pad, dilation=a.get("dilations", [1] * nd), groups=int(a.get("group", 1)))
        elif op == "MaxPool":   # padding cells never win: -inf, as the operator specification says
            nd = i[0].dim() - 2
            pads = list(a.get("pads", [0] * (2 * nd)))
            x = i[0]
            if any(pads):
                x = F.pad(x, [p for d in reversed(range(nd)) for p in (pads[d], pads[nd + d])], value=float("-inf"))
            if any(int(v) != 1 for v in a.get("dilations", [1] * nd)):
                raise NotImplementedError("MaxPool with dilations")
            o = F.max_pool2d(x, list(a["kernel_shape"]), list(a.get("strides", [1] * nd)), 0, 1, bool(a.get("ceil_mode", 0)))
        elif op == "BatchNormalization":   # inference form: (x - mean) / sqrt(var + eps) * scale + B
            o = F.batch_norm(i[0], i[3], i[4], i[1], i[2], False, 0.0, float(a.get("epsilon", 1e-5)))
        elif op == "MatMul":
            o = torch.matmul(i[0], i[1])
        elif op == "Gemm":
            A = i[0].t() if a.get("transA", 0) else i[0]
            Bm = i[1].t() if a.get("transB", 0) else i[1]
            o = float(a.get("alpha", 1.0)) * (A @ Bm)
            if get(2) is not None:
                o = o + float(a.get("beta", 1.0)) * i[2]
        elif op in ("Add", "Sub", "Mul", "Div", "Pow"):
            o = {"Add": torch.add, "Sub": torch.sub, "Mul": torch.mul, "Div": torch.div, "Pow": torch.pow}[op](i[0], i[1])
            if op == "Div" and not i[0].is_floating_point():
                o = torch.div(i[0], i[1], rounding_mode="trunc")
        elif op in ("Sqrt", "Erf", "Tanh", "Sigmoid", "Relu", "Exp", "Log", "Reciprocal", "Neg", "Abs", "Floor", "Ceil", "Not", "IsNaN"):
            o = {"Sqrt": torch.sqrt, "Erf": torch.erf, "Tanh": torch.tanh, "Sigmoid": torch.sigmoid, "Relu": torch.relu, "Exp": torch.exp, "Log": torch.log,
                 "Reciprocal": torch.reciprocal, "Neg": torch.neg, "Abs": torch.abs, "Floor": torch.floor, "Ceil": torch.ceil, "Not": torch.logical_not,
                 "IsNaN": torch.isnan}[op](i[0])
        elif op == "IsInf":
            o = torch.isinf(i[0])
            if not a.get("detect_negative", 1):
                o = o & (i[0] > 0)
            if not a.get("detect_positive", 1):
                o = o & (i[0] < 0)
        elif op in ("Max", "Min"):
            o = i[0]
            for t in i[1:]:
                o = torch.maximum(o, t) if op == "Max" else torch.minimum(o, t)
        elif op == "Clip":
            o = i[0]
            if get(1) is not None:
                o = torch.maximum(o, i[1])
            if get(2) is not None:
                o = torch.minimum(o, i[2])
        elif op in ("Equal", "Less", "Greater", "LessOrEqual", "GreaterOrEqual", "Or", "And"):
            o = {"Equal": torch.eq, "Less": torch.lt, "Greater": torch.gt, "LessOrEqual": torch.le, "GreaterOrEqual": torch.ge, "Or": torch.logical_or,
                 "And": torch.logical_and}[op](i[0], i[1])
        elif op == "Where":
            o = torch.where(i[0], i[1], i[2])
        elif op == "Cast":
            o = i[0].to(_TORCH_OF[int(a["to"])])
        elif op == "ReduceMean":
            o = i[0].mean(dim=list(a["axes"]), keepdim=bool(a.get("keepdims", 1)))
        elif op == "ReduceL2":
            o = torch.sqrt((i[0] * i[0]).sum(dim=list(a["axes"]), keepdim=bool(a.get("keepdims", 1))))
        elif op == "ReduceSum":     # opset 13: axes is an input
            ax = _ints(i[1]) if get(1) is not None else list(range(i[0].dim()))
            o = i[0].sum(dim=ax, keepdim=bool(a.get("keepdims", 1)))
        elif op == "Softmax":
            o = torch.softmax(i[0], dim=int(a.get("axis", -1)))
        elif op == "Reshape":
            shp = _ints(i[1])
            shp = [i[0].shape[k] if d == 0 else d for k, d in enumerate(shp)]
            o = i[0].reshape(shp)
        elif op == "Flatten":
            ax = int(a.get("axis", 1))
            o = i[0].reshape(int(np.prod(i[0].shape[:ax])) if ax else 1, -1)
        elif op == "Transpose":
            o = i[0].permute(list(a["perm"])) if "perm" in a else i[0].permute(list(reversed(range(i[0].dim()))))
        elif op == "Concat":
            o = torch.cat(i, dim=int(a["axis"]))
        elif op == "Slice":
            st, en = _ints(i[1]), _ints(i[2])
            axes = _ints(i[3]) if get(3) is not None else list(range(len(st)))
            steps = _ints(i[4]) if get(4) is not None else [1] * len(st)
            idx = [slice(None)] * i[0].dim()
            back = []   # axes walked backwards (negative step): an index_select after the forward slices
            for s0, e0, ax, sp in zip(st, en, axes, steps):
                dim = i[0].shape[ax]
                if sp < 0:      # starts clamp to [0, dim - 1], ends to [-1, dim - 1] (operator specification)
                    s0 = max(0, min(dim - 1, s0 + dim if s0 < 0 else s0))
                    e0 = max(-1, min(dim - 1, e0 + dim if e0 < 0 else e0))
                    back.append((ax, torch.arange(s0, e0, sp)))
                    continue
                s0 = max(0, min(dim, s0 + dim if s0 < 0 else s0))
                e0 = max(0, min(dim, e0 + dim if e0 < 0 else e0))
                idx[ax] = slice(s0, e0, sp)
            o = i[0][tuple(idx)]
            for ax, sel in back:
                o = torch.index_select(o, ax, sel)
        elif op == "Gather":
            ax = int(a.get("axis", 0))
            ind = i[1].long()
            ind = torch.where(ind < 0, ind + i[0].shape[ax], ind)
            o = torch.index_select(i[0], ax, ind.reshape(-1)).reshape(list(i[0].shape[:ax]) + list(ind.shape) + list(i[0].shape[ax + 1:]))
        elif op == "Unsqueeze":
            o = i[0]
            for ax in sorted(ax0 + (o.dim() + len(_ints(i[1])) if ax0 < 0 else 0) for ax0 in _ints(i[1])):
                o = o.unsqueeze(ax)
        elif op == "Squeeze":
            o = i[0]
            if get(1) is None:
                o = o.squeeze()
            else:
                for ax in sorted((ax0 + o.dim() if ax0 < 0 else ax0 for ax0 in _ints(i[1])), reverse=True):
                    o = o.squeeze(ax)
        elif op == "Expand":
            shp = _ints(i[1])
            o = i[0].expand(torch.broadcast_shapes(tuple(i[0].shape), tuple(shp)))
        elif op == "Shape":
            o = torch.tensor(list(i[0].shape), dtype=torch.int64)
        elif op == "ConstantOfShape":
            v = a.get("value")
            fill = torch.from_numpy(np.array(v)) if v is not None else torch.zeros(1)
            o = fill.reshape(()).expand(_ints(i[0])).clone()
        elif op == "Split":
            ax = int(a.get("axis", 0))
            parts = _ints(i[1]) if get(1) is not None else [i[0].shape[ax] // len(n.outputs)] * len(n.outputs)
            for name, t in zip(n.outputs, torch.split(i[0], parts, dim=ax)):
                env[name] = t
            continue
        elif op == "Pad":        # opset 13: pads (begin of every axis, then end of every axis) and the constant are inputs
            if a.get("mode", "constant") != "constant":
                raise NotImplementedError(f"Pad mode {a.get('mode')}")
            pads = _ints(i[1])
            nd = i[0].dim()
            value = float(i[2].reshape(-1)[0]) if get(2) is not None and i[2].numel() else 0.0
            o = F.pad(i[0], [p for d in reversed(range(nd)) for p in (pads[d], pads[nd + d])], value=value)
        elif op == "Resize":
            o = _resize(i[0], get(2), get(3), a)
        else:
            raise NotImplementedError(f"ONNX op {op} is not in the evaluator's subset")
        env[n.outputs[0]] = o
    return [env[k] for k in want]
