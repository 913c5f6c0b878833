"""ONNX export of SOccDPT_V3 -- the counterpart of /root/reference/SOccDPT/scripts/export_SOccDPT.py (same flags; :122-141 calls
torch.onnx.export(net, x, export_path, opset_version=13, input_names=["input"], output_names=["output"], dynamic batch)).

The reference traces the torch forward.  This build's forward is a launch sequence of HIP kernels behind a C ABI, so there is nothing to trace:
the graph of `SOccDPT_V3.forward` with compute_occ=False -- Swin-V2 (tiny_256, base_384) or ViT-hybrid (dpt_hybrid_384) encoder, DPT decoder, depth + seg
heads, bicubic / nearest up-sampling and
the back-projection to camera-frame points (model/SOccDPT.py:264-372, 681-685) -- is written node by node in standard opset-13 operators from
the model's own weights (`soccdpt_amd.utils.onnx_proto`, a protobuf writer; the `onnx` package is not in the image).  What the reference's
constant folding would fold is folded here too: the continuous-position-bias MLP, the logit scales and the shift masks become initializers, and so do
the standardised StdConv2dSame filters of the hybrid's ResNetV2.

Outputs (the 3-output graph run_SOccDPT_onnx.py:165-176 consumes): "output" = inverse depth [B, Hc, Wc], "segmentation" [B, C, Hc, Wc], "points"
[B, Hc, Wc, 3] (incl. the reference's 3-pixel pc_scale quirk); input "input" [B, 3, S, S]; batch is dynamic.  `soccdpt_amd.utils.onnx_eval`
interprets the file with torch CPU ops (tests/test_onnx_export.py checks it against the fp32 oracle and the golden fixtures).

    python -m soccdpt_amd.scripts.export_SOccDPT -v 3 -dt bdd -t dpt_swin2_tiny_256 -e onnx/SOccDPT.onnx [-l checkpoint.pth]
"""
from __future__ import annotations

import argparse
import math
import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from ..utils import onnx_proto as P


class GraphBuilder:
    """Tiny tracing API: every method appends one node and returns the name of its output."""

    def __init__(self):
        self.g = P.Graph("SOccDPT_V3")
        self._n = 0
        self._const: Dict[tuple, str] = {}

    def _name(self, hint: str) -> str:
        self._n += 1
        return f"{hint}_{self._n}"

    def init(self, array, hint: str = "c") -> str:
        a = np.ascontiguousarray(array)
        if a.size <= 64:   # small constants are shared
            key = (a.dtype.str, a.shape, a.tobytes())
            if key in self._const:
                return self._const[key]
        name = self._name(hint)
        self.g.initializers.append(P.Tensor(name, a))
        if a.size <= 64:
            self._const[key] = name
        return name

    def weight(self, t: torch.Tensor, hint: str) -> str:
        return self.init(t.detach().cpu().to(torch.float32).numpy(), hint)

    def i64(self, vals: Sequence[int]) -> str:
        return self.init(np.asarray(vals, dtype=np.int64), "i")

    def f32(self, v) -> str:
        return self.init(np.asarray(v, dtype=np.float32), "f")

    def op(self, op_type: str, inputs: Sequence[str], hint: Optional[str] = None, n_out: int = 1, **attrs):
        outs = [self._name(hint or op_type.lower()) for _ in range(n_out)]
        self.g.nodes.append(P.Node(op_type, list(inputs), outs, dict(attrs), name=outs[0]))
        return outs[0] if n_out == 1 else outs

    # ---- shorthands ----
    def reshape(self, x, shape): return self.op("Reshape", [x, self.i64(shape)])
    def transpose(self, x, perm): return self.op("Transpose", [x], perm=list(perm))
    def add(self, a, b): return self.op("Add", [a, b])
    def sub(self, a, b): return self.op("Sub", [a, b])
    def mul(self, a, b): return self.op("Mul", [a, b])
    def div(self, a, b): return self.op("Div", [a, b])
    def matmul(self, a, b): return self.op("MatMul", [a, b])
    def relu(self, x): return self.op("Relu", [x])
    def concat(self, xs, axis): return self.op("Concat", list(xs), axis=int(axis))

    def slice(self, x, starts, ends, axes, steps=None):
        ins = [x, self.i64(starts), self.i64(ends), self.i64(axes)]
        if steps is not None:
            ins.append(self.i64(steps))
        return self.op("Slice", ins)

    def linear(self, x, w: torch.Tensor, b: Optional[torch.Tensor], hint: str):
        y = self.matmul(x, self.weight(w.t().contiguous(), hint + "_wT"))
        return self.add(y, self.weight(b, hint + "_b")) if b is not None else y

    def layer_norm(self, x, g: torch.Tensor, b: torch.Tensor, eps: float, hint: str):
        mean = self.op("ReduceMean", [x], axes=[-1], keepdims=1)
        d = self.sub(x, mean)
        var = self.op("ReduceMean", [self.mul(d, d)], axes=[-1], keepdims=1)
        y = self.div(d, self.op("Sqrt", [self.add(var, self.f32(eps))]))
        return self.add(self.mul(y, self.weight(g, hint + "_g")), self.weight(b, hint + "_b"))

    def gelu(self, x):   # exact (erf) GELU, nn.GELU()
        e = self.op("Erf", [self.div(x, self.f32(math.sqrt(2.0)))])
        return self.mul(self.mul(x, self.add(e, self.f32(1.0))), self.f32(0.5))

    def conv(self, x, w: torch.Tensor, b: Optional[torch.Tensor], hint: str, stride: int = 1, pad: int = 0, pads: Optional[Sequence[int]] = None):
        k = int(w.shape[-1])
        ins = [x, self.weight(w, hint + "_w")] + ([self.weight(b, hint + "_b")] if b is not None else [])
        return self.op("Conv", ins, kernel_shape=[k, k], strides=[stride, stride], pads=list(pads) if pads is not None else [pad] * 4, dilations=[1, 1], group=1)

    def group_norm(self, x, C: int, H: int, W: int, g: torch.Tensor, b: torch.Tensor, groups: int, eps: float, hint: str):
        """nn.GroupNorm on [B, C, H, W] (opset 13 has no GroupNormalization): per (sample, group) statistics over a [B, groups, -1] view."""
        v = self.reshape(x, [0, groups, -1])
        mean = self.op("ReduceMean", [v], axes=[-1], keepdims=1)
        d = self.sub(v, mean)
        var = self.op("ReduceMean", [self.mul(d, d)], axes=[-1], keepdims=1)
        y = self.reshape(self.div(d, self.op("Sqrt", [self.add(var, self.f32(eps))])), [0, C, H, W])
        return self.add(self.mul(y, self.weight(g.reshape(1, C, 1, 1), hint + "_g")), self.weight(b.reshape(1, C, 1, 1), hint + "_b"))

    def resize(self, x, sh: float, sw: float, mode: str):
        attrs = {"bilinear_ac": dict(mode="linear", coordinate_transformation_mode="align_corners"),
                 "bicubic": dict(mode="cubic", coordinate_transformation_mode="half_pixel", cubic_coeff_a=-0.75),
                 "nearest": dict(mode="nearest", coordinate_transformation_mode="asymmetric", nearest_mode="floor")}[mode]
        return self.op("Resize", [x, "", self.f32([1.0, 1.0, sh, sw])], **attrs)

    def resize_to(self, x, H: int, W: int, mode: str):
        """F.interpolate(x, size=(H, W)): Resize with the `sizes` input ([batch, channels] taken from Shape(x), so the batch stays dynamic), as torch.onnx emits
        for size=.  With `scales` = H / S an exported graph computes floor(S * float32(H / S)), one pixel short whenever H / S is not exact in float32
        (width 1241 at S = 384) -- and the constant camera-size initialisers would no longer broadcast."""
        attrs = {"bilinear_ac": dict(mode="linear", coordinate_transformation_mode="align_corners"),
                 "bicubic": dict(mode="cubic", coordinate_transformation_mode="half_pixel", cubic_coeff_a=-0.75),
                 "nearest": dict(mode="nearest", coordinate_transformation_mode="asymmetric", nearest_mode="floor")}[mode]
        sizes = self.concat([self.slice(self.op("Shape", [x]), [0], [2], [0]), self.i64([int(H), int(W)])], 0)
        return self.op("Resize", [x, "", "", sizes], **attrs)

    def roll2(self, x, shift: int, res: int):
        """torch.roll(x, (shift, shift), dims=(1, 2)) of [B, res, res, C] for a shift in (-res, res)."""
        s = shift % res
        if s == 0:
            return x
        for ax in (1, 2):
            x = self.concat([self.slice(x, [res - s], [res], [ax]), self.slice(x, [0], [res - s], [ax])], ax)
        return x


# ---- constants the reference computes from the weights at every forward (timm WindowAttention); folded like do_constant_folding=True would ----
def _cpb_bias(sd, pfx: str, ws: int, pretrained_ws: int, heads: int) -> torch.Tensor:
    r = torch.arange(-(ws - 1), ws, dtype=torch.float32)
    tab = torch.stack(torch.meshgrid(r, r, indexing="ij"), dim=-1)
    tab = tab / ((pretrained_ws - 1) if pretrained_ws > 0 else (ws - 1)) * 8
    tab = (torch.sign(tab) * torch.log2(torch.abs(tab) + 1.0) / math.log2(8)).reshape(-1, 2)
    h = torch.relu(torch.nn.functional.linear(tab, sd[pfx + "cpb_mlp.0.weight"], sd[pfx + "cpb_mlp.0.bias"]))
    t = torch.nn.functional.linear(h, sd[pfx + "cpb_mlp.2.weight"])
    c = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = c[:, :, None] - c[:, None, :]
    idx = ((rel[0] + ws - 1) * (2 * ws - 1) + (rel[1] + ws - 1)).reshape(-1)
    n = ws * ws
    return 16.0 * torch.sigmoid(t[idx].reshape(n, n, heads).permute(2, 0, 1).contiguous())


def _shift_mask(res: int, ws: int, shift: int) -> torch.Tensor:
    idx = torch.arange(res)
    reg = (idx >= res - ws).long() + (idx >= res - shift).long()
    img = (reg[:, None] * 3 + reg[None, :]).float()
    nw = res // ws
    win = img.reshape(nw, ws, nw, ws).permute(0, 2, 1, 3).reshape(nw * nw, ws * ws)
    diff = win[:, None, :] - win[:, :, None]
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


SWIN = {   # model/dpt.py:51-89 (hooks), backbones/swin2.py:15-30 (timm model names)
    "swin2t16_256": dict(img=256, patch=4, embed=96, depths=(2, 2, 6, 2), heads=(3, 6, 12, 24), window=16, pretrained=(0, 0, 0, 0), hooks=(1, 1, 5, 1)),
    "swin2b24_384": dict(img=384, patch=4, embed=128, depths=(2, 2, 18, 2), heads=(4, 8, 16, 32), window=24, pretrained=(12, 12, 12, 6), hooks=(1, 1, 17, 1)),
}


def _swin_encoder(b: GraphBuilder, sd, A) -> List[str]:
    """timm SwinTransformerV2 (backbones/swin2.py:15-30) with the forward hooks + act_postprocess of backbones/swin_common.py:12-54: the four NCHW maps."""
    ENC = "depth_net.pretrained.model."
    S, C0 = A["img"], A["embed"]
    t = b.conv("input", sd[ENC + "patch_embed.proj.weight"], sd[ENC + "patch_embed.proj.bias"], "patch_embed", stride=A["patch"])
    t = b.transpose(b.reshape(t, [0, C0, -1]), [0, 2, 1])
    t = b.layer_norm(t, sd[ENC + "patch_embed.norm.weight"], sd[ENC + "patch_embed.norm.bias"], 1e-5, "patch_norm")
    res, feats = S // A["patch"], []
    for s, depth in enumerate(A["depths"]):
        C, heads = C0 << s, A["heads"][s]
        d = C // heads
        for j in range(depth):
            pfx = f"{ENC}layers.{s}.blocks.{j}."
            ws = min(res, A["window"])
            shift = 0 if (j % 2 == 0 or res <= A["window"]) else A["window"] // 2
            nw, N, L = res // ws, ws * ws, res * res
            # window partition (+ cyclic shift)
            h = b.roll2(b.reshape(t, [-1, res, res, C]), -shift, res)
            win = b.reshape(b.transpose(b.reshape(h, [-1, nw, ws, nw, ws, C]), [0, 1, 3, 2, 4, 5]), [-1, N, C])
            qkv_bias = torch.cat([sd[pfx + "attn.q_bias"], torch.zeros_like(sd[pfx + "attn.v_bias"]), sd[pfx + "attn.v_bias"]])
            qkv = b.linear(win, sd[pfx + "attn.qkv.weight"], qkv_bias, f"s{s}b{j}_qkv")
            q, k, v = [b.transpose(b.reshape(b.slice(qkv, [i * C], [(i + 1) * C], [2]), [-1, N, heads, d]), [0, 2, 1, 3]) for i in range(3)]
            eps = b.f32(1e-12)
            qn = b.div(q, b.op("Max", [b.op("ReduceL2", [q], axes=[-1], keepdims=1), eps]))      # F.normalize: x / max(||x||, eps)
            kn = b.div(k, b.op("Max", [b.op("ReduceL2", [k], axes=[-1], keepdims=1), eps]))
            attn = b.matmul(qn, b.transpose(kn, [0, 1, 3, 2]))
            scale = torch.clamp(sd[pfx + "attn.logit_scale"], max=math.log(1.0 / 0.01)).exp().reshape(1, heads, 1, 1)
            attn = b.mul(attn, b.weight(scale, f"s{s}b{j}_scale"))
            attn = b.add(attn, b.weight(_cpb_bias(sd, pfx + "attn.", ws, A["pretrained"][s], heads).unsqueeze(0), f"s{s}b{j}_cpb"))
            if shift > 0:
                mask = _shift_mask(res, ws, shift).reshape(1, nw * nw, 1, N, N)
                attn = b.reshape(b.add(b.reshape(attn, [-1, nw * nw, heads, N, N]), b.weight(mask, f"s{s}b{j}_mask")), [-1, heads, N, N])
            attn = b.op("Softmax", [attn], axis=-1)
            o = b.reshape(b.transpose(b.matmul(attn, v), [0, 2, 1, 3]), [-1, N, C])
            o = b.linear(o, sd[pfx + "attn.proj.weight"], sd[pfx + "attn.proj.bias"], f"s{s}b{j}_proj")
            o = b.reshape(b.transpose(b.reshape(o, [-1, nw, nw, ws, ws, C]), [0, 1, 3, 2, 4, 5]), [-1, res, res, C])
            o = b.reshape(b.roll2(o, shift, res), [-1, L, C])
            # res-post-norm block: x + LN(attn(x)); then + LN(mlp(.))
            t = b.add(t, b.layer_norm(o, sd[pfx + "norm1.weight"], sd[pfx + "norm1.bias"], 1e-5, f"s{s}b{j}_n1"))
            m = b.gelu(b.linear(t, sd[pfx + "mlp.fc1.weight"], sd[pfx + "mlp.fc1.bias"], f"s{s}b{j}_fc1"))
            m = b.linear(m, sd[pfx + "mlp.fc2.weight"], sd[pfx + "mlp.fc2.bias"], f"s{s}b{j}_fc2")
            t = b.add(t, b.layer_norm(m, sd[pfx + "norm2.weight"], sd[pfx + "norm2.bias"], 1e-5, f"s{s}b{j}_n2"))
            if j == A["hooks"][s]:   # forward hooks + act_postprocess: Transpose(1, 2) + Unflatten (backbones/swin_common.py:38-52)
                feats.append(b.reshape(b.transpose(t, [0, 2, 1]), [-1, C, res, res]))
        if s < len(A["depths"]) - 1:   # PatchMerging: 2x2 gather, Linear(4C, 2C), LayerNorm
            pfx = f"{ENC}layers.{s}.downsample."
            h = b.reshape(t, [-1, res, res, C])
            parts = [b.slice(h, [y0, x0], [res, res], [1, 2], [2, 2]) for (y0, x0) in ((0, 0), (1, 0), (0, 1), (1, 1))]
            h = b.reshape(b.concat(parts, 3), [-1, res * res // 4, 4 * C])
            t = b.layer_norm(b.linear(h, sd[pfx + "reduction.weight"], None, f"merge{s}"), sd[pfx + "norm.weight"], sd[pfx + "norm.bias"], 1e-5, f"merge{s}_n")
            res //= 2

    return feats


HYBRID = dict(img=384, patch=16, embed=768, depth=12, heads=12, layers=(3, 4, 9), hooks=(0, 1, 8, 11), features=(256, 512, 768, 768))   # model/dpt.py:86, model/blocks.py:103-112


def _same_pads(i: int, k: int, s: int) -> List[int]:
    """timm pad_same / TF 'SAME': total = max((ceil(i / s) - 1) s + k - i, 0), the odd pixel goes bottom / right.  ONNX order [top, left, bottom, right]."""
    t = max((math.ceil(i / s) - 1) * s + k - i, 0)
    return [t // 2, t // 2, t - t // 2, t - t // 2]


def _std_weight(w: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    """timm StdConv2dSame: per-output-channel weight standardisation (biased variance, evaluated through F.batch_norm like timm does).  A function of
    the weights alone: folded into the initializer, like do_constant_folding would."""
    return torch.nn.functional.batch_norm(w.reshape(1, w.shape[0], -1), None, None, training=True, momentum=0.0, eps=eps).reshape_as(w)


def _hybrid_encoder(b: GraphBuilder, sd, A) -> List[str]:
    """timm vit_base_resnet50_384 as backbones/vit.py:147-258 wires it: ResNetV2 (3, 4, 9) stem / stages (StdConv2dSame + GroupNormAct(32) + MaxPool2dSame),
    1x1 projection, class token + position embedding, 12 pre-norm ViT-B blocks; hooks on stages 0 / 1 and blocks 8 / 11; act_postprocess3 / 4 =
    ProjectReadout + Transpose + Unflatten + Conv1x1 (+ Conv3x3 stride 2) (backbones/utils.py:27-40,84-133).  Returns the four NCHW maps."""
    M = "depth_net.pretrained.model."
    BB = M + "patch_embed.backbone."
    S, E, heads = A["img"], A["embed"], A["heads"]

    def std_conv(x, key, hint, size, stride=1):
        w = _std_weight(sd[key])
        k = int(w.shape[-1])
        pads = [(k - 1) // 2] * 4 if stride == 1 else _same_pads(size, k, stride)     # static symmetric padding at stride 1, dynamic SAME padding otherwise
        return b.conv(x, w, None, hint, stride=stride, pads=pads)

    def gn(x, C, size, key, hint, relu=True):
        y = b.group_norm(x, C, size, size, sd[key + ".weight"], sd[key + ".bias"], 32, 1e-5, hint)
        return b.relu(y) if relu else y

    size = S // 2
    y = gn(std_conv("input", BB + "stem.conv.weight", "stem", S, 2), 64, size, BB + "stem.norm", "stem_gn")
    y = b.op("MaxPool", [y], kernel_shape=[3, 3], strides=[2, 2], pads=_same_pads(size, 3, 2))    # MaxPool2dSame: the padding is -inf, as in ONNX's MaxPool
    size //= 2
    stages = []
    cin = 64
    for s, depth in enumerate(A["layers"]):
        cout, mid = 256 << s, 64 << s
        for j in range(depth):
            p, hint = f"{BB}stages.{s}.blocks.{j}.", f"rn{s}b{j}"
            stride = 2 if (s > 0 and j == 0) else 1
            osz = size // stride
            sc = y
            if p + "downsample.conv.weight" in sd:
                sc = gn(std_conv(y, p + "downsample.conv.weight", hint + "_ds", size, stride), cout, osz, p + "downsample.norm", hint + "_dsn", relu=False)
            t = gn(std_conv(y, p + "conv1.weight", hint + "_c1", size), mid, size, p + "norm1", hint + "_n1")
            t = gn(std_conv(t, p + "conv2.weight", hint + "_c2", size, stride), mid, osz, p + "norm2", hint + "_n2")
            t = gn(std_conv(t, p + "conv3.weight", hint + "_c3", osz), cout, osz, p + "norm3", hint + "_n3", relu=False)
            y = b.relu(b.add(t, sc))
            size, cin = osz, cout
        stages.append(y)
    g = S // A["patch"]
    assert size == g, (size, g)
    # tokens: 1x1 projection, flatten(2).transpose(1, 2), class token in front, position embedding (resized like backbones/vit.py:23-41 when the grid differs)
    t = b.conv(stages[-1], sd[M + "patch_embed.proj.weight"], sd[M + "patch_embed.proj.bias"], "pe")
    t = b.transpose(b.reshape(t, [0, E, -1]), [0, 2, 1])
    pos = sd[M + "pos_embed"]
    g_old = int(math.sqrt(pos.shape[1] - 1))
    if g_old != g:
        grid = torch.nn.functional.interpolate(pos[0, 1:].reshape(1, g_old, g_old, -1).permute(0, 3, 1, 2), size=(g, g), mode="bilinear")
        pos = torch.cat([pos[:, :1], grid.permute(0, 2, 3, 1).reshape(1, g * g, -1)], dim=1)
    batch = b.op("Gather", [b.op("Shape", [t]), b.i64([0])], axis=0)
    cls = b.op("Expand", [b.weight(sd[M + "cls_token"], "cls"), b.concat([batch, b.i64([1, E])], 0)])
    t = b.add(b.concat([cls, t], 1), b.weight(pos, "pos_embed"))
    N, d = g * g + 1, E // heads
    hooked = {}
    for i in range(A["depth"]):
        p, hint = f"{M}blocks.{i}.", f"vit{i}"
        h = b.layer_norm(t, sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6, hint + "_n1")
        qkv = b.linear(h, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"], hint + "_qkv")
        q, k, v = [b.transpose(b.reshape(b.slice(qkv, [c * E], [(c + 1) * E], [2]), [0, N, heads, d]), [0, 2, 1, 3]) for c in range(3)]
        a = b.op("Softmax", [b.mul(b.matmul(q, b.transpose(k, [0, 1, 3, 2])), b.f32(d ** -0.5))], axis=-1)
        o = b.reshape(b.transpose(b.matmul(a, v), [0, 2, 1, 3]), [0, N, E])
        t = b.add(t, b.linear(o, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"], hint + "_proj"))
        h = b.layer_norm(t, sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6, hint + "_n2")
        h = b.gelu(b.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"], hint + "_fc1"))
        t = b.add(t, b.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"], hint + "_fc2"))
        if i in A["hooks"][2:]:
            hooked[i] = t           # (the final model.norm is dead on this path: both hooks fire before it)
    feats = [stages[0], stages[1]]  # act_postprocess1 / 2 are identities (backbones/vit.py:179-182)
    for n, hk in ((3, A["hooks"][2]), (4, A["hooks"][3])):
        ap = f"depth_net.pretrained.act_postprocess{n}."
        x = hooked[hk]
        tok = b.slice(x, [1], [N], [1])
        readout = b.op("Expand", [b.slice(x, [0], [1], [1]), b.op("Shape", [tok])])
        f = b.gelu(b.linear(b.concat([tok, readout], 2), sd[ap + "0.project.0.weight"], sd[ap + "0.project.0.bias"], f"ro{n}"))
        f = b.reshape(b.transpose(f, [0, 2, 1]), [0, E, g, g])
        f = b.conv(f, sd[ap + "3.weight"], sd[ap + "3.bias"], f"pp{n}_c1")
        if n == 4:
            f = b.conv(f, sd[ap + "4.weight"], sd[ap + "4.bias"], "pp4_c3", stride=2, pad=1)
        feats.append(f)
    return feats


def build_graph(net, opset: int = 13) -> P.Model:
    """The ONNX model of `net` (a soccdpt_amd SOccDPT_V3 with a Swin-V2 or the ViT-hybrid backbone), weights taken from its state dict."""
    backbone = net._engine_backbone()
    if backbone not in SWIN and backbone != "vitb_rn50_384":
        raise NotImplementedError(f"ONNX export is written for {list(SWIN) + ['vitb_rn50_384']}; got {backbone}")
    A = SWIN.get(backbone, HYBRID)
    sd = {k: v.detach().cpu().float() for k, v in net.state_dict().items()}
    SCR = "depth_net.scratch."
    b = GraphBuilder()
    S = A["img"]
    b.g.inputs.append(P.ValueInfo("input", P.FLOAT, ["batch_size", 3, S, S]))
    feats = _hybrid_encoder(b, sd, A) if backbone == "vitb_rn50_384" else _swin_encoder(b, sd, A)

    # ---------------- decoder (model/dpt.py:152-182, model/blocks.py:391-497) ----------------
    def rcu(x, pfx, hint):
        o = b.conv(b.relu(x), sd[pfx + "conv1.weight"], sd[pfx + "conv1.bias"], hint + "_c1", pad=1)
        o = b.conv(b.relu(o), sd[pfx + "conv2.weight"], sd[pfx + "conv2.bias"], hint + "_c2", pad=1)
        return b.add(o, x)

    def fusion(xs, r):
        pfx = f"{SCR}refinenet{r}."
        o = xs[0]
        if len(xs) == 2:
            o = b.add(o, rcu(xs[1], pfx + "resConfUnit1.", f"ref{r}_rcu1"))
        o = rcu(o, pfx + "resConfUnit2.", f"ref{r}_rcu2")
        o = b.resize(o, 2.0, 2.0, "bilinear_ac")     # every level of these backbones doubles (size= and scale_factor=2 coincide)
        return b.conv(o, sd[pfx + "out_conv.weight"], sd[pfx + "out_conv.bias"], f"ref{r}_oc")

    lr = [b.conv(feats[i], sd[f"{SCR}layer{i + 1}_rn.weight"], None, f"layer{i + 1}_rn", pad=1) for i in range(4)]
    p4 = fusion([lr[3]], 4)
    p3 = fusion([p4, lr[2]], 3)
    p2 = fusion([p3, lr[1]], 2)
    p1 = fusion([p2, lr[0]], 1)
    h = b.conv(p1, sd[SCR + "output_conv.0.weight"], sd[SCR + "output_conv.0.bias"], "d0", pad=1)
    h = b.resize(h, 2.0, 2.0, "bilinear_ac")
    h = b.relu(b.conv(h, sd[SCR + "output_conv.2.weight"], sd[SCR + "output_conv.2.bias"], "d2", pad=1))
    inv = b.relu(b.conv(h, sd[SCR + "output_conv.4.weight"], sd[SCR + "output_conv.4.bias"], "d4"))       # [B, 1, S, S] (non_negative=True)
    # seg head (model/SOccDPT.py:660-674), eval mode: BatchNorm on its running statistics, Dropout inactive
    g = b.conv(p1, sd["seg_head.0.weight"], None, "s0", pad=1)
    g = b.op("BatchNormalization", [g, b.weight(sd["seg_head.1.weight"], "bn_g"), b.weight(sd["seg_head.1.bias"], "bn_b"),
                                    b.weight(sd["seg_head.1.running_mean"], "bn_m"), b.weight(sd["seg_head.1.running_var"], "bn_v")], epsilon=1e-5)
    g = b.conv(b.relu(g), sd["seg_head.4.weight"], sd["seg_head.4.bias"], "s4")
    g = b.resize(g, 2.0, 2.0, "bilinear_ac")
    seg = b.op("Sigmoid", [g]) if net.sigmoid else b.add(b.mul(b.op("Tanh", [g]), b.f32(0.5)), b.f32(0.5))   # ScaledTanh (model/scaled_tanh.py)

    # ---------------- get_semantic_occupancy without the voxel grid (model/SOccDPT.py:264-364) ----------------
    Hc, Wc = int(net.height), int(net.width)
    inv_up = b.resize_to(inv, Hc, Wc, "bicubic")                         # F.interpolate(..., size=(height, width), mode="bicubic", align_corners=False)
    seg_up = b.resize_to(seg, Hc, Wc, "nearest")
    inv_up = b.reshape(inv_up, [-1, Hc, Wc])
    small = b.f32(1e-8)
    inv_up = b.op("Where", [b.op("Less", [inv_up, small]), small, inv_up])                               # inv[inv < 1e-8] = 1e-8 (NaN stays NaN)
    depth = b.op("Reciprocal", [inv_up])
    depth = b.op("Where", [b.op("Or", [b.op("IsInf", [depth]), b.op("IsNaN", [depth])]), b.f32(float("inf")), depth])
    V = b.init(np.arange(Wc, dtype=np.float32).reshape(1, 1, Wc), "V")
    U = b.init(np.arange(Hc, dtype=np.float32).reshape(1, Hc, 1), "U")
    X = b.div(b.mul(b.sub(V, b.f32(np.float32(net.cx))), depth), b.f32(np.float32(net.fx)))
    Y = b.div(b.mul(b.sub(U, b.f32(np.float32(net.cy))), depth), b.f32(np.float32(net.fy)))
    un = lambda x: b.op("Unsqueeze", [x, b.i64([3])])
    pts = b.concat([un(X), un(Y), un(depth)], 3)                                                         # torch.stack([X, Y, Z], dim=3)
    # the reference's quirk (:351-353): pc_scale / pc_shift hit POINTS 0, 1, 2 of every image (all three coordinates), nothing else
    sc_row = np.ones((1, 1, Wc, 1), np.float32); sh_row = np.zeros((1, 1, Wc, 1), np.float32)
    for n in range(3):
        sc_row[0, 0, n, 0] = np.float32(net.pc_scale[n]); sh_row[0, 0, n, 0] = np.float32(net.pc_shift[n])
    row0 = np.zeros((1, Hc, 1, 1), np.float32); row0[0, 0] = 1.0
    scale_map = b.add(b.mul(b.init(row0, "row0"), b.init(sc_row - 1.0, "pc_scale_m1")), b.f32(1.0))
    shift_map = b.mul(b.init(row0, "row0"), b.init(sh_row, "pc_shift"))
    pts = b.add(b.mul(pts, scale_map), shift_map)

    for src, name, shape in ((inv_up, "output", ["batch_size", Hc, Wc]), (seg_up, "segmentation", ["batch_size", int(net.num_classes), Hc, Wc]),
                             (pts, "points", ["batch_size", Hc, Wc, 3])):
        b.g.nodes.append(P.Node("Identity", [src], [name], name=name))
        b.g.outputs.append(P.ValueInfo(name, P.FLOAT, shape))
    model = P.Model(b.g, opset=opset)
    # names of intermediate tensors (not serialised): lets a checker feed the hooked feature maps / read the half-resolution outputs
    model.tensor_names = {**{f"feat{i}": feats[i] for i in range(4)}, "path1": p1, "inv256": inv, "seg256": seg}
    return model


def export(net, export_path: str) -> P.Model:
    model = build_graph(net)
    d = os.path.dirname(export_path)
    if d:
        os.makedirs(d, exist_ok=True)
    P.save(model, export_path)
    return model


def build_parser() -> argparse.ArgumentParser:
    from ..model.SOccDPT import model_types
    p = argparse.ArgumentParser(description="Export SOccDPT to ONNX")
    p.add_argument("-v", "--version", choices=[1, 2, 3], required=True, type=int, help="SOccDPT version")
    p.add_argument("-dt", "--dataset", choices=["bdd", "idd", "idd+bdd"], required=True, help="Dataset (the reference traces one of its frames; unused here)")
    p.add_argument("-t", "--model_type", choices=model_types, required=True, help="Model architecture to use")
    p.add_argument("-d", "--device", default="cpu", help="unused: the graph is written from the weights, nothing runs")
    p.add_argument("-l", "--load", default=None, help="Load model from a .pth file")
    p.add_argument("-ld", "--load_depth", default=None, help="Load depth model from a .pth file")
    p.add_argument("-ls", "--load_seg", default=None, help="(V1 only in the reference)")
    p.add_argument("-b", "--base_path", default=os.path.expanduser("~/Datasets/Depth_Dataset_Bengaluru"))
    p.add_argument("-e", "--export_path", required=True, help="Path to export model")
    p.add_argument("--camera_intrinsics_yaml", default=None, help="calibration file (default: the reference's DEFAULT_CALIB)")
    return p


def main(args) -> int:
    from ..model.SOccDPT import SOccDPT_versions
    assert args.version == 3, "only SOccDPT_V3 is built (SURVEY.md section 2)"
    assert args.load_seg in (None, False), "V3 does not support loading seg"
    kw = dict(load_depth=args.load_depth if args.load_depth else False, model_type=args.model_type, path=args.load)
    if args.camera_intrinsics_yaml:
        kw["camera_intrinsics_yaml"] = args.camera_intrinsics_yaml
    net = SOccDPT_versions[3](**kw).eval()
    model = export(net, args.export_path)
    print(f"wrote {args.export_path}: {len(model.graph.nodes)} nodes, {len(model.graph.initializers)} initializers, opset {model.opset}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main(build_parser().parse_args()))
