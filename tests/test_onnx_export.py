"""CPU: the ONNX export (/root/reference/SOccDPT/scripts/export_SOccDPT.py:122-141).

The image has neither `onnx` nor `onnxruntime`, so three things are pinned separately:
  1. the WIRE FORMAT of soccdpt_amd.utils.onnx_proto against a real producer: torch's own C++ ONNX serialiser (reachable through torch.onnx internals
     without the `onnx` package) writes small graphs, the codec decodes / re-encodes them;
  2. the OPERATOR SEMANTICS of soccdpt_amd.utils.onnx_eval against torch: those torch-exported graphs (Conv, MatMul, LayerNorm decomposition, Erf-GELU,
     L2 normalisation, Softmax, roll = Slice + Concat, the three Resize flavours the SOccDPT graph uses) evaluate to what the torch modules return;
  3. the EXPORTED SOccDPT_V3 GRAPH, run by that evaluator: against the fp32 oracle end to end (dynamic batch: B = 1 and 2), and its decoder + heads
     against the fixture recorded from the REFERENCE's own modules (tests/golden/decoder_B1_*.npz)."""
import contextlib
import importlib
import io
import os
import tempfile
import warnings

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import soccdpt_ref as R
from soccdpt_amd.utils import onnx_eval as E
from soccdpt_amd.utils import onnx_proto as P


def _torch_export(module, x):
    """ONNX bytes of a small torch module from torch's own exporter internals (TorchScript tracer + C++ serialiser; no `onnx` package needed)."""
    warnings.filterwarnings("ignore")
    from torch.onnx._internal.torchscript_exporter import utils as TU
    from torch.onnx._internal.torchscript_exporter._globals import GLOBALS
    for v in range(9, 14):
        importlib.import_module(f"torch.onnx._internal.torchscript_exporter.symbolic_opset{v}")
    GLOBALS.export_onnx_opset_version = 13
    da = {"input": {0: "batch_size"}, "output": {0: "batch_size"}}
    with torch.no_grad():
        graph, params, _ = TU._model_to_graph(module.eval(), (x,), input_names=["input"], output_names=["output"], dynamic_axes=da)
    return graph._export_onnx(params, 13, da, False, torch._C._onnx.OperatorExportTypes.ONNX, True, False, {}, True, "", {})[0]


class _Interp(nn.Module):
    def __init__(self, mode, ac, size):
        super().__init__()
        self.mode, self.ac, self.size = mode, ac, size

    def forward(self, x):
        return F.interpolate(x, size=self.size, mode=self.mode, align_corners=self.ac)


class _Block(nn.Module):
    def __init__(self):
        super().__init__()
        self.l, self.n = nn.Linear(8, 16), nn.LayerNorm(16)

    def forward(self, x):
        y = self.n(F.gelu(self.l(x)))
        q = F.normalize(y, dim=-1)
        a = torch.softmax(q @ q.transpose(-2, -1), dim=-1)
        return torch.roll(a @ y, shifts=(1,), dims=(1,))


class _SamePool(nn.Module):   # timm MaxPool2dSame at an even size: one -inf row / column at the bottom / right
    def forward(self, x):
        return F.max_pool2d(F.pad(x, [0, 1, 0, 1], value=float("-inf")), 3, 2)


class _ClsTokens(nn.Module):   # class token expanded over a dynamic batch, read-out broadcast over the tokens (Shape / Gather / Expand / Concat)
    def __init__(self):
        super().__init__()
        self.cls = nn.Parameter(torch.randn(1, 1, 8))

    def forward(self, x):
        t = torch.cat((self.cls.expand(x.shape[0], -1, -1), x), dim=1)
        return torch.cat((t[:, 1:], t[:, 0].unsqueeze(1).expand_as(t[:, 1:])), -1)


_CASES = {
    "conv": (lambda: nn.Sequential(nn.Conv2d(3, 4, 3, padding=1), nn.ReLU(), nn.Conv2d(4, 2, 4, stride=4)), (2, 3, 8, 8)),
    "maxpool_padded": (lambda: nn.Sequential(nn.Conv2d(3, 4, 3, stride=2), nn.MaxPool2d(3, 2, padding=1)), (2, 3, 17, 17)),
    "maxpool_same": (_SamePool, (2, 3, 8, 8)),
    "cls_tokens": (_ClsTokens, (2, 5, 8)),
    "block": (_Block, (2, 5, 8)),
    "bicubic": (lambda: _Interp("bicubic", False, (9, 13)), (2, 3, 4, 5)),
    "bilinear_align_corners": (lambda: _Interp("bilinear", True, (8, 10)), (2, 3, 4, 5)),
    "nearest": (lambda: _Interp("nearest", None, (9, 13)), (2, 3, 4, 5)),
}


@pytest.mark.parametrize("case", sorted(_CASES))
def test_codec_and_evaluator_against_torchs_own_exporter(case):
    try:
        import torch.onnx._internal.torchscript_exporter.utils  # noqa: F401
    except Exception:
        pytest.skip("this torch build has no TorchScript ONNX exporter internals")
    torch.manual_seed(0)
    make, shape = _CASES[case]
    m, x = make(), torch.randn(*shape)
    pb = _torch_export(m, x)
    model = P.Model.decode(pb)
    assert model.ir_version == 7 and model.opset == 13
    assert model.graph.inputs[0].name == "input" and model.graph.inputs[0].shape[0] == "batch_size"
    assert model.graph.outputs[0].name == "output" and model.graph.outputs[0].shape[0] == "batch_size"
    y = E.run(model, {"input": x})[0]
    assert float((y - m(x)).abs().max()) < 1e-5
    # what the codec writes is what it read: decode(encode(decode(bytes))) evaluates identically, node for node
    again = P.Model.decode(model.encode())
    assert [(n.op_type, n.inputs, n.outputs, sorted(n.attrs)) for n in again.graph.nodes] == [(n.op_type, n.inputs, n.outputs, sorted(n.attrs)) for n in model.graph.nodes]
    assert torch.equal(E.run(again, {"input": x})[0], y)
    if "Resize" in [n.op_type for n in model.graph.nodes]:   # the attribute values the SOccDPT exporter writes are the ones torch writes
        from soccdpt_amd.scripts.export_SOccDPT import GraphBuilder
        b = GraphBuilder()
        b.resize("x", 2.0, 2.0, {"bicubic": "bicubic", "bilinear_align_corners": "bilinear_ac", "nearest": "nearest"}[case])
        mine = b.g.nodes[-1].attrs
        theirs = [n for n in model.graph.nodes if n.op_type == "Resize"][0].attrs
        for k, v in mine.items():
            assert theirs[k] == pytest.approx(v) if isinstance(v, float) else theirs[k] == v, (k, theirs[k], v)


@pytest.fixture(scope="module")
def exported():
    from soccdpt_amd.model.SOccDPT import SOccDPT_V3
    from soccdpt_amd.scripts.export_SOccDPT import export
    from soccdpt_amd.utils.synth import synth_state_dict, write_synth_calib
    tmp = tempfile.mkdtemp()
    calib = write_synth_calib(os.path.join(tmp, "calib.yaml"))
    with contextlib.redirect_stdout(io.StringIO()):
        net = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=False)
    sd = synth_state_dict(alias_pretrained=True)
    net.load_state_dict(sd, strict=False)
    path = os.path.join(tmp, "onnx", "SOccDPT.onnx")
    built = export(net.eval(), path)
    return path, built, sd


def test_exported_file_structure(exported):
    path, built, _ = exported
    model = P.load(path)                       # what a consumer reads back from disk
    g = model.graph
    assert model.opset == 13 and model.ir_version == 7
    assert [(v.name, v.shape) for v in g.inputs] == [("input", ["batch_size", 3, 256, 256])]
    assert [(v.name, v.shape) for v in g.outputs] == [("output", ["batch_size", 1080, 1920]), ("segmentation", ["batch_size", 3, 1080, 1920]),
                                                     ("points", ["batch_size", 1080, 1920, 3])]
    ops = {n.op_type for n in g.nodes}
    standard = {"Conv", "MatMul", "Add", "Sub", "Mul", "Div", "Sqrt", "Erf", "Tanh", "Relu", "Reciprocal", "Max", "Less", "Or", "IsInf", "IsNaN", "Where",
                "ReduceMean", "ReduceL2", "Softmax", "Reshape", "Transpose", "Concat", "Slice", "Unsqueeze", "Resize", "BatchNormalization", "Identity", "Shape"}
    assert ops <= standard, ops - standard      # opset-13 operators of the default domain only
    produced = {o for n in g.nodes for o in n.outputs} | {t.name for t in g.initializers} | {"input", ""}
    assert all(x in produced for n in g.nodes for x in n.inputs)       # no dangling edges
    n_weights = sum(t.array.size for t in g.initializers)
    assert 38e6 < n_weights < 50e6                                      # the 42 M parameters + folded bias tables / masks
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "soccdpt_amd", "scripts", "export_SOccDPT.py")).read()
    assert "oracle" not in src.replace("fp32 oracle", "")              # the writer does not route through the test oracle


def test_camera_size_that_is_not_a_float32_multiple_of_the_network_size(tmp_path):
    """Resize by `scales` = 1241 / 256 gives floor(256 * float32(4.84765625...)) -- exact here, but not for every width; the exporter now writes the
    `sizes` input (size=(height, width) as the reference calls F.interpolate, model/SOccDPT.py:262-285), so the outputs have the calibration's size whatever it is."""
    from soccdpt_amd.model.SOccDPT import SOccDPT_V3
    from soccdpt_amd.scripts.export_SOccDPT import export
    from soccdpt_amd.utils.synth import synth_input, synth_state_dict, write_synth_calib
    calib = write_synth_calib(str(tmp_path / "calib.yaml"), **{"Camera.width": 1241, "Camera.height": 377})
    with contextlib.redirect_stdout(io.StringIO()):
        net = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=False)
    sd = synth_state_dict(alias_pretrained=True)
    net.load_state_dict(sd, strict=False)
    path = str(tmp_path / "odd.onnx")
    export(net.eval(), path)
    model = P.load(path)
    rs = [n for n in model.graph.nodes if n.op_type == "Resize"][-2:]
    assert all(n.inputs[1] == "" and n.inputs[2] == "" and n.inputs[3] for n in rs)        # sizes, not scales
    assert [(v.name, v.shape[1:]) for v in model.graph.outputs] == [("output", [377, 1241]), ("segmentation", [3, 377, 1241]), ("points", [377, 1241, 3])]
    torch.set_num_threads(8)
    x = synth_input(1, seed0=2)
    inv, seg, pts = E.run(model, {"input": x})
    assert tuple(inv.shape) == (1, 377, 1241) and tuple(seg.shape) == (1, 3, 377, 1241) and tuple(pts.shape) == (1, 377, 1241, 3)


@pytest.mark.parametrize("B", [1, 2])
def test_exported_graph_matches_the_fp32_oracle(exported, B):
    from soccdpt_amd.utils.synth import synth_input
    path, _, sd = exported
    model = P.load(path)
    x = synth_input(B, seed0=4)
    torch.set_num_threads(8)
    inv, seg, pts = E.run(model, {"input": x})
    o_inv, o_seg, o_pts, _ = R.soccdpt_v3_forward(sd, x, sigmoid=False, compute_occ=False)
    if B == 1:      # the reference squeezes the batch dimension of the segmentation away at B = 1 (model/SOccDPT.py:276-285); the graph keeps it (dynamic batch)
        o_seg = o_seg.reshape(1, 3, 1080, 1920)
    rel = lambda a, b: float((a - b).norm() / b.norm())
    assert tuple(inv.shape) == (B, 1080, 1920) and tuple(seg.shape) == (B, 3, 1080, 1920) and tuple(pts.shape) == (B, 1080, 1920, 3)
    assert rel(inv, o_inv) < 1e-5 and rel(seg, o_seg) < 1e-4
    fin = torch.isfinite(o_pts)
    assert torch.equal(torch.isfinite(pts), fin) and rel(pts[fin], o_pts[fin]) < 1e-5
    assert torch.allclose(pts[:, 0, :3], o_pts[:, 0, :3], rtol=1e-5)        # the three pc_scale / pc_shift pixels of the reference's quirk


@pytest.mark.parametrize("name", ["tanh"])
def test_exported_decoder_and_heads_match_the_reference_golden(exported, golden_dir, name):
    """Feed the hooked feature maps the fixture was recorded with (tests/golden_inputs.decoder_features) into the exported graph: inverse depth and
    class probabilities at network resolution equal what the REFERENCE's own DPT decoder / depth head / seg head returned (oracle/make_golden.py)."""
    from tests.golden_inputs import decoder_features
    _, built, _ = exported
    g = np.load(os.path.join(golden_dir, f"decoder_B1_{name}.npz"))
    feats = decoder_features()
    names = built.tensor_names
    inv, seg = E.run(built, {names[f"feat{i}"]: feats[i] for i in range(4)}, outputs=[names["inv256"], names["seg256"]])
    assert float((inv[:, 0] - torch.from_numpy(g["inv256"])).abs().max()) < 1e-4 * float(np.abs(g["inv256"]).max())
    assert float((seg - torch.from_numpy(g["seg256"])).abs().max()) < 1e-5


def _export_other(model_type, backbone):
    from soccdpt_amd.model.SOccDPT import SOccDPT_V3
    from soccdpt_amd.scripts.export_SOccDPT import build_graph
    from soccdpt_amd.utils.synth import synth_state_dict, write_synth_calib
    calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
    with contextlib.redirect_stdout(io.StringIO()):
        net = SOccDPT_V3(sigmoid=False, load_depth=False, camera_intrinsics_yaml=calib, compute_occ=False, model_type=model_type)
    sd = synth_state_dict(backbone, alias_pretrained=True)
    net.load_state_dict(sd, strict=False)
    return build_graph(net.eval()), sd


@pytest.mark.parametrize("model_type,backbone", [("dpt_hybrid_384", "vitb_rn50_384"), ("dpt_swin2_base_384", "swin2b24_384")])
def test_exported_graphs_of_the_384_models_match_the_fp32_oracle(model_type, backbone):
    """The other two model types of the reference's config files: the ViT-hybrid graph (weight-standardised SAME convolutions folded into the initializers,
    GroupNorm decomposition, MaxPool with -inf SAME padding, class token / read-out over a dynamic batch) and the Swin-V2 base graph (24 x 24 / 12 x 12
    windows, log-spaced CPB with pretrained window sizes), evaluated on CPU against the fp32 oracle: the four hooked maps and the network outputs."""
    from soccdpt_amd.utils.synth import synth_input
    built, sd = _export_other(model_type, backbone)
    again = P.Model.decode(built.encode())                     # through the wire format, like a file on disk
    x = synth_input(1, size=384, seed0=8)
    torch.set_num_threads(8)
    names = built.tensor_names
    outs = E.run(again, {"input": x}, outputs=[names[f"feat{i}"] for i in range(4)] + [names["inv256"], names["seg256"]])
    with torch.no_grad():
        layers = R.hybrid_encoder(sd, x) if backbone == "vitb_rn50_384" else R.swin_encoder(sd, x, R.ARCHS[backbone])
        o_inv, o_p1 = R.dpt_decoder(sd, layers)
        o_seg = R.seg_head(sd, o_p1, sigmoid=False)
    rel = lambda a, b: float((a - b).norm() / b.norm())
    for i in range(4):
        assert tuple(outs[i].shape) == tuple(layers[i].shape) and rel(outs[i], layers[i]) < 2e-5, (i, rel(outs[i], layers[i]))
    assert rel(outs[4][:, 0], o_inv) < 5e-5 and rel(outs[5], o_seg.reshape(outs[5].shape)) < 2e-4


def test_export_script_takes_the_reference_flags(tmp_path):
    """`python -m soccdpt_amd.scripts.export_SOccDPT -v 3 -dt bdd -t <model_type> -e <path>` (scripts/export_SOccDPT.py of the reference: same flags) writes a
    file the codec reads back; versions 1 / 2 are refused like everywhere else in this build."""
    from soccdpt_amd.scripts.export_SOccDPT import build_parser, main
    from soccdpt_amd.utils.synth import write_synth_calib
    calib = write_synth_calib(str(tmp_path / "calib.yaml"))
    out = str(tmp_path / "onnx" / "m.onnx")
    with contextlib.redirect_stdout(io.StringIO()) as log:
        assert main(build_parser().parse_args(["-v", "3", "-dt", "bdd", "-t", "dpt_swin2_tiny_256", "-e", out, "--camera_intrinsics_yaml", calib])) == 0
    assert "nodes" in log.getvalue()
    m = P.load(out)
    assert [v.name for v in m.graph.outputs] == ["output", "segmentation", "points"] and m.opset == 13
    with pytest.raises(AssertionError):
        main(build_parser().parse_args(["-v", "2", "-dt", "bdd", "-t", "dpt_swin2_tiny_256", "-e", out, "--camera_intrinsics_yaml", calib]))
    with pytest.raises(SystemExit):
        build_parser().parse_args(["-v", "3", "-dt", "bdd", "-t", "dpt_swin2_tiny_256"])       # -e is required, as in the reference
