"""Generate tests/golden/*.npz (TEST INFRASTRUCTURE; runs only in the build container).

What it does
------------
1. Imports the reference's OWN decoder / heads / projection code from
   /root/reference (SURVEY.md Appendix A recipe: empty stand-ins for the missing
   third-party modules timm / cv2 / torchvision are placed in sys.modules so the
   reference's import statements succeed; none of the stand-ins is ever called on
   the path exercised here).
2. Checks that oracle/soccdpt_ref.py reproduces the reference bit-for-bit on CPU
   for (a) the projection and (b) decoder + heads + projection with the synthetic
   state dict loaded through the reference's own load_state_dict (which also pins
   the checkpoint key layout).
3. Cross-checks the Swin-V2 encoder restatement against HF transformers'
   Swinv2Model (independent port; the timm original is not installable) —
   the encoder stays "parity unpinned".
4. Writes small fixtures: inputs are regenerated from seeds by the tests, only
   expected outputs are stored.

Only data (inputs/expected outputs) is written; no reference source travels.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import tempfile
import types
import zlib

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from oracle import soccdpt_ref as R  # noqa: E402
from soccdpt_amd.utils.synth import synth_state_dict, synth_input, write_synth_calib  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")


from tests.golden_inputs import proj_inputs, decoder_features, metrics_inputs  # noqa: E402


def sha(t: torch.Tensor) -> str:
    return hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest()


def pack_occ(occ: torch.Tensor) -> np.ndarray:
    """[256,256,32,3] {0,1} float -> uint32 words, bit index == linear index."""
    flat = occ.reshape(-1).numpy() != 0
    return np.packbits(flat, bitorder="little").view(np.uint32)


# -- reference import (stub recipe) --
def import_reference():
    def stub(name, **a):
        m = types.ModuleType(name)
        m.__dict__.update(a)
        sys.modules[name] = m
    stub("timm", create_model=None)
    stub("timm.models")
    stub("timm.models.layers", get_act_layer=lambda n: None)
    stub("timm.models.beit", gen_relative_position_index=lambda ws: None)
    stub("cv2", INTER_AREA=3, INTER_CUBIC=2, INTER_NEAREST=0, COLOR_BGR2RGB=4)
    stub("torchvision")
    stub("torchvision.transforms", Compose=lambda l: l)
    stub("wandb")
    stub("matplotlib", colormaps={"viridis": (lambda v: v)})
    sys.path.insert(0, "/root/reference")
    from SOccDPT.model import SOccDPT as S, dpt, blocks
    from SOccDPT.model.backbones.swin_common import _make_swin_backbone
    return S, dpt, blocks, _make_swin_backbone


class _FixedBlock(torch.nn.Module):
    """Emits a fixed [B, L, C] tensor; stands in for a hooked timm block."""
    def __init__(self):
        super().__init__()
        self.value = None

    def forward(self, x):
        return self.value


class _FixedEncoder(torch.nn.Module):
    def __init__(self, depths=(2, 2, 6, 2)):
        super().__init__()
        self.layers = torch.nn.ModuleList()
        for d in depths:
            st = torch.nn.Module()
            st.blocks = torch.nn.ModuleList([_FixedBlock() for _ in range(d)])
            self.layers.append(st)
        self.hooks = (1, 1, 5, 1)

    def set_features(self, feats):
        for i, f in enumerate(feats):
            B, C, H, W = f.shape
            self.layers[i].blocks[self.hooks[i]].value = f.reshape(B, C, H * W).transpose(1, 2).contiguous()

    def forward_features(self, x):
        for i, st in enumerate(self.layers):
            st.blocks[self.hooks[i]](x)
        return x


class _StdConv(torch.nn.Module):
    """Holds a StdConv2dSame weight under timm's name; arithmetic = oracle restatement (R.std_conv_same)."""
    def __init__(self, cin, cout, k, stride=1):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.zeros(cout, cin, k, k))
        self.stride = stride

    def forward(self, x):
        return R.std_conv_same(x, self.weight, self.stride)


class _GN(torch.nn.Module):
    def __init__(self, c, relu=True):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.ones(c))
        self.bias = torch.nn.Parameter(torch.zeros(c))
        self.relu = relu

    def forward(self, x):
        return R.group_norm_act(x, self.weight, self.bias, self.relu)


class _Bottleneck(torch.nn.Module):
    def __init__(self, cin, cout, stride, proj):
        super().__init__()
        mid = cout // 4
        if proj:
            self.downsample = torch.nn.Module()
            self.downsample.conv = _StdConv(cin, cout, 1, stride)
            self.downsample.norm = _GN(cout, relu=False)
        self.conv1, self.norm1 = _StdConv(cin, mid, 1), _GN(mid)
        self.conv2, self.norm2 = _StdConv(mid, mid, 3, stride), _GN(mid)
        self.conv3, self.norm3 = _StdConv(mid, cout, 1), _GN(cout, relu=False)
        self.proj = proj

    def forward(self, x):
        sc = self.downsample.norm(self.downsample.conv(x)) if self.proj else x
        y = self.norm1(self.conv1(x))
        y = self.norm2(self.conv2(y))
        y = self.norm3(self.conv3(y))
        return torch.nn.functional.relu(y + sc)


class _Stage(torch.nn.Module):
    def __init__(self, cin, cout, depth, stride):
        super().__init__()
        self.blocks = torch.nn.Sequential(*[_Bottleneck(cin if j == 0 else cout, cout, stride if j == 0 else 1, j == 0) for j in range(depth)])

    def forward(self, x):
        return self.blocks(x)


class _ResNetV2(torch.nn.Module):
    """Module tree / names of timm ResNetV2(layers=(3,4,9), preact=False, stem_type='same') [3p-recall]; arithmetic from the oracle."""
    def __init__(self):
        super().__init__()
        self.stem = torch.nn.Module()
        self.stem.conv, self.stem.norm = _StdConv(3, 64, 7, 2), _GN(64)
        self.stages = torch.nn.Sequential(_Stage(64, 256, 3, 1), _Stage(256, 512, 4, 2), _Stage(512, 1024, 9, 2))

    def forward(self, x):
        y = self.stem.norm(self.stem.conv(x))
        y = torch.nn.functional.max_pool2d(R.pad_same(y, 3, 2, value=float("-inf")), 3, 2)
        return self.stages(y)


class _VitBlock(torch.nn.Module):
    def __init__(self, E=768, heads=12):
        super().__init__()
        self.heads = heads
        self.norm1 = torch.nn.LayerNorm(E, eps=1e-6)
        self.attn = torch.nn.Module()
        self.attn.qkv, self.attn.proj = torch.nn.Linear(E, 3 * E), torch.nn.Linear(E, E)
        self.norm2 = torch.nn.LayerNorm(E, eps=1e-6)
        self.mlp = torch.nn.Module()
        self.mlp.fc1, self.mlp.fc2 = torch.nn.Linear(E, 4 * E), torch.nn.Linear(4 * E, E)

    def forward(self, x):
        sd = {"b." + k: v for k, v in self.state_dict().items()}
        return R.vit_block(sd, "b.", x, self.heads)


class _HybridViT(torch.nn.Module):
    """Stand-in for timm's vit_base_resnet50_384 VisionTransformer: the attributes the reference's forward_flex touches
    (backbones/vit.py:44-85) under timm's parameter names, arithmetic from the oracle restatement."""
    def __init__(self):
        super().__init__()
        self.cls_token = torch.nn.Parameter(torch.zeros(1, 1, 768))
        self.pos_embed = torch.nn.Parameter(torch.zeros(1, 577, 768))
        self.patch_embed = torch.nn.Module()
        self.patch_embed.backbone = _ResNetV2()
        self.patch_embed.proj = torch.nn.Conv2d(1024, 768, 1)
        self.pos_drop = torch.nn.Identity()
        self.blocks = torch.nn.Sequential(*[_VitBlock() for _ in range(12)])
        self.norm = torch.nn.LayerNorm(768, eps=1e-6)
        self.head = torch.nn.Linear(768, 1000)
        self.no_embed_class = False
        self.dist_token = None


def make_reference_hybrid_backbone(vit_mod, utils_mod):
    """What backbones/vit.py:147-241 (_make_vit_b_rn50_backbone) evidently means to build -- the snapshot's lines 181-182 / 222-223
    assign the Sequential to `_` and then exec("...=value") (NameError), upstream MiDaS has `value = nn.Sequential(...)` -- assembled
    from the reference's OWN pieces: get_activation hooks, get_readout_oper("project"), Transpose, forward_flex, _resize_pos_embed."""
    nn = torch.nn
    features, vit_features, size, hooks, number_stages = [256, 512, 768, 768], 768, [384, 384], [0, 1, 8, 11], 2
    pretrained = nn.Module()
    pretrained.model = _HybridViT()
    for s in range(number_stages):
        pretrained.model.patch_embed.backbone.stages[s].register_forward_hook(utils_mod.get_activation(str(s + 1)))
    for s in range(number_stages, 4):
        pretrained.model.blocks[hooks[s]].register_forward_hook(utils_mod.get_activation(str(s + 1)))
    pretrained.activations = utils_mod.activations
    readout_oper = utils_mod.get_readout_oper(vit_features, features, "project", 1)
    for s in range(number_stages):
        setattr(pretrained, f"act_postprocess{s + 1}", nn.Sequential(nn.Identity(), nn.Identity(), nn.Identity()))
    for s in range(number_stages, 4):
        final_layer = nn.Conv2d(features[3], features[3], kernel_size=3, stride=2, padding=1) if s > number_stages else None
        layers = [readout_oper[s], utils_mod.Transpose(1, 2), nn.Unflatten(2, torch.Size([size[0] // 16, size[1] // 16])),
                  nn.Conv2d(vit_features, features[s], kernel_size=1, stride=1, padding=0)]
        if final_layer is not None:
            layers.append(final_layer)
        setattr(pretrained, f"act_postprocess{s + 1}", nn.Sequential(*layers))
    pretrained.model.start_index = 1
    pretrained.model.patch_size = [16, 16]
    pretrained.model.forward_flex = types.MethodType(vit_mod.forward_flex, pretrained.model)
    pretrained.model._resize_pos_embed = types.MethodType(vit_mod._resize_pos_embed, pretrained.model)
    return pretrained


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-hf", action="store_true")
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    S, dpt, blocks, _make_swin_backbone = import_reference()
    calib = write_synth_calib(os.path.join(tempfile.mkdtemp(), "calib.yaml"))
    report = {}

    # ---- (1) projection: reference vs oracle, bit-for-bit ----
    inv, seg = proj_inputs()
    ref_model = S.SOccDPT(camera_intrinsics_yaml=calib, compute_occ=True)
    with torch.no_grad():
        r_inv, r_seg, r_pts, r_occ = ref_model.get_semantic_occupancy(inv.clone(), seg.clone())
        o_inv, o_seg, o_pts, o_occ = R.project(inv.clone(), seg.clone())
    eq = lambda a, b: bool(torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0)))  # noqa: E731
    report["projection_bit_exact"] = dict(inv=eq(r_inv, o_inv), seg=eq(r_seg, o_seg), pts=eq(r_pts, o_pts), occ=eq(r_occ, o_occ))
    assert all(report["projection_bit_exact"].values()), report
    assert torch.equal(r_occ[0], r_occ[1])
    rows = [0, 1, 5, 100, 539, 540, 777, 1079]
    np.savez_compressed(
        os.path.join(GOLD, "projection_B2.npz"),
        seed=np.int64(1234),
        occ_bits=pack_occ(r_occ[0]),
        occ_count=np.int64(r_occ[0].sum().item()),
        rows=np.array(rows),
        inv_up_rows=r_inv[:, rows].numpy(),
        points_rows=r_pts[:, rows].numpy(),
        seg_up_rows=r_seg[:, :, rows].numpy(),
        inv_up_sha=np.array(sha(r_inv)), points_sha=np.array(sha(r_pts)), seg_up_sha=np.array(sha(r_seg)),
    )
    print("projection golden: occupied voxel-classes =", int(r_occ[0].sum().item()))

    # B = 1 shape quirk (model/SOccDPT.py:276-285): seg squeezed to [3,H,W]
    with torch.no_grad():
        r1 = ref_model.get_semantic_occupancy(inv[:1].clone(), seg[:1].clone())
        o1 = R.project(inv[:1].clone(), seg[:1].clone())
    assert [tuple(t.shape) for t in r1] == [tuple(t.shape) for t in o1]
    report["b1_shapes"] = [list(t.shape) for t in r1]

    # ---- (2) decoder + heads + projection through the reference's own modules ----
    enc = _FixedEncoder()
    dpt._make_encoder = lambda backbone, features, use_pretrained, groups=1, expand=False, exportable=True, hooks=None, \
        use_vit_only=False, use_readout="ignore", in_features=None: (
        _make_swin_backbone(enc, hooks=hooks, patch_grid=[64, 64]),
        blocks._make_scratch([96, 192, 384, 768], features, groups=groups, expand=expand))
    for sigmoid in (True, False):
        net = S.SOccDPT_V3(sigmoid=sigmoid, load_depth=False, path=None, camera_intrinsics_yaml=calib, compute_occ=True).eval()
        sd = synth_state_dict()
        dec_sd = {k: v for k, v in sd.items() if not k.startswith("depth_net.pretrained.")}
        missing = net.load_state_dict(dec_sd, strict=False)
        assert not missing.unexpected_keys, missing.unexpected_keys
        not_loaded = [k for k in missing.missing_keys if "num_batches_tracked" not in k]
        assert not not_loaded, not_loaded
        feats = decoder_features()
        enc.set_features(feats)
        with torch.no_grad():
            r_out = net(torch.zeros(1, 3, 256, 256))
            r_inv256, r_path1 = net.depth_net.forward(torch.zeros(1, 3, 256, 256))
            r_seg256 = net.seg_head(r_path1)
            o_inv256, o_path1 = R.dpt_decoder(sd, feats)
            o_seg256 = R.seg_head(sd, o_path1, sigmoid)
            o_out = R.project(o_inv256, o_seg256)
        ok = dict(inv256=eq(r_inv256, o_inv256), path1=eq(r_path1, o_path1), seg256=eq(r_seg256, o_seg256),
                  out=all(eq(a, b) for a, b in zip(r_out, o_out)))
        report[f"decoder_bit_exact_sigmoid={sigmoid}"] = ok
        assert all(ok.values()), ok
        np.savez_compressed(
            os.path.join(GOLD, f"decoder_B1_{'sigmoid' if sigmoid else 'tanh'}.npz"),
            seed=np.int64(77),
            inv256=r_inv256.numpy(), seg256=r_seg256.numpy(),
            path1_sample=r_path1[0, ::16, ::8, ::8].numpy(),
            path1_sha=np.array(sha(r_path1)),
            occ_bits=pack_occ(r_out[3][0]), occ_count=np.int64(r_out[3][0].sum().item()),
        )
        print(f"decoder golden (sigmoid={sigmoid}): inv256 range", float(r_inv256.min()), float(r_inv256.max()),
              "occ", int(r_out[3][0].sum().item()))

    # param order / key layout of the reference's decoder+heads
    names = [n for n, _ in net.named_parameters()]
    with open(os.path.join(GOLD, "param_order_decoder.json"), "w") as f:
        json.dump(dict(named_parameters=names, state_dict_keys=list(net.state_dict().keys())), f, indent=0)

    # ---- (2c) dpt_hybrid_384 (BASELINE configs[2]): the reference's own DPT.forward / forward_vit / forward_adapted_unflatten /
    # forward_flex / ProjectReadout / fusion blocks / heads / projection at 384 x 384 with the [256, 512, 768, 768] pyramid; the
    # timm arithmetic (ResNetV2 + ViT blocks) is the oracle's restatement inside a stand-in module (parity unpinned there).
    from SOccDPT.model.backbones import vit as ref_vit, utils as ref_utils
    dpt._make_encoder = lambda backbone, features, use_pretrained, groups=1, expand=False, exportable=True, hooks=None, \
        use_vit_only=False, use_readout="ignore", in_features=None: (
        make_reference_hybrid_backbone(ref_vit, ref_utils),
        blocks._make_scratch([256, 512, 768, 768], features, groups=groups, expand=expand))
    hnet = S.SOccDPT_V3(sigmoid=False, load_depth=False, path=None, camera_intrinsics_yaml=calib, compute_occ=True, model_type="dpt_hybrid_384").eval()
    assert hnet.depth_net.forward_transformer is ref_vit.forward_vit
    hsd = synth_state_dict("vitb_rn50_384", alias_pretrained=True)
    res = hnet.load_state_dict(hsd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert not [k for k in res.missing_keys if "num_batches_tracked" not in k], res.missing_keys
    xh = synth_input(1, size=384, seed0=21)
    with torch.no_grad():
        r_layers = ref_vit.forward_vit(hnet.depth_net.pretrained, xh)
        o_layers = R.hybrid_encoder(hsd, xh)
        r_out = hnet(xh)
        r_inv, r_p1 = hnet.depth_net.forward(xh)
        r_seg = hnet.seg_head(r_p1)
        o_inv, o_seg, o_p1 = R.soccdpt_v3_network(hsd, xh, backbone="vitb_rn50_384", sigmoid=False)
        o_out = R.project(o_inv, o_seg)
    ok = dict(layers=all(eq(a, b) for a, b in zip(r_layers, o_layers)), inv=eq(r_inv, o_inv), path1=eq(r_p1, o_p1), seg=eq(r_seg, o_seg),
              out=all(eq(a, b) for a, b in zip(r_out, o_out)))
    report["hybrid_384_reference_adapters_decoder_bit_exact"] = ok
    assert all(ok.values()), ok
    np.savez_compressed(
        os.path.join(GOLD, "hybrid_B1_tanh.npz"), seed=np.int64(21),
        inv384=r_inv.numpy(), seg384_sample=r_seg[0, :, ::4, ::4].numpy(), path1_sample=r_p1[0, ::16, ::12, ::12].numpy(),
        **{f"layer{i + 1}_sample": l[0, ::16, ::3, ::3].numpy() for i, l in enumerate(r_layers)},
        occ_bits=pack_occ(r_out[3][0]), occ_count=np.int64(r_out[3][0].sum().item()))
    hnames = [n for n, _ in hnet.named_parameters()]
    with open(os.path.join(GOLD, "param_order_hybrid.json"), "w") as f:
        json.dump(dict(named_parameters=hnames, state_dict_keys=list(hnet.state_dict().keys()),
                       note="decoder / act_postprocess / seg_head names come from the reference's modules; depth_net.pretrained.model.* from the "
                            "timm-named stand-in (3p-recall, unpinned)"), f, indent=0)
    print("hybrid golden: inv range", float(r_inv.min()), float(r_inv.max()), "occ", int(r_out[3][0].sum().item()))

    # ---- (2b) evaluation metrics: the reference's own functions vs oracle/metrics_ref.py ----
    from SOccDPT.loss.ssi_loss import compute_scale_and_shift as ref_css
    from SOccDPT.utils import compute_masked_errors as ref_cme
    from oracle import metrics_ref as MR
    pred, gt, mask, seg_pred, seg_gt = metrics_inputs()
    r_scale, r_shift = ref_css(pred, gt, mask)
    r_ssi = r_scale.view(-1, 1, 1) * pred + r_shift.view(-1, 1, 1)
    r_m = ref_cme(gt.numpy(), r_ssi.numpy(), mask.numpy())
    o_m = MR.depth_metrics_batch(pred, gt, mask)
    assert np.array_equal(np.array(r_m, dtype=np.float64), np.array(o_m[:7], dtype=np.float64)), (r_m, o_m[:7])
    assert torch.equal(r_scale, torch.from_numpy(o_m[7])) and torch.equal(r_shift, torch.from_numpy(o_m[8]))
    r_iou = 0.0   # evaluate_seg loop body (utils/__init__.py:314-330)
    for c in range(3):
        pm, ym = seg_pred[:, c] > 0.5, seg_gt[:, c] > 0.5
        r_iou = r_iou + torch.logical_and(pm, ym).sum(dim=(1, 2)) / (torch.logical_or(pm, ym).sum(dim=(1, 2)) + 1e-7)
    r_iou = (r_iou / 3).numpy()
    assert np.array_equal(r_iou, MR.iou_batch(seg_pred, seg_gt))
    report["metrics_bit_exact"] = True
    np.savez_compressed(os.path.join(GOLD, "metrics.npz"), seed=np.int64(321), depth=np.array(r_m, dtype=np.float64),
                        scale=r_scale.numpy(), shift=r_shift.numpy(), iou=r_iou)
    print("metrics golden:", [float(v) for v in r_m], r_iou)

    # ---- (3) encoders: oracle vs the independent HF ports (parity stays unpinned at the timm boundary) ----
    if not args.skip_hf:
        for name in [m for m in sys.modules if m.split(".")[0] in ("timm", "cv2", "torchvision", "wandb", "matplotlib")]:
            del sys.modules[name]  # drop the stand-ins before importing transformers
        from oracle.hf_crosscheck import hybrid_hf_features, swinv2_hf_features
        for backbone, img in (("swin2t16_256", 256), ("swin2b24_384", 384)):
            sd = synth_state_dict(backbone)
            x = synth_input(1, size=img)
            with torch.no_grad():
                my_feats = R.swin_encoder(sd, x, R.ARCHS[backbone])
            errs = [float((a - b).abs().max() / b.abs().max()) for a, b in zip(swinv2_hf_features(sd, x, backbone), my_feats)]
            report[f"encoder_vs_hf_max_rel[{backbone}]"] = errs
            print(f"{backbone} encoder vs HF Swinv2Model, max|diff|/max|ref| per stage:", errs)
            assert max(errs) < 2e-5, errs
            if backbone == "swin2t16_256":
                np.savez_compressed(
                    os.path.join(GOLD, "encoder_B1_unpinned.npz"),
                    seed=np.int64(0),
                    **{f"stage{i}_sample": f[0, ::8, ::4, ::4].numpy() for i, f in enumerate(my_feats)},
                    **{f"stage{i}_absmean": np.float64(f.abs().mean().item()) for i, f in enumerate(my_feats)},
                )
        sd = synth_state_dict("vitb_rn50_384")
        x = synth_input(1, size=384)
        with torch.no_grad():
            my_feats = R.hybrid_encoder(sd, x)
        errs = [float((a - b).abs().max() / b.abs().max()) for a, b in zip(hybrid_hf_features(sd, x), my_feats)]
        report["encoder_vs_hf_max_rel[vitb_rn50_384]"] = errs
        print("vitb_rn50_384 encoder + reassemble vs HF DPT-hybrid, max|diff|/max|ref| per level:", errs)
        assert max(errs) < 2e-5, errs

    # ---- (4) end-to-end oracle output summary for the synthetic model (regression pin) ----
    sd = synth_state_dict()
    x = synth_input(1)
    with torch.no_grad():
        inv256, seg256, p1 = R.soccdpt_v3_network(sd, x, sigmoid=False)
        out = R.project(inv256, seg256)
    np.savez_compressed(
        os.path.join(GOLD, "e2e_B1_tanh_oracle.npz"),
        inv256=inv256.numpy(), seg256_sample=seg256[0, :, ::4, ::4].numpy(),
        occ_bits=pack_occ(out[3][0]), occ_count=np.int64(out[3][0].sum().item()),
    )
    with open(os.path.join(GOLD, "report.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
