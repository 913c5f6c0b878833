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

    # ---- (3) encoder: oracle vs HF Swinv2 (independent port; parity unpinned) ----
    if not args.skip_hf:
        for name in [m for m in sys.modules if m.split(".")[0] in ("timm", "cv2", "torchvision")]:
            del sys.modules[name]  # drop the stand-ins before importing transformers
        from transformers import Swinv2Config, Swinv2Model
        arch = R.ARCHS["swin2t16_256"]
        cfg = Swinv2Config(image_size=256, patch_size=4, embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24],
                           window_size=16, pretrained_window_sizes=[0, 0, 0, 0], drop_path_rate=0.0)
        hf = Swinv2Model(cfg, add_pooling_layer=False).eval()
        sd = synth_state_dict()
        pfx = "depth_net.pretrained.model."
        hsd = {}
        hsd["embeddings.patch_embeddings.projection.weight"] = sd[pfx + "patch_embed.proj.weight"]
        hsd["embeddings.patch_embeddings.projection.bias"] = sd[pfx + "patch_embed.proj.bias"]
        hsd["embeddings.norm.weight"] = sd[pfx + "patch_embed.norm.weight"]
        hsd["embeddings.norm.bias"] = sd[pfx + "patch_embed.norm.bias"]
        for s, depth in enumerate(arch.depths):
            C = arch.embed << s
            for j in range(depth):
                t = f"{pfx}layers.{s}.blocks.{j}."
                h = f"encoder.layers.{s}.blocks.{j}."
                hsd[h + "attention.self.logit_scale"] = sd[t + "attn.logit_scale"]
                for m in ("0.weight", "0.bias", "2.weight"):
                    hsd[h + "attention.self.continuous_position_bias_mlp." + m] = sd[t + "attn.cpb_mlp." + m]
                w = sd[t + "attn.qkv.weight"]
                hsd[h + "attention.self.query.weight"] = w[:C]
                hsd[h + "attention.self.key.weight"] = w[C:2 * C]
                hsd[h + "attention.self.value.weight"] = w[2 * C:]
                hsd[h + "attention.self.query.bias"] = sd[t + "attn.q_bias"]
                hsd[h + "attention.self.value.bias"] = sd[t + "attn.v_bias"]
                hsd[h + "attention.output.dense.weight"] = sd[t + "attn.proj.weight"]
                hsd[h + "attention.output.dense.bias"] = sd[t + "attn.proj.bias"]
                hsd[h + "layernorm_before.weight"] = sd[t + "norm1.weight"]
                hsd[h + "layernorm_before.bias"] = sd[t + "norm1.bias"]
                hsd[h + "intermediate.dense.weight"] = sd[t + "mlp.fc1.weight"]
                hsd[h + "intermediate.dense.bias"] = sd[t + "mlp.fc1.bias"]
                hsd[h + "output.dense.weight"] = sd[t + "mlp.fc2.weight"]
                hsd[h + "output.dense.bias"] = sd[t + "mlp.fc2.bias"]
                hsd[h + "layernorm_after.weight"] = sd[t + "norm2.weight"]
                hsd[h + "layernorm_after.bias"] = sd[t + "norm2.bias"]
            if s < 3:
                hsd[f"encoder.layers.{s}.downsample.reduction.weight"] = sd[f"{pfx}layers.{s}.downsample.reduction.weight"]
                hsd[f"encoder.layers.{s}.downsample.norm.weight"] = sd[f"{pfx}layers.{s}.downsample.norm.weight"]
                hsd[f"encoder.layers.{s}.downsample.norm.bias"] = sd[f"{pfx}layers.{s}.downsample.norm.bias"]
        hsd["layernorm.weight"] = sd[pfx + "norm.weight"]
        hsd["layernorm.bias"] = sd[pfx + "norm.bias"]
        res = hf.load_state_dict(hsd, strict=False)
        assert not res.unexpected_keys, res.unexpected_keys
        assert not [k for k in res.missing_keys if "relative" not in k], res.missing_keys
        x = synth_input(1)
        with torch.no_grad():
            emb, dims = hf.embeddings(x)
            eo = hf.encoder(emb, dims, output_hidden_states=True, output_hidden_states_before_downsampling=True)
            hf_feats = eo.reshaped_hidden_states[1:]
            my_feats = R.swin_encoder(sd, x, arch)
        errs = []
        for a, b in zip(hf_feats, my_feats):
            errs.append(float((a - b).abs().max() / b.abs().max()))
        report["encoder_vs_hf_max_rel"] = errs
        print("encoder vs HF Swinv2Model, max|diff|/max|ref| per stage:", errs)
        assert max(errs) < 2e-5, errs
        np.savez_compressed(
            os.path.join(GOLD, "encoder_B1_unpinned.npz"),
            seed=np.int64(0),
            **{f"stage{i}_sample": f[0, ::8, ::4, ::4].numpy() for i, f in enumerate(my_feats)},
            **{f"stage{i}_absmean": np.float64(f.abs().mean().item()) for i, f in enumerate(my_feats)},
        )

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
