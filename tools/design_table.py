"""Regenerates the kernel table of DESIGN.md section 4 from the committed bench line and PMC summary (profiles/<tag>_bench.json,
profiles/<tag>_pmc_traffic.json): python tools/design_table.py r04"""
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
d = json.load(open(os.path.join(REPO, "profiles", f"{tag}_bench.json")))
pmc, stall = {}, {}
try:
    pmc = json.load(open(os.path.join(REPO, "profiles", f"{tag}_pmc_traffic.json")))["kernels"]
except Exception:
    pass
try:
    stall = json.load(open(os.path.join(REPO, "profiles", f"{tag}_pmc_stall.json")))["kernels"]
except Exception:
    pass
WHAT = {
    "igemm_f16": "Linear / 1x1 / 3x3 convolution, fp16 operands (`igemm_kernel<BM,BN,BK,...>`)",
    "igemm_x3": "the same template on x3 split-fp16 operands (three MFMAs per product; groups the precision map promotes)",
    "igemm_x2w": "the same template with fp16 activations and x3 weight pairs (two MFMAs per product: round 5, section 11.3)",
    "igemm_bf16": "the same template, bf16 operands",
    "window_attention_qkv": "qkv projection + Swin-V2 cosine window attention in one launch (`attention_qkv.hip`, round 6: stages 0-1 where the qkv group reads 16-bit activations)",
    "window_attention": "Swin-V2 cosine window attention + roll / partition / reverse (`attention.hip`)",
    "ln_residual": "post-norm `x + LN(y)` (C >= 192), operand copy + hooked halo image",
    "project_voxelise": "`get_semantic_occupancy` + `rotate_points` + voxel index pass (`projection.hip`, 4 camera rows per workgroup)",
    "depth_tail_fused": "Interpolate x2 + Conv3x3(128->32) + ReLU + Conv1x1 + ReLU (`depth_tail.hip`)",
    "occ_expand": "packed bits -> B dense f32 rows",
    "seg_tail": "seg head: n-tile partial logits + bias, bilinear x2, ScaledTanh / Sigmoid (the Conv1x1(256->3) itself rides in the `_dot3` launch)",
    "bilinear_resize": "`F.interpolate(bilinear, align_corners=True)` of path_1 into the zero-halo operand image",
    "mlp_ln_fused": "fc1 + GELU + fc2 + LayerNorm + residual in one launch (C <= 128, `mlp_fused.hip`)",
    "patch_embed_ln": "PatchEmbed conv 4x4 / 4 + LayerNorm",
    "merge_gather": "PatchMerging 2x2 gather (only where the producing LayerNorm could not write the merged layout)",
}
rows = ["| kernel family | replaces | bound | launches | us / forward | achieved | of peak | PMC (HBM MB / launch, MFMA busy) | SQ stall shares (parked / issue-stalled / issuing) |", "|---|---|---|---|---|---|---|---|---|"]
for k in d["kernels"]:
    name = k["name"]
    what = next((v for p, v in WHAT.items() if name.startswith(p)), "")
    if name.endswith("_dot3"):
        what = "seg head Conv3x3(256->256) + BN + ReLU with the Conv1x1(256->3) classifier in its epilogue (section 10.3); since round 5 on the 16-wave 256 x 256 tile (section 11.4)"
    if "tflops" in k:
        peak = 2500.0 / 3 if name.startswith("igemm_x3") else (1250.0 if name.startswith("igemm_x2w") else 2500.0)
        ach, frac, bound = f"{k['tflops']:.0f} TFLOP/s", f"{k['tflops'] / peak:.3f}", "MFMA" if k["tflops"] > 300 else "MFMA nominally; launch / L2->LDS fill latency in practice"
    elif "gbs" in k:
        ach, frac, bound = f"{k['gbs'] / 1e3:.2f} TB/s", f"{k['gbs'] / 8000.0:.3f}", "HBM"
    else:
        ach, frac, bound = "", "", "latency"
    pm = pmc.get(name)
    pmtxt = f"{pm['hbm_bytes_per_launch'] / 1e6:.1f}" + (f", {pm['mfma_util']:.3f}" if pm and "mfma_util" in pm else "") if pm else ""
    st = stall.get(name)
    sttxt = f"{st['wait_any']:.2f} / {st['wait_inst_any']:.2f} / {st['active_inst_any']:.2f}" if st else ""
    rows.append(f"| `{name}` | {what} | {bound} | {k['launches_per_step']:g} | {k['ms_per_step'] * 1e3:.1f} | {ach} | {frac} | {pmtxt} | {sttxt} |")
r = d["roofline"]
head = (f"Forward: {d['ms_per_step']} ms per step = **{d['value']:.0f} frames/s** ({d['dtype'].split(' (')[0]}), {d['launches_per_step']} launches, kernels sum to "
        f"{d['device_ms_per_step']} ms; igemm template as a whole {r['achieved']} TFLOP/s = {r['frac']} of the 2500 TFLOP/s 16-bit peak ({r.get('frac_vs_blended_peak', '?')} of its blended {r.get('blended_peak', '?')} TFLOP/s); whole forward {r.get('whole_forward_frac', '?')}; "
        f"B = 1 latency {d.get('latency_b1', {}).get('ms_per_frame', '?')} ms.\n\n")
table = "<!-- r-table-begin -->\n" + head + "\n".join(rows) + "\n<!-- r-table-end -->"
p = os.path.join(REPO, "DESIGN.md")
s = open(p).read()
if "@@R04_KERNEL_TABLE@@" in s:
    s = s.replace("@@R04_KERNEL_TABLE@@", table)
else:
    s = re.sub(r"<!-- r-table-begin -->.*?<!-- r-table-end -->", lambda m: table, s, flags=re.S)
open(p, "w").write(s)
print(table)
