"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md §HBM
prescribes) into per-kernel HBM bytes per launch.  Units: the counters are in KiB; on gfx950 FETCH_SIZE reports
half of the bytes of wide coalesced reads, so the read side is doubled (upper bound for other access shapes).
usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [<mfma_counter_collection.csv>]"""
import collections, csv, json, re, sys

def family(kernel_name: str) -> str:
    """Kernel symbol -> the family name bench.py's HIP-event profiler reports (model.cpp PROF scopes)."""
    n = kernel_name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("soccdpt::", "")
    m = re.match(r"igemm_kernel<Cfg<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)(?:, (\d+))?>, ([\w: ]+), (true|false), (true|false)(?:, (?:true|false))*>", n)   # Cfg<BM, BN, BK, WM, WN, NS[, MF]>, T, LN, SK[, ST, GEN]
    if m:
        mf32 = m.group(7) == "32"
        m = re.match(r"igemm_kernel<Cfg<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)(?:, \d+)?>, ([\w: ]+), (true|false), (true|false)(?:, (?:true|false))*>", n)
        t = m.group(7)
        if "x2w_t" in t:   # one-sided split (round 5): fp16 activations, x3 weight pairs (igemm.hip kCfgNamesX2W)
            w8 = int(m.group(4)) * int(m.group(5)) == 8 and (m.group(1), m.group(2)) in (("128", "128"), ("64", "64"))
            return f"igemm_x2w_{m.group(1)}x{m.group(2)}x{m.group(3)}_s{m.group(6)}" + ("_w8" if w8 else "") + ("_ln" if m.group(8) == "true" else "")
        if t == "float" or "x3_t" in t:   # 4 bytes per element: the k-tile holds BK / 2 elements (igemm.hip kCfgNamesF32 / kCfgNamesX3)
            w8 = int(m.group(4)) * int(m.group(5)) == 8 and ((m.group(1), m.group(2)) in (("128", "128"), ("64", "64")) or ((m.group(1), m.group(2)) == ("32", "64") and "x3_t" in t))
            return (f"igemm_{'f32' if t == 'float' else 'x3'}_{m.group(1)}x{m.group(2)}x{int(m.group(3)) // 2}_s{m.group(6)}" + ("_w8" if w8 else "") +
                    ("_splitk" if m.group(9) == "true" else ""))
        if mf32:
            return f"igemm_{'f16' if 'f16' in t else 'bf16'}_{m.group(1)}x{m.group(2)}x{m.group(3)}_s{m.group(6)}_m32"
        # 8-wave forms of tiles that also exist with 4 waves carry a _w8 suffix in igemm.hip's kCfgNames
        w8 = int(m.group(4)) * int(m.group(5)) == 8 and (m.group(1), m.group(2)) in (("128", "128"), ("32", "64"), ("64", "64"))
        flags = re.findall(r"true|false", n[n.index(">,") :])   # LN, SK[, ST, GEN[, D3]]
        w16 = int(m.group(4)) * int(m.group(5)) == 16   # the 16-wave tiles of round 5
        return (f"igemm_{'f16' if 'f16' in t else 'bf16'}_{m.group(1)}x{m.group(2)}x{m.group(3)}_s{m.group(6)}" + ("_w8" if w8 else "") + ("_w16" if w16 else "") +
                ("_splitk" if m.group(9) == "true" else "") + ("_dot3" if len(flags) >= 5 and flags[4] == "true" else ""))
    for prefix, fam in (("window_attention_qkv", "window_attention_qkv"), ("window_attention", "window_attention"), ("mlp_ln_kernel", "mlp_ln_fused"), ("project_", "project_voxelise"), ("occ_expand", "occ_expand"),
                        ("ln_residual", "ln_residual"), ("depth_tail", "depth_tail_fused"), ("patch_embed", "patch_embed_ln"),
                        ("bilinear", "bilinear_resize"), ("merge_gather", "merge_gather"), ("conv1x1_c3", "seg_tail"), ("seg_up_act", "seg_tail"), ("seg_logits_finish", "seg_tail")):
        if n.startswith(prefix):
            return fam
    return n

def load(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):   # per-dispatch rocprofv3 rows, or the per-kernel aggregate tools/install_profiles.py writes
        if r["Counter_Name"] == counter:
            a = agg[family(r["Kernel_Name"])]
            a[0] += int(r["Dispatches"]) if "Dispatches" in r else 1
            a[1] += float(r["Counter_Value_Sum"]) if "Counter_Value_Sum" in r else float(r["Counter_Value"])
    return agg

if __name__ == "__main__":
    f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in f:
        fl, wl = f[k], w.get(k, [1, 0.0])
        out[k] = dict(launches=fl[0], fetch_bytes_per_launch_raw=fl[1] / fl[0] * 1024, fetch_bytes_per_launch=2 * fl[1] / fl[0] * 1024,
                      write_bytes_per_launch=wl[1] / max(wl[0], 1) * 1024)
        out[k]["hbm_bytes_per_launch"] = out[k]["fetch_bytes_per_launch"] + out[k]["write_bytes_per_launch"]
    if len(sys.argv) > 4:  # optional third pass: SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE -> MFMA pipe utilisation per kernel family
        mf, ga = load(sys.argv[4], "SQ_VALU_MFMA_BUSY_CYCLES"), load(sys.argv[4], "GRBM_GUI_ACTIVE")
        for k in out:
            if k in mf and k in ga and ga[k][1] > 0:
                # SQ_VALU_MFMA_BUSY_CYCLES is summed over the 256 CUs x 4 SIMDs (16 cycles per v_mfma_f32_16x16x32: checked against the
                # analytic MFMA count of the 256x256 conv, 9,437,184 x 16 = 150,994,944); GRBM_GUI_ACTIVE is summed over the 8 XCDs
                out[k]["mfma_busy_cycles_per_launch"] = mf[k][1] / mf[k][0]
                out[k]["gpu_active_cycles_per_launch_sum_of_8_xcds"] = ga[k][1] / ga[k][0]
                out[k]["mfma_util"] = mf[k][1] / (ga[k][1] / 8.0 * 256 * 4)
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from soccdpt_amd.lib import csrc_sha  # noqa: E402
    json.dump(dict(csrc_sha=csrc_sha(), precision=os.environ.get("SOCCDPT_PROFILE_PRECISION", "mixed"),   # the arithmetic of the profiled headline leg (bench.py's default)
                   note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over `bench.py --headline-only --steps 3 --warmup 1`; KiB->bytes; "
                        "FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM (gfx950 reports half of wide coalesced reads)", kernels=out),
              open(sys.argv[3], "w"), indent=1)
    print("wrote", sys.argv[3], len(out), "kernels")
