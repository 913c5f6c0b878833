"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md §HBM
prescribes) into per-kernel HBM bytes per launch.  Units: the counters are in KiB; on gfx950 FETCH_SIZE reports
half of the bytes of wide coalesced reads, so the read side is doubled (upper bound for other access shapes).
usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import collections, csv, json, re, sys

def family(kernel_name: str) -> str:
    n = kernel_name.split("(")[0].replace("void ", "").replace("soccdpt::", "")
    m = re.match(r"igemm_kernel<Cfg<(\d+), (\d+), (\d+), \d+, \d+, (\d+)> >", n)
    if m:
        return f"igemm_bf16_{m.group(1)}x{m.group(2)}x{m.group(3)}_s{m.group(4)}"
    return {"project_kernel<3, 4>": "project_voxelise", "occ_expand_kernel": "occ_expand",
            "window_attention_kernel<16>": "window_attention", "window_attention_kernel<8>": "window_attention_8"}.get(n, n)

def load(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            a = agg[family(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return agg

f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in f:
    fl, wl = f[k], w.get(k, [1, 0.0])
    out[k] = dict(launches=fl[0], fetch_bytes_per_launch_raw=fl[1] / fl[0] * 1024, fetch_bytes_per_launch=2 * fl[1] / fl[0] * 1024,
                  write_bytes_per_launch=wl[1] / max(wl[0], 1) * 1024)
    out[k]["hbm_bytes_per_launch"] = out[k]["fetch_bytes_per_launch"] + out[k]["write_bytes_per_launch"]
json.dump(dict(note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over `bench.py --steps 3 --warmup 1`; KiB->bytes; "
                    "FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM (gfx950 reports half of wide coalesced reads)", kernels=out),
          open(sys.argv[3], "w"), indent=1)
print("wrote", sys.argv[3], len(out), "kernels")
