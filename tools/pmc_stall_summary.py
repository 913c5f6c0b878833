"""Summarise ONE rocprofv3 --pmc pass of SQ stall-reason counters (SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS: the eight SQ slots of one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots") into per-kernel-family
shares of the wave cycles:  wait_any = waves parked on s_waitcnt / barriers, wait_inst = issue stalls (MFMA RAW / pipe busy), of which wait_lds = LDS
issue stalls, active = cycles an instruction issued; the three are disjoint and add up to ~ the wave cycles.
usage: python tools/pmc_stall_summary.py <counter_collection.csv> <out.json>"""
import collections, csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import family  # noqa: E402

NAMES = ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_INSTS_VALU", "SQ_INSTS_LDS")

if __name__ == "__main__":
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(sys.argv[1])):
        fam = family(r["Kernel_Name"])
        agg[fam][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[fam].add(r.get("Dispatch_Id", r.get("Correlation_Id")))
    out = {}
    for fam, c in agg.items():
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if wc <= 0:
            continue
        n = max(len(disp[fam]), 1)
        out[fam] = dict(launches=n, wave_cycles_per_launch=wc / n, busy_cycles_per_launch=c.get("SQ_BUSY_CYCLES", 0.0) / n,
                        wait_any=round(c.get("SQ_WAIT_ANY", 0.0) / wc, 4), wait_inst_any=round(c.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4),
                        active_inst_any=round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 4), wait_inst_lds=round(c.get("SQ_WAIT_INST_LDS", 0.0) / wc, 4),
                        insts_valu_per_launch=c.get("SQ_INSTS_VALU", 0.0) / n, insts_lds_per_launch=c.get("SQ_INSTS_LDS", 0.0) / n,
                        # occupancy proxy: wave cycles per busy cycle = average waves resident per SQ while it was busy
                        waves_per_busy_cycle=round(wc / c["SQ_BUSY_CYCLES"], 3) if c.get("SQ_BUSY_CYCLES") else None)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from soccdpt_amd.lib import csrc_sha  # noqa: E402
    json.dump(dict(csrc_sha=csrc_sha(), note="rocprofv3 --pmc " + " ".join(NAMES) + " over `bench.py --headline-only --steps 3 --warmup 1 --prewarm 0`; shares are of "
                   "SQ_WAVE_CYCLES (quad-cycle units on both sides); wait_any + wait_inst_any + active_inst_any ~ 1", kernels=out), open(sys.argv[2], "w"), indent=1)
    for fam, v in sorted(out.items(), key=lambda kv: -kv[1]["wave_cycles_per_launch"] * kv[1]["launches"])[:40]:
        print(f"{fam:42s} n={v['launches']:4d} wait_any {v['wait_any']:.2f} wait_inst {v['wait_inst_any']:.2f} (lds {v['wait_inst_lds']:.2f}) active {v['active_inst_any']:.2f} "
              f"valu/launch {v['insts_valu_per_launch']:.3g} lds/launch {v['insts_lds_per_launch']:.3g} waves/busy {v['waves_per_busy_cycle']}")
