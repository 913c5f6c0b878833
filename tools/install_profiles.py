"""Copy the summaries collected by tools/collect_profiles.sh from gpurun_out/<tag>/ into profiles/ (tracked), aggregating the
per-dispatch PMC rows per (kernel, counter), and write profiles/<tag>_pmc_traffic.json.
usage: python tools/install_profiles.py r01f"""
import csv, os, shutil, subprocess, sys
tag = sys.argv[1]
src, dst = f"gpurun_out/{tag}", "profiles"
shutil.copy(f"{src}/bench.json", f"{dst}/{tag}_bench.json")
shutil.copy(f"{src}/bench_under_rocprof.json", f"{dst}/{tag}_bench_under_rocprof.json")
shutil.copy(f"{src}/stats/s_kernel_stats.csv", f"{dst}/{tag}_kernel_stats.csv")
import collections
for name, f in (("fetch_size", "pmc_fetch/f"), ("write_size", "pmc_write/w"), ("mfma", "pmc_mfma/m")):
    agg = collections.OrderedDict()   # (kernel, counter) -> [dispatches, sum]: the per-dispatch CSVs are megabytes
    for r in csv.DictReader(open(f"{src}/{f}_counter_collection.csv")):
        a = agg.setdefault((r["Kernel_Name"], r["Counter_Name"]), [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    with open(f"{dst}/{tag}_pmc_{name}_counter_collection.csv", "w", newline="") as out:
        w = csv.writer(out)
        w.writerow(["Kernel_Name", "Counter_Name", "Dispatches", "Counter_Value_Sum"])
        for (k, c), (n, v) in agg.items():
            w.writerow([k, c, n, v])
if os.path.exists(f"{src}/pmc_stall.json"):   # round 6: SQ stall-reason shares per kernel family (tools/pmc_stall_summary.py on the box)
    shutil.copy(f"{src}/pmc_stall.json", f"{dst}/{tag}_pmc_stall.json")
subprocess.check_call([sys.executable, "tools/pmc_summary.py", f"{dst}/{tag}_pmc_fetch_size_counter_collection.csv",
                       f"{dst}/{tag}_pmc_write_size_counter_collection.csv", f"{dst}/{tag}_pmc_traffic.json",
                       f"{dst}/{tag}_pmc_mfma_counter_collection.csv"])
