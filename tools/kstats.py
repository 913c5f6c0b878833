"""Print a rocprofv3 --stats kernel summary (t_kernel_stats.csv) sorted by total time.  usage: python tools/kstats.py <csv> [rows] [per-N divisor]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
div = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:top]:
    t = float(r["TotalDurationNs"])
    print(f"{t / 1e6 / div:9.3f} ms {100 * t / tot:5.1f}% calls {int(r['Calls']) / div:8.1f} avg {float(r['AverageNs']) / 1e3:9.1f} us  {r['Name'][:120]}")
print(f"total {tot / 1e6 / div:.3f} ms")
