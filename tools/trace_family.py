"""Per-launch device durations of ONE kernel family inside one forward, from a `rocprofv3 --kernel-trace --output-format csv` run of bench.py:
    python tools/trace_family.py <kernel_trace.csv> <substring of the kernel name> [anchor substring = first kernel of a forward] [forward from the end]
prints every launch of the family in launch order with its grid, duration and the idle gap in front of it."""
import csv, re, sys
path, fam = sys.argv[1], sys.argv[2]
anchor = sys.argv[3] if len(sys.argv) > 3 else "stem_im2col"
back = int(sys.argv[4]) if len(sys.argv) > 4 else 5
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
i0, i1 = starts[-back], starts[-back + 1]
prev_end, tot, n = None, 0.0, 0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if fam in r["Kernel_Name"]:
        name = re.sub(r"soccdpt::|\(anonymous namespace\)::|void |\(.*", "", r["Kernel_Name"]).strip()
        print(f"{name[:60]:60s} grid {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):6d} x{r['Workgroup_Size_X']:>4s}  {(e - s) / 1e3:8.2f} us  gap {0 if prev_end is None else (s - prev_end) / 1e3:6.2f}")
        tot += (e - s) / 1e3; n += 1
    prev_end = e
print(f"{n} launches of '{fam}' in one forward: {tot:.1f} us; forward span {(int(rows[i1 - 1]['End_Timestamp']) - int(rows[i0]['Start_Timestamp'])) / 1e3:.1f} us, {i1 - i0} launches")
