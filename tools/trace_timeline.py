"""Per-dispatch timeline of ONE forward from a `rocprofv3 --kernel-trace --output-format csv` run of bench.py:
prints every launch of the chosen step in order with its grid, device duration and the idle gap before it."""
import csv, sys, re
path = sys.argv[1]
step_from_end = int(sys.argv[2]) if len(sys.argv) > 2 else 12   # which patch_embed occurrence (from the end) starts the step
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "patch_embed_kernel" in r["Kernel_Name"]]
i0 = starts[-step_from_end]; i1 = starts[-step_from_end + 1]
prev_end = None; tot = 0; gap_tot = 0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"soccdpt::|void |unsigned short|\(.*", "", r["Kernel_Name"]).strip()
    gap = 0 if prev_end is None else s - prev_end
    tot += e - s; gap_tot += max(gap, 0)
    print(f"{name[:58]:58s} grid {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):6d} x{r['Workgroup_Size_X']:>4s}  {(e-s)/1e3:8.2f} us  gap {gap/1e3:7.2f}")
    prev_end = e
print(f"launches {i1-i0}  kernel time {tot/1e3:.1f} us  gaps {gap_tot/1e3:.1f} us  span {(int(rows[i1-1]['End_Timestamp'])-int(rows[i0]['Start_Timestamp']))/1e3:.1f} us")
