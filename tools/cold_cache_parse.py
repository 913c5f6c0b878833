import csv, json, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "igemm_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
order = json.load(open(sys.argv[2]))
assert len(rows) == len(order), (len(rows), len(order))
acc = collections.OrderedDict()
for r, (name, mode) in zip(rows, order):
    acc.setdefault((name, mode), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (name, mode), v in acc.items():
    v = sorted(v)
    print(f"{name:12s} {mode:14s} median {v[len(v)//2]:7.2f} us   min {v[0]:7.2f}")
