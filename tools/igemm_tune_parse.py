import csv, json, sys, collections, re
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "igemm_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
order = json.load(open(sys.argv[2]))
assert len(rows) == len(order), (len(rows), len(order))
acc = collections.OrderedDict()
for r, (name, t, M, N, K) in zip(rows, order):
    cfg = re.search(r"Cfg<([\d, ]+)>", r["Kernel_Name"]).group(1).replace(" ", "")
    acc.setdefault((name, M, N, K), collections.OrderedDict()).setdefault((t, cfg), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (name, M, N, K), d in acc.items():
    best = min(d.items(), key=lambda kv: min(kv[1]))
    auto = ([(k, v) for k, v in d.items() if k[0] == -1] or [best])[0]
    line = f"{name:9s} M={M:6d} N={N:5d} K={K:5d} | auto {auto[0][1]:>16s} {min(auto[1]):7.2f} | best {best[0][1]:>16s} {min(best[1]):7.2f} |"
    for (t, cfg), v in d.items():
        if t >= 0:
            line += f" {t}:{min(v):.1f}"
    print(line)
