"""Diff two tools/site_times.py outputs per launch site / kernel family: python tools/site_diff.py a.json b.json"""
import json, sys
def _load(f):
    t = open(f).read()
    return json.loads(t[t.index('{'):])   # the model's constructor prints two lines first
a, b = (_load(f) for f in sys.argv[1:3])
ra, rb = ({r["name"]: r for r in d["rows"]} for d in (a, b))
rows = []
for k in sorted(set(ra) | set(rb)):
    ua, ub = ra.get(k, {}).get("us_per_forward", 0.0), rb.get(k, {}).get("us_per_forward", 0.0)
    rows.append((ub - ua, k, ua, ub, ra.get(k, rb.get(k)).get("cfg", "")))
rows.sort(reverse=True)
print(f"{'site':28s} {a['precision']:>10s} {b['precision']:>10s} {'delta':>8s}")
for d, k, ua, ub, cfg in rows:
    if abs(d) >= 0.5:
        print(f"{k:28s} {ua:10.1f} {ub:10.1f} {d:8.1f}  {cfg}")
print("total", sum(r[2] for r in rows), sum(r[3] for r in rows))
