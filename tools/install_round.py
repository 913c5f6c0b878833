"""Copy what tools/collect_round.sh <tag> left under gpurun_out/ into profiles/ (tracked).  usage: python tools/install_round.py r03"""
import os, shutil, subprocess, sys
tag = sys.argv[1]
for t in (tag, f"{tag}_base384", f"{tag}_hybrid384"):
    subprocess.check_call([sys.executable, "tools/install_profiles.py", t])
src, dst = f"gpurun_out/{tag}_extra", "profiles"
for f in sorted(os.listdir(src)):
    p = os.path.join(src, f)
    if f.startswith("bench_") and f.endswith(".json") and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(dst, f"{tag}_{f}"))
for a, name in (("0", "f32"), ("x3", "x3"), ("bf16", "bf16"), ("hybrid_bf16", "hybrid384_bf16")):
    p = os.path.join(src, f"train_{a}", "t_kernel_stats.csv")
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, f"{tag}_train_{name}_kernel_stats.csv"))
        shutil.copy(os.path.join(src, f"train_{a}.json"), os.path.join(dst, f"{tag}_train_{name}_bench_under_rocprof.json"))
# the bench lines re-run AFTER the PMC summaries were installed (only then can bench.py fill roofline.traffic from a matching file)
for t in (tag, f"{tag}_base384", f"{tag}_hybrid384"):
    p = f"gpurun_out/{t}_bench_final.json"
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(dst, f"{t}_bench.json"))
print("installed", tag)
