"""Scan the built objects for global loads that sit under their own branch and are waited for one by one (`c < C ? p[c] : 0` compiles to
s_cbranch_execz / global_load / s_waitcnt vmcnt(0): N such loads are N dependent memory round trips).  Prints, per kernel, how many s_waitcnt vmcnt(0)
follow a load with a branch in between since the previous wait -- a rough count of serialised round trips.
    python tools/isa_scan.py [objects...]   (default: soccdpt_amd/csrc/build/*.o)"""
import glob, os, re, subprocess, sys, tempfile
objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
objs = sys.argv[1:] or sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "soccdpt_amd/csrc/build/*.o")))
for o in objs:
    with tempfile.TemporaryDirectory() as d:
        subprocess.run(["cp", o, d + "/x.o"], check=True)
        subprocess.run([objdump, "--offloading", "x.o"], cwd=d, capture_output=True)
        co = [f for f in os.listdir(d) if "gfx950" in f]
        if not co:
            continue
        asm = subprocess.run([objdump, "-d", "--no-show-raw-insn", os.path.join(d, co[0])], capture_output=True, text=True).stdout
    name, loads, branch, chain, waits0 = None, 0, False, 0, 0
    out = []
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            if name and chain >= 3:
                out.append((chain, waits0, name))
            name, loads, branch, chain, waits0 = m.group(1), 0, False, 0, 0
            continue
        t = line.strip()
        if t.startswith("global_load") or t.startswith("buffer_load"):
            loads += 1
        elif t.startswith("s_cbranch"):
            branch = True
        elif t.startswith("s_waitcnt") and "vmcnt(0)" in t:
            waits0 += 1
            if loads and branch:
                chain += 1
            loads, branch = 0, False
    if name and chain >= 3:
        out.append((chain, waits0, name))
    for chain, waits0, name in sorted(out, reverse=True):
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        print(f"{os.path.basename(o):22s} serial-load waits {chain:3d} (vmcnt(0) waits {waits0:3d})  {dem[:150]}")
