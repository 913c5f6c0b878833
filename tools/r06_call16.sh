#!/bin/bash
set -o pipefail
O=gpurun_out/r06_c16
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; echo "gpu tests rc $?"; tail -3 $O/tests_gpu.log
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for a in "--amp bf16" "--amp x3"; do python3 bench.py --train-step $a --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$a', d['value'], d['ms_per_step'], d['launches_per_step'], d['split_ms'])"; done
python3 bench.py --headline-only --steps 200 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('forward', d['value'], d['ms_per_step'])"
