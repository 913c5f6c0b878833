#!/bin/bash
set -o pipefail
O=gpurun_out/r06_c8
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; echo "gpu tests rc $?"; tail -6 $O/tests_gpu.log
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 600 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; python3 -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('frac_vs_blended_peak'), d['tolerance']['worst_measured'], json.dumps(d['pipelined'])[:200]); print({k:(v['calibrated']['value'], v['uncalibrated']['value'], v['calibrated']['heldout_worst']) for k,v in d['other_weights']['sets'].items()})"
