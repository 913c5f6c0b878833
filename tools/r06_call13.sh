#!/bin/bash
set -o pipefail
O=gpurun_out/r06_c13
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_calibrate_gpu.py -x -q -s > $O/tests_cal.log 2>&1; echo "calibrate tests rc $?"; grep -E "calibrated map|passed|failed|Error" $O/tests_cal.log | cut -c1-200 | tail -8
for c in 1 3 2; do timeout -k 10 600 python3 bench.py --config $c --no-cpu-baseline --in-flight 1 > $O/bench_c$c.json 2> $O/bench_c$c.err; python3 -c "
import json; d=json.loads(open('$O/bench_c$c.json').read().strip().splitlines()[-1]); print('config $c', d['value'], {k:(v['uncalibrated']['value'], v['calibrated']['value'], v['calibrated']['n_x3'], v['calibrated']['n_x2w'], v['calibrated']['heldout_worst'], v['vs_headline']) for k,v in d['other_weights']['sets'].items()})"; done
