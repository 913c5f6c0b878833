#!/bin/bash
set -o pipefail
O=gpurun_out/r06_c7
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for n in 2 3; do timeout -k 10 600 python3 bench.py --no-side-modes --no-cpu-baseline --in-flight $n > $O/bench_if$n.json 2> $O/bench_if$n.err; echo "rc $?"; python3 -c "
import json; d=json.loads(open('$O/bench_if$n.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], json.dumps(d.get('pipelined')))"; done
for c in 2 3; do timeout -k 10 600 python3 bench.py --no-side-modes --no-cpu-baseline --config $c > $O/bench_c$c.json 2> $O/bench_c$c.err; python3 -c "
import json; d=json.loads(open('$O/bench_c$c.json').read().strip().splitlines()[-1]); print('config $c', d['value'], d['ms_per_step'], json.dumps(d.get('pipelined')))"; done
