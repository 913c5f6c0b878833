#!/bin/bash
set -o pipefail
O=gpurun_out/r06_c6
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SOCCDPT_PROJECT_ROWS8=1 timeout -k 10 600 python3 -m pytest tests/test_projection_gpu.py -x -q > $O/tests_proj8.log 2>&1; echo "projection tests (8 rows) rc $?"; tail -3 $O/tests_proj8.log
for i in 1 2 3; do for r in 0 1; do SOCCDPT_PROJECT_ROWS8=$r python3 bench.py --headline-only --steps 200 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); k={r['name']:(r['ms_per_step'], r['launches_per_step']) for r in d['kernels']}; print('rows8=$r', d['value'], d['ms_per_step'], 'project', k.get('project_voxelise'))"; done; done 2>&1 | tee $O/ab_rows8.txt
