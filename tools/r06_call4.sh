#!/bin/bash
set -o pipefail
O=gpurun_out/r06_c4
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "window_attention" > $O/tests_kernel.log 2>&1; rc=$?; echo "kernel tests rc $rc"; tail -3 $O/tests_kernel.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python3 tools/wattn_qkv_stamps.py 2>&1 | grep -v "^Loading\|amdgpu.ids" | tee $O/wattn_qkv_stamps.txt
for p in mixed f16; do for m in 0 1 4 8 15; do SOCCDPT_FUSE_QKV_STAGES=$m python3 bench.py --headline-only --steps 200 --precision $p 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); k={r['name']:(r['ms_per_step'], r['launches_per_step']) for r in d['kernels']}; print('$p stage mask $m', d['value'], d['ms_per_step'], 'wattn', k.get('window_attention'), 'wattn_qkv', k.get('window_attention_qkv'))"; done; done 2>&1 | tee $O/ab_fuse_stage.txt
