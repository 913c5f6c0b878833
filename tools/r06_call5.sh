#!/bin/bash
set -o pipefail
O=gpurun_out/r06_c5
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "depth_tail or window_attention" > $O/tests_kernel.log 2>&1; rc=$?; echo "kernel tests rc $rc"; tail -3 $O/tests_kernel.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 900 python3 -m pytest tests/test_mixed_gpu.py tests/test_network_gpu.py -x -q > $O/tests_net.log 2>&1; echo "network tests rc $?"; tail -4 $O/tests_net.log
for i in 1 2; do for c in 1 2 3; do python3 bench.py --headline-only --steps 200 --config $c 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); k={r['name']:(r['ms_per_step'], r['launches_per_step']) for r in d['kernels']}; print('config $c', d['value'], d['ms_per_step'], 'depth_tail', k.get('depth_tail_fused'), 'wattn_qkv', k.get('window_attention_qkv'))"; done; done 2>&1 | tee $O/bench3.txt
