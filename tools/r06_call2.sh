#!/bin/bash
# round 6, GPU call 2: fused qkv + attention kernel: kernel tests, network tests, A/B against the two-launch chain
set -o pipefail
O=gpurun_out/r06_c2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -s -k "window_attention" > $O/tests_kernel.log 2>&1; rc=$?; echo "kernel tests rc $rc"; grep -E "fused qkv|passed|failed|Error|error" $O/tests_kernel.log | tail -50
[ $rc -eq 0 ] || exit 1
timeout -k 10 900 python3 -m pytest tests/test_mixed_gpu.py tests/test_network_gpu.py -x -q > $O/tests_net.log 2>&1; echo "network tests rc $?"; tail -8 $O/tests_net.log
for i in 1 2; do
  for f in 0 1; do
    for c in 1 3; do
      SOCCDPT_FUSE_QKV=$f python3 bench.py --headline-only --steps 200 --config $c 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); k={r['name']:(r['ms_per_step'], r['launches_per_step']) for r in d['kernels']}; print('fuse $f config $c', d['value'], d['ms_per_step'], 'wattn', k.get('window_attention'), 'wattn_qkv', k.get('window_attention_qkv'), 'launches', d['launches_per_step'])"
    done
  done
done 2>&1 | tee $O/ab_fuse.txt
for m in 1 2 4 8; do SOCCDPT_FUSE_QKV_STAGES=$m python3 bench.py --headline-only --steps 200 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); k={r['name']:(r['ms_per_step'], r['launches_per_step']) for r in d['kernels']}; print('stage mask $m', d['value'], d['ms_per_step'], 'wattn', k.get('window_attention'), 'wattn_qkv', k.get('window_attention_qkv'))"; done 2>&1 | tee -a $O/ab_fuse.txt
