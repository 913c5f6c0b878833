#!/bin/bash
set -o pipefail
O=gpurun_out/r06_c15
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_train_step_gpu.py tests/test_autograd_bridge_gpu.py tests/test_adam_gpu.py -x -q > $O/tests_train.log 2>&1; echo "train tests rc $?"; tail -4 $O/tests_train.log
for a in "--amp bf16" "--amp x3" "" "--amp bf16 --config 2 --model-type dpt_hybrid_384 --batch 4" "--amp bf16 --model-type dpt_swin2_base_384"; do python3 bench.py --train-step $a --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$a', d['value'], d['ms_per_step'], d['launches_per_step'], d['split_ms'], d['train_workspace_gib'])"; done
