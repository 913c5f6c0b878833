#!/bin/bash
set -o pipefail
O=gpurun_out/r06_c10
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -s -k "winograd" > $O/tests_wino.log 2>&1; echo "rc $?"; grep -E "winograd|passed|failed|Error|assert" $O/tests_wino.log | tail -20
