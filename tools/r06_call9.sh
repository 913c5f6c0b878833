#!/bin/bash
set -o pipefail
O=gpurun_out/r06_c9
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 tools/wino_bench.py 2>&1 | grep -v "^Loading\|amdgpu.ids" | tee $O/wino_bench.txt
