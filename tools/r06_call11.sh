#!/bin/bash
set -o pipefail
O=gpurun_out/r06_c11
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in dpt_swin2_tiny_256 dpt_hybrid_384; do timeout -k 10 500 python3 tools/headroom_probe.py $m 2>&1 | grep headroom | tee -a $O/headroom_probe.txt; done
