#!/bin/bash
set -o pipefail
O=gpurun_out/r06_c12
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 -m pytest tests/test_x2w_gpu.py -x -q > $O/tests_x2w.log 2>&1; echo "x2w tests rc $?"; tail -3 $O/tests_x2w.log
timeout -k 10 300 python3 tools/prec_costs.py dpt_swin2_tiny_256 8 > $O/r06_prec_costs_tiny256.json 2> $O/costs_tiny.err; echo "tiny rc $?"; grep "^head\|^ref0\|^oc0" $O/costs_tiny.err
timeout -k 10 400 python3 tools/prec_costs.py dpt_swin2_base_384 8 > $O/r06_prec_costs_base384.json 2> $O/costs_base.err; echo "base rc $?"; grep "^head" $O/costs_base.err
timeout -k 10 400 python3 tools/prec_costs.py dpt_hybrid_384 4 > $O/r06_prec_costs_hybrid384.json 2> $O/costs_hybrid.err; echo "hybrid rc $?"; grep "^head" $O/costs_hybrid.err
