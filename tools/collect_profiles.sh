#!/bin/bash
# Collects, on the GPU box, everything profiles/ holds for one round tag (run through gpurun from the repo root):
#   bench JSON of the default command, rocprofv3 --kernel-trace --stats of the HEADLINE leg only (bench.py --headline-only: no fp16 /
#   two-stream / CPU legs, so the per-kernel averages are those of the benchmarked launches alone), and three separate --pmc passes
#   (FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES+GRBM_GUI_ACTIVE) over a short run, as MI355X_MICROARCH.md prescribes.
# usage: bash tools/collect_profiles.sh r01f     (outputs under gpurun_out/<tag>/; copy the summaries into profiles/ afterwards)
set -eo pipefail
TAG=${1:-rXX}
EXTRA="${@:2}"   # e.g. --model-type dpt_hybrid_384 --batch 4
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py $EXTRA > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 bench.py --headline-only $EXTRA > $OUT/bench_under_rocprof.json 2>> $OUT/bench.err
rm -f $OUT/stats/s_kernel_trace.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 bench.py --headline-only --steps 3 --warmup 1 --prewarm 0 $EXTRA > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 bench.py --headline-only --steps 3 --warmup 1 --prewarm 0 $EXTRA > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o m -- python3 bench.py --headline-only --steps 3 --warmup 1 --prewarm 0 $EXTRA > /dev/null 2>&1
# round 6: one pass of SQ stall-reason counters (eight SQ slots: MI355X_MICROARCH.md "rocprofv3 PMC slots") -> <tag>_pmc_stall.json (tools/pmc_stall_summary.py)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_stall -o s -- python3 bench.py --headline-only --steps 3 --warmup 1 --prewarm 0 $EXTRA > /dev/null 2>&1
python3 tools/pmc_stall_summary.py $(find $OUT/pmc_stall -name "*counter_collection.csv" | head -1) $OUT/pmc_stall.json > $OUT/pmc_stall.txt
rm -rf $OUT/pmc_stall
ls -la $OUT $OUT/stats
