#!/bin/bash
# One --pmc pass of SQ stall-reason counters per model: eight SQ counters fit one pass.  Outputs under gpurun_out/<tag>_stall/.
# usage (on the GPU box, from the repo root): bash tools/collect_stall.sh r06
set -eo pipefail
TAG=${1:-rXX}
OUT=gpurun_out/${TAG}_stall
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CNT="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS"
for m in tiny:1 hybrid384:2 base384:3; do
  name=${m%%:*}; cfg=${m##*:}
  rocprofv3 --pmc $CNT --output-format csv -d $OUT/$name -o s -- python3 bench.py --headline-only --steps 3 --warmup 1 --prewarm 0 --config $cfg > /dev/null 2> $OUT/$name.err
  f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  python3 tools/pmc_stall_summary.py $f $OUT/${name}_pmc_stall.json > $OUT/${name}_stall.txt
  rm -rf $OUT/$name
  echo "$name stall pass done"
done
