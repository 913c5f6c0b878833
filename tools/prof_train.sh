#!/bin/bash
# rocprofv3 kernel statistics of the training step (tools/train_bench.py); run through gpurun from the repo root.
# usage: bash tools/prof_train.sh <tag> [train_bench args]
set -eo pipefail
TAG=${1:-rXX}
ARGS="${@:2}"
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 tools/train_bench.py $ARGS > $OUT/train_bench.json 2> $OUT/train_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o t -- python3 tools/train_bench.py $ARGS > $OUT/train_bench_under_rocprof.json 2>> $OUT/train_bench.err
rm -f $OUT/stats/t_kernel_trace.csv
cat $OUT/train_bench.json
