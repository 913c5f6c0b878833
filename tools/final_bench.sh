#!/bin/bash
# Last step of a collection (tools/collect_round.sh <tag> PART=1..3 + tools/install_round.py <tag> come first): the three default bench lines again AFTER the PMC summaries of
# this tree were installed under profiles/ (bench.py fills roofline.traffic only from a file whose csrc_sha matches), then tools/install_round.py <tag> once more.
# usage (through gpurun, from the repo root): bash tools/final_bench.sh r06
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py > gpurun_out/${TAG}_bench_final.json 2> gpurun_out/bf.err
python3 bench.py --config 3 > gpurun_out/${TAG}_base384_bench_final.json 2>> gpurun_out/bf.err
python3 bench.py --config 2 > gpurun_out/${TAG}_hybrid384_bench_final.json 2>> gpurun_out/bf.err
python3 -c "
import json
for t in ('$TAG','${TAG}_base384','${TAG}_hybrid384'):
    d=json.loads(open('gpurun_out/%s_bench_final.json'%t).read().strip().splitlines()[-1]); print(t, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline_hbm']['frac'], d['tolerance']['worst_measured'], d['pipelined']['value'])
"
