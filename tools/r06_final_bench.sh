#!/bin/bash
# the three bench lines again AFTER the PMC summaries of this tree were installed under profiles/ (bench.py fills roofline.traffic only from a matching file)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py > gpurun_out/r06_bench_final.json 2> gpurun_out/bf.err
python3 bench.py --config 3 > gpurun_out/r06_base384_bench_final.json 2>> gpurun_out/bf.err
python3 bench.py --config 2 > gpurun_out/r06_hybrid384_bench_final.json 2>> gpurun_out/bf.err
python3 -c "
import json
for t in ('r06','r06_base384','r06_hybrid384'):
    d=json.loads(open('gpurun_out/%s_bench_final.json'%t).read().strip().splitlines()[-1]); print(t, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline_hbm']['frac'], d['tolerance']['worst_measured'], d['pipelined']['value'])
"
