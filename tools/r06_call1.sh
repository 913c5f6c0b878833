#!/bin/bash
# round 6, GPU call 1: calibration v2 tests, ViT attention LDS fix, default bench (other_weights), packed-f32 attention A/B, stall-counter passes
set -o pipefail
O=gpurun_out/r06_c1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_calibrate_gpu.py tests/test_hybrid_gpu.py::test_vit_attention_vs_torch tests/test_capi_contract_gpu.py -x -q -s > $O/tests.log 2>&1; echo "tests rc $?" | tee -a $O/tests.log
tail -5 $O/tests.log
timeout -k 10 600 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python3 -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], json.dumps(d.get('other_weights'))[:3000])"
for i in 1 2 3; do
  for L in soccdpt_amd/libsoccdpt_hip.so soccdpt_amd/csrc/build_ab/libsoccdpt_hip_pk.so; do
    for c in 1 3 2; do
      SOCCDPT_LIB_PATH=$GRAFT_REPO_ROOT/$L python3 bench.py --headline-only --steps 200 --config $c 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); k={r['name']:r['ms_per_step'] for r in d['kernels']}; print('$L', 'config $c', d['value'], d['ms_per_step'], 'wattn', k.get('window_attention'), 'vit', k.get('vit_attention'))"
    done
  done
done 2>&1 | tee $O/ab_packed.txt
bash tools/collect_stall.sh r06 2>&1 | tail -5
cat gpurun_out/r06_stall/tiny_stall.txt | head -30
