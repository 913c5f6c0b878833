#!/bin/bash
set -o pipefail
O=gpurun_out/r06_c3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 tools/wattn_qkv_stamps.py 2>&1 | grep -v "^Loading\|amdgpu.ids" | tee $O/wattn_qkv_stamps.txt
for p in f16 mixed; do for m in 0 1 2 3; do SOCCDPT_FUSE_QKV_STAGES=$m python3 bench.py --headline-only --steps 200 --precision $p 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); k={r['name']:(r['ms_per_step'], r['launches_per_step']) for r in d['kernels']}; print('$p stage mask $m', d['value'], d['ms_per_step'], 'wattn', k.get('window_attention'), 'wattn_qkv', k.get('window_attention_qkv'))"; done; done 2>&1 | tee $O/ab_fuse_stage.txt
