#!/bin/bash
# A/B of an environment switch inside one GPU call: bash tools/ab_env.sh VAR=VALUE [rounds] [bench args...] -> alternating default / switched headline runs
SW=$1; R=${2:-3}; shift; shift
for i in $(seq 1 $R); do
  python bench.py --headline-only "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default ', d['value'], d['ms_per_step'])"
  env $SW python bench.py --headline-only "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$SW', d['value'], d['ms_per_step'])"
done
