#!/bin/bash
# A/B of two builds of libsoccdpt_hip.so on the training step inside ONE GPU call: alternates the two libraries.   usage: bash tools/ab_train.sh <lib_a.so> <lib_b.so> [rounds] [bench args]
A=$1; B=$2; R=${3:-3}; shift 3 || true
for i in $(seq 1 $R); do
  for L in "$A" "$B"; do
    SOCCDPT_LIB_PATH=$L python3 bench.py --train-step --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', d['value'], d['ms_per_step'], d['launches_per_step'], d['split_ms'])"
  done
done
