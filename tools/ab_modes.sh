#!/bin/bash
# A/B of two builds of libsoccdpt_hip.so over the three models in one GPU call: headline leg (mixed) of each, alternating the libraries.
# usage: bash tools/ab_modes.sh <lib_a.so> <lib_b.so> [rounds] [extra bench args, e.g. --precision f16x3]
A=$1; B=$2; R=${3:-2}; shift 3 || true
for cfg in 1 3 2; do
  for i in $(seq 1 $R); do
    for L in "$A" "$B"; do
      SOCCDPT_LIB_PATH=$L python3 bench.py --config $cfg --headline-only --steps 200 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('config $cfg', '$L', d['value'], d['ms_per_step'])"
    done
  done
done
