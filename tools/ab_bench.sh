#!/bin/bash
# A/B of two builds of libsoccdpt_hip.so inside ONE GPU call (a fresh box differs from the next by more than most changes are worth):
# alternates the two libraries, N rounds each, headline leg only.   usage: bash tools/ab_bench.sh <lib_a.so> <lib_b.so> [rounds] [bench args]
A=$1; B=$2; R=${3:-3}; shift 3 || true
for i in $(seq 1 $R); do
  for L in "$A" "$B"; do
    SOCCDPT_LIB_PATH=$L python3 bench.py --headline-only --steps 300 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', d['value'], d['ms_per_step'])"
  done
done
