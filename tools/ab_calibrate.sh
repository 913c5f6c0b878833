#!/bin/bash
# Shipped map vs soccdpt_prec_calibrate's (three formats) vs the two-format calibration of round 4, alternated in one GPU call.
#   bash tools/ab_calibrate.sh [rounds] [bench args...]
R=${1:-2}; shift
show='import sys,json; d=json.loads(sys.stdin.read()); c=d["config"]; r=c.get("precision_map_calibration") or {}; print(sys.argv[1], d["value"], d["ms_per_step"], "x3", len(c["precision_map_x3_groups"] or []), "x2w", r.get("n_x2w"), "worst", r.get("worst_calibrated"), "cost_us", r.get("cost_us_calibrated"), "forwards", r.get("forwards"))'
for i in $(seq 1 $R); do
  python bench.py --headline-only "$@" 2>/dev/null | python -c "$show" shipped
  python bench.py --headline-only --calibrate "$@" 2>/dev/null | python -c "$show" calibrated-3fmt
  SOCCDPT_CALIB_NO_X2W=1 python bench.py --headline-only --calibrate "$@" 2>/dev/null | python -c "$show" calibrated-2fmt
done
