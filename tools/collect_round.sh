#!/bin/bash
# One GPU call that collects everything profiles/ holds for a round tag (run through gpurun from the repo root, then tools/install_round.py <tag> here):
#   tools/collect_profiles.sh for the three models (bench JSON, rocprofv3 kernel stats of the headline leg, three separate --pmc passes each),
#   bench JSONs of the other arithmetic modes, the training-step bench lines and rocprofv3 kernel stats of one training step per amp mode.
# usage: bash tools/collect_round.sh r03
set -eo pipefail
TAG=${1:-rXX}
OUT=gpurun_out/${TAG}_extra
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# PART=1: tiny_256 + base_384 forward profiles; PART=2: hybrid_384 + the other arithmetic modes; PART=3 (or TRAIN_ONLY=1): the training-step lines and tables.
# Unset: everything (does not fit one 1200 s gpurun call any more: run the three parts as three calls).
PART=${PART:-all}
if [ -n "$TRAIN_ONLY" ]; then PART=3; fi
if [ "$PART" = "all" ] || [ "$PART" = "1" ]; then
bash tools/collect_profiles.sh ${TAG} > $OUT/collect_tiny.log 2>&1
echo "tiny done"
bash tools/collect_profiles.sh ${TAG}_base384 --config 3 > $OUT/collect_base.log 2>&1
echo "base done"
fi
if [ "$PART" = "all" ] || [ "$PART" = "2" ]; then
bash tools/collect_profiles.sh ${TAG}_hybrid384 --config 2 > $OUT/collect_hybrid.log 2>&1
echo "hybrid done"
for p in f16 f16x3 f32; do python3 bench.py --precision $p --no-cpu-baseline > $OUT/bench_$p.json 2>> $OUT/bench.err; done
for w in salt1 salt2 trained_like; do python3 bench.py --weights $w --no-cpu-baseline --no-other-weights > $OUT/bench_weights_$w.json 2>> $OUT/bench.err; done
python3 bench.py --no-cpu-baseline --no-side-modes --in-flight 3 > $OUT/bench_in_flight3.json 2>> $OUT/bench.err
python3 bench.py --config 2 --precision f16x3 --no-cpu-baseline > $OUT/bench_hybrid384_f16x3.json 2>> $OUT/bench.err
python3 bench.py --config 2 --precision f32 --no-cpu-baseline > $OUT/bench_hybrid384_f32.json 2>> $OUT/bench.err
python3 bench.py --config 3 --precision f16x3 --no-cpu-baseline > $OUT/bench_base384_f16x3.json 2>> $OUT/bench.err
echo "modes done"
fi
if [ "$PART" = "1" ] || [ "$PART" = "2" ]; then exit 0; fi
python3 bench.py --train-step > $OUT/bench_train_step.json 2>> $OUT/bench.err
python3 bench.py --train-step --amp x3 > $OUT/bench_train_step_x3.json 2>> $OUT/bench.err
python3 bench.py --train-step --amp bf16 > $OUT/bench_train_step_amp.json 2>> $OUT/bench.err
python3 bench.py --train-step --amp f16 > $OUT/bench_train_step_amp_f16.json 2>> $OUT/bench.err
python3 bench.py --train-step --batch 3 --encoder-percentage 0.5 --patchwise-percentage 0.5 > $OUT/bench_train_step_B3_enc50_patch50.json 2>> $OUT/bench.err
python3 bench.py --train-step --amp bf16 --batch 3 --encoder-percentage 0.5 --patchwise-percentage 0.5 > $OUT/bench_train_step_amp_B3_enc50_patch50.json 2>> $OUT/bench.err
python3 bench.py --train-step --model-type dpt_swin2_base_384 --batch 8 > $OUT/bench_train_step_base384.json 2>> $OUT/bench.err
python3 bench.py --train-step --model-type dpt_swin2_base_384 --batch 8 --amp bf16 > $OUT/bench_train_step_amp_base384.json 2>> $OUT/bench.err
python3 bench.py --train-step --model-type dpt_hybrid_384 --batch 4 > $OUT/bench_train_step_hybrid384.json 2>> $OUT/bench.err
python3 bench.py --train-step --model-type dpt_hybrid_384 --batch 4 --amp bf16 > $OUT/bench_train_step_amp_hybrid384.json 2>> $OUT/bench.err
echo "train lines done"
for a in 0 x3 bf16; do
  TRAIN_AMP=$a rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_$a -o t -- python3 tools/train_bench.py 8 6 > $OUT/train_$a.json 2>> $OUT/bench.err
  rm -f $OUT/train_$a/t_kernel_trace.csv
done
TRAIN_AMP=bf16 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_hybrid_bf16 -o t -- python3 tools/train_bench.py 4 4 1.0 1.0 dpt_hybrid_384 > $OUT/train_hybrid_bf16.json 2>> $OUT/bench.err
rm -f $OUT/train_hybrid_bf16/t_kernel_trace.csv
echo "all done"
