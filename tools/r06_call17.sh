#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_c17
bash tools/ab_train.sh $GRAFT_REPO_ROOT/soccdpt_amd/csrc/build_ab/libsoccdpt_hip_head.so $GRAFT_REPO_ROOT/soccdpt_amd/libsoccdpt_hip.so 3 --amp bf16 2>&1 | tee gpurun_out/r06_c17/ab_train_deferred_reduce.txt
bash tools/ab_train.sh $GRAFT_REPO_ROOT/soccdpt_amd/csrc/build_ab/libsoccdpt_hip_head.so $GRAFT_REPO_ROOT/soccdpt_amd/libsoccdpt_hip.so 2 --amp x3 2>&1 | tee -a gpurun_out/r06_c17/ab_train_deferred_reduce.txt
