#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_c14
timeout -k 10 900 python3 tools/derive_shipped_maps.py 2>/dev/null | grep -v "^Loading\|^Model loaded\|soccdpt_amd:" | tee gpurun_out/r06_c14/shipped_maps.txt | cut -c1-400
