#!/bin/bash
# Load / wait / branch skeleton of one kernel of a built object: shows whether global loads are issued back to back or each under its own branch + s_waitcnt.
# usage: bash tools/isa_loads.sh soccdpt_amd/csrc/build/elementwise.o <mangled-name substring>
set -e
O=$1; K=$2; D=$(mktemp -d); cp $O $D/x.o; (cd $D && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading x.o > /dev/null 2>&1)
F=$(ls $D | grep gfx950 | head -1)
/opt/rocm/lib/llvm/bin/llvm-objdump -d $D/$F --no-show-raw-insn | awk -v k="$K" '$0 ~ k && />:$/{p=1; print} p && /global_load|buffer_load|s_waitcnt vmcnt|s_cbranch|global_store|s_barrier|s_endpgm/{sub(/\/\/.*/,""); print} /s_endpgm/{if(p) exit}'
rm -rf $D
