#!/bin/bash
# usage: tools/pmc.sh <outdir> <kbench args...>   — one rocprofv3 pass per counter group
# (each pass bounded by its own timeout; progress is echoed so the run never looks hung)
out=$1; shift
export TMPDIR=/tmp
mkdir -p $out
i=0
while read -r group; do
  i=$((i+1))
  echo "pmc group $i: $group"
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $group --output-format csv -d $out/g$i -o p -- python3 tools/kbench.py "$@" > $out/g$i.log 2>&1 || echo "group $i failed/timeout"
done <<GROUPS
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES
GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
GROUPS
