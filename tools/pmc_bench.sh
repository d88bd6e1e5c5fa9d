#!/bin/bash
# usage: tools/pmc_bench.sh <outdir> <workload> — SQ / TCP counters of every kernel of the frame loop
# while bench.py runs: one rocprofv3 pass per counter group (each under its own timeout);
# tools/pmc_summary.py prints per-kernel means.
out=${1:-gpurun_out/pmc}; wl=${2:-rgbd}
export TMPDIR=/tmp
mkdir -p $out
i=0
while read -r group; do
  i=$((i+1))
  echo "pmc group $i: $group"
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $group --output-format csv -d $out/g$i -o p -- python3 bench.py --workload $wl --only --steps 30 --warmup 10 --cpu-seconds 0 > $out/g$i.log 2>&1 || echo "group $i failed/timeout"
done <<GROUPS
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum
GROUPS
python3 tools/pmc_summary.py $out
