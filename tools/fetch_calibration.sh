#!/bin/bash
# usage: tools/fetch_calibration.sh <outdir> — FETCH_SIZE (and the fabric request counters behind it) of tools/probe's gather
# kernels, one rocprofv3 pass per counter group; tools/fetch_calibration.py summary turns them into
# <outdir>/fetch_calibration.json (committed as profiles/r06_fetch_calibration.json)
out=${1:-gpurun_out/fetch_calibration}
export TMPDIR=/tmp
mkdir -p $out
i=0
while read -r group; do
  i=$((i+1))
  echo "calibration pass $i: $group"
  rm -rf $out/g$i
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $group --output-format csv -d $out/g$i -o p -- python3 tools/fetch_calibration.py run $out > $out/g$i.log 2>&1 || echo "pass $i ($group) failed/timeout"
done <<GROUPS
FETCH_SIZE
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
GROUPS
python3 tools/fetch_calibration.py summary $out
