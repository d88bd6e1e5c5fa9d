#!/bin/bash
# usage: tools/traffic.sh <outdir> <workload: rgbd|depth>  — HBM-side bytes of the integrate and raycast
# kernels during bench.py: one rocprofv3 pass per counter (FETCH_SIZE, WRITE_SIZE: they do not fit one
# pass), each under its own timeout; tools/traffic_summary.py turns the two passes into the
# profiles/rNN_*_traffic.json files bench.py reports as roofline.traffic.
out=${1:-gpurun_out/traffic}; wl=${2:-rgbd}
export TMPDIR=/tmp
mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  echo "traffic pass: $c ($wl)"
  rm -rf $out/$c
  timeout -k 5 240 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -o p -- python3 bench.py --workload $wl --only --steps 60 --warmup 20 --cpu-seconds 0 > $out/$c.log 2>&1 || echo "pass $c failed/timeout"
done
python3 tools/traffic_summary.py $out $wl
