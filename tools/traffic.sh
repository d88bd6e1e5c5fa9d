#!/bin/bash
# usage: tools/traffic.sh [outdir]  — HBM-side bytes of the integrate kernel during bench.py:
# one rocprofv3 pass per counter (FETCH_SIZE, WRITE_SIZE), each under its own timeout;
# tools/traffic_summary.py turns the two passes into profiles/r01_integrate_traffic.json.
out=${1:-gpurun_out/traffic}
export TMPDIR=/tmp
mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  echo "traffic pass: $c"
  rm -rf $out/$c
  timeout -k 5 240 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -o p -- python3 bench.py --steps 60 --warmup 20 --cpu-frames 0 > $out/$c.log 2>&1 || echo "pass $c failed/timeout"
done
python3 tools/traffic_summary.py $out
