#!/bin/bash
# bench.py's headline loop on torch's legacy default stream against a created one (VK_BENCH_STREAM=legacy | created, the
# default since the end of round 5), alternating; then the C++ loop (which creates its own stream)
for r in 1 2 3; do
  for m in ${STREAMS:-legacy created}; do
    export VK_BENCH_STREAM=$m
    python bench.py --only --cpu-seconds 0 --steps 200 --warmup 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$m', round(d['value']), round(1e3*d['ms_per_step'],2))"
  done
done
unset VK_BENCH_STREAM
vulcan_amd/host/bin/fuse_sequence 300 0 0 0 1 | grep "^frames"
