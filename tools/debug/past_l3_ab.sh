#!/bin/bash
# the integrate launches past L3, unprofiled, timed by their own events (tools/past_l3_profile.py), per library build, alternating
for r in 1 2; do
  for lib in "$@"; do
    VK_HIP_LIBRARY=$PWD/$lib python3 tools/past_l3_profile.py rgbd 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', round(d['avg_launch_us'],2), round(d['frac_by_events'],4))"
  done
done
