#!/usr/bin/env python3
"""usage: tools/frame_timeline.py <rocprofv3 output dir> — the tracked frame's launches in stream order with the idle time in
front of each (a kernel trace of `bench.py --workload rgbd-icp --only`): one frame printed, and per launch position the mean
duration and the mean gap over all frames of the steady part. The gap in front of the first launch behind the Track is the
host's round trip for the pose (Tracker::EndSolve + the calls that need the pose as launch arguments)."""
import csv
import glob
import json
import sys
from collections import defaultdict

KINDS = (("track_loop", "L"), ("pyramid_level", "P"), ("create_requests", "R"), ("handle_visibility", "H"), ("integrate_pipelined", "I"),
         ("compute_points", "T"), ("trace_and_pyramid", "TP"), ("trace_and_request", "TR"), ("compute_normals", "N"), ("frame_mask", "M"))
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    kind = next((k for pat, k in KINDS if pat in r["Kernel_Name"]), None)
    if kind:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind))
rows.sort()
rows = rows[len(rows) // 3:]                    # the steady part
# frames start at a pyramid launch
starts = [i for i, r in enumerate(rows) if r[2] == "P"]
frames = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
shape = max(set(tuple(k for _, _, k in fr) for fr in frames), key=lambda s: sum(1 for fr in frames if tuple(k for _, _, k in fr) == s))
same = [fr for fr in frames if tuple(k for _, _, k in fr) == shape]
dur, gap = defaultdict(list), defaultdict(list)
for fi, fr in enumerate(same):
    for j, (s, e, k) in enumerate(fr):
        dur[j].append(e - s)
        if j > 0:
            gap[j].append(s - fr[j - 1][1])
period = [b[0][0] - a[0][0] for a, b in zip(same[:-1], same[1:]) if b[0][0] - a[0][0] < 2e6]
print(f"{f}: {len(same)} frames of the shape {' '.join(shape)}; frame period mean {sum(period) / len(period) / 1e3:.1f} us")
print("launch  mean us  mean gap in front, us")
for j, k in enumerate(shape):
    g = sum(gap[j]) / len(gap[j]) / 1e3 if gap[j] else 0.0
    print(f"{k:4s}   {sum(dur[j]) / len(dur[j]) / 1e3:7.1f}  {g:7.1f}")
print(json.dumps({"frames": len(same), "shape": shape, "period_us": sum(period) / len(period) / 1e3,
                  "sum_kernels_us": sum(sum(dur[j]) / len(dur[j]) for j in range(len(shape))) / 1e3,
                  "sum_gaps_us": sum(sum(g) / len(g) for g in gap.values() if g) / 1e3}))
