#!/usr/bin/env python3
"""usage: tools/sequence_overlap.py <rocprofv3 output dir> [frames to skip]
Reads the kernel trace of `bench.py --workload rgbd-icp --sequences S` (one stream = one HSA queue per sequence) and says what
the sequences' chains did to each other ("the trace that shows why"):
  * per kernel: launches, mean duration when it ran ALONE on the device and when another queue's kernel overlapped it;
  * the device's timeline over the steady part: time with 0, 1 and >= 2 kernels in flight;
  * per pair of kernel kinds: how long they were in flight together.
A loop kernel (track_loop_kernel) needs all its workgroups resident; an integrate workgroup takes 40.7 KiB of a CU's 160 KiB
LDS (four fill it) and a raycast workgroup five waves per SIMD: what overlaps in TIME here still shares the CUs."""
import csv
import glob
import json
import sys
from collections import defaultdict

path = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
f = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)[0]
KINDS = (("track_loop", "loop"), ("pyramid_level", "pyramid"), ("create_requests", "requests"), ("handle_visibility", "handle+visibility"),
         ("integrate_pipelined", "integrate"), ("compute_points", "raycast"), ("trace_and_request", "raycast+requests"),
         ("compute_normals", "normals"), ("frame_mask", "mask"))
rows = []
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    kind = next((k for pat, k in KINDS if pat in name), None)
    if kind:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, r.get("Queue_Id", "?")))
rows.sort()
t_first, t_last = rows[0][0], rows[-1][1]
t_from = t_first + skip * (t_last - t_first)
rows = [r for r in rows if r[0] >= t_from]
queues = sorted({r[3] for r in rows})
# sweep: events
events = []
for i, (s, e, k, q) in enumerate(rows):
    events.append((s, 1, i))
    events.append((e, 0, i))
events.sort()
active = set()
depth_time = defaultdict(int)
pair_time = defaultdict(int)
overlapped = [0] * len(rows)             # ns of each kernel during which another QUEUE had a kernel in flight
last = events[0][0]
for t, is_start, i in events:
    dt = t - last
    if dt > 0:
        depth_time[min(len(active), 3)] += dt
        act = list(active)
        for a in act:
            if any(rows[b][3] != rows[a][3] for b in act if b != a):
                overlapped[a] += dt
        kinds = sorted({rows[a][2] for a in act})
        if len(act) >= 2:
            for x in range(len(kinds)):
                for y in range(x, len(kinds)):
                    if x != y or sum(1 for a in act if rows[a][2] == kinds[x]) >= 2:
                        pair_time[(kinds[x], kinds[y])] += dt
    last = t
    (active.add if is_start else active.discard)(i)
total = rows[-1][1] - rows[0][0]
per_kind = defaultdict(lambda: {"n": 0, "alone": [], "shared": []})
for (s, e, k, q), ov in zip(rows, overlapped):
    d = e - s
    per_kind[k]["n"] += 1
    (per_kind[k]["shared"] if ov > 0.5 * d else per_kind[k]["alone"]).append(d)
out = {"trace": f, "queues": queues, "window_us": total / 1e3,
       "device_time_fraction_by_kernels_in_flight": {str(k) if k < 3 else "3+": v / total for k, v in sorted(depth_time.items())},
       "kernels": {}, "in_flight_together_us": {f"{a} + {b}": v / 1e3 for (a, b), v in sorted(pair_time.items(), key=lambda kv: -kv[1])[:12]}}
for k, d in per_kind.items():
    mean = lambda v: (sum(v) / len(v) / 1e3) if v else None
    out["kernels"][k] = {"launches": d["n"], "mean_us_alone": mean(d["alone"]), "launches_alone": len(d["alone"]),
                         "mean_us_overlapped_by_another_queue": mean(d["shared"]), "launches_overlapped": len(d["shared"])}
busy = sum(e - s for s, e, _, _ in rows)
out["sum_of_kernel_durations_over_window"] = busy / total
print(json.dumps(out, indent=1))
