#!/usr/bin/env python3
"""The integrate kernel past the Infinity Cache, timed by rocprofv3 instead of by bench.py's event pairs (an event pair
costs 2 - 3.5 us of the bracket, profiles/README.md: the in-cache launch reads 34.5 us in the kernel trace and 36.3 us
between events). Runs bench.past_l3 — eight replica volumes in lock step, so a volume's voxels have left the 256 MiB
cache when its turn comes again — and nothing else:

  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pl3 -o p -- python3 tools/past_l3_profile.py [rgbd|depth] > gpurun_out/pl3/run.json
  python3 tools/past_l3_profile.py --parse gpurun_out/pl3

--parse: the average duration of the integrate launches of the TIMED frames in the kernel trace (the last `launches_timed`
launches of the integrate kernel) against the algorithmic bytes of exactly those launches."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(workload):
    import numpy as np
    import torch
    import bench
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scenes
    from vulcan_amd import api
    torch.cuda.set_device(0)
    api.lib()
    warmup, frames = 10, 12
    poses = [scenes.orbit_pose(i, bench.YAW_STEP) for i in range(warmup + frames)]
    nvis, _ = bench.visible_counts(poses)
    out = bench.past_l3(workload, poses, warmup, frames, nvis)
    image = bench.IMAGE_BYTES["depth" if workload == "depth" else "rgbd"]
    alg = [float(nvis[i] * bench.BYTES_PER_BLOCK + image) for i in range(warmup, warmup + frames) for _ in range(out["replica_volumes"])]
    out.update({"workload": workload, "algorithmic_bytes_per_timed_launch": alg, "frac_by_events": out["achieved"] / bench.HBM_PEAK_GBS})
    print(json.dumps(out))


def parse(directory):
    run_json = json.load(open(os.path.join(directory, "run.json")))
    alg = run_json["algorithmic_bytes_per_timed_launch"]
    trace = glob.glob(os.path.join(directory, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(trace)) if "integrate_pipelined_kernel" in r["Kernel_Name"] or "integrate_ring_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-len(alg):]
    ns = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    gbps = sum(alg) / sum(ns)
    print(json.dumps({"workload": run_json["workload"], "kernel": rows[0]["Kernel_Name"][:80], "launches": len(ns),
                      "avg_us_rocprof": sum(ns) / len(ns) / 1e3, "avg_us_between_events": run_json["avg_launch_us"],
                      "algorithmic_GBps_rocprof": gbps, "frac_of_8TBps_rocprof": gbps / 8000.0,
                      "frac_of_8TBps_between_events": run_json["frac_by_events"],
                      "voxel_working_set_bytes": run_json["voxel_working_set_bytes"], "exceeds_l3": run_json["exceeds_l3"]}))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--parse":
        parse(sys.argv[2])
    else:
        run(sys.argv[1] if len(sys.argv) > 1 else "rgbd")
