#!/usr/bin/env python3
"""(Test infrastructure: a fixture generator — it runs the oracle, like make_fixtures.py.) The CPU oracle's OWN tracked loop over the soak's sequence (tools/soak.py leg 2, the looped room sequence), for as many
frames as one cares to wait for: ComputeNormals -> PyramidTracker<DepthTracker>::Track against the previous raycast ->
SetView x3 -> depth + shaded colour -> Trace at the TRACKED pose (apps/vulcan/vulcan.cu:297-325) — the restated reference
kernels, float64 sums, no GPU. Prints the pose error against the ground truth every --every frames and writes
tests/golden/soak_oracle_drift.json: what the soak's drift is compared with (is the slow creep of the tracked pose the
algorithm's, or the device path's?).

  python3 tests/golden/make_soak_oracle_drift.py --frames 2000 --every 100       (about 0.35 s per frame on 8 cores)"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1500)
    ap.add_argument("--every", type=int, default=100)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "soak_oracle_drift.json"))
    args = ap.parse_args()
    import bench
    import make_fixtures as mf
    from oracle import oracle as orc
    from vulcan_amd import vk_types as T
    orc.build()
    orc.set_threads(args.threads)
    k, inputs = mf.soak_inputs(min(args.frames, mf.SOAK_CYCLE))
    light = T.Light.make(2.0, (0.025, 0.08, 0.0))
    hv = orc.HostVolume(65024, 8192, voxel_length=0.005, truncation_length=0.04)
    pose, key, reports, worst = inputs[0][2], None, [], (0.0, 0.0)
    t0 = time.time()
    for i in range(args.frames):
        depth, color, truth = inputs[i % mf.SOAK_CYCLE]
        hf = orc.HostFrame(depth, k, pose, color=color)
        hf.compute_normals()
        if i > 0:
            pose, _ = orc.pyramid_track(key, hf)
        hf.depth_to_world = pose
        for _ in range(3):
            hv.set_view(hf, orc.POLICY_MAXKEY)
        orc.integrate_depth(hv, hf)
        orc.integrate_light_color(hv, hf, light, orc.light_frame_mask(hf, 0.2))
        odepth, ocolor, onormals, _ = orc.trace(hv, hf)
        key = orc.HostFrame(odepth, k, pose, color=ocolor, normals=onormals)
        e = bench.pose_error(pose, truth)
        worst = (max(worst[0], e[0]), max(worst[1], e[1]))
        if (i + 1) % args.every == 0 or i + 1 == args.frames:
            reports.append({"frame": i + 1, "pose_error_max": {"translation_m": worst[0], "rotation_deg": worst[1]},
                            "pose_error_last": {"translation_m": e[0], "rotation_deg": e[1]},
                            "allocated_blocks": int(65024 + 8192 - 1 - hv.counters[T.VK_CTR_VOXEL_PTR]),
                            "dropped_requests": int(hv.counters[T.VK_CTR_DROPPED]), "seconds": time.time() - t0})
            print(json.dumps(reports[-1]), flush=True)
            worst = (0.0, 0.0)
            with open(args.out, "w") as f:
                json.dump({"what": "tests/golden/make_soak_oracle_drift.py: the CPU oracle's tracked loop on the soak's sequence (pose_error_max: "
                                   "over the frames since the previous report)", "threads": args.threads, "reports": reports}, f, indent=1)


if __name__ == "__main__":
    main()
