#!/usr/bin/env python3
"""Per-stage timing of the bench workload (HIP events, many repetitions per
stage on the steady-state volume). Development aid; bench.py is the contract.

  python tools/kbench.py [--frames 40] [--reps 50] [--only integrate,points]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--only", default="")
    ap.add_argument("--inner", type=int, default=10, help="back-to-back launches per timed repetition")
    ap.add_argument("--color", action="store_true", help="RGB-D workload (ColorIntegrator)")
    args = ap.parse_args()

    import torch
    import bench
    import scenes
    from vulcan_amd import api, vk_types as T

    k = T.Projection.make(*scenes.APP_INTRINSICS)
    depth = bench.sphere_room_depth(k)
    color = scenes.checker_color(bench.W, bench.H, 0.1, 0.9) if args.color else None
    poses = [scenes.orbit_pose(i, bench.YAW_STEP) for i in range(args.frames + 1)]
    vol = api.Volume(bench.MAIN, bench.EXCESS, voxel_length=bench.VOXEL, truncation_length=bench.TRUNC)
    frame = api.Frame(depth, k, poses[0], color=color)
    out = api.Frame(torch.zeros((bench.H, bench.W), dtype=torch.float32, device="cuda"), k, poses[0])
    integ = api.ColorIntegrator(vol) if args.color else api.DepthIntegrator(vol)
    light_integ = None
    if args.color:
        frame.compute_normals()
        light_integ = api.LightIntegrator(vol)
        light_integ.light = T.Light.make(2.0, (0.025, 0.08, 0.0))
    tracer = api.Tracer(vol)
    for i in range(args.frames):
        frame.depth_to_world = poses[i]
        out.depth_to_world = poses[i]
        vol.set_view(frame)
        integ.integrate(frame)
        tracer.trace(out)
    torch.cuda.synchronize()
    nvis = vol.visible_count
    print(f"steady state after {args.frames} frames: {nvis} visible blocks, counters {vol.read_counters().tolist()}")

    frame.depth_to_world = poses[args.frames]
    out.depth_to_world = poses[args.frames]
    out2 = torch.zeros((bench.H, bench.W), dtype=torch.float32, device="cuda")
    col2 = torch.zeros((bench.H, bench.W, 3), dtype=torch.float32, device="cuda")

    stages = {
        "set_view": lambda: vol.set_view(frame),
        "reset_vis": lambda: vol.reset_block_visibility(),
        "requests": lambda: vol.create_allocation_requests(frame),
        "handle": lambda: vol.handle_allocation_requests(),
        "visibility": lambda: vol.update_block_visibility(frame),
        "integrate": lambda: integ.integrate(frame),
        "integrate_light": lambda: light_integ.integrate(frame) if light_integ else None,
        "block_bounds": lambda: tracer.compute_block_bounds(frame),
        "points": lambda: tracer.compute_points(frame, out2, col2),
        "normals": lambda: out.compute_normals(),
        "trace": lambda: tracer.trace(out),
        "points_v1": lambda: (api.lib().vk_probe_points_variant(1), tracer.compute_points(frame, out2, col2), api.lib().vk_probe_points_variant(0)),
        "points_v2": lambda: (api.lib().vk_probe_points_variant(2), tracer.compute_points(frame, out2, col2), api.lib().vk_probe_points_variant(0)),
        "points_v3": lambda: (api.lib().vk_probe_points_variant(3), tracer.compute_points(frame, out2, col2), api.lib().vk_probe_points_variant(0)),
        "integ_block": lambda: (api.lib().vk_probe_integrate(None, None, None, 10, None), integ.integrate(frame), api.lib().vk_probe_integrate(None, None, None, 15, None)),
        "integ_pipe4": lambda: (api.lib().vk_probe_integrate(None, None, None, 14, None), integ.integrate(frame)),
        "integ_pipe5": lambda: (api.lib().vk_probe_integrate(None, None, None, 15, None), integ.integrate(frame)),
        "integ_pipe6": lambda: (api.lib().vk_probe_integrate(None, None, None, 16, None), integ.integrate(frame)),
        "integ_pipe8": lambda: (api.lib().vk_probe_integrate(None, None, None, 18, None), integ.integrate(frame), api.lib().vk_probe_integrate(None, None, None, 15, None)),
        "integ_pipe2": lambda: (api.lib().vk_probe_integrate(None, None, None, 12, None), integ.integrate(frame), api.lib().vk_probe_integrate(None, None, None, 15, None)),
        "integ_pipe3": lambda: (api.lib().vk_probe_integrate(None, None, None, 13, None), integ.integrate(frame), api.lib().vk_probe_integrate(None, None, None, 15, None)),
        "integ_v0": lambda: api.check(api.lib().vk_probe_integrate(api._ref(vol.desc()), api._ref(integ.params), api._ref(frame.desc()), 0, api.stream()), "p"),
        "integ_v1": lambda: api.check(api.lib().vk_probe_integrate(api._ref(vol.desc()), api._ref(integ.params), api._ref(frame.desc()), 1, api.stream()), "p"),
        "integ_v2": lambda: api.check(api.lib().vk_probe_integrate(api._ref(vol.desc()), api._ref(integ.params), api._ref(frame.desc()), 2, api.stream()), "p"),
        "integ_v3": lambda: api.check(api.lib().vk_probe_integrate(api._ref(vol.desc()), api._ref(integ.params), api._ref(frame.desc()), 3, api.stream()), "p"),
        "integ_v4": lambda: api.check(api.lib().vk_probe_integrate(api._ref(vol.desc()), api._ref(integ.params), api._ref(frame.desc()), 4, api.stream()), "p"),
        "probe_rmw": lambda: api.check(api.lib().vk_probe_block_rmw(api._ref(vol.desc()), api.stream()), "probe"),
        "probe_rmw1": lambda: (api.lib().vk_probe_block_rmw_mode(1), api.check(api.lib().vk_probe_block_rmw(api._ref(vol.desc()), api.stream()), "probe"), api.lib().vk_probe_block_rmw_mode(0)),
        "probe_rmw2": lambda: (api.lib().vk_probe_block_rmw_mode(2), api.check(api.lib().vk_probe_block_rmw(api._ref(vol.desc()), api.stream()), "probe"), api.lib().vk_probe_block_rmw_mode(0)),
        "probe_rmw3": lambda: (api.lib().vk_probe_block_rmw_mode(3), api.check(api.lib().vk_probe_block_rmw(api._ref(vol.desc()), api.stream()), "probe"), api.lib().vk_probe_block_rmw_mode(0)),
    }
    only = [s for s in args.only.split(",") if s]
    for name, fn in stages.items():
        if only and name not in only:
            continue
        if name == "handle":
            vol.create_allocation_requests(frame)
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        times = []
        for _ in range(args.reps):
            e0.record()
            for _ in range(args.inner):
                fn()
            e1.record()
            e1.synchronize()
            times.append(e0.elapsed_time(e1) * 1e3 / args.inner)
        t = np.array(times)
        extra = ""
        if name == "integrate":
            b = nvis * bench.BYTES_PER_BLOCK + bench.W * bench.H * 4 * (4 if args.color else 1)
            extra = f"  {b / (np.median(t) * 1e-6) / 1e9:7.0f} GB/s algorithmic"
        print(f"{name:13s} median {np.median(t):8.1f} us  min {t.min():8.1f}  max {t.max():8.1f}{extra}")


if __name__ == "__main__":
    main()
