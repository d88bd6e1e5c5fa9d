#!/usr/bin/env python3
"""Does the photometric loop drift because of the kernels or because of the algorithm?

The app's frame loop with the LightTracker closed around fusion (apps/vulcan/vulcan.cu:89-111,297-325) in the room
scene, run twice on the same input: by the CPU oracle (the restated reference) and by the device. Per frame: the
pose error of each against the ground truth, and the largest difference between the two poses. If both drift
alike, the drift is the reference algorithm's answer to this scene (a lamp-lit, mostly dark room, tracked
frame-to-model with a photometric residual), not something the HIP kernels add.

    python tools/light_loop_drift.py [--frames 40] [--steps 20] [--size 320x240]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

LIGHT = (2.0, (0.025, 0.08, 0.0))


def pose_error(got, truth):
    d = got.matrix().astype(np.float64) @ truth.inverse_matrix().astype(np.float64)
    angle = np.degrees(np.arccos(np.clip((np.trace(d[:3, :3]) - 1.0) / 2.0, -1.0, 1.0)))
    return float(np.linalg.norm(got.matrix()[:3, 3].astype(np.float64) - truth.matrix()[:3, 3])), float(angle)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--steps", type=int, default=20, help="Gauss-Newton steps per frame (the app: 1)")
    ap.add_argument("--size", default="320x240")
    ap.add_argument("--tracker", default="light", choices=["light", "depth"])
    ap.add_argument("--cycle", type=int, default=240, help="frames per cycle of the camera's path (240: the bench's motion)")
    ap.add_argument("--ambient", type=float, default=0.0, help="ambient term added to the shaded colour (0: the lamp alone)")
    args = ap.parse_args()
    import torch
    import scenes
    from oracle import oracle as orc
    from vulcan_amd import api, vk_types as T
    w, h = (int(v) for v in args.size.split("x"))
    k = T.Projection.make(*(np.float32(w / 640.0) * np.float32(v) for v in scenes.APP_INTRINSICS))
    light = T.Light.make(*LIGHT)
    params = T.Integrator(0.1, 5.0, 100.0, 16.0)             # vulcan.cu:92-93
    truth = [scenes.room_pose(30 * args.cycle // 240 + i, frames_per_cycle=args.cycle) for i in range(args.frames)]
    inputs = [scenes.room_frame(k, p, w, h, light=LIGHT) for p in truth]
    orc.set_threads(min(16, os.cpu_count() or 1))

    hv = orc.HostVolume(65024, 8192, voxel_length=0.005, truncation_length=0.04)
    dv = api.Volume(65024, 8192, voxel_length=0.005, truncation_length=0.04)
    integ, tracer = api.LightIntegrator(dv), api.Tracer(dv)
    integ.light, integ.params = light, params
    if args.tracker == "light":
        tracker = api.LightTracker()
        tracker.light, tracker.max_iterations = light, args.steps
    else:
        tracker = api.PyramidTracker()
    keys = [api.Frame(torch.zeros((h, w), dtype=torch.float32, device="cuda"), k, truth[0],
                      color=torch.zeros((h, w, 3), dtype=torch.float32, device="cuda"),
                      normals=torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")) for _ in range(2)]
    opose = dpose = truth[0]
    hkey = None
    print(f"# {args.tracker} tracker, {args.steps} steps per frame, {w}x{h}, {args.frames} frames of the room sequence, "
          f"{args.cycle} frames per cycle")
    print("# frame | oracle error mm / deg | device error mm / deg | max |device pose - oracle pose| | dropped requests (device)")
    for i, (depth, color) in enumerate(inputs):
        hf = orc.HostFrame(depth, k, opose, color=color)
        hf.compute_normals()
        df = api.Frame(depth, k, dpose, color=color)
        df.compute_normals()
        if i > 0:
            if args.tracker == "light":
                opose, _ = orc.light_track(hkey, hf, light, args.steps)
            else:
                opose, _ = orc.pyramid_track(hkey, hf)
            tracker.keyframe = keys[(i - 1) & 1]
            dpose = tracker.track(df)
        hf.depth_to_world = opose
        df.depth_to_world = dpose
        for _ in range(3):
            hv.set_view(hf, orc.POLICY_MAXKEY)
        orc.integrate_depth(hv, hf, params)
        orc.integrate_light_color(hv, hf, light, orc.light_frame_mask(hf, 0.2), params)
        odepth, ocolor, onormals, _ = orc.trace(hv, hf)
        hkey = orc.HostFrame(odepth, k, opose, color=ocolor, normals=onormals)
        out = keys[i & 1]
        out.depth_to_world = dpose
        dv.set_view(df, rounds=3)
        integ.integrate(df)
        tracer.trace(out)
        torch.cuda.synchronize()
        eo, ed = pose_error(opose, truth[i]), pose_error(dpose, truth[i])
        apart = float(np.abs(dpose.matrix() - opose.matrix()).max())
        print(f"{i:3d} | {eo[0] * 1e3:6.2f} {eo[1]:6.3f} | {ed[0] * 1e3:6.2f} {ed[1]:6.3f} | {apart:.1e} | {int(dv.read_counters()[T.VK_CTR_DROPPED])}")


if __name__ == "__main__":
    main()
