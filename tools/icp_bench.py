#!/usr/bin/env python3
"""Time per Gauss-Newton iteration of the three trackers at the three pyramid
resolutions of BASELINE configs[3] (320x240, 640x480, 1280x960). Each Track() is one
C call that enqueues `--iterations` steps; the pose is perturbed so the solve does not
converge early. Development aid; bench.py is the contract."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=20)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()

    import torch
    from vulcan_amd import api, vk_types as T

    for (w, h) in ((320, 240), (640, 480), (1280, 960)):
        f = 547.0 * w / 640
        k = T.Projection.make(f, f, w / 2, h / 2)
        y, x = np.mgrid[0:h, 0:w]
        depth = (1.0 + 0.05 * np.cos(3.0 * x / w) * np.sin(2.0 * y / h)).astype(np.float32)
        c = (0.5 + 0.245 * np.cos(3 * np.pi * x / (w - 1)) + 0.245 * np.cos(3 * np.pi * y / (h - 1))).astype(np.float32)
        color = np.repeat(c[:, :, None], 3, axis=2).copy()
        key = api.Frame(depth, k, T.Transform.identity(), color=color)
        key.compute_normals()
        moved = T.Transform.translate(0.004, -0.003, 0.002) * T.Transform.rotate(0.999995, 0.002, -0.0015, 0.001)
        row = []
        for name, cls in (("depth", api.DepthTracker), ("colour", api.ColorTracker), ("light", api.LightTracker)):
            tracker = cls()
            tracker.keyframe = key
            tracker.max_iterations = args.iterations
            if name == "light":
                tracker.light = T.Light.make(2.0, (0.1, 0.0, 0.0))
            frame = api.Frame(key.depth, k, moved, color=key.color, normals=key.normals)
            times = []
            for _ in range(args.reps):
                frame.depth_to_world = moved
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                tracker.track(frame)            # ends with the blocking pose readback
                times.append(time.perf_counter() - t0)
            ran = int(tracker.state.cpu().numpy()[0])
            row.append(f"{name} {1e6 * np.median(times):7.1f} us/Track ({ran} steps run, {1e6 * np.median(times) / args.iterations:5.1f} us/step enqueued)")
        print(f"{w}x{h}: " + " | ".join(row))


if __name__ == "__main__":
    main()
