#!/usr/bin/env python3
"""Microseconds per Gauss-Newton step of PyramidTracker<DepthTracker> in the bench's own
rgbd-icp loop (tracking a frame against the previous raycast: 15 + 20 steps, the update
never drops below 1e-6), from HIP events around the tracking calls. Development aid."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
import scenes

frames = 60
poses = [scenes.orbit_pose(i, bench.YAW_STEP) for i in range(frames)]
loop = bench.FrameLoop("rgbd-icp", poses)
lib, s = loop.lib, loop.stream
ev = [(loop.make_event(), loop.make_event()) for _ in range(frames)]
steps = []
t = loop.tracker.tracker
for i in range(frames):
    if i > 0:
        a = loop.track_args
        lib.vk_event_record(ev[i][0], s)
        lib.vk_transform_upload(loop.pose_dev, C.byref(poses[i - 1]), s)
        lib.vk_icp_pyramid_track(a[0], C.byref(poses[i - 1]), *a[2:], s)
        lib.vk_event_record(ev[i][1], s)
        lib.vk_memcpy_d2h(C.byref(loop.tracked), loop.pose_dev, 128, s)
        steps.append(int(t.state.cpu()[0]))
    saved, loop.tracker = loop.tracker, None
    loop.step(i)
    loop.tracker = saved
torch.cuda.synchronize()
ms = np.array([loop.elapsed_ms(e0, e1) for e0, e1 in ev[10:]])
print(f"track: median {1e3 * np.median(ms):.1f} us, full-level steps run {np.median(steps[9:]):.0f} of 20 "
      f"-> {1e3 * np.median(ms) / (15 + np.median(steps[9:])):.2f} us per step over both levels (incl. the pyramid launch)")
print("per frame (us, full-level steps):", [(round(1e3 * float(m)), int(n)) for m, n in zip(ms[:16], steps[9:25])])
err = np.abs(loop.tracked.matrix() - poses[frames - 2].matrix()).max()
print(f"last tracked pose vs the keyframe's pose: max |diff| {err:.2e}")
