#!/usr/bin/env python3
"""The bench's own rgbd-icp loop (closed-loop tracking in the room scene) for 60 frames: Gauss-Newton
steps per frame at the full-resolution level, and — with a -DVK_LOOP_TIMING build of the library
(tools/icp_variants.sh) and VK_LOOP_TIMING_DUMP=1 — the phase times of every step on stderr.
Development aid."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
import scenes
from vulcan_amd import vk_types as T

frames = 60
seq = bench.RoomSequence(frames, T.Projection.make(*scenes.APP_INTRINSICS))
loop = bench.FrameLoop("rgbd-icp", seq.truth, sequence=seq)
ev = [(loop.make_event(), loop.make_event()) for _ in range(frames)]
for i in range(frames):
    loop.lib.vk_event_record(ev[i][0], loop.stream)
    loop.step(i)
    loop.lib.vk_event_record(ev[i][1], loop.stream)
torch.cuda.synchronize()
ms = np.array([loop.elapsed_ms(e0, e1) for e0, e1 in ev[10:]])
steps = np.array(loop.gn_steps[9:])
errors = [bench.pose_error(p, seq.truth[i]) for i, p in enumerate(loop.tracked_poses)]
print(f"per frame: median {1e3 * np.median(ms):.1f} us; full-level steps: median {np.median(steps):.0f} (min {steps.min()}, max {steps.max()}); "
      f"pose error max {1e3 * max(e[0] for e in errors):.2f} mm / {max(e[1] for e in errors):.3f} deg")
