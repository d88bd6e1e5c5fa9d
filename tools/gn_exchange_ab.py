#!/usr/bin/env python3
"""Pose spread of PyramidTracker<DepthTracker>::Track over repeats of the SAME Track (same keyframe, frame and start
pose): 0 for the product's fixed-order exchange, whatever arrival order does to the float sums for the
-DVK_LOOP_ATOMIC_EXCHANGE build (VK_HIP_LIBRARY selects the build; tools/icp_variants.sh makes them). Development aid."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import scenes
from vulcan_amd import api, vk_types as T

w, h = 640, 480
k = T.Projection.make(*scenes.APP_INTRINSICS)
pose0, pose1 = scenes.room_pose(10), scenes.room_pose(11)
d0, c0 = scenes.room_frame(k, pose0, w, h, light=(2.0, (0.025, 0.08, 0.0)))
d1, c1 = scenes.room_frame(k, pose1, w, h, light=(2.0, (0.025, 0.08, 0.0)))
key = api.Frame(d0, k, pose0, color=c0)
key.compute_normals()
tracker = api.PyramidTracker()
tracker.keyframe = key
poses, steps = [], []
for rep in range(20):
    frame = api.Frame(d1, k, pose0, color=c1)          # starts from the keyframe's pose
    frame.compute_normals()
    got = tracker.track(frame)
    torch.cuda.synchronize()
    poses.append(got.matrix().astype(np.float64))
    steps.append(int(tracker.tracker.state.cpu()[0]))
poses = np.array(poses)
spread = poses.max(axis=0) - poses.min(axis=0)
err = np.abs(poses[0] - pose1.matrix()).max()
print(f"library {os.path.basename(api.LIB_PATH)}: 20 Tracks of one frame: steps at the full level {sorted(set(steps))}, "
      f"max spread of a pose entry {spread.max():.3e}, distinct poses {len({p.tobytes() for p in poses})}, "
      f"|pose - truth| max {err:.3e}")
