#!/usr/bin/env python3
"""Timing of the §8f side rows that are not in the frame loop: Detector::Detect on a 640x480
point cloud and Frame::FilterDepths (HIP events, averages of 50). Development aid."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench, scenes
from vulcan_amd import api, vk_types as T

def timed(fn, n=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s = torch.cuda.current_stream()
    e0.record(s)
    for _ in range(n): fn()
    e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

k = T.Projection.make(*scenes.APP_INTRINSICS)
depth = bench.sphere_room_depth(k)
frame = api.Frame(depth, k, T.Transform.identity())
print(f"FilterDepths 640x480: {timed(frame.filter_depths):.1f} us")
print(f"ComputeNormals 640x480: {timed(frame.compute_normals):.1f} us")
rng = np.random.default_rng(1)
pts = (rng.standard_normal((bench.W * bench.H, 3)) * 0.3 + np.array([0.1, -0.2, 1.5])).astype(np.float32)
d = api.Detector()
dev = torch.from_numpy(pts).cuda()
print(f"Detector::Detect, {len(pts)} points: {timed(lambda: d.enqueue(dev)):.1f} us (8 launches, no readback)")
