#!/usr/bin/env python3
"""What the slowest waves of the raycast do, pass by pass (tools/probe, count_points_kernel with PointParams::trip_log):
for every pass through the march loop the clock, how many lanes still march, how many of them stand in a block that is
not there (in their last trip), how many took a sample in their last trip.
    python tools/trip_log.py [--room]"""
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
from vulcan_amd import api, vk_types as T

ROOM = "--room" in sys.argv
PASSES = 96
k = T.Projection.make(*scenes.APP_INTRINSICS)
depth = bench.sphere_room_depth(k)
vol = api.Volume(bench.MAIN, bench.EXCESS, voxel_length=bench.VOXEL, truncation_length=bench.TRUNC)
frame = api.Frame(depth, k, T.Transform.identity(), color=scenes.checker_color(bench.W, bench.H, 0.1, 0.9))
out = api.Frame(torch.zeros((bench.H, bench.W), dtype=torch.float32, device="cuda"), k, T.Transform.identity())
integ, tracer = api.ColorIntegrator(vol), api.Tracer(vol)
for i in range(30):
    if ROOM:
        pose = scenes.room_pose(i)
        d, c = scenes.room_frame(k, pose, bench.W, bench.H)
        frame = api.Frame(d, k, pose, color=c)
        out.depth_to_world = pose
    else:
        frame.depth_to_world = out.depth_to_world = scenes.orbit_pose(i, bench.YAW_STEP)
    vol.set_view(frame, rounds=3)
    integ.integrate(frame)
    tracer.trace(out)
torch.cuda.synchronize()
pl = C.CDLL(os.environ.get("VK_PROBE_LIBRARY") or os.path.join(ROOT, "vulcan_amd", "lib", "libvk_probe.so"))   # (a variant build of the probe: experiments)
tiles = (bench.W // 16) * (bench.H // 16)
waves = 4 * tiles
clocks = torch.zeros(2 * waves, dtype=torch.int64, device="cuda")
touched = torch.zeros(vol.max, dtype=torch.uint8, device="cuda")
log = torch.zeros(waves * PASSES, dtype=torch.int64, device="cuda")
steps = torch.zeros((bench.H, bench.W), dtype=torch.int32, device="cuda")
d2 = torch.zeros((bench.H, bench.W), dtype=torch.float32, device="cuda")
c2 = torch.zeros((bench.H, bench.W, 3), dtype=torch.float32, device="cuda")
F = C.c_float
for rep in range(3):
    log.zero_()
    rc = pl.vk_probe_trace_log(C.c_void_p(vol.hash_entries.data_ptr()), C.c_void_p(vol.voxels.data_ptr()),
                               C.c_void_p(tracer.bounds.data_ptr()), vol.main, F(8 * bench.VOXEL), F(bench.VOXEL), F(bench.TRUNC),
                               C.byref(out.depth_to_world), C.byref(k), C.c_void_p(d2.data_ptr()), C.c_void_p(c2.data_ptr()),
                               bench.W, bench.H, 80, 60, C.c_void_p(touched.data_ptr()), C.c_void_p(clocks.data_ptr()),
                               C.c_void_p(steps.data_ptr()), C.c_void_p(log.data_ptr()), PASSES, api.stream())
    assert rc == 0, rc
torch.cuda.synchronize()
assert torch.equal(d2, out.depth)
t = clocks.cpu().numpy().reshape(-1, 2).astype(np.float64)
t0 = t[:, 0].min()
life = (t[:, 1] - t[:, 0]) / 100.0
words = log.cpu().numpy().view(np.uint64).reshape(waves, PASSES)
print(f"launch span {(t[:, 1].max() - t0) / 100.0:.1f} us (the counting kernel: 64-bit voxel addresses, slower than the product's)")
order = np.argsort(-life)
passes_all = (words != 0).sum(axis=1)
print(f"passes per wave: mean {passes_all.mean():.1f} p90 {np.percentile(passes_all, 90):.0f} p99 {np.percentile(passes_all, 99):.0f} max {passes_all.max()}")
print(f"wave life / passes: mean {np.mean(life / np.maximum(passes_all, 1)):.2f} us per pass over all waves")
for w in order[:6]:
    n = int(passes_all[w])
    row = words[w, :n]
    clock = (row >> np.uint64(32)).astype(np.int64)
    dt = np.diff(clock) & 0xffffffff
    marching = ((row >> np.uint64(24)) & np.uint64(0xff)).astype(int)
    absent = ((row >> np.uint64(16)) & np.uint64(0xff)).astype(int)
    sampling = ((row >> np.uint64(8)) & np.uint64(0xff)).astype(int)
    looked = (row & np.uint64(0xff)).astype(int)
    tile, wave = divmod(int(w), 4)
    print(f"wave {w} (tile row {tile // 40} col {tile % 40}, wave {wave}): life {life[w]:.1f} us, {n} passes, mean {np.mean(dt) / 100.0:.2f} us per pass")
    print("  pass: us | marching | last trip: no block, sampled")
    for i in range(n):
        print(f"  {i:3d}: {(dt[i] / 100.0 if i < n - 1 else float('nan')):5.2f} | {marching[i]:2d} | {absent[i]:2d} {sampling[i]:2d}")
