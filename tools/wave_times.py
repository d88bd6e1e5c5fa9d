#!/usr/bin/env python3
"""Distribution of wave lifetimes of the raycast kernel on the bench scene (tools/probe,
count_points_kernel with wave clocks): how long waves live vs how long the launch lasts."""
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

ROOM = "--room" in sys.argv          # the tracking workload's scene (true poses) instead of the sphere
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
pl = C.CDLL(os.path.join(ROOT, "vulcan_amd", "lib", "libvk_probe.so"))
tiles = (bench.W // 16) * (bench.H // 16)
clocks = torch.zeros(2 * 4 * tiles, dtype=torch.int64, device="cuda")
touched = torch.zeros(vol.max, dtype=torch.uint8, device="cuda")
d2 = torch.zeros((bench.H, bench.W), dtype=torch.float32, device="cuda")
c2 = torch.zeros((bench.H, bench.W, 3), dtype=torch.float32, device="cuda")
F = C.c_float
for use_touched in (False, True):
    for rep in range(3):
        rc = pl.vk_probe_trace_touched(C.c_void_p(vol.hash_entries.data_ptr()), C.c_void_p(vol.voxels.data_ptr()),
                                       C.c_void_p(tracer.bounds.data_ptr()), vol.main, F(8 * bench.VOXEL), F(bench.VOXEL), F(bench.TRUNC),
                                       C.byref(out.depth_to_world), C.byref(k), C.c_void_p(d2.data_ptr()), C.c_void_p(c2.data_ptr()),
                                       bench.W, bench.H, 80, 60, C.c_void_p(touched.data_ptr()) if use_touched else None,
                                       C.c_void_p(clocks.data_ptr()), api.stream())
        assert rc == 0, rc
    torch.cuda.synchronize()
    t = clocks.cpu().numpy().reshape(-1, 2).astype(np.float64)
    t0 = t[:, 0].min()
    start, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0          # us
    life = end - start
    print("counting kernel" if use_touched else "plain kernel (POOL32)")
    print(f"  launch span {end.max():.1f} us; wave start: median {np.median(start):.1f} p99 {np.percentile(start, 99):.1f} max {start.max():.1f}")
    print(f"  wave life: mean {life.mean():.1f} median {np.median(life):.1f} p90 {np.percentile(life, 90):.1f} p99 {np.percentile(life, 99):.1f} max {life.max():.1f}")
    print(f"  wave end: median {np.median(end):.1f} p90 {np.percentile(end, 90):.1f} p99 {np.percentile(end, 99):.1f}")
    lt = life.reshape(30, 40, 4).max(axis=2)
    print("  life by tile row (max over the row):", np.round(lt.max(axis=1), 1))
    if os.environ.get("VK_WAVE_MAP"):
        np.set_printoptions(linewidth=250)
        print("  life per 16x16 tile (max of its 4 waves), rows top to bottom:")
        print(np.round(lt).astype(int))
        print("  wave START per tile (us):")
        print(np.round(start.reshape(30, 40, 4).max(axis=2)).astype(int))
    assert torch.equal(d2, out.depth)
print("blocks touched by rays (Nhit):", int(touched.sum()), "visible:", vol.visible_count)
# trips through the march loop per pixel (the counting kernel again, with its step image)
steps = torch.zeros((bench.H, bench.W), dtype=torch.int32, device="cuda")
pl.vk_probe_trace_steps(C.c_void_p(vol.hash_entries.data_ptr()), C.c_void_p(vol.voxels.data_ptr()),
                        C.c_void_p(tracer.bounds.data_ptr()), vol.main, F(8 * bench.VOXEL), F(bench.VOXEL), F(bench.TRUNC),
                        C.byref(out.depth_to_world), C.byref(k), C.c_void_p(d2.data_ptr()), C.c_void_p(c2.data_ptr()),
                        bench.W, bench.H, 80, 60, C.c_void_p(touched.data_ptr()), C.c_void_p(clocks.data_ptr()),
                        C.c_void_p(steps.data_ptr()), api.stream())
torch.cuda.synchronize()
st = steps.cpu().numpy()
print(f"march trips per ray: mean {st.mean():.1f} median {np.median(st):.0f} p90 {np.percentile(st, 90):.0f} p99 {np.percentile(st, 99):.0f} max {st.max()}")
tile_max = st.reshape(bench.H // 8, 8, bench.W // 8, 8).max(axis=(1, 3))
print(f"per 8x8 tile, the slowest ray: mean {tile_max.mean():.1f} median {np.median(tile_max):.0f} p90 {np.percentile(tile_max, 90):.0f} p99 {np.percentile(tile_max, 99):.0f} max {tile_max.max()}")
b = tracer.bounds.cpu().numpy()
span = (b[..., 1] - b[..., 0])
print(f"bounds span per cell (m): mean {span[span > 0].mean():.2f} p90 {np.percentile(span[span > 0], 90):.2f} max {span.max():.2f}")
if os.environ.get("VK_WAVE_MAP"):
    np.set_printoptions(linewidth=250)
    print("slowest ray per 16x16 tile:")
    print(st.reshape(30, 16, 40, 16).max(axis=(1, 3)))
    print("bounds span (cm) per 2x2 cells:")
    print(np.round(100 * span.reshape(30, 2, 40, 2).max(axis=(1, 3))).astype(int))
