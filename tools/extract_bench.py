import os, sys, ctypes as C
ROOT = "/root/repo" if os.path.isdir("/root/repo/tests") else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench, scenes
from vulcan_amd import api, vk_types as T
k = T.Projection.make(*scenes.APP_INTRINSICS)
depth = bench.sphere_room_depth(k)
vol = api.Volume(bench.MAIN, bench.EXCESS, voxel_length=bench.VOXEL, truncation_length=bench.TRUNC)
frame = api.Frame(depth, k, T.Transform.identity(), color=scenes.checker_color(bench.W, bench.H, 0.1, 0.9))
integ = api.ColorIntegrator(vol)
for i in range(30):
    frame.depth_to_world = scenes.orbit_pose(i, bench.YAW_STEP)
    vol.set_view(frame); integ.integrate(frame)
torch.cuda.synchronize()
for all_alloc in (False, True):
    ex = api.Extractor(vol); ex.all_allocated = all_alloc
    m = ex.extract(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s = torch.cuda.current_stream()
    e0.record(s)
    for _ in range(20): m = ex.extract()
    e1.record(s); torch.cuda.synchronize()
    p, f = m.host()
    print(f"all_allocated={all_alloc}: {e0.elapsed_time(e1)/20*1e3:.1f} us per extraction, {len(p)} points, {len(f)} faces, blocks {vol.visible_count if not all_alloc else 'all'}")
