"""Experiment: does giving every XCD the blocks of its own image bands speed the light integrate up?
The visible list is re-ordered on the host so that item i — taken by wave i % W, i.e. by a workgroup on
XCD (i / 4) % 8 — belongs to a band of that XCD; the kernel is untouched."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench, scenes
from vulcan_amd import api, vk_types as T

loop = bench.FrameLoop("rgbd", [scenes.orbit_pose(i, bench.YAW_STEP) for i in range(40)])
for i in range(30):
    loop.step(i)
torch.cuda.synchronize()
vv = loop.vols[0]; vol = vv["vol"]; lib, s = loop.lib, loop.stream
n = vol.visible_count
vis = vol.visible_blocks[:n].cpu().numpy().copy()
entries = vol.host_entries()
origin = entries["block"]["origin"][vis].astype(np.float64)
centre = (origin + 0.5) * (8 * bench.VOXEL)
pose = loop.poses[29]
Tdw = pose.inverse_matrix().astype(np.float64)
p = centre @ Tdw[:3, :3].T + Tdw[:3, 3]
k = loop.k
v = k.fy * p[:, 1] / np.maximum(p[:, 2], 1e-3) + k.cy
u = k.fx * p[:, 0] / np.maximum(p[:, 2], 1e-3) + k.cx

def timed(order, reps=40):
    vol.visible_blocks[:n].copy_(torch.from_numpy(order.astype(np.int32)).cuda())
    e0, e1 = loop.make_event(), loop.make_event()
    def launch():
        lib.vk_integrate_ahead(vv["vref"], vv["pref"], loop.fref, loop.mode, vv["lref"], loop.m_ptr, loop.r_ptr, None, s)
    for _ in range(5): launch()
    lib.vk_event_record(e0, s)
    for _ in range(reps): launch()
    lib.vk_event_record(e1, s)
    return loop.elapsed_ms(e0, e1) / reps * 1e3

def xcd_order(band, bands):
    """list position i -> XCD (i // 4) % 8; give each position a block of a band owned by that XCD"""
    queues = [list(vis[(band % 8) == x][np.argsort(band[(band % 8) == x], kind="stable")]) for x in range(8)]
    out, spare = [], []
    i = 0
    while len(out) < n:
        x = (len(out) // 4) % 8
        if queues[x]:
            out.append(queues[x].pop(0))
        else:
            rest = [q for q in queues if q]
            if not rest: break
            out.append(max(rest, key=len).pop(0))
    return np.array(out)

timed(vis, 300)     # reach the steady state of repeated integration of one frame (weights saturate)
print("visible", n, "unsorted (table order):", round(timed(vis), 2), "us")
for bands in (8, 32, 128):
    band = np.clip((v * bands / bench.H).astype(int), 0, bands - 1)
    print(f"{bands} row bands, XCD = band % 8:", round(timed(xcd_order(band, bands)), 2), "us")
tiles = np.clip((v * 8 / bench.H).astype(int), 0, 7) * 8 + np.clip((u * 8 / bench.W).astype(int), 0, 7)
print("64 tiles, XCD = tile % 8:", round(timed(xcd_order(tiles, 64)), 2), "us")
print("sorted by band, contiguous (no XCD mapping):", round(timed(vis[np.argsort(np.clip((v * 32 / bench.H).astype(int), 0, 31), kind='stable')]), 2), "us")
print("unsorted again:", round(timed(vis), 2), "us")
band = np.clip((v * 32 / bench.H).astype(int), 0, 31)
o32 = xcd_order(band, 32)
for _ in range(3):
    print("  unsorted", round(timed(vis), 2), " 32 bands by XCD", round(timed(o32), 2))
