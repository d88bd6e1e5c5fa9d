"""Experiment: the whole frame with the visible list re-ordered on the host (between SetView and Integrate) so that
list position i — integrated by a workgroup on XCD (i / 4) % 8 — holds a block of one of the two image row bands
(of 16) that the raycast gives to that XCD. Integrate and raycast times from HIP events, with and without."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench, scenes
from vulcan_amd import api, vk_types as T

FRAMES, WARM = 70, 30
poses = [scenes.orbit_pose(i, bench.YAW_STEP) for i in range(FRAMES)]


def xcd_order(vis, band):
    queues = [list(vis[(band % 8) == x][np.argsort(band[(band % 8) == x], kind="stable")]) for x in range(8)]
    out = []
    while len(out) < len(vis):
        x = (len(out) // 4) % 8
        if queues[x]:
            out.append(queues[x].pop(0))
        else:
            rest = [q for q in queues if q]
            out.append(max(rest, key=len).pop(0))
    return np.array(out)


def run(reorder):
    loop = bench.FrameLoop("rgbd", poses)
    vv = loop.vols[0]; vol = vv["vol"]; lib, s = loop.lib, loop.stream
    integ, trace = [], []
    for i in range(FRAMES):
        pose = poses[i]
        loop.fdesc.depth_to_world = pose
        loop.kdesc.depth_to_world = pose
        loop.fdesc.content_id += 2
        vv["tracer"].view_bounds.valid = 0
        loop.prep.normals_out = loop.n_ptr.value
        lib.vk_volume_set_view_rounds(vv["vref"], loop.fref, loop.pprep, 3, s)
        loop.prep.valid = 0
        if reorder is not None:
            torch.cuda.synchronize()
            n = vol.visible_count
            vis = vol.visible_blocks[:n].cpu().numpy().copy()
            origin = vol.host_entries()["block"]["origin"][vis].astype(np.float64)
            centre = (origin + 0.5) * (8 * bench.VOXEL)
            Tdw = pose.inverse_matrix().astype(np.float64)
            p = centre @ Tdw[:3, :3].T + Tdw[:3, 3]
            v = loop.k.fy * p[:, 1] / np.maximum(p[:, 2], 1e-3) + loop.k.cy
            band = np.clip((v * 16 / bench.H).astype(int), 0, 15)
            order = xcd_order(vis, band) if reorder else vis       # (False: the same traffic, the order untouched)
            vol.visible_blocks[:n].copy_(torch.from_numpy(order.astype(np.int32)).cuda())
            torch.cuda.synchronize()
        ev = [loop.make_event() for _ in range(4)]
        lib.vk_event_record(ev[0], s)
        lib.vk_integrate_ahead(vv["vref"], vv["pref"], loop.fref, loop.mode, vv["lref"], loop.m_ptr, loop.r_ptr, vv["bref"], s)
        lib.vk_event_record(ev[1], s)
        lib.vk_event_record(ev[2], s)
        lib.vk_trace_ahead(vv["vref"], loop.kref, vv["bref"], *loop.out_ptrs, s)
        lib.vk_event_record(ev[3], s)
        torch.cuda.synchronize()
        if i >= WARM:
            integ.append(loop.elapsed_ms(ev[0], ev[1]) * 1e3)
            trace.append(loop.elapsed_ms(ev[2], ev[3]) * 1e3)
    return np.mean(integ), np.mean(trace), vol.host_voxels()["distance"].sum()


z = run(None)
print(f"untouched:    integrate {z[0]:.2f} us  raycast + normals {z[1]:.2f} us   (checksum {z[2]:.3f})")
for _ in range(2):
    a = run(False)
    b = run(True)
    print(f"table order:  integrate {a[0]:.2f} us  raycast + normals {a[1]:.2f} us   (checksum {a[2]:.3f})")
    print(f"bands by XCD: integrate {b[0]:.2f} us  raycast + normals {b[1]:.2f} us   (checksum {b[2]:.3f})")
