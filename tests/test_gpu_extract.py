"""Mesh extraction on the device (vk_extract_mesh) against the oracle: the reference's
extractor is unfinished and untested, so the oracle states the
finished algorithm and the device must reproduce it exactly — same vertices, same faces,
in the same order (both sides walk blocks, cubes and table entries in one defined order)."""
import numpy as np
import pytest

import scenes
from test_gpu_parity import api, assert_volume_equal, frames, make_pair, sync  # noqa: F401
from vulcan_amd import io as vio, vk_types as T

pytestmark = pytest.mark.gpu


def fused(api, orc, depth, color, k, pose, main=8192, excess=2048, voxel=0.008, rounds=3):
    hf, df = frames(api, orc, depth, k, pose, color=color)
    hv, dv = make_pair(api, orc, main, excess, voxel, 0.04)
    for _ in range(6):
        hv.set_view(hf, orc.POLICY_MAXKEY)
        dv.set_view(df)
    integ = api.ColorIntegrator(dv)
    for _ in range(rounds):
        orc.integrate_depth(hv, hf)
        orc.integrate_color(hv, hf)
        integ.integrate(df)
    return hv, dv


@pytest.mark.parametrize("scene", ["plane", "sphere", "ripple-tilted"])
def test_mesh_matches_oracle(api, orc, scene, tmp_path):
    w, h = 160, 120
    k = T.Projection.make(136, 136, 80, 60)
    if scene == "plane":
        depth, pose = scenes.plane(w, h, 1.5), T.Transform.identity()
    elif scene == "sphere":
        y, x = np.mgrid[0:h, 0:w]
        r2 = (x - 80.0) ** 2 + (y - 60.0) ** 2
        depth = np.where(r2 < 50.0 ** 2, 2.0 - 0.6 * np.sqrt(np.maximum(50.0 ** 2 - r2, 0)) / 50.0, 0.0).astype(np.float32)
        pose = T.Transform.identity()
    else:
        depth, pose = (scenes.ripple(w, h) * 1.3).astype(np.float32), scenes.tracer_test_pose()
    hv, dv = fused(api, orc, depth, scenes.checker_color(w, h, 0.1, 0.9), k, pose)
    assert_volume_equal(dv, hv)
    ex = api.Extractor(dv)
    for all_allocated in (True, False):
        for interpolate in (True, False):
            ex.all_allocated, ex.interpolate = all_allocated, interpolate
            mesh = ex.extract()
            sync()
            got_p, got_f = mesh.host()
            # the visible list is a set (its order is unspecified, volume.cu:80-83): give the oracle the device's order
            hv.visible_blocks[:dv.visible_count] = dv.visible()
            want_p, want_f, want_skipped = orc.extract_mesh(hv, all_allocated, interpolate)
            assert len(want_f) > 1000
            assert got_p.shape == want_p.shape and got_f.shape == want_f.shape
            assert np.array_equal(got_p.view(np.uint32), want_p.view(np.uint32))      # bit for bit
            assert np.array_equal(got_f, want_f)
            assert ex.skipped == want_skipped
            if all_allocated:
                assert ex.skipped == 0 and ex.blocks == int((hv.hash_entries["data"] >= 0).sum())
    # too small a capacity: nothing past it is written, the totals are still reported
    ex.all_allocated, ex.interpolate = True, True
    import torch
    points = torch.full((100, 3), -7.0, dtype=torch.float32, device="cuda")
    faces = torch.full((50, 3), -7, dtype=torch.int32, device="cuda")
    api.check(api.lib().vk_extract_mesh(api._ref(dv.desc()), 1, 1, api._ptr(points[:64]), 64, api._ptr(faces[:32]), 32,
                                        api._ptr(ex.counts), api._ptr(ex.workspace), api.stream()), "vk_extract_mesh")
    sync()
    counts = ex.counts.cpu().numpy()
    assert counts[0] == len(want_p) or counts[0] > 64
    assert torch.all(points[64:] == -7.0) and torch.all(faces[32:] == -7)
    # and out through the exporter
    ex.all_allocated = True
    mesh = ex.extract()
    path = str(tmp_path / f"{scene}.ply")
    vio.write_ply(path, *mesh.host())
    v, c, f = vio.read_ply(path)
    assert len(v) == len(mesh.points) and np.array_equal(f, mesh.faces.cpu().numpy())


def test_full_size_mesh_properties(api, orc):
    """BASELINE sizes (640x480, 5 mm, Volume(65024, 8192)): the 2 m sphere room seen from its
    centre. Every vertex must lie on the sphere, every face must look inwards, no vertex is
    duplicated — properties that need no oracle at this size."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    k = T.Projection.make(*scenes.APP_INTRINSICS)
    depth = bench.sphere_room_depth(k)
    dv = api.Volume(bench.MAIN, bench.EXCESS, voxel_length=bench.VOXEL, truncation_length=bench.TRUNC)
    df = api.Frame(depth, k, T.Transform.identity())
    integ = api.DepthIntegrator(dv)
    for i in range(4):
        df.depth_to_world = scenes.orbit_pose(i, bench.YAW_STEP)
        dv.set_view(df)
        integ.integrate(df)
    ex = api.Extractor(dv)
    ex.all_allocated = True
    mesh = ex.extract()
    sync()
    p, f = mesh.host()
    assert len(p) > 150000 and len(f) > 300000 and ex.skipped == 0
    radius = np.linalg.norm(p.astype(np.float64), axis=1)
    assert np.abs(radius - 2.0).max() < 0.004                 # within a voxel of the sphere
    assert np.percentile(np.abs(radius - 2.0), 99) < 0.0015
    # every vertex once (two vertices coincide only where a voxel's distance is exactly 0 and two of
    # its edges are cut: both land on the voxel itself)
    assert len(p) - len(np.unique(p.view([("x", "f4"), ("y", "f4"), ("z", "f4")]))) <= len(p) // 10000
    tri = p[f].astype(np.float64)
    normals = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    big = np.linalg.norm(normals, axis=1) > 1e-12
    inward = -(tri.mean(axis=1))
    assert ((normals * inward).sum(axis=1)[big] > 0).mean() > 0.999       # towards the camera: the positive side
    # a 2-manifold with boundary: no edge belongs to more than two faces, an edge of two faces is
    # walked once in each direction (consistent orientation), and the edges of ONE face — where the
    # observed surface ends — are few and form closed loops (every boundary vertex has an even degree)
    f64 = f.astype(np.int64)
    directed = np.concatenate([f64[:, [0, 1]], f64[:, [1, 2]], f64[:, [2, 0]]])
    assert (directed[:, 0] != directed[:, 1]).all()
    n = len(p)
    dkey = directed[:, 0] * n + directed[:, 1]
    assert len(np.unique(dkey)) == len(dkey)                      # no directed edge twice
    lo, hi = directed.min(axis=1), directed.max(axis=1)
    ukey, counts = np.unique(lo * n + hi, return_counts=True)
    assert counts.max() == 2
    interior = ukey[counts == 2]
    reverse = directed[:, 1] * n + directed[:, 0]
    both_ways = np.isin(dkey, reverse)
    assert both_ways.sum() == 2 * len(interior)                   # each shared edge: once (a, b), once (b, a)
    boundary = ukey[counts == 1]
    assert len(boundary) < 0.02 * len(ukey)
    ends = np.concatenate([boundary // n, boundary % n])
    _, degree = np.unique(ends, return_counts=True)
    assert (degree % 2 == 0).all()
    # reproducible
    again = ex.extract()
    sync()
    p2, f2 = again.host()
    assert np.array_equal(p2, p) and np.array_equal(f2, f)
