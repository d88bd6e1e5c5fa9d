"""Mesh extraction, PLY export and sequence files on the CPU: the
triangle table against the conventions the reference's extractor fixes, the oracle's mesh
against geometry that is known in closed form, and the file formats against restatements of
src/exporter.cpp and include/vulcan/image.h. (Upstream has no test for any of this:
tests/extractor_test.cpp, exporter_test.cpp and mesh_test.cpp are empty.)"""
import itertools
import os

import numpy as np
import pytest

import scenes
from vulcan_amd import io as vio, vk_types as T

CORNER = [(c & 1, (c >> 1) & 1, (c >> 2) & 1) for c in range(8)]
V = [0, 1, 3, 2, 4, 5, 7, 6]          # classic vertex order -> binary corner
EDGE = [(V[0], V[1]), (V[1], V[2]), (V[2], V[3]), (V[3], V[0]), (V[4], V[5]), (V[5], V[6]), (V[6], V[7]), (V[7], V[4]),
        (V[0], V[4]), (V[1], V[5]), (V[2], V[6]), (V[3], V[7])]


def test_triangle_table_follows_the_reference_conventions(orc):
    """extractor.cu:16-118: state bit c <-> corner (c & 1, c >> 1 & 1, c >> 2) with distance > 0;
    edges 0, 3, 8 leave corner 0 along +x, +y, +z; edge_counts[s] = how many of those three are
    cut = [b0 != b1] + [b0 != b2] + [b0 != b4] (the table upstream spells out, :33-50)."""
    assert EDGE[0] == (0, 1) and set(EDGE[3]) == {0, 2} and EDGE[8] == (0, 4)
    upstream_counts_row0 = [0, 3, 1, 2, 1, 2, 2, 1, 0, 3, 1, 2, 1, 2, 2, 1]        # extractor.cu:35
    upstream_counts_row1 = [1, 2, 2, 1, 2, 1, 3, 0, 1, 2, 2, 1, 2, 1, 3, 0]        # extractor.cu:36
    total = 0
    for s in range(256):
        b = [(s >> c) & 1 for c in range(8)]
        owned = (b[0] != b[1]) + (b[0] != b[2]) + (b[0] != b[4])
        assert owned == (upstream_counts_row0 if (s >> 4) % 2 == 0 else upstream_counts_row1)[s & 15]
        tris = orc.mc_triangles(s)
        total += len(tris)
        cut = {e for e, (p, q) in enumerate(EDGE) if b[p] != b[q]}
        used = set(int(e) for e in tris.reshape(-1))
        assert used == cut, s                                   # every cut edge carries a vertex of some triangle
        # closed surface inside the cube: an edge between two vertices is shared by two triangles unless
        # it lies in a cube face (where the neighbouring cube supplies the other triangle)
        seen = {}
        for t in tris:
            for k in range(3):
                key = (int(t[k]), int(t[(k + 1) % 3]))
                seen[key] = seen.get(key, 0) + 1
        for (p, q), n in seen.items():
            assert n == 1 and seen.get((q, p), 0) <= 1         # consistent orientation: a directed edge once
        # normals point to the positive side: for a single positive corner the triangle faces it
    assert total == 820                                         # the classic table's triangle count
    assert len(orc.mc_triangles(0)) == 0 and len(orc.mc_triangles(255)) == 0
    one = orc.mc_triangles(1)
    assert len(one) == 1 and set(int(e) for e in one[0]) == {0, 3, 8}
    mid = [np.mean([CORNER[a] for a in EDGE[int(e)]], axis=0) for e in one[0]]
    normal = np.cross(mid[1] - mid[0], mid[2] - mid[0])
    assert np.dot(normal, np.array([0, 0, 0]) - np.mean(mid, axis=0)) > 0      # towards corner 0, the positive one
    # complementary states cut the same edges with the opposite orientation
    for s in range(256):
        a, b_ = orc.mc_triangles(s), orc.mc_triangles(255 - s)
        assert set(map(int, a.reshape(-1))) == set(map(int, b_.reshape(-1)))


def _fused_plane(orc, depth_value=1.5, frames=3, voxel=0.008):
    w, h = 160, 120
    k = T.Projection.make(136, 136, 80, 60)
    hf = orc.HostFrame(scenes.plane(w, h, depth_value), k, T.Transform.identity(), color=scenes.constant_color(w, h))
    hv = orc.HostVolume(8192, 2048, voxel_length=voxel, truncation_length=0.04)
    for _ in range(6):
        hv.set_view(hf, orc.POLICY_MAXKEY)
    for _ in range(frames):
        orc.integrate_depth(hv, hf)
    return hv, hf


def _check_mesh(points, faces, voxel):
    assert faces.min() >= 0 and faces.max() < len(points)
    assert len(points) - len(np.unique(points, axis=0)) <= len(points) // 10000     # every vertex once: shared, not duplicated
    # manifold with boundary: a directed edge belongs to at most one triangle
    edges = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]])
    packed = edges[:, 0].astype(np.int64) * (1 << 32) + edges[:, 1]
    assert len(np.unique(packed)) == len(packed)
    tri = points[faces]
    normals = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    area = np.linalg.norm(normals, axis=1)
    assert np.all(np.linalg.norm(tri[:, 1] - tri[:, 0], axis=1) <= voxel * 1.75)     # within one cube
    return normals, area


def test_plane_mesh_lies_on_the_plane(orc):
    """A plane fused from the front: the zero crossing of the truncated distance is at
    z = 1.5 along every camera ray; the mesh must lie there, face the camera, and cover the
    visible patch without holes inside it."""
    voxel = 0.008
    hv, hf = _fused_plane(orc)
    points, faces, skipped = orc.extract_mesh(hv, all_allocated=True, interpolate=True)
    assert len(points) > 5000 and len(faces) > 10000
    normals, area = _check_mesh(points, faces, voxel)
    assert np.abs(points[:, 2] - 1.5).max() < 0.3 * voxel           # linear interpolation of a linear field: on the plane
    # normals point towards free space = the positive side = towards the camera (-z)
    big = area > 1e-9
    assert (normals[big, 2] < 0).mean() > 0.999
    # watertight inside the patch: directed edges whose reverse is missing are boundary edges; they must sit
    # on the patch outline, i.e. be few (~perimeter) compared with the interior
    edges = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]])
    fwd = set(map(tuple, edges))
    boundary = [e for e in fwd if (e[1], e[0]) not in fwd]
    assert len(boundary) < 0.05 * len(edges)
    assert skipped == 0
    # midpoint placement (what upstream's active code does, extractor.cu:361): within half a voxel
    mid, mfaces, _ = orc.extract_mesh(hv, all_allocated=True, interpolate=False)
    assert len(mid) == len(points) and np.array_equal(mfaces, faces)
    assert np.abs(mid[:, 2] - 1.5).max() <= 0.5 * voxel + 1e-6
    # visible list (upstream's block source): the same surface, cubes on the outline of the list skipped
    vis_points, vis_faces, vis_skipped = orc.extract_mesh(hv, all_allocated=False, interpolate=True)
    assert 0 < len(vis_faces) <= len(faces)
    _check_mesh(vis_points, vis_faces, voxel)


def test_unknown_voxels_make_holes_not_garbage(orc):
    """extractor.cu:202-212: a cube with a corner that was never integrated is empty."""
    hv, hf = _fused_plane(orc)
    before = orc.extract_mesh(hv, True, True)
    d = hv.voxels["distance"].reshape(-1, 512)
    seen = hv.voxels["distance_weight"].reshape(-1, 512) > 0
    crossing = np.nonzero(((d > 0) & seen).any(axis=1) & ((d <= 0) & seen).any(axis=1))[0]
    slot = int(crossing[len(crossing) // 2])                             # a block the surface passes through
    hv.voxels["distance_weight"][slot * 512:(slot + 1) * 512] = 0       # forget it
    points, faces, _ = orc.extract_mesh(hv, True, True)
    assert 0 < len(faces) < len(before[1])
    _check_mesh(points, faces, 0.008)
    empty = orc.HostVolume(256, 64)
    p, f, s = orc.extract_mesh(empty, True, True)
    assert len(p) == 0 and len(f) == 0 and s == 0


def test_ply_bytes(tmp_path):
    """src/exporter.cpp:19-71 restated: header lines, `x y z c c c`, `3 i j k`."""
    points = np.array([[0.1, -0.25, 0.35], [1.5, 2.0, 0.85], [1e-7, 123456.789, 1.35], [0, 0, 0.6]], dtype=np.float32)
    faces = np.array([[0, 1, 2], [2, 1, 3]], dtype=np.int32)
    path = str(tmp_path / "mesh.ply")
    vio.write_ply(path, points, faces)
    text = open(path).read()
    want = ("ply\nformat ascii 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\n"
            "property uchar red\nproperty uchar green\nproperty uchar blue\nelement face 2\n"
            "property list uchar int vertex_indices\nend_header\n"
            "0.1 -0.25 0.35 0 0 0\n"              # (0.35 - 0.35) / (1.35 - 0.35) = 0
            "1.5 2 0.85 127 127 127\n"            # int(255 * 0.5) = 127
            "1e-07 123457 1.35 255 255 255\n"     # ostream << float: 6 significant digits
            "0 0 0.6 63 63 63\n"                  # int(255 * 0.25) = 63
            "3 0 1 2\n3 2 1 3\n")
    assert text == want
    v, c, f = vio.read_ply(path)
    assert np.array_equal(f, faces) and np.allclose(v, points, rtol=1e-5, atol=1e-6)
    vio.write_ply(path, np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32))
    assert open(path).read().count("\n") == 12


def test_image_files_round_trip(tmp_path):
    """Image::Load / Save, ColorImage::Load / Save (image.h:100-133,228-253; image.cu:213-221,264-273)
    on PGM / PPM: Load = pixel * scale, Save = saturate(round(v * alpha + beta))."""
    rng = np.random.default_rng(5)
    depth = rng.uniform(0.3, 4.0, (48, 64)).astype(np.float32)
    path = str(tmp_path / "d.pgm")
    vio.save_depth(path, depth, bits=16, alpha=1000.0)                    # millimetres
    head = open(path, "rb").read(15)
    assert head.startswith(b"P5\n64 48\n65535\n")
    back = vio.load_depth(path, 0.001)
    assert back.dtype == np.float32 and np.abs(back - depth).max() <= 0.0005 + 1e-6
    assert np.array_equal(back, (np.rint(depth.astype(np.float64) * 1000.0).astype(np.float32) * np.float32(0.001)))
    vio.save_depth(path, np.array([[-3.0, 0.4999, 0.5, 1.5, 2.5, 300.0]]), bits=8)
    assert vio.load_depth(path).tolist() == [[0.0, 0.0, 0.0, 2.0, 2.0, 255.0]]      # round half to even, saturate
    color = rng.uniform(0, 1, (48, 64, 3)).astype(np.float32)
    cpath = str(tmp_path / "c.ppm")
    vio.save_color(cpath, color, alpha=255.0)
    cback = vio.load_color(cpath, 1.0 / 255.0)
    assert cback.shape == (48, 64, 3) and np.abs(cback - color).max() <= 0.5 / 255 + 1e-6
    assert np.array_equal(vio.load_color(path).shape, (1, 6, 3))                   # grey file -> three equal channels


def test_sequence_round_trip(tmp_path):
    k = T.Projection.make(*scenes.APP_INTRINSICS)
    w = vio.SequenceWriter(str(tmp_path / "seq"), 64, 48, k, depth_scale=0.0002)
    rng = np.random.default_rng(9)
    frames = []
    for i in range(3):
        depth = rng.uniform(0.5, 3.0, (48, 64)).astype(np.float32)
        color = rng.uniform(0, 1, (48, 64, 3)).astype(np.float32)
        pose = scenes.orbit_pose(i, 2.0)
        w.append(depth, color, pose)
        frames.append((depth, color, pose))
    w.close()
    r = vio.SequenceReader(str(tmp_path / "seq"))
    assert len(r) == 3 and (r.width, r.height) == (64, 48)
    assert np.allclose(r.depth_projection, [k.fx, k.fy, k.cx, k.cy], rtol=1e-7)
    for i, (depth, color, pose) in enumerate(frames):
        d, c, m = r.frame(i)
        assert np.abs(d - depth).max() <= 0.0001 + 1e-6 and np.abs(c - color).max() <= 0.5 / 255 + 1e-6
        assert np.array_equal(m, pose.matrix().astype(np.float32))
