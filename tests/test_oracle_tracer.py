"""Pins the oracle's raycaster against the reference's tracer tests:
  tracer_test.cu:22-166   ComputePatches — 5 hand-placed blocks
  tracer_test.cu:168-244  ComputeBounds  — 6 literal patches
  tracer_test.cu:246-390  ComputePoints  — plane at 1.5 m from a rotated pose,
                          depth +-0.01, colour +-0.001, 2/5 px border exempt
  tracer_test.cu:392-588  ComputeNormals — sphere cap, depth +-0.05, colour
                          +-0.005 away from the rim and the checker edges
"""
import numpy as np
import pytest

import scenes
from vulcan_amd import vk_types as T

F = np.float32
BW, BH = 80, 60


def _tcw():
    return T.Transform.translate(0.3, -1.3, 3.7) * T.Transform.rotate(0.7474, 0.3438, -0.3884, 0.4152)


def test_compute_patches_reference_case(orc):
    block_length, min_depth, max_depth = F(0.008), F(0.1), F(5.0)
    w, h = 640, 480
    tcw = _tcw()
    k = T.Projection.make(346.723, 353.914, 321.294, 239.052)
    M, Minv = tcw.matrix().astype(np.float64), tcw.inverse_matrix().astype(np.float64)
    points = [(320.0, 240.0, 2.5), (120.0, 340.0, 1.5), (420.0, 240.0, 0.6),
              (-20.0, -40.0, 1.0), (720.0, 580.0, 1.0)]

    entries = np.zeros(len(points), dtype=T.hash_entry_dtype)
    expected = []
    for i, (u, v, d) in enumerate(points):
        # tracer_test.cu:54-66 (float32 unproject, then world = Tcw^-1 * Xcp)
        ifx, ify = F(1) / F(k.fx), F(1) / F(k.fy)
        xcp = np.array([ifx * F(u) - F(k.cx) * ifx, ify * F(v) - F(k.cy) * ify, F(1)], np.float32) * F(d)
        xwp = (tcw.inverse_matrix() @ np.append(xcp, F(1)).astype(np.float32)).astype(np.float32)
        b = [np.int16(int(F(c) / block_length)) for c in xwp[:3]]   # short = float / float, truncation
        entries["block"]["origin"][i] = b
        entries["data"][i], entries["next"][i] = 0, -1

        # tracer_test.cu:68-121 expected patches (int bounds, unclamped depth range)
        bmin, bmax = [2**31 - 1, 2**31 - 1], [-2**31, -2**31]
        drng = [np.inf, -np.inf]
        for cz in (0, 1):
            for cy in (0, 1):
                for cx in (0, 1):
                    corner = np.array([block_length * F(cx + b[0]), block_length * F(cy + b[1]),
                                       block_length * F(cz + b[2]), 1.0])
                    xc = M @ corner
                    uu = k.fx * xc[0] / xc[2] + k.cx
                    vv = k.fy * xc[1] / xc[2] + k.cy
                    uu, vv = BW * (uu / w), BH * (vv / h)
                    bmin[0] = min(max(min(int(np.floor(uu)), bmin[0]), 0), BW - 1)
                    bmin[1] = min(max(min(int(np.floor(vv)), bmin[1]), 0), BH - 1)
                    bmax[0] = min(max(max(int(np.ceil(uu)), bmax[0]), 0), BW - 1)
                    bmax[1] = min(max(max(int(np.ceil(vv)), bmax[1]), 0), BH - 1)
                    drng = [min(xc[2], drng[0]), max(xc[2], drng[1])]
        gx = (bmax[0] - bmin[0] + 15) // 16
        gy = (bmax[1] - bmin[1] + 15) // 16
        for j in range(gy):
            oy = bmin[1] + 16 * j
            for kk in range(gx):
                ox = bmin[0] + 16 * kk
                expected.append((ox, oy, min(bmax[0] - ox + 1, 16), min(bmax[1] - oy + 1, 16), drng[0], drng[1]))

    patches, count = orc.compute_patches(np.arange(len(points)), entries, tcw, k, block_length,
                                         min_depth, max_depth, w, h, BW, BH)
    assert count == len(expected) == len(patches)
    remaining = list(expected)
    for p in patches:
        hit = [e for e in remaining if abs(e[4] - p["bounds"][0]) < 1e-4 and abs(e[5] - p["bounds"][1]) < 1e-4
               and (e[0], e[1]) == tuple(p["origin"]) and (e[2], e[3]) == tuple(p["size"])]
        assert hit, p
        remaining.remove(hit[0])
    assert not remaining


def test_compute_bounds_reference_case(orc):
    lit = [((23, 46), (5, 2), (1.237, 1.523)), ((3, 9), (1, 1), (2.021, 3.214)),
           ((20, 43), (5, 8), (0.856, 1.014)), ((0, 0), (2, 2), (1.256, 2.114)),
           ((79, 59), (1, 1), (0.256, 1.314)), ((3, 9), (3, 3), (0.256, 1.314))]
    patches = np.zeros(len(lit), dtype=T.patch_dtype)
    exp = np.zeros((BH, BW, 2), np.float32)
    exp[..., 0], exp[..., 1] = np.finfo(np.float32).max, -np.finfo(np.float32).max
    for i, (o, s, b) in enumerate(lit):
        patches[i] = (o, s, b)
        for yy in range(o[1], o[1] + s[1]):
            for xx in range(o[0], o[0] + s[0]):
                exp[yy, xx, 0] = min(F(b[0]), exp[yy, xx, 0])
                exp[yy, xx, 1] = max(F(b[1]), exp[yy, xx, 1])
    got = orc.compute_bounds(patches, BW, BH)
    assert np.array_equal(got, exp)


def _fuse(orc, depth, color, main, excess):
    k = T.Projection.make(*scenes.TRACER_TEST_INTRINSICS)
    frame = orc.HostFrame(depth, k, scenes.tracer_test_pose(), color=color)
    v = orc.HostVolume(main, excess, voxel_length=0.008, truncation_length=0.04)
    prev = -1
    for _ in range(64):   # tracer_test.cu:298-303
        v.set_view(frame)
        if v.visible_count == prev:
            break
        prev = v.visible_count
    orc.integrate_depth(v, frame)
    orc.integrate_color(v, frame)
    return v, frame


@pytest.fixture(scope="module")
def plane_case(orc):
    orc.set_threads(8)
    w, h = 640, 480
    v, frame = _fuse(orc, scenes.plane(w, h, 1.5), scenes.constant_color(w, h), 4096, 2048)
    out = orc.trace(v, frame, want_steps=True)
    orc.set_threads(1)
    return v, frame, out


def test_compute_points_plane(plane_case):
    v, frame, (depth, color, normals, bounds, steps) = plane_case
    assert np.abs(depth[3:-2, 3:-2] - 1.5).max() < 0.01               # tracer_test.cu:355-370
    err = np.abs(color[6:-5, 6:-5] - np.array([0.1, 0.2, 0.3], np.float32))
    assert err.max() < 0.001                                            # :372-389
    assert steps.max() < 500 and steps[3:-2, 3:-2].min() >= 1


def test_normals_of_raycast_plane(plane_case):
    """frame.cu:9-122 on the raycast depth: a fronto-parallel plane has normal
    (0,0,-1) under the reference's dy x dx convention."""
    v, frame, (depth, color, normals, bounds, steps) = plane_case
    n = normals[8:-8, 8:-8]
    assert np.abs(np.linalg.norm(n, axis=-1) - 1).max() < 1e-5
    assert np.abs(n[..., 2] + 1).max() < 2e-2


def test_compute_normals_sphere(orc):
    """tracer_test.cu:392-588"""
    w, h = 640, 480
    orc.set_threads(8)
    depth_in, color_in = scenes.sphere(w, h), scenes.checker_color(w, h)
    color_in[depth_in == 0] = 0
    v, frame = _fuse(orc, depth_in, color_in, 8192, 4096)
    assert v.counters[T.VK_CTR_DROPPED] == 0
    depth, color, normals, bounds = orc.trace(v, frame)
    orc.set_threads(1)

    y, x = np.mgrid[0:h, 0:w]
    r = np.hypot(x + 0.5 - w / 2, y + 0.5 - h / 2)
    keep = ~((r >= 180) & (r <= 203))
    assert np.abs(depth - depth_in)[keep].max() < 0.05                 # :549-563
    inner = keep & ~((x % 20 < 5) | (x % 20 > 15) | (y % 20 < 5) | (y % 20 > 15))
    assert np.abs(color - color_in)[inner].max() < 0.005               # :565-587
    # normals are unit length wherever the raycast hit (never asserted upstream, "TODO" :456)
    hit = depth > 0
    nn = np.linalg.norm(normals, axis=-1)
    assert np.all((np.abs(nn - 1) < 1e-4)[hit & (r < 170)])
    assert np.all(nn[~hit] == 0)


# ---- the march's 500-step cap (tracer.cu:437-442) ---------------------------------------------------------------

def test_step_cap_is_reached_and_paints_the_pixel_red(orc):
    """tracer.cu:437-442: `if (++iters >= 500) { color = Vector3f(1, 0, 0); break; }` — the loop is left with final_depth
    still 0. No scene a depth camera produces gets there (test_compute_points_plane asserts that for the plane); the
    hand-built slab of tests/scenes.py STEP_CAP_* does: the left half of the image is capped, the right half hits the
    surface behind the slab after some 250 steps."""
    import step_cap
    orc.set_threads(8)
    hv, hf = step_cap.build_host(orc)
    depth, color, normals, bounds, steps = orc.trace(hv, hf, want_steps=True)
    orc.set_threads(1)
    capped, hit = step_cap.classify(depth, color)
    assert capped.sum() > 1200 and hit.sum() > 1200 and (capped | hit).all() and not (capped & hit).any()
    assert np.all(steps[capped] == 500) and np.all(depth[capped] == 0.0)           # :437-442, and final_depth untouched
    assert steps[hit].max() < 500 and steps[hit].min() > 200
    # the hits are the plane z = STEP_CAP_SURFACE in the camera frame, to a voxel
    assert np.abs(depth[hit] - scenes.STEP_CAP_SURFACE).max() < 4 * scenes.STEP_CAP_VOXEL
    # their colour is the mean of the corners' colours (tracer.cu:282-310): the slab's constant colour
    assert np.abs(color[hit] - np.array([0.2, 0.4, 0.6], np.float32)).max() < 1e-6
    w, h = scenes.STEP_CAP_SIZE
    xs = np.nonzero(capped.any(axis=0))[0]
    assert xs.max() < w // 2 and np.nonzero(hit.any(axis=0))[0].min() >= w // 2 - 1   # left half / right half


def test_step_cap_by_hand(orc):
    """The same cap re-derived without the oracle's march, for four capped pixels: replay tracer.cu:358-444 in plain Python
    with what the slab makes of it — outside an allocated block the step is one block length (:431), inside the slab the
    nearest voxel holds STEP_CAP_SDF <= 0.1, the trilinear sample of eight equal corners is that value again (to rounding,
    and far from the branch points 0 and voxel_length / trunc), so the step is max(voxel_length, trunc * sdf) =
    voxel_length (:425) — and count the trips: the 500th comes while the ray is still inside the slab and in front of
    the far bound, so the reference leaves the loop through :437-442."""
    import step_cap
    hv, hf = step_cap.build_host(orc)
    depth, color, normals, bounds, steps = orc.trace(hv, hf, want_steps=True)
    capped, _ = step_cap.classify(depth, color)
    allocated = {tuple(int(c) for c in e["block"]["origin"]) for e in hv.hash_entries if e["data"] >= 0}
    w, h = scenes.STEP_CAP_SIZE
    fx, fy, cx, cy = scenes.STEP_CAP_INTRINSICS
    Twc = hf.depth_to_world.matrix().astype(np.float64)
    Tcw = hf.depth_to_world.inverse_matrix().astype(np.float64)
    block, voxel = 8 * scenes.STEP_CAP_VOXEL, scenes.STEP_CAP_VOXEL
    assert scenes.STEP_CAP_TRUNC * scenes.STEP_CAP_SDF < voxel and scenes.STEP_CAP_SDF <= 0.1
    ys, xs = np.nonzero(capped)
    for i in np.linspace(0, len(ys) - 1, 4).astype(int):
        x, y = int(xs[i]), int(ys[i])
        near, far = bounds[(60 * y) // h, (80 * x) // w]          # tracer.cu:329-331 (80 x 60 grid)
        assert near < far
        cam = np.array([(x + 0.5 - cx) / fx * near, (y + 0.5 - cy) / fy * near, near])
        p = Twc[:3, :3] @ cam + Twc[:3, 3]
        d = Twc[:3, :3] @ cam
        d /= np.linalg.norm(d)
        iters, inside = 0, 0
        while True:
            b = tuple(int(c) for c in np.floor(p / block))
            if b in allocated:
                p = p + voxel * d
                inside += 1
            else:
                p = p + block * d
            z = (Tcw[:3, :3] @ p + Tcw[:3, 3])[2]
            iters += 1
            if iters >= 500:
                break
            assert z < far, "the ray left its bound before the cap"
        assert inside > 400 and steps[y, x] == 500
