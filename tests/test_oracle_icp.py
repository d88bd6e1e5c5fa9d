"""Pins the oracle's ICP residual / Jacobian against the reference's
depth_tracker_test.cu: keyframe depth == 1 at identity (:12-66), frame = rippled
plane at Translate(.001,-.002,.003)*Rotate(.9998719,.0085884,-.0104268,.0085884)
(:68-126), f = 547, c = (320, 240).
  Residuals (:384-422): identical frames -> exactly 0; rippled frame within 1e-6
                        of a float64 restatement (:128-204).
  Jacobian  (:348-382): within 7e-4 of central differences, steps 1e-2 rot /
                        1e-3 trans, through GetTransform (:241-300).
Also the Gauss-Newton update (tracker.cpp:124-163, depth_tracker.cpp:22-86) and
the pyramid downsample (image.cu:101-165).
"""
import numpy as np
import pytest

import scenes
from vulcan_amd import vk_types as T

W, H = 640, 480


def _k():
    return T.Projection.make(547, 547, 320, 240)


def _frame_pose():
    return T.Transform.translate(0.001, -0.002, 0.003) * T.Transform.rotate(0.9998719, 0.0085884, -0.0104268, 0.0085884)


@pytest.fixture(scope="module")
def frames(orc):
    orc.set_threads(8)
    key = orc.HostFrame(scenes.plane(W, H, 1.0), _k(), T.Transform.identity())
    key.compute_normals()
    frm = orc.HostFrame(scenes.ripple(W, H), _k(), _frame_pose())
    frm.compute_normals()
    return key, frm


def _residuals64(key, frm, Twc, base_Twc):
    """depth_tracker_test.cu:128-204 in float64, vectorised."""
    k = key.depth_projection
    K = np.array([[k.fx, 0, k.cx], [0, k.fy, k.cy], [0, 0, 1]], np.float64)
    Kinv = np.array([[1 / k.fx, 0, -k.cx / k.fx], [0, 1 / k.fy, -k.cy / k.fy], [0, 0, 1]], np.float64)
    y, x = np.mgrid[0:H, 0:W]
    uv1 = np.stack([x + 0.5, y + 0.5, np.ones_like(x, np.float64)], -1)
    d = frm.depth.astype(np.float64)
    Xcp = d[..., None] * (uv1 @ Kinv.T)
    M, Mb = Twc.matrix().astype(np.float64), base_Twc.matrix().astype(np.float64)
    Twm, Tmw = key.depth_to_world.matrix().astype(np.float64), key.depth_to_world.inverse_matrix().astype(np.float64)
    Xwp = Xcp @ M[:3, :3].T + M[:3, 3]
    bXwp = Xcp @ Mb[:3, :3].T + Mb[:3, 3]
    bXmp = bXwp @ Tmw[:3, :3].T + Tmw[:3, 3]
    huv = bXmp @ K.T
    ku, kv = huv[..., 0] / huv[..., 2], huv[..., 1] / huv[..., 2]
    ok = (d > 0) & (ku >= 0) & (ku < W) & (kv >= 0) & (kv < H)
    kx, ky = np.clip(ku.astype(np.int64), 0, W - 1), np.clip(kv.astype(np.int64), 0, H - 1)
    kd = key.depth[ky, kx].astype(np.float64)
    ok &= kd > 0
    fn = frm.normals.astype(np.float64) @ M[:3, :3].T
    kn = key.normals[ky, kx].astype(np.float64) @ Twm[:3, :3].T
    ok &= ((kn * kn).sum(-1) > 0) & ((fn * kn).sum(-1) > 0.5)
    fuv1 = np.stack([kx + 0.5, ky + 0.5, np.ones_like(kx, np.float64)], -1)
    Ymp = kd[..., None] * (fuv1 @ Kinv.T)
    Ywp = Ymp @ Twm[:3, :3].T + Twm[:3, 3]
    ok &= ((bXwp - Ywp) ** 2).sum(-1) < 0.05
    return np.where(ok, ((Xwp - Ywp) * kn).sum(-1), 0.0)


def test_residuals_identical_frames_are_zero(orc, frames):
    key, _ = frames
    r = orc.icp_residuals(key, key)
    assert np.all(r == 0)


def test_residuals_vs_float64(orc, frames):
    key, frm = frames
    r = orc.icp_residuals(key, frm)
    exp = _residuals64(key, frm, frm.depth_to_world, frm.depth_to_world)
    assert (r != 0).sum() > 0.9 * W * H
    assert np.abs(r - exp).max() < 1e-6


def test_jacobian_vs_central_differences(orc, frames):
    key, frm = frames
    J = orc.icp_jacobian(key, frm, True)
    base = frm.depth_to_world
    steps = [1e-2] * 3 + [1e-3] * 3
    eye = np.zeros(21, np.float32)
    eye[[0, 2, 5, 9, 14, 20]] = 1    # packed lower triangle of I6
    for i in range(6):
        res = []
        for sgn in (+1, -1):
            upd = np.zeros(6, np.float32)
            upd[i] = sgn * steps[i]
            Twc, x, _ = orc.icp_solve_update(eye, -upd, base, True)   # x = -H^-1 g = upd
            np.testing.assert_allclose(x, upd, atol=1e-9)
            res.append(_residuals64(key, frm, Twc, base))
        fd = (res[0] - res[1]) / (2 * steps[i])
        assert np.abs(fd - J[..., i]).max() < 7e-4, i


def test_system_is_sum_of_outer_products(orc, frames):
    key, frm = frames
    J = orc.icp_jacobian(key, frm, True).reshape(-1, 6).astype(np.float64)
    r = orc.icp_residuals(key, frm).reshape(-1).astype(np.float64)
    Hs, g = orc.icp_system(key, frm, True)
    full = J.T @ J
    packed = np.array([full[i, j] for i in range(6) for j in range(i + 1)])
    np.testing.assert_allclose(Hs, packed, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(g, J.T @ r, rtol=1e-5, atol=1e-7)


def test_gauss_newton_recovers_the_pose(orc, frames):
    """No DepthTracker.Track test exists upstream; LightTracker.Track
    (light_tracker_test.cu:585-669) asserts convergence to the keyframe pose. Here:
    a planar keyframe constrains rotation about x/y and translation along z."""
    key, _ = frames
    true_pose = T.Transform.translate(0.0, 0.0, 0.004) * T.Transform.rotate(0.99999, 0.003, -0.002, 0.0)
    # render what the moved camera sees of the plane z = 1 (closed form)
    R, t = true_pose.matrix()[:3, :3].astype(np.float64), true_pose.matrix()[:3, 3].astype(np.float64)
    y, x = np.mgrid[0:H, 0:W]
    k = key.depth_projection
    rays = np.stack([(x + 0.5 - k.cx) / k.fx, (y + 0.5 - k.cy) / k.fy, np.ones((H, W))], -1)
    depth = ((1.0 - t[2]) / (rays @ R[2])).astype(np.float32)
    frm = orc.HostFrame(depth, _k(), T.Transform.identity())
    frm.compute_normals()
    pose = frm.depth_to_world
    for _ in range(20):   # tracker.cpp:12 max_iterations_
        frm.depth_to_world = pose
        Hs, g = orc.icp_system(key, frm, True)
        Hreg = Hs.copy()
        # the plane leaves x, y translation and z rotation unobservable: the reference would
        # hand a singular H to Eigen. Pin those with a unit diagonal for this test only.
        for i in (2, 3, 4):
            Hreg[i * (i + 1) // 2 + i] += 1.0
        pose, upd, norm = orc.icp_solve_update(Hreg, g, pose, True)
        if norm < 1e-6:
            break
    got = pose.matrix().astype(np.float64)
    assert abs(got[2, 3] - t[2]) < 2e-4
    assert np.abs(got[2, :3] - R[2]).max() < 2e-4


def test_downsample(orc):
    """image.cu:101-165: nearest = top-left of each 2x2, else box mean in the
    kernel's summation order."""
    rng = np.random.default_rng(0)
    img = rng.random((48, 64), dtype=np.float32)
    assert np.array_equal(orc.downsample(img, True), img[::2, ::2])
    box = ((img[0::2, 1::2] + img[0::2, 0::2]) + img[1::2, 1::2] + img[1::2, 0::2]) * np.float32(0.25)
    assert np.array_equal(orc.downsample(img, False), box.astype(np.float32))
    rgb = rng.random((48, 64, 3), dtype=np.float32)
    assert np.array_equal(orc.downsample(rgb, True), rgb[::2, ::2])
    box = ((rgb[0::2, 1::2] + rgb[0::2, 0::2]) + rgb[1::2, 1::2] + rgb[1::2, 0::2]) * np.float32(0.25)
    assert np.array_equal(orc.downsample(rgb, False), box.astype(np.float32))


# ---- pins asked for by the round-1 review ------------------------------------------

def _get_transform_f32(update, Twc):
    """tests/depth_tracker_test.cu:250-300 GetTransform (the reference test's own copy of
    DepthTracker::ApplyUpdate, including Tinc(1,2) = +update[0]) in numpy float32, with the
    reference's operation order: Matrix product accumulates from 0 in index order
    (matrix.h:297-318), Normalize multiplies by 1/sqrt(dot) (matrix.h:131-154)."""
    F = np.float32
    u = np.asarray(update, dtype=F)
    Tinc = np.array([[1, -u[2], +u[1], u[3]],
                     [+u[2], 1, +u[0], u[4]],
                     [-u[1], +u[0], 1, u[5]],
                     [0, 0, 0, 1]], dtype=F)
    A = Twc.matrix().astype(F)
    M = np.zeros((4, 4), dtype=F)
    for r in range(4):
        for c in range(4):
            acc = F(0)
            for n in range(4):
                acc = F(acc + F(Tinc[r, n] * A[n, c]))
            M[r, c] = acc

    def dot(a, b):
        acc = F(0)
        for i in range(3):
            acc = F(acc + F(a[i] * b[i]))
        return acc

    def normalized(a):
        return (a * F(F(1) / np.sqrt(dot(a, a), dtype=F))).astype(F)

    def cross(a, b):
        return np.array([F(a[1] * b[2]) - F(a[2] * b[1]), F(a[2] * b[0]) - F(a[0] * b[2]),
                         F(a[0] * b[1]) - F(a[1] * b[0])], dtype=F)
    x_axis, y_axis = normalized(M[:3, 0]), normalized(M[:3, 1])
    z_axis = cross(x_axis, y_axis)
    y_axis = cross(z_axis, x_axis)
    out = np.eye(4, dtype=F)
    out[:3, 0], out[:3, 1], out[:3, 2], out[:3, 3] = x_axis, y_axis, z_axis, M[:3, 3]
    return out


def test_apply_update_equals_the_reference_test_helper(orc):
    """DepthTracker::ApplyUpdate (depth_tracker.cpp:22-86) vs the helper the reference's own
    Jacobian test perturbs poses with (tests/depth_tracker_test.cu:250-300). The update is
    injected through the solve with H = I, g = -x (an identity LDL^T is exact)."""
    rng = np.random.default_rng(7)
    eye_packed = np.zeros(21, dtype=np.float32)
    eye_packed[[i * (i + 1) // 2 + i for i in range(6)]] = 1.0
    for trial in range(50):
        x = (rng.standard_normal(6) * np.array([1e-2, 1e-2, 1e-2, 1e-3, 1e-3, 1e-3]) * 10 ** rng.uniform(-2, 1)).astype(np.float32)
        q = rng.standard_normal(4)
        q /= np.linalg.norm(q)
        Twc = T.Transform.translate(*rng.uniform(-2, 2, 3)) * T.Transform.rotate(*q)
        got, upd, norm = orc.icp_solve_update(eye_packed, -x, Twc, True)
        assert np.array_equal(upd, x)
        assert np.float32(norm) == np.sqrt(np.sum(x.astype(np.float32) ** 2, dtype=np.float32), dtype=np.float32) \
            or abs(norm - np.linalg.norm(x)) <= 1e-6 * np.linalg.norm(x)
        want = _get_transform_f32(x, Twc)
        # Transform::Translate(t) * Transform::Rotate(R): the product with an identity-rotation
        # translation adds exact zeros, so the 3x4 part is R | t bit for bit
        assert np.array_equal(got.matrix()[:3, :3], want[:3, :3]), trial
        assert np.array_equal(got.matrix()[:3, 3], want[:3, 3]), trial
        # the cached inverse is Rotate(R).Inverse() * Translate(-t) with Rotate's inverse = R^T
        # (transform.h:74-99,146-159). z = x X y is not re-normalised (depth_tracker.cpp:62-65), so
        # R is orthonormal only to first order in the update: that is the reference's behaviour
        assert np.array_equal(got.inverse_matrix()[:3, :3], want[:3, :3].T), trial


def test_ldlt_against_float64_solve(orc):
    """The 6x6 solve is the one piece with no reference-held vector (Eigen, unversioned). Bound
    it instead: float32 LDL^T vs numpy's float64 solve on normal systems J^T J of widely
    varying conditioning, error <= 8 * cond(H) * eps32 * |x| (backward-stable Cholesky-type
    bound with a small constant)."""
    rng = np.random.default_rng(11)
    eps = np.finfo(np.float32).eps
    worst = 0.0
    for trial in range(300):
        n = 6 if trial % 3 else 3
        m = int(rng.integers(20, 2000))
        scale = 10 ** rng.uniform(-3, 1, size=n)               # ill-scaled columns, like X x n vs n
        J = rng.standard_normal((m, n)) * scale
        r = rng.standard_normal(m) * 10 ** rng.uniform(-4, 0)
        H = (J.T @ J).astype(np.float32)
        g = (J.T @ r).astype(np.float32)
        packed = np.zeros(21, dtype=np.float32)
        idx = 0
        for i in range(n):
            for j in range(i + 1):
                packed[idx] = H[i, j]
                idx += 1
        gg = np.zeros(6, dtype=np.float32)
        gg[:n] = g
        _, upd, _ = orc.icp_solve_update(packed, gg, T.Transform.identity(), n == 6)
        H64 = np.tril(H.astype(np.float64)) + np.tril(H.astype(np.float64), -1).T
        x64 = -np.linalg.solve(H64, g.astype(np.float64))
        cond = np.linalg.cond(H64)
        err = np.linalg.norm(upd[:n] - x64) / max(np.linalg.norm(x64), 1e-30)
        worst = max(worst, err / (cond * eps))
        assert err <= 8 * cond * eps, (trial, n, cond, err)
        assert np.all(upd[n:] == 0)
    assert worst > 0


def _seeded_systems():
    rng = np.random.default_rng(11)
    for trial in range(300):
        n = 6 if trial % 3 else 3
        m = int(rng.integers(20, 2000))
        scale = 10 ** rng.uniform(-3, 1, size=n)
        J = rng.standard_normal((m, n)) * scale
        r = rng.standard_normal(m) * 10 ** rng.uniform(-4, 0)
        yield (J.T @ J).astype(np.float32), (J.T @ r).astype(np.float32)


def _room_systems(orc):
    """Normal systems of the closed-loop tracking workload (tests/scenes.py room): consecutive frames
    of the sequence, the frame placed at the previous frame's pose — the first Gauss-Newton step."""
    w, h = 160, 120
    k = T.Projection.make(*(0.25 * np.float32(v) for v in scenes.APP_INTRINSICS))
    for i in range(0, 300, 12):
        a, b = scenes.room_pose(i), scenes.room_pose(i + 1)
        key = orc.HostFrame(scenes.room_depth(k, a, w, h), k, a)
        key.compute_normals()
        frm = orc.HostFrame(scenes.room_depth(k, b, w, h), k, a)
        frm.compute_normals()
        packed, g = orc.icp_system(key, frm, True)
        H = np.zeros((6, 6), dtype=np.float32)
        idx = 0
        for r in range(6):
            for c in range(r + 1):
                H[r, c] = H[c, r] = packed[idx]
                idx += 1
        yield H, np.asarray(g, dtype=np.float32)[:6]


def test_pivoting_stays_inside_the_forward_error_bound(orc):
    """The reference solves with Eigen::LDLT, which pivots on the largest remaining diagonal entry;
    oracle and device factor without pivoting. On symmetric positive definite systems both are
    backward stable, so they may differ by no more than the bound each already meets against
    float64: checked on the 300 seeded systems of the test above and on the normal systems of the
    tracking workload. The worst observed ratio is recorded in DESIGN.md (section 2)."""
    eps = np.finfo(np.float32).eps
    worst = {"seeded": 0.0, "room": 0.0}
    swapped = 0
    for name, systems in (("seeded", _seeded_systems()), ("room", _room_systems(orc))):
        count = 0
        for H, g in systems:
            n = len(g)
            plain = orc.ldlt_solve(H, g, pivoted=False)
            pivoted = orc.ldlt_solve(H, g, pivoted=True)
            H64 = H.astype(np.float64)
            x64 = np.linalg.solve(H64, g.astype(np.float64))
            cond = np.linalg.cond(H64)
            scale = max(np.linalg.norm(x64), 1e-30)
            for x in (plain, pivoted):
                assert np.linalg.norm(x - x64) / scale <= 8 * cond * eps, (name, count, cond)
            diff = np.linalg.norm(plain.astype(np.float64) - pivoted) / scale
            assert diff <= 8 * cond * eps, (name, count, cond, diff)
            worst[name] = max(worst[name], diff / (cond * eps))
            swapped += int(np.argmax(np.abs(np.diag(H))) != 0)
            count += 1
        assert count >= 20
    print(f"|x_plain - x_pivoted| / (cond * eps * |x|): seeded {worst['seeded']:.3f}, room {worst['room']:.3f}")
    assert swapped > 100                         # pivoting really reorders most of these systems
    assert max(worst.values()) < 8.0


def test_pivoted_solve_handles_the_degenerate_cases_like_eigen(orc):
    """A zero matrix solves to 0 (LDLT.h: the first pivot is invalid, D = 0, solve zeroes every
    component); a rank-1 system solves its one observable component."""
    assert np.array_equal(orc.ldlt_solve(np.zeros((6, 6)), np.ones(6), pivoted=True), np.zeros(6, np.float32))
    H = np.zeros((6, 6), np.float32)
    H[3, 3] = 4.0
    g = np.zeros(6, np.float32)
    g[3] = 2.0
    assert np.array_equal(orc.ldlt_solve(H, g, pivoted=True), np.array([0, 0, 0, 0.5, 0, 0], np.float32))
    assert np.array_equal(orc.ldlt_solve(H, g, pivoted=False), np.array([0, 0, 0, 0.5, 0, 0], np.float32))


@pytest.mark.parametrize("translation", [True, False])
def test_empty_system_solves_to_zero(orc, translation):
    """H = 0, g = 0 (no valid correspondence): Eigen's LDLT returns x = 0 (zero pivots are
    skipped), so |update| < 1e-6 stops the loop and the pose is only re-orthonormalised
    (tracker.cpp:153-162). A rank-deficient system must stay finite too."""
    zero_h, zero_g = np.zeros(21, np.float32), np.zeros(6, np.float32)
    start = T.Transform.translate(0.01, 0.02, -0.01) * T.Transform.rotate(0.9998, 0.01, -0.01, 0.012)
    pose, upd, norm = orc.icp_solve_update(zero_h, zero_g, start, translation)
    assert np.all(upd == 0) and norm == 0
    assert np.all(np.isfinite(pose.matrix()))
    np.testing.assert_allclose(pose.matrix(), start.matrix(), atol=1e-6)
    # rank 1: only rotation about x is observable
    h = zero_h.copy()
    g = zero_g.copy()
    h[0], g[0] = 4.0, 2.0
    pose, upd, norm = orc.icp_solve_update(h, g, start, translation)
    assert np.array_equal(upd, np.array([-0.5, 0, 0, 0, 0, 0], np.float32))
    assert np.all(np.isfinite(pose.matrix()))
    # an all-hole frame through the whole loop
    key = orc.HostFrame(scenes.ripple(160, 120), T.Projection.make(136, 136, 80, 60), T.Transform.identity())
    key.compute_normals()
    frm = orc.HostFrame(np.zeros((120, 160), np.float32), T.Projection.make(136, 136, 80, 60), start)
    frm.compute_normals()
    pose, iters = orc.icp_track(key, frm, 20, translation)
    assert iters == 1 and np.all(np.isfinite(pose.matrix()))


def test_pyramid_loop_recovers_the_pose(orc):
    """pyramid_tracker.cpp:52-90 on the oracle: half level (15) then full level (20)."""
    w, h = 320, 240
    y, x = np.mgrid[0:h, 0:w]
    depth = (1.0 + 0.05 * np.cos(3.0 * x / w) * np.sin(2.0 * y / h)).astype(np.float32)
    k = T.Projection.make(273.5, 273.5, 160, 120)
    key = orc.HostFrame(depth, k, T.Transform.identity(), color=scenes.checker_color(w, h))
    key.compute_normals()
    start = T.Transform.translate(0.002, -0.001, 0.003) * T.Transform.rotate(0.999995, 0.002, -0.0015, 0.001)
    frm = orc.HostFrame(depth, k, start, color=scenes.checker_color(w, h))
    frm.compute_normals()
    half = frm.downsample()
    assert (half.width, half.height) == (160, 120) and np.array_equal(half.depth, depth[::2, ::2])
    assert half.depth_projection.fx == np.float32(273.5) * np.float32(0.5)
    pose, iters = orc.pyramid_track(key, frm)
    np.testing.assert_allclose(pose.matrix(), np.eye(4), atol=5e-4)
