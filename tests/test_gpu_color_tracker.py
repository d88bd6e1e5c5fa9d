"""vulcan::ColorTracker on the device (vk_color_*) vs the oracle: image operators,
residuals and Jacobians bit for bit, the 27 sums to float-tree tolerance, the pose
update bit for bit given the same system, Track() against the oracle's loop."""
import ctypes as C

import numpy as np
import pytest

import color_scenes as cs
from test_gpu_parity import api, sync  # noqa: F401  (fixture)
from vulcan_amd import vk_types as T

pytestmark = pytest.mark.gpu


def bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def pair(api, orc):
    k = cs.projection()
    kd, kc = cs.keyframe_images()
    fd, fc = cs.frame_images()
    hk = orc.HostFrame(kd, k, cs.keyframe_pose(), color=kc)
    hf = orc.HostFrame(fd, k, cs.frame_pose(), color=fc)
    hk.compute_normals()
    hf.compute_normals()
    dk = api.Frame(kd, k, cs.keyframe_pose(), color=kc, normals=hk.normals)
    df = api.Frame(fd, k, cs.frame_pose(), color=fc, normals=hf.normals)
    return hk, hf, orc.ColorSide(hk, False), orc.ColorSide(hf, True), dk, df


def test_convert_and_gradients_match(api, orc):
    import torch
    rng = np.random.default_rng(3)
    for (h, w) in ((480, 640), (77, 101), (1, 1), (5, 130)):
        rgb = rng.random((h, w, 3), dtype=np.float32)
        d_rgb = torch.from_numpy(rgb).cuda()
        gray = torch.empty((h, w), dtype=torch.float32, device="cuda")
        api.check(api.lib().vk_color_image_convert(h * w, api._ptr(d_rgb), api._ptr(gray), api.stream()), "convert")
        gx, gy = torch.empty_like(gray), torch.empty_like(gray)
        api.check(api.lib().vk_image_gradients(w, h, api._ptr(gray), api._ptr(gx), api._ptr(gy), api.stream()), "grad")
        sync()
        want = orc.color_convert(rgb)
        assert np.array_equal(bits(gray.cpu().numpy()), bits(want))
        wx, wy = orc.image_gradients(want)
        assert np.array_equal(bits(gx.cpu().numpy()), bits(wx))
        assert np.array_equal(bits(gy.cpu().numpy()), bits(wy))


def test_residuals_and_jacobian_match(api, orc, pair):
    hk, hf, ks, fs, dk, df = pair
    tracker = api.ColorTracker()
    tracker.keyframe = dk
    Tcm = orc.color_tcm(hk, hf)
    assert np.array_equal(np.array(tracker.tcm(df).m[:]), np.array(Tcm.m[:]))
    r = tracker.compute_residuals(df)
    sync()
    assert np.array_equal(bits(r.cpu().numpy()), bits(orc.color_residuals(ks, fs, Tcm)))
    for translation in (True, False):
        tracker.translation_enabled = translation
        J = tracker.compute_jacobian(df)
        sync()
        assert np.array_equal(bits(J.cpu().numpy()), bits(orc.color_jacobian(ks, fs, Tcm, translation)))


def test_system_matches(api, orc, pair):
    hk, hf, ks, fs, dk, df = pair
    tracker = api.ColorTracker()
    tracker.keyframe = dk
    Tcm = orc.color_tcm(hk, hf)
    J = orc.color_jacobian(ks, fs, Tcm, True).reshape(-1, 6).astype(np.float64)
    r = orc.color_residuals(ks, fs, Tcm).reshape(-1).astype(np.float64)
    for translation in (True, False):
        tracker.translation_enabled = translation
        tracker.compute_system(df)
        sync()
        h, g = orc.color_system(ks, fs, Tcm, translation)
        n = 6 if translation else 3
        got_h, got_g = tracker.hessian.cpu().numpy().astype(np.float64), tracker.gradient.cpu().numpy().astype(np.float64)
        scale_g = np.abs(J[:, :n] * r[:, None]).sum(0)
        assert np.all(np.abs(got_g[:n] - g[:n]) <= 2e-5 * scale_g + 1e-12)
        idx = 0
        for rr in range(n):
            for c in range(rr + 1):
                scale = np.abs(J[:, rr] * J[:, c]).sum()
                assert abs(got_h[idx] - h[idx]) <= 2e-5 * scale + 1e-12
                idx += 1
        assert np.all(got_h[idx:] == 0) and np.all(got_g[n:] == 0)


def test_solve_update_matches(api, orc, pair):
    """Same packed system in, same pose and Tcm out (bit for bit)."""
    import torch
    hk, hf, ks, fs, dk, df = pair
    Tcm = orc.color_tcm(hk, hf)
    h, g = orc.color_system(ks, fs, Tcm, True)
    h32, g32 = h.astype(np.float32), g.astype(np.float32)
    key_Twc = (hk.depth_to_color * hk.depth_to_world.inverse()).inverse()
    for translation in (True, False):
        pose = T.ColorPose()
        pose.depth_to_world = hf.depth_to_world
        packed_h = h32 if translation else np.array([h32[0], h32[1], h32[2], h32[3], h32[4], h32[5]], np.float32)
        want_update, _ = orc.color_solve_update(packed_h, g32, hf.depth_to_color, key_Twc, pose, translation)

        dev_pose = torch.from_numpy(np.frombuffer(bytes(_pose_of(hf)), dtype=np.uint8).copy()).cuda()
        dh = torch.zeros(36, dtype=torch.float32, device="cuda")
        dh[: len(packed_h)] = torch.from_numpy(packed_h)
        dg = torch.from_numpy(np.concatenate([g32, np.zeros(0, np.float32)])).cuda()
        state = torch.zeros(2, dtype=torch.int32, device="cuda")
        upd = torch.zeros(6, dtype=torch.float32, device="cuda")
        api.check(api.lib().vk_color_tracker_solve_update(api._ptr(dh), api._ptr(dg), int(translation),
                                                          api._ref(hf.depth_to_color), api._ref(key_Twc), api._ptr(dev_pose),
                                                          api._ptr(state), api._ptr(upd), api.stream()), "solve")
        sync()
        got = T.ColorPose.from_buffer_copy(dev_pose.cpu().numpy().tobytes())
        assert np.array_equal(bits(upd.cpu().numpy()), bits(want_update))
        for name in ("depth_to_world", "Tcm"):
            a, b = getattr(got, name), getattr(pose, name)
            assert np.array_equal(bits(a.m[:]), bits(b.m[:])) and np.array_equal(bits(a.inv[:]), bits(b.inv[:]))
        assert state.cpu().numpy().tolist() == [1, 0]


def _pose_of(frame):
    pose = T.ColorPose()
    pose.depth_to_world = frame.depth_to_world
    return pose


def test_track_follows_the_oracle_loop(api, orc):
    """Track() = BeginSolve + n x (system, solve, update) on the device; the oracle runs the
    same loop with double sums. The keyframe and the frame show the SAME scene here (a curved
    surface with the reference's test texture, seen from a slightly wrong pose): on the
    reference's own pair of unrelated images (color_tracker_test.cu:12-148) Gauss-Newton
    diverges until no pixel overlaps, and both sides ended in NaN poses that compared equal.
    Float tree sums vs double sums differ by 2e-5 relative per system; three steps at a
    condition number of ~1e3 keep the poses within 2e-4."""
    k = cs.projection()
    _, kc = cs.keyframe_images()
    y, x = np.mgrid[0:cs.H, 0:cs.W]
    depth = (1.0 + 0.05 * np.cos(3.0 * x / cs.W) * np.sin(2.0 * y / cs.H)).astype(np.float32)
    hk = orc.HostFrame(depth, k, T.Transform.identity(), color=kc)
    hk.compute_normals()
    start = T.Transform.translate(0.002, -0.001, 0.003) * T.Transform.rotate(0.999995, 0.002, -0.0015, 0.001)
    hf = orc.HostFrame(depth, k, start, color=kc, normals=hk.normals)
    ks, fs = orc.ColorSide(hk, False), orc.ColorSide(hf, True)
    dk = api.Frame(depth, k, T.Transform.identity(), color=kc, normals=hk.normals)
    moved = api.Frame(depth, k, start, color=kc, normals=hk.normals)

    for enabled in (True, False):
        tracker = api.ColorTracker()
        tracker.keyframe = dk
        tracker.max_iterations = 3
        tracker.translation_enabled = enabled
        moved.depth_to_world = start
        before = float((tracker.compute_residuals(moved).double() ** 2).sum())
        got = tracker.track(moved)
        sync()
        after = float((tracker.compute_residuals(moved).double() ** 2).sum())
        assert after < 0.5 * before
        assert tracker.state.cpu().numpy().tolist() == [3, 0]

        pose = T.ColorPose()
        pose.depth_to_world = start
        key_Twc = (hk.depth_to_color * hk.depth_to_world.inverse()).inverse()
        orc.lib().orc_color_tracker_tcm(C.byref(hf.depth_to_color), C.byref(key_Twc), C.byref(pose))
        for _ in range(3):
            h, g = orc.color_system(ks, fs, pose.Tcm, enabled)
            _, norm = orc.color_solve_update(h, g, hf.depth_to_color, key_Twc, pose, enabled)
        assert np.all(np.isfinite(got.matrix()))
        np.testing.assert_allclose(got.matrix(), pose.depth_to_world.matrix(), atol=2e-4)
        np.testing.assert_allclose(got.inverse_matrix(), pose.depth_to_world.inverse_matrix(), atol=2e-4)


def test_fused_begin_and_frame_downsample_match_the_staged_calls(api, orc):
    """vk_color_tracker_begin (BeginSolve of the colour / light trackers as one launch) and
    vk_frame_downsample (Frame::Downsample as one launch) leave the same bits as the staged
    entry points they fuse, also for sizes that are no multiples of their tiles."""
    import torch
    rng = np.random.default_rng(9)
    for (w, h) in ((640, 480), (202, 150), (64, 2)):
        k = T.Projection.make(0.85 * w, 0.85 * w, w / 2, h / 2)
        depth = (1.0 + 0.3 * rng.random((h, w), dtype=np.float32)).astype(np.float32)
        depth[rng.random((h, w)) < 0.05] = 0.0
        kcol = rng.random((h, w, 3), dtype=np.float32)
        fcol = rng.random((h, w, 3), dtype=np.float32)
        key = api.Frame(depth, k, T.Transform.identity(), color=kcol)
        frm = api.Frame(depth, k, T.Transform.translate(0.01, 0.02, -0.01), color=fcol)
        key.compute_normals()
        frm.compute_normals()
        lib, s = api.lib(), api.stream()
        new = lambda *shape: torch.full(shape, -7.0, dtype=torch.float32, device="cuda")

        # staged
        ki, fi, gx, gy, mask = new(h, w), new(h, w), new(h, w), new(h, w), new(h, w)
        api.check(lib.vk_color_image_convert(h * w, api._ptr(key.color), api._ptr(ki), s), "convert")
        api.check(lib.vk_color_image_convert(h * w, api._ptr(frm.color), api._ptr(fi), s), "convert")
        api.check(lib.vk_image_gradients(w, h, api._ptr(fi), api._ptr(gx), api._ptr(gy), s), "gradients")
        api.check(lib.vk_light_compute_frame_mask(C.byref(frm.desc()), 0.2, api._ptr(mask), s), "mask")
        # fused
        ki2, fi2, gx2, gy2, mask2 = new(h, w), new(h, w), new(h, w), new(h, w), new(h, w)
        pose_dev = torch.zeros(64, dtype=torch.float32, device="cuda")
        state = torch.full((2,), 5, dtype=torch.int32, device="cuda")
        start = T.Transform.translate(0.3, -0.2, 0.1) * T.Transform.rotate(0.9998719, 0.0085884, -0.0104268, 0.0085884)
        api.check(lib.vk_color_tracker_begin(C.byref(key.desc()), C.byref(frm.desc()), api._ptr(ki2), api._ptr(fi2),
                                             api._ptr(gx2), api._ptr(gy2), 0.2, api._ptr(mask2), C.byref(start),
                                             api._ptr(pose_dev), api._ptr(state), s), "vk_color_tracker_begin")
        sync()
        for a, b in ((ki, ki2), (fi, fi2), (gx, gx2), (gy, gy2), (mask, mask2)):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (w, h)
        assert pose_dev[:32].cpu().numpy().tobytes() == bytes(start) and state.cpu().tolist() == [0, 0]
        # colour tracker: no mask, no pose
        gx3 = new(h, w)
        api.check(lib.vk_color_tracker_begin(C.byref(key.desc()), C.byref(frm.desc()), api._ptr(ki2), api._ptr(fi2),
                                             api._ptr(gx3), api._ptr(gy2), 0.0, None, None, None, None, s), "begin")
        sync()
        assert torch.equal(gx.view(torch.int32), gx3.view(torch.int32))

        # Frame::Downsample
        d1, c1, n1 = new(h // 2, w // 2), new(h // 2, w // 2, 3), new(h // 2, w // 2, 3)
        api.check(lib.vk_image_downsample(w, h, api._ptr(frm.depth), api._ptr(d1), 1, s), "down")
        api.check(lib.vk_color_image_downsample(w, h, api._ptr(frm.color), api._ptr(c1), 0, s), "down3")
        api.check(lib.vk_color_image_downsample(w, h, api._ptr(frm.normals), api._ptr(n1), 1, s), "down3")
        d2, c2, n2 = new(h // 2, w // 2), new(h // 2, w // 2, 3), new(h // 2, w // 2, 3)
        api.check(lib.vk_frame_downsample(C.byref(frm.desc()), api._ptr(d2), api._ptr(c2), api._ptr(n2), s), "frame down")
        sync()
        for a, b in ((d1, d2), (c1, c2), (n1, n2)):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (w, h)
