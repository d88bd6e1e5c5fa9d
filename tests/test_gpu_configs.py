"""BASELINE.json configs that round 1 left without a GPU parity test, plus the
cases the round-1 review asked for:

  configs[2]  640x480 RGB-D, LightIntegrator (mask + depth + shaded colour) + Tracer at
              full size into Volume(65024, 8192) @ 5 mm — every voxel byte and every
              raycast image against the oracle (light_integrator.cu:270-354)
  configs[3]  PyramidTracker<DepthTracker>::Track at 320x240 / 640x480 / 1280x960 bases
              against the oracle's pyramid loop (pyramid_tracker.cpp:52-90)
  a colour image whose size differs from the depth image (color_integrator.cu:183-184)
  an empty normal system (all-hole frame) must leave the pose finite (tracker.cpp:153-162)
  two Tracers sharing one Volume
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import scenes
from test_gpu_parity import api, assert_volume_equal, frames, make_pair, sync  # noqa: F401
from vulcan_amd import vk_types as T

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ configs[2] --

@pytest.mark.parametrize("streams", ["requests ahead", "two streams", "one stream"])
def test_rgbd_bench_sequence_matches_oracle(api, orc, streams, monkeypatch):
    """The exact step `bench.py --workload rgbd` times — bench.FrameLoop.step itself, i.e. ONE
    vk_volume_set_view_rounds(.., 3) that also computes the input frame's normals and prepares the
    light integrator's records (vk_light_prep.normals_out), vk_integrate_ahead with those records and
    the raycast bounds riding along, vk_trace_ahead — at 640x480, 5 mm, Volume(65024, 8192), light
    (2, (.025,.08,0)) as in apps/vulcan/vulcan.cu:87-88, six frames from the empty volume. Oracle side:
    Frame::ComputeNormals, three SetView calls, the frame mask, depth and shaded-colour passes, Trace
    (vulcan.cu:297,316-325). Every image of every frame and, at the end, every voxel byte.

    "requests ahead" is bench.py's default: the request pass of frame i + 1 (with its normals and the light
    preparation) rides behind the raycast of frame i in one launch (vk_trace_ahead_requests), and SetView(i + 1)
    launches only its handle + visibility pass (vk_volume_set_view_rounds_ahead). "two streams"
    (VK_BENCH_SPLIT_STREAMS=1): the request pass of frame i on a stream of its own, beside the raycast of frame
    i - 1 (vk_volume_set_view_rounds_split). Both are run twice: frame by frame with every image compared, and with
    all six frames enqueued back to back — no host synchronisation in between, so that the request pass of a frame
    really runs while the raycast before it does — and the end state compared."""
    sys.path.insert(0, ROOT)
    import bench
    count = 6
    k = T.Projection.make(*scenes.APP_INTRINSICS)
    depth = bench.sphere_room_depth(k)
    color = scenes.checker_color(bench.W, bench.H, 0.1, 0.9)
    light = T.Light.make(*bench.LIGHT)
    poses = [scenes.orbit_pose(i, bench.YAW_STEP) for i in range(count)]
    assert bench.SET_VIEW_ROUNDS == 3 and bench.NORMALS_IN_SET_VIEW
    monkeypatch.setattr(bench, "SPLIT_STREAMS", streams == "two streams")
    monkeypatch.setattr(bench, "REQUESTS_AHEAD", streams == "requests ahead")
    ahead = streams == "requests ahead"

    orc.set_threads(16)
    hv = orc.HostVolume(bench.MAIN, bench.EXCESS, voxel_length=bench.VOXEL, truncation_length=bench.TRUNC)
    hf = orc.HostFrame(depth, k, T.Transform.identity(), color=color)
    want = []
    for i in range(count):
        hf.depth_to_world = poses[i]
        hf.compute_normals()                                  # vulcan.cu:297
        for _ in range(3):                                    # vulcan.cu:316-318
            hv.set_view(hf, orc.POLICY_MAXKEY)
        orc.integrate_depth(hv, hf)
        mask = orc.light_frame_mask(hf, 0.2)
        orc.integrate_light_color(hv, hf, light, mask)
        odepth, ocolor, onormals, obounds = orc.trace(hv, hf)
        want.append(dict(normals_in=hf.normals.copy(), mask=mask, depth=odepth, color=ocolor, normals=onormals, bounds=obounds,
                         visible=hv.visible_count, table=hv.hash_entries.copy()))
    orc.set_threads(1)

    def compare_images(loop, w, w_in=None):
        # (requests ahead: the input normals and the mask in memory are already the NEXT frame's, w_in)
        w_in = w if w_in is None else w_in
        tracer = loop.vols[0]["tracer"]
        assert tracer.view_bounds.valid == 1                  # the bounds came with the integrate launch
        assert np.array_equal(loop.frame.normals.cpu().numpy(), w_in["normals_in"], equal_nan=True)
        assert np.array_equal(loop.mask.cpu().numpy(), w_in["mask"])
        assert np.array_equal(tracer.bounds.cpu().numpy(), w["bounds"])
        assert np.array_equal(loop.key.depth.cpu().numpy(), w["depth"])
        assert np.array_equal(loop.key.color.cpu().numpy(), w["color"])
        assert np.array_equal(loop.key.normals.cpu().numpy(), w["normals"], equal_nan=True)

    # ---- frame by frame
    loop = bench.FrameLoop("rgbd", poses)
    assert (loop.split is not None) == (streams == "two streams")
    assert (loop.ahead is not None) == ahead
    dv = loop.vols[0]["vol"]
    rounds_before = 0
    for i in range(count):
        if not ahead or i == 0:
            loop.frame.normals.fill_(-7.0)                    # whatever the step leaves here, it computed itself
            loop.mask.fill_(-7.0)
        sync()
        loop.step(i)
        sync()
        if ahead:
            # the record names the next frame (none behind the last one), and SetView used the previous one
            assert loop.ahead.valid == (1 if i + 1 < count else 0)
            if i + 1 < count:
                assert loop.ahead.content_id == loop.fdesc.content_id + 2 and loop.prep.valid == 1
        ctr = dv.read_counters()
        rounds_before, rounds = int(ctr[T.VK_CTR_ROUNDS]), int(ctr[T.VK_CTR_ROUNDS]) - rounds_before
        assert 1 <= rounds <= 3
        if i == 0:
            assert rounds > 1                                 # 7 k blocks at once: some lose their bucket
        assert dv.visible_count == want[i]["visible"] > 5000
        assert np.array_equal(dv.host_entries(), want[i]["table"])
        compare_images(loop, want[i], want[min(i + 1, count - 1)] if ahead else None)
    assert_volume_equal(dv, hv)
    got = dv.host_voxels()
    assert (got["color_weight"] > 0).sum() > 500000          # the colour pass really ran

    # ---- all frames enqueued back to back (what the bench's timed region does)
    if streams != "one stream":
        del loop, dv
        loop = bench.FrameLoop("rgbd", poses)
        for i in range(count):
            loop.step(i)
        sync()
        compare_images(loop, want[-1])
        assert_volume_equal(loop.vols[0]["vol"], hv)


def test_requests_ahead_record_for_another_frame_is_refused(api):
    """vk_volume_set_view_rounds_ahead with a VALID record made for another pose returns VK_ERR_ARGUMENT and launches nothing
    (the announced frame's requests are in the volume); with the announced frame it goes through, and a record that is
    not valid means the whole call."""
    sys.path.insert(0, ROOT)
    import bench
    poses = [scenes.orbit_pose(i, bench.YAW_STEP) for i in range(4)]
    loop = bench.FrameLoop("depth", poses)
    assert loop.ahead is not None
    loop.step(0)
    sync()
    assert loop.ahead.valid == 1
    vv, lib = loop.vols[0], loop.lib
    before = vv["vol"].read_counters().copy()
    announced = T.Transform.from_buffer_copy(bytes(loop.ndesc.depth_to_world))
    loop.fdesc.depth_to_world = poses[2]                     # not the frame the pass was made for
    loop.fdesc.content_id += 2
    rc = lib.vk_volume_set_view_rounds_ahead(vv["vref"], loop.fref, None, 3, loop.aref, loop.stream)
    sync()
    assert rc == -1 and loop.ahead.valid == 1        # VK_ERR_ARGUMENT
    assert np.array_equal(vv["vol"].read_counters(), before)
    loop.fdesc.depth_to_world = announced
    rc = lib.vk_volume_set_view_rounds_ahead(vv["vref"], loop.fref, None, 3, loop.aref, loop.stream)
    sync()
    assert rc == 0 and loop.ahead.valid == 0
    after = vv["vol"].read_counters()
    assert after[T.VK_CTR_ROUNDS] > before[T.VK_CTR_ROUNDS]
    # an invalid record: the whole call
    rc = lib.vk_volume_set_view_rounds_ahead(vv["vref"], loop.fref, None, 3, loop.aref, loop.stream)
    sync()
    assert rc == 0


@pytest.mark.parametrize("size", [(1280, 960), (96, 1040), (200, 150)])
def test_requests_ahead_at_other_image_sizes(api, size):
    """Tracer.trace(frame, next_frame=) + set_view against trace() + set_view on a second, identical volume: the same images and
    the same volume, bit for bit — at 1280x960 (120 rows of tiles for the normals' counters), at a height beyond the 1024 rows the
    counters cover (the normals are then a launch of their own again) and at a size that is no multiple of the tiles."""
    import torch
    w, h = size
    s = w / 640.0
    k = T.Projection.make(547.0 * s, 547.0 * s, 0.5 * w, 0.5 * h)
    depth = curved_depth(w, h)
    color = scenes.checker_color(w, h, 0.1, 0.9)
    poses = [scenes.yaw(1.0 * i) for i in range(3)]
    results = []
    for ahead in (False, True):
        vol = api.Volume(16384, 4096, voxel_length=0.01, truncation_length=0.05)
        integ, tracer = api.LightIntegrator(vol), api.Tracer(vol)
        integ.light = T.Light.make(2.0, (0.025, 0.08, 0.0))
        fs = [api.Frame(depth, k, p, color=color) for p in poses]
        outs = []
        for i, f in enumerate(fs):
            if not (ahead and i > 0):
                f.compute_normals()
            vol.set_view(f, rounds=3)
            integ.integrate(f)
            out = api.Frame(torch.zeros((h, w), dtype=torch.float32, device="cuda"), k, poses[i])
            if ahead and i + 1 < len(fs):
                tracer.trace(out, next_frame=fs[i + 1], next_needs_normals=True)
                assert vol.requests_ahead.valid == 1
            else:
                tracer.trace(out)
            sync()
            outs.append((out.depth.cpu().numpy(), out.color.cpu().numpy(), out.normals.cpu().numpy(), f.normals.cpu().numpy()))
        results.append((outs, vol.host_voxels().tobytes(), vol.host_entries().tobytes(), vol.visible_count))
    (plain, pv, pe, pc), (made, mv, me, mc) = results
    assert pc == mc > 100 and pv == mv and pe == me
    for a, b in zip(plain, made):
        for x, y in zip(a, b):
            assert np.array_equal(x, y, equal_nan=True)


# ------------------------------------------------------------------ configs[3] --

def curved_depth(w, h):
    y, x = np.mgrid[0:h, 0:w]
    return (1.0 + 0.05 * np.cos(3.0 * x / w) * np.sin(2.0 * y / h)).astype(np.float32)


@pytest.mark.parametrize("size", [(320, 240), (640, 480), (1280, 960)])
def test_pyramid_tracker_two_views_matches_oracle(api, orc, size):
    """The second case bench.py's pyramid-icp leg times (configs[3], round 6; bench.pyramid_case "two views"): the room scene
    rendered from two poses 33 mm / 2.6 deg apart, the frame started at the key frame's pose — different images, so both
    levels run several Gauss-Newton steps. Same bar as the same-surface case: the oracle's pose within 2e-5 per matrix
    entry, run-to-run identical bits; and the true pose is found to a tenth of a millimetre."""
    sys.path.insert(0, ROOT)
    import bench
    w, h = size
    k, key_depth, key_pose, frame_depth, start, truth = bench.pyramid_case("two views", w, h)
    hk, dk = frames(api, orc, key_depth, k, key_pose)
    hf, df = frames(api, orc, frame_depth, k, start)
    orc.set_threads(8)
    for f in (hk, hf):
        f.compute_normals()
    for f in (dk, df):
        f.compute_normals()
    want, iters = orc.pyramid_track(hk, hf)
    orc.set_threads(1)
    tracker = api.PyramidTracker()
    tracker.keyframe = dk
    got = tracker.track(df)
    sync()
    assert int(tracker.tracker.state.cpu()[1]) == 1 and int(tracker.tracker.state.cpu()[0]) >= 2
    np.testing.assert_allclose(got.matrix(), want.matrix(), atol=2e-5)
    np.testing.assert_allclose(got.inverse_matrix(), want.inverse_matrix(), atol=2e-5)
    t_err, r_err = bench.pose_error(got, truth)
    assert t_err < 1.5e-4 and r_err < 0.05, (t_err, r_err)
    df.depth_to_world = start
    assert bytes(tracker.track(df)) == bytes(got)


@pytest.mark.parametrize("size", [(320, 240), (640, 480), (1280, 960)])
def test_pyramid_tracker_matches_oracle(api, orc, size):
    """PyramidTracker<DepthTracker>::Track: downsampled levels bit-exact, final pose within
    2e-5 per matrix entry of the oracle's pyramid loop (float tree sums vs float64 sums are
    the only difference between the two), and the pose is recovered."""
    w, h = size
    s = w / 640.0
    k = T.Projection.make(547.0 * s, 547.0 * s, 320.0 * s, 240.0 * s)     # depth_tracker_test.cu:20-23 scaled
    key_depth = curved_depth(w, h)
    hk, dk = frames(api, orc, key_depth, k, T.Transform.identity(), color=scenes.checker_color(w, h, 0.1, 0.9))
    hk.compute_normals()
    dk.compute_normals()
    start = T.Transform.translate(0.002, -0.001, 0.003) * T.Transform.rotate(0.999995, 0.002, -0.0015, 0.001)
    hf, df = frames(api, orc, key_depth, k, start, color=scenes.checker_color(w, h, 0.1, 0.9))
    hf.compute_normals()
    df.compute_normals()
    sync()
    assert np.array_equal(dk.normals.cpu().numpy(), hk.normals, equal_nan=True)

    # Frame::Downsample, both pyramid levels (frame.cpp:38-58)
    hh, dh = hf.downsample(), df.downsample()
    hq, dq = hh.downsample(), dh.downsample()
    sync()
    for host, dev in ((hh, dh), (hq, dq)):
        assert (dev.width, dev.height) == (host.width, host.height)
        assert np.array_equal(dev.depth.cpu().numpy(), host.depth)
        assert np.array_equal(dev.normals.cpu().numpy(), host.normals, equal_nan=True)
        assert np.array_equal(dev.color.cpu().numpy(), host.color)
        assert bytes(dev.depth_projection) == bytes(host.depth_projection)

    want, iters = orc.pyramid_track(hk, hf)
    tracker = api.PyramidTracker()
    tracker.keyframe = dk
    got = tracker.track(df)
    sync()
    np.testing.assert_allclose(got.matrix(), want.matrix(), atol=2e-5)
    np.testing.assert_allclose(got.inverse_matrix(), want.inverse_matrix(), atol=2e-5)
    np.testing.assert_allclose(got.matrix(), np.eye(4), atol=5e-4)          # the keyframe's pose
    np.testing.assert_allclose(got.matrix() @ got.inverse_matrix(), np.eye(4), atol=1e-5)
    # run-to-run identical (fixed-order reduction)
    df.depth_to_world = start
    again = tracker.track(df)
    assert bytes(again) == bytes(got)
    # frame.compute_normals() + track() as one call (vk_icp_pyramid_track_frame: the normal image and the start pose travel with
    # the launch that builds the pyramid): the same normal image, the same half-resolution level, the same pose, bit for bit
    want_normals = df.normals.cpu().numpy().copy()
    half_before = tracker._pyramid.cpu().numpy().copy()
    df.normals.fill_(-3.0)
    tracker._pyramid.fill_(-5.0)
    df.depth_to_world = start
    fused = tracker.track(df, compute_normals=True)
    sync()
    assert bytes(fused) == bytes(got)
    assert np.array_equal(df.normals.cpu().numpy(), want_normals, equal_nan=True)
    assert np.array_equal(tracker._pyramid.cpu().numpy(), half_before, equal_nan=True)


# ------------------------------------------------- colour image of another size --

def test_color_image_of_a_different_size(api, orc):
    """frame.h:21-25: three independent images. ColorIntegrator::IntegrateColor tests and
    strides with the COLOUR image's size (color_integrator.cu:183-184)."""
    w, h = 160, 120
    kd = T.Projection.make(136, 136, 80, 60)
    depth = scenes.plane(w, h, 1.5)
    for cw, ch in ((320, 240), (96, 72)):
        sx = cw / w
        kc = T.Projection.make(136 * sx, 136 * sx, 80 * sx + 1.5, 60 * sx - 0.5)
        y, x = np.mgrid[0:ch, 0:cw]
        color = np.stack([0.1 + 0.8 * x / cw, 0.1 + 0.8 * y / ch, 0.5 + 0 * x], -1).astype(np.float32)
        tcd = T.Transform.translate(0.01, -0.005, 0.0)
        hf = orc.HostFrame(depth, kd, T.Transform.identity(), color=color, color_projection=kc, depth_to_color=tcd)
        df = api.Frame(depth, kd, T.Transform.identity(), color=color, color_projection=kc, depth_to_color=tcd)
        assert (df.desc().color_width, df.desc().color_height) == (cw, ch)
        for fused in (True, False):
            hv, dv = make_pair(api, orc, 4096, 1024, 0.008, 0.04)
            for _ in range(4):
                hv.set_view(hf, orc.POLICY_MAXKEY)
                dv.set_view(df)
            orc.integrate_depth(hv, hf)
            orc.integrate_color(hv, hf)
            integ = api.ColorIntegrator(dv)
            if fused:
                integ.integrate(df)
            else:
                integ.integrate_depth(df)
                integ.integrate_color(df)
            assert_volume_equal(dv, hv)
            assert (hv.voxels["color_weight"] > 0).sum() > 10000
        # the light path indexes colour with the depth size upstream (light_integrator.cu:333-334):
        # a mismatching frame is refused instead of read out of bounds
        df.compute_normals()
        mask = api.LightIntegrator(dv)
        with pytest.raises(api.VkError):
            mask.compute_frame_mask(df)
        lib = api.lib()
        import torch
        m = torch.zeros((h, w), dtype=torch.float32, device="cuda")
        rc = lib.vk_integrate_depth_light(api._ref(dv.desc()), api._ref(mask.params), api._ref(mask.light),
                                          api._ptr(m), api._ref(df.desc()), api.stream())
        assert rc == -1                                        # VK_ERR_ARGUMENT


# ----------------------------------------------------------- empty normal system --

@pytest.mark.parametrize("translation", [True, False])
def test_empty_system_leaves_pose_finite(api, orc, translation):
    """No valid correspondence (an all-hole frame): H = 0, g = 0. Eigen's LDLT solves that to
    x = 0, |x| < 1e-6 ends the loop and the pose stays what it was (tracker.cpp:153-162); an
    unguarded LDL^T returns 0/0 = NaN and poisons the pose."""
    w, h = 160, 120
    k = T.Projection.make(136, 136, 80, 60)
    hk, dk = frames(api, orc, curved_depth(w, h), k, T.Transform.identity())
    hk.compute_normals()
    dk.compute_normals()
    start = T.Transform.translate(0.01, 0.02, -0.01) * T.Transform.rotate(0.9998, 0.01, -0.01, 0.012)
    hf, df = frames(api, orc, np.zeros((h, w), np.float32), k, start)
    hf.compute_normals()
    df.compute_normals()
    tracker = api.DepthTracker()
    tracker.keyframe = dk
    tracker.translation_enabled = translation
    got = tracker.track(df)
    sync()
    state = tracker.state.cpu().numpy()
    assert np.all(np.isfinite(got.matrix())) and np.all(np.isfinite(got.inverse_matrix()))
    assert state[1] == 1 and state[0] == 1                      # stopped after the first (zero) step
    assert np.all(tracker.update.cpu().numpy() == 0)
    want, iters = orc.icp_track(hk, hf, 20, translation)
    assert iters == 1
    assert bytes(got) == bytes(want)                            # both re-orthonormalise the same pose
    np.testing.assert_allclose(got.matrix(), start.matrix(), atol=1e-6)
    # the pyramid on the same input
    pyr = api.PyramidTracker()
    pyr.keyframe = dk
    df.depth_to_world = start
    got = pyr.track(df)
    assert np.all(np.isfinite(got.matrix()))
    np.testing.assert_allclose(got.matrix(), start.matrix(), atol=1e-6)


# ------------------------------------------------------- two tracers, one volume --

def test_two_tracers_share_a_volume(api, orc):
    """Every attached Tracer's cached bounds go stale when the visible list changes, not
    only the most recently attached one's."""
    import torch
    w, h = 160, 120
    k = T.Projection.make(136, 136, 80, 60)
    pose = scenes.tracer_test_pose()
    hf, df = frames(api, orc, scenes.plane(w, h, 1.5), k, pose)
    hv, dv = make_pair(api, orc, 4096, 1024, 0.008, 0.04)
    first, second = api.Tracer(dv), api.Tracer(dv)
    integ = api.DepthIntegrator(dv)
    out_a = api.Frame(torch.zeros((h, w), dtype=torch.float32, device="cuda"), k, pose)
    out_b = api.Frame(torch.zeros((h, w), dtype=torch.float32, device="cuda"), k, pose)
    for round_ in range(4):          # same pose every round; the visible list keeps growing
        hv.set_view(hf, orc.POLICY_MAXKEY)
        orc.integrate_depth(hv, hf)
        dv.set_view(df)
        assert first.view_bounds.valid == 0 and second.view_bounds.valid == 0
        integ.integrate(df)
        first.trace(out_a)
        second.trace(out_b)
        sync()
        odepth, ocolor, onormals, obounds = orc.trace(hv, hf)
        for tracer, out in ((first, out_a), (second, out_b)):
            assert np.array_equal(tracer.bounds.cpu().numpy(), obounds), round_
            assert np.array_equal(out.depth.cpu().numpy(), odepth), round_


# ------------------------------------------------- the rig step through libvk_comm --

def test_rig_step_through_vk_comm(api, orc):
    """configs[4] on the one GPU a test box has: the Gauss-Newton loop with the shipped C
    hook (vk_comm_reduce_hook) between the system and the solve. With one rank the sum is the
    identity, so the pose must equal the hook-less loop bit for bit — through the loopback
    communicator and through a real one-rank RCCL communicator (ncclCommInitRank, ncclAllReduce
    on the compute stream)."""
    import torch
    from vulcan_amd import comm
    w, h = 320, 240
    k = T.Projection.make(273.5, 273.5, 160, 120)
    key_depth = curved_depth(w, h)
    hk, dk = frames(api, orc, key_depth, k, T.Transform.identity())
    dk.compute_normals()
    start = T.Transform.translate(0.002, -0.001, 0.003) * T.Transform.rotate(0.999995, 0.002, -0.0015, 0.001)
    hf, df = frames(api, orc, key_depth, k, start)
    df.compute_normals()
    tracker = api.DepthTracker()
    tracker.keyframe = dk
    plain = tracker.track(df)
    steps_plain = int(tracker.state.cpu()[0])
    for make in (lambda: comm.Communicator(None, 0, 1), lambda: comm.Communicator(comm.unique_id(), 0, 1)):
        c = make()
        buf = torch.arange(48, dtype=torch.float32, device="cuda")
        c.allreduce_system(buf)
        sync()
        assert torch.equal(buf.cpu(), torch.arange(48, dtype=torch.float32))
        df.depth_to_world = start
        got = c.track(tracker, df)
        sync()
        assert bytes(got) == bytes(plain)
        assert int(tracker.state.cpu()[0]) == steps_plain
        assert c.time_allreduce(20) > 0
        tracker.comm = None
        c.close()
    # and a python hook through the same C loop (what vulcan_amd.dist.allreduce_system plugs into)
    calls = []
    tracker.reduce_hook = lambda system: calls.append(system.data_ptr())
    df.depth_to_world = start
    got = tracker.track(df)
    assert bytes(got) == bytes(plain) and len(calls) >= steps_plain


def test_early_exit_matches_the_full_loop(api, orc):
    """vk_track_poll: stopping the enqueue once the loop has converged gives the pose of the
    full-length loop (steps after convergence are no-ops), and stops within one chunk."""
    w, h = 320, 240
    k = T.Projection.make(273.5, 273.5, 160, 120)
    key_depth = curved_depth(w, h)
    hk, dk = frames(api, orc, key_depth, k, T.Transform.identity())
    dk.compute_normals()
    start = T.Transform.translate(0.002, -0.001, 0.003) * T.Transform.rotate(0.999995, 0.002, -0.0015, 0.001)
    hf, df = frames(api, orc, key_depth, k, start)
    df.compute_normals()
    poses, steps = [], []
    for chunk in (0, 1, 3, 4, 50):
        tracker = api.DepthTracker()
        tracker.poll_chunk = chunk
        tracker.keyframe = dk
        df.depth_to_world = start
        poses.append(bytes(tracker.track(df)))
        sync()
        st = tracker.state.cpu().numpy()
        steps.append(int(st[0]))
        assert st[1] == 1
    assert len(set(poses)) == 1 and len(set(steps)) == 1


@pytest.mark.parametrize("w,h", [(640, 480), (1280, 960)])
def test_loop_kernel_is_reproducible_and_continues_past_one_launch(api, orc, w, h):
    """The depth tracker's loop is one launch whose workgroups exchange their sums inside it
    (vk_icp_track without a reduce hook). Twice the same call gives the same bits; a loop longer
    than the 1023 steps one launch can tag continues in a further launch and ends where the
    converged loop ended; with the rig's hook (launch per stage, same order of summation) the
    pose is the same bit for bit — also at 1280x960, where the image has more pixel groups than
    the device holds workgroups and a workgroup publishes several groups."""
    k = T.Projection.make(547.0 * w / 640, 547.0 * w / 640, w / 2, h / 2)
    key_depth = curved_depth(w, h)
    hk, dk = frames(api, orc, key_depth, k, T.Transform.identity())
    dk.compute_normals()
    start = T.Transform.translate(0.002, -0.001, 0.003) * T.Transform.rotate(0.999995, 0.002, -0.0015, 0.001)
    hf, df = frames(api, orc, key_depth, k, start)
    df.compute_normals()
    results = []
    for iterations, hook in ((20, False), (20, False), (1100, False), (20, True)):
        tracker = api.DepthTracker()
        tracker.keyframe = dk
        tracker.max_iterations = iterations
        if hook:
            tracker.reduce_hook = lambda system: None
        df.depth_to_world = start
        pose = bytes(tracker.track(df))
        sync()
        st = tracker.state.cpu().numpy()
        assert st[1] == 1 and 0 < st[0] < 20
        results.append((pose, int(st[0]), tracker.system.cpu().numpy().tobytes()))
    assert len(set(results)) == 1


def test_track_wait_hands_over_the_device_pose(api, orc):
    """vk_track_wait (Tracker::EndSolve): the pose a Track leaves in pinned host memory is the pose
    on the device, for the one-launch loop, the rig's launch-per-stage loop and the colour tracker;
    waiting without a Track that could have left one reports it instead of blocking."""
    import color_scenes as cs
    w, h = 320, 240
    k = T.Projection.make(273.5, 273.5, 160, 120)
    key_depth = curved_depth(w, h)
    hk, dk = frames(api, orc, key_depth, k, T.Transform.identity())
    dk.compute_normals()
    start = T.Transform.translate(0.002, -0.001, 0.003) * T.Transform.rotate(0.999995, 0.002, -0.0015, 0.001)
    hf, df = frames(api, orc, key_depth, k, start)
    df.compute_normals()
    for hook in (False, True):
        tracker = api.DepthTracker()
        tracker.keyframe = dk
        if hook:
            tracker.reduce_hook = lambda system: None
        df.depth_to_world = start
        got = tracker.track(df)
        sync()
        assert bytes(got) == tracker.pose.cpu().numpy().tobytes()
        assert bytes(got) != bytes(start)
    # a poll block no Track has used: the stream drains, nothing arrives
    fresh = api.DepthTracker()
    fresh._poll()
    assert api.lib().vk_track_wait(C.byref(fresh._poll_desc), api.stream()) == -2      # VK_ERR_UNSUPPORTED

    kc = cs.keyframe_images()[1][:h, :w].copy()
    key = api.Frame(key_depth, k, T.Transform.identity(), color=kc, normals=dk.normals)
    moved = api.Frame(key_depth, k, start, color=kc, normals=dk.normals)
    ct = api.ColorTracker()
    ct.keyframe = key
    ct.max_iterations = 3
    got = ct.track(moved)
    sync()
    assert bytes(got) == ct.pose.cpu().numpy().tobytes()[:128]


def test_set_view_prepares_the_light_integrator(api, orc):
    """vk_volume_set_view_prepare: the request pass of SetView also leaves LightIntegrator's frame
    mask and per-pixel records — the same bits as vk_light_prepare — and the volume ends up exactly
    as with the separate pass; a frame without normals, a second integrate of the same frame and a
    larger frame fall back to the pass of their own."""
    import torch
    w, h = 200, 150                       # not a multiple of the request pass's 64x4 patches
    k = T.Projection.make(170.0, 170.0, 100.3, 74.6)
    rng = np.random.default_rng(5)
    depth = (1.2 + 0.1 * np.sin(np.arange(w)[None, :] / 9.0) * np.cos(np.arange(h)[:, None] / 7.0)).astype(np.float32)
    depth[rng.random((h, w)) < 0.02] = 0.0                       # holes: the 7x7 window sees them
    depth[40:60, 80:100] += 0.5                                  # a step edge: mask 0 around it
    color = rng.random((h, w, 3), dtype=np.float32)              # some pixels outside [0.02, 0.98]
    pose = T.Transform.translate(0.01, -0.02, 0.03)

    def run(fused):
        vol = api.Volume(4096, 1024, voxel_length=0.008, truncation_length=0.04)
        integ = api.LightIntegrator(vol)
        integ.light = T.Light.make(1.3, (0.02, -0.01, 0.0))
        frame = api.Frame(depth, k, pose, color=color)
        frame.compute_normals()
        for i in range(3):
            frame.depth_to_world = T.Transform.translate(0.01 * i, -0.02, 0.03)
            if not fused:
                vol.light_prep = None                            # plain vk_volume_set_view every time
            vol.set_view(frame)
            if fused and i > 0:
                assert integ._prep.valid == 1                    # prepared by set_view
            integ.integrate(frame)
            if fused:
                assert integ._prep.valid == 0                    # used once
        sync()
        return (vol.voxels.cpu().numpy().tobytes(), vol.hash_entries.cpu().numpy().tobytes(),
                integ.frame_mask.cpu().numpy().tobytes(), integ.pixel_records.cpu().numpy().tobytes())

    assert run(True) == run(False)

    # fallbacks: no normals -> no ride; a frame larger than the registered buffers -> no ride
    vol = api.Volume(4096, 1024, voxel_length=0.008, truncation_length=0.04)
    integ = api.LightIntegrator(vol)
    frame = api.Frame(depth, k, pose, color=color)
    frame.compute_normals()
    vol.set_view(frame)
    integ.integrate(frame)                                       # registers w x h buffers
    bare = api.Frame(depth, k, pose, color=color)                # no normal image
    vol.set_view(bare)
    assert integ._prep.valid == 0
    big = api.Frame(np.full((h + 8, w), 1.5, dtype=np.float32), k, pose, color=np.full((h + 8, w, 3), 0.5, dtype=np.float32))
    big.compute_normals()
    vol.set_view(big)
    assert integ._prep.valid == 0
    sync()


def test_fewer_workgroups_than_pixel_groups_gives_the_same_bits(api, orc):
    """The one-launch loops publish one slot per pixel GROUP: when the device holds fewer workgroups
    than the image has groups (fewer CUs, lower occupancy — forced here with vk_test_hooks.loop_grid_cap), a
    workgroup takes several groups and the poses come out bit for bit the same."""
    import color_scenes as cs
    w, h = 640, 480
    k = T.Projection.make(547.0, 547.0, 320, 240)
    key_depth = curved_depth(w, h)
    hk, dk = frames(api, orc, key_depth, k, T.Transform.identity())
    dk.compute_normals()
    start = T.Transform.translate(0.002, -0.001, 0.003) * T.Transform.rotate(0.999995, 0.002, -0.0015, 0.001)
    hf, df = frames(api, orc, key_depth, k, start)
    df.compute_normals()
    kc = cs.keyframe_images()[1]
    key = api.Frame(key_depth, k, T.Transform.identity(), color=kc, normals=dk.normals)
    moved = api.Frame(key_depth, k, start, color=kc, normals=dk.normals)

    def run():
        out = []
        tracker = api.PyramidTracker()
        tracker.keyframe = dk
        df.depth_to_world = start
        out.append(bytes(tracker.track(df)))
        ct = api.ColorTracker()
        ct.keyframe = key
        ct.max_iterations = 4
        moved.depth_to_world = start
        out.append(bytes(ct.track(moved)))
        sync()
        assert int(tracker.tracker.state.cpu()[1]) in (0, 1) and int(ct.state.cpu()[1]) in (0, 1)
        return out

    full = run()
    for cap in (100, 7, 1):
        with api.test_hooks(loop_grid_cap=cap):
            assert run() == full, cap


def test_two_streams_track_concurrently(api, orc):
    """Two host threads, each with a stream and a tracker of its own, track at the same time on
    one device. A loop kernel needs all of its workgroups resident and one fills the device, so
    two that start together could starve each other; the library chains loop kernels of different
    streams (vk_loop_launch_begin). Every Track must finish un-aborted with the single-stream bits."""
    import threading
    import torch
    w, h = 640, 480
    k = T.Projection.make(547.0, 547.0, 320, 240)
    key_depth = curved_depth(w, h)
    hk, dk = frames(api, orc, key_depth, k, T.Transform.identity())
    dk.compute_normals()
    start = T.Transform.translate(0.002, -0.001, 0.003) * T.Transform.rotate(0.999995, 0.002, -0.0015, 0.001)
    _, df0 = frames(api, orc, key_depth, k, start)
    df0.compute_normals()
    tracker = api.PyramidTracker()
    tracker.keyframe = dk
    want = bytes(tracker.track(df0))
    sync()

    results, errors = {}, []

    def work(name):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                _, df = frames(api, orc, key_depth, k, start)
                df.normals = df0.normals.clone()
                t = api.PyramidTracker()
                t.keyframe = dk
                out = []
                for _ in range(40):
                    df.depth_to_world = start
                    out.append(bytes(t.track(df)))
                    assert int(t.tracker.state.cpu()[1]) in (0, 1)
                stream.synchronize()
                results[name] = out
        except Exception as e:            # noqa: BLE001 — reported by the main thread
            errors.append((name, repr(e)))

    threads = [threading.Thread(target=work, args=(n,)) for n in ("a", "b")]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for name in ("a", "b"):
        assert all(p == want for p in results[name]), name


def test_stale_light_preparation_is_never_used(api, orc):
    """Upstream's SetView reads the depth image only, so SetView(frame); <change normals or colours>;
    Integrate(frame) is legal. The preparation SetView made ahead is tied to the frame's content
    (vk_frame.content_id), not to its pointers: after an in-place change Integrate prepares again."""
    import torch
    w, h = 160, 120
    k = T.Projection.make(136, 136, 80, 60)
    depth = scenes.sphere(4 * w, 4 * h)[::4, ::4].copy()
    color = scenes.checker_color(w, h, 0.1, 0.9)
    light = T.Light.make(2.0, (0.025, 0.08, 0.0))
    hf = orc.HostFrame(depth, k, T.Transform.identity(), color=color)
    df = api.Frame(depth, k, T.Transform.identity(), color=color)
    hf.compute_normals()
    df.compute_normals()
    hv, dv = make_pair(api, orc, 8192, 2048, 0.008, 0.04)
    integ = api.LightIntegrator(dv)
    integ.light = light

    def oracle_step():
        hv.set_view(hf, orc.POLICY_MAXKEY)
        orc.integrate_depth(hv, hf)
        orc.integrate_light_color(hv, hf, light, orc.light_frame_mask(hf, 0.2))

    oracle_step()
    dv.set_view(df)
    integ.integrate(df)
    assert_volume_equal(dv, hv)

    # (1) an in-place torch write between SetView and Integrate (torch's version counter sees it)
    dv.set_view(df)
    assert integ._prep.valid == 1
    df.normals.mul_(-1.0)
    df.color.mul_(0.5).add_(0.2)
    hf.normals, hf.color = -hf.normals, (hf.color * np.float32(0.5) + np.float32(0.2)).astype(np.float32)
    assert not api.lib().vk_light_prepared(api._ref(integ._prep), api._ref(df.desc()), integ.depth_threshold)
    oracle_step()
    integ.integrate(df)
    assert_volume_equal(dv, hv)

    # (2) a write through the raw pointer (a kernel of the caller's own), announced with touch()
    dv.set_view(df)
    raw = torch.full_like(df.normals, 0.0)
    raw[..., 2] = -1.0
    api.check(api.lib().vk_memcpy_d2d(api._ptr(df.normals), api._ptr(raw), raw.numel() * 4, api.stream()), "copy")
    df.touch()
    hf.normals = raw.cpu().numpy()
    oracle_step()
    integ.integrate(df)
    assert_volume_equal(dv, hv)

    # (3) nothing changed: the preparation made by SetView is used
    hv.set_view(hf, orc.POLICY_MAXKEY)
    dv.set_view(df)
    assert api.lib().vk_light_prepared(api._ref(integ._prep), api._ref(df.desc()), integ.depth_threshold)

    # a frame that does not say what it holds (content_id 0) is never prepared ahead
    desc = df.desc()
    desc.content_id = 0
    api.check(api.lib().vk_volume_set_view_prepare(api._ref(dv.desc()), api._ref(desc), api._ref(integ._prep), api.stream()),
              "vk_volume_set_view_prepare")
    assert integ._prep.valid == 0


def test_configs0_dense_128_on_the_device(api, orc):
    """BASELINE configs[0] (SURVEY 8d Config 1): one 640x480 depth frame into a dense 128^3 voxel
    region = 16^3 = 4096 hand-placed blocks of 8^3 straddling the surface (the layout bench.py's
    cpu_baseline times on the host; tests/integrator_test.cu:141-199 is the arithmetic). The device
    integrates the same hand-placed volume: every voxel byte equals the oracle's."""
    import bench
    k = T.Projection.make(*scenes.APP_INTRINSICS)
    depth = bench.sphere_room_depth(k)
    hv = orc.HostVolume(8192, 1024, voxel_length=bench.VOXEL, truncation_length=bench.TRUNC)
    origin = np.array([[x, y, z] for z in range(42, 58) for y in range(-8, 8) for x in range(-8, 8)], dtype=np.int16)
    n = len(origin)
    assert n == 4096
    hv.hash_entries["block"]["origin"][:n] = origin
    hv.hash_entries["data"][:n] = np.arange(n)
    hv.hash_entries["next"][:n] = -1
    hv.visible_blocks[:n] = np.arange(n)
    hv.counters[T.VK_CTR_VISIBLE] = n
    dv = api.Volume(8192, 1024, voxel_length=bench.VOXEL, truncation_length=bench.TRUNC)
    dv.upload(hv)
    hf = orc.HostFrame(depth, k, T.Transform.identity())
    df = api.Frame(depth, k, T.Transform.identity())
    integ = api.DepthIntegrator(dv)
    for _ in range(2):
        orc.integrate_depth(hv, hf)
        integ.integrate(df)
        sync()
        assert dv.host_voxels().tobytes() == hv.voxels.tobytes()
    updated = int((hv.voxels["distance_weight"][:n * 512] > 0).sum())
    assert updated > 500000 and hv.voxels["distance_weight"].max() == 2      # the band through 2 097 152 voxels
