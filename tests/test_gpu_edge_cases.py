"""Edge cases on the device vs the oracle: empty and ragged inputs, odd table
sizes, pool exhaustion, invalid depth, two volumes in one process."""
import numpy as np
import pytest

import scenes
from test_gpu_parity import api, assert_volume_equal, frames, make_pair, sync  # noqa: F401  (fixtures/helpers)
from vulcan_amd import vk_types as T

pytestmark = pytest.mark.gpu


def test_empty_frame_is_a_no_op(api, orc):
    """No valid depth: nothing allocated, nothing visible, integrate and trace run on an empty list."""
    import torch
    w, h = 160, 120
    k = T.Projection.make(136, 136, 80, 60)
    depth = np.zeros((h, w), np.float32)
    depth[::7, ::5] = 9.0            # beyond max depth: ignored too
    hf, df = frames(api, orc, depth, k, T.Transform.identity(), color=scenes.constant_color(w, h))
    hv, dv = make_pair(api, orc, 1000, 37, 0.008, 0.04)        # table sizes that are multiples of nothing
    hv.set_view(hf, orc.POLICY_MAXKEY)
    dv.set_view(df)
    assert_volume_equal(dv, hv)
    assert dv.visible_count == 0
    api.ColorIntegrator(dv).integrate(df)
    orc.integrate_depth(hv, hf)
    out = api.Frame(torch.full((h, w), 7.0, dtype=torch.float32, device="cuda"), k, T.Transform.identity())
    api.Tracer(dv).trace(out)
    sync()
    assert_volume_equal(dv, hv)
    assert np.all(out.depth.cpu().numpy() == 0) and np.all(out.color.cpu().numpy() == 0)
    assert np.all(out.normals.cpu().numpy() == 0)


@pytest.mark.parametrize("size", [(101, 77), (66, 50), (17, 9)])
def test_ragged_image_sizes(api, orc, size):
    """Widths/heights that are not multiples of the 8/16/64-wide tiles the kernels use."""
    import torch
    w, h = size
    k = T.Projection.make(0.85 * w, 0.85 * w, 0.49 * w, 0.52 * h)
    y, x = np.mgrid[0:h, 0:w]
    depth = (1.2 + 0.3 * np.sin(x / 9.0) * np.cos(y / 7.0)).astype(np.float32)
    depth[h // 3, w // 4:w // 2] = 0.0
    color = np.stack([0.2 + 0.6 * (x % 7) / 7.0, 0.5 + 0 * x, 0.3 + 0.5 * (y % 5) / 5.0], -1).astype(np.float32)
    pose = scenes.tracer_test_pose()
    hf, df = frames(api, orc, depth, k, pose, color=color)
    hf.compute_normals()
    df.compute_normals()
    sync()
    assert np.array_equal(df.normals.cpu().numpy(), hf.normals, equal_nan=True)
    hv, dv = make_pair(api, orc, 4099, 1021, 0.01, 0.04)
    for _ in range(6):
        hv.set_view(hf, orc.POLICY_MAXKEY)
        dv.set_view(df)
    orc.integrate_depth(hv, hf)
    orc.integrate_color(hv, hf)
    api.ColorIntegrator(dv).integrate(df)
    assert_volume_equal(dv, hv)
    odepth, ocolor, onormals, obounds = orc.trace(hv, hf)
    out = api.Frame(torch.zeros((h, w), dtype=torch.float32, device="cuda"), k, pose)
    tracer = api.Tracer(dv)
    tracer.trace(out)
    sync()
    assert np.array_equal(tracer.bounds.cpu().numpy(), obounds)
    assert np.array_equal(out.depth.cpu().numpy(), odepth)
    assert np.array_equal(out.color.cpu().numpy(), ocolor)
    assert np.array_equal(out.normals.cpu().numpy(), onormals, equal_nan=True)
    # ICP hooks on the same ragged frame
    tracker = api.DepthTracker()
    tracker.keyframe = df
    hk = orc.HostFrame(depth, k, pose, normals=hf.normals)
    moved = T.Transform.translate(0.002, -0.001, 0.001) * pose
    hf2, df2 = frames(api, orc, depth, k, moved, normals=hf.normals)
    assert np.array_equal(tracker.compute_residuals(df2).cpu().numpy(), orc.icp_residuals(hk, hf2))
    assert np.array_equal(tracker.compute_jacobian(df2).cpu().numpy(), orc.icp_jacobian(hk, hf2, True))


def test_pool_and_excess_exhaustion(api, orc):
    """More blocks than the pool holds: the same requests are dropped on both sides
    (volume.cu:356) and the volume stays consistent."""
    w, h = 160, 120
    k = T.Projection.make(136, 136, 80, 60)
    hf, df = frames(api, orc, scenes.plane(w, h, 1.5), k, scenes.tracer_test_pose())
    hv, dv = make_pair(api, orc, 257, 40, 0.008, 0.04)
    for _ in range(5):
        hv.set_view(hf, orc.POLICY_MAXKEY)
        dv.set_view(df)
        assert_volume_equal(dv, hv, voxels=False)
    assert hv.counters[T.VK_CTR_DROPPED] > 100
    entries = dv.host_entries()
    used = entries["data"][entries["data"] >= 0]
    assert len(np.unique(used)) == len(used) <= hv.max
    api.DepthIntegrator(dv).integrate(df)
    orc.integrate_depth(hv, hf)
    assert_volume_equal(dv, hv)


def test_two_volumes_in_one_process(api, orc):
    """The reference keeps its counters in process-wide __device__ symbols
    (volume.cu:17-21), so two Volumes corrupt each other; here they are independent."""
    w, h = 160, 120
    k = T.Projection.make(136, 136, 80, 60)
    hfa, dfa = frames(api, orc, scenes.plane(w, h, 1.5), k, T.Transform.identity())
    hfb, dfb = frames(api, orc, scenes.ramp(w, h), k, scenes.tracer_test_pose())
    hva, dva = make_pair(api, orc, 4096, 1024, 0.008, 0.04)
    hvb, dvb = make_pair(api, orc, 2048, 512, 0.02, 0.08)
    for _ in range(4):                 # interleaved on the same stream
        dva.set_view(dfa)
        dvb.set_view(dfb)
        api.DepthIntegrator(dva).integrate(dfa)
        api.DepthIntegrator(dvb).integrate(dfb)
        hva.set_view(hfa, orc.POLICY_MAXKEY)
        orc.integrate_depth(hva, hfa)
        hvb.set_view(hfb, orc.POLICY_MAXKEY)
        orc.integrate_depth(hvb, hfb)
    assert_volume_equal(dva, hva)
    assert_volume_equal(dvb, hvb)


def test_argument_errors_are_reported_not_thrown(api):
    lib = api.lib()
    import ctypes as C
    v = T.Volume()          # all-null descriptor
    assert lib.vk_volume_set_view(C.byref(v), None, None) == -1
    assert lib.vk_volume_initialize(C.byref(v), None) == -1
    assert lib.vk_trace_reset_bounds(None, 10, None) == -1
    assert lib.vk_image_downsample(641, 480, None, None, 1, None) == -1
    with pytest.raises(api.VkError):
        api.check(-1, "vk_volume_set_view")


def test_bounds_ahead_hits_and_falls_back(api, orc):
    """The raycast bounds prepared inside the integrate launch (vk_view_bounds) are used
    only for the very view they were computed for; every other case recomputes them,
    and all of them give the oracle's images."""
    import torch
    w, h = 320, 240
    k = T.Projection.make(270, 270, 160, 120)
    y, x = np.mgrid[0:h, 0:w]
    depth = (1.3 + 0.2 * np.sin(x / 23.0) * np.cos(y / 17.0)).astype(np.float32)
    pose_a = scenes.tracer_test_pose()
    pose_b = T.Transform.translate(0.03, -0.02, 0.01) * pose_a
    hf, df = frames(api, orc, depth, k, pose_a, color=scenes.checker_color(w, h, 0.2, 0.8))
    hv, dv = make_pair(api, orc, 16384, 2048, 0.008, 0.04)
    integ, tracer = api.ColorIntegrator(dv), api.Tracer(dv)
    vb = tracer.view_bounds
    assert dv.view_bounds is vb and vb.valid == 0

    def check(pose):
        hf.depth_to_world = pose
        want = orc.trace(hv, hf)
        out = api.Frame(torch.zeros((h, w), dtype=torch.float32, device="cuda"), k, pose)
        tracer.trace(out)
        sync()
        assert np.array_equal(tracer.bounds.cpu().numpy(), want[3])
        assert np.array_equal(out.depth.cpu().numpy(), want[0])
        assert np.array_equal(out.color.cpu().numpy(), want[1])
        assert np.array_equal(out.normals.cpu().numpy(), want[2], equal_nan=True)
        hf.depth_to_world = pose_a

    for _ in range(4):
        hv.set_view(hf, orc.POLICY_MAXKEY)
        dv.set_view(df)
        assert vb.valid == 0                      # a new visible list invalidates the record
    orc.integrate_depth(hv, hf)
    orc.integrate_color(hv, hf)
    integ.integrate(df)
    assert vb.valid == 1                          # prepared by the integrate launch
    check(pose_a)                                 # hit
    assert vb.valid == 1
    check(pose_b)                                 # other view: recomputed, record now holds pose_b
    assert vb.valid == 1 and np.array_equal(np.array(vb.depth_to_world.m[:]), np.array(pose_b.m[:]))
    check(pose_a)                                 # and back
    tracer.depth_range = (0.5, 1.35)              # other tracer settings: recomputed
    hf_bounds = orc.trace(hv, hf, depth_range=(0.5, 1.35))
    out = api.Frame(torch.zeros((h, w), dtype=torch.float32, device="cuda"), k, pose_a)
    tracer.trace(out)
    sync()
    assert np.array_equal(tracer.bounds.cpu().numpy(), hf_bounds[3])
    assert np.array_equal(out.depth.cpu().numpy(), hf_bounds[0])


def test_pool_beyond_4_gib_gives_the_same_images(api):
    """A 12.9 GB voxel pool (byte offsets past 2^32, slots past 2^20) against a small
    one: slot numbers and table sizes differ, the fused surface and its raycast
    cannot. Checks 64-bit addressing end to end without a CPU mirror of the pool."""
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 20 << 30:
        pytest.skip("needs 20 GB of free HBM")
    w, h = 320, 240
    k = T.Projection.make(270, 270, 160, 120)
    y, x = np.mgrid[0:h, 0:w]
    depth = (1.3 + 0.2 * np.sin(x / 23.0) * np.cos(y / 17.0)).astype(np.float32)
    color = scenes.checker_color(w, h, 0.2, 0.8)
    outputs = []
    for main, excess in ((16384, 2048), (1200007, 60001)):
        vol = api.Volume(main, excess, voxel_length=0.008, truncation_length=0.04)
        integ, tracer = api.ColorIntegrator(vol), api.Tracer(vol)
        for i in range(3):
            pose = scenes.orbit_pose(i, 2.0)
            frame = api.Frame(depth, k, pose, color=color)
            for _ in range(3):
                vol.set_view(frame)
            integ.integrate(frame)
        out = api.Frame(torch.zeros((h, w), dtype=torch.float32, device="cuda"), k, scenes.orbit_pose(1, 2.0))
        tracer.trace(out)
        sync()
        ctr = vol.read_counters()
        assert ctr[T.VK_CTR_DROPPED] == 0
        if main > 1 << 20:
            # the free list hands out the highest slots first: the blocks in use sit past 4 GiB
            slots = vol.host_entries()["data"]
            assert int(slots.max()) * 10240 > 1 << 32
        outputs.append((vol.visible_count, out.depth.cpu().numpy(), out.color.cpu().numpy(), out.normals.cpu().numpy()))
        del vol, integ, tracer, out
        torch.cuda.empty_cache()
    small, big = outputs
    assert small[0] == big[0] > 500
    for a, b in zip(small[1:], big[1:]):
        assert np.array_equal(a, b, equal_nan=True)
    assert (small[1] > 0).mean() > 0.8


def test_separate_colour_camera(api, orc):
    """Colour camera with its own intrinsics and a depth->colour offset (a real RGB-D
    rig): the colour and light integrators project voxels into it, the colour and
    light trackers work in it. Everything else in the suite uses registered cameras."""
    import torch
    w, h = 320, 240
    kd = T.Projection.make(270, 271, 160.5, 119.25)
    kc = T.Projection.make(281, 279, 157.0, 123.5)
    Tcd = T.Transform.translate(0.025, -0.003, 0.004) * T.Transform.rotate(0.9999, 0.004, -0.01, 0.006)
    y, x = np.mgrid[0:h, 0:w]
    depth = (1.3 + 0.2 * np.sin(x / 23.0) * np.cos(y / 17.0)).astype(np.float32)
    color = (0.25 + 0.5 * scenes.checker_color(w, h, 0.2, 0.8) * (0.6 + 0.4 * np.sin(x / 11.0))[..., None]).astype(np.float32)
    pose = scenes.tracer_test_pose()

    def make(p):
        hf = orc.HostFrame(depth, kd, p, color=color, color_projection=kc, depth_to_color=Tcd)
        hf.compute_normals()
        df = api.Frame(depth, kd, p, color=color, normals=hf.normals, color_projection=kc, depth_to_color=Tcd)
        return hf, df

    hf, df = make(pose)
    light = T.Light.make(2.0, (0.025, 0.08, 0.0))

    # colour integrator (fused and two-pass), then light integrator on a second volume
    for kind in ("color", "light"):
        hv, dv = make_pair(api, orc, 16384, 2048, 0.008, 0.04)
        for _ in range(4):
            hv.set_view(hf, orc.POLICY_MAXKEY)
            dv.set_view(df)
        orc.integrate_depth(hv, hf)
        if kind == "color":
            orc.integrate_color(hv, hf)
            api.ColorIntegrator(dv).integrate(df)
        else:
            mask = orc.light_frame_mask(hf, 0.2)
            orc.integrate_light_color(hv, hf, light, mask)
            integ = api.LightIntegrator(dv)
            integ.light = light
            integ.integrate(df)
        sync()
        assert_volume_equal(dv, hv)
        assert (hv.voxels["color_weight"] > 0).sum() > 10000

    # colour and light trackers: keyframe = this frame, frame = slightly moved copy
    moved = T.Transform.translate(0.003, -0.002, 0.001) * pose
    hf2, df2 = make(moved)
    ks, fs = orc.ColorSide(hf, False), orc.ColorSide(hf2, True)
    Tcm = orc.color_tcm(hf, hf2)
    ct = api.ColorTracker()
    ct.keyframe = df
    assert np.array_equal(np.array(ct.tcm(df2).m[:]), np.array(Tcm.m[:]))
    assert np.array_equal(ct.compute_residuals(df2).cpu().numpy().view(np.uint32), orc.color_residuals(ks, fs, Tcm).view(np.uint32))
    assert np.array_equal(ct.compute_jacobian(df2).cpu().numpy().view(np.uint32), orc.color_jacobian(ks, fs, Tcm, True).view(np.uint32))
    lt = api.LightTracker()
    lt.keyframe = df
    lt.light = light
    mask = orc.light_frame_mask(hf2, 0.2)
    terms = orc.light_terms(hf2, light, mask)
    dmask = torch.from_numpy(mask).cuda()
    assert np.array_equal(lt.compute_residuals(df2, dmask).cpu().numpy().view(np.uint32), orc.light_residuals(ks, fs, terms, Tcm).view(np.uint32))
    assert np.array_equal(lt.compute_jacobian(df2, dmask).cpu().numpy().view(np.uint32), orc.light_jacobian(ks, fs, terms, Tcm, True).view(np.uint32))
    assert np.count_nonzero(orc.light_residuals(ks, fs, terms, Tcm)) > 10000


def test_plain_entry_points_equal_the_ahead_ones(api, orc):
    """vk_integrate_depth / _depth_color / _depth_light and vk_trace (the entry points a
    maintainer binds one to one) give the same voxels and images as vk_integrate_ahead /
    vk_trace_ahead, which the Python and C++ class layers call."""
    import ctypes as C
    import torch
    w, h = 320, 240
    k = T.Projection.make(270, 270, 160, 120)
    y, x = np.mgrid[0:h, 0:w]
    depth = (1.3 + 0.2 * np.sin(x / 23.0) * np.cos(y / 17.0)).astype(np.float32)
    pose = scenes.tracer_test_pose()
    hf, df = frames(api, orc, depth, k, pose, color=scenes.checker_color(w, h, 0.2, 0.8))
    df.compute_normals()
    light = T.Light.make(2.0, (0.025, 0.08, 0.0))
    lib = api.lib()
    for mode in (0, 1, 2):
        vols = []
        for ahead in (False, True):
            vol = api.Volume(16384, 2048, voxel_length=0.008, truncation_length=0.04)
            tracer = api.Tracer(vol)
            for _ in range(4):
                vol.set_view(df)
            params = T.Integrator.default()
            mask = torch.empty((h, w), dtype=torch.float32, device="cuda")
            api.check(lib.vk_light_compute_frame_mask(api._ref(df.desc()), 0.2, api._ptr(mask), api.stream()), "mask")
            out = api.Frame(torch.zeros((h, w), dtype=torch.float32, device="cuda"), k, pose)
            out.color = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
            out.normals = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
            v, f, o = api._ref(vol.desc()), api._ref(df.desc()), api._ref(out.desc())
            if ahead:
                api.check(lib.vk_integrate_ahead(v, api._ref(params), f, mode, api._ref(light), api._ptr(mask), None,
                                                 api._ref(tracer.view_bounds), api.stream()), "ahead")
                assert tracer.view_bounds.valid == 1
                api.check(lib.vk_trace_ahead(v, o, api._ref(tracer.view_bounds), api._ptr(out.depth), api._ptr(out.color),
                                             api._ptr(out.normals), api.stream()), "trace_ahead")
            else:
                if mode == 0:
                    api.check(lib.vk_integrate_depth(v, api._ref(params), f, api.stream()), "depth")
                elif mode == 1:
                    api.check(lib.vk_integrate_depth_color(v, api._ref(params), f, api.stream()), "depth_color")
                else:
                    api.check(lib.vk_integrate_depth_light(v, api._ref(params), api._ref(light), api._ptr(mask), f,
                                                           api.stream()), "depth_light")
                api.check(lib.vk_trace(v, o, 0.1, 5.0, api._ptr(tracer.bounds_scratch), tracer.BOUNDS_W, tracer.BOUNDS_H,
                                       api._ptr(out.depth), api._ptr(out.color), api._ptr(out.normals), api.stream()), "trace")
            sync()
            vols.append((vol.host_voxels().tobytes(), out.depth.cpu().numpy(), out.color.cpu().numpy(),
                         out.normals.cpu().numpy(), tracer.bounds.cpu().numpy()))
        a, b = vols
        assert a[0] == b[0]
        for p, q in zip(a[1:], b[1:]):
            assert np.array_equal(p, q, equal_nan=True)
        assert (a[1] > 0).mean() > 0.5


def test_reduce_hook_between_system_and_solve(api, orc):
    """The multi-GPU hook is called once per Gauss-Newton step with the packed
    48-float system on the device. Two identical ranks: the all-reduce doubles the system,
    which leaves every step — and so the final pose — bit-identical (a power-of-two scale
    is exact through the LDL^T solve)."""
    import torch
    import color_scenes as cs
    k = cs.projection()
    kd, kc = cs.keyframe_images()
    key_h = orc.HostFrame(kd, k, cs.keyframe_pose(), color=kc)
    key_h.compute_normals()
    key = api.Frame(kd, k, cs.keyframe_pose(), color=kc, normals=key_h.normals)
    start = T.Transform.translate(0.004, -0.002, 0.001) * cs.keyframe_pose()
    for cls in (api.DepthTracker, api.ColorTracker):
        poses, calls = [], []
        for hooked in (False, True):
            tracker = cls()
            tracker.keyframe = key
            tracker.max_iterations = 6
            if hooked:
                def hook(system):
                    assert system.is_cuda and system.numel() == 48
                    system.mul_(2.0)
                    calls.append(1)
                    return system
                tracker.reduce_hook = hook
            frame = api.Frame(key.depth, k, start, color=key.color, normals=key.normals)
            tracker.track(frame)
            sync()
            poses.append(np.array(frame.depth_to_world.m[:], dtype=np.float32))
        # one call per step that was ENQUEUED: every poll_chunk steps the loop looks at the
        # convergence mirror — at the state one chunk back, so that a chunk of launches is always
        # queued behind the look — and stops enqueuing once |update| < 1e-6 (vk_track_poll)
        steps, converged = (int(v) for v in tracker.state.cpu().numpy())
        chunk = tracker.poll_chunk or 6
        expected = min(6, (-(-steps // chunk) + 1) * chunk) if converged else 6
        assert len(calls) == expected and steps <= len(calls)
        assert np.array_equal(poses[0].view(np.uint32), poses[1].view(np.uint32))
        assert not np.array_equal(poses[0], np.array(start.m[:], dtype=np.float32))
