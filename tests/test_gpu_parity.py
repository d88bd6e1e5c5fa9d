"""Parity of the HIP path (through the C ABI, include/vk.h) with the CPU oracle
on identical inputs. Integer / index state must be bit-exact; float32 results
are computed in the same operation order with contraction off on both sides, so
they are compared bit-exact too where no transcendental is involved (tolerances
are written next to each assertion otherwise).

Run on the GPU box:  python -m pytest tests -m gpu -x -q
"""
import numpy as np
import pytest

import scenes
from vulcan_amd import vk_types as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    import torch
    assert torch.cuda.is_available()
    from vulcan_amd import api as a
    a.lib()
    return a


def sync():
    import torch
    torch.cuda.synchronize()


def assert_volume_equal(dv, hv, voxels=True, order_free_visible=True):
    """Whole-state comparison; allocation is deterministic on both sides (max-key
    winner, scan-ordered slots), so raw buffers must match byte for byte."""
    sync()
    ctr = dv.read_counters()
    for i in (T.VK_CTR_VISIBLE, T.VK_CTR_VOXEL_PTR, T.VK_CTR_EXCESS_PTR, T.VK_CTR_DROPPED):
        assert ctr[i] == hv.counters[i], (i, ctr, hv.counters)
    assert np.array_equal(dv.host_entries(), hv.hash_entries)
    assert np.array_equal(dv.host_visibility(), hv.block_visibility)
    assert np.array_equal(dv.host_allocation_types(), hv.allocation_types)
    assert np.array_equal(dv.host_allocation_blocks(), hv.allocation_blocks)
    n = int(ctr[T.VK_CTR_VISIBLE])
    got = dv.visible_blocks[:n].cpu().numpy()
    assert np.array_equal(np.sort(got), hv.visible())     # order is unspecified (volume.cu:80-83)
    if voxels:
        assert dv.host_voxels().tobytes() == hv.voxels.tobytes()


def make_pair(api, orc, main, excess, voxel, trunc):
    hv = orc.HostVolume(main, excess, voxel_length=voxel, truncation_length=trunc)
    dv = api.Volume(main, excess, voxel_length=voxel, truncation_length=trunc)
    return hv, dv


def frames(api, orc, depth, k, pose, color=None, normals=None):
    hf = orc.HostFrame(depth, k, pose, color=color, normals=normals)
    df = api.Frame(depth, k, pose, color=color, normals=normals)
    return hf, df


# ------------------------------------------------------------------ volume --

def test_initialize(api, orc):
    hv, dv = make_pair(api, orc, 1024, 512, 0.008, 0.04)
    assert_volume_equal(dv, hv)
    assert np.array_equal(dv.free_voxel_blocks.cpu().numpy(), hv.free_voxel_blocks)


def test_reset_block_visibility(api, orc):
    """volume_test.cpp:100-122 pattern plus UNKNOWN entries and an odd length"""
    hv, dv = make_pair(api, orc, 1021, 510, 0.008, 0.04)   # 1531 entries: exercises the byte tail
    hv.block_visibility[:] = np.arange(hv.max) % 3
    dv.upload(hv)
    hv.reset_block_visibility()
    dv.reset_block_visibility()
    sync()
    assert np.array_equal(dv.host_visibility(), hv.block_visibility)


def test_create_allocation_requests_reference_case(api, orc):
    """volume_test.cpp:250-431 scene: ramp depth, translated pose, one forced collision."""
    depth = scenes.ramp(64, 48)
    k, pose = T.Projection.make(32, 32, 32, 24), T.Transform.translate(-10.73, 2.11, -33.54)
    hf, df = frames(api, orc, depth, k, pose)
    hv, dv = make_pair(api, orc, 1024, 512, 0.02, 0.20)
    hv.hash_entries["block"]["origin"][5] = (-1, -1, -1)
    hv.hash_entries["data"][5] = 0
    dv.upload(hv)
    hv.create_allocation_requests(hf, orc.POLICY_MAXKEY)
    dv.create_allocation_requests(df)
    assert_volume_equal(dv, hv, voxels=False)
    assert (hv.allocation_types == T.ALLOC_EXCESS).sum() >= 1


def test_handle_allocation_requests_reference_case(api, orc):
    """volume_test.cpp:433-556"""
    hv, dv = make_pair(api, orc, 1024, 512, 0.008, 0.04)
    for blocks, typ in ((((0, (1, 2, 3)), (323, (7, 3, -1))), T.ALLOC_MAIN),
                        (((0, (7, 3, 0)), (323, (-9, 1, -2))), T.ALLOC_EXCESS)):
        for idx, b in blocks:
            hv.allocation_blocks["origin"][idx] = b
            hv.allocation_blocks["pad"][idx] = typ
            hv.allocation_types[idx] = typ
        dv.upload(hv)
        hv.handle_allocation_requests()
        dv.handle_allocation_requests()
        assert_volume_equal(dv, hv, voxels=False)
    e = dv.host_entries()
    assert tuple(e[e[0]["next"]]["block"]["origin"]) == (7, 3, 0)


def test_handle_pool_exhaustion(api, orc):
    """volume.cu:356 guard: requests beyond the pool / excess list are dropped, counted."""
    hv, dv = make_pair(api, orc, 64, 2, 0.008, 0.04)
    hv.hash_entries["data"][:8] = np.arange(8)            # 8 occupied buckets -> EXCESS requests
    hv.hash_entries["block"]["origin"][:8] = 99
    for idx in range(8):
        hv.allocation_blocks["origin"][idx] = (idx, 1, 2)
        hv.allocation_blocks["pad"][idx] = T.ALLOC_EXCESS
        hv.allocation_types[idx] = T.ALLOC_EXCESS
    dv.upload(hv)
    hv.handle_allocation_requests()
    dv.handle_allocation_requests()
    assert_volume_equal(dv, hv, voxels=False)
    assert hv.counters[T.VK_CTR_DROPPED] == 6


def test_update_block_visibility_reference_case(api, orc):
    """volume_test.cpp:124-248"""
    hv, dv = make_pair(api, orc, 1024, 512, 0.008, 0.04)
    hf, df = frames(api, orc, np.zeros((480, 640), np.float32), T.Projection.make(320, 320, 320, 240),
                    T.Transform.translate(10, -2, 30))
    hv.block_visibility[[7, 32, 123]] = T.VISIBILITY_TRUE
    hv.block_visibility[[3, 17, 315]] = T.VISIBILITY_UNKNOWN
    scale = np.float32(1.0) / (np.float32(8) * np.float32(hv.voxel_length))
    for idx, p in ((3, (10, -2, 33)), (17, (-10, -2, 28)), (315, (11, -1, 53))):
        hv.hash_entries["block"]["origin"][idx] = [np.int16(int(np.float32(c) * scale)) for c in p]
    dv.upload(hv)
    hv.update_block_visibility(hf)
    dv.update_block_visibility(df)
    assert_volume_equal(dv, hv, voxels=False)
    assert sorted(dv.visible().tolist()) == [3, 7, 32, 123, 315]


@pytest.mark.parametrize("scene", ["plane", "sphere", "ramp"])
def test_set_view_rounds_match_oracle(api, orc, scene):
    """Every SetView round (not only the fixed point) leaves identical state."""
    w, h = 320, 240
    depth = {"plane": scenes.plane(w, h, 1.5), "sphere": scenes.sphere(2 * w, 2 * h)[::2, ::2].copy(),
             "ramp": scenes.ramp(w, h)}[scene]
    k = T.Projection.make(272.0, 272.0, 155.6, 117.4)
    pose = scenes.tracer_test_pose()
    hf, df = frames(api, orc, depth, k, pose)
    hv, dv = make_pair(api, orc, 8192, 4096, 0.01, 0.04)
    prev = -1
    for _ in range(12):
        hv.set_view(hf, orc.POLICY_MAXKEY)
        dv.set_view(df)
        assert_volume_equal(dv, hv, voxels=False)
        if hv.visible_count == prev:
            break
        prev = hv.visible_count
    assert hv.visible_count > 500 and hv.counters[T.VK_CTR_EXCESS_PTR] > hv.main   # chains exercised


# -------------------------------------------------------------- integrators --

def fused_pair(api, orc, depth, color, k, pose, main=8192, excess=4096, voxel=0.008, trunc=0.04, normals=False):
    hf, df = frames(api, orc, depth, k, pose, color=color)
    if normals:
        hf.compute_normals()
        df.compute_normals()
    hv, dv = make_pair(api, orc, main, excess, voxel, trunc)
    for _ in range(8):
        hv.set_view(hf, orc.POLICY_MAXKEY)
        dv.set_view(df)
    return hv, dv, hf, df


def test_integrate_reference_case(api, orc):
    """integrator_test.cu:82-221 on the device; two passes."""
    w, h = 160, 120
    hv, dv, hf, df = fused_pair(api, orc, scenes.plane(w, h, 1.5), scenes.constant_color(w, h, (1, 2, 3)),
                                T.Projection.make(80, 80, 80, 60), T.Transform.identity(), 4096, 2048, 0.008, 0.02)
    integ = api.ColorIntegrator(dv)
    for _ in range(2):
        orc.integrate_depth(hv, hf)
        orc.integrate_color(hv, hf)
        integ.integrate(df)
        assert_volume_equal(dv, hv)
    assert hv.voxels["distance_weight"].max() == 2


@pytest.mark.parametrize("mode", ["depth", "color_two_pass", "fused"])
def test_integrate_modes_match_oracle(api, orc, mode):
    w, h = 320, 240
    k = T.Projection.make(272.0, 272.0, 155.6, 117.4)
    hv, dv, hf, df = fused_pair(api, orc, scenes.sphere(2 * w, 2 * h)[::2, ::2].copy(),
                                scenes.checker_color(w, h), k, scenes.tracer_test_pose(), voxel=0.01)
    integ = api.ColorIntegrator(dv)
    for rep in range(3):
        orc.integrate_depth(hv, hf)
        if mode != "depth":
            orc.integrate_color(hv, hf)
        if mode == "depth":
            api.DepthIntegrator(dv).integrate(df)
        elif mode == "color_two_pass":
            integ.integrate_depth(df)
            integ.integrate_color(df)
        else:
            integ.integrate(df)
        assert_volume_equal(dv, hv)
    assert (hv.voxels["distance_weight"] == 3).sum() > 100000


def test_weight_clamp_and_depth_range(api, orc):
    w, h = 160, 120
    depth = scenes.plane(w, h, 1.5)
    depth[:, :40] = 7.0      # beyond max depth: ignored (depth_integrator.cu:54)
    depth[:, 120:] = 0.0     # invalid
    hv, dv, hf, df = fused_pair(api, orc, depth, scenes.constant_color(w, h), T.Projection.make(136, 136, 80, 60),
                                T.Transform.identity(), 4096, 1024)
    integ = api.ColorIntegrator(dv)
    integ.params = T.Integrator(0.1, 5.0, 3.0, 2.0)
    for _ in range(5):
        orc.integrate_depth(hv, hf, integ.params)
        orc.integrate_color(hv, hf, integ.params)
        integ.integrate(df)
    assert_volume_equal(dv, hv)
    assert hv.voxels["distance_weight"].max() == 3 and hv.voxels["color_weight"].max() == 2


def test_light_integrator_matches_oracle(api, orc):
    """light_integrator.cu: mask (asymmetric 7x7 window), depth, shaded colour."""
    w, h = 320, 240
    k = T.Projection.make(272.0, 272.0, 155.6, 117.4)
    depth = scenes.sphere(2 * w, 2 * h)[::2, ::2].copy()
    depth[depth == 0] = 3.9
    color = scenes.checker_color(w, h, 0.1, 0.9)
    color[:20, :, 0] = 0.99   # saturated rows fail the mask
    hv, dv, hf, df = fused_pair(api, orc, depth, color, k, T.Transform.identity(), voxel=0.01, normals=True)
    light = T.Light.make(2.0, (0.025, 0.08, 0.0))        # apps/vulcan/vulcan.cu:87-88
    integ = api.LightIntegrator(dv)
    integ.light = light
    mask_h = orc.light_frame_mask(hf, 0.2)
    mask_d = integ.compute_frame_mask(df)
    sync()
    assert np.array_equal(mask_d.cpu().numpy(), mask_h)
    assert 0.05 < mask_h.mean() < 0.99
    assert np.array_equal(df.normals.cpu().numpy(), hf.normals, equal_nan=True)
    for rep in range(2):
        orc.integrate_depth(hv, hf)
        orc.integrate_light_color(hv, hf, light, mask_h)
        if rep == 0:
            integ.integrate(df)                 # fused depth + light colour
        else:
            integ.integrate_depth(df)           # two-pass form
            integ.integrate_color(df)
        assert_volume_equal(dv, hv)
    assert (hv.voxels["color_weight"] > 0).sum() > 10000


# ------------------------------------------------------------------- tracer --

def test_compute_patches_and_bounds_reference_cases(api, orc):
    """tracer_test.cu:22-244 inputs on the device vs the oracle."""
    import torch
    block_length = np.float32(0.008)
    entries, tcw, k = scenes.tracer_patch_kat()
    want, count = orc.compute_patches(np.arange(5), entries, tcw, k, block_length, 0.1, 5.0, 640, 480)

    d_entries = torch.from_numpy(np.frombuffer(entries.tobytes(), np.uint8).copy()).cuda()
    d_idx = torch.arange(5, dtype=torch.int32, device="cuda")
    d_patches = torch.zeros(64 * 16, dtype=torch.uint8, device="cuda")
    d_count = torch.zeros(1, dtype=torch.int32, device="cuda")
    api.check(api.lib().vk_trace_compute_patches(
        api._ptr(d_idx), api._ptr(d_entries), api._ref(tcw), api._ref(k), block_length, 0.1, 5.0, 5, None,
        640, 480, 80, 60, api._ptr(d_patches), 64, api._ptr(d_count), api.stream()), "patches")
    sync()
    assert int(d_count.cpu()[0]) == count
    got = api.to_numpy(d_patches[:count * 16], T.patch_dtype)
    assert sorted(got.tobytes()[i:i + 16] for i in range(0, 16 * count, 16)) == \
        sorted(want.tobytes()[i:i + 16] for i in range(0, 16 * count, 16))

    lit = [((23, 46), (5, 2), (1.237, 1.523)), ((3, 9), (1, 1), (2.021, 3.214)),
           ((20, 43), (5, 8), (0.856, 1.014)), ((0, 0), (2, 2), (1.256, 2.114)),
           ((79, 59), (1, 1), (0.256, 1.314)), ((3, 9), (3, 3), (0.256, 1.314))]
    patches = np.zeros(len(lit), dtype=T.patch_dtype)
    for i, p in enumerate(lit):
        patches[i] = p
    d_p = torch.from_numpy(np.frombuffer(patches.tobytes(), np.uint8).copy()).cuda()
    d_b = torch.zeros((60, 80, 2), dtype=torch.float32, device="cuda")
    api.check(api.lib().vk_trace_reset_bounds(api._ptr(d_b), 4800, api.stream()), "reset")
    api.check(api.lib().vk_trace_compute_bounds(api._ptr(d_p), api._ptr(d_b), 80, len(lit), None, api.stream()), "bounds")
    sync()
    assert np.array_equal(d_b.cpu().numpy(), orc.compute_bounds(patches))


@pytest.fixture(scope="module")
def traced(api, orc):
    """tracer_test.cu:392-588 scene (sphere cap + checker), fused on both sides."""
    w, h = 640, 480
    orc.set_threads(16)
    depth, color = scenes.sphere(w, h), scenes.checker_color(w, h)
    color[depth == 0] = 0
    hv, dv, hf, df = fused_pair(api, orc, depth, color, T.Projection.make(*scenes.TRACER_TEST_INTRINSICS),
                                scenes.tracer_test_pose(), 8192, 4096)
    orc.integrate_depth(hv, hf)
    orc.integrate_color(hv, hf)
    api.ColorIntegrator(dv).integrate(df)
    assert_volume_equal(dv, hv)
    return hv, dv, hf, df


def test_patches_bounds_and_fused_bounds(api, orc, traced):
    hv, dv, hf, df = traced
    tracer = api.Tracer(dv)
    tracer.compute_patches(df)
    tracer.compute_bounds()
    sync()
    block_length = np.float32(8) * np.float32(hv.voxel_length)
    want, count = orc.compute_patches(hv.visible(), hv.hash_entries, hf.depth_to_world.inverse(),
                                      hf.depth_projection, block_length, 0.1, 5.0, hf.width, hf.height)
    got = tracer.host_patches()
    assert len(got) == count
    key = lambda a: sorted(a.tobytes()[i:i + 16] for i in range(0, 16 * len(a), 16))
    assert key(got) == key(want)
    bounds = orc.compute_bounds(want)
    via_patches = tracer.bounds.cpu().numpy().copy()
    assert np.array_equal(via_patches, bounds)
    tracer.bounds.zero_()
    tracer.compute_block_bounds(df)
    sync()
    assert np.array_equal(tracer.bounds.cpu().numpy(), bounds)   # fused path is bit-identical


def test_trace_matches_oracle(api, orc, traced):
    import torch
    hv, dv, hf, df = traced
    odepth, ocolor, onormals, obounds, steps = orc.trace(hv, hf, want_steps=True)
    out = api.Frame(torch.zeros((hf.height, hf.width), dtype=torch.float32, device="cuda"),
                    hf.depth_projection, hf.depth_to_world)
    tracer = api.Tracer(dv)
    tracer.trace(out)
    sync()
    d, c, n = out.depth.cpu().numpy(), out.color.cpu().numpy(), out.normals.cpu().numpy()
    assert np.array_equal(tracer.bounds.cpu().numpy(), obounds)     # Tracer::bounds_ after Trace
    assert (odepth > 0).sum() > 100000 and steps.max() < 500
    assert np.array_equal(d, odepth)                        # bit-exact raycast depth
    assert np.array_equal(c, ocolor)
    assert np.array_equal(n, onormals, equal_nan=True)
    # and both satisfy the reference's own bar (tracer_test.cu:549-563)
    y, x = np.mgrid[0:hf.height, 0:hf.width]
    r = np.hypot(x + 0.5 - hf.width / 2, y + 0.5 - hf.height / 2)
    keep = ~((r >= 180) & (r <= 203))
    assert np.abs(d - hf.depth)[keep].max() < 0.05


def test_normals_and_filter(api, orc):
    w, h = 640, 480
    depth = scenes.ripple(w, h)
    depth[100:110, 200:260] = 0      # holes: neighbours fall back to the centre (frame.cu:54-57)
    k = T.Projection.make(*scenes.APP_INTRINSICS)
    hf, df = frames(api, orc, depth, k, T.Transform.identity())
    orc.set_threads(16)
    want = hf.compute_normals()
    got = df.compute_normals()
    sync()
    assert np.array_equal(got.cpu().numpy(), want, equal_nan=True)
    df.filter_depths()
    sync()
    # frame.cu:126-181 uses expf: libm vs device expf differ in the last ulp
    np.testing.assert_allclose(df.depth.cpu().numpy(), orc.filter_depths(depth), rtol=2e-6, atol=1e-7)


def test_downsample(api, orc):
    rng = np.random.default_rng(0)
    depth = rng.random((480, 640), dtype=np.float32)
    color = rng.random((480, 640, 3), dtype=np.float32)
    k = T.Projection.make(*scenes.APP_INTRINSICS)
    hf, df = frames(api, orc, depth, k, T.Transform.identity(), color=color, normals=color[::-1].copy())
    half = df.downsample()
    sync()
    assert np.array_equal(half.depth.cpu().numpy(), orc.downsample(depth, True))
    assert np.array_equal(half.color.cpu().numpy(), orc.downsample(color, False))
    assert np.array_equal(half.normals.cpu().numpy(), orc.downsample(color[::-1].copy(), True))
    assert half.depth_projection.fx == np.float32(k.fx) * np.float32(0.5)


# ---------------------------------------------------------------------- ICP --

@pytest.fixture(scope="module")
def icp_frames(api, orc):
    """depth_tracker_test.cu:12-126"""
    w, h = 640, 480
    k = T.Projection.make(547, 547, 320, 240)
    pose = T.Transform.translate(0.001, -0.002, 0.003) * T.Transform.rotate(0.9998719, 0.0085884, -0.0104268, 0.0085884)
    orc.set_threads(16)
    hk, dk = frames(api, orc, scenes.plane(w, h, 1.0), k, T.Transform.identity())
    hf, df = frames(api, orc, scenes.ripple(w, h), k, pose)
    for f in (hk, dk, hf, df):
        f.compute_normals()
    return hk, dk, hf, df


def test_icp_residuals_and_jacobian(api, orc, icp_frames):
    hk, dk, hf, df = icp_frames
    tracker = api.DepthTracker()
    tracker.keyframe = dk
    r = tracker.compute_residuals(dk)
    sync()
    assert np.all(r.cpu().numpy() == 0)                         # depth_tracker_test.cu:396-406
    r = tracker.compute_residuals(df).cpu().numpy()
    assert np.array_equal(r, orc.icp_residuals(hk, hf))
    for enabled in (True, False):
        tracker.translation_enabled = enabled
        J = tracker.compute_jacobian(df).cpu().numpy()
        assert np.array_equal(J, orc.icp_jacobian(hk, hf, enabled))


def test_icp_system(api, orc, icp_frames):
    """27 sums: float32 tree sums on the device vs float64 sums of the same
    float32 terms; |diff| <= 2e-5 * sum|terms| (float32 eps * log-depth slack)."""
    hk, dk, hf, df = icp_frames
    tracker = api.DepthTracker()
    tracker.keyframe = dk
    J = orc.icp_jacobian(hk, hf, True).reshape(-1, 6).astype(np.float64)
    r = orc.icp_residuals(hk, hf).reshape(-1).astype(np.float64)
    for enabled in (True, False):
        tracker.translation_enabled = enabled
        tracker.compute_system(df)
        sync()
        H, g = tracker.hessian.cpu().numpy(), tracker.gradient.cpu().numpy()
        wantH, wantg = orc.icp_system(hk, hf, enabled)
        n = 6 if enabled else 3
        packed_abs = np.array([np.abs(J[:, i] * J[:, j]).sum() for i in range(n) for j in range(i + 1)])
        cnt = len(packed_abs)
        assert np.all(np.abs(H[:cnt] - wantH[:cnt]) <= 2e-5 * packed_abs + 1e-12)
        assert np.all(H[cnt:] == 0)
        gabs = np.abs(J[:, :n] * r[:, None]).sum(0)
        assert np.all(np.abs(g[:n] - wantg[:n]) <= 2e-5 * gabs + 1e-12)
        first = (H.copy(), g.copy())
        tracker.compute_system(df)
        sync()
        assert np.array_equal(tracker.hessian.cpu().numpy(), first[0])   # fixed-order reduction: reproducible
        assert np.array_equal(tracker.gradient.cpu().numpy(), first[1])


def test_icp_track_converges(api, orc, icp_frames):
    """Device-side Gauss-Newton (no host round trip per iteration) against the oracle's
    host loop on the same system; tracker.cpp:53-63,124-163."""
    hk, dk, hf, df = icp_frames
    w, h = hk.width, hk.height
    k = hk.depth_projection
    true_pose = T.Transform.translate(0.0, 0.0, 0.004) * T.Transform.rotate(0.99999, 0.003, -0.002, 0.0)
    R, t = true_pose.matrix()[:3, :3].astype(np.float64), true_pose.matrix()[:3, 3].astype(np.float64)
    y, x = np.mgrid[0:h, 0:w]
    rays = np.stack([(x + 0.5 - k.cx) / k.fx, (y + 0.5 - k.cy) / k.fy, np.ones((h, w))], -1)
    # a gently curved keyframe makes all six parameters observable
    key_depth = (1.0 + 0.05 * np.cos(3 * x / w) * np.sin(2 * y / h)).astype(np.float32)
    hk2, dk2 = frames(api, orc, key_depth, k, T.Transform.identity())
    hk2.compute_normals()
    dk2.compute_normals()
    # the frame is the keyframe itself seen from a slightly wrong initial pose
    start = T.Transform.translate(0.002, -0.001, 0.003) * T.Transform.rotate(0.999995, 0.002, -0.0015, 0.001)
    hf2, df2 = frames(api, orc, key_depth, k, start)
    hf2.compute_normals()
    df2.compute_normals()

    tracker = api.DepthTracker()
    tracker.keyframe = dk2
    got = tracker.track(df2)
    sync()
    state = tracker.state.cpu().numpy()
    pose = hf2.depth_to_world
    for _ in range(20):
        hf2.depth_to_world = pose
        Hs, g = orc.icp_system(hk2, hf2, True)
        pose, upd, norm = orc.icp_solve_update(Hs, g, pose, True)
        if norm < 1e-6:
            break
    assert state[0] >= 3
    np.testing.assert_allclose(got.matrix(), pose.matrix(), atol=2e-5)
    np.testing.assert_allclose(got.matrix(), np.eye(4), atol=5e-4)       # converged to the keyframe pose
    np.testing.assert_allclose(got.matrix() @ got.inverse_matrix(), np.eye(4), atol=1e-5)


# ------------------------------------------- full-size properties (5 mm, 640x480) --

def test_full_size_properties(api, orc):
    """BASELINE configs[1] sizes (Volume(65024, 8192), 5 mm voxels): properties that
    do not need the oracle at full size."""
    import torch
    w, h = 640, 480
    k = T.Projection.make(*scenes.APP_INTRINSICS)
    pose = scenes.tracer_test_pose()
    dv = api.Volume(65024, 8192, voxel_length=0.005, truncation_length=0.04)
    df = api.Frame(scenes.plane(w, h, 1.5), k, pose)
    counts = []
    for _ in range(6):
        dv.set_view(df)
        counts.append(dv.visible_count)
    assert counts[-1] == counts[-2] > 3000                        # allocation reaches a fixed point
    ctr = dv.read_counters()
    assert ctr[T.VK_CTR_DROPPED] == 0
    entries = dv.host_entries()
    alloc = entries[entries["data"] >= 0]
    assert len(np.unique(alloc["data"])) == len(alloc)            # no pool slot handed out twice
    assert len(alloc) == dv.max - 1 - ctr[T.VK_CTR_VOXEL_PTR]
    origins = {tuple(o) for o in alloc["block"]["origin"]}
    assert len(origins) == len(alloc)                             # no block allocated twice
    vis = dv.visible()
    assert len(np.unique(vis)) == len(vis)

    integ = api.DepthIntegrator(dv)
    integ.integrate(df)
    sync()
    v1 = dv.host_voxels()
    integ.integrate(df)
    sync()
    v2 = dv.host_voxels()
    touched = v1["distance_weight"] > 0
    assert touched.sum() > 1000000
    assert np.array_equal(v2["distance_weight"][touched], 2 * v1["distance_weight"][touched])
    assert np.abs(v2["distance"] - v1["distance"]).max() <= 1e-6   # running mean of equal samples
    assert np.all(v2["distance_weight"][~touched] == 0) and np.all(v2["distance"][~touched] == 1)

    out = api.Frame(torch.zeros((h, w), dtype=torch.float32, device="cuda"), k, pose)
    api.Tracer(dv).trace(out)
    sync()
    d = out.depth.cpu().numpy()
    assert np.abs(d[3:-2, 3:-2] - 1.5).max() < 0.01               # tracer_test.cu:355-370 bar at 5 mm
    n = out.normals.cpu().numpy()[8:-8, 8:-8]
    assert np.abs(n[..., 2] + 1).max() < 2e-2


def test_bench_sequence_matches_oracle(api, orc):
    """The exact bench.py workload (BASELINE configs[1]: 640x480, 5 mm voxels,
    Volume(65024, 8192), camera yawing inside a 2 m sphere): every frame's
    SetView + Integrate + Trace on the device equals the oracle's, bit for bit."""
    import sys, os, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    k = T.Projection.make(*scenes.APP_INTRINSICS)
    depth = bench.sphere_room_depth(k)
    orc.set_threads(16)
    hv = orc.HostVolume(bench.MAIN, bench.EXCESS, voxel_length=bench.VOXEL, truncation_length=bench.TRUNC)
    dv = api.Volume(bench.MAIN, bench.EXCESS, voxel_length=bench.VOXEL, truncation_length=bench.TRUNC)
    hf = orc.HostFrame(depth, k, T.Transform.identity())
    df = api.Frame(depth, k, T.Transform.identity())
    out = api.Frame(torch.zeros((bench.H, bench.W), dtype=torch.float32, device="cuda"), k, T.Transform.identity())
    integ, tracer = api.DepthIntegrator(dv), api.Tracer(dv)
    for i in range(5):
        pose = scenes.orbit_pose(i, bench.YAW_STEP)
        hf.depth_to_world = df.depth_to_world = out.depth_to_world = pose
        hv.set_view(hf, orc.POLICY_MAXKEY)
        orc.integrate_depth(hv, hf)
        odepth, ocolor, onormals, obounds = orc.trace(hv, hf)
        dv.set_view(df)
        integ.integrate(df)
        tracer.trace(out)
        sync()
        assert dv.visible_count == hv.visible_count > 5000
        assert np.array_equal(tracer.bounds.cpu().numpy(), obounds)
        assert np.array_equal(out.depth.cpu().numpy(), odepth)
        assert np.array_equal(out.color.cpu().numpy(), ocolor)
        assert np.array_equal(out.normals.cpu().numpy(), onormals, equal_nan=True)
    assert_volume_equal(dv, hv)
    # and the result is the scene: a sphere of radius 2 m seen from its centre
    assert np.abs(out.depth.cpu().numpy()[4:-4, 4:-4] - depth[4:-4, 4:-4]).max() < 0.01
