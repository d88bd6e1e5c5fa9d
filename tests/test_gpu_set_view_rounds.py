"""vk_volume_set_view_rounds: the reference's frame loop calls SetView three times per frame
(apps/vulcan/vulcan.cu:316-318) because a bucket takes one request per call. One call with
`rounds` must leave exactly the state of that many consecutive SetView calls — every buffer,
the visible set, the pool pointers and the request / drop counters — although the rays are
walked once: the later rounds are replayed from the list of requests that lost their bucket,
inside the handle launch, and only when there are any.

The checker is the oracle's set_view called `rounds` times.
"""
import os

import numpy as np
import pytest

import scenes
from test_gpu_parity import api, assert_volume_equal, frames, make_pair, sync  # noqa: F401
from vulcan_amd import vk_types as T

pytestmark = pytest.mark.gpu

K_SMALL = T.Projection.make(272.0, 272.0, 155.6, 117.4)


def oracle_rounds(orc, hv, hf, rounds):
    for _ in range(rounds):
        hv.set_view(hf, orc.POLICY_MAXKEY)


def assert_same_requests(dv, hv):
    ctr = dv.read_counters()
    assert ctr[T.VK_CTR_REQUESTS] == hv.counters[T.VK_CTR_REQUESTS], (ctr, hv.counters)
    return ctr


@pytest.fixture(params=["two launches", "three launches"])
def launches(request):
    """The call as two launches (requests; handle + visibility, the handle pass working from the list
    of posted buckets) and as three (vk_test_hooks.set_view_unfused: handle and visibility apart, the handle
    pass counting the request flags — also the form for tables beyond 67 M entries)."""
    from vulcan_amd import api as a
    if request.param == "three launches":
        with a.test_hooks(set_view_unfused=1):
            yield request.param
    else:
        yield request.param


@pytest.mark.parametrize("scene", ["sphere", "ramp"])
@pytest.mark.parametrize("rounds", [2, 3, 6])
def test_rounds_equal_consecutive_set_views(api, orc, scene, rounds, launches):
    """A table far too small for the scene (1024 buckets for ~2 000 blocks): every round loses
    requests to bucket contests and grows chains, so each round changes the state."""
    w, h = 320, 240
    depth = {"sphere": scenes.sphere(2 * w, 2 * h)[::2, ::2].copy(), "ramp": scenes.ramp(w, h)}[scene]
    hv, dv = make_pair(api, orc, 1024, 16384, 0.01, 0.04)
    for step, pose in enumerate((scenes.tracer_test_pose(), scenes.yaw(4.0) * scenes.tracer_test_pose())):
        hf, df = frames(api, orc, depth, K_SMALL, pose)
        before = int(dv.read_counters()[T.VK_CTR_ROUNDS])
        oracle_rounds(orc, hv, hf, rounds)
        dv.set_view(df, rounds=rounds)
        assert_volume_equal(dv, hv, voxels=False)
        ctr = assert_same_requests(dv, hv)
        ran = int(ctr[T.VK_CTR_ROUNDS]) - before
        assert 1 <= ran <= rounds
        if step == 0 and rounds <= 3:
            assert ran == rounds                              # contested: no round could be skipped
        assert ctr[T.VK_CTR_UNSETTLED] in (0, 1)
    assert hv.counters[T.VK_CTR_EXCESS_PTR] > hv.main + 500   # chains exercised
    assert hv.counters[T.VK_CTR_DROPPED] == 0


def test_later_rounds_are_skipped_when_nothing_is_pending(api, orc):
    """A table with room: the first frame needs a second round for its few collisions, the same
    view again needs none — and the state still equals three SetView calls."""
    w, h = 320, 240
    hf, df = frames(api, orc, scenes.plane(w, h, 1.5), K_SMALL, scenes.tracer_test_pose())
    hv, dv = make_pair(api, orc, 65024, 8192, 0.01, 0.04)
    ran = []
    for _ in range(3):
        before = int(dv.read_counters()[T.VK_CTR_ROUNDS])
        oracle_rounds(orc, hv, hf, 3)
        dv.set_view(df, rounds=3)
        assert_volume_equal(dv, hv, voxels=False)
        ctr = assert_same_requests(dv, hv)
        ran.append(int(ctr[T.VK_CTR_ROUNDS]) - before)
        assert ctr[T.VK_CTR_UNSETTLED] == 0
    assert ran[0] in (1, 2, 3) and ran[1:] == [1, 1], ran


def test_rounds_end_with_the_first_round_that_drops_a_request(api, orc, launches):
    """A pool of 600 blocks for a scene that needs more. Upstream asks again on every call, drops
    again and links excess entries it never writes (volume.cu:337-356); from there on its state is
    inconsistent. The rounds therefore end with the first round that drops a request (vk.h): the
    state is that of so many SetView calls — here the first call fills the 512 main entries, the
    second exhausts the pool — and VK_CTR_UNSETTLED says that requests are unanswered."""
    w, h = 320, 240
    hf, df = frames(api, orc, scenes.sphere(2 * w, 2 * h)[::2, ::2].copy(), K_SMALL, scenes.tracer_test_pose())
    hv, dv = make_pair(api, orc, 512, 88, 0.01, 0.04)
    dv.set_view(df, rounds=5)
    ctr = dv.read_counters()
    assert ctr[T.VK_CTR_ROUNDS] == 2 and ctr[T.VK_CTR_UNSETTLED] == 1
    oracle_rounds(orc, hv, hf, 2)
    assert_volume_equal(dv, hv, voxels=False)
    assert_same_requests(dv, hv)
    assert hv.counters[T.VK_CTR_DROPPED] > 100 and hv.counters[T.VK_CTR_VOXEL_PTR] < 0
    # a pool that is exhausted in the FIRST round: one round, whatever was asked for
    hv, dv = make_pair(api, orc, 2048, 512, 0.01, 0.04)
    hv.counters[T.VK_CTR_VOXEL_PTR] = 99
    dv.upload(hv)
    oracle_rounds(orc, hv, hf, 1)
    dv.set_view(df, rounds=3)
    assert_volume_equal(dv, hv, voxels=False)
    ctr = assert_same_requests(dv, hv)
    assert hv.counters[T.VK_CTR_DROPPED] > 100 and ctr[T.VK_CTR_UNSETTLED] == 1 and ctr[T.VK_CTR_ROUNDS] == 1


def test_a_retry_list_that_is_too_small_ends_the_rounds(api, orc):
    """More lost requests than the retry list holds (forced: vk_test_hooks.retry_capacity = 64): the rounds
    stop, the state is ONE SetView call's with VK_CTR_UNSETTLED = 1, and later calls settle it."""
    w, h = 320, 240
    hf, df = frames(api, orc, scenes.ramp(w, h), K_SMALL, scenes.tracer_test_pose())
    hv, dv = make_pair(api, orc, 2048, 8192, 0.01, 0.04)
    with api.test_hooks(retry_capacity=64):
        oracle_rounds(orc, hv, hf, 1)
        dv.set_view(df, rounds=4)
        assert_volume_equal(dv, hv, voxels=False)
        ctr = dv.read_counters()
        assert ctr[T.VK_CTR_UNSETTLED] == 1 and ctr[T.VK_CTR_ROUNDS] == 1
    for _ in range(6):
        oracle_rounds(orc, hv, hf, 4)
        dv.set_view(df, rounds=4)
        assert_volume_equal(dv, hv, voxels=False)
        assert_same_requests(dv, hv)
    assert dv.read_counters()[T.VK_CTR_UNSETTLED] == 0
    assert hv.counters[T.VK_CTR_EXCESS_PTR] > hv.main + 200


def test_rounds_with_the_light_preparation_riding_along(api, orc):
    """The mask / record pass of LightIntegrator rides in the first round's request launch only."""
    w, h = 320, 240
    depth = scenes.sphere(2 * w, 2 * h)[::2, ::2].copy()
    color = scenes.checker_color(w, h, 0.1, 0.9)
    hf, df = frames(api, orc, depth, K_SMALL, scenes.tracer_test_pose(), color=color)
    hf.compute_normals()
    df.compute_normals()
    hv, dv = make_pair(api, orc, 1024, 8192, 0.01, 0.04)
    light = T.Light.make(2.0, (0.025, 0.08, 0.0))
    integ = api.LightIntegrator(dv)
    integ.light = light
    for i in range(3):
        oracle_rounds(orc, hv, hf, 3)
        dv.set_view(df, rounds=3)
        if i > 0:                                   # the first Integrate registers the integrator's buffers
            assert integ._prep.valid == 1
        orc.integrate_depth(hv, hf)
        orc.integrate_light_color(hv, hf, light, orc.light_frame_mask(hf, 0.2))
        integ.integrate(df)
        assert_volume_equal(dv, hv)


def test_bench_frame_zero(api, orc, launches):
    """The first frame of bench.py's sequence (640x480, 5 mm, Volume(65024, 8192)) allocates ~7 k
    blocks at once: hundreds of bucket contests. rounds=3 equals the app's three SetView calls;
    with enough rounds nothing is left pending, and that state is the oracle's fixed point."""
    import bench
    k = T.Projection.make(*scenes.APP_INTRINSICS)
    depth = bench.sphere_room_depth(k)
    pose = scenes.orbit_pose(0, bench.YAW_STEP)
    hf, df = frames(api, orc, depth, k, pose)

    hv, dv = make_pair(api, orc, bench.MAIN, bench.EXCESS, bench.VOXEL, bench.TRUNC)
    oracle_rounds(orc, hv, hf, 3)
    dv.set_view(df, rounds=3)
    assert_volume_equal(dv, hv, voxels=False)
    ctr = assert_same_requests(dv, hv)
    assert ctr[T.VK_CTR_ROUNDS] >= 2 and hv.visible_count > 5000
    pending_after_three = int(ctr[T.VK_CTR_UNSETTLED])

    hv2, dv2 = make_pair(api, orc, bench.MAIN, bench.EXCESS, bench.VOXEL, bench.TRUNC)
    dv2.set_view(df, rounds=16)
    ctr2 = dv2.read_counters()
    assert ctr2[T.VK_CTR_UNSETTLED] == 0 and ctr2[T.VK_CTR_ROUNDS] < 16
    calls = 0
    while True:                                   # the oracle's fixed point: a call that sees no request
        hv2.set_view(hf, orc.POLICY_MAXKEY)
        calls += 1
        if hv2.counters[T.VK_CTR_REQUESTS] == 0:
            break
    # the device stops after the first round that loses nothing; one more oracle call only confirms it
    assert calls in (ctr2[T.VK_CTR_ROUNDS], ctr2[T.VK_CTR_ROUNDS] + 1)
    assert_volume_equal(dv2, hv2, voxels=False)
    assert (pending_after_three == 0) == (int(ctr2[T.VK_CTR_ROUNDS]) <= 3)


@pytest.mark.parametrize("size", [(320, 240), (200, 150)])
def test_set_view_computes_the_frame_normals_on_the_way(api, orc, size):
    """vk_light_prep.normals_out: Frame::ComputeNormals is still due and SetView's request pass does it
    (the 5-tap stencil reads the depth tile that pass stages for the light mask): the normal image,
    the preparation and the integration that follows equal the separate calls bit for bit.
    200x150: image edges inside a workgroup's 64x4 pixels, ragged last tiles."""
    w, h = size
    k = T.Projection.make(272.0 * w / 320, 272.0 * w / 320, 155.6 * w / 320, 117.4 * h / 240)
    depth = scenes.sphere(2 * w, 2 * h)[::2, ::2].copy()
    depth[10:14, 20:40] = 0.0                      # holes: taps without a measurement
    color = scenes.checker_color(w, h, 0.1, 0.9)
    pose = scenes.tracer_test_pose()
    hf, df = frames(api, orc, depth, k, pose, color=color)
    hf.compute_normals()
    reference = api.Frame(depth, k, pose, color=color)
    reference.compute_normals()
    assert np.array_equal(reference.normals.cpu().numpy(), hf.normals, equal_nan=True)

    hv, dv = make_pair(api, orc, 2048, 8192, 0.01, 0.04)
    light = T.Light.make(2.0, (0.025, 0.08, 0.0))
    integ = api.LightIntegrator(dv)
    integ.light = light
    for i in range(2):
        if i == 1:
            df.normals.fill_(7.0)                   # stale content that must be replaced
        oracle_rounds(orc, hv, hf, 3)
        dv.set_view(df, rounds=3, compute_normals=True)
        sync()
        assert np.array_equal(df.normals.cpu().numpy(), hf.normals, equal_nan=True)
        if i == 1:
            assert integ._prep.valid == 1 and not integ._prep.normals_out
        orc.integrate_depth(hv, hf)
        orc.integrate_light_color(hv, hf, light, orc.light_frame_mask(hf, 0.2))
        integ.integrate(df)
        assert_volume_equal(dv, hv)


def test_normals_are_computed_also_when_the_preparation_cannot_ride(api, orc):
    """A frame without colour: the preparation does not ride along, the normals are computed by the
    launch of their own, and the request is cleared."""
    w, h = 320, 240
    depth = scenes.plane(w, h, 1.5)
    hf, df = frames(api, orc, depth, K_SMALL, scenes.tracer_test_pose())
    hf.compute_normals()
    hv, dv = make_pair(api, orc, 4096, 1024, 0.01, 0.04)
    prep = T.LightPrep()
    import torch
    mask = torch.zeros(w * h, dtype=torch.float32, device="cuda")
    records = torch.zeros(4 * w * h, dtype=torch.float32, device="cuda")
    prep.mask, prep.records, prep.capacity, prep.depth_threshold = mask.data_ptr(), records.data_ptr(), w * h, 0.2
    dv.attach_light_preparation(prep)
    oracle_rounds(orc, hv, hf, 1)
    dv.set_view(df, compute_normals=True)
    sync()
    assert np.array_equal(df.normals.cpu().numpy(), hf.normals, equal_nan=True)
    assert prep.valid == 0 and not prep.normals_out
    assert_volume_equal(dv, hv, voxels=False)


def test_the_posted_list_and_the_flag_scan_give_the_same_table(api, orc):
    """The handle pass works from the list of posted buckets when it holds them all, and from the
    request flags otherwise. Forced both ways on the same frames (vk_test_hooks.posted_capacity), at
    the boundary: a list that holds exactly the requests of the pass, and one that is one short.
    Odd table sizes: 1021 + 6001 entries (flag and visibility tails that are not a multiple of 4 / 16)."""
    w, h = 320, 240
    depth = scenes.sphere(2 * w, 2 * h)[::2, ::2].copy()
    poses = (scenes.tracer_test_pose(), scenes.yaw(3.0) * scenes.tracer_test_pose())
    # how many buckets the first pass posts to: the requests the first round sees
    hv0 = orc.HostVolume(1021, 6001, voxel_length=0.01, truncation_length=0.04)
    hf0 = orc.HostFrame(depth, K_SMALL, poses[0])
    hv0.set_view(hf0, orc.POLICY_MAXKEY)
    posted = int(hv0.counters[T.VK_CTR_REQUESTS])
    assert 256 < posted < 1021
    for capacity in (posted, posted - 1, 0, -1):                             # -1: the library's own capacity
        with api.test_hooks(posted_capacity=capacity):
            hv, dv = make_pair(api, orc, 1021, 6001, 0.01, 0.04)
            for pose in poses:
                hf, df = frames(api, orc, depth, K_SMALL, pose)
                oracle_rounds(orc, hv, hf, 3)
                dv.set_view(df, rounds=3)
                assert_volume_equal(dv, hv, voxels=False)
                assert_same_requests(dv, hv)
            assert hv.counters[T.VK_CTR_EXCESS_PTR] > hv.main + 100          # chains: EXCESS requests, new entries visible
            assert hv.counters[T.VK_CTR_DROPPED] == 0


@pytest.mark.parametrize("hooks", [{}, {"posted_capacity": 0}, {"set_view_unfused": 1}])
def test_a_request_pass_made_behind_the_raycast_gives_the_same_table(api, orc, hooks):
    """vk_trace_ahead_requests + vk_volume_set_view_rounds_ahead (Tracer.trace(frame, next_frame=) and the set_view that
    follows it) against the oracle's Trace and three SetView calls — on a table far too small for the scene (1024 buckets for
    ~2 000 blocks: every round loses requests, chains grow), with the handle pass working from the posted list, from the
    request flags (posted_capacity 0) and with handle and visibility as launches of their own (set_view_unfused): the pass made
    ahead must leave the later rounds exactly what the pass inside SetView leaves them. Every frame's raycast (depth, colour, normals) is compared as well."""
    w, h = 320, 240
    depth = scenes.sphere(2 * w, 2 * h)[::2, ::2].copy()
    poses = [scenes.yaw(2.0 * i) * scenes.tracer_test_pose() for i in range(4)]
    with api.test_hooks(**hooks):
        hv, dv = make_pair(api, orc, 1024, 16384, 0.01, 0.04)
        integ, tracer = api.DepthIntegrator(dv), api.Tracer(dv)
        pairs = [frames(api, orc, depth, K_SMALL, pose) for pose in poses]
        for i, pose in enumerate(poses):
            hf, df = pairs[i]
            before = int(dv.read_counters()[T.VK_CTR_ROUNDS])
            oracle_rounds(orc, hv, hf, 3)
            assert (dv.requests_ahead is not None and dv.requests_ahead.valid == 1) == (i > 0)
            dv.set_view(df, rounds=3)
            assert dv.requests_ahead is None or dv.requests_ahead.valid == 0
            assert_volume_equal(dv, hv, voxels=False)
            ctr = assert_same_requests(dv, hv)
            assert 1 <= int(ctr[T.VK_CTR_ROUNDS]) - before <= 3
            orc.integrate_depth(hv, hf)
            integ.integrate(df)
            odepth, ocolor, onormals, _ = orc.trace(hv, hf)
            out = api.Frame(np.zeros((h, w), np.float32), K_SMALL, pose)
            if i + 1 < len(poses):
                tracer.trace(out, next_frame=pairs[i + 1][1])     # the very frame the next set_view is handed
            else:
                tracer.trace(out)
            sync()
            assert np.array_equal(out.depth.cpu().numpy(), odepth)
            assert np.array_equal(out.color.cpu().numpy(), ocolor)
            assert np.array_equal(out.normals.cpu().numpy(), onormals, equal_nan=True)
        assert_volume_equal(dv, hv)
        assert hv.counters[T.VK_CTR_EXCESS_PTR] > hv.main + 500   # chains exercised
