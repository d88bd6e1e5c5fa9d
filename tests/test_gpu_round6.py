"""Round 6: what VERDICT r5 / ADVICE r5 named.

  * an announce WITHOUT the next frame's normals (Tracer.trace(out, next_frame=f) — next_needs_normals defaults to False)
    followed by Volume.set_view(f, compute_normals=True): the normals are still due and the light preparation that rode
    with the pass (made from the OLD normal image) must not be used;
"""
import ctypes as C
import os

import numpy as np
import pytest

import scenes
from test_gpu_parity import api, assert_volume_equal, sync  # noqa: F401
from vulcan_amd import vk_types as T

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rgbd_inputs(w=320, h=240):
    k = T.Projection.make(273.5, 273.5, 160.0, 120.0)
    y, x = np.mgrid[0:h, 0:w]
    depth = (1.4 + 0.1 * np.cos(5.0 * x / w) * np.sin(4.0 * y / h + 0.3)).astype(np.float32)
    return k, depth, scenes.checker_color(w, h, 0.1, 0.9)


# --------------------------------------------------- announce without the normals --

def test_announce_without_normals_then_set_view_computes_them(api, orc):
    """Trace(key, next) with next_needs_normals = false, then ComputeNormalsAndSetView(next). Until round 5 the
    normals were skipped on the record's validity alone and LightIntegrator shaded from whatever the normal image held.
    Now vk_requests_ahead.normals_made says whether they came with the pass; when not, they are computed in place, the
    announced SetView runs, and the preparation made from the old normals is void. The result is the unannounced
    sequence's, bit for bit, and the normal image is the oracle's."""
    import torch
    w, h = 320, 240
    k, depth, color = _rgbd_inputs(w, h)
    results = []
    for announce in (None, "without normals", "with normals"):
        vol = api.Volume(16384, 4096, voxel_length=0.01, truncation_length=0.05)
        integ, tracer = api.LightIntegrator(vol), api.Tracer(vol)
        integ.light = T.Light.make(2.0, (0.025, 0.08, 0.0))
        f0, f1 = api.Frame(depth, k, scenes.yaw(0.0), color=color), api.Frame(depth, k, scenes.yaw(1.0), color=color)
        # a normal image that holds something else (a recycled frame): what round 5's code shaded from
        f1.normals = torch.full((h, w, 3), 0.57735, dtype=torch.float32, device="cuda")
        out = api.Frame(torch.zeros((h, w), dtype=torch.float32, device="cuda"), k, scenes.yaw(0.0))
        vol.set_view(f0, rounds=3, compute_normals=True)
        integ.integrate(f0)
        if announce is None:
            tracer.trace(out)
        else:
            tracer.trace(out, next_frame=f1, next_needs_normals=(announce == "with normals"))
            assert vol.requests_ahead.valid == 1
            assert vol.requests_ahead.normals_made == (1 if announce == "with normals" else 0)
        vol.set_view(f1, rounds=3, compute_normals=True)
        assert vol.requests_ahead is None or vol.requests_ahead.valid == 0
        integ.integrate(f1)
        sync()
        results.append((vol.host_voxels().tobytes(), vol.host_entries().tobytes(), f1.normals.cpu().numpy()))
    want_normals = orc.compute_normals(depth, k)
    for voxels, entries, normals in results:
        assert np.array_equal(normals, want_normals, equal_nan=True)
        assert voxels == results[0][0] and entries == results[0][1]
    # and the shading did depend on the normals: integrating from the stale image gives another volume
    vol = api.Volume(16384, 4096, voxel_length=0.01, truncation_length=0.05)
    integ = api.LightIntegrator(vol)
    integ.light = T.Light.make(2.0, (0.025, 0.08, 0.0))
    f0, f1 = api.Frame(depth, k, scenes.yaw(0.0), color=color), api.Frame(depth, k, scenes.yaw(1.0), color=color)
    f1.normals = torch.full((h, w, 3), 0.57735, dtype=torch.float32, device="cuda")
    vol.set_view(f0, rounds=3, compute_normals=True)
    integ.integrate(f0)
    vol.set_view(f1, rounds=3)
    integ.integrate(f1)
    sync()
    assert vol.host_voxels().tobytes() != results[0][0], "the test's stale normal image does not change the shading"


def test_class_layer_records_normals_made_by_its_own_launch(api):
    """Without a LightIntegrator attached the announce cannot carry the normals; Tracer.trace(.., next_needs_normals=True) then
    computes them with a launch of its own IN FRONT of the announce, and the record says so (the class layer's flag)."""
    import torch
    w, h = 320, 240
    k, depth, _ = _rgbd_inputs(w, h)
    vol = api.Volume(16384, 4096, voxel_length=0.01, truncation_length=0.05)
    integ, tracer = api.DepthIntegrator(vol), api.Tracer(vol)
    f0, f1 = api.Frame(depth, k, scenes.yaw(0.0)), api.Frame(depth, k, scenes.yaw(1.0))
    out = api.Frame(torch.zeros((h, w), dtype=torch.float32, device="cuda"), k, scenes.yaw(0.0))
    vol.set_view(f0)
    integ.integrate(f0)
    tracer.trace(out, next_frame=f1, next_needs_normals=True)
    assert vol.requests_ahead.valid == 1 and vol.requests_ahead.normals_made == 1
    made = f1.normals.clone()
    vol.set_view(f1, compute_normals=True)                  # nothing is due: the frame is taken as announced
    sync()
    assert vol.requests_ahead.valid == 0 and torch.equal(f1.normals, made)


# -------------------------------------------- several sequences on one GPU --

@pytest.mark.parametrize("workload", ["rgbd", "rgbd-icp"])
def test_two_sequences_on_one_gpu_equal_each_alone(api, workload):
    """bench.MultiLoop: two independent sequences — a replica volume, a stream and a FrameLoop each —
    issued in lock step by one host thread, so that one sequence's integrate / raycast / request pass runs under the other's
    Gauss-Newton loop and raycast tail. Each sequence's volume, hash table, raycast images and (tracked) poses are the ones
    the same sequence gives when it runs alone, bit for bit. (Alone, bench.FrameLoop.step is held to the oracle by
    tests/test_gpu_configs.py.)"""
    import torch
    import bench
    frames = 6
    k = T.Projection.make(*scenes.APP_INTRINSICS)
    if workload == "rgbd-icp":
        room = bench.RoomSequence(frames + 8, k)
        sequences = [room.view(0, frames), room.view(8, frames)]
        pose_lists = [s.truth for s in sequences]
    else:
        sequences = None
        pose_lists = [[scenes.orbit_pose(i + 7 * j, bench.YAW_STEP) for i in range(frames)] for j in range(2)]
    sync()
    multi = bench.MultiLoop(workload, pose_lists, sequences)
    assert len({loop.stream.value for loop in multi.loops}) == 2, "the two sequences share a stream"
    for i in range(frames):
        multi.step(i)
    sync()

    def state(loop):
        vol = loop.vols[0]["vol"]
        ctr = torch.from_numpy(np.array([vol.read_counters()[c] for c in (T.VK_CTR_VISIBLE, T.VK_CTR_VOXEL_PTR, T.VK_CTR_DROPPED)]))
        return {"voxels": vol.voxels, "entries": vol.hash_entries, "visible / free pointer / dropped": ctr,
                "depth": loop.key.depth, "color": loop.key.color}

    for j in range(2):
        alone = bench.FrameLoop(workload, pose_lists[j], sequence=None if sequences is None else sequences[j])
        for i in range(frames):
            alone.step(i)
        sync()
        got, want = state(multi.loops[j]), state(alone)
        for name in want:
            assert torch.equal(got[name], want[name]), f"sequence {j}: {name} differs from the sequence run alone"
        if workload == "rgbd-icp":
            assert len(alone.tracked_poses) == frames
            assert [bytes(p) for p in multi.loops[j].tracked_poses] == [bytes(p) for p in alone.tracked_poses], f"sequence {j}: poses"
            assert multi.loops[j].gn_steps == alone.gn_steps
        assert int(alone.vols[0]["vol"].read_counters()[T.VK_CTR_VISIBLE]) > 1000
        del alone
        torch.cuda.empty_cache()


# -------------------------------------- a timed launch that never happened --

def test_timing_events_are_used_up_by_a_call_that_fails(api):
    """vk_integrate_time_next arms a pair of events for the next integrate launch of the thread. If the next
    vk_integrate_* call returns before it launches (an argument error), the pair used to stay armed and was recorded by a
    later, unrelated launch. Now the pair is used up by the CALL."""
    lib = api.lib()
    k, depth, color = _rgbd_inputs()
    vol = api.Volume(16384, 4096, voxel_length=0.01, truncation_length=0.05)
    integ = api.DepthIntegrator(vol)
    f = api.Frame(depth, k, scenes.yaw(0.0))
    vol.set_view(f)
    pair = []
    for _ in range(2):
        e = C.c_void_p()
        api.check(lib.vk_event_create(C.byref(e)), "vk_event_create")
        pair.append(e)
    api.check(lib.vk_integrate_time_next(pair[0], pair[1]), "vk_integrate_time_next")
    assert lib.vk_integrate_depth(None, None, None, None) == -1          # fails before its launch
    integ.integrate(f)                                                    # an unrelated launch: must not record the pair
    sync()
    ms = C.c_float(-1.0)
    assert lib.vk_event_elapsed_ms(pair[0], pair[1], C.byref(ms)) != 0, "the stale pair was recorded by a later launch"


# --------------------------------- the next Track's pyramid behind the raycast --

def test_pyramid_made_behind_the_raycast_equals_the_pyramid_launch(api, orc):
    """vk_trace_ahead_pyramid: the raycast's launch also makes the NEXT Track's pyramid — the next input
    frame's normal image and half-resolution level, the raycast's own normal image and half-resolution level (behind the
    raycast's row counters) — and vk_icp_pyramid_track_built then launches the two loops only. Against Tracer.trace +
    PyramidTracker.track (the pyramid launch): the same raycast, the same four images, the same pyramid buffer, the same pose,
    bit for bit; the normal images are the oracle's; the record serves once and names its images."""
    import torch
    import bench
    lib = api.lib()
    k = T.Projection.make(*scenes.APP_INTRINSICS)
    poses = [scenes.room_pose(30 + i) for i in range(3)]
    inputs = [scenes.room_frame(k, p, bench.W, bench.H, light=bench.LIGHT) for p in poses]
    vol = api.Volume(bench.MAIN, bench.EXCESS, voxel_length=bench.VOXEL, truncation_length=bench.TRUNC)
    integ, tracer = api.LightIntegrator(vol), api.Tracer(vol)
    integ.light = T.Light.make(*bench.LIGHT)
    f0 = api.Frame(inputs[0][0], k, poses[0], color=inputs[0][1])
    vol.set_view(f0, rounds=3, compute_normals=True)
    integ.integrate(f0)
    sync()

    def fresh(depth):
        return api.Frame(depth, k, poses[0], normals=torch.full((bench.H, bench.W, 3), -7.0, dtype=torch.float32, device="cuda"))

    # ---- the reference form: Trace, then Track (its pyramid launch computes the frame's normals)
    key_a = api.Frame(torch.zeros((bench.H, bench.W), dtype=torch.float32, device="cuda"), k, poses[0])
    tracer.trace(key_a)
    next_a = fresh(inputs[1][0])
    track_a = api.PyramidTracker()
    track_a.keyframe = key_a
    pose_a = track_a.track(next_a, compute_normals=True)
    sync()

    # ---- the riding form, through the C ABI
    key_b = api.Frame(torch.zeros((bench.H, bench.W), dtype=torch.float32, device="cuda"), k, poses[0],
                      color=torch.zeros((bench.H, bench.W, 3), dtype=torch.float32, device="cuda"),
                      normals=torch.full((bench.H, bench.W, 3), -7.0, dtype=torch.float32, device="cuda"))
    next_b = fresh(inputs[1][0])
    track_b = api.PyramidTracker()
    t = track_b.tracker
    n = int(lib.vk_icp_pyramid_floats(bench.W, bench.H, bench.W, bench.H))
    pyramid = torch.full((n,), -5.0, dtype=torch.float32, device="cuda")
    built = T.PyramidAhead()
    next_view, key_view = t._view(next_b), t._view(key_b)
    vb = tracer.view_bounds
    vb.valid = 0
    api.check(lib.vk_trace_ahead_pyramid(C.byref(vol.desc()), C.byref(key_b.desc()), C.byref(vb), key_b.depth.data_ptr(),
                                         key_b.color.data_ptr(), key_b.normals.data_ptr(), C.byref(next_view), pyramid.data_ptr(),
                                         C.byref(built), api.stream()), "vk_trace_ahead_pyramid")
    sync()
    assert built.valid == 1 and built.key_depths == key_b.depth.data_ptr() and built.frame_normals == next_b.normals.data_ptr()
    assert torch.equal(key_b.depth, key_a.depth) and torch.equal(key_b.color, key_a.color)
    want_key_normals = orc.compute_normals(key_a.depth.cpu().numpy(), k)
    assert np.array_equal(key_b.normals.cpu().numpy(), want_key_normals, equal_nan=True)
    assert np.array_equal(next_b.normals.cpu().numpy(), orc.compute_normals(inputs[1][0], k), equal_nan=True)
    assert torch.equal(next_b.normals, next_a.normals)
    assert np.array_equal(pyramid.cpu().numpy(), track_a._pyramid.cpu().numpy()[:n], equal_nan=True), "the half-resolution level differs"
    poll = t._poll()
    t.state.zero_()
    start = poses[0]
    api.check(lib.vk_icp_pyramid_track_built(C.byref(key_view), C.byref(key_b.depth_to_world), C.byref(next_view), t.pose.data_ptr(),
                                             C.byref(start), 1 | 2, C.byref(built), pyramid.data_ptr(),
                                             t._workspace(next_b).data_ptr(), t.system.data_ptr(), t.state.data_ptr(),
                                             t.update.data_ptr(), None, None, poll, api.stream()), "vk_icp_pyramid_track_built")
    pose_b = t._wait_pose()
    assert built.valid == 0                                                 # served once
    assert bytes(pose_b) == bytes(pose_a)
    # a record for OTHER images is not used: the call is then vk_icp_pyramid_track_frame, with its pyramid launch
    built.valid = 1
    built.frame_depths = key_b.depth.data_ptr()
    pyramid.fill_(-5.0)
    next_b.normals.fill_(-7.0)
    t.state.zero_()
    api.check(lib.vk_icp_pyramid_track_built(C.byref(key_view), C.byref(key_b.depth_to_world), C.byref(next_view), t.pose.data_ptr(),
                                             C.byref(start), 1, C.byref(built), pyramid.data_ptr(),
                                             t._workspace(next_b).data_ptr(), t.system.data_ptr(), t.state.data_ptr(),
                                             t.update.data_ptr(), None, None, poll, api.stream()), "vk_icp_pyramid_track_built")
    assert bytes(t._wait_pose()) == bytes(pose_a) and built.valid == 0
    assert torch.equal(next_b.normals, next_a.normals)


def test_tracked_loop_with_the_pyramid_behind_the_raycast_equals_the_loop_without(api, monkeypatch):
    """bench.FrameLoop('rgbd-icp') — the step bench.py times — with the pyramid riding behind the raycast (VK_BENCH_PYRAMID_AHEAD=1:
    built and measured to lose, so off in reported runs) and with the pyramid launch in front of the loops: six frames, every tracked pose, the volume, the table
    and the raycast images bit for bit."""
    import torch
    import bench
    frames = 6
    k = T.Projection.make(*scenes.APP_INTRINSICS)
    room = bench.RoomSequence(frames, k)
    sync()
    states = []
    for ahead in (True, False):
        monkeypatch.setattr(bench, "PYRAMID_AHEAD", ahead)
        monkeypatch.setattr(bench, "SET_VIEW_AT_DEVICE_POSE", 0)
        loop = bench.FrameLoop("rgbd-icp", room.truth, sequence=room)
        assert (loop.built is not None) == ahead
        for i in range(frames):
            loop.step(i)
            if ahead and 0 < i < frames - 1:
                assert loop.built.valid == 1, "the raycast did not carry the next Track's pyramid"
        sync()
        vol = loop.vols[0]["vol"]
        states.append(([bytes(p) for p in loop.tracked_poses], list(loop.gn_steps), vol.voxels.clone(), vol.hash_entries.clone(),
                       loop.key.depth.clone(), loop.key.color.clone(), loop.key.normals.clone(), loop.frame.normals.clone()))
        del loop
    a, b = states
    assert a[0] == b[0], "tracked poses differ"
    assert a[1] == b[1]
    for x, y, name in zip(a[2:], b[2:], ("voxels", "table", "key depth", "key colour", "key normals", "frame normals")):
        if name in ("key normals", "frame normals"):
            assert np.array_equal(x.cpu().numpy(), y.cpu().numpy(), equal_nan=True), name
        else:
            assert torch.equal(x, y), name


# --------------------------------- SetView's request pass at the pose on the device --

def test_tracked_loop_with_set_view_at_the_device_pose_equals_the_loop_without(api, monkeypatch):
    """bench.FrameLoop('rgbd-icp') with SetView(i) enqueued behind Track(i) at the pose the tracker leaves on the device — the
    whole call (vk_volume_set_view_at_device_pose, bench.py's default) or its request pass only
    (vk_volume_requests_at_device_pose) — and with the host's round trip in front of SetView
    (VK_BENCH_SET_VIEW_AT_DEVICE_POSE=0, the form the oracle's closed loop is held against in tests/test_gpu_closed_loop.py):
    six frames, every tracked pose, the volume, the table, the visibility bytes, the visible list's size and the raycast and
    light-preparation images, bit for bit."""
    import torch
    import bench
    frames = 6
    k = T.Projection.make(*scenes.APP_INTRINSICS)
    room = bench.RoomSequence(frames, k)
    sync()
    states = []
    for mode in (2, 1, 0):
        monkeypatch.setattr(bench, "SET_VIEW_AT_DEVICE_POSE", mode)
        monkeypatch.setattr(bench, "PYRAMID_AHEAD", False)
        loop = bench.FrameLoop("rgbd-icp", room.truth, sequence=room)
        assert (loop.early is not None) == (mode == 1) and loop.set_view_early == (mode == 2)
        for i in range(frames):
            loop.begin(i)
            if mode == 1 and i > 0:
                assert loop.early.valid == 1 and loop.early.pose_on_device == 1, "the request pass was not made behind the Track"
            if mode == 2 and i > 0:
                assert loop.set_view_done
            loop.finish(i)
            if mode == 1:
                assert loop.early.valid == 0                   # SetView used the record
        sync()
        vol = loop.vols[0]["vol"]
        ctr = vol.read_counters()
        states.append(([bytes(p) for p in loop.tracked_poses], list(loop.gn_steps), int(ctr[T.VK_CTR_VISIBLE]), int(ctr[T.VK_CTR_VOXEL_PTR]),
                       vol.voxels.clone(), vol.hash_entries.clone(), vol.block_visibility.clone(), loop.key.depth.clone(),
                       loop.key.color.clone(), loop.mask.clone(), loop.records.clone()))
        del loop
    for a in states[:2]:
        b = states[2]
        assert a[0] == b[0], "tracked poses differ"
        assert a[1:4] == b[1:4]
        for x, y, name in zip(a[4:], b[4:], ("voxels", "table", "visibility", "key depth", "key colour", "mask", "records")):
            assert np.array_equal(x.cpu().numpy(), y.cpu().numpy(), equal_nan=True), name


def test_requests_at_the_device_pose_record_and_an_aborted_track(api):
    """The record of vk_volume_requests_at_device_pose: it announces the frame (another frame is refused by SetView, a second
    pass by the call itself), its pose is taken on trust (pose_on_device), and after a Track that ABORTED — the device pose
    is then still the start pose — vk_requests_ahead_cancel completes the SetView at the start pose: the state of a plain
    SetView with the untracked frame, bit for bit."""
    import torch
    lib = api.lib()
    w, h = 320, 240
    k, depth, color = _rgbd_inputs(w, h)
    start = scenes.yaw(1.0)

    def volume():
        vol = api.Volume(16384, 4096, voxel_length=0.01, truncation_length=0.05)
        f0 = api.Frame(depth, k, scenes.yaw(0.0), color=color)
        f0.compute_normals()
        vol.set_view(f0, rounds=3)
        api.DepthIntegrator(vol).integrate(f0)
        return vol

    # the reference state: SetView of the frame at the start pose
    want = volume()
    f = api.Frame(depth, k, start, color=color)
    f.compute_normals()
    want.set_view(f, rounds=3)
    sync()
    # the early pass, reading the pose from device memory that holds the start pose (what an aborted Track leaves)
    got = volume()
    pose_dev = torch.from_numpy(np.frombuffer(bytes(start), dtype=np.uint8).copy()).cuda()
    record = T.RequestsAhead()
    g = api.Frame(depth, k, start, color=color)
    g.compute_normals()
    gd = g.desc()
    api.check(lib.vk_volume_requests_at_device_pose(C.byref(got.desc()), C.byref(gd), pose_dev.data_ptr(), None, C.byref(record),
                                                    api.stream()), "vk_volume_requests_at_device_pose")
    assert record.valid == 1 and record.pose_on_device == 1
    assert lib.vk_volume_requests_at_device_pose(C.byref(got.desc()), C.byref(gd), pose_dev.data_ptr(), None, C.byref(record),
                                                 api.stream()) == -1                  # a second pass on top: refused
    other = api.Frame(depth, k, scenes.yaw(5.0), color=color).desc()
    assert lib.vk_volume_set_view_rounds_ahead(C.byref(got.desc()), C.byref(other), None, 3, C.byref(record), api.stream()) == -1
    assert record.valid == 1                                                          # kept: cancel is the way out
    api.check(lib.vk_requests_ahead_cancel(C.byref(got.desc()), C.byref(record), 3, api.stream()), "vk_requests_ahead_cancel")
    sync()
    assert record.valid == 0
    for name in ("hash_entries", "block_visibility", "allocation_types"):
        assert torch.equal(getattr(got, name), getattr(want, name)), name
    cg, cw = got.read_counters(), want.read_counters()
    for c in (T.VK_CTR_VISIBLE, T.VK_CTR_VOXEL_PTR, T.VK_CTR_EXCESS_PTR, T.VK_CTR_DROPPED):
        assert cg[c] == cw[c]
    n = int(cw[T.VK_CTR_VISIBLE])
    # (the visible list's ORDER is the compaction's; its content is the state)
    assert n > 100 and torch.equal(torch.sort(got.visible_blocks[:n]).values, torch.sort(want.visible_blocks[:n]).values)
    # the same pass with a pose the HOST has (trusted, pose_on_device): SetView takes the record for the frame
    again = volume()
    rec2 = T.RequestsAhead()
    api.check(lib.vk_volume_requests_at_device_pose(C.byref(again.desc()), C.byref(gd), pose_dev.data_ptr(), None, C.byref(rec2),
                                                    api.stream()), "vk_volume_requests_at_device_pose")
    api.check(lib.vk_volume_set_view_rounds_ahead(C.byref(again.desc()), C.byref(gd), None, 3, C.byref(rec2), api.stream()),
              "vk_volume_set_view_rounds_ahead")
    sync()
    assert rec2.valid == 0 and torch.equal(again.hash_entries, want.hash_entries)
    assert again.read_counters()[T.VK_CTR_VISIBLE] == cw[T.VK_CTR_VISIBLE]


@pytest.mark.parametrize("with_prep", [False, True], ids=["no preparation", "light preparation rides"])
def test_set_view_at_the_device_pose_equals_set_view_rounds(api, with_prep):
    """vk_volume_set_view_at_device_pose against vk_volume_set_view_rounds with the same pose as a launch argument, from an EMPTY
    volume (7 k blocks requested at once: buckets are contested, the later rounds run inside the handle + visibility launch) and
    once more from the state that leaves: table, visibility bytes, request flags, the visible set, the pool pointers, the rounds
    run — and, with a LightIntegrator's preparation riding, its mask and records — bit for bit."""
    import torch
    import bench
    lib = api.lib()
    k = T.Projection.make(*scenes.APP_INTRINSICS)
    depth = bench.sphere_room_depth(k)
    color = scenes.checker_color(bench.W, bench.H, 0.1, 0.9)
    poses = [scenes.orbit_pose(0, bench.YAW_STEP), scenes.orbit_pose(9, bench.YAW_STEP)]
    states = []
    for at_device in (False, True):
        vol = api.Volume(bench.MAIN, bench.EXCESS, voxel_length=bench.VOXEL, truncation_length=bench.TRUNC)
        frame = api.Frame(depth, k, poses[0], color=color)
        frame.compute_normals()
        prep = pprep = None
        if with_prep:
            mask = torch.full((bench.H, bench.W), -1.0, dtype=torch.float32, device="cuda")
            records = torch.full((bench.H, bench.W, 4), -1.0, dtype=torch.float32, device="cuda")
            prep = T.LightPrep()
            prep.depth_threshold, prep.mask, prep.records, prep.capacity = 0.2, mask.data_ptr(), records.data_ptr(), bench.W * bench.H
            pprep = C.byref(prep)
        rounds = []
        for pose in poses:
            frame.depth_to_world = pose
            frame.touch()
            fd = frame.desc()
            if at_device:
                fd.depth_to_world = T.Transform.identity()             # ignored: the pose comes from the device
                pose_dev = torch.from_numpy(np.frombuffer(bytes(pose), dtype=np.uint8).copy()).cuda()
                api.check(lib.vk_volume_set_view_at_device_pose(C.byref(vol.desc()), C.byref(fd), pose_dev.data_ptr(), pprep, 3, api.stream()),
                          "vk_volume_set_view_at_device_pose")
            else:
                api.check(lib.vk_volume_set_view_rounds(C.byref(vol.desc()), C.byref(fd), pprep, 3, api.stream()), "vk_volume_set_view_rounds")
            sync()
            if with_prep:
                assert prep.valid == 1
            rounds.append(int(vol.read_counters()[T.VK_CTR_ROUNDS]))
        ctr = vol.read_counters()
        n = int(ctr[T.VK_CTR_VISIBLE])
        state = {"table": vol.hash_entries.clone(), "visibility": vol.block_visibility.clone(), "request flags": vol.allocation_types.clone(),
                 "request words": vol.allocation_blocks.clone(), "visible set": torch.sort(vol.visible_blocks[:n]).values.clone(),
                 "counters": torch.tensor([int(ctr[c]) for c in (T.VK_CTR_VISIBLE, T.VK_CTR_VOXEL_PTR, T.VK_CTR_EXCESS_PTR, T.VK_CTR_DROPPED,
                                                                 T.VK_CTR_REQUESTS)] + rounds)}
        if with_prep:
            state["mask"], state["records"] = mask.clone(), records.clone()
        states.append(state)
        assert n > 5000 and rounds[0] > 1, "the first SetView from an empty volume runs more than one round"
    for name in states[0]:
        assert torch.equal(states[0][name], states[1][name]), name


def test_python_track_with_set_view_equals_the_two_calls(api):
    """api.PyramidTracker.track(frame, set_view_of=volume, rounds=3) — the Python mirror of
    PyramidTracker<DepthTracker>::ComputeNormalsTrackAndSetView — against track() followed by Volume.set_view(): the same pose,
    table, visibility bytes and visible set; and under a forced loop abort the call still ends with the frame's SetView."""
    import torch
    import bench
    k = T.Projection.make(*scenes.APP_INTRINSICS)
    poses = [scenes.room_pose(30), scenes.room_pose(31)]
    inputs = [scenes.room_frame(k, p, bench.W, bench.H, light=bench.LIGHT) for p in poses]
    states = []
    for combined in (False, True, "aborted"):
        vol = api.Volume(bench.MAIN, bench.EXCESS, voxel_length=bench.VOXEL, truncation_length=bench.TRUNC)
        integ, tracer = api.LightIntegrator(vol), api.Tracer(vol)
        integ.light = T.Light.make(*bench.LIGHT)
        f0 = api.Frame(inputs[0][0], k, poses[0], color=inputs[0][1])
        vol.set_view(f0, rounds=3, compute_normals=True)
        integ.integrate(f0)
        key = api.Frame(torch.zeros((bench.H, bench.W), dtype=torch.float32, device="cuda"), k, poses[0])
        tracer.trace(key)
        f1 = api.Frame(inputs[1][0], k, poses[0], color=inputs[1][1])
        tracker = api.PyramidTracker()
        tracker.keyframe = key
        if combined is False:
            pose = tracker.track(f1, compute_normals=True)
            vol.set_view(f1, rounds=3)
        elif combined is True:
            pose = tracker.track(f1, compute_normals=True, set_view_of=vol, rounds=3)
            assert tracker._set_view_done, "SetView was not enqueued behind the Track"
        else:
            with api.test_hooks(force_loop_abort=1):
                pose = tracker.track(f1, compute_normals=True, set_view_of=vol, rounds=3)
            assert not tracker._set_view_done
        integ.integrate(f1)
        sync()
        ctr = vol.read_counters()
        n = int(ctr[T.VK_CTR_VISIBLE])
        states.append((bytes(pose), vol.hash_entries.clone(), vol.block_visibility.clone(), torch.sort(vol.visible_blocks[:n]).values.clone(), n))
    a, b, c = states
    assert a[0] == b[0] and a[4] == b[4] > 1000
    for x, y in zip(a[1:4], b[1:4]):
        assert torch.equal(x, y)
    # aborted: the staged Track gives the oracle-level pose (within the loop forms' tolerance) and the frame's own SetView ran
    # after an extra SetView at the start pose: at least the blocks of the plain sequence are visible
    pa, pc = T.Transform.from_buffer_copy(a[0]), T.Transform.from_buffer_copy(c[0])
    assert np.abs(pa.matrix() - pc.matrix()).max() < 2e-5
    assert c[4] >= a[4]
