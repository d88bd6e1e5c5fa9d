"""A one-launch Gauss-Newton loop that cannot get all of its workgroups onto the device together
gives up (VK_TRACK_ABORTED, vk.h) — two processes on one GPU, or a foreign kernel that holds
compute units while the loop spins. The hosts' way out: the same Track again from the start pose
on the launch-per-stage path, which waits for nobody. The abort is forced here through the test
aid vk_test_hooks.force_loop_abort (the loop kernels end at once, as after the exchange timeout); the
result must equal the launch-per-stage path's bit for bit. Run once; never looped.
"""
import os
import subprocess

import numpy as np
import pytest

import color_scenes as cs
import scenes
from test_gpu_parity import api, sync  # noqa: F401  (fixture)
from vulcan_amd import vk_types as T

pytestmark = pytest.mark.gpu


def forced_abort():
    from vulcan_amd import api as a
    return a.test_hooks(force_loop_abort=1)


def depth_pair(api):
    w, h = 320, 240
    k = T.Projection.make(*(0.5 * np.float32(v) for v in scenes.APP_INTRINSICS))
    key = api.Frame(scenes.room_depth(k, scenes.room_pose(40), w, h), k, scenes.room_pose(40))
    key.compute_normals()
    frame = api.Frame(scenes.room_depth(k, scenes.room_pose(42), w, h), k, scenes.room_pose(40))
    frame.compute_normals()
    return key, frame


def test_depth_tracker_and_pyramid_fall_back_to_the_staged_path(api):
    key, frame = depth_pair(api)
    start = frame.depth_to_world

    def staged(make, track):
        t = make()
        t.reduce_hook = lambda system: None          # a hook that changes nothing: one launch per stage
        frame.depth_to_world = start
        return track(t), t

    def fallen_back(make, track):
        t = make()
        frame.depth_to_world = start
        with forced_abort():
            return track(t), t

    # DepthTracker::Track
    def make_depth():
        t = api.DepthTracker()
        t.keyframe = key
        return t
    want, _ = staged(make_depth, lambda t: t.track(frame))
    got, tracker = fallen_back(make_depth, lambda t: t.track(frame))
    sync()
    assert bytes(got) == bytes(want) and bytes(frame.depth_to_world) == bytes(want)
    assert int(tracker.state.cpu()[1]) != T.VK_TRACK_ABORTED and int(tracker.state.cpu()[0]) >= 1
    assert np.abs(got.matrix() - start.matrix()).max() > 1e-3           # it did track

    # without the host's fallback the call reports the abort instead of a pose
    t = make_depth()
    frame.depth_to_world = start
    with forced_abort():
        with pytest.raises(api.TrackAborted):
            t._track(frame)
    assert int(t.state.cpu()[1]) == T.VK_TRACK_ABORTED
    assert bytes(frame.depth_to_world) == bytes(start)

    # PyramidTracker<DepthTracker>::Track: an abort of the HALF level must not be lost by the full one
    def make_pyramid():
        p = api.PyramidTracker()
        p.keyframe = key
        return p

    def pyramid_staged(p):
        p.tracker.reduce_hook = lambda system: None
        return p.track(frame)
    frame.depth_to_world = start
    want = pyramid_staged(make_pyramid())
    got, _ = fallen_back(make_pyramid, lambda p: p.track(frame))
    sync()
    assert bytes(got) == bytes(want)
    p = make_pyramid()
    frame.depth_to_world = start
    with forced_abort():
        with pytest.raises(api.TrackAborted):
            p._track_depth_once(frame)
    assert int(p.tracker.state.cpu()[1]) == T.VK_TRACK_ABORTED


def test_light_tracker_falls_back_to_the_staged_path(api, orc):
    k, light = cs.projection(), cs.test_light()
    kd, kc = cs.plane_frame(cs.light_keyframe_pose(), False, light)
    fd, fc = cs.plane_frame(cs.light_frame_pose(), True, light)
    hk = orc.HostFrame(kd, k, cs.light_keyframe_pose(), color=kc)
    hf = orc.HostFrame(fd, k, cs.light_frame_pose(), color=fc)
    hk.compute_normals()
    hf.compute_normals()
    dk = api.Frame(kd, k, cs.light_keyframe_pose(), color=kc, normals=hk.normals)
    start = T.Transform.translate(0.01, -0.005, 0.008) * cs.light_frame_pose()
    df = api.Frame(fd, k, start, color=fc, normals=hf.normals)

    def make():
        t = api.LightTracker()
        t.keyframe = dk
        t.light = light
        return t
    t = make()
    t.reduce_hook = lambda system: None
    want = t.track(df)
    df.depth_to_world = start
    t = make()
    with forced_abort():
        got = t.track(df)
    sync()
    assert bytes(got) == bytes(want)
    assert np.abs(got.matrix() - start.matrix()).max() > 1e-3


def test_cpp_class_layer_falls_back(api):
    """The C++ trackers (Tracker::Track, DepthTracker::TrackPyramid, ColorTracker::TrackCoarseToFine)
    under a forced abort: their re-authored Track tests pass on the fallback path."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "vulcan_amd", "host", "bin", "host_tests")
    out = subprocess.run([exe, "Track", "--force-loop-abort"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                         timeout=600).stdout
    assert " 0 failed" in out, out
    for name in ("DepthTracker.Track", "PyramidTracker.Track", "ColorTracker.Track", "LightTracker.Track"):
        assert f"[  OK  ] {name}" in out, out
