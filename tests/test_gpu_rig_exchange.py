"""vk_icp_track_rig on one GPU: a rig of ONE rank (loopback communicator, its own area only) must give
vk_icp_track's bits — the publish / gather / rank-ordered sum sit between the local sums and the
solve of every step — over many Tracks in a row (sequence numbers, step and Track parities). More
ranks have never run on hardware; the protocol itself is rehearsed on host threads
(tests/test_rig_protocol.py)."""
import numpy as np
import pytest

import scenes
from test_gpu_parity import api, sync  # noqa: F401
from vulcan_amd import vk_types as T

pytestmark = pytest.mark.gpu


def test_rig_of_one_equals_the_plain_track(api):
    from vulcan_amd import comm
    w, h = 320, 240
    k = T.Projection.make(*(0.5 * np.float32(v) for v in scenes.APP_INTRINSICS))
    key = api.Frame(scenes.room_depth(k, scenes.room_pose(40), w, h), k, scenes.room_pose(40))
    key.compute_normals()
    c = comm.Communicator(None, 0, 1)
    x = c.attach_exchange()
    assert x.world == 1 and x.rank == 0 and x.sequence == 1 and x.areas[0]
    plain, rig = api.DepthTracker(), api.DepthTracker()
    plain.keyframe = rig.keyframe = key
    for i, (iterations, translation) in enumerate([(20, True), (1, True), (3, False), (20, True), (2, True), (1, True), (7, True)]):
        frame = api.Frame(scenes.room_depth(k, scenes.room_pose(41 + i), w, h), k, scenes.room_pose(40))
        frame.compute_normals()
        start = frame.depth_to_world
        for t in (plain, rig):
            t.max_iterations, t.translation_enabled = iterations, translation
        want = plain.track(frame)
        frame.depth_to_world = start
        got = c.track_rig(rig, frame)
        sync()
        assert bytes(got) == bytes(want), i
        assert np.array_equal(rig.system.cpu().numpy(), plain.system.cpu().numpy())
        assert np.array_equal(rig.state.cpu().numpy(), plain.state.cpu().numpy())
        assert c.exchange.sequence == i + 2
    c.close()
