"""The Gauss-Newton exchange's launch tags repeat every 2^22 - 1 loop launches (about ten minutes of tracking): VERDICT r5
weak #7 / next #3. A workspace that sat idle for a whole period — a second tracker, a rig peer, a pyramid level not run after
an abort — comes back to a launch whose tag equals the tag of the words it still holds. The library makes that a non-event
(vk_runtime.hip vk_loop_epoch_begin: an exchange area is cleared in front of a launch when it is new, 2^21 launches old, grown,
or was written by a launch-per-stage loop); vk_test_hooks_loop_count puts the count right in front of the repeat.

  * forty Tracks on two PyramidTrackers used alternately, across the wrap, one of them idle for exactly one period in
    between: every pose against the oracle's (2e-5) and, bit for bit, against the pose the same start gave long before the wrap;
  * the adversarial case, deterministic: the idle workspace holds, in EVERY slot, a word that carries exactly the tag of the
    launch that comes next (what a launch one period earlier leaves behind, with values that would wreck the sums) — the pose
    must be the fresh tracker's, bit for bit. Built with -DVK_LOOP_EPOCH_UNGUARDED (no clears) this test fails
    (tools/epoch_wrap_proof.sh, profiles/r06_epoch_wrap.txt): it bites.

ref: src/tracker.cpp:53-63 (a loop that is simply correct for ever)."""
import ctypes as C

import numpy as np
import pytest

import scenes
from test_gpu_parity import api, frames, sync  # noqa: F401
from vulcan_amd import vk_types as T

pytestmark = pytest.mark.gpu

PERIOD = (1 << 22) - 1


def loop_count(api, set_to=None):
    """(count, clears) of the library image under test; `set_to` replaces the count first"""
    now, clears = C.c_uint64(), C.c_uint64()
    arg = None if set_to is None else C.byref(C.c_uint64(set_to))
    api.check(api.lib().vk_test_hooks_loop_count(arg, C.byref(now), C.byref(clears)), "vk_test_hooks_loop_count")
    return now.value, clears.value


def epoch_of(count):
    return 1 + count % PERIOD


STARTS = [T.Transform.translate(0.002, -0.001, 0.003) * T.Transform.rotate(0.999995, 0.002, -0.0015, 0.001),
          T.Transform.translate(-0.003, 0.002, 0.001) * T.Transform.rotate(0.999995, -0.001, 0.002, -0.0015),
          T.Transform.translate(0.001, 0.003, -0.002) * T.Transform.rotate(0.999997, 0.0015, 0.001, 0.001),
          T.Transform.translate(-0.001, -0.002, -0.003) * T.Transform.rotate(0.999996, -0.002, -0.001, 0.0015)]


def curved_depth(w, h):
    y, x = np.mgrid[0:h, 0:w]
    return (1.0 + 0.05 * np.cos(3.0 * x / w) * np.sin(2.0 * y / h)).astype(np.float32)


def _scene(api, orc, w=320, h=240):
    s = w / 640.0
    k = T.Projection.make(547.0 * s, 547.0 * s, 320.0 * s, 240.0 * s)
    depth = curved_depth(w, h)
    hk, dk = frames(api, orc, depth, k, T.Transform.identity())
    hk.compute_normals()
    dk.compute_normals()
    return k, depth, hk, dk


def test_forty_tracks_across_the_wrap_on_two_trackers(api, orc):
    k, depth, hk, dk = _scene(api, orc)
    want = []
    for start in STARTS:
        hf = orc.HostFrame(depth, k, start, normals=hk.normals)
        pose, _ = orc.pyramid_track(hk, hf)
        want.append(pose)

    def track(tracker, j):
        f = api.Frame(dk.depth, k, STARTS[j], normals=dk.normals)
        got = tracker.track(f)
        assert int(tracker.tracker.state.cpu()[1]) in (0, 1), "the loop kernel aborted"
        return got

    # long before the wrap: the poses a fresh tracker gives (the exchange is fixed-order: run-to-run identical bits)
    count0, _ = loop_count(api)
    fresh = api.PyramidTracker()
    fresh.keyframe = dk
    ref = [bytes(track(fresh, j)) for j in range(len(STARTS))]
    for j, r in enumerate(ref):
        got = T.Transform.from_buffer_copy(r)
        np.testing.assert_allclose(got.matrix(), want[j].matrix(), atol=2e-5)

    a, b = api.PyramidTracker(), api.PyramidTracker()
    a.keyframe = b.keyframe = dk
    # 21 loop launches in front of a wrap of the 22-bit epoch
    base = (count0 // PERIOD + 2) * PERIOD - 21
    loop_count(api, set_to=base)
    done = 0
    b_first, _ = loop_count(api)
    assert epoch_of(b_first) == PERIOD - 20
    assert bytes(track(b, 0)) == ref[0]                       # B's one Track in front of the wrap: two launches, then idle
    done += 1
    for i in range(14):                                        # A across the wrap (28 launches: the epoch passes 2^22 - 1 -> 1)
        assert bytes(track(a, (i + 1) % 4)) == ref[(i + 1) % 4], f"A, Track {i}"
        done += 1
    now, clears_before = loop_count(api)
    assert now == b_first + 2 + 28 and epoch_of(now) < 16, "the wrap was not crossed"
    # A goes on for the rest of the period (counted, not run): B's next launches get EXACTLY the tags of its last ones
    loop_count(api, set_to=b_first + PERIOD)
    assert epoch_of(b_first + PERIOD) == epoch_of(b_first)
    assert bytes(track(b, 1)) == ref[1], "the tracker that sat idle for a whole period"
    done += 1
    _, clears_after = loop_count(api)
    assert clears_after > clears_before, "an area 2^22 - 1 launches old was launched on without a clear"
    for i in range(12):                                        # and on, alternately
        assert bytes(track(a, (i + 2) % 4)) == ref[(i + 2) % 4]
        assert bytes(track(b, (i + 3) % 4)) == ref[(i + 3) % 4]
        done += 2
        if i == 0:
            _, clears_steady = loop_count(api)                 # (A's area is a period old as well: cleared at its first Track)
            assert clears_steady > clears_after
    assert done == 40
    _, clears_end = loop_count(api)
    assert clears_end == clears_steady, "areas in steady use are not cleared again"


@pytest.mark.parametrize("tracker_kind", ["depth", "pyramid"])
def test_an_idle_area_full_of_the_next_launchs_tag_is_cleared_first(api, orc, tracker_kind):
    """Deterministic form of the hazard: every 64-bit word of the idle workspace = {the tag the NEXT launch's step 0 reads
    for, 1e6}. A reader that gets to a slot before its writer takes 1e6 for a sum. The guard clears the area first (it is a
    whole period old), so the pose is the fresh tracker's."""
    import torch
    k, depth, hk, dk = _scene(api, orc)
    make = api.DepthTracker if tracker_kind == "depth" else api.PyramidTracker

    def track(tracker, j):
        f = api.Frame(dk.depth, k, STARTS[j], normals=dk.normals)
        return bytes(tracker.track(f))

    fresh = make()
    fresh.keyframe = dk
    ref = [track(fresh, j) for j in (0, 1)]
    idle = make()
    idle.keyframe = dk
    first, _ = loop_count(api)
    assert track(idle, 0) == ref[0]
    sync()
    inner = idle if tracker_kind == "depth" else idle.tracker
    words = inner.workspace[:inner.workspace.numel() - 32].view(torch.int64)     # the exchange area (a pose follows it)
    # one period later the idle tracker's launches carry the tags of its last ones; its area "still holds" step 0's words
    nxt = first + PERIOD
    loop_count(api, set_to=nxt)
    tag = (epoch_of(nxt) << 10) | 1                                               # exchange_tag(epoch, step 0)
    stale = (tag << 32) | int(np.float32(1.0e6).view(np.uint32))
    words.fill_(stale - (1 << 64) if stale >= (1 << 63) else stale)
    sync()
    _, clears_before = loop_count(api)
    got = track(idle, 1)
    _, clears_after = loop_count(api)
    assert clears_after > clears_before
    assert got == ref[1], "stale words with this launch's tag were taken for this launch's sums"


def test_a_launch_per_stage_loop_in_between_marks_the_area(api, orc):
    """The launch-per-stage loop (the fallback after VK_TRACK_ABORTED, and the rig's reduce-hook loop) keeps float partials
    where the loop kernels keep tagged words; the next one-launch loop on that workspace clears it first."""
    k, depth, hk, dk = _scene(api, orc)
    t = api.DepthTracker()
    t.keyframe = dk
    f = api.Frame(dk.depth, k, STARTS[0], normals=dk.normals)
    ref = bytes(t.track(f))
    _, c0 = loop_count(api)
    f.depth_to_world = STARTS[0]
    assert bytes(t.track(f)) == ref
    _, c1 = loop_count(api)
    assert c1 == c0                                              # steady use: no clear
    f.depth_to_world = STARTS[0]
    t.compute_system(f)                                          # float partials over the tagged words
    f.depth_to_world = STARTS[0]
    assert bytes(t.track(f)) == ref
    _, c2 = loop_count(api)
    assert c2 == c1 + 1
