"""The frame loop of the reference app with tracking closed around fusion
(apps/vulcan/vulcan.cu:297-325): ComputeNormals -> PyramidTracker<DepthTracker>::Track against the
previous raycast, starting from the previous tracked pose -> SetView x3 -> LightIntegrator ->
Tracer at the TRACKED pose. Scene: the room of tests/scenes.py, in which all six pose parameters
are observable and the camera really moves (the true poses only score the result).

Two more forms of the same loop: one at the benchmark's full size (640x480, three frames), and the
shipped app's own configuration (vulcan.cu:89-111): a plain LightTracker with SetMaxIterations(1) and a
LightIntegrator with weight caps 100 / 16 — also with 20 steps per frame.

Checked against the oracle's own closed loop:
  * the device's tracked poses stay within 2e-5 (per matrix entry) of the oracle's;
  * given the oracle's poses, the device's volume and raycasts equal the oracle's bit for bit.
"""
import numpy as np
import pytest

import scenes
from test_gpu_parity import api, assert_volume_equal, sync  # noqa: F401
from vulcan_amd import vk_types as T

pytestmark = pytest.mark.gpu

LIGHT = (2.0, (0.025, 0.08, 0.0))          # apps/vulcan/vulcan.cu:87-88


def pose_error(got, truth):
    d = got.matrix().astype(np.float64) @ truth.inverse_matrix().astype(np.float64)
    angle = np.degrees(np.arccos(np.clip((np.trace(d[:3, :3]) - 1.0) / 2.0, -1.0, 1.0)))
    return float(np.linalg.norm(got.matrix()[:3, 3].astype(np.float64) - truth.matrix()[:3, 3])), float(angle)


def closed_loop(api, orc, size, count, stride, tracking, free_running_atol=2e-5):
    """`tracking`: ("depth-pyramid",) = PyramidTracker<DepthTracker>; ("light", steps) = LightTracker with
    that many Gauss-Newton steps per frame and the app's integrator caps (vulcan.cu:92-93,111)."""
    import torch
    w, h = size
    scale = w / 640.0
    k = T.Projection.make(*(np.float32(scale) * np.float32(v) for v in scenes.APP_INTRINSICS))
    truth = [scenes.room_pose(30 + stride * i) for i in range(count)]      # 30: away from the turning point
    inputs = [scenes.room_frame(k, p, w, h, light=LIGHT) for p in truth]
    light = T.Light.make(*LIGHT)
    orc.set_threads(16)
    main, excess, voxel, trunc = 65024, 8192, 0.005, 0.04
    photometric = tracking[0] == "light"
    params = T.Integrator(0.1, 5.0, 100.0, 16.0) if photometric else T.Integrator.default()

    # ---- the oracle's closed loop
    hv = orc.HostVolume(main, excess, voxel_length=voxel, truncation_length=trunc)
    pose, hkey, oracle = truth[0], None, []
    for i, (depth, color) in enumerate(inputs):
        hf = orc.HostFrame(depth, k, pose, color=color)
        hf.compute_normals()
        if i > 0:
            if photometric:
                pose, _ = orc.light_track(hkey, hf, light, tracking[1])
            else:
                pose, _ = orc.pyramid_track(hkey, hf)
        hf.depth_to_world = pose
        for _ in range(3):
            hv.set_view(hf, orc.POLICY_MAXKEY)
        orc.integrate_depth(hv, hf, params)
        orc.integrate_light_color(hv, hf, light, orc.light_frame_mask(hf, 0.2), params)
        odepth, ocolor, onormals, obounds = orc.trace(hv, hf)
        hkey = orc.HostFrame(odepth, k, pose, color=ocolor, normals=onormals)
        oracle.append(dict(pose=pose, depth=odepth, color=ocolor, normals=onormals, bounds=obounds,
                           visible=hv.visible_count))

    def device_loop(follow_oracle):
        dv = api.Volume(main, excess, voxel_length=voxel, truncation_length=trunc)
        integ, tracer = api.LightIntegrator(dv), api.Tracer(dv)
        integ.light, integ.params = light, params
        if photometric:
            tracker = api.LightTracker()
            tracker.light, tracker.max_iterations = light, tracking[1]
        else:
            tracker = api.PyramidTracker()
        keys = [api.Frame(torch.zeros((h, w), dtype=torch.float32, device="cuda"), k, truth[0],
                          color=torch.zeros((h, w, 3), dtype=torch.float32, device="cuda"),
                          normals=torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")) for _ in range(2)]
        dpose, poses, single = truth[0], [], []
        for i, (depth, color) in enumerate(inputs):
            df = api.Frame(depth, k, dpose, color=color)
            df.compute_normals()
            if i > 0:
                tracker.keyframe = keys[(i - 1) & 1]
                dpose = tracker.track(df)
            if follow_oracle:
                # one Track from the oracle's pose against the oracle's keyframe (the raycast is bit-exact)
                single.append(float(np.abs(dpose.matrix() - oracle[i]["pose"].matrix()).max()))
                dpose = oracle[i]["pose"]
            df.depth_to_world = dpose
            out = keys[i & 1]
            out.depth_to_world = dpose
            dv.set_view(df, rounds=3)
            integ.integrate(df)
            tracer.trace(out)
            sync()
            poses.append(dpose)
            if follow_oracle:
                want = oracle[i]
                assert dv.visible_count == want["visible"] > 3000
                assert np.array_equal(tracer.bounds.cpu().numpy(), want["bounds"])
                assert np.array_equal(out.depth.cpu().numpy(), want["depth"])
                assert np.array_equal(out.color.cpu().numpy(), want["color"])
                assert np.array_equal(out.normals.cpu().numpy(), want["normals"], equal_nan=True)
        if follow_oracle:
            print("   one Track from the oracle's state, max |device - oracle| per frame: " + ", ".join(f"{d:.1e}" for d in single))
            assert max(single) <= 2e-5
        return dv, poses

    # ---- the device's own closed loop: poses against the oracle's loop and against the truth
    _, poses = device_loop(follow_oracle=False)
    print("   closed loop, max |device - oracle| per frame: "
          + ", ".join(f"{np.abs(poses[i].matrix() - oracle[i]['pose'].matrix()).max():.1e}" for i in range(count)))
    for i in range(count):
        np.testing.assert_allclose(poses[i].matrix(), oracle[i]["pose"].matrix(), atol=free_running_atol, rtol=0)
        np.testing.assert_allclose(poses[i].inverse_matrix(), oracle[i]["pose"].inverse_matrix(), atol=free_running_atol, rtol=0)
    errors = [pose_error(p, t) for p, t in zip(poses, truth)]
    moved = pose_error(truth[-1], truth[0])
    print(f"{w}x{h}, {tracking}, stride {stride}: camera moved {moved[0] * 1e3:.1f} mm / {moved[1]:.2f} deg; "
          "tracking error per frame " + ", ".join(f"{e[0] * 1e3:.2f} mm / {e[1]:.3f} deg" for e in errors))
    assert moved[0] > 0.005 and moved[1] > 0.5                      # the camera really moved ...

    # ---- the device's fusion and raycast at the oracle's poses: bit-exact
    dv, _ = device_loop(follow_oracle=True)
    assert_volume_equal(dv, hv)
    orc.set_threads(1)
    return errors, moved


@pytest.mark.parametrize("stride", [1, 3])
def test_closed_loop_in_the_room_matches_the_oracle(api, orc, stride):
    """`stride` = every how many poses of the room sequence a frame is taken: 1 is the bench's
    motion (<= 0.8 deg and ~5 mm per frame), 3 three times that."""
    errors, moved = closed_loop(api, orc, (320, 240), 5, stride, ("depth-pyramid",))
    assert moved[0] > 0.01
    assert max(e[0] for e in errors) < 0.1 * moved[0] + 0.002      # ... and the tracker followed it
    assert max(e[1] for e in errors) < 0.1 * moved[1] + 0.02


def test_closed_loop_at_the_benchmark_size(api, orc):
    """640x480, the size `bench.py --workload rgbd-icp` runs: three frames."""
    errors, moved = closed_loop(api, orc, (640, 480), 3, 2, ("depth-pyramid",))
    assert max(e[0] for e in errors) < 0.1 * moved[0] + 0.002
    assert max(e[1] for e in errors) < 0.1 * moved[1] + 0.02


@pytest.mark.parametrize("steps", [1, 20])
def test_closed_loop_with_the_light_tracker(api, orc, steps):
    """apps/vulcan/vulcan.cu:89-111 as shipped (`steps` = 1): LightTracker, one Gauss-Newton step per frame,
    LightIntegrator capped at 100 / 16 — and the same loop with the tracker's default 20 steps. The
    photometric tracker is scored against the oracle's loop only; how far either is from the truth is
    printed. Every Track that starts from the oracle's state (its pose, its raycast — which the device reproduces bit
    for bit) must land within 2e-5 of the oracle's in both forms (measured: 3e-7). The free-running loops are
    compared at 2e-5 with 20 steps; with ONE step per frame the loop is never converged and carries a difference of
    one unit in the last place to 1e-4 ... 1e-3 within three frames — in the oracle alone
    (tests/test_oracle_closed_loop.py) — so there the free-running poses are only held to the tracker's own
    accuracy (2e-3)."""
    closed_loop(api, orc, (320, 240), 5, 1, ("light", steps), free_running_atol=2e-5 if steps == 20 else 2e-3)
