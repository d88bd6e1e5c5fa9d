"""vulcan::LightTracker on the device (vk_light_tracker_*) vs the oracle: residuals
and Jacobians bit for bit in both branches (photometric with shading, point-to-plane
fallback), the 27 sums to float-tree tolerance, and the reference's Track test
(light_tracker_test.cu:586-669) on the device."""
import numpy as np
import pytest

import color_scenes as cs
from test_gpu_parity import api, sync  # noqa: F401  (fixture)
from vulcan_amd import vk_types as T

pytestmark = pytest.mark.gpu


def bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def scene(api, orc):
    k, light = cs.projection(), cs.test_light()
    kd, kc = cs.plane_frame(cs.light_keyframe_pose(), False, light)
    fd, fc = cs.plane_frame(cs.light_frame_pose(), True, light)
    hk = orc.HostFrame(kd, k, cs.light_keyframe_pose(), color=kc)
    hf = orc.HostFrame(fd, k, cs.light_frame_pose(), color=fc)
    hk.compute_normals()
    hf.compute_normals()
    dk = api.Frame(kd, k, cs.light_keyframe_pose(), color=kc, normals=hk.normals)
    df = api.Frame(fd, k, cs.light_frame_pose(), color=fc, normals=hf.normals)
    return hk, hf, orc.ColorSide(hk, False), orc.ColorSide(hf, True), dk, df, light


def test_mask_residuals_jacobian_match(api, orc, scene):
    import torch
    hk, hf, ks, fs, dk, df, light = scene
    tracker = api.LightTracker()
    tracker.keyframe = dk
    tracker.light = light
    Tcm = orc.color_tcm(hk, hf)
    real_mask = orc.light_frame_mask(hf, 0.2)
    dev_mask = tracker.compute_frame_mask(df)
    sync()
    assert np.array_equal(dev_mask.cpu().numpy(), real_mask)
    assert 0.05 < real_mask.mean() < 0.999          # both branches occur with the real mask
    for mask in (real_mask, np.ones_like(real_mask), np.zeros_like(real_mask)):
        terms = orc.light_terms(hf, light, mask)
        dmask = torch.from_numpy(mask).cuda()
        r = tracker.compute_residuals(df, dmask)
        sync()
        assert np.array_equal(bits(r.cpu().numpy()), bits(orc.light_residuals(ks, fs, terms, Tcm)))
        for translation in (True, False):
            tracker.translation_enabled = translation
            J = tracker.compute_jacobian(df, dmask)
            sync()
            assert np.array_equal(bits(J.cpu().numpy()), bits(orc.light_jacobian(ks, fs, terms, Tcm, translation)))
        tracker.translation_enabled = True


def test_system_matches(api, orc, scene):
    import torch
    hk, hf, ks, fs, dk, df, light = scene
    tracker = api.LightTracker()
    tracker.keyframe = dk
    tracker.light = light
    Tcm = orc.color_tcm(hk, hf)
    mask = orc.light_frame_mask(hf, 0.2)
    terms = orc.light_terms(hf, light, mask)
    J = orc.light_jacobian(ks, fs, terms, Tcm, True).reshape(-1, 6).astype(np.float64)
    r = orc.light_residuals(ks, fs, terms, Tcm).reshape(-1).astype(np.float64)
    tracker.compute_system(df, torch.from_numpy(mask).cuda())
    sync()
    h, g = orc.light_system(ks, fs, terms, Tcm, True)
    got_h, got_g = tracker.hessian.cpu().numpy().astype(np.float64), tracker.gradient.cpu().numpy().astype(np.float64)
    assert np.all(np.abs(got_g - g) <= 2e-5 * np.abs(J * r[:, None]).sum(0) + 1e-12)
    idx = 0
    for rr in range(6):
        for c in range(rr + 1):
            assert abs(got_h[idx] - h[idx]) <= 2e-5 * np.abs(J[:, rr] * J[:, c]).sum() + 1e-12
            idx += 1


def test_track_holds_and_recovers_the_pose(api, orc, scene):
    """light_tracker_test.cu:586-669 on the device: 20 x Track from the true pose stays
    within 1e-5 of it; from a pose perturbed by 10 cm and ~1.2 degrees it comes back."""
    hk, hf, ks, fs, dk, df, light = scene
    tracker = api.LightTracker()
    tracker.keyframe = dk
    tracker.light = light
    true_pose = hf.depth_to_world
    frame = api.Frame(df.depth, df.depth_projection, true_pose, color=df.color, normals=df.normals)
    for _ in range(20):
        tracker.track(frame)
    diff = T.Transform._matmul(true_pose.inverse_matrix(), frame.depth_to_world.matrix())
    assert np.abs(diff - np.eye(4)).max() < 1e-5

    frame.depth_to_world = T.Transform.translate(0.1, 0.1, 0.1) * T.Transform.rotate(0.999871, 0.008638, -0.010375, 0.008638) * true_pose
    for _ in range(20):
        tracker.track(frame)
    diff = T.Transform._matmul(true_pose.inverse_matrix(), frame.depth_to_world.matrix())
    assert np.abs(diff - np.eye(4)).max() < 1e-5
