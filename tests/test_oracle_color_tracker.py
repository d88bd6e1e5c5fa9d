"""The oracle's colour tracker against the reference's own test cases
(tests/color_tracker_test.cu: ColorTracker.Residuals :494-540 and
ColorTracker.Jacobian :397-492), plus the two image operators it builds on."""
import numpy as np
import pytest

import color_scenes as cs
from vulcan_amd import vk_types as T


@pytest.fixture(scope="module")
def sides(orc):
    k = cs.projection()
    kd, kc = cs.keyframe_images()
    fd, fc = cs.frame_images()
    key = orc.HostFrame(kd, k, cs.keyframe_pose(), color=kc)
    frm = orc.HostFrame(fd, k, cs.frame_pose(), color=fc)
    key.compute_normals()
    frm.compute_normals()
    return key, frm, orc.ColorSide(key, False), orc.ColorSide(frm, True)


def test_convert_and_gradients(orc):
    rng = np.random.default_rng(0)
    rgb = rng.random((7, 9, 3), dtype=np.float32)
    gray = orc.color_convert(rgb)
    want = ((rgb[..., 0] + rgb[..., 1]) + rgb[..., 2]) / np.float32(3.0)       # image.cu:17
    assert np.array_equal(gray, want.astype(np.float32))
    gx, gy = orc.image_gradients(gray)
    p = np.pad(gray.astype(np.float64), 1)
    ex = (0.125 * p[:-2, 2:] + 0.25 * p[1:-1, 2:] + 0.125 * p[2:, 2:]) - (0.125 * p[:-2, :-2] + 0.25 * p[1:-1, :-2] + 0.125 * p[2:, :-2])
    ey = (0.125 * p[2:, :-2] + 0.25 * p[2:, 1:-1] + 0.125 * p[2:, 2:]) - (0.125 * p[:-2, :-2] + 0.25 * p[:-2, 1:-1] + 0.125 * p[:-2, 2:])
    np.testing.assert_allclose(gx, ex, atol=1e-6)
    np.testing.assert_allclose(gy, ey, atol=1e-6)
    ramp = np.tile(np.arange(12, dtype=np.float32), (8, 1))                     # interior of a unit ramp: gx = 1
    gx, gy = orc.image_gradients(ramp)
    assert np.all(gx[1:-1, 1:-1] == 1.0) and np.all(gy[1:-1, 1:-1] == 0.0)


def test_residuals_identical_frames_are_zero(orc, sides):
    """color_tracker_test.cu:512-521: the keyframe tracked against itself"""
    key, _, key_side, _ = sides
    same = orc.ColorSide(key, True)
    r = orc.color_residuals(key_side, same, orc.color_tcm(key, key))
    assert np.abs(r).max() <= 1e-6


def test_residuals_match_double_precision(orc, sides):
    """color_tracker_test.cu:523-539: 1e-4 wherever the double-precision replay says visible"""
    key, frm, key_side, frm_side = sides
    Tcm = orc.color_tcm(key, frm)
    found = orc.color_residuals(key_side, frm_side, Tcm)
    expected, visible = cs.residuals64(cs.projection(), Tcm, key.depth, key.normals, key_side.intensities,
                                       frm.depth, frm.normals, frm_side.intensities)
    assert visible.mean() > 0.9
    assert np.abs(found - expected)[visible].max() < 1e-4


def test_jacobian_matches_central_differences(orc, sides):
    """color_tracker_test.cu:397-492: step 1e-2 per parameter through GetTransformX,
    relative criterion |d / min| < |0.065 / mean| away from the image border."""
    key, frm, key_side, frm_side = sides
    k = cs.projection()
    found = orc.color_jacobian(key_side, frm_side, orc.color_tcm(key, frm), True).astype(np.float64)

    expected = np.zeros((cs.H, cs.W, 6))
    usable = np.ones((cs.H, cs.W, 6), dtype=bool)
    base = frm.depth_to_world
    for i in range(6):
        res = []
        for sign in (+1, -1):
            u = np.zeros(6, dtype=np.float32)
            u[i] = sign * 1e-2
            moved = orc.HostFrame(frm.depth, k, cs.transform_of_update(u, base), color=frm.color, normals=frm.normals)
            res.append(cs.residuals64(k, orc.color_tcm(key, moved), key.depth, key.normals, key_side.intensities,
                                      frm.depth, frm.normals, frm_side.intensities))
        expected[..., i] = (res[0][0] - res[1][0]) / (2 * 1e-2)
        usable[..., i] = res[0][1] & res[1][1]

    # :463-475: keyframe pixels 5 px inside, landing 5 px inside the frame
    y, x = np.mgrid[0:cs.H, 0:cs.W]
    Xm = np.stack([(x + 0.5 - 320) / 547, (y + 0.5 - 240) / 547, np.ones_like(x, dtype=np.float64)], -1)
    Twm, Tcw = key.depth_to_world.matrix().astype(np.float64), frm.depth_to_world.inverse_matrix().astype(np.float64)
    Xw = Xm @ Twm[:3, :3].T + Twm[:3, 3]
    Xc = Xw @ Tcw[:3, :3].T + Tcw[:3, 3]
    uu, vv = 547 * Xc[..., 0] / Xc[..., 2] + 320, 547 * Xc[..., 1] / Xc[..., 2] + 240
    region = (x >= 5) & (x < cs.W - 5) & (y >= 5) & (y < cs.H - 5)
    region &= ~((uu < 5) | (uu > cs.W - 5) | (vv < 5) | (vv > cs.H - 5))
    check = usable & region[..., None]
    assert check.mean() > 0.8

    d = np.abs(found - expected)
    n = np.minimum(np.abs(found), np.abs(expected))
    p = 0.5 * (np.abs(found) + np.abs(expected))
    with np.errstate(divide="ignore", invalid="ignore"):
        r = np.abs(d / n)
        limit = np.abs(0.065 / p)
    bad = check & ~(r < limit)
    assert not bad.any(), f"{bad.sum()} entries fail, worst d={d[bad].max()}"


def test_solve_update_moves_towards_the_keyframe(orc, sides):
    """No upstream Track test for ColorTracker; Gauss-Newton on the two test frames
    must shrink the photometric cost (color_tracker.cpp:34-96 + tracker.cpp:53-63)."""
    key, frm, key_side, frm_side = sides
    pose = T.ColorPose()
    pose.depth_to_world = frm.depth_to_world
    Tcd = frm.depth_to_color
    key_Twc = (key.depth_to_color * key.depth_to_world.inverse()).inverse()
    import ctypes as C
    orc.lib().orc_color_tracker_tcm(C.byref(Tcd), C.byref(key_Twc), C.byref(pose))
    first = orc.color_tcm(key, frm)
    assert np.array_equal(np.array(pose.Tcm.m[:]), np.array(first.m[:]))
    costs = []
    for _ in range(6):
        r = orc.color_residuals(key_side, frm_side, pose.Tcm)
        costs.append(float((r.astype(np.float64) ** 2).sum()))
        h, g = orc.color_system(key_side, frm_side, pose.Tcm, True)
        orc.color_solve_update(h, g, Tcd, key_Twc, pose, True)
    assert costs[-1] < 0.5 * costs[0]


def test_gradients_equal_sobel_over_eight(orc):
    """image_test.cu:36-99 Image.GetGradients: on the test's 640x480 image the result
    equals a 3x3 Sobel filter / 8 to 1e-6 two pixels inside the border."""
    w, h = 640, 480
    y, x = np.mgrid[0:h, 0:w]
    f32 = np.float32
    value = (f32(0.25) + f32(0.25) * np.cos((64 * np.pi * x / (w - 1)).astype(np.float32), dtype=np.float32)).astype(np.float32)
    value = value + (f32(0.25) + f32(0.25) * np.cos((64 * np.pi * y / (h - 1)).astype(np.float32), dtype=np.float32)).astype(np.float32)
    image = (value * value).astype(np.float32)
    gx, gy = orc.image_gradients(image)
    p = image.astype(np.float64)
    sx = (p[:-2, 2:] + 2 * p[1:-1, 2:] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[1:-1, :-2] + p[2:, :-2])
    sy = (p[2:, :-2] + 2 * p[2:, 1:-1] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[:-2, 1:-1] + p[:-2, 2:])
    assert np.abs(gx[1:-1, 1:-1] - sx / 8)[1:-1, 1:-1].max() < 1e-6
    assert np.abs(gy[1:-1, 1:-1] - sy / 8)[1:-1, 1:-1].max() < 1e-6
