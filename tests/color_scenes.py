"""The two synthetic frames of the reference's ColorTracker tests
(tests/color_tracker_test.cu:12-148 CreateKeyframeX / CreateFrameX) and the
double-precision checks that file applies (:150-395), restated with numpy. The
reference's Frame then had one pose `Twc` and one `projection`; in today's fields
that is depth_to_world = Twc, depth_to_color = identity, both projections equal."""
import numpy as np

from vulcan_amd import vk_types as T

W, H = 640, 480


def projection():
    return T.Projection.make(547, 547, 320, 240)


def _colors(freq):
    y, x = np.mgrid[0:H, 0:W]
    xr = (x.astype(np.float32) / np.float32(W - 1)).astype(np.float32)
    yr = (y.astype(np.float32) / np.float32(H - 1)).astype(np.float32)
    c = np.full((H, W), 0.5, dtype=np.float32)
    # color += 0.245 * cosf(freq * M_PI * ratio): double product rounded to float on the add
    c = (c.astype(np.float64) + 0.245 * np.cos((freq * np.pi * xr.astype(np.float64)).astype(np.float32)).astype(np.float32)).astype(np.float32)
    c = (c.astype(np.float64) + 0.245 * np.cos((freq * np.pi * yr.astype(np.float64)).astype(np.float32)).astype(np.float32)).astype(np.float32)
    return np.repeat(c[:, :, None], 3, axis=2).copy()


def keyframe_pose():
    return T.Transform.translate(0.0011, -0.0019, 0.0031) * T.Transform.rotate(0.9998715, 0.0086385, -0.0103759, 0.0086385)


def frame_pose():
    return T.Transform.translate(0.001, -0.002, 0.003) * T.Transform.rotate(0.9998719, 0.0085884, -0.0104268, 0.0085884)


def keyframe_images():
    depth = np.ones((H, W), dtype=np.float32)
    return depth, _colors(3.0)


def frame_images():
    y, x = np.mgrid[0:H, 0:W]
    d = np.ones((H, W), dtype=np.float64)
    d = (d.astype(np.float32) + 0.01 * np.cos(16 * np.pi * x / (W - 1))).astype(np.float32)
    d = (d + 0.01 * np.cos(16 * np.pi * y / (H - 1))).astype(np.float32)
    return d, _colors(4.0)


def sample64(values, u, v):
    """color_tracker_test.cu:150-174 SampleX"""
    x = np.floor(u - 0.5).astype(np.int64)
    y = np.floor(v - 0.5).astype(np.int64)
    w = values.shape[1]
    flat = values.reshape(-1).astype(np.float64)
    v00 = flat[(y + 0) * w + (x + 0)]
    v01 = flat[(y + 0) * w + (x + 1)]
    v10 = flat[(y + 1) * w + (x + 0)]
    v11 = flat[(y + 1) * w + (x + 1)]
    u1 = u - (x + 0.5)
    v1 = v - (y + 0.5)
    u0 = 1.0 - u1
    v0 = 1.0 - v1
    return (v0 * u0 * v00) + (v0 * u1 * v01) + (v1 * u0 * v10) + (v1 * u1 * v11)


def residuals64(k, Tcm, key_depth, key_normals, key_int, frm_depth, frm_normals, frm_int):
    """color_tracker_test.cu:176-246 ComputeResidual for every keyframe pixel:
    returns (residual, visible)."""
    h, w = key_depth.shape
    fx, fy, cx, cy = float(k.fx), float(k.fy), float(k.cx), float(k.cy)
    M = Tcm.matrix().astype(np.float64)
    y, x = np.mgrid[0:h, 0:w]
    d = key_depth.astype(np.float64)
    u, v = x + 0.5, y + 0.5
    Xm = np.stack([d * (u / fx - cx / fx), d * (v / fy - cy / fy), d], -1)
    Xc = Xm @ M[:3, :3].T + M[:3, 3]
    fu = fx * Xc[..., 0] / Xc[..., 2] + cx
    fv = fy * Xc[..., 1] / Xc[..., 2] + cy
    ok = (d > 0.001) & (fu >= 1.0) & (fu < w - 1.0) & (fv >= 1.0) & (fv < h - 1.0)
    fu_s, fv_s = np.where(ok, fu, 1.5), np.where(ok, fv, 1.5)
    fi = fv_s.astype(np.int64) * w + fu_s.astype(np.int64)
    ok &= np.abs(frm_depth.reshape(-1)[fi].astype(np.float64) - Xc[..., 2]) < 0.099
    fn = frm_normals.reshape(-1, 3)[fi].astype(np.float64)
    kn = key_normals.astype(np.float64) @ M[:3, :3].T
    ok &= ((kn * kn).sum(-1) > 0.001) & ((fn * kn).sum(-1) > 0.501)
    Ic = sample64(frm_int, fu_s, fv_s)
    return np.where(ok, Ic - key_int.astype(np.float64), 0.0), ok


def transform_of_update(update, Twc):
    """color_tracker_test.cu:307-372 GetTransformX (float32, as color_tracker.cpp:34-96)"""
    f = np.float32
    u = np.zeros(6, dtype=np.float32)
    u[:] = update
    Tinc = np.array([[1, -u[2], u[1], u[3]], [u[2], 1, -u[0], u[4]], [-u[1], u[0], 1, u[5]], [0, 0, 0, 1]], dtype=np.float32)
    M = T.Transform._matmul(Tinc, Twc.inverse_matrix())
    xa, ya = M[:3, 0].copy(), M[:3, 1].copy()

    def dot(a, b):
        acc = f(0)
        for i in range(3):
            acc = f(acc + f(a[i] * b[i]))
        return acc

    def normalized(a):
        return (a * f(f(1) / np.sqrt(dot(a, a), dtype=np.float32))).astype(np.float32)

    def cross(a, b):
        return np.array([f(a[1] * b[2]) - f(a[2] * b[1]), f(a[2] * b[0]) - f(a[0] * b[2]), f(a[0] * b[1]) - f(a[1] * b[0])], dtype=np.float32)

    xa, ya = normalized(xa), normalized(ya)
    za = cross(xa, ya)
    ya = cross(za, xa)
    R = np.eye(4, dtype=np.float32)
    R[:3, 0], R[:3, 1], R[:3, 2] = xa, ya, za
    rot = T.Transform.from_matrices(R, R.T)
    return (T.Transform.translate(M[0, 3], M[1, 3], M[2, 3]) * rot).inverse()


# ---- LightTracker scenes (tests/light_tracker_test.cu:12-189 CreateKeyframeY / CreateFrameY) ----

def light_keyframe_pose():
    return T.Transform.translate(0.0011, -0.0019, -0.5531) * T.Transform.rotate(0.9998715, 0.0086385, -0.0103759, 0.0086385)


def light_frame_pose():
    return T.Transform.translate(0.0010, -0.002, -0.4030) * T.Transform.rotate(0.9998719, 0.0085884, -0.0104268, 0.0085884)


def test_light():
    return T.Light.make(2.0, (0.1, 0.0, 0.0))          # light_tracker_test.cu:465-467


def plane_frame(pose, shaded, light):
    """A camera at `pose` looking at the textured plane z = 1 of the world; returns
    (depth, colour). shaded: colour = albedo * light.GetShading(Xcp, n) with the
    light expressed in the camera frame (light_tracker_test.cu:57-95)."""
    k = projection()
    M, Mi = pose.matrix().astype(np.float64), pose.inverse_matrix().astype(np.float64)
    y, x = np.mgrid[0:H, 0:W]
    Xc = np.stack([(x + 0.5 - k.cx) / k.fx, (y + 0.5 - k.cy) / k.fy, np.ones((H, W))], -1)
    d = Xc @ M[:3, :3].T
    origin = M[:3, 3]
    length = (1.0 - origin[2]) / d[..., 2]
    Xw = origin + length[..., None] * d
    Xcp = Xw @ Mi[:3, :3].T + Mi[:3, 3]
    depth = Xcp[..., 2].astype(np.float32)
    c = 0.5 + 0.245 * np.cos(3.0 * np.pi * Xw[..., 0]) + 0.245 * np.cos(3.0 * np.pi * Xw[..., 1])
    if shaded:
        n = Mi[:3, :3] @ np.array([0.0, 0.0, -1.0])
        delta = np.array(light.position[:], dtype=np.float64) - Xcp
        d2 = (delta * delta).sum(-1)
        cos_theta = (delta / np.sqrt(d2)[..., None]) @ n
        c = c * (float(light.intensity) * cos_theta / d2)
    return depth, np.repeat(c.astype(np.float32)[:, :, None], 3, axis=2).copy()


def light_residuals64(k, Tcm, light, key_depth, key_normals, key_albedo, frm_depth, frm_normals, frm_int):
    """light_tracker_test.cu:216-268 ComputeResidualY for every keyframe pixel:
    photometric model everywhere (the test has no mask); returns (residual, visible)."""
    h, w = key_depth.shape
    fx, fy, cx, cy = float(k.fx), float(k.fy), float(k.cx), float(k.cy)
    M = Tcm.matrix().astype(np.float64)
    y, x = np.mgrid[0:h, 0:w]
    d = key_depth.astype(np.float64)
    Xm = np.stack([d * ((x + 0.5 - cx) / fx), d * ((y + 0.5 - cy) / fy), d], -1)
    Xc = Xm @ M[:3, :3].T + M[:3, 3]
    fu = fx * Xc[..., 0] / Xc[..., 2] + cx
    fv = fy * Xc[..., 1] / Xc[..., 2] + cy
    ok = (d > 0.001) & (fu >= 1.0) & (fu < w - 1.0) & (fv >= 1.0) & (fv < h - 1.0)
    fu_s, fv_s = np.where(ok, fu, 1.5), np.where(ok, fv, 1.5)
    fi = fv_s.astype(np.int64) * w + fu_s.astype(np.int64)
    ok &= np.abs(frm_depth.reshape(-1)[fi].astype(np.float64) - Xc[..., 2]) < 0.099
    fn = frm_normals.reshape(-1, 3)[fi].astype(np.float64)
    kn = key_normals.astype(np.float64) @ M[:3, :3].T
    ok &= ((kn * kn).sum(-1) > 0.501) & ((fn * kn).sum(-1) > 0.501)
    aa = key_albedo.astype(np.float64)
    ok &= aa > 0
    delta = np.array(light.position[:], dtype=np.float64) - Xc
    d2 = (delta * delta).sum(-1)
    shading = float(light.intensity) * ((delta / np.sqrt(d2)[..., None]) * kn).sum(-1) / d2
    Ic = sample64(frm_int, fu_s, fv_s)
    return np.where(ok, Ic - shading * aa, 0.0), ok
