"""Synthetic inputs shared by the tests and bench.py.

Every scene is closed-form, taken from the reference's own tests:
  plane   depth == d                                   tracer_test.cu:279-285, integrator_test.cu:87-92
  sphere  4 - 1.5*sqrt(200^2 - r^2)/200, r < 200 px    tracer_test.cu:441-459
  ripple  1 + .01cos(16pi x/(w-1)) + .01cos(16pi y/(h-1))   depth_tracker_test.cu:89-97
  ramp    1 + 3*((x+y)%100)/99                         volume_test.cpp:288
"""
import numpy as np

from vulcan_amd import vk_types as T

APP_INTRINSICS = (544.162, 544.3847, 311.2701, 234.7798)      # apps/vulcan/vulcan.cu:283-287
TRACER_TEST_INTRINSICS = (546.723, 553.914, 321.294, 239.052)  # tracer_test.cu:275-277


def plane(w, h, d=1.5):
    return np.full((h, w), d, dtype=np.float32)


def sphere(w, h):
    y, x = np.mgrid[0:h, 0:w]
    u = x.astype(np.float32) + np.float32(0.5) - np.float32(0.5 * w)
    v = y.astype(np.float32) + np.float32(0.5) - np.float32(0.5 * h)
    rr = (u * u + v * v).astype(np.float32)
    r = np.sqrt(rr)
    z = np.sqrt(np.maximum(200.0 * 200.0 - rr.astype(np.float64), 0.0))
    depth = np.where(r < 200, 4.0 - 1.5 * (z / 200.0), 0.0)
    return depth.astype(np.float32)


def ripple(w, h):
    y, x = np.mgrid[0:h, 0:w]
    d = np.ones((h, w), dtype=np.float32)
    # `float += double`: add in double, round to float after each statement
    d = (d.astype(np.float64) + 0.01 * np.cos(16 * np.pi * x / (w - 1))).astype(np.float32)
    d = (d.astype(np.float64) + 0.01 * np.cos(16 * np.pi * y / (h - 1))).astype(np.float32)
    return d


def ramp(w, h):
    y, x = np.mgrid[0:h, 0:w]
    return (1 + 3 * (((x + y) % 100) / 99.0)).astype(np.float32)


def constant_color(w, h, rgb=(0.1, 0.2, 0.3)):
    c = np.zeros((h, w, 3), dtype=np.float32)
    c[:] = np.asarray(rgb, dtype=np.float32)
    return c


def checker_color(w, h, lo=0.0, hi=1.0):
    """tracer_test.cu:455: (0, (x%40<20)^(y%40<20), 1); lo/hi let the light-mask
    tests keep values inside (.02,.98)."""
    y, x = np.mgrid[0:h, 0:w]
    g = ((x % 40 < 20) ^ (y % 40 < 20)).astype(np.float32)
    c = np.zeros((h, w, 3), dtype=np.float32)
    c[..., 0] = lo
    c[..., 1] = lo + (hi - lo) * g
    c[..., 2] = hi
    return c


def tracer_test_pose():
    """Tcw of tracer_test.cu:267-271; returns Twc = Tcw.Inverse()."""
    tcw = T.Transform.translate(0.3, -1.3, 3.7) * T.Transform.rotate(0.7474, 0.3438, -0.3884, 0.4152)
    return tcw.inverse()


def yaw(deg):
    a = np.deg2rad(deg) / 2.0
    return T.Transform.rotate(np.cos(a), 0.0, np.sin(a), 0.0)


def orbit_pose(i, step_deg=0.5, base=None):
    """Pose i of the bench sequence: `base` pre-multiplied by a yaw of i*step_deg
    (SURVEY §8d 'orbit')."""
    base = base or T.Transform.identity()
    return yaw(step_deg * i) * base


def block_map(entries, voxels=None):
    """{(bx,by,bz): entry index} over allocated entries — parity is keyed on block
    origin, never on pool slot."""
    out = {}
    idx = np.nonzero(entries["data"] >= 0)[0]
    for i in idx:
        o = entries["block"]["origin"][i]
        out[(int(o[0]), int(o[1]), int(o[2]))] = int(i)
    return out


def tracer_patch_kat():
    """tracer_test.cu:22-66: five blocks placed by unprojecting image points at
    known depths through Tcw^-1; returns (entries, Tcw, projection)."""
    F = np.float32
    block_length = F(0.008)
    tcw = T.Transform.translate(0.3, -1.3, 3.7) * T.Transform.rotate(0.7474, 0.3438, -0.3884, 0.4152)
    k = T.Projection.make(346.723, 353.914, 321.294, 239.052)
    points = [(320.0, 240.0, 2.5), (120.0, 340.0, 1.5), (420.0, 240.0, 0.6),
              (-20.0, -40.0, 1.0), (720.0, 580.0, 1.0)]
    entries = np.zeros(len(points), dtype=T.hash_entry_dtype)
    for i, (u, v, d) in enumerate(points):
        ifx, ify = F(1) / F(k.fx), F(1) / F(k.fy)
        xcp = np.array([ifx * F(u) - F(k.cx) * ifx, ify * F(v) - F(k.cy) * ify, F(1)], np.float32) * F(d)
        xwp = (tcw.inverse_matrix() @ np.append(xcp, F(1)).astype(np.float32)).astype(np.float32)
        entries["block"]["origin"][i] = [np.int16(int(F(c) / block_length)) for c in xwp[:3]]
        entries["data"][i], entries["next"][i] = 0, -1
    return entries, tcw, k
