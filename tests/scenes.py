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


# ---------------------------------------------------------------- the room ----
# A closed-form scene in which a depth camera observes all six degrees of freedom: a box room
# (three pairs of orthogonal walls) with spheres standing on the floor, hanging from the ceiling
# and attached to the walls. Every ray hits something, so the depth image has no holes; the
# intersections are analytic (a slab test and a quadratic per sphere, float64, rounded once).
# Used by the tracking workloads of bench.py / fuse_sequence and by the closed-loop parity test:
# the sphere-centred scene of the fusion benchmark is invariant under rotation, hence useless
# for tracking.
ROOM_HALF = (2.2, 1.5, 2.6)                                   # box half extents (x, y, z), metres; +y is down
ROOM_SPHERES = ((1.2, 0.9, 1.6, 0.6), (-1.4, 1.0, 1.2, 0.5), (-1.0, 0.95, -1.7, 0.55), (1.5, 1.05, -1.3, 0.45),
                (0.2, -0.9, 2.1, 0.5), (-1.9, -0.2, -0.3, 0.5), (1.9, 0.1, 0.2, 0.45), (0.1, 1.1, 2.2, 0.4),
                (-0.3, 1.15, -2.2, 0.35))                     # (cx, cy, cz, radius)


def room_pose(i, frames_per_cycle=240, yaw_amplitude_deg=24.0, pitch_deg=12.0):
    """Camera pose i of the room sequence (depth-to-world): the yaw swings +-24 deg (at most
    0.63 deg per frame), the camera is pitched 12 deg towards the floor and its centre moves on a
    small closed curve (up to 6.5 mm per frame), so depth changes in every pixel, every frame. A
    whole cycle shows the camera ~56 k blocks of 4 cm: it fits the app's Volume(65024, 8192)."""
    phase = 2.0 * np.pi * i / frames_per_cycle
    a = np.deg2rad(yaw_amplitude_deg) * np.sin(phase)
    p = np.deg2rad(pitch_deg)
    position = (0.25 * np.sin(phase), 0.04 * np.sin(2.0 * phase), 0.25 * (np.cos(phase) - 1.0))
    yaw_q = T.Transform.rotate(np.cos(a / 2), 0.0, np.sin(a / 2), 0.0)
    pitch_q = T.Transform.rotate(np.cos(p / 2), np.sin(p / 2), 0.0, 0.0)
    return T.Transform.translate(*position) * yaw_q * pitch_q


def _room_hit(k, pose, w, h, with_normal=True):
    """Per pixel: depth z (the ray is unproject(u, v) with z = 1, so its parameter IS the depth),
    the world point and the inward surface normal (world) as three planes each, float64."""
    m = pose.matrix().astype(np.float64)
    rot, pos = m[:3, :3], m[:3, 3]
    rx = ((np.arange(w) + 0.5 - k.cx) / k.fx)[None, :]
    ry = ((np.arange(h) + 0.5 - k.cy) / k.fy)[:, None]
    d = [rot[j, 0] * rx + rot[j, 1] * ry + rot[j, 2] for j in range(3)]      # world direction, |d| != 1
    s = np.full((h, w), np.inf)
    axis = np.zeros((h, w), dtype=np.int8)
    with np.errstate(divide="ignore", invalid="ignore"):
        for j in range(3):
            sj = np.where(d[j] == 0, np.inf, (np.sign(d[j]) * ROOM_HALF[j] - pos[j]) / d[j])   # exit parameter
            closer = sj < s
            s = np.where(closer, sj, s)
            axis[closer] = j
    normal = [np.where(axis == j, -np.sign(d[j]), 0.0) for j in range(3)] if with_normal else None
    aa = d[0] * d[0] + d[1] * d[1] + d[2] * d[2]
    for cx, cy, cz, r in ROOM_SPHERES:
        oc = pos - np.array([cx, cy, cz])
        b = d[0] * oc[0] + d[1] * oc[1] + d[2] * oc[2]
        disc = b * b - aa * (oc @ oc - r * r)
        t = (-b - np.sqrt(np.maximum(disc, 0.0))) / aa
        hit = (disc > 0) & (t > 0) & (t < s)
        if not hit.any():
            continue
        s = np.where(hit, t, s)
        if with_normal:                                        # outward from the ball = into the room
            normal = [np.where(hit, (oc[j] + t * d[j]) / r, normal[j]) for j in range(3)]
    point = [pos[j] + s * d[j] for j in range(3)]
    return s, point, normal, (np.broadcast_to(rx, (h, w)), np.broadcast_to(ry, (h, w))), rot


def room_depth(k, pose, w, h):
    return _room_hit(k, pose, w, h, with_normal=False)[0].astype(np.float32)


def room_albedo(point):
    return 0.5 + 0.15 * np.cos(5.0 * point[0] + 1.0) * np.cos(4.0 * point[2]) + 0.15 * np.cos(6.0 * point[1])


def room_frame(k, pose, w, h, light=None):
    """(depth, colour) of the room from `pose`. With a light (intensity, position in the camera
    frame: the app's lamp sits next to the camera, apps/vulcan/vulcan.cu:87-88) the colour is
    albedo x Light::GetShading (light.h:53-60: intensity * cos / distance^2), grey; without, the
    bare albedo. Values stay inside (.02, .98), the range LightIntegrator accepts."""
    s, point, normal, (rx, ry), rot = _room_hit(k, pose, w, h, with_normal=light is not None)
    value = room_albedo(point)
    if light is not None:
        intensity, lp = light
        pc = (s * rx, s * ry, s)                               # the point in the camera frame
        nc = [rot[0, j] * normal[0] + rot[1, j] * normal[1] + rot[2, j] * normal[2] for j in range(3)]   # R^T n
        to_light = [lp[j] - pc[j] for j in range(3)]
        dist2 = to_light[0] ** 2 + to_light[1] ** 2 + to_light[2] ** 2
        cos = (nc[0] * to_light[0] + nc[1] * to_light[1] + nc[2] * to_light[2]) / np.sqrt(dist2)
        value = value * np.clip(intensity * cos / dist2, 0.0, None)
    value = np.clip(value, 0.03, 0.97).astype(np.float32)
    color = np.empty((h, w, 3), dtype=np.float32)
    color[...] = value[..., None]
    return s.astype(np.float32), color


# ---- the march's step cap (tracer.cu:437-442): a hand-built volume whose rays take 500 steps -------------------
# A ray's step through an allocated block is max(voxel_length, trunc * sdf) (tracer.cu:425): a slab of blocks whose
# voxels all hold the small positive distance STEP_CAP_SDF (trunc * sdf < voxel_length) is marched one voxel length at
# a time, and a slab deeper than 500 voxel lengths along the ray ends the march at the cap: colour (1, 0, 0), depth 0.
# The slab is ALLOCATED by SetView itself with a truncation length of half its depth (the DDA of volume.cu:87-301
# requests every block within +-truncation of the depth pixel), then filled by step_cap_fill and raycast with the
# reference's truncation length. Camera-frame left half (x < 0): the deep slab, every ray capped. Right half: at
# STEP_CAP_SURFACE the distances turn negative, so those rays hit a surface after ~250 steps — both exits in one image.
STEP_CAP_SIZE = (64, 48)
STEP_CAP_INTRINSICS = (800.0, 800.0, 32.0, 24.0)
STEP_CAP_VOLUME = (8192, 2048)                    # main, excess
STEP_CAP_VOXEL, STEP_CAP_TRUNC = 0.002, 0.04      # 2 mm voxels (16 mm blocks), the reference's truncation (volume.cu:375)
STEP_CAP_ALLOC_DEPTH, STEP_CAP_ALLOC_TRUNC = 0.85, 0.55     # blocks from 0.30 to 1.40 m along every ray
STEP_CAP_SDF, STEP_CAP_BEHIND, STEP_CAP_SURFACE = 0.04, -0.1, 0.8


def step_cap_pose():
    from vulcan_amd import vk_types as T
    return T.Transform.translate(0.53, -0.31, 0.22) * T.Transform.rotate(0.9961947, 0.0, 0.0871557, 0.0)   # 10 degrees about y


def step_cap_fill(entries, voxels, pose):
    """Fills every ALLOCATED block of a volume (numpy views of the hash table and the pool, vk_types dtypes) in place:
    distance STEP_CAP_SDF with weight 1 and colour (.2, .4, .6) with weight 1 everywhere, STEP_CAP_BEHIND where the
    voxel centre lies in the camera's right half (x >= 0) deeper than STEP_CAP_SURFACE. Returns the allocated count."""
    alloc = np.nonzero(entries["data"] >= 0)[0]
    slots = entries["data"][alloc].astype(np.int64)
    origin = entries["block"]["origin"][alloc].astype(np.float64)            # (n, 3) block coordinates
    idx = np.arange(8, dtype=np.float64) + 0.5
    zz, yy, xx = np.meshgrid(idx, idx, idx, indexing="ij")                   # voxel index z*64 + y*8 + x
    local = np.stack([xx, yy, zz], axis=-1).reshape(512, 3)
    world = (origin[:, None, :] * 8.0 + local[None, :, :]) * STEP_CAP_VOXEL  # (n, 512, 3) voxel centres
    Tcw = pose.inverse_matrix().astype(np.float64)
    cam = world @ Tcw[:3, :3].T + Tcw[:3, 3]
    behind = (cam[..., 0] >= 0.0) & (cam[..., 2] > STEP_CAP_SURFACE)
    v = voxels.reshape(-1, 512)
    v["distance"][slots] = np.where(behind, np.float32(STEP_CAP_BEHIND), np.float32(STEP_CAP_SDF))
    v["distance_weight"][slots] = 1
    v["color"][slots] = np.array([0.2, 0.4, 0.6], dtype=np.float32)
    v["color_weight"][slots] = 1
    return len(alloc)
