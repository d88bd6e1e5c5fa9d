"""ctypes loader + numpy host containers for the CPU oracle (liboracle.so).

TEST INFRASTRUCTURE ONLY (see oracle/oracle.h): imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg — never by vulcan_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from vulcan_amd import vk_types as T

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

POLICY_SERIAL, POLICY_MAXKEY = 0, 1


def build(force=False):
    if os.environ.get("ORACLE_LIBRARY"):
        return os.environ["ORACLE_LIBRARY"]
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    srcs.append(os.path.join(_HERE, "..", "include", "vk.h"))
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        # ORACLE_LIBRARY: another build of the same sources (the AddressSanitizer / UBSan build,
        # `make -C oracle asan`)
        so = os.environ.get("ORACLE_LIBRARY") or os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            so = build()
        _LIB = C.CDLL(so)
        _LIB.orc_icp_solve_update.restype = C.c_float
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class HostVolume:
    """Host mirror of vulcan::Volume's buffers (volume.h:89-116) as numpy arrays."""

    def __init__(self, main_block_count, excess_block_count, voxel_length=0.008,
                 truncation_length=0.04, depth_range=(0.1, 5.0)):
        n = main_block_count + excess_block_count
        self.main, self.excess, self.max = main_block_count, excess_block_count, n
        self.voxels = np.zeros(n * 512, dtype=T.voxel_dtype)
        self.hash_entries = np.zeros(n, dtype=T.hash_entry_dtype)
        self.free_voxel_blocks = np.zeros(n, dtype=np.int32)
        self.allocation_types = np.zeros(main_block_count, dtype=np.uint8)
        self.allocation_blocks = np.zeros(main_block_count, dtype=T.block_dtype)
        self.block_visibility = np.zeros(n, dtype=np.uint8)
        self.visible_blocks = np.zeros(n, dtype=np.int32)
        self.counters = np.zeros(T.VK_CTR_COUNT, dtype=np.int32)
        self.voxel_length = voxel_length
        self.truncation_length = truncation_length
        self.depth_range = depth_range
        lib().orc_volume_initialize(C.byref(self.desc()))

    def desc(self):
        d = T.Volume()
        d.voxels = self.voxels.ctypes.data
        d.hash_entries = self.hash_entries.ctypes.data
        d.free_voxel_blocks = self.free_voxel_blocks.ctypes.data
        d.allocation_types = self.allocation_types.ctypes.data
        d.allocation_blocks = self.allocation_blocks.ctypes.data
        d.block_visibility = self.block_visibility.ctypes.data
        d.visible_blocks = self.visible_blocks.ctypes.data
        d.counters = self.counters.ctypes.data
        d.main_block_count, d.excess_block_count = self.main, self.excess
        d.voxel_length, d.truncation_length = self.voxel_length, self.truncation_length
        d.min_depth, d.max_depth = self.depth_range
        return d

    @property
    def visible_count(self):
        return int(self.counters[T.VK_CTR_VISIBLE])

    def visible(self):
        return self.visible_blocks[:self.visible_count]

    # volume.cu:430-437 and its four stages
    def set_view(self, frame, policy=POLICY_SERIAL):
        lib().orc_volume_set_view(C.byref(self.desc()), C.byref(frame.desc()), policy)

    def reset_block_visibility(self):
        lib().orc_volume_reset_block_visibility(C.byref(self.desc()))

    def create_allocation_requests(self, frame, policy=POLICY_SERIAL):
        lib().orc_volume_create_allocation_requests(
            C.byref(self.desc()), _p(frame.depth), frame.width, frame.height,
            C.byref(frame.depth_projection), C.byref(frame.depth_to_world), policy)

    def handle_allocation_requests(self):
        lib().orc_volume_handle_allocation_requests(C.byref(self.desc()))

    def update_block_visibility(self, frame):
        tdw = frame.depth_to_world.inverse()
        lib().orc_volume_update_block_visibility(
            C.byref(self.desc()), frame.width, frame.height,
            C.byref(frame.depth_projection), C.byref(tdw))


class HostFrame:
    """Host mirror of vulcan::Frame (frame.h:11-32)."""

    def __init__(self, depth, depth_projection, depth_to_world=None, color=None,
                 normals=None, color_projection=None, depth_to_color=None):
        self.depth = np.ascontiguousarray(depth, dtype=np.float32)
        self.height, self.width = self.depth.shape
        self.color = None if color is None else np.ascontiguousarray(color, dtype=np.float32)
        self.normals = None if normals is None else np.ascontiguousarray(normals, dtype=np.float32)
        self.depth_projection = depth_projection
        self.color_projection = color_projection or depth_projection
        self.depth_to_world = depth_to_world or T.Transform.identity()
        self.depth_to_color = depth_to_color or T.Transform.identity()

    def desc(self):
        d = T.Frame()
        d.depth = self.depth.ctypes.data
        d.color = None if self.color is None else self.color.ctypes.data
        d.normals = None if self.normals is None else self.normals.ctypes.data
        d.width, d.height = self.width, self.height
        if self.color is not None:
            d.color_height, d.color_width = self.color.shape[0], self.color.shape[1]
        d.depth_projection, d.color_projection = self.depth_projection, self.color_projection
        d.depth_to_world, d.depth_to_color = self.depth_to_world, self.depth_to_color
        return d

    def compute_normals(self):
        self.normals = compute_normals(self.depth, self.depth_projection)
        return self.normals

    def downsample(self):
        """Frame::Downsample (frame.cpp:38-58): nearest depth / normals, 2x2 box colour,
        intrinsics / 2 (Vector2f / 2 multiplies by 1/2, matrix.h:279-295)."""
        f32 = np.float32

        def half(k):
            return T.Projection.make(f32(k.fx) * f32(0.5), f32(k.fy) * f32(0.5),
                                     f32(k.cx) * f32(0.5), f32(k.cy) * f32(0.5))
        return HostFrame(downsample(self.depth, True), half(self.depth_projection), self.depth_to_world,
                         None if self.color is None else downsample(self.color, False),
                         None if self.normals is None else downsample(self.normals, True),
                         half(self.color_projection), self.depth_to_color)


def set_threads(n):
    lib().orc_set_threads(int(n))


# ---- integrators ---------------------------------------------------------

def integrate_depth(vol, frame, params=None):
    params = params or T.Integrator.default()
    lib().orc_integrate_depth(C.byref(vol.desc()), C.byref(params), C.byref(frame.desc()))


def integrate_color(vol, frame, params=None):
    params = params or T.Integrator.default()
    lib().orc_integrate_color(C.byref(vol.desc()), C.byref(params), C.byref(frame.desc()))


def light_frame_mask(frame, depth_threshold=0.2):
    mask = np.zeros((frame.height, frame.width), dtype=np.float32)
    lib().orc_light_compute_frame_mask(C.byref(frame.desc()), C.c_float(depth_threshold), _p(mask))
    return mask


def integrate_light_color(vol, frame, light, mask, params=None):
    params = params or T.Integrator.default()
    lib().orc_integrate_light_color(C.byref(vol.desc()), C.byref(params), C.byref(light),
                                    _p(mask), C.byref(frame.desc()))


# ---- tracer --------------------------------------------------------------

def compute_patches(indices, entries, Tcw, projection, block_length, min_depth, max_depth,
                    image_width, image_height, bounds_width=80, bounds_height=60, capacity=262144):
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    patches = np.zeros(capacity, dtype=T.patch_dtype)
    count = np.zeros(1, dtype=np.int32)
    lib().orc_trace_compute_patches(
        _p(indices), _p(entries), C.byref(Tcw), C.byref(projection), C.c_float(block_length),
        C.c_float(min_depth), C.c_float(max_depth), len(indices), image_width, image_height,
        bounds_width, bounds_height, _p(patches), capacity, _p(count))
    return patches[:min(int(count[0]), capacity)].copy(), int(count[0])


def compute_bounds(patches, bounds_width=80, bounds_height=60):
    bounds = np.zeros((bounds_height, bounds_width, 2), dtype=np.float32)
    lib().orc_trace_reset_bounds(_p(bounds), bounds_width * bounds_height)
    patches = np.ascontiguousarray(patches)
    lib().orc_trace_compute_bounds(_p(patches), _p(bounds), bounds_width, len(patches))
    return bounds


def compute_points(vol, bounds, Twc, projection, image_width, image_height, want_steps=False):
    depths = np.zeros((image_height, image_width), dtype=np.float32)
    colors = np.zeros((image_height, image_width, 3), dtype=np.float32)
    steps = np.zeros((image_height, image_width), dtype=np.int32) if want_steps else None
    bh, bw = bounds.shape[:2]
    block_length = np.float32(8) * np.float32(vol.voxel_length)
    lib().orc_trace_compute_points(
        _p(vol.hash_entries), _p(vol.voxels), _p(bounds), vol.main, C.c_float(block_length),
        C.c_float(vol.voxel_length), C.c_float(vol.truncation_length), C.byref(Twc),
        C.byref(projection), _p(depths), _p(colors), image_width, image_height, bw, bh, _p(steps))
    return (depths, colors, steps) if want_steps else (depths, colors)


def compute_normals(depths, projection):
    depths = np.ascontiguousarray(depths, dtype=np.float32)
    h, w = depths.shape
    normals = np.zeros((h, w, 3), dtype=np.float32)
    lib().orc_frame_compute_normals(_p(depths), C.byref(projection), _p(normals), w, h)
    return normals


def filter_depths(depths):
    depths = np.ascontiguousarray(depths, dtype=np.float32)
    h, w = depths.shape
    out = np.zeros_like(depths)
    lib().orc_frame_filter_depths(w, h, _p(depths), _p(out))
    return out


def trace(vol, frame, depth_range=(0.1, 5.0), bounds_width=80, bounds_height=60, want_steps=False):
    """tracer.cpp:41-47 Tracer::Trace on the host: returns depth, colour, normals."""
    block_length = np.float32(8) * np.float32(vol.voxel_length)
    patches, count = compute_patches(vol.visible(), vol.hash_entries, frame.depth_to_world.inverse(),
                                     frame.depth_projection, block_length, depth_range[0],
                                     depth_range[1], frame.width, frame.height,
                                     bounds_width, bounds_height)
    bounds = compute_bounds(patches, bounds_width, bounds_height)
    out = compute_points(vol, bounds, frame.depth_to_world, frame.depth_projection,
                         frame.width, frame.height, want_steps)
    normals = compute_normals(out[0], frame.depth_projection)
    return (out[0], out[1], normals, bounds) + ((out[2],) if want_steps else ())


# ---- image ---------------------------------------------------------------

def downsample(img, nearest):
    img = np.ascontiguousarray(img, dtype=np.float32)
    h, w = img.shape[:2]
    if img.ndim == 2:
        out = np.zeros((h // 2, w // 2), dtype=np.float32)
        lib().orc_image_downsample(w, h, _p(img), _p(out), int(nearest))
    else:
        out = np.zeros((h // 2, w // 2, 3), dtype=np.float32)
        lib().orc_color_image_downsample(w, h, _p(img), _p(out), int(nearest))
    return out


# ---- ICP -----------------------------------------------------------------

def _view(depths, normals, projection):
    v = T.IcpView()
    v.depths, v.normals = depths.ctypes.data, normals.ctypes.data
    v.height, v.width = depths.shape
    v.projection = projection
    return v


def icp_residuals(key, frame):
    kv = _view(key.depth, key.normals, key.depth_projection)
    fv = _view(frame.depth, frame.normals, frame.depth_projection)
    out = np.zeros((frame.height, frame.width), dtype=np.float32)
    lib().orc_icp_compute_residuals(C.byref(kv), C.byref(key.depth_to_world), C.byref(fv),
                                    C.byref(frame.depth_to_world), _p(out))
    return out


def icp_jacobian(key, frame, translation_enabled=True):
    kv = _view(key.depth, key.normals, key.depth_projection)
    fv = _view(frame.depth, frame.normals, frame.depth_projection)
    out = np.zeros((frame.height, frame.width, 6), dtype=np.float32)
    lib().orc_icp_compute_jacobian(C.byref(kv), C.byref(key.depth_to_world), C.byref(fv),
                                   C.byref(frame.depth_to_world), int(translation_enabled), _p(out))
    return out


def icp_system(key, frame, translation_enabled=True):
    kv = _view(key.depth, key.normals, key.depth_projection)
    fv = _view(frame.depth, frame.normals, frame.depth_projection)
    hessian = np.zeros(21, dtype=np.float64)
    gradient = np.zeros(6, dtype=np.float64)
    lib().orc_icp_compute_system(C.byref(kv), C.byref(key.depth_to_world), C.byref(fv),
                                 C.byref(frame.depth_to_world), int(translation_enabled),
                                 _p(hessian), _p(gradient))
    return hessian, gradient


def ldlt_solve(A, b, pivoted=False):
    """x with A x = b: the unpivoted LDL^T the device runs, or Eigen's pivoted one (orc_ldlt_solve_pivoted)."""
    A = np.ascontiguousarray(A, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    n = len(b)
    x = np.zeros(n, dtype=np.float32)
    (lib().orc_ldlt_solve_pivoted if pivoted else lib().orc_ldlt_solve)(n, _p(A), _p(b), _p(x))
    return x


def icp_solve_update(hessian_packed, gradient, Twc, translation_enabled=True):
    h = np.ascontiguousarray(hessian_packed, dtype=np.float32)
    g = np.ascontiguousarray(gradient, dtype=np.float32)
    out = T.Transform.from_matrices(Twc.matrix(), Twc.inverse_matrix())
    update = np.zeros(6, dtype=np.float32)
    norm = lib().orc_icp_solve_update(_p(h), _p(g), int(translation_enabled), C.byref(out), _p(update))
    return out, update, float(norm)


def icp_track(key, frame, max_iterations=20, translation_enabled=True):
    """Tracker::Track for DepthTracker (tracker.cpp:53-63,124-163): Gauss-Newton until
    max_iterations or |update| < 1e-6. Updates frame.depth_to_world; returns (pose, iterations run)."""
    pose, it = frame.depth_to_world, 0
    while it < max_iterations:
        frame.depth_to_world = pose
        Hs, g = icp_system(key, frame, translation_enabled)
        pose, _, norm = icp_solve_update(Hs, g, pose, translation_enabled)
        it += 1
        if norm < 1e-6:
            break
    frame.depth_to_world = pose
    return pose, it


def pyramid_track(key, frame):
    """PyramidTracker<DepthTracker>::Track (pyramid_tracker.cpp:52-90): half resolution
    (15 iterations) then full resolution (20); the quarter level is built but unused."""
    half_frame, half_key = frame.downsample(), key.downsample()
    icp_track(half_key, half_frame, 15, True)
    frame.depth_to_world = half_frame.depth_to_world
    return icp_track(key, frame, 20, True)


def extract_mesh(vol, all_allocated=False, interpolate=True):
    """orc_extract_mesh: (points [n, 3] float32, faces [m, 3] int32, cubes skipped)."""
    counts = np.zeros(4, dtype=np.int32)
    pc, fc = 1, 1
    for _ in range(2):
        points = np.zeros((pc, 3), dtype=np.float32)
        faces = np.zeros((fc, 3), dtype=np.int32)
        rc = lib().orc_extract_mesh(C.byref(vol.desc()), int(all_allocated), int(interpolate), _p(points), pc,
                                    _p(faces), fc, _p(counts))
        if rc == 0:
            return points[:counts[0]].copy(), faces[:counts[1]].copy(), int(counts[2])
        pc, fc = max(int(counts[0]), 1), max(int(counts[1]), 1)
    raise RuntimeError("orc_extract_mesh: capacities did not settle")


def mc_triangles(state):
    edges = np.zeros(15, dtype=np.int8)
    n = lib().orc_mc_triangles(int(state), _p(edges))
    return edges[:3 * n].reshape(n, 3).copy()


def detect(points, params):
    """orc_detect: returns (DetectState, inlier points [m,3])."""
    pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
    inliers = np.zeros((max(len(pts), 1), 3), dtype=np.float32)
    state = T.DetectState()
    lib().orc_detect(C.byref(params), _p(pts), len(pts), _p(inliers), C.byref(state))
    return state, inliers[: state.inlier_count].copy()


# ---- colour tracker ------------------------------------------------------

def color_convert(colors):
    c = np.ascontiguousarray(colors, dtype=np.float32)
    out = np.zeros(c.shape[:-1], dtype=np.float32)
    lib().orc_color_image_convert(out.size, _p(c), _p(out))
    return out


def image_gradients(img):
    img = np.ascontiguousarray(img, dtype=np.float32)
    h, w = img.shape
    gx, gy = np.zeros_like(img), np.zeros_like(img)
    lib().orc_image_gradients(w, h, _p(img), _p(gx), _p(gy))
    return gx, gy


class ColorSide:
    """What ColorTracker keeps per frame (color_tracker.h:38-44): depth, normals,
    intensities and (frame side) their gradients, with the colour intrinsics."""

    def __init__(self, frame, with_gradients):
        self.frame = frame
        self.depth = np.ascontiguousarray(frame.depth, dtype=np.float32)
        self.normals = np.ascontiguousarray(frame.normals, dtype=np.float32)
        self.intensities = color_convert(frame.color)
        self.gx, self.gy = image_gradients(self.intensities) if with_gradients else (None, None)

    def view(self):
        v = T.ColorView()
        v.depths, v.normals, v.intensities = _p(self.depth).value, _p(self.normals).value, _p(self.intensities).value
        v.gradient_x = _p(self.gx).value if self.gx is not None else None
        v.gradient_y = _p(self.gy).value if self.gy is not None else None
        v.height, v.width = self.depth.shape
        v.projection = self.frame.color_projection
        return v


def color_tcm(key_frame, frame):
    """color_tracker.cu:312-320"""
    key_Tcw = key_frame.depth_to_color * key_frame.depth_to_world.inverse()
    frame_Tcw = frame.depth_to_color * frame.depth_to_world.inverse()
    return frame_Tcw * key_Tcw.inverse()


def color_residuals(key, frm, Tcm):
    out = np.zeros(key.depth.shape, dtype=np.float32)
    kv, fv = key.view(), frm.view()
    lib().orc_color_tracker_compute_residuals(C.byref(kv), C.byref(fv), C.byref(Tcm), _p(out))
    return out


def color_jacobian(key, frm, Tcm, translation_enabled=True):
    out = np.zeros(key.depth.shape + (6,), dtype=np.float32)
    kv, fv = key.view(), frm.view()
    lib().orc_color_tracker_compute_jacobian(C.byref(kv), C.byref(fv), C.byref(Tcm), int(translation_enabled), _p(out))
    return out


def color_system(key, frm, Tcm, translation_enabled=True):
    h = np.zeros(21, dtype=np.float64)
    g = np.zeros(6, dtype=np.float64)
    kv, fv = key.view(), frm.view()
    lib().orc_color_tracker_compute_system(C.byref(kv), C.byref(fv), C.byref(Tcm), int(translation_enabled), _p(h), _p(g))
    return h, g


def color_solve_update(hessian_packed, gradient, frame_Tcd, key_Twc, pose, translation_enabled=True):
    h = np.ascontiguousarray(hessian_packed, dtype=np.float32)
    g = np.ascontiguousarray(gradient, dtype=np.float32)
    update = np.zeros(6, dtype=np.float32)
    lib().orc_color_tracker_solve_update.restype = C.c_float
    norm = lib().orc_color_tracker_solve_update(_p(h), _p(g), int(translation_enabled), C.byref(frame_Tcd),
                                                C.byref(key_Twc), C.byref(pose), _p(update))
    return update, float(norm)


# ---- light tracker -------------------------------------------------------

def light_terms(frame, light, mask):
    """vk_light_terms for `frame`; `mask` must stay alive while the struct is used."""
    t = T.LightTerms()
    t.frame_mask = _p(mask).value
    t.light = light
    t.frame_Tcd = frame.depth_to_color
    return t


def light_residuals(key, frm, terms, Tcm):
    out = np.zeros(key.depth.shape, dtype=np.float32)
    kv, fv = key.view(), frm.view()
    lib().orc_light_tracker_compute_residuals(C.byref(kv), C.byref(fv), C.byref(terms), C.byref(Tcm), _p(out))
    return out


def light_jacobian(key, frm, terms, Tcm, translation_enabled=True):
    out = np.zeros(key.depth.shape + (6,), dtype=np.float32)
    kv, fv = key.view(), frm.view()
    lib().orc_light_tracker_compute_jacobian(C.byref(kv), C.byref(fv), C.byref(terms), C.byref(Tcm),
                                             int(translation_enabled), _p(out))
    return out


def light_system(key, frm, terms, Tcm, translation_enabled=True):
    h = np.zeros(21, dtype=np.float64)
    g = np.zeros(6, dtype=np.float64)
    kv, fv = key.view(), frm.view()
    lib().orc_light_tracker_compute_system(C.byref(kv), C.byref(fv), C.byref(terms), C.byref(Tcm),
                                           int(translation_enabled), _p(h), _p(g))
    return h, g


def light_track(key, frame, light, max_iterations=20, depth_threshold=0.2, translation_enabled=True):
    """Tracker::Track for LightTracker (tracker.cpp:53-63 with light_tracker.cpp:34-41 as BeginSolve:
    the frame mask and both intensity images are made once, then Gauss-Newton until max_iterations
    or |update| < 1e-6). Updates frame.depth_to_world; returns (pose, iterations run)."""
    key_side, frame_side = ColorSide(key, False), ColorSide(frame, True)
    mask = light_frame_mask(frame, depth_threshold)
    key_Twc = (key.depth_to_color * key.depth_to_world.inverse()).inverse()
    p = T.ColorPose()
    p.depth_to_world = frame.depth_to_world
    lib().orc_color_tracker_tcm(C.byref(frame.depth_to_color), C.byref(key_Twc), C.byref(p))
    it = 0
    while it < max_iterations:
        h, g = light_system(key_side, frame_side, light_terms(frame, light, mask), p.Tcm, translation_enabled)
        _, norm = color_solve_update(h, g, frame.depth_to_color, key_Twc, p, translation_enabled)
        it += 1
        if norm < 1e-6:
            break
    frame.depth_to_world = T.Transform.from_buffer_copy(bytes(p.depth_to_world))
    return frame.depth_to_world, it
