"""ctypes / numpy mirrors of the PODs declared in include/vk.h.

Layouts follow the reference's types (sizes verified against the reference
headers): Voxel 20 B (voxel.h:43-49), Block 8 B (block.h:81-85),
HashEntry 16 B (hash.h:51-55), Patch 16 B (tracer.h:13-22), Projection 16 B
(projection.h:111-113), Transform 128 B (transform.h:168-170), Light 16 B.
"""
import ctypes as C

import numpy as np

VK_CTR_VISIBLE, VK_CTR_VOXEL_PTR, VK_CTR_EXCESS_PTR, VK_CTR_PATCHES = 0, 1, 2, 3
VK_CTR_REQUESTS, VK_CTR_DROPPED, VK_CTR_PENDING_ALL, VK_CTR_PENDING_EXCESS = 4, 5, 6, 7
VK_CTR_ROUNDS, VK_CTR_UNSETTLED, VK_CTR_CONTENDED, VK_CTR_PUBLIC = 8, 9, 10, 24
VK_RETRY_SLOTS, VK_RETRY_KEYS, VK_POSTED_SLOTS = 65536, 8192, 2048
# counters, two key sets, two slot lists, the posted buckets and their chains' last entries
VK_ABI_VERSION = 7                       # include/vk.h
VK_CTR_BANDED, VK_BANDS, VK_BAND_SLOTS = 20, 8, 16384
VK_CTR_COUNT = (VK_CTR_PUBLIC + 2 * 2 * VK_RETRY_SLOTS + 2 * VK_RETRY_KEYS + 2 * VK_POSTED_SLOTS
                + VK_BANDS + VK_BANDS * VK_BAND_SLOTS)
VK_TRACK_ABORTED = -1
VK_ERR_UNSUPPORTED = -2                  # include/vk.h vk_status
VISIBILITY_UNKNOWN, VISIBILITY_FALSE, VISIBILITY_TRUE = 0, 1, 2
ALLOC_NONE, ALLOC_MAIN, ALLOC_EXCESS = 0, 1, 2
BLOCK_RESOLUTION, BLOCK_VOXELS, PATCH_MAX_SIZE = 8, 512, 16

voxel_dtype = np.dtype([("distance", "<f4"), ("color", "<f4", (3,)),
                        ("distance_weight", "<i2"), ("color_weight", "<i2")])
block_dtype = np.dtype([("origin", "<i2", (3,)), ("pad", "<i2")])
hash_entry_dtype = np.dtype([("block", block_dtype), ("data", "<i4"), ("next", "<i4")])
patch_dtype = np.dtype([("origin", "<i2", (2,)), ("size", "<i2", (2,)), ("bounds", "<f4", (2,))])
assert voxel_dtype.itemsize == 20 and block_dtype.itemsize == 8
assert hash_entry_dtype.itemsize == 16 and patch_dtype.itemsize == 16


class Projection(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float)]

    @staticmethod
    def make(fx, fy, cx, cy):
        return Projection(np.float32(fx), np.float32(fy), np.float32(cx), np.float32(cy))


class Transform(C.Structure):
    """4x4 column-major matrix + cached inverse (transform.h). Host-side helpers
    follow the reference's float32 arithmetic (transform.h:62-159)."""
    _fields_ = [("m", C.c_float * 16), ("inv", C.c_float * 16)]

    @staticmethod
    def from_matrices(m, inv):
        t = Transform()
        mm = np.asarray(m, dtype=np.float32).reshape(4, 4)
        ii = np.asarray(inv, dtype=np.float32).reshape(4, 4)
        t.m[:] = mm.T.reshape(-1).tolist()     # column-major storage
        t.inv[:] = ii.T.reshape(-1).tolist()
        return t

    def matrix(self):
        return np.array(self.m[:], dtype=np.float32).reshape(4, 4).T.copy()

    def inverse_matrix(self):
        return np.array(self.inv[:], dtype=np.float32).reshape(4, 4).T.copy()

    @staticmethod
    def identity():
        return Transform.from_matrices(np.eye(4), np.eye(4))

    @staticmethod
    def translate(x, y, z):
        m = np.eye(4, dtype=np.float32)
        i = np.eye(4, dtype=np.float32)
        m[:3, 3] = np.array([x, y, z], dtype=np.float32)
        i[:3, 3] = -np.array([x, y, z], dtype=np.float32)
        return Transform.from_matrices(m, i)

    @staticmethod
    def rotate(w, x, y, z):
        """transform.h:108-136 Rotate(w, x, y, z), float32, same expression order."""
        f = np.float32
        w, x, y, z = f(w), f(x), f(y), f(z)
        m = np.zeros((4, 4), dtype=np.float32)
        m[0, 0] = f(1) - f(2) * (y * y + z * z)
        m[0, 1] = f(2) * (x * y - w * z)
        m[0, 2] = f(2) * (x * z + w * y)
        m[1, 0] = f(2) * (x * y + w * z)
        m[1, 1] = f(1) - f(2) * (x * x + z * z)
        m[1, 2] = f(2) * (y * z - w * x)
        m[2, 0] = f(2) * (x * z - w * y)
        m[2, 1] = f(2) * (y * z + w * x)
        m[2, 2] = f(1) - f(2) * (x * x + y * y)
        m[3, 3] = f(1)
        return Transform.from_matrices(m, m.T)

    @staticmethod
    def _matmul(a, b):
        # matrix.h:297-318: result(m,p) = 0; result += A(m,n) * B(n,p), n ascending
        out = np.zeros((4, 4), dtype=np.float32)
        for p in range(4):
            for m in range(4):
                acc = np.float32(0)
                for n in range(4):
                    acc = np.float32(acc + np.float32(a[m, n] * b[n, p]))
                out[m, p] = acc
        return out

    def __mul__(self, other):
        """transform.h:62-66: (A*B).m = A.m*B.m, (A*B).inv = B.inv*A.inv"""
        return Transform.from_matrices(
            Transform._matmul(self.matrix(), other.matrix()),
            Transform._matmul(other.inverse_matrix(), self.inverse_matrix()))

    def inverse(self):
        return Transform.from_matrices(self.inverse_matrix(), self.matrix())


class Light(C.Structure):
    _fields_ = [("intensity", C.c_float), ("position", C.c_float * 3)]

    @staticmethod
    def make(intensity, position):
        l = Light()
        l.intensity = intensity
        l.position[:] = [float(p) for p in position]
        return l


class Volume(C.Structure):
    _fields_ = [("voxels", C.c_void_p), ("hash_entries", C.c_void_p),
                ("free_voxel_blocks", C.c_void_p), ("allocation_types", C.c_void_p),
                ("allocation_blocks", C.c_void_p), ("block_visibility", C.c_void_p),
                ("visible_blocks", C.c_void_p), ("counters", C.c_void_p),
                ("main_block_count", C.c_int32), ("excess_block_count", C.c_int32),
                ("voxel_length", C.c_float), ("truncation_length", C.c_float),
                ("min_depth", C.c_float), ("max_depth", C.c_float)]


class Frame(C.Structure):
    _fields_ = [("depth", C.c_void_p), ("color", C.c_void_p), ("normals", C.c_void_p),
                ("width", C.c_int32), ("height", C.c_int32),
                ("color_width", C.c_int32), ("color_height", C.c_int32),
                ("depth_projection", Projection), ("color_projection", Projection),
                ("depth_to_world", Transform), ("depth_to_color", Transform), ("content_id", C.c_uint64)]


class Integrator(C.Structure):
    _fields_ = [("min_depth", C.c_float), ("max_depth", C.c_float),
                ("max_distance_weight", C.c_float), ("max_color_weight", C.c_float)]

    @staticmethod
    def default():
        return Integrator(0.1, 5.0, 16.0, 16.0)   # integrator.cu:7-13


class ViewBounds(C.Structure):
    _fields_ = [("scratch", C.c_void_p), ("bounds_width", C.c_int32), ("bounds_height", C.c_int32),
                ("min_depth", C.c_float), ("max_depth", C.c_float),
                ("valid", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("block_length", C.c_float), ("visible_blocks", C.c_void_p),
                ("projection", Projection), ("depth_to_world", Transform),
                ("counted_scratch", C.c_void_p), ("counted_width", C.c_int32), ("counted_height", C.c_int32),
                ("trace_launches", C.c_uint32), ("pad_", C.c_int32),
                ("late_host", C.c_void_p), ("last_depths", C.c_void_p), ("last_normals", C.c_void_p),
                ("last_width", C.c_int32), ("last_height", C.c_int32), ("last_projection", Projection),
                ("counted_stream", C.c_void_p)]


class ColorView(C.Structure):
    _fields_ = [("depths", C.c_void_p), ("normals", C.c_void_p), ("intensities", C.c_void_p),
                ("gradient_x", C.c_void_p), ("gradient_y", C.c_void_p),
                ("width", C.c_int32), ("height", C.c_int32), ("projection", Projection)]


class LightTerms(C.Structure):
    _fields_ = [("frame_mask", C.c_void_p), ("light", Light), ("frame_Tcd", Transform)]


class TrackPoll(C.Structure):
    _fields_ = [("host_state", C.c_void_p), ("chunk", C.c_int32), ("host_pose", C.c_void_p)]


class LightPrep(C.Structure):
    _fields_ = [("depth_threshold", C.c_float), ("mask", C.c_void_p), ("records", C.c_void_p),
                ("normals_out", C.c_void_p), ("capacity", C.c_int32),
                ("valid", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("depth", C.c_void_p),
                ("color", C.c_void_p), ("normals", C.c_void_p), ("prepared_threshold", C.c_float),
                ("depth_to_color", Transform), ("content_id", C.c_uint64)]


class RigExchange(C.Structure):
    """vk_rig_exchange (vk.h): the peers' areas of the rig's in-launch exchange."""
    _fields_ = [("areas", C.c_void_p * 8), ("rank", C.c_int32), ("world", C.c_int32), ("sequence", C.c_uint32)]


class RequestsAhead(C.Structure):
    """vk_requests_ahead (vk.h): the frame a request pass was made for ahead of its SetView"""
    _fields_ = [("counters", C.c_void_p), ("depth", C.c_void_p), ("prep", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32),
                ("depth_projection", Projection), ("depth_to_world", Transform), ("content_id", C.c_uint64),
                ("valid", C.c_int32), ("normals_made", C.c_int32), ("pose_on_device", C.c_int32), ("pad_", C.c_int32)]


class PyramidAhead(C.Structure):
    """vk_pyramid_ahead (vk.h): the images and the buffer a Track's pyramid was built from / into behind the raycast"""
    _fields_ = [("key_depths", C.c_void_p), ("key_normals", C.c_void_p), ("frame_depths", C.c_void_p), ("frame_normals", C.c_void_p),
                ("pyramid", C.c_void_p), ("key_width", C.c_int32), ("key_height", C.c_int32), ("frame_width", C.c_int32),
                ("frame_height", C.c_int32), ("valid", C.c_int32), ("pad_", C.c_int32)]


class TestHooks(C.Structure):
    """vk_test_hooks (vk.h)"""
    _fields_ = [("posted_capacity", C.c_int32), ("retry_capacity", C.c_int32), ("set_view_unfused", C.c_int32),
                ("force_loop_abort", C.c_int32), ("loop_grid_cap", C.c_int32), ("loop_cooperative", C.c_int32),
                ("force_normals_expiry", C.c_int32)]


class ColorPose(C.Structure):
    _fields_ = [("depth_to_world", Transform), ("Tcm", Transform)]


class Detector(C.Structure):
    _fields_ = [("radius", C.c_float), ("origin", C.c_float * 3), ("bounds", (C.c_float * 2) * 3),
                ("min_inlier_count", C.c_int32), ("bounds_use_own_axis", C.c_int32)]

    @staticmethod
    def default():
        """detector.cu:66-72,214-221: radius 2, origin 0, all intervals open, 100 inliers."""
        d = Detector()
        d.radius = 2.0
        d.min_inlier_count = 100
        for a in range(3):
            d.bounds[a][0], d.bounds[a][1] = 1.0, -1.0
        return d


class DetectState(C.Structure):
    _fields_ = [("filtered_count", C.c_int32), ("inlier_count", C.c_int32), ("detected", C.c_int32),
                ("reserved", C.c_int32), ("center", C.c_float * 3), ("limit", C.c_float),
                ("position", C.c_float * 3), ("squared_error", C.c_float)]


class IcpView(C.Structure):
    _fields_ = [("depths", C.c_void_p), ("normals", C.c_void_p),
                ("width", C.c_int32), ("height", C.c_int32), ("projection", Projection)]
