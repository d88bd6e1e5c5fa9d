"""ctypes bindings of libvk_hip.so (include/vk.h) and thin host classes that mirror
the reference's Volume / Integrator / Tracer / Frame surface for tests and
bench.py. Device memory lives in torch tensors (plumbing only): every compute
call goes through the C ABI on the HIP stream torch is currently using.

There is NO CPU fallback: loading fails loudly when the library is missing, and
every wrapper raises VkError on a non-zero return code.
"""
import ctypes as C
import os

import numpy as np

from . import vk_types as T

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libvk_hip.so")
_LIB = None

# name -> argtypes, in the order of include/vk.h (restype is int unless noted)
_P, _I, _F, _SZ = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_PP = C.POINTER(C.c_void_p)
_SIGNATURES = {
    "vk_error_string": ([_I], C.c_char_p),
    "vk_version": ([], _I),
    "vk_abi_version": ([], _I),
    "vk_abi_check": ([_I, _SZ, _SZ, _I], _I),
    "vk_device_count": ([C.POINTER(_I)], _I),
    "vk_set_device": ([_I], _I),
    "vk_device_name": ([C.c_char_p, _SZ], _I),
    "vk_stream_create": ([_PP], _I),
    "vk_stream_destroy": ([_P], _I),
    "vk_stream_synchronize": ([_P], _I),
    "vk_malloc": ([_PP, _SZ], _I),
    "vk_free": ([_P], _I),
    "vk_memcpy_h2d": ([_P, _P, _SZ, _P], _I),
    "vk_memcpy_d2h": ([_P, _P, _SZ, _P], _I),
    "vk_memcpy_d2d": ([_P, _P, _SZ, _P], _I),
    "vk_memset": ([_P, _I, _SZ, _P], _I),
    "vk_malloc_host": ([_PP, _SZ], _I),
    "vk_free_host": ([_P], _I),
    "vk_event_create": ([_PP], _I),
    "vk_event_destroy": ([_P], _I),
    "vk_event_record": ([_P, _P], _I),
    "vk_event_elapsed_ms": ([_P, _P, C.POINTER(_F)], _I),
    "vk_memcpy_h2d_async": ([_P, _P, _SZ, _P], _I),
    "vk_event_create_ordering": ([_PP, _I], _I),
    "vk_stream_wait_event": ([_P, _P], _I),
    "vk_event_synchronize": ([_P], _I),
    "vk_volume_initialize": ([_P, _P], _I),
    "vk_volume_reset_block_visibility": ([_P, _P], _I),
    "vk_volume_create_allocation_requests": ([_P, _P, _I, _I, _P, _P, _P], _I),
    "vk_volume_handle_allocation_requests": ([_P, _P], _I),
    "vk_volume_update_block_visibility": ([_P, _I, _I, _P, _P, _P], _I),
    "vk_volume_set_view": ([_P, _P, _P], _I),
    "vk_volume_read_counters_sync": ([_P, _P, _P], _I),
    "vk_integrate_depth": ([_P, _P, _P, _P], _I),
    "vk_integrate_color": ([_P, _P, _P, _P], _I),
    "vk_integrate_depth_color": ([_P, _P, _P, _P], _I),
    "vk_light_compute_frame_mask": ([_P, _F, _P, _P], _I),
    "vk_integrate_light_color": ([_P, _P, _P, _P, _P, _P], _I),
    "vk_integrate_depth_light": ([_P, _P, _P, _P, _P, _P], _I),
    "vk_trace_compute_patches": ([_P, _P, _P, _P, _F, _F, _F, _I, _P, _I, _I, _I, _I, _P, _I, _P, _P], _I),
    "vk_trace_compute_bounds": ([_P, _P, _I, _I, _P, _P], _I),
    "vk_trace_reset_bounds": ([_P, _I, _P], _I),
    "vk_trace_compute_block_bounds": ([_P, _P, _P, _P, _F, _F, _F, _I, _P, _I, _I, _I, _I, _P, _P], _I),
    "vk_trace_compute_points": ([_P, _P, _P, _I, _F, _F, _F, _P, _P, _P, _P, _I, _I, _I, _I, _P], _I),
    "vk_frame_compute_normals": ([_P, _P, _P, _I, _I, _P], _I),
    "vk_frame_filter_depths": ([_I, _I, _P, _P, _P], _I),
    "vk_trace_bounds_floats": ([_I, _I], _SZ),
    "vk_trace": ([_P, _P, _F, _F, _P, _I, _I, _P, _P, _P, _P], _I),
    "vk_light_prepare": ([_P, _F, _P, _P, _P], _I),
    "vk_integrate_ahead": ([_P, _P, _P, _I, _P, _P, _P, _P, _P], _I),
    "vk_integrate_time_next": ([_P, _P], _I),
    "vk_trace_ahead": ([_P, _P, _P, _P, _P, _P, _P], _I),
    "vk_image_downsample": ([_I, _I, _P, _P, _I, _P], _I),
    "vk_color_image_downsample": ([_I, _I, _P, _P, _I, _P], _I),
    "vk_icp_compute_residuals": ([_P, _P, _P, _P, _P, _P], _I),
    "vk_icp_compute_jacobian": ([_P, _P, _P, _P, _I, _P, _P], _I),
    "vk_icp_workspace_floats": ([_I, _I], _SZ),
    "vk_icp_compute_system": ([_P, _P, _P, _P, _P, _I, _P, _P, _P, _P], _I),
    "vk_icp_solve_update": ([_P, _P, _I, _P, _P, _P, _P], _I),
    "vk_icp_track": ([_P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P], _I),
    "vk_transform_upload": ([_P, _P, _P], _I),
    "vk_icp_pyramid_floats": ([_I, _I, _I, _I], _SZ),
    "vk_icp_pyramid_track": ([_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P], _I),
    "vk_icp_pyramid_track_frame": ([_P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P], _I),
    "vk_track_wait": ([_P, _P], _I),
    "vk_reduce_nothing": ([_P, _I, _P, _P], _I),
    "vk_rig_area_bytes": ([], _SZ),
    "vk_icp_track_rig": ([_P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P], _I),
    "vk_color_tracker_begin": ([_P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P], _I),
    "vk_frame_downsample": ([_P, _P, _P, _P, _P], _I),
    "vk_volume_set_view_prepare": ([_P, _P, _P, _P], _I),
    "vk_volume_set_view_rounds": ([_P, _P, _P, _I, _P], _I),
    "vk_volume_set_view_rounds_split": ([_P, _P, _P, _I, _P, _P, _P], _I),
    "vk_trace_ahead_requests": ([_P, _P, _P, _P, _P, _P, _P, _P, _P, _P], _I),
    "vk_volume_set_view_rounds_ahead": ([_P, _P, _P, _I, _P, _P], _I),
    "vk_requests_ahead_cancel": ([_P, _P, _I, _P], _I),
    "vk_trace_normals_settle": ([_P, _P], _I),
    "vk_light_prepared": ([_P, _P, C.c_float], _I),
    "vk_color_image_convert": ([_I, _P, _P, _P], _I),
    "vk_image_gradients": ([_I, _I, _P, _P, _P, _P], _I),
    "vk_color_tracker_compute_residuals": ([_P, _P, _P, _P, _P], _I),
    "vk_color_tracker_compute_jacobian": ([_P, _P, _P, _I, _P, _P], _I),
    "vk_color_tracker_compute_system": ([_P, _P, _P, _P, _I, _P, _P, _P, _P], _I),
    "vk_color_tracker_solve_update": ([_P, _P, _I, _P, _P, _P, _P, _P, _P], _I),
    "vk_color_tracker_track": ([_P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P], _I),
    "vk_light_tracker_compute_residuals": ([_P, _P, _P, _P, _P, _P], _I),
    "vk_light_tracker_compute_jacobian": ([_P, _P, _P, _P, _I, _P, _P], _I),
    "vk_light_tracker_compute_system": ([_P, _P, _P, _P, _P, _I, _P, _P, _P, _P], _I),
    "vk_light_tracker_track": ([_P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P], _I),
    "vk_compact_workspace_bytes": ([C.c_int32], _SZ),
    "vk_compact_offsets": ([_P, C.c_int32, _P, _P, _P, _P], _I),
    "vk_extract_workspace_bytes": ([C.c_int32, C.c_int32], _SZ),
    "vk_extract_mesh": ([_P, _I, _I, _P, C.c_int32, _P, C.c_int32, _P, _P, _P], _I),
    "vk_detect_workspace_bytes": ([C.c_int32], _SZ),
    "vk_detect_filter": ([_P, _P, C.c_int32, _P, _P, _P, _P], _I),
    "vk_detect": ([_P, _P, C.c_int32, _P, _P, _P, _P], _I),
    "vk_test_hooks_set": ([_P], _I),
    "vk_test_hooks_get": ([_P], _I),
    "vk_test_hooks_loop_count": ([_P, _P, _P], _I),
    "vk_trace_ahead_pyramid": ([_P, _P, _P, _P, _P, _P, _P, _P, _P, _P], _I),
    "vk_volume_requests_at_device_pose": ([_P, _P, _P, _P, _P, _P], _I),
    "vk_volume_set_view_at_device_pose": ([_P, _P, _P, _P, _I, _P], _I),
    "vk_icp_pyramid_track_built": ([_P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P], _I),
}
EXPORTS = tuple(_SIGNATURES)
_REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p)   # vk_icp_reduce_fn


class test_hooks:
    """`with api.test_hooks(set_view_unfused=1): ...` — vk_test_hooks (vk.h) for the duration of the block: the
    test suites' switches (a small posted / retry list, the three-launch SetView, a forced loop abort, a capped
    loop grid) and the cooperative-launch option. The library reads no environment variable on a call path."""

    def __init__(self, **fields):
        self.fields = fields

    def __enter__(self):
        self.saved = T.TestHooks()
        check(lib().vk_test_hooks_get(C.byref(self.saved)), "vk_test_hooks_get")
        new = T.TestHooks.from_buffer_copy(bytes(self.saved))
        for name, value in self.fields.items():
            setattr(new, name, int(value))
        check(lib().vk_test_hooks_set(C.byref(new)), "vk_test_hooks_set")
        return self

    def __exit__(self, *exc):
        check(lib().vk_test_hooks_set(C.byref(self.saved)), "vk_test_hooks_set")
        return False


class TrackAborted(RuntimeError):
    """A Gauss-Newton loop kernel gave up waiting for the other workgroups (vk.h VK_TRACK_ABORTED)."""


class VkError(RuntimeError):
    pass


def lib():
    """Load libvk_hip.so. torch is imported first so the HIP runtime it bundles
    (libamdhip64.so.7) is the one the library binds to."""
    global _LIB, LIB_PATH
    if _LIB is None:
        # VK_HIP_LIBRARY selects another build of the same ABI (the debug build that syncs and
        # checks after every launch, libvk_hip_debug.so, or an experiment) — never a fallback
        LIB_PATH = os.environ.get("VK_HIP_LIBRARY") or LIB_PATH
        if not os.path.exists(LIB_PATH):
            raise VkError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)")
        import torch  # noqa: F401  (loads the HIP runtime)
        handle = C.CDLL(LIB_PATH)
        for name, (argtypes, restype) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.argtypes, fn.restype = argtypes, restype
        # the structs and sizes of vk_types.py against the ones the library was compiled with (vk.h VK_ABI_VERSION)
        if handle.vk_abi_check(T.VK_ABI_VERSION, C.sizeof(T.Volume), C.sizeof(T.Frame), T.VK_CTR_COUNT) != 0:
            raise VkError(f"{LIB_PATH} has binary interface version {handle.vk_abi_version()}, these bindings were written "
                          f"for {T.VK_ABI_VERSION} (or a struct / VK_CTR_COUNT differs): rebuild the library")
        _LIB = handle
    return _LIB


def check(code, what=""):
    if code != 0:
        raise VkError(f"{what}: {lib().vk_error_string(code).decode()} [{code}]")


def _ptr(t):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def _ref(s):
    return C.byref(s)


def stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev_bytes(n, device):
    import torch
    return torch.empty(int(n), dtype=torch.uint8, device=device)


def to_numpy(t, dtype):
    """Device byte tensor -> numpy structured array."""
    return np.frombuffer(t.cpu().numpy().tobytes(), dtype=dtype).copy()


_STAMP = [0]


def _next_stamp():
    _STAMP[0] += 1
    return _STAMP[0]


class Frame:
    """vulcan::Frame (frame.h:11-32) with images in device memory.

    vk_frame.content_id (which content the images hold) = the stamp of the last change this class
    saw — a tensor assigned to .depth / .color / .normals, compute_normals, filter_depths, a raycast
    into the frame — mixed with the tensors' own torch version counters (in-place torch ops).
    A kernel of your own that writes through data_ptr() is invisible: call touch() after it."""

    def __setattr__(self, name, value):
        if name in ("depth", "color", "normals"):
            object.__setattr__(self, "_stamp", _next_stamp())
        object.__setattr__(self, name, value)

    def touch(self):
        self._stamp = _next_stamp()

    def content_id(self):
        h = self._stamp
        for t in (self.depth, self.color, self.normals):
            h = (h * 0x9E3779B97F4A7C15 + (0 if t is None else (t._version + 1) * 0xC2B2AE3D27D4EB4F + t.data_ptr())) & (2 ** 64 - 1)
        return h | 1

    def __init__(self, depth, depth_projection, depth_to_world=None, color=None, normals=None,
                 color_projection=None, depth_to_color=None, device="cuda"):
        import torch
        self.device = device
        self.depth = torch.as_tensor(np.ascontiguousarray(depth, dtype=np.float32)).to(device) \
            if not hasattr(depth, "data_ptr") else depth
        self.height, self.width = self.depth.shape
        self.color = None
        self.normals = None
        if color is not None:
            self.color = torch.as_tensor(np.ascontiguousarray(color, dtype=np.float32)).to(device) \
                if not hasattr(color, "data_ptr") else color
        if normals is not None:
            self.normals = torch.as_tensor(np.ascontiguousarray(normals, dtype=np.float32)).to(device) \
                if not hasattr(normals, "data_ptr") else normals
        self.depth_projection = depth_projection
        self.color_projection = color_projection or depth_projection
        self.depth_to_world = depth_to_world or T.Transform.identity()
        self.depth_to_color = depth_to_color or T.Transform.identity()

    def desc(self):
        d = T.Frame()
        d.depth = self.depth.data_ptr()
        d.color = None if self.color is None else self.color.data_ptr()
        d.normals = None if self.normals is None else self.normals.data_ptr()
        d.width, d.height = self.width, self.height
        if self.color is not None:
            d.color_height, d.color_width = self.color.shape[0], self.color.shape[1]
        d.depth_projection, d.color_projection = self.depth_projection, self.color_projection
        d.depth_to_world, d.depth_to_color = self.depth_to_world, self.depth_to_color
        d.content_id = self.content_id()
        return d

    def compute_normals(self):
        """Frame::ComputeNormals (frame.cpp:21-36)"""
        import torch
        if self.normals is None:
            self.normals = torch.empty((self.height, self.width, 3), dtype=torch.float32, device=self.device)
        k = self.depth_projection
        check(lib().vk_frame_compute_normals(_ptr(self.depth), _ref(k), _ptr(self.normals),
                                             self.width, self.height, stream()), "vk_frame_compute_normals")
        self.touch()
        return self.normals

    def filter_depths(self):
        """Frame::FilterDepths (frame.cpp:8-19)"""
        import torch
        tmp = torch.empty_like(self.depth)
        check(lib().vk_frame_filter_depths(self.width, self.height, _ptr(self.depth), _ptr(tmp), stream()),
              "vk_frame_filter_depths")
        self.depth.copy_(tmp)
        self.touch()

    def downsample(self):
        """Frame::Downsample (frame.cpp:38-58): nearest depth/normals, box colour,
        intrinsics halved."""
        import torch
        w2, h2 = self.width // 2, self.height // 2
        depth = torch.empty((h2, w2), dtype=torch.float32, device=self.device)
        check(lib().vk_image_downsample(self.width, self.height, _ptr(self.depth), _ptr(depth), 1, stream()),
              "vk_image_downsample")
        color = normals = None
        if self.color is not None:
            ch, cw = self.color.shape[0], self.color.shape[1]     # ColorImage::Downsample uses its own size
            color = torch.empty((ch // 2, cw // 2, 3), dtype=torch.float32, device=self.device)
            check(lib().vk_color_image_downsample(cw, ch, _ptr(self.color), _ptr(color), 0,
                                                  stream()), "vk_color_image_downsample")
        if self.normals is not None:
            normals = torch.empty((h2, w2, 3), dtype=torch.float32, device=self.device)
            check(lib().vk_color_image_downsample(self.width, self.height, _ptr(self.normals), _ptr(normals), 1,
                                                  stream()), "vk_color_image_downsample")
        f32 = np.float32

        def half(k):  # Vector2f / 2 multiplies by 1/2 (matrix.h:279-295)
            return T.Projection.make(f32(k.fx) * f32(0.5), f32(k.fy) * f32(0.5),
                                     f32(k.cx) * f32(0.5), f32(k.cy) * f32(0.5))
        return Frame(depth, half(self.depth_projection), self.depth_to_world, color, normals,
                     half(self.color_projection), self.depth_to_color, self.device)


class Volume:
    """vulcan::Volume (volume.h:16-117): owns the device buffers, forwards to vk_volume_*."""

    def __init__(self, main_block_count, excess_block_count, voxel_length=0.008,
                 truncation_length=0.04, depth_range=(0.1, 5.0), device="cuda"):
        import torch
        self.device = device
        n = main_block_count + excess_block_count
        self.main, self.excess, self.max = main_block_count, excess_block_count, n
        self.voxels = _dev_bytes(n * 512 * 20, device)
        self.hash_entries = _dev_bytes(n * 16, device)
        self.free_voxel_blocks = torch.empty(n, dtype=torch.int32, device=device)
        self.allocation_types = _dev_bytes(main_block_count, device)
        self.allocation_blocks = _dev_bytes(main_block_count * 8, device)
        self.block_visibility = _dev_bytes(n, device)
        self.visible_blocks = torch.empty(n, dtype=torch.int32, device=device)
        self.counters = torch.zeros(T.VK_CTR_COUNT, dtype=torch.int32, device=device)
        self.voxel_length = voxel_length
        self.truncation_length = truncation_length
        self.depth_range = depth_range
        # raycast bounds computed ahead, inside the integrate launch (vk_view_bounds);
        # a Tracer attaches its scratch and settings here
        self.view_bounds = None
        self._view_records = []      # every Tracer's record: all go stale when the visible list changes
        check(lib().vk_volume_initialize(_ref(self.desc()), stream()), "vk_volume_initialize")

    def attach_view_bounds(self, record):
        """A Tracer registers its vk_view_bounds: the integrators prepare the bounds of
        the most recently attached one; every attached record is invalidated together."""
        self._view_records.append(record)
        self.view_bounds = record

    def _view_changed(self):
        for record in self._view_records:
            record.valid = 0

    def desc(self):
        d = T.Volume()
        d.voxels = self.voxels.data_ptr()
        d.hash_entries = self.hash_entries.data_ptr()
        d.free_voxel_blocks = self.free_voxel_blocks.data_ptr()
        d.allocation_types = self.allocation_types.data_ptr()
        d.allocation_blocks = self.allocation_blocks.data_ptr()
        d.block_visibility = self.block_visibility.data_ptr()
        d.visible_blocks = self.visible_blocks.data_ptr()
        d.counters = self.counters.data_ptr()
        d.main_block_count, d.excess_block_count = self.main, self.excess
        d.voxel_length, d.truncation_length = self.voxel_length, self.truncation_length
        d.min_depth, d.max_depth = self.depth_range
        return d

    # -- Volume::SetView and its four protected stages (volume.cu:430-535)
    light_prep = None      # T.LightPrep of an attached LightIntegrator (attach_light_preparation)
    requests_ahead = None  # T.RequestsAhead: the frame Tracer.trace(.., next_frame=) announced (vk_requests_ahead)

    def attach_light_preparation(self, prep):
        """A LightIntegrator's mask / record buffers: set_view fills them in its own request pass
        (vk_volume_set_view_prepare) for the integrate that follows."""
        self.light_prep = prep

    def set_view(self, frame, rounds=1, compute_normals=False):
        """Volume::SetView. `rounds` > 1: the state of that many consecutive SetView calls with this
        frame (the reference's frame loop makes three, apps/vulcan/vulcan.cu:316-318), in one call.
        `compute_normals`: frame.compute_normals() is still due and is done by this call — inside its
        request pass when a LightIntegrator's preparation rides along (vk_light_prep.normals_out)."""
        import torch
        self._view_changed()
        announced = self.requests_ahead is not None and self.requests_ahead.valid == 1
        stale_prep = False
        if announced and compute_normals:
            # Tracer.trace(.., next_frame=frame, next_needs_normals=True) has computed this frame's normals with the pass
            # (or, without a riding preparation, as a launch of their own): nothing is due any more. Touching the frame
            # now would give it a content id the record does not name, and the library would refuse the very frame that
            # was announced. Any OTHER frame is refused below, by the library, with the record kept.
            r = self.requests_ahead
            if not (r.depth == (frame.depth.data_ptr() if frame.depth is not None else None) and r.content_id == frame.desc().content_id):
                raise VkError("set_view(compute_normals=True): another frame was announced by Tracer.trace(next_frame=...); "
                              "fuse that frame first, or Volume.cancel_requests_ahead()")
            if not r.normals_made:
                # announced WITHOUT its normals (trace(next_frame=frame), next_needs_normals left False): they are still due
                # (skipped until round 5 on the record's validity alone). Written in place and without a new stamp
                # — the record names the frame by its content id — and whatever the pass prepared from the old normals is
                # void after this SetView: LightIntegrator.integrate prepares again from the new ones.
                if frame.normals is None:
                    raise VkError("set_view(compute_normals=True): the announced frame has no normal image to write to")
                check(lib().vk_frame_compute_normals(_ptr(frame.depth), _ref(frame.depth_projection), _ptr(frame.normals),
                                                     frame.width, frame.height, stream()), "vk_frame_compute_normals")
                stale_prep = self.light_prep is not None
            compute_normals = False
        if compute_normals:
            if self.light_prep is None:
                frame.compute_normals()
            else:
                if frame.normals is None:
                    frame.normals = torch.empty((frame.height, frame.width, 3), dtype=torch.float32, device=frame.device)
                frame.touch()                                  # the normal image's content is new
                self.light_prep.normals_out = frame.normals.data_ptr()
        prep = _ref(self.light_prep) if self.light_prep is not None else None
        if announced:
            # Tracer.trace(frame, next_frame=...) announced a frame: its request pass is made, this must be its SetView
            # (vk_volume_set_view_rounds_ahead refuses any other frame)
            check(lib().vk_volume_set_view_rounds_ahead(_ref(self.desc()), _ref(frame.desc()), prep, int(rounds),
                                                        _ref(self.requests_ahead), stream()), "vk_volume_set_view_rounds_ahead")
        elif compute_normals and prep is not None:
            check(lib().vk_volume_set_view_rounds(_ref(self.desc()), _ref(frame.desc()), prep, int(rounds), stream()),
                  "vk_volume_set_view_rounds")
        elif rounds != 1:
            check(lib().vk_volume_set_view_rounds(_ref(self.desc()), _ref(frame.desc()), prep, int(rounds), stream()),
                  "vk_volume_set_view_rounds")
        elif prep is not None:
            check(lib().vk_volume_set_view_prepare(_ref(self.desc()), _ref(frame.desc()), prep, stream()),
                  "vk_volume_set_view_prepare")
        else:
            check(lib().vk_volume_set_view(_ref(self.desc()), _ref(frame.desc()), stream()), "vk_volume_set_view")
        if stale_prep:
            self.light_prep.valid = 0

    def set_view_at_device_pose(self, frame, pose_dev, rounds=1):
        """Volume::SetViewAtDevicePose (not upstream, round 6): SetView(frame, rounds) at the pose in `pose_dev` (a device
        vk_transform: what a tracker's launches in front of this call leave there), enqueued before the host has that pose.
        frame.depth_to_world is ignored. False: not possible in this state (an announced frame, the three-launch test form,
        normals still due) — nothing was launched, call set_view once the pose is known."""
        if self.requests_ahead is not None and self.requests_ahead.valid == 1:
            return False
        prep = _ref(self.light_prep) if self.light_prep is not None else None
        if self.light_prep is not None and self.light_prep.normals_out:
            return False
        rc = lib().vk_volume_set_view_at_device_pose(_ref(self.desc()), _ref(frame.desc()), _ptr(pose_dev), prep, int(rounds), stream())
        if rc == T.VK_ERR_UNSUPPORTED:
            return False
        check(rc, "vk_volume_set_view_at_device_pose")
        self._view_changed()
        return True

    def cancel_requests_ahead(self, rounds=1):
        """vk_requests_ahead_cancel: the way out of an announced frame that will not be fused as announced — its SetView is
        completed from the record (handle + visibility pass), after which any frame may follow."""
        if self.requests_ahead is None:
            return
        self._view_changed()
        check(lib().vk_requests_ahead_cancel(_ref(self.desc()), _ref(self.requests_ahead), int(rounds), stream()),
              "vk_requests_ahead_cancel")

    def _no_requests_pending(self, stage):
        # the staged SetView stages on top of an announced frame's requests would mix two frames' state (vk.h)
        if self.requests_ahead is not None and self.requests_ahead.valid == 1:
            raise VkError(f"{stage}: a frame announced by Tracer.trace(next_frame=...) has its requests in the volume; "
                          "set_view(that frame) or cancel_requests_ahead() first")

    def reset_block_visibility(self):
        self._no_requests_pending("reset_block_visibility")
        check(lib().vk_volume_reset_block_visibility(_ref(self.desc()), stream()), "reset_block_visibility")

    def create_allocation_requests(self, frame):
        self._no_requests_pending("create_allocation_requests")
        check(lib().vk_volume_create_allocation_requests(
            _ref(self.desc()), _ptr(frame.depth), frame.width, frame.height,
            _ref(frame.depth_projection), _ref(frame.depth_to_world), stream()), "create_allocation_requests")

    def handle_allocation_requests(self):
        self._no_requests_pending("handle_allocation_requests")
        check(lib().vk_volume_handle_allocation_requests(_ref(self.desc()), stream()), "handle_allocation_requests")

    def update_block_visibility(self, frame):
        self._view_changed()
        tdw = frame.depth_to_world.inverse()
        check(lib().vk_volume_update_block_visibility(
            _ref(self.desc()), frame.width, frame.height, _ref(frame.depth_projection), _ref(tdw), stream()),
            "update_block_visibility")

    # -- host views (blocking)
    def read_counters(self):
        out = (C.c_int32 * T.VK_CTR_PUBLIC)()
        check(lib().vk_volume_read_counters_sync(_ref(self.desc()), out, stream()), "read_counters")
        return np.array(out[:], dtype=np.int32)

    @property
    def visible_count(self):
        return int(self.read_counters()[T.VK_CTR_VISIBLE])

    def visible(self):
        return self.visible_blocks[:self.visible_count].cpu().numpy()

    def host_voxels(self):
        return to_numpy(self.voxels, T.voxel_dtype)

    def host_entries(self):
        return to_numpy(self.hash_entries, T.hash_entry_dtype)

    def host_visibility(self):
        return self.block_visibility.cpu().numpy()

    def host_allocation_types(self):
        return self.allocation_types.cpu().numpy()

    def host_allocation_blocks(self):
        return to_numpy(self.allocation_blocks, T.block_dtype)

    def upload(self, host):
        """Copy an oracle HostVolume (same sizes) into this volume's buffers — lets a
        test start the device path from an exactly known state."""
        import torch
        assert host.main == self.main and host.excess == self.excess
        self._view_changed()
        for name in ("voxels", "hash_entries", "allocation_types", "allocation_blocks", "block_visibility"):
            src = torch.from_numpy(np.frombuffer(getattr(host, name).tobytes(), dtype=np.uint8).copy())
            getattr(self, name).copy_(src.to(self.device))
        self.free_voxel_blocks.copy_(torch.from_numpy(host.free_voxel_blocks).to(self.device))
        self.visible_blocks.copy_(torch.from_numpy(host.visible_blocks).to(self.device))
        self.counters.copy_(torch.from_numpy(host.counters).to(self.device))
        self.voxel_length, self.truncation_length = host.voxel_length, host.truncation_length
        self.depth_range = host.depth_range


class Integrator:
    """vulcan::Integrator (integrator.h:12-45): depth range (0.1,5), max weights 16."""

    def __init__(self, volume):
        self.volume = volume
        self.params = T.Integrator.default()

    def _call(self, fn, frame, *extra):
        check(getattr(lib(), fn)(_ref(self.volume.desc()), _ref(self.params), *extra,
                                 _ref(frame.desc()), stream()), fn)

    def _fused(self, frame, mode, light=None, mask=None, records=None):
        """depth (+ colour) in one pass; when a Tracer is attached to the volume the
        same launch also prepares the raycast bounds of this frame's view."""
        vb = self.volume.view_bounds
        check(lib().vk_integrate_ahead(_ref(self.volume.desc()), _ref(self.params), _ref(frame.desc()), mode,
                                       _ref(light) if light is not None else None, _ptr(mask), _ptr(records),
                                       _ref(vb) if vb is not None else None, stream()), "vk_integrate_ahead")


class DepthIntegrator(Integrator):
    def integrate(self, frame):                      # depth_integrator.cu:89-115
        self._fused(frame, 0)


class ColorIntegrator(Integrator):
    def integrate(self, frame):                      # color_integrator.cu:144-148, one pass
        self._fused(frame, 1)

    def integrate_depth(self, frame):                # color_integrator.cu:150-176
        self._call("vk_integrate_depth", frame)

    def integrate_color(self, frame):                # color_integrator.cu:178-204
        self._call("vk_integrate_color", frame)


class LightIntegrator(Integrator):
    def __init__(self, volume):
        super().__init__(volume)
        self.light = T.Light.make(1.0, (0, 0, 0))    # light.h:14-18
        self.depth_threshold = 0.2                   # light_integrator.cu:256
        self.frame_mask = None
        self.pixel_records = None

    def compute_frame_mask(self, frame):             # light_integrator.cu:277-293
        import torch
        if self.frame_mask is None or tuple(self.frame_mask.shape) != (frame.height, frame.width):
            self.frame_mask = torch.empty((frame.height, frame.width), dtype=torch.float32, device=frame.device)
        check(lib().vk_light_compute_frame_mask(_ref(frame.desc()), self.depth_threshold,
                                                _ptr(self.frame_mask), stream()), "vk_light_compute_frame_mask")
        return self.frame_mask

    def prepare(self, frame):
        """ComputeFrameMask + the per-pixel half of IntegrateColor (vk_light_prepare): the mask
        and one {Tcd * normal, mask} record per pixel."""
        import torch
        shape = (frame.height, frame.width)
        if self.frame_mask is None or tuple(self.frame_mask.shape) != shape:
            self.frame_mask = torch.empty(shape, dtype=torch.float32, device=frame.device)
        if self.pixel_records is None or tuple(self.pixel_records.shape[:2]) != shape:
            self.pixel_records = torch.empty(shape + (4,), dtype=torch.float32, device=frame.device)
        prep = getattr(self, "_prep", None)
        if prep is None or self.volume.light_prep is not prep or prep.mask != self.frame_mask.data_ptr() \
                or prep.records != self.pixel_records.data_ptr():
            # (re)attach: the volume's next set_view prepares these buffers in its request pass
            prep = T.LightPrep()
            prep.mask, prep.records = self.frame_mask.data_ptr(), self.pixel_records.data_ptr()
            prep.capacity = frame.width * frame.height
            self._prep = prep
            self.volume.attach_light_preparation(prep)
        prep.depth_threshold = self.depth_threshold
        if lib().vk_light_prepared(_ref(prep), _ref(frame.desc()), self.depth_threshold):
            prep.valid = 0                           # used once: a second integrate of the frame prepares again
            return
        check(lib().vk_light_prepare(_ref(frame.desc()), self.depth_threshold, _ptr(self.frame_mask),
                                     _ptr(self.pixel_records), stream()), "vk_light_prepare")

    def integrate(self, frame):                      # light_integrator.cu:270-275
        self.prepare(frame)
        self._fused(frame, 2, self.light, self.frame_mask, self.pixel_records)

    def integrate_depth(self, frame):
        self._call("vk_integrate_depth", frame)

    def integrate_color(self, frame):                # light_integrator.cu:323-354
        self._call("vk_integrate_light_color", frame, _ref(self.light), _ptr(self.frame_mask))


class Tracer:
    """vulcan::Tracer (tracer.h:24-77): 80x60 bounds grid, 262144 patch capacity."""

    BOUNDS_W, BOUNDS_H, PATCH_CAPACITY = 80, 60, 262144   # tracer.cpp:56-57,134

    def __init__(self, volume):
        import torch
        self.volume = volume
        self.depth_range = (0.1, 5.0)
        dev = volume.device
        n = int(lib().vk_trace_bounds_floats(self.BOUNDS_W, self.BOUNDS_H))
        self.bounds_scratch = torch.empty(n, dtype=torch.float32, device=dev)
        self.bounds = self.bounds_scratch[:self.BOUNDS_W * self.BOUNDS_H * 2].view(self.BOUNDS_H, self.BOUNDS_W, 2)
        self.patches = _dev_bytes(self.PATCH_CAPACITY * 16, dev)
        self.patch_count = torch.zeros(1, dtype=torch.int32, device=dev)
        # let the volume's integrators prepare this tracer's bounds ahead of time
        vb = T.ViewBounds()
        vb.scratch = self.bounds_scratch.data_ptr()
        vb.bounds_width, vb.bounds_height = self.BOUNDS_W, self.BOUNDS_H
        vb.min_depth, vb.max_depth = self.depth_range
        # the pinned word a normals workgroup sets when its bounded wait expires (vk.h vk_view_bounds.late_host)
        self._late = C.c_void_p()
        check(lib().vk_malloc_host(C.byref(self._late), 4), "vk_malloc_host")
        C.memset(self._late, 0, 4)
        vb.late_host = self._late.value
        self.view_bounds = vb
        volume.attach_view_bounds(vb)

    def __del__(self):
        late = getattr(self, "_late", None)
        if late is not None and late.value and _LIB is not None:
            self.view_bounds.late_host = None
            _LIB.vk_free_host(late)
            self._late = None

    def settle_normals(self):
        """vk_trace_normals_settle: synchronises, and raises VkError (VK_ERR_TIMEOUT) when a normals workgroup of the last
        trace(next_frame=...) gave up its wait — the normals have then been recomputed by a launch of their own."""
        check(lib().vk_trace_normals_settle(_ref(self.view_bounds), stream()), "vk_trace_normals_settle")

    def trace(self, frame, next_frame=None, next_needs_normals=False, normals=True):
        """Tracer::Trace (tracer.cpp:41-47): writes frame.depth / color / normals. `normals=False` (not upstream): the last
        stage is left out — for the tracking loop, whose next PyramidTracker.track(.., keyframe_normals_due=True) computes the
        key frame's normal image in its pyramid launch (one launch less per frame). `next_frame` (not upstream): the frame the
        volume's next set_view will be called with — its request pass (and, with `next_needs_normals` and a LightIntegrator's
        preparation attached, its normal image) is made behind the raycast's workgroups, in the same launch
        (vk_trace_ahead_requests)."""
        import torch
        if frame.color is None:
            frame.color = torch.empty((frame.height, frame.width, 3), dtype=torch.float32, device=frame.device)
        if frame.normals is None:
            frame.normals = torch.empty((frame.height, frame.width, 3), dtype=torch.float32, device=frame.device)
        vb = self.view_bounds
        if (vb.min_depth, vb.max_depth) != tuple(np.float32(d) for d in self.depth_range):
            vb.min_depth, vb.max_depth = self.depth_range
            vb.valid = 0
        if next_frame is None:
            check(lib().vk_trace_ahead(_ref(self.volume.desc()), _ref(frame.desc()), _ref(vb), _ptr(frame.depth),
                                       _ptr(frame.color), _ptr(frame.normals) if normals else None, stream()), "vk_trace_ahead")
            frame.touch()
            return
        v = self.volume
        if v.requests_ahead is None:
            v.requests_ahead = T.RequestsAhead()
        prep = v.light_prep
        if next_needs_normals:
            if prep is None:
                next_frame.compute_normals()
            else:
                if next_frame.normals is None:
                    next_frame.normals = torch.empty((next_frame.height, next_frame.width, 3), dtype=torch.float32, device=next_frame.device)
                next_frame.touch()
                prep.normals_out = next_frame.normals.data_ptr()
        check(lib().vk_trace_ahead_requests(_ref(v.desc()), _ref(frame.desc()), _ref(vb), _ptr(frame.depth), _ptr(frame.color),
                                            _ptr(frame.normals), _ref(next_frame.desc()), _ref(prep) if prep is not None else None,
                                            _ref(v.requests_ahead), stream()), "vk_trace_ahead_requests")
        if next_needs_normals and prep is None and v.requests_ahead.valid == 1:
            v.requests_ahead.normals_made = 1              # (the library knows of the riding normals only)
        if prep is not None and prep.normals_out and v.requests_ahead.valid != 1:
            prep.normals_out = None                        # the pass could not be made ahead: the normals as a launch of their own
            next_frame.compute_normals()
        frame.touch()

    # the tracer.cuh free functions, for the stage-by-stage tests
    def compute_patches(self, frame, block_count=None):
        v = self.volume
        self.patch_count.zero_()
        tcw = frame.depth_to_world.inverse()
        n = v.max if block_count is None else block_count
        dev_count = C.c_void_p(v.counters.data_ptr()) if block_count is None else None
        check(lib().vk_trace_compute_patches(
            _ptr(v.visible_blocks), _ptr(v.hash_entries), _ref(tcw), _ref(frame.depth_projection),
            np.float32(8) * np.float32(v.voxel_length), self.depth_range[0], self.depth_range[1], n, dev_count,
            frame.width, frame.height, self.BOUNDS_W, self.BOUNDS_H, _ptr(self.patches), self.PATCH_CAPACITY,
            _ptr(self.patch_count), stream()), "vk_trace_compute_patches")

    def compute_bounds(self):
        check(lib().vk_trace_reset_bounds(_ptr(self.bounds), self.BOUNDS_W * self.BOUNDS_H, stream()),
              "vk_trace_reset_bounds")
        check(lib().vk_trace_compute_bounds(_ptr(self.patches), _ptr(self.bounds), self.BOUNDS_W,
                                            self.PATCH_CAPACITY, _ptr(self.patch_count), stream()),
              "vk_trace_compute_bounds")

    def compute_block_bounds(self, frame):
        v = self.volume
        tcw = frame.depth_to_world.inverse()
        check(lib().vk_trace_compute_block_bounds(
            _ptr(v.visible_blocks), _ptr(v.hash_entries), _ref(tcw), _ref(frame.depth_projection),
            np.float32(8) * np.float32(v.voxel_length), self.depth_range[0], self.depth_range[1], v.max,
            C.c_void_p(v.counters.data_ptr()), frame.width, frame.height, self.BOUNDS_W, self.BOUNDS_H,
            _ptr(self.bounds), stream()), "vk_trace_compute_block_bounds")

    def compute_points(self, frame, depth_out, color_out):
        v = self.volume
        check(lib().vk_trace_compute_points(
            _ptr(v.hash_entries), _ptr(v.voxels), _ptr(self.bounds), v.main,
            np.float32(8) * np.float32(v.voxel_length), v.voxel_length, v.truncation_length,
            _ref(frame.depth_to_world), _ref(frame.depth_projection), _ptr(depth_out), _ptr(color_out),
            frame.width, frame.height, self.BOUNDS_W, self.BOUNDS_H, stream()), "vk_trace_compute_points")

    def host_patches(self):
        n = min(int(self.patch_count.cpu()[0]), self.PATCH_CAPACITY)
        return to_numpy(self.patches[:n * 16], T.patch_dtype)


class _PollMixin:
    """Early exit of the device-side Gauss-Newton loop (vk_track_poll): a pinned
    {iterations, converged} mirror the enqueuing call looks at every `poll_chunk` steps."""

    poll_chunk = int(os.environ.get("VK_TRACK_POLL_CHUNK", "4"))   # 0: enqueue every step, never block

    def _poll(self):
        if getattr(self, "_poll_desc", None) is None:
            host, pose = C.c_void_p(), C.c_void_p()
            check(lib().vk_malloc_host(C.byref(host), 16), "vk_malloc_host")
            check(lib().vk_malloc_host(C.byref(pose), C.sizeof(T.Transform)), "vk_malloc_host")
            C.memset(host, 0, 16)
            self._poll_host, self._pose_host = host, pose
            self._poll_desc = T.TrackPoll(host.value, 0, pose.value)
        self._poll_desc.chunk = int(self.poll_chunk)          # 0: enqueue every step, never look
        return _ref(self._poll_desc)

    def _wait_pose(self):
        """Tracker::EndSolve (tracker.cpp:78-82): the pose of the Track just issued, picked up from
        pinned memory (vk_track_wait) instead of a copy + stream synchronisation."""
        rc = lib().vk_track_wait(_ref(self._poll_desc), stream())
        if rc != 0 and int(self.state.cpu()[1]) == T.VK_TRACK_ABORTED:
            raise TrackAborted("the one-launch loop ended with VK_TRACK_ABORTED")
        check(rc, "vk_track_wait")
        return T.Transform.from_buffer_copy(C.string_at(self._pose_host, C.sizeof(T.Transform)))

    _staged = False                          # falling back from an aborted one-launch loop

    def _with_fallback(self, frame, run):
        """A one-launch loop that could not get its workgroups onto the device together ends with
        VK_TRACK_ABORTED and no pose: run the Track again from the start pose with a reduce hook
        that changes nothing, i.e. one launch per stage (vk_reduce_nothing, vk.h)."""
        start = frame.depth_to_world
        try:
            return run(frame)
        except TrackAborted:
            frame.depth_to_world = start
            self._staged = True
            try:
                return run(frame)
            finally:
                self._staged = False

    comm = None                              # vulcan_amd.comm.Communicator: the rig's all-reduce, from C

    def _c_hook(self):
        """(vk_icp_reduce_fn, user): the communicator's C hook, or the python reduce hook
        wrapped in a callback, or (None, None)."""
        fn = self._c_hook_fn()
        return fn, (self.comm.handle if self.comm is not None else None)

    def _c_hook_fn(self):
        if self.comm is not None:
            return self.comm.hook_fn
        if self.reduce_hook is None:
            if self._staged:
                return C.cast(lib().vk_reduce_nothing, C.c_void_p)
            return None
        system, py_hook = self.system, self.reduce_hook

        def _call(ptr, count, user, strm):
            py_hook(system)
            return 0
        self._hook_keepalive = _REDUCE_FN(_call)          # must outlive the C call
        return C.cast(self._hook_keepalive, C.c_void_p)

    def __del__(self):
        # a loop kernel that is still running writes both pinned blocks: drain the stream first
        host, pose = getattr(self, "_poll_host", None), getattr(self, "_pose_host", None)
        if (host is not None or pose is not None) and _LIB is not None:
            try:
                _LIB.vk_stream_synchronize(stream())
            except Exception:      # noqa: BLE001  (interpreter shutdown: torch may be gone; the process ends anyway)
                return
            for block in (host, pose):
                if block is not None:
                    _LIB.vk_free_host(block)
            self._poll_host = self._pose_host = None


class DepthTracker(_PollMixin):
    """vulcan::DepthTracker (depth_tracker.h, tracker.h): Gauss-Newton ICP against a
    keyframe; the pose, the 27-float system and the solve stay on the device."""

    def __init__(self, device="cuda"):
        import torch
        self.device = device
        self.translation_enabled = True      # tracker.cpp:11
        self.max_iterations = 20             # tracker.cpp:12
        self.keyframe = None
        # one 48-float buffer = 36 hessian + 6 gradient + pad, so a multi-GPU rig
        # needs ONE all-reduce per Gauss-Newton iteration
        self.system = torch.zeros(48, dtype=torch.float32, device=device)
        self.hessian = self.system[:36]
        self.gradient = self.system[36:42]
        self.pose = _dev_bytes(128, device)
        self.state = torch.zeros(2, dtype=torch.int32, device=device)
        self.update = torch.zeros(6, dtype=torch.float32, device=device)
        self.workspace = None
        self.reduce_hook = None              # e.g. an all-reduce over ranks

    @staticmethod
    def _view(frame):
        v = T.IcpView()
        v.depths, v.normals = frame.depth.data_ptr(), frame.normals.data_ptr()
        v.width, v.height = frame.width, frame.height
        v.projection = frame.depth_projection
        return v

    def _workspace(self, frame):
        import torch
        n = int(lib().vk_icp_workspace_floats(frame.width, frame.height))
        if self.workspace is None or self.workspace.numel() < n:
            # zeroed: memory that comes back from the allocator may hold anything, and the library — which clears a workspace
            # only when it is new to it, old, grown or overwritten by a staged loop (vk.h) — cannot see that
            self.workspace = torch.zeros(n, dtype=torch.float32, device=self.device)
        return self.workspace

    def compute_residuals(self, frame):      # depth_tracker.cu:272-300
        import torch
        out = torch.empty((frame.height, frame.width), dtype=torch.float32, device=self.device)
        check(lib().vk_icp_compute_residuals(_ref(self._view(self.keyframe)), _ref(self.keyframe.depth_to_world),
                                             _ref(self._view(frame)), _ref(frame.depth_to_world), _ptr(out),
                                             stream()), "vk_icp_compute_residuals")
        return out

    def compute_jacobian(self, frame):       # depth_tracker.cu:302-336
        import torch
        out = torch.empty((frame.height, frame.width, 6), dtype=torch.float32, device=self.device)
        check(lib().vk_icp_compute_jacobian(_ref(self._view(self.keyframe)), _ref(self.keyframe.depth_to_world),
                                            _ref(self._view(frame)), _ref(frame.depth_to_world),
                                            int(self.translation_enabled), _ptr(out), stream()),
              "vk_icp_compute_jacobian")
        return out

    def compute_system(self, frame, pose_on_device=False):   # depth_tracker.cu:338-378
        ws = self._workspace(frame)
        check(lib().vk_icp_compute_system(
            _ref(self._view(self.keyframe)), _ref(self.keyframe.depth_to_world), _ref(self._view(frame)),
            _ref(frame.depth_to_world), _ptr(self.pose) if pose_on_device else None,
            int(self.translation_enabled), _ptr(ws), _ptr(self.hessian), _ptr(self.gradient), stream()),
            "vk_icp_compute_system")

    def track(self, frame):
        return self._with_fallback(frame, self._track)

    def _track(self, frame):
        """Tracker::Track (tracker.cpp:53-63): <= max_iterations Gauss-Newton steps,
        all enqueued without a host sync; one 128-byte readback of the pose at the end."""
        import torch
        check(lib().vk_transform_upload(_ptr(self.pose), _ref(frame.depth_to_world), stream()), "vk_transform_upload")
        self.state.zero_()
        # one C call: the whole loop is one launch (with a reduce hook: 3 launches per step,
        # enqueuing stops once the loop has converged)
        check(lib().vk_icp_track(_ref(self._view(self.keyframe)), _ref(self.keyframe.depth_to_world),
                                 _ref(self._view(frame)), _ptr(self.pose), self.max_iterations,
                                 int(self.translation_enabled), _ptr(self._workspace(frame)), _ptr(self.system),
                                 _ptr(self.state), _ptr(self.update), *self._c_hook(), self._poll(), stream()),
              "vk_icp_track")
        out = self._wait_pose()
        frame.depth_to_world = out
        return out


def track_rig(tracker, frame, exchange):
    """vk_icp_track_rig: DepthTracker::Track of one camera of a rigid rig, the ranks' normal systems
    added inside the one-launch loop through `exchange` (a T.RigExchange, vulcan_amd.comm)."""
    check(lib().vk_transform_upload(_ptr(tracker.pose), _ref(frame.depth_to_world), stream()), "vk_transform_upload")
    tracker.state.zero_()
    check(lib().vk_icp_track_rig(_ref(tracker._view(tracker.keyframe)), _ref(tracker.keyframe.depth_to_world),
                                 _ref(tracker._view(frame)), _ptr(tracker.pose), tracker.max_iterations,
                                 int(tracker.translation_enabled), _ptr(tracker._workspace(frame)), _ptr(tracker.system),
                                 _ptr(tracker.state), _ptr(tracker.update), _ref(exchange), tracker._poll(), stream()),
          "vk_icp_track_rig")
    out = tracker._wait_pose()
    frame.depth_to_world = out
    return out


class ColorTracker(_PollMixin):
    """vulcan::ColorTracker (color_tracker.h): photometric Gauss-Newton tracking of a
    frame against a keyframe — intensity residuals sampled bilinearly in the frame,
    one per keyframe pixel. Pose, system and solve stay on the device."""

    def __init__(self, device="cuda"):
        import torch
        self.device = device
        self.translation_enabled = True      # tracker.cpp:11
        self.max_iterations = 20             # tracker.cpp:12
        self._keyframe = None
        self._key_side = None
        self.system = torch.zeros(48, dtype=torch.float32, device=device)
        self.hessian = self.system[:36]
        self.gradient = self.system[36:42]
        self.pose = _dev_bytes(C.sizeof(T.ColorPose), device)
        self.state = torch.zeros(2, dtype=torch.int32, device=device)
        self.update = torch.zeros(6, dtype=torch.float32, device=device)
        self.workspace = None
        self.reduce_hook = None

    @property
    def keyframe(self):
        return self._keyframe

    @keyframe.setter
    def keyframe(self, frame):
        self._keyframe = frame
        self._key_side = None

    # -- ColorTracker::ComputeKeyframeIntensities / FrameIntensities / FrameGradients ------
    def _side(self, frame, with_gradients):
        import torch
        h, w = frame.height, frame.width
        inten = torch.empty((h, w), dtype=torch.float32, device=self.device)
        check(lib().vk_color_image_convert(h * w, _ptr(frame.color), _ptr(inten), stream()), "vk_color_image_convert")
        gx = gy = None
        if with_gradients:
            gx, gy = torch.empty_like(inten), torch.empty_like(inten)
            check(lib().vk_image_gradients(w, h, _ptr(inten), _ptr(gx), _ptr(gy), stream()), "vk_image_gradients")
        v = T.ColorView()
        v.depths, v.normals, v.intensities = frame.depth.data_ptr(), frame.normals.data_ptr(), inten.data_ptr()
        v.gradient_x = gx.data_ptr() if gx is not None else None
        v.gradient_y = gy.data_ptr() if gy is not None else None
        v.width, v.height = w, h
        v.projection = frame.color_projection
        return v, (inten, gx, gy)          # the tensors must outlive the view

    def _key(self):
        if self._key_side is None:
            self._key_side = self._side(self._keyframe, False)
        return self._key_side

    def tcm(self, frame):
        """color_tracker.cu:312-320"""
        key_Tcw = self._keyframe.depth_to_color * self._keyframe.depth_to_world.inverse()
        frame_Tcw = frame.depth_to_color * frame.depth_to_world.inverse()
        return frame_Tcw * key_Tcw.inverse()

    def _workspace(self):
        import torch
        n = int(lib().vk_icp_workspace_floats(self._keyframe.width, self._keyframe.height))
        if self.workspace is None or self.workspace.numel() < n:
            # zeroed: memory that comes back from the allocator may hold anything, and the library — which clears a workspace
            # only when it is new to it, old, grown or overwritten by a staged loop (vk.h) — cannot see that
            self.workspace = torch.zeros(n, dtype=torch.float32, device=self.device)
        return self.workspace

    def compute_residuals(self, frame):      # color_tracker.cu:296-344
        import torch
        kv, keep_k = self._key()
        fv, keep_f = self._side(frame, False)
        out = torch.empty((self._keyframe.height, self._keyframe.width), dtype=torch.float32, device=self.device)
        check(lib().vk_color_tracker_compute_residuals(_ref(kv), _ref(fv), _ref(self.tcm(frame)), _ptr(out), stream()),
              "vk_color_tracker_compute_residuals")
        return out

    def compute_jacobian(self, frame):       # color_tracker.cu:346-410
        import torch
        kv, keep_k = self._key()
        fv, keep_f = self._side(frame, True)
        out = torch.empty((self._keyframe.height, self._keyframe.width, 6), dtype=torch.float32, device=self.device)
        check(lib().vk_color_tracker_compute_jacobian(_ref(kv), _ref(fv), _ref(self.tcm(frame)),
                                                      int(self.translation_enabled), _ptr(out), stream()),
              "vk_color_tracker_compute_jacobian")
        return out

    def compute_system(self, frame):         # color_tracker.cu:412-470
        kv, keep_k = self._key()
        fv, keep_f = self._side(frame, True)
        check(lib().vk_color_tracker_compute_system(_ref(kv), _ref(fv), _ref(self.tcm(frame)), None,
                                                    int(self.translation_enabled), _ptr(self._workspace()),
                                                    _ptr(self.hessian), _ptr(self.gradient), stream()),
              "vk_color_tracker_compute_system")

    def track(self, frame):
        return self._with_fallback(frame, self._track)

    def _track(self, frame):
        """Tracker::Track (tracker.cpp:53-63) with ColorTracker::BeginSolve
        (color_tracker.cpp:19-25): intensities and gradients once, then
        max_iterations steps enqueued without a host sync."""
        import torch
        kv, keep_k = self._key()
        fv, keep_f = self._side(frame, True)
        pose = T.ColorPose()
        pose.depth_to_world = frame.depth_to_world
        host = np.frombuffer(bytes(pose), dtype=np.uint8).copy()
        self.pose.copy_(torch.from_numpy(host).to(self.device))
        self.state.zero_()
        key_Twc = (self._keyframe.depth_to_color * self._keyframe.depth_to_world.inverse()).inverse()
        hook, hook_user = self._c_hook()
        check(lib().vk_color_tracker_track(_ref(kv), _ref(fv), _ref(frame.depth_to_color), _ref(key_Twc),
                                           _ptr(self.pose), self.max_iterations, int(self.translation_enabled),
                                           _ptr(self._workspace()), _ptr(self.system), _ptr(self.state),
                                           _ptr(self.update), hook, hook_user, self._poll(), stream()),
              "vk_color_tracker_track")
        out = self._wait_pose()
        frame.depth_to_world = out
        return out


class LightTracker(ColorTracker):
    """vulcan::LightTracker (light_tracker.h): the colour tracker with a shading model —
    residual Ic - albedo * light.GetShading(Xcp, n) where the frame mask is set,
    point-to-plane elsewhere. The keyframe's colour image is read as albedo."""

    def __init__(self, device="cuda"):
        super().__init__(device)
        self.light = T.Light.make(1.0, (0, 0, 0))    # light.h:14-18
        self.depth_threshold = 0.2                   # light_tracker.cpp:13-14
        self.frame_mask = None

    def compute_frame_mask(self, frame):             # light_tracker.cu:548-564
        import torch
        self.frame_mask = torch.empty((frame.height, frame.width), dtype=torch.float32, device=self.device)
        check(lib().vk_light_compute_frame_mask(_ref(frame.desc()), self.depth_threshold, _ptr(self.frame_mask),
                                                stream()), "vk_light_compute_frame_mask")
        return self.frame_mask

    def _terms(self, frame, mask=None):
        t = T.LightTerms()
        m = mask if mask is not None else self.compute_frame_mask(frame)
        t.frame_mask = m.data_ptr()
        t.light = self.light
        t.frame_Tcd = frame.depth_to_color
        return t, m

    def compute_residuals(self, frame, mask=None):   # light_tracker.cu:566-608 (the mask is computed here)
        import torch
        kv, keep_k = self._key()
        fv, keep_f = self._side(frame, False)
        terms, keep_m = self._terms(frame, mask)
        out = torch.empty((self._keyframe.height, self._keyframe.width), dtype=torch.float32, device=self.device)
        check(lib().vk_light_tracker_compute_residuals(_ref(kv), _ref(fv), _ref(terms), _ref(self.tcm(frame)),
                                                       _ptr(out), stream()), "vk_light_tracker_compute_residuals")
        return out

    def compute_jacobian(self, frame, mask=None):    # light_tracker.cu:610-668
        import torch
        kv, keep_k = self._key()
        fv, keep_f = self._side(frame, True)
        terms, keep_m = self._terms(frame, mask)
        out = torch.empty((self._keyframe.height, self._keyframe.width, 6), dtype=torch.float32, device=self.device)
        check(lib().vk_light_tracker_compute_jacobian(_ref(kv), _ref(fv), _ref(terms), _ref(self.tcm(frame)),
                                                      int(self.translation_enabled), _ptr(out), stream()),
              "vk_light_tracker_compute_jacobian")
        return out

    def compute_system(self, frame, mask=None):      # light_tracker.cu:670-732
        kv, keep_k = self._key()
        fv, keep_f = self._side(frame, True)
        terms, keep_m = self._terms(frame, mask)
        check(lib().vk_light_tracker_compute_system(_ref(kv), _ref(fv), _ref(terms), _ref(self.tcm(frame)), None,
                                                    int(self.translation_enabled), _ptr(self._workspace()),
                                                    _ptr(self.hessian), _ptr(self.gradient), stream()),
              "vk_light_tracker_compute_system")

    def track(self, frame):
        return self._with_fallback(frame, self._track)

    def _track(self, frame):
        """Tracker::Track with LightTracker::BeginSolve (light_tracker.cpp:34-41)."""
        import torch
        kv, keep_k = self._key()
        fv, keep_f = self._side(frame, True)
        terms, keep_m = self._terms(frame)
        pose = T.ColorPose()
        pose.depth_to_world = frame.depth_to_world
        host = np.frombuffer(bytes(pose), dtype=np.uint8).copy()
        self.pose.copy_(torch.from_numpy(host).to(self.device))
        self.state.zero_()
        key_Twc = (self._keyframe.depth_to_color * self._keyframe.depth_to_world.inverse()).inverse()
        check(lib().vk_light_tracker_track(_ref(kv), _ref(fv), _ref(terms), _ref(key_Twc), _ptr(self.pose),
                                           self.max_iterations, int(self.translation_enabled), _ptr(self._workspace()),
                                           _ptr(self.system), _ptr(self.state), _ptr(self.update), *self._c_hook(),
                                           self._poll(), stream()), "vk_light_tracker_track")
        out = self._wait_pose()
        frame.depth_to_world = out
        return out


class PyramidTracker:
    """vulcan::PyramidTracker<DepthTracker> (pyramid_tracker.cpp:52-90): half
    resolution (15 iterations) then full resolution (20)."""

    def __init__(self, tracker=None, device="cuda"):
        self.tracker = tracker or DepthTracker(device)
        self.keyframe = None

    def track(self, frame, compute_normals=False, keyframe_normals_due=False, set_view_of=None, rounds=1):
        """`compute_normals` (not upstream): frame.compute_normals() is still due; with a DepthTracker it is done by the launch
        that builds the pyramid (vk_icp_pyramid_track_frame). `keyframe_normals_due`: the key frame came from
        Tracer.trace(.., normals=False); its normal image is computed by the same launch. `set_view_of` (round 6;
        PyramidTracker<DepthTracker>::ComputeNormalsTrackAndSetView): a Volume whose set_view(frame, rounds) follows this Track
        — with a DepthTracker it is enqueued BEHIND the Track at the pose the loop leaves on the device, before this call waits
        for that pose (Volume.set_view_at_device_pose); otherwise, and after a Track that aborted, it is called afterwards."""
        if set_view_of is not None:
            self._set_view_of, self._set_view_rounds, self._set_view_done = set_view_of, int(rounds), False
            try:
                pose = self.track(frame, compute_normals, keyframe_normals_due)
            finally:
                done, self._set_view_of = self._set_view_done, None
            if not done:
                set_view_of.set_view(frame, rounds=rounds)
            return pose
        t = self.tracker
        if keyframe_normals_due and not isinstance(t, DepthTracker):
            self.keyframe.compute_normals()
            keyframe_normals_due = False
        self._key_normals_due = bool(keyframe_normals_due)
        if isinstance(t, DepthTracker):
            if compute_normals:
                import torch
                if frame.normals is None:
                    frame.normals = torch.empty((frame.height, frame.width, 3), dtype=torch.float32, device=frame.device)
                frame.touch()
            self._normals_due = bool(compute_normals)
            try:
                return self._track_depth(frame)
            finally:
                self._normals_due = False
        if compute_normals:
            frame.compute_normals()
        half_frame = frame.downsample()
        half_key = self.keyframe.downsample()
        t.max_iterations, t.translation_enabled = 15, True
        t.keyframe = half_key
        t.track(half_frame)
        t.max_iterations = 20
        frame.depth_to_world = half_frame.depth_to_world
        t.keyframe = self.keyframe
        return t.track(frame)


    def _track_depth(self, frame):
        return self.tracker._with_fallback(frame, self._track_depth_once)

    def _track_depth_once(self, frame):
        """PyramidTracker<DepthTracker>::Track as ONE call (vk_icp_pyramid_track): both levels
        are enqueued from C, the pose stays on the device in between, one readback at the end."""
        import torch
        t, key = self.tracker, self.keyframe
        n = int(lib().vk_icp_pyramid_floats(key.width, key.height, frame.width, frame.height))
        if getattr(self, "_pyramid", None) is None or self._pyramid.numel() < n:
            self._pyramid = torch.empty(n, dtype=torch.float32, device=t.device)
        t.max_iterations, t.translation_enabled, t.keyframe = 20, True, key
        due = (1 if getattr(self, "_normals_due", False) else 0) | (2 if getattr(self, "_key_normals_due", False) else 0)
        self._normals_due = self._key_normals_due = False       # (a second attempt after an aborted loop finds them computed)
        check(lib().vk_icp_pyramid_track_frame(_ref(t._view(key)), _ref(key.depth_to_world), _ref(t._view(frame)),
                                               _ptr(t.pose), _ref(frame.depth_to_world), due, _ptr(self._pyramid),
                                               _ptr(t._workspace(frame)), _ptr(t.system), _ptr(t.state), _ptr(t.update),
                                               *t._c_hook(), t._poll(), stream()),
              "vk_icp_pyramid_track_frame")
        # (round 6) the caller's SetView at the pose this Track leaves on the device, enqueued before the host waits for it;
        # the first attempt only: after an aborted loop the staged Track follows and the caller's own SetView after it
        volume = getattr(self, "_set_view_of", None)
        early = False
        if volume is not None and not t._staged and t.reduce_hook is None and t.comm is None:
            early = volume.set_view_at_device_pose(frame, t.pose, self._set_view_rounds)
        out = t._wait_pose()
        if volume is not None:
            self._set_view_done = early
        frame.depth_to_world = out
        return out


class Mesh:
    """vulcan::DeviceMesh (mesh.h:16-21): points [n, 3] float32, faces [m, 3] int32 on the device."""

    def __init__(self, points, faces):
        self.points, self.faces = points, faces

    def host(self):
        return self.points.cpu().numpy(), self.faces.cpu().numpy()


class Extractor:
    """vulcan::Extractor (extractor.h:116-134): the whole volume in four launches."""

    def __init__(self, volume):
        import torch
        self.volume = volume
        self.all_allocated = False       # upstream walks the visible blocks (extractor.cu:455-457)
        self.interpolate = True
        n = int(lib().vk_extract_workspace_bytes(volume.main, volume.excess))
        self.workspace = torch.empty(n, dtype=torch.uint8, device=volume.device)
        self.counts = torch.zeros(4, dtype=torch.int32, device=volume.device)

    def extract(self, point_capacity=None, face_capacity=None):
        """Extractor::Extract(DeviceMesh&). Capacities default to upstream's ResizeMesh bound
        for the visible blocks (extractor.cu:700-716: 3 points per voxel, 5 faces per cube),
        capped; a second call with the reported totals follows when they did not suffice."""
        import torch
        v = self.volume
        blocks = v.max if self.all_allocated else max(v.visible_count, 1)
        pc = point_capacity if point_capacity is not None else min(blocks * 1536, 1 << 24)
        fc = face_capacity if face_capacity is not None else min(blocks * 5 * 512, 1 << 25)
        for _ in range(2):
            points = torch.empty((pc, 3), dtype=torch.float32, device=v.device)
            faces = torch.empty((fc, 3), dtype=torch.int32, device=v.device)
            check(lib().vk_extract_mesh(_ref(v.desc()), int(self.all_allocated), int(self.interpolate), _ptr(points), pc,
                                        _ptr(faces), fc, _ptr(self.counts), _ptr(self.workspace), stream()), "vk_extract_mesh")
            np_, nf, self.skipped, self.blocks = (int(c) for c in self.counts.cpu())
            if np_ <= pc and nf <= fc:
                return Mesh(points[:np_], faces[:nf])
            pc, fc = max(np_, 1), max(nf, 1)
        raise VkError("vk_extract_mesh: capacities did not settle")


class Detector:
    """vulcan::Detector (detector.h:10-72): radius / interval filter, 1.5-sigma
    outlier removal, centroid. Everything stays on the device until `detect`
    reads the 48-byte state back."""

    def __init__(self, device="cuda"):
        import torch
        self.device = device
        self.params = T.Detector.default()
        self.state = torch.zeros(C.sizeof(T.DetectState) // 4, dtype=torch.int32, device=device)
        self.inliers = None
        self._workspace = None

    # -- reference accessors -------------------------------------------------------------
    @property
    def radius(self):
        return self.params.radius

    @radius.setter
    def radius(self, value):
        self.params.radius = float(value)

    @property
    def origin(self):
        return tuple(self.params.origin)

    @origin.setter
    def origin(self, value):
        self.params.origin[:] = [float(v) for v in value]

    def get_bounds(self, axis):
        return tuple(self.params.bounds[axis])

    def set_bounds(self, axis, bounds):
        self.params.bounds[axis][0], self.params.bounds[axis][1] = float(bounds[0]), float(bounds[1])

    @property
    def min_inlier_count(self):
        return self.params.min_inlier_count

    @min_inlier_count.setter
    def min_inlier_count(self, value):
        self.params.min_inlier_count = int(value)

    # -- device work ---------------------------------------------------------------------
    def _prepare(self, points):
        import torch
        assert points.dtype == torch.float32 and points.is_contiguous() and points.shape[-1] == 3
        count = points.numel() // 3
        need = lib().vk_detect_workspace_bytes(count)
        if self._workspace is None or self._workspace.numel() < need:
            self._workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
        if self.inliers is None or self.inliers.shape[0] < max(count, 1):
            self.inliers = torch.empty((max(count, 1), 3), dtype=torch.float32, device=self.device)
        return count

    def filter(self, points):
        """Detector::Filter; returns nothing, see read_state()/inlier_points()."""
        count = self._prepare(points)
        check(lib().vk_detect_filter(_ref(self.params), _ptr(points), count, _ptr(self.inliers), _ptr(self.state),
                                     _ptr(self._workspace), stream()), "vk_detect_filter")

    def enqueue(self, points):
        """Detector::Detect without the readback."""
        count = self._prepare(points)
        check(lib().vk_detect(_ref(self.params), _ptr(points), count, _ptr(self.inliers), _ptr(self.state),
                              _ptr(self._workspace), stream()), "vk_detect")

    def read_state(self):
        return T.DetectState.from_buffer_copy(self.state.cpu().numpy().tobytes())

    def inlier_points(self):
        return self.inliers[: self.read_state().inlier_count]

    def detect(self, points):
        """Detector::Detect: the box position as 3 floats (NaN when not detected)."""
        self.enqueue(points)
        return np.array(self.read_state().position, dtype=np.float32)
