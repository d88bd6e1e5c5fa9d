"""The input side of a frame (vk.h vk_memcpy_h2d_async / vk_event_create_ordering / vk_stream_wait_event;
ref: include/vulcan/image.h:100-123 and apps/vulcan/vulcan.cu:220,232, where every frame is uploaded with a blocking
copy): `bench.py --stream-input` uploads the frame's depth and colour images every frame — pinned staging, a copy
stream, two slots of device images, frame i + 1 crossing while frame i is fused. The streamed loop must leave the very
volume and raycasts the resident loop leaves (whose parity with the oracle is tests/test_gpu_configs.py's business),
also when the two slots hold DIFFERENT images, i.e. when a slot that is overwritten too early or read too early shows."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import scenes
from test_gpu_parity import api, sync  # noqa: F401
from vulcan_amd import vk_types as T

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_streamed_input_equals_resident_input(api):
    sys.path.insert(0, ROOT)
    import bench
    count = 10                     # every slot is used more than once
    poses = [scenes.orbit_pose(i, bench.YAW_STEP) for i in range(count)]
    # (no request pass made ahead: the images alternate, and FrameLoop announces the next frame with the current images)
    resident = bench.FrameLoop("rgbd", poses, requests_ahead=False)
    streamed = bench.FrameLoop("rgbd", poses, stream_input=True)
    # the slots' staging buffers hold different frames: even frames the bench's image, odd frames a dimmer one
    # 3 mm further away; the resident loop is handed the same alternation
    depth_b = (streamed.depth_np + np.float32(0.003)).astype(np.float32)
    color_b = (streamed.color_np * np.float32(0.8)).astype(np.float32)
    assert streamed.upload.SLOTS % 2 == 0
    for slot in streamed.upload.slots[1::2]:
        (pinned_d, _, nbytes_d), (pinned_c, _, nbytes_c) = slot["images"]
        C.memmove(pinned_d, depth_b.ctypes.data, nbytes_d)
        C.memmove(pinned_c, color_b.ctypes.data, nbytes_c)
    import torch
    res_d = [torch.from_numpy(streamed.depth_np).cuda(), torch.from_numpy(depth_b).cuda()]
    res_c = [torch.from_numpy(streamed.color_np).cuda(), torch.from_numpy(color_b).cuda()]
    for i in range(count):
        resident.fdesc.depth, resident.fdesc.color = res_d[i & 1].data_ptr(), res_c[i & 1].data_ptr()
        resident.step(i)
        streamed.step(i)
        sync()
        for name in ("depth", "color", "normals"):
            a, b = getattr(resident.key, name).cpu().numpy(), getattr(streamed.key, name).cpu().numpy()
            assert np.array_equal(a, b, equal_nan=True), (i, name)
    va, vb = resident.vols[0]["vol"], streamed.vols[0]["vol"]
    assert va.visible_count == vb.visible_count > 5000
    assert np.array_equal(va.host_entries(), vb.host_entries())
    assert va.host_voxels().tobytes() == vb.host_voxels().tobytes()
    assert streamed.upload.bytes_per_frame == bench.W * bench.H * 16
    assert streamed.upload.h2d_rate_GBps(reps=5) > 1.0
