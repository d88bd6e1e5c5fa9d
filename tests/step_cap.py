"""The step-cap case (tests/scenes.py STEP_CAP_*; ref: src/tracer.cu:437-442) built on either side: the slab is allocated by
the side's own SetView, filled by the closed form scenes.step_cap_fill, and raycast with the reference's truncation."""
import numpy as np

import scenes
from vulcan_amd import vk_types as T


def inputs():
    w, h = scenes.STEP_CAP_SIZE
    return w, h, T.Projection.make(*scenes.STEP_CAP_INTRINSICS), scenes.step_cap_pose(), \
        scenes.plane(w, h, scenes.STEP_CAP_ALLOC_DEPTH)


def build_host(orc):
    """-> (HostVolume ready to be raycast, HostFrame of the view)"""
    w, h, k, pose, depth = inputs()
    hv = orc.HostVolume(*scenes.STEP_CAP_VOLUME, voxel_length=scenes.STEP_CAP_VOXEL, truncation_length=scenes.STEP_CAP_ALLOC_TRUNC)
    hf = orc.HostFrame(depth, k, pose)
    prev = -1
    for _ in range(64):                                   # tracer_test.cu:298-303: until the visible count settles
        hv.set_view(hf, orc.POLICY_MAXKEY)
        if hv.visible_count == prev:
            break
        prev = hv.visible_count
    assert hv.counters[T.VK_CTR_DROPPED] == 0
    scenes.step_cap_fill(hv.hash_entries, hv.voxels, pose)
    hv.truncation_length = scenes.STEP_CAP_TRUNC
    return hv, hf


def build_device(api):
    """-> (api.Volume, api.Frame): the device's own SetView allocates, the voxels are filled on the host and uploaded"""
    import torch
    w, h, k, pose, depth = inputs()
    dv = api.Volume(*scenes.STEP_CAP_VOLUME, voxel_length=scenes.STEP_CAP_VOXEL, truncation_length=scenes.STEP_CAP_ALLOC_TRUNC)
    df = api.Frame(depth, k, pose)
    prev = -1
    for _ in range(64):
        dv.set_view(df)
        if dv.visible_count == prev:
            break
        prev = dv.visible_count
    assert dv.read_counters()[T.VK_CTR_DROPPED] == 0
    entries, voxels = dv.host_entries(), dv.host_voxels().copy()
    scenes.step_cap_fill(entries, voxels, pose)
    dv.voxels.copy_(torch.from_numpy(np.frombuffer(voxels.tobytes(), dtype=np.uint8).copy()).to(dv.device))
    dv.truncation_length = scenes.STEP_CAP_TRUNC
    dv._view_changed()
    return dv, df


def classify(depth, color):
    """(capped, hit) pixel masks: the cap paints (1, 0, 0) and leaves depth 0 (tracer.cu:437-442)"""
    capped = (color[..., 0] == 1) & (color[..., 1] == 0) & (color[..., 2] == 0)
    return capped, depth > 0
