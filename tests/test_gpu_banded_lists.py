"""The banded visible lists (include/vk.h VK_BANDS, VK_CTR_BANDED; round 4): the visibility pass of
vk_volume_set_view* lists every visible entry a second time, binned by the image row band of a depth pixel whose
ray touched the block, and the integrate kernels deal the bands to the XCDs. No reference counterpart: the order in
which visible blocks are integrated is unspecified upstream (volume.cu:80-83) and does not change a voxel — which
the parity tests check with the lists in use (tests/test_gpu_configs.py integrates the bench sequence through them).
Here: the lists themselves.
"""
import numpy as np
import pytest

import scenes
from test_gpu_parity import api, assert_volume_equal, frames, make_pair, sync  # noqa: F401
from vulcan_amd import vk_types as T

pytestmark = pytest.mark.gpu

BASE = T.VK_CTR_COUNT - T.VK_BANDS - T.VK_BANDS * T.VK_BAND_SLOTS


def banded(dv):
    sync()
    c = dv.counters.cpu().numpy()
    sizes = c[BASE:BASE + T.VK_BANDS]
    lists = c[BASE + T.VK_BANDS:].reshape(T.VK_BANDS, T.VK_BAND_SLOTS)
    return int(c[T.VK_CTR_BANDED]), int(c[T.VK_CTR_VISIBLE]), sizes, lists


def test_every_visible_entry_is_in_exactly_one_band_and_the_bands_follow_the_image_rows(api, orc):
    import bench
    k = T.Projection.make(*scenes.APP_INTRINSICS)
    depth = bench.sphere_room_depth(k)
    hf, df = frames(api, orc, depth, k, T.Transform.identity())
    hv, dv = make_pair(api, orc, bench.MAIN, bench.EXCESS, bench.VOXEL, bench.TRUNC)
    valid_frames = 0
    for i in range(4):
        pose = scenes.orbit_pose(i, bench.YAW_STEP)
        hf.depth_to_world = df.depth_to_world = pose
        for _ in range(3):
            hv.set_view(hf, orc.POLICY_MAXKEY)
        dv.set_view(df, rounds=3)
        assert_volume_equal(dv, hv, voxels=False)
        flag, visible, sizes, lists = banded(dv)
        assert flag in (-1, visible)
        if flag != visible:
            continue                      # later rounds appended to the plain list (the first frame: 7 k requests at once)
        valid_frames += 1
        assert sizes.sum() == visible and (sizes <= T.VK_BAND_SLOTS).all() and (sizes > 0).all()
        members = np.concatenate([lists[b, :sizes[b]] for b in range(T.VK_BANDS)])
        assert np.array_equal(np.sort(members), np.sort(dv.visible()))           # a partition of the visible list
        # the bands are bands of image rows: project every block's centre (float64) and compare
        entries = dv.host_entries()
        inv = pose.inverse_matrix().astype(np.float64)
        block = 8 * np.float64(np.float32(bench.VOXEL))
        inside = []
        for b in range(T.VK_BANDS):
            origin = entries["block"]["origin"][lists[b, :sizes[b]]].astype(np.float64)
            cam = ((origin + 0.5) * block) @ inv[:3, :3].T + inv[:3, 3]
            row = k.fy * cam[:, 1] / cam[:, 2] + k.cy
            lo, hi = b * bench.H / T.VK_BANDS, (b + 1) * bench.H / T.VK_BANDS
            margin = 30.0                # a block at 2 m spans ~11 rows; any of its rays may have tagged it
            inside.append(float(((row > lo - margin) & (row < hi + margin)).mean()))
        assert min(inside) > 0.97, inside
    assert valid_frames >= 2


def test_the_staged_visibility_pass_invalidates_the_lists(api, orc):
    w, h = 320, 240
    k = T.Projection.make(272.0, 272.0, 155.6, 117.4)
    hf, df = frames(api, orc, scenes.plane(w, h, 1.5), k, scenes.tracer_test_pose(), color=scenes.constant_color(w, h))
    hv, dv = make_pair(api, orc, 8192, 4096, 0.01, 0.04)
    for _ in range(4):
        hv.set_view(hf, orc.POLICY_MAXKEY)
        dv.set_view(df)
    flag, visible, sizes, _ = banded(dv)
    assert flag == visible > 100 and sizes.sum() == visible
    integ = api.ColorIntegrator(dv)
    orc.integrate_depth(hv, hf)
    orc.integrate_color(hv, hf)
    integ.integrate(df)                                   # through the banded lists
    assert_volume_equal(dv, hv)
    hv.update_block_visibility(hf)                        # volume.cu:473-495 as a stage of its own
    dv.update_block_visibility(df)
    flag, visible, _, _ = banded(dv)
    assert flag == -1 and visible == hv.visible_count
    orc.integrate_depth(hv, hf)
    orc.integrate_color(hv, hf)
    integ.integrate(df)                                   # through the plain list
    assert_volume_equal(dv, hv, voxels=True)
