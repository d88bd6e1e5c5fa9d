"""Pins the oracle's integrate arithmetic against the reference's own expected-
value loop, integrator_test.cu:82-221 (`Integrator.Integrate`): 160x120 depth
== 1.5, f = 80, c = (80, 60), identity pose, voxel .008, trunc .02; distance and
weight within 1e-5, voxels that project onto the image border exempt
(:160-168,203); a second Integrate leaves the distance and doubles the weight
(:210-220). Also BASELINE configs[0]: one 640x480 frame into a dense 128^3
region (4096 blocks) on the CPU integrator.
"""
import numpy as np

import scenes
from vulcan_amd import vk_types as T

W, H = 160, 120
TRUNC, VOXEL = 0.02, 0.008


def _expected(v, frame, trunc, max_weight=16.0):
    """integrator_test.cu:141-199 vectorised over the visible blocks (float32)."""
    F = np.float32
    vis = v.visible()
    ent = v.hash_entries[vis]
    block_length = F(8) * F(v.voxel_length)
    z, y, x = np.meshgrid(np.arange(8), np.arange(8), np.arange(8), indexing="ij")
    off = np.stack([x, y, z], -1).reshape(-1, 3).astype(np.float32)
    voxel_offset = F(v.voxel_length) * (off + F(0.5))
    Xwp = (block_length * ent["block"]["origin"].astype(np.float32))[:, None, :] + voxel_offset[None]
    Xcp = Xwp  # identity pose
    k = frame.depth_projection
    inv_w = F(1) / Xcp[..., 2]
    u = inv_w * F(k.fx) * Xcp[..., 0] + F(k.cx)
    w_ = inv_w * F(k.fy) * Xcp[..., 1] + F(k.cy)
    border = (np.abs(u) < 1e-6) | (np.abs(u - W) < 1e-6) | (np.abs(w_) < 1e-6) | (np.abs(w_ - H) < 1e-6)
    inside = (u >= 0) & (u < W) & (w_ >= 0) & (w_ < H)
    px = np.clip(u.astype(np.int64), 0, W - 1)
    py = np.clip(w_.astype(np.int64), 0, H - 1)
    depth = frame.depth[py, px]
    distance = depth - Xcp[..., 2]
    upd = inside & (distance > -F(trunc))
    exp_dist = np.where(upd, np.minimum(F(1), distance / F(trunc)), F(1)).astype(np.float32)
    exp_w = np.where(upd, 1, 0)
    index = ent["data"].astype(np.int64)[:, None] * 512 + np.arange(512)[None]
    return index, exp_dist, exp_w, border


def test_integrate_matches_reference_test(orc):
    frame = orc.HostFrame(scenes.plane(W, H, 1.5), T.Projection.make(80, 80, 80, 60),
                          color=scenes.constant_color(W, H, (1, 2, 3)))
    v = orc.HostVolume(4096, 2048, voxel_length=VOXEL, truncation_length=TRUNC)
    v.set_view(frame)
    assert v.visible_count > 0 and v.counters[T.VK_CTR_DROPPED] == 0
    orc.integrate_depth(v, frame)
    orc.integrate_color(v, frame)

    index, exp_dist, exp_w, border = _expected(v, frame, TRUNC)
    found = v.voxels[index]
    check = ~border | ((exp_w > 0) & (found["distance_weight"] > 0))
    assert np.abs(found["distance"] - exp_dist)[check].max() <= 1e-5
    assert np.array_equal(found["distance_weight"][check], exp_w[check])
    assert (exp_w > 0).sum() > 10000
    # voxels outside the visible list are untouched
    untouched = np.ones(len(v.voxels), bool)
    untouched[index.reshape(-1)] = False
    assert np.all(v.voxels["distance_weight"][untouched] == 0)
    # colour follows the same running mean with weight 1: constant (1,2,3) where |D| < 1
    near = (np.abs(found["distance"]) < 1) & (found["color_weight"] > 0)
    assert near.sum() > 1000
    np.testing.assert_allclose(found["color"][near], np.broadcast_to([1, 2, 3], found["color"][near].shape), atol=1e-5)

    orc.integrate_depth(v, frame)  # integrator_test.cu:210-220
    found = v.voxels[index]
    assert np.abs(found["distance"] - exp_dist)[check].max() <= 1e-5
    assert np.array_equal(found["distance_weight"][check], 2 * exp_w[check])


def test_weight_clamps_at_max(orc):
    """integrator.cu:7-13 max weight 16; depth_integrator.cu:72 clamp."""
    frame = orc.HostFrame(scenes.plane(W, H, 1.5), T.Projection.make(80, 80, 80, 60))
    v = orc.HostVolume(4096, 2048, voxel_length=VOXEL, truncation_length=TRUNC)
    v.set_view(frame)
    p = T.Integrator(0.1, 5.0, 3.0, 16.0)
    for _ in range(5):
        orc.integrate_depth(v, frame, p)
    assert v.voxels["distance_weight"].max() == 3


def test_config0_dense_128cubed_cpu(orc):
    """BASELINE configs[0]: 640x480 depth frame -> dense 128^3 TSDF (16^3 blocks
    of 8^3) on the CPU integrator; closed form for a fronto-parallel plane."""
    w, h = 640, 480
    voxel, trunc = 0.005, 0.04
    frame = orc.HostFrame(scenes.plane(w, h, 1.5), T.Projection.make(*scenes.APP_INTRINSICS))
    v = orc.HostVolume(4096, 0, voxel_length=voxel, truncation_length=trunc)
    # dense region: blocks [-8,8) x [-8,8) x [30,46) -> z in [1.2, 1.84), straddles the plane
    n = 0
    for bz in range(30, 46):
        for by in range(-8, 8):
            for bx in range(-8, 8):
                v.hash_entries["block"]["origin"][n] = (bx, by, bz)
                v.hash_entries["data"][n] = n
                v.visible_blocks[n] = n
                n += 1
    v.counters[T.VK_CTR_VISIBLE] = n
    orc.integrate_depth(v, frame)

    F = np.float32
    zc = (F(8 * voxel) * np.arange(30, 46, dtype=np.float32))[:, None] + F(voxel) * (np.arange(8, dtype=np.float32) + F(0.5))[None]
    zc = zc.reshape(-1)                                # 128 voxel-centre depths
    dist = F(1.5) - zc
    exp = np.where(dist > -F(trunc), np.minimum(F(1), dist / F(trunc)), F(1)).astype(np.float32)
    vox = v.voxels[:n * 512].reshape(16, 16, 16, 8, 8, 8)   # bz,by,bx,z,y,x
    got = vox["distance"].transpose(0, 3, 1, 4, 2, 5).reshape(128, 128, 128)   # Z, Y, X
    assert np.array_equal(got, np.broadcast_to(exp[:, None, None], got.shape))
    wgt = vox["distance_weight"].transpose(0, 3, 1, 4, 2, 5).reshape(128, 128, 128)
    assert np.array_equal(wgt, np.broadcast_to((dist > -F(trunc)).astype(np.int16)[:, None, None], wgt.shape))
