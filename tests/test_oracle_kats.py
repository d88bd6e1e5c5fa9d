"""Pins the CPU oracle's math layer and volume bookkeeping against the
reference's known answers (tests/golden/reference_kats.json) and
the reference's host-only / white-box volume tests:
  block_test.cpp:7-10, hash_test.cpp:9-54, voxel_test.cpp:8-37,
  volume_test.cpp:35-98 (Constructor), :100-122 (ResetBlockVisibility),
  :124-248 (UpdateBlockVisibility), :433-556 (HandleAllocationRequests).
"""
import ctypes as C
import json
import os

import numpy as np

from vulcan_amd import vk_types as T

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")
MAIN, EXCESS = 1024, 512   # volume_test.cpp:12-14
MAX = MAIN + EXCESS


def kats():
    with open(GOLDEN) as f:
        return json.load(f)


def test_pod_sizes(orc):
    k = kats()["sizeof"]
    out = (C.c_int * 8)()
    orc.lib().orc_kat_sizes(out)
    assert list(out)[:7] == [k["Voxel"], k["Block"], k["HashEntry"], k["Patch"],
                             k["Projection"], k["Transform"], k["Light"]]
    assert out[7] == kats()["offsets"]["Voxel.distance_weight"]
    assert T.voxel_dtype.itemsize == k["Voxel"] and T.hash_entry_dtype.itemsize == k["HashEntry"]
    assert T.voxel_dtype.fields["color"][1] == 4 and T.voxel_dtype.fields["color_weight"][1] == 18
    assert T.hash_entry_dtype.fields["data"][1] == 8 and T.hash_entry_dtype.fields["next"][1] == 12
    assert C.sizeof(T.Projection) == 16 and C.sizeof(T.Transform) == 128 and C.sizeof(T.Light) == 16


def test_hash_known_answers(orc):
    """hash.h / volume.cu:168-180: ((bx*P1) ^ (by*P2) ^ (bz*P3)) % K in uint32 — recomputed here
    with Python integers, so the stored codes are checked twice."""
    f = orc.lib().orc_kat_hash
    f.restype = C.c_uint32
    P1, P2, P3 = 73856093, 19349669, 83492791
    for case in kats()["hash"]:
        bx, by, bz = case["block"]
        by_hand = (((bx * P1) & 0xffffffff) ^ ((by * P2) & 0xffffffff) ^ ((bz * P3) & 0xffffffff)) % case["K"]
        assert by_hand == case["code"]
        assert f(bx, by, bz, C.c_uint32(case["K"])) == case["code"]
    rng = np.random.default_rng(0)
    for _ in range(200):
        bx, by, bz = (int(v) for v in rng.integers(-3000, 3000, 3))
        K = int(rng.integers(1, 200000))
        by_hand = (((bx * P1) & 0xffffffff) ^ ((by * P2) & 0xffffffff) ^ ((bz * P3) & 0xffffffff)) % K
        assert f(bx, by, bz, C.c_uint32(K)) == by_hand


def test_projection_known_answer(orc):
    for case in kats()["project"]:
        k = T.Projection.make(*case["projection"])
        uv = (C.c_float * 2)()
        x, y, z = case["point"]
        orc.lib().orc_kat_project(C.byref(k), C.c_float(x), C.c_float(y), C.c_float(z), uv)
        # projection.h:63-70 in numpy float32: inv_w = 1 / z; u = inv_w * fx * x + cx
        F = np.float32
        inv_w = F(1) / F(z)
        assert F(F(F(inv_w * F(k.fx)) * F(x)) + F(k.cx)) == F(case["uv"][0])
        assert F(F(F(inv_w * F(k.fy)) * F(y)) + F(k.cy)) == F(case["uv"][1])
        assert np.float32(uv[0]) == np.float32(case["uv"][0])
        assert np.float32(uv[1]) == np.float32(case["uv"][1])
        xyz = (C.c_float * 3)()
        orc.lib().orc_kat_unproject(C.byref(k), C.c_float(uv[0]), C.c_float(uv[1]), C.c_float(z), xyz)
        np.testing.assert_allclose(list(xyz), [x, y, z], atol=1e-6)


def test_volume_constructor(orc):
    """volume_test.cpp:35-98"""
    v = orc.HostVolume(MAIN, EXCESS)
    e = kats()["voxel_empty"]
    assert np.all(v.voxels["distance"] == e["distance"]) and np.all(v.voxels["color"] == 0)
    assert np.all(v.voxels["distance_weight"] == 0) and np.all(v.voxels["color_weight"] == 0)
    assert np.all(v.hash_entries["data"] == -1) and np.all(v.hash_entries["next"] == -1)
    assert np.all(v.hash_entries["block"]["origin"] == 0)
    assert np.all(v.block_visibility == T.VISIBILITY_FALSE)
    assert np.all(v.allocation_types == T.ALLOC_NONE)
    assert np.array_equal(v.free_voxel_blocks, np.arange(MAX, dtype=np.int32))
    assert v.visible_count == 0
    assert v.counters[T.VK_CTR_VOXEL_PTR] == MAX - 1 and v.counters[T.VK_CTR_EXCESS_PTR] == MAIN


def test_reset_block_visibility(orc):
    """volume_test.cpp:100-122"""
    v = orc.HostVolume(MAIN, EXCESS)
    v.block_visibility[:] = np.where(np.arange(MAX) % 7 == 0, T.VISIBILITY_TRUE, T.VISIBILITY_FALSE)
    v.reset_block_visibility()
    want = np.where(np.arange(MAX) % 7 == 0, T.VISIBILITY_UNKNOWN, T.VISIBILITY_FALSE)
    assert np.array_equal(v.block_visibility, want)


def _visibility_frame(orc):
    # volume_test.cpp:132-135
    return orc.HostFrame(np.zeros((480, 640), np.float32), T.Projection.make(320, 320, 320, 240),
                         T.Transform.translate(10, -2, 30))


def test_update_block_visibility(orc):
    """volume_test.cpp:124-248: nothing / three TRUE / three TRUE + three UNKNOWN
    of which two are inside the frustum."""
    v = orc.HostVolume(MAIN, EXCESS)
    frame = _visibility_frame(orc)

    v.update_block_visibility(frame)
    assert v.visible_count == 0

    v.block_visibility[[7, 32, 123]] = T.VISIBILITY_TRUE
    v.update_block_visibility(frame)
    assert sorted(v.visible().tolist()) == [7, 32, 123]

    v.block_visibility[[3, 17, 315]] = T.VISIBILITY_UNKNOWN
    scale = np.float32(1.0) / (np.float32(8) * np.float32(v.voxel_length))
    for idx, p in ((3, (10, -2, 33)), (17, (-10, -2, 28)), (315, (11, -1, 53))):
        v.hash_entries["block"]["origin"][idx] = [np.int16(int(np.float32(c) * scale)) for c in p]
    v.update_block_visibility(frame)
    assert sorted(v.visible().tolist()) == [3, 7, 32, 123, 315]
    assert v.block_visibility[17] == T.VISIBILITY_FALSE
    # UNKNOWN entries found visible keep UNKNOWN (volume.cu:75 only writes FALSE)
    assert v.block_visibility[3] == T.VISIBILITY_UNKNOWN and v.block_visibility[315] == T.VISIBILITY_UNKNOWN


def test_handle_allocation_requests(orc):
    """volume_test.cpp:433-556 — slots in {last, last-1}, excess indices in
    {main, main+1}; the oracle's serial order picks the first of each pair."""
    v = orc.HostVolume(MAIN, EXCESS)
    v.allocation_blocks["origin"][0] = (1, 2, 3)
    v.allocation_types[0] = T.ALLOC_MAIN
    v.allocation_blocks["origin"][323] = (7, 3, -1)
    v.allocation_types[323] = T.ALLOC_MAIN
    v.handle_allocation_requests()
    assert np.all(v.allocation_types == T.ALLOC_NONE)

    e0, e323 = v.hash_entries[0], v.hash_entries[323]
    assert tuple(e0["block"]["origin"]) == (1, 2, 3) and e0["next"] == -1
    assert tuple(e323["block"]["origin"]) == (7, 3, -1) and e323["next"] == -1
    assert {int(e0["data"]), int(e323["data"])} == {MAX - 1, MAX - 2}

    v.allocation_blocks["origin"][0] = (7, 3, 0)
    v.allocation_types[0] = T.ALLOC_EXCESS
    v.allocation_blocks["origin"][323] = (-9, 1, -2)
    v.allocation_types[323] = T.ALLOC_EXCESS
    v.handle_allocation_requests()
    assert np.all(v.allocation_types == T.ALLOC_NONE)

    e0, e323 = v.hash_entries[0], v.hash_entries[323]
    assert tuple(e0["block"]["origin"]) == (1, 2, 3) and tuple(e323["block"]["origin"]) == (7, 3, -1)
    assert {int(e0["next"]), int(e323["next"])} == {MAIN, MAIN + 1}
    n0, n323 = v.hash_entries[e0["next"]], v.hash_entries[e323["next"]]
    assert tuple(n0["block"]["origin"]) == (7, 3, 0) and n0["next"] == -1
    assert tuple(n323["block"]["origin"]) == (-9, 1, -2) and n323["next"] == -1
    assert {int(n0["data"]), int(n323["data"])} == {MAX - 3, MAX - 4}
    assert v.block_visibility[e0["next"]] == T.VISIBILITY_TRUE
    assert v.block_visibility[e323["next"]] == T.VISIBILITY_TRUE
