"""The committed golden vectors (tests/golden/scenes_160x120.npz, written by
tests/golden/make_fixtures.py from the CPU oracle) still are what the oracle computes: the
file and the oracle cannot drift apart unnoticed."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_fixtures as mf  # noqa: E402


@pytest.fixture(scope="module")
def golden():
    assert os.path.exists(mf.FILE), "run python tests/golden/make_fixtures.py"
    return np.load(mf.FILE)


def compare(got, golden, name):
    for key, value in got.items():
        want = golden[f"{name}/{key}"]
        value = np.asarray(value)
        if value.dtype.kind in "US":
            assert str(value) == str(want), (name, key)
        elif value.dtype.kind == "f":
            assert np.array_equal(value.view(np.uint32), want.view(np.uint32)), (name, key)     # bit for bit, NaN included
        else:
            assert np.array_equal(value, want), (name, key)


@pytest.mark.parametrize("name", mf.SCENES)
def test_oracle_reproduces_the_golden_vectors(orc, golden, name):
    orc.set_threads(8)
    compare(mf.run_scene(mf.oracle_backend, name), golden, name)
    orc.set_threads(1)


def test_fixture_is_small_and_complete(golden):
    assert os.path.getsize(mf.FILE) < 4 << 20
    for name in mf.SCENES:
        for key in ("depth", "color", "normals", "bounds", "counters", "voxels_sha256", "mesh_counts", "icp_residuals"):
            assert f"{name}/{key}" in golden.files
        assert golden[f"{name}/depth"].shape == (mf.H, mf.W)
        assert (golden[f"{name}/depth"] > 0).sum() > 1000           # the scene was really raycast


def test_oracle_reproduces_the_step_cap_vectors(orc):
    """tests/golden/step_cap_64x48.npz: the hand-built slab whose rays reach the march's 500-step cap (tracer.cu:437-442)"""
    assert os.path.exists(mf.STEP_CAP_FILE), "run python tests/golden/make_fixtures.py --step-cap-only"
    golden = np.load(mf.STEP_CAP_FILE)
    orc.set_threads(8)
    compare(mf.step_cap_oracle(), golden, "step_cap")
    orc.set_threads(1)
    color, depth = golden["step_cap/color"], golden["step_cap/depth"]
    capped = (color[..., 0] == 1) & (color[..., 1] == 0) & (color[..., 2] == 0)
    assert capped.sum() > 1200 and np.all(depth[capped] == 0) and (depth > 0).sum() > 1200
    assert os.path.getsize(mf.STEP_CAP_FILE) < 200 << 10


def test_oracle_reproduces_the_soak_fixtures_first_checkpoint(orc):
    """tests/golden/soak_room_640x480.json (make_fixtures.py --soak; tools/soak.py holds the device to it over 2 000 frames):
    the oracle's first 20 frames of the looped room sequence at the bench's size still give the digests on file — the
    2 000-frame file and the oracle cannot drift apart unnoticed (the whole run takes the oracle five minutes: made once)."""
    import json
    assert os.path.exists(mf.SOAK_FILE), "run python tests/golden/make_fixtures.py --soak"
    golden = json.load(open(mf.SOAK_FILE))["checkpoints"]
    assert sorted(int(n) for n in golden) == sorted(mf.SOAK_CHECKPOINTS)
    orc.set_threads(8)
    got = mf.soak_oracle(20, checkpoints=(20,))
    orc.set_threads(1)
    assert got["20"] == golden["20"]
    # the later checkpoints: no request was ever dropped, and the table stops changing once the camera has seen the room
    assert all(golden[n]["dropped"] == 0 for n in golden)
    assert golden["240"]["entries_sha256"] == golden["2000"]["entries_sha256"] != golden["20"]["entries_sha256"]
    assert golden["240"]["voxels_sha256"] != golden["2000"]["voxels_sha256"]
