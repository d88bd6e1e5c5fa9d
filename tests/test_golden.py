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
