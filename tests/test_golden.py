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


def test_oracle_reproduces_the_soak_drift_fixtures_first_report(orc):
    """tests/golden/soak_oracle_drift.json (make_soak_oracle_drift.py: the oracle's OWN tracked loop over the soak's sequence —
    what tools/soak.py compares the device's creeping pose with, profiles/r06_soak.json finding 2): the first 100 frames of that
    loop still give the file's first report, to the bit (30 s of oracle; the whole file took ten minutes, once)."""
    import json
    import subprocess
    path = os.path.join(os.path.dirname(mf.SOAK_FILE), "soak_oracle_drift.json")
    golden = json.load(open(path))["reports"]
    assert [r["frame"] for r in golden][:3] == [100, 200, 300] and golden[-1]["frame"] == 2000
    out = os.path.join(os.environ.get("TMPDIR", "/tmp"), "soak_drift_check.json")
    script = os.path.join(os.path.dirname(mf.SOAK_FILE), "make_soak_oracle_drift.py")
    subprocess.run([sys.executable, script, "--frames", "100", "--every", "100", "--out", out], check=True, stdout=subprocess.DEVNULL,
                   timeout=600)
    got = json.load(open(out))["reports"][0]
    for key in ("pose_error_max", "pose_error_last", "allocated_blocks", "dropped_requests"):
        assert got[key] == golden[0][key], key
    # and what the file says: the error grows steadily (the algorithm's creep), nothing is dropped in 2 000 frames
    assert golden[-1]["pose_error_max"]["translation_m"] > 1.8 * golden[0]["pose_error_max"]["translation_m"]
    assert all(r["dropped_requests"] == 0 for r in golden)
