"""The C++ class layer (vulcan_amd/host: Volume / Integrator / Tracer / Frame /
DepthTracker / PyramidTracker over the C ABI) runs the reference's gtest cases,
re-authored in vulcan_amd/host/tests/host_tests.cpp, and the demo frame loop."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "vulcan_amd", "host", "bin")


@pytest.mark.gpu
def test_cpp_host_tests_pass():
    exe = os.path.join(BIN, "host_tests")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    proc = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    print(proc.stdout)
    assert proc.returncode == 0, proc.stdout[-4000:]
    assert re.search(r"\d+ test\(s\), 0 failed", proc.stdout)
    for name in ("Volume.CreateAllocationRequests", "Integrator.Integrate", "Tracer.ComputePoints",
                 "Tracer.ComputeNormals", "DepthTracker.Jacobian", "PyramidTracker.Track"):
        assert f"[  OK  ] {name}" in proc.stdout


@pytest.mark.gpu
def test_cpp_frame_loop_runs_and_tracks():
    exe = os.path.join(BIN, "fuse_sequence")
    out = subprocess.run([exe, "60", "0"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600).stdout
    m = re.search(r"frames 60 .* fps ([\d.]+) +visible (\d+) +allocated (\d+) +dropped (\d+)", out)
    assert m, out
    assert int(m.group(2)) > 3000 and int(m.group(4)) == 0
    # mode 1: PyramidTracker<DepthTracker> in front of every frame; mode 2: the shipped app's
    # set-up (PyramidTracker<LightTracker> + LightIntegrator). The depth image is the same from
    # every pose, so tracking must hold the pose it starts from.
    for mode, label in (("1", "depth"), ("2", "light")):
        out = subprocess.run([exe, "60", mode], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600).stdout
        assert re.search(r"dropped 0 +tracking " + label, out), out
        row = re.search(r"final pose row0: ([-\d.]+) ([-\d.]+) ([-\d.]+) ([-\d.]+)", out)
        assert row and abs(float(row.group(1)) - 1.0) < 1e-3 and abs(float(row.group(4))) < 0.01, out


def test_host_layer_builds_and_links_only_the_c_abi():
    """libvulcan.so needs libvk_hip.so and nothing HIP-specific of its own."""
    lib = os.path.join(ROOT, "vulcan_amd", "lib", "libvulcan.so")
    assert os.path.exists(lib), "run __graft_entry__.build() first"
    needed = subprocess.run(["readelf", "-d", lib], stdout=subprocess.PIPE, text=True).stdout
    assert "libvk_hip.so" in needed and "libamdhip64" not in needed
    syms = subprocess.run(["nm", "-D", "--defined-only", lib], stdout=subprocess.PIPE, text=True).stdout
    for cls in ("Volume", "DepthIntegrator", "ColorIntegrator", "LightIntegrator", "Tracer", "DepthTracker",
                "PyramidTracker"):
        assert f"6vulcan{len(cls)}{cls}" in syms or cls in syms, cls
