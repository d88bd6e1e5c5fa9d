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
    # the same loop with every raycast announcing the next frame (Tracer::Trace(keyframe, next_frame)): the same map
    out = subprocess.run([exe, "60", "0", "0", "0", "1"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600).stdout
    ahead = re.search(r"frames 60 .* fps ([\d.]+) +visible (\d+) +allocated (\d+) +dropped (\d+) +input resident, requests made ahead", out)
    assert ahead, out
    assert ahead.group(2, 3, 4) == m.group(2, 3, 4)
    # mode 1: PyramidTracker<DepthTracker> in front of every frame; mode 2: PyramidTracker<LightTracker> +
    # LightIntegrator (the line upstream keeps commented out); mode 3, the shipped app's own set-up, is run by
    # tools/round_run.sh (its one-step tracker lags this fast camera: profiles/). Closed loop in the room scene
    # (room_scene.h): the camera turns ~24 degrees and moves ~0.35 m over 60 frames, every frame
    # is fused and raycast at its TRACKED pose, and the true poses only score the result.
    import json
    for mode, label in (("1", "depth"), ("2", "light")):
        out = subprocess.run([exe, "60", mode], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600).stdout
        assert re.search(r"dropped 0 +input resident +tracking " + label, out), out
        line = [text for text in out.splitlines() if text.startswith("{")]
        assert line, out
        rec = json.loads(line[-1])
        assert rec["tracked_pose_drives_fusion"] is True and rec["frames"] == 60
        assert rec["camera_motion_over_run"]["rotation_deg"] > 15 and rec["camera_motion_over_run"]["translation_m"] > 0.2
        print(mode, rec["pose_error_max"], rec["gn_steps_median"], rec["frames_per_s"])
        # the geometric tracker holds the truth to ~1 mm; the photometric one (shading model on a
        # lamp-lit, mostly dark room) drifts by centimetres — still a small fraction of the motion
        limit_m, limit_deg = (0.01, 0.2) if mode == "1" else (0.06, 2.0)
        assert rec["pose_error_max"]["translation_m"] < limit_m and rec["pose_error_max"]["rotation_deg"] < limit_deg, rec
        assert 1 <= rec["gn_steps_median"] <= 20 and 1.0 <= rec["set_view_rounds_run_per_frame"] <= 3.0


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
