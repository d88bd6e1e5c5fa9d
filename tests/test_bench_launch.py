"""`python bench.py --gpus N` must start its own ranks (the round-1 bench died with
"--gpus 2 but WORLD_SIZE=1" unless a launcher had been wrapped around it). This runs the
launch path on CPU: the parent spawns fresh children with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* set, the children rendezvous over gloo, run the rig's 48-float all-reduce once
and exit without touching a GPU (`--selftest-launch`)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_self_launch_two_ranks():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "5", "--selftest-launch"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=_clean_env(), timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                      # ONE json line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["system_sum"] == 3.0       # 1 + 2: the collective ran over both ranks
    assert out["max_over_ranks"] == 2.0 and out["frames_all_ranks"] == 10.0
    assert out["local_rank_env"] == 0
    # the line says what every rank took, and whether RCCL (here: the rehearsal) saw all ranks
    assert out["per_rank_ms_per_step"] == [1.0, 2.0]
    assert out["collective"]["vk_comm_count"] == 2 and out["collective"]["update_identical_on_all_ranks"] is True


def test_self_launch_eight_ranks():
    """north_star's machine is 8 ranks; the launch path had only ever seen 2. `--gpus 8` over gloo:
    eight fresh children rendezvous, the rig's collective sees eight ranks, rank 0 prints ONE line that says what each took."""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "5", "--selftest-launch"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=_clean_env(), timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["system_sum"] == 36.0      # 1 + 2 + ... + 8
    assert out["max_over_ranks"] == 8.0 and out["frames_all_ranks"] == 40.0
    assert out["per_rank_ms_per_step"] == [float(r + 1) for r in range(8)]
    assert out["collective"]["vk_comm_count"] == 8 and out["collective"]["update_identical_on_all_ranks"] is True


def test_self_launch_reports_a_dying_rank():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--selftest-launch", "--selftest-fail-rank", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=_clean_env(), timeout=300)
    assert p.returncode != 0


def test_joins_an_external_launcher():
    """Under torchrun-style env the script must NOT spawn: it joins the job it was given,
    and refuses a job of the wrong size."""
    env = _clean_env()
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--selftest-launch"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    assert p.returncode == 2 and "WORLD_SIZE=1" in p.stderr
    p = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--selftest-launch"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    assert p.returncode == 0 and json.loads(p.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_a_rank_failing_inside_the_rig_step_ends_every_rank_non_zero():
    """A step of the rig that raises on ONE rank (bench.vk_comm_rig's steps all go through vd.agreed_step; rehearsed here
    over gloo with the same functions and the same exit path): every rank learns of it, the line is still printed — the
    headline is measured before the rig step — with the error in it, and EVERY rank exits non-zero, well within the
    rig's timeout."""
    import time
    t0 = time.time()
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "5", "--selftest-launch", "--selftest-rig-fail-rank", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=_clean_env(), timeout=300)
    assert time.time() - t0 < 120
    assert p.returncode == 4, (p.returncode, p.stderr[-2000:])
    assert "rank 0 exited 4" in p.stderr and "rank 1 exited 4" in p.stderr, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["system_sum"] == 3.0                          # what was measured before the failure is in the line
    vk = out["collective"]["vk_comm"]
    assert vk["ok"] is False and "rig step" in vk["error"]
    # rank 0 did not fail itself: it reports the peer's failure
    assert "another rank failed" in vk["error"]


def test_recorded_bench_line_carries_the_raycast_figures():
    """SURVEY 8(d)'s raycast figures were silently missing from two rounds of driver records, and
    rounds_run_per_frame read 6.0 of at most 3. The newest bench line on file (profiles/r05_*_bench.json, written by
    tools/round_run.sh on the GPU box) must carry them."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r05_*_bench.json")))
    assert files, "no round-5 bench line recorded under profiles/"
    d = json.load(open(files[-1]))
    ray = d["roofline"]["raycast"]
    assert "error" not in ray, ray.get("error")
    for key in ("blocks_touched", "algorithmic_bytes", "algorithmic_GBps", "traffic", "gather_amplification", "avg_us"):
        assert key in ray and ray[key], key
    assert 1000 < ray["blocks_touched"] < 73216
    assert 0.0 < ray["algorithmic_GBps"] < 8000.0
    rounds = d["config"]["set_view"]["rounds_run_per_frame"]
    assert 1.0 <= rounds <= d["config"]["set_view"]["max_rounds"] == 3
    assert len(d["per_rank_ms_per_step"]) == d["n_gpus"]


def test_recorded_past_l3_trace_adds_up():
    """roofline.past_l3.by_kernel_trace is read from profiles/r*_integrate_past_l3.json (rocprofv3 cannot run inside the
    bench): what is on file must follow from its own numbers."""
    sys.path.insert(0, ROOT)
    import bench
    for wl in ("rgbd", "depth"):
        d = bench.recorded_past_l3(wl)
        assert d is not None and "NOT measured in this run" in d["source"]
        assert abs(d["frac"] - d["achieved"] / bench.HBM_PEAK_GBS) < 1e-9
        assert 20.0 < d["avg_launch_us"] < 80.0 and d["launches"] >= 48
        assert d["frac"] < 1.0
    assert bench.recorded_past_l3("rgbd-icp") is None


def test_recorded_round6_bench_line_is_complete():
    """the driver-run line carries every BASELINE config and a spread. The newest line on file
    (profiles/r06_*_bench.json, written by tools/round_run.sh on the GPU box) must hold: the nine-window spread; frac_hbm beside
    frac, and what `frac` is; the bracket figure beside the dispatch figure; configs[3] (pyramid-icp) at three sizes; configs[0]
    on the device beside the CPU's; the multi-sequence legs; dropped_requests in every entry and no `error` anywhere."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r06_*_bench.json")))
    assert files, "no round-6 bench line recorded under profiles/"
    d = json.load(open(files[-1]))
    assert "error" not in d
    w = d["windows"]
    assert w["count"] >= 1 and len(w["frames_per_s"]) == w["count"] and w["min"] <= w["median"] <= w["max"]
    assert w["frames_per_s"][0] == d["value"]
    roof = d["roofline"]
    assert 0.0 < roof["frac_hbm"] < roof["frac"] < 1.0 and "cache-assisted" in roof["frac_is"]
    assert roof["frac_hbm"] == roof["frac_past_l3"]
    assert roof["timed_by"] == "dispatch" and 0.0 < roof["by_bracket"]["frac"] <= roof["frac"] * 1.02
    others = d["other_workloads"]
    for name in ("depth", "rgbd-icp", "rgbd-requests-inside-set-view", "rgbd-icp x1", "rgbd-icp x2", "rgbd-icp x4", "rgbd x1", "rgbd x2"):
        assert name in others, name
        assert "error" not in others[name], (name, others[name].get("error"))
        dropped = others[name]["dropped_requests"]
        assert dropped == 0 or (isinstance(dropped, list) and not any(dropped)), name
    for name in ("rgbd-icp x2", "rgbd-icp x4", "rgbd x2"):
        assert others[name]["aggregate_over_single"] > 0.9 and others[name]["single_sequence_value"] > 0
    pyr = others["pyramid-icp"]
    assert "error" not in pyr
    for size in ("320x240", "640x480", "1280x960"):
        for case in ("same surface", "two views"):
            e = pyr["sizes"][size][case]
            assert "error" not in e and 20.0 < e["us_per_track"] < 2000.0
            assert e["steps_run"]["half_resolution"] >= 1 and e["steps_run"]["full_resolution"] >= 1
            assert e["pose_error_after_track"]["translation_m"] < 2e-4 and e["algorithmic_GBps"] > 0 and e["frac_of_8TBps"] < 1.0
    dense = d["cpu_baseline"]["configs0_dense_128"]
    gpu = dense["gpu"]
    assert gpu["voxels"] == dense["voxels"] == 128 ** 3 and dense["same_voxels_updated"] is True
    assert 0.001 < gpu["ms"] < 1.0 and 0.0 < gpu["past_l3"]["frac_of_8TBps"] < 1.0 and "frac_of_8TBps" not in gpu
    assert dense["gpu_over_cpu"]["1_core"] > 10
