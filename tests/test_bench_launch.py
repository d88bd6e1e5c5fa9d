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
