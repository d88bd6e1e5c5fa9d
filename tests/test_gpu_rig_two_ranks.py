"""The rig's in-launch exchange (vk_icp_track_rig, vulcan_amd/csrc/vk_rig_protocol.h) ACROSS TWO PROCESSES: each rank's
area in device memory, mapped into the other process through an IPC handle, the 27 {tag, value} words of every
Gauss-Newton step written into the peer's area and read from the own one — everything but the xGMI link, which this
pool's single-GPU boxes do not have: both ranks use the one GPU (RCCL refuses that, so the handles travel over gloo:
vk_comm_exchange_create / vk_comm_exchange_attach_handles), each loop kernel capped to a part of the device so that
both are resident together (vk_test_hooks.loop_grid_cap). Three Tracks; the poses must equal — bit for bit — those of
the reduce-hook path (the systems added by a gloo all-reduce between launches), and the last update must be the same
on both ranks. If the two loops cannot be resident together the kernels give up after two seconds (VK_TRACK_ABORTED)
and the test says so instead of hanging."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_processes_exchange_through_mapped_areas():
    world, port = 2, free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rig_two_ranks_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    results = []
    for p in procs:
        try:
            out, err = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()                       # exact children, by handle
            raise
        assert p.returncode == 0, err[-3000:]
        results.append(json.loads([l for l in out.splitlines() if l.startswith("{")][-1]))
    print(results)
    for r in results:
        assert r["areas_mapped"] == world
        assert not r["aborted"], r
        assert r["poses_equal_the_hook_path"], r
        assert r["update_identical_on_all_ranks"], r
        assert r["sequence_after"] == 4                       # three Tracks from 1
        assert r["pose_error_after_track"] < 5e-4
        assert all(2 <= s <= 20 for s in r["rig_steps"])
    assert results[0]["rig_steps"] == results[1]["rig_steps"]  # the same solve: the same number of steps
