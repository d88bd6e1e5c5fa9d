"""The N>1 path on CPU: two gloo ranks shard the frame sequence, each fuses its
share into a replica volume (oracle stands in for the device kernels here — this
test covers the sharding / collective plumbing, not kernels), the ICP system is
all-reduced, and timing / throughput aggregation follows the bench contract."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from vulcan_amd import dist as vd, vk_types as T
    from oracle import oracle as orc
    import scenes

    r, lr, w = vd.init(backend="gloo")
    assert (r, w) == (rank, world)
    frames = list(range(8))
    mine = vd.shard(frames, rank, world)
    assert mine == [i for i in frames if i % world == rank]

    # per-rank ICP system: keyframe plane, frame = ripple at a rank-specific pose
    k = T.Projection.make(136.75, 136.75, 80, 60)
    key = orc.HostFrame(scenes.plane(160, 120, 1.0), k, T.Transform.identity())
    key.compute_normals()
    frm = orc.HostFrame(scenes.ripple(160, 120), k, T.Transform.translate(0.001 * (rank + 1), -0.002, 0.003))
    frm.compute_normals()
    H, g = orc.icp_system(key, frm, True)
    system = torch.zeros(48, dtype=torch.float32)
    system[:21] = torch.from_numpy(H.astype(np.float32))
    system[36:42] = torch.from_numpy(g.astype(np.float32))
    local = system.clone()
    vd.allreduce_system(system)

    t = vd.max_over_ranks(1.0 + rank)
    n = vd.sum_over_ranks(len(mine))
    vd.barrier()
    out[rank] = (local.numpy(), system.numpy(), t, n)
    vd.shutdown()


def test_two_rank_gloo_sharding_and_icp_allreduce():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    l0, s0, t0, n0 = out[0]
    l1, s1, t1, n1 = out[1]
    np.testing.assert_allclose(s0, l0 + l1, rtol=1e-6)
    assert np.array_equal(s0, s1)                       # every rank solves the same system
    assert not np.array_equal(l0, l1)
    assert t0 == t1 == 2.0 and n0 == n1 == 8.0          # max-over-ranks time, whole-job frame count


def test_single_process_helpers_are_noops():
    sys.path.insert(0, ROOT)
    from vulcan_amd import dist as vd
    x = torch.arange(48, dtype=torch.float32)
    assert torch.equal(vd.allreduce_system(x.clone()), x)
    assert vd.max_over_ranks(3.5) == 3.5 and vd.sum_over_ranks(2) == 2.0
    assert vd.shard(list(range(5)), 0, 1) == list(range(5))
