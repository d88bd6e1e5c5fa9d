"""Multi-GPU plumbing: one process per GPU, torch.distributed over RCCL (backend
"nccl" on ROCm) — or gloo on CPU for the tests.

The fusion + raycast path shards by frame / camera: every rank owns a replica
volume and its own frames, and no voxel data ever crosses xGMI. The only
exchange step the path has is the ICP normal system of a rigid multi-camera rig
(SURVEY.md §8e): 27 useful floats, summed over ranks once per Gauss-Newton
iteration, after which every rank solves the same 6x6 system. At 192 bytes the
collective is latency bound, so it is issued as ONE all-reduce of the packed
48-float buffer on the compute stream, straight from device memory.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), \
        int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None):
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.
    Returns (rank, local_rank, world_size); a single process needs no group."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # VK_DIST_BACKEND=gloo rehearses a multi-rank run on a box with fewer GPUs
            # than ranks (ranks then share devices; RCCL cannot do that)
            backend = os.environ.get("VK_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()


def shard(items, rank, world):
    """Frames / views owned by `rank`: item i goes to rank i % world."""
    return [x for i, x in enumerate(items) if i % world == rank]


def allreduce_system(system):
    """Sum the packed ICP system (api.DepthTracker.system, 48 floats) over ranks, in
    place. No-op for a single process."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(system, op=dist.ReduceOp.SUM)
    return system


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    """Max of a python float over ranks (bench timing contract)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def all_ok(ok, device="cpu"):
    """True when EVERY rank passed True: what the ranks call after a step that can fail on one of them alone (a
    rank-local check, an allocation, a mapping) and before the next collective, so that all of them take the same
    branch — a rank that raised on its own would leave the others inside that collective for ever."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return bool(ok)
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def gather_over_ranks(value, device="cpu"):
    """The python float of every rank, in rank order (bench.py: per-rank ms_per_step next to the max)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return [float(value)]
    mine = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [float(t.item()) for t in out]


class StepFailed(RuntimeError):
    """Raised by agreed_step on EVERY rank when the step failed on ANY rank."""


def agreed_step(what, step, device="cpu"):
    """Runs a rank-local step that may fail on one rank alone (creating a communicator, mapping a peer's memory, a
    check of what RCCL reports) and makes its OUTCOME collective: every rank learns whether all ranks passed before any of
    them enters the next collective. Returns the step's value; raises StepFailed — on every rank, with this rank's own
    error text or "another rank failed" — if any rank failed. A rank that raised on its own would leave the others
    inside the next collective until a timeout."""
    value, error = None, None
    try:
        value = step()
    except Exception as e:     # noqa: BLE001  (the text travels in the JSON line)
        error = f"{type(e).__name__}: {e}"[:300]
    if not all_ok(error is None, device=device):
        raise StepFailed(f"{what}: {error}" if error else f"another rank failed in: {what}")
    return value
