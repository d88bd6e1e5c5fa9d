"""One rank of tests/test_gpu_rig_two_ranks.py: started as a fresh process (RANK / WORLD_SIZE / MASTER_* in the
environment), joins a gloo group BEFORE anything touches the GPU, then runs the rig's Track twice on the one GPU the
ranks share: with the ranks' normal systems added through a reduce hook (a gloo all-reduce: launch-per-stage loop), and
with the in-launch exchange through peer-mapped areas (vk_comm_exchange_create / _attach_handles: RCCL refuses two ranks
on one device, the handles travel over gloo). Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import numpy as np
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)      # CPU only: no GPU touched yet

    import scenes
    from vulcan_amd import api, comm, vk_types as T
    torch.cuda.set_device(0)
    api.lib()
    out = {"rank": rank, "world": world}

    w, h = 320, 240
    k = T.Projection.make(*(np.float32(0.5) * np.float32(v) for v in scenes.APP_INTRINSICS))
    y, x = np.mgrid[0:h, 0:w]
    depth = (1.5 + 0.08 * np.cos(3.0 * x / w + rank) * np.sin(2.0 * y / h + 0.5 * rank)).astype(np.float32)
    rig_pose = scenes.yaw(360.0 * rank / world)                               # camera `rank` of the ring
    key = api.Frame(depth, k, rig_pose)
    key.compute_normals()
    errors = [T.Transform.translate(0.003, -0.002, 0.004) * T.Transform.rotate(0.999995, 0.002, -0.0015, 0.001),
              T.Transform.translate(-0.002, 0.001, 0.002), T.Transform.translate(0.001, 0.003, -0.002)]

    def gloo_sum(system):
        t = system.cpu()
        dist.all_reduce(t)
        system.copy_(t)

    def all_gather(blob):
        got = [None] * world
        dist.all_gather_object(got, blob)
        return got

    agree = comm.agree_over_torch_group()

    # the loops of both ranks must be resident together: each takes at most a part of the device
    with api.test_hooks(loop_grid_cap=int(os.environ.get("VK_TEST_RIG_GRID_CAP", "60"))):
        hooked = api.DepthTracker()
        hooked.keyframe = key
        hooked.reduce_hook = gloo_sum
        want = []
        for e in errors:
            frame = api.Frame(depth, k, e * rig_pose, normals=key.normals)
            want.append(bytes(hooked.track(frame)))
            torch.cuda.synchronize()
        out["hook_steps"] = int(hooked.state.cpu()[0])

        c = comm.Communicator.without_rccl(rank, world)
        x = c.attach_exchange_with(all_gather, agree)
        out["areas_mapped"] = int(sum(1 for r in range(world) if x.areas[r]))
        rig = api.DepthTracker()
        rig.keyframe = key
        got, steps, updates_equal = [], [], []
        try:
            for e in errors:
                frame = api.Frame(depth, k, e * rig_pose, normals=key.normals)
                got.append(bytes(c.track_rig(rig, frame)))
                torch.cuda.synchronize()
                steps.append(int(rig.state.cpu()[0]))
                upd = rig.update.cpu()
                everyone = [None] * world
                dist.all_gather_object(everyone, upd.numpy().tobytes())
                updates_equal.append(all(b == everyone[0] for b in everyone))
            out["aborted"] = False
        except api.TrackAborted as e:
            out["aborted"] = True
            out["error"] = str(e)
        out["sequence_after"] = int(c.exchange.sequence)
        out["rig_steps"] = steps
        out["poses_equal_the_hook_path"] = got == want[:len(got)] and len(got) == len(errors)
        out["update_identical_on_all_ranks"] = bool(updates_equal) and all(updates_equal)
        # the pose came back to the rig's: the ranks really solved ONE system
        if got:
            last = T.Transform.from_buffer_copy(got[-1])
            out["pose_error_after_track"] = float(np.abs(last.matrix() @ rig_pose.inverse_matrix() - np.eye(4)).max())
        dist.barrier()
        c.close()
    print(json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
