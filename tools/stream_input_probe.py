#!/usr/bin/env python3
"""Where the streamed-input frame's time goes (development aid for bench.py --stream-input): the rgbd loop with parts of
the upload protocol removed one at a time — unsafe forms included, for timing only."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import bench
    import scenes
    frames = 220
    poses = [scenes.orbit_pose(i, bench.YAW_STEP) for i in range(frames)]

    def run(name, patch=None, stream_input=True, workload="rgbd"):
        loop = bench.FrameLoop(workload, poses, stream_input=stream_input)
        if patch:
            patch(loop)
        for i in range(20):
            loop.step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(20, frames):
            loop.step(i)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / (frames - 20) * 1e6
        print(f"{name:64s} {us:7.1f} us/frame  {1e6 / us:7.0f} frames/s", flush=True)
        del loop
        torch.cuda.empty_cache()

    run("resident input", stream_input=False)
    run("streamed (the product)")

    def no_copy(loop):
        up = loop.upload
        lib = loop.lib

        def submit():
            slot = up.slots[up.count % up.SLOTS]
            if slot["consumed_recorded"]:
                lib.vk_event_synchronize(slot["events"][1])
            lib.vk_event_record(slot["events"][0], up.copy_stream)
            up.count += 1
        up.submit = submit
    run("streamed, events and waits but NO copies", no_copy)

    def no_waits(loop):
        up = loop.upload
        up.acquire = lambda n: [dev for _, dev, _ in up.slots[n % up.SLOTS]["images"]]
        up.release = lambda n: None
    run("streamed, copies but NO waits / records on the compute stream", no_waits)

    def no_consumed(loop):
        up = loop.upload
        up.release = lambda n: None
    run("streamed, no `consumed` record (nobody waits for the readers)", no_consumed)

    def device_wait(loop):
        up, lib = loop.upload, loop.lib

        def submit():
            slot = up.slots[up.count % up.SLOTS]
            if slot["consumed_recorded"]:
                lib.vk_stream_wait_event(up.copy_stream, slot["events"][1])
            for pinned, dev, nbytes in slot["images"]:
                lib.vk_memcpy_h2d_async(bench.C.c_void_p(dev.data_ptr()), pinned, nbytes, up.copy_stream)
            lib.vk_event_record(slot["events"][0], up.copy_stream)
            up.count += 1
        up.submit = submit
    run("streamed, the COPY STREAM waits for the readers on the device", device_wait)
    run("depth only: resident", stream_input=False, workload="depth")
    run("depth only: streamed", workload="depth")


if __name__ == "__main__":
    main()
