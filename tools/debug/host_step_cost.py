#!/usr/bin/env python3
"""What one bench.FrameLoop.step costs the HOST (development aid): the loop of bench.py's headline workload, timed (a) as it is
and (b) with the time spent inside each C ABI call separated from the Python around it, by wrapping the library's functions."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench        # noqa: E402
import scenes       # noqa: E402
import torch        # noqa: E402
from vulcan_amd import api   # noqa: E402

torch.cuda.set_device(0)
api.lib()
n = 140          # (2 x 140 + 41 frames at 0.5 degrees: within what the app's Volume(65024, 8192) holds of this scene — its excess list is
                 # exhausted after ~35 k blocks, some 540 frames, and every frame from then on drops requests and runs all three rounds)
poses = [scenes.orbit_pose(i, bench.YAW_STEP) for i in range(2 * n + 41)]
loop = bench.FrameLoop("rgbd", poses)
for i in range(40):
    loop.step(i)
torch.cuda.synchronize()
for rep in range(2):      # (the first repetition runs while the device's clocks are still coming up)
    t0 = time.perf_counter()
    for i in range(n):
        loop.step(40 + rep * n + i)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"as it is, repetition {rep}: {1e6 * t_all / n:.1f} us per step; the host had issued everything after {1e6 * t_issue / n:.1f} us per step")

# the same with every library call timed
inside = {}


class Timed:
    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        f = getattr(self._lib, name)

        def call(*a):
            t = time.perf_counter()
            r = f(*a)
            inside[name] = inside.get(name, 0.0) + time.perf_counter() - t
            return r
        setattr(self, name, call)
        return call


loop2 = bench.FrameLoop("rgbd", poses[:n + 41])
loop2.lib = Timed(loop2.lib)
for i in range(40):
    loop2.step(i)
torch.cuda.synchronize()
inside.clear()
t0 = time.perf_counter()
for i in range(n):
    loop2.step(40 + i)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"wrapped: issued after {1e6 * t_issue / n:.1f} us per step, of which inside the library:")
for k, v in sorted(inside.items(), key=lambda kv: -kv[1]):
    print(f"   {k:40s} {1e6 * v / n:7.2f} us per step")
print(f"   {'python around the calls':40s} {1e6 * (t_issue - sum(inside.values())) / n:7.2f} us per step")
