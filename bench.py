#!/usr/bin/env python3
"""bench.py — throughput of the fusion + raycast hot path on MI355X.

Workload (BASELINE.json configs[1]): a 640x480 depth-only sequence fused into a
5 mm hashed volume, Volume(65024, 8192) (apps/vulcan/vulcan.cu:12-13). One
"step" = one frame = Volume::SetView -> DepthIntegrator::Integrate ->
Tracer::Trace through the C ABI (include/vk.h). The camera sits at the centre of
a 2 m sphere and yaws 0.5 deg per frame, so the depth image (resident in HBM) is
the same closed form every frame while new blocks are allocated every frame.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU, every rank owns a replica volume and its own frame
sequence (weak scaling, no data-path collective: SURVEY.md §8e); value = frames
of all ranks / max-over-ranks time.

Prints ONE JSON line on rank 0 with `roofline` (integrate kernel, HBM bound)
and `cpu_baseline` (the CPU oracle timed on a bounded sample of the same frames).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

W, H = 640, 480
VOXEL, TRUNC = 0.005, 0.04
MAIN, EXCESS = 65024, 8192
RADIUS, YAW_STEP = 2.0, 0.5
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_BLOCK = 4 + 16 + 2 * 512 * 20   # index + entry + voxel read + write


def sphere_room_depth(k):
    """Depth seen from the centre of a sphere of radius RADIUS: z = R / |unproject(u,v)|
    (identical for every camera rotation about the centre)."""
    y, x = np.mgrid[0:H, 0:W]
    rx = (x + 0.5 - k.cx) / k.fx
    ry = (y + 0.5 - k.cy) / k.fy
    return (RADIUS / np.sqrt(rx * rx + ry * ry + 1.0)).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--cpu-frames", type=int, default=6, help="frames of the CPU-oracle sample (0 = skip)")
    ap.add_argument("--workload", default="depth", choices=["depth", "rgbd", "rgbd-icp"],
                    help="depth = BASELINE configs[1] (the headline line); rgbd = configs[2] without tracking "
                         "(light integrator + tracer); rgbd-icp = configs[2] with the pyramid ICP tracker. "
                         "The extra workloads print the same JSON shape without roofline/cpu_baseline.")
    args = ap.parse_args()
    if args.workload != "depth":
        return extra_workload(args)

    import torch
    from vulcan_amd import api, dist as vd, vk_types as T
    import scenes

    rank, local_rank, world = vd.init()
    assert world == max(1, args.gpus), f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local_rank % torch.cuda.device_count())   # == local_rank on a full node
    lib = api.lib()

    k = T.Projection.make(*scenes.APP_INTRINSICS)
    depth_np = sphere_room_depth(k)
    total_frames = args.warmup + args.steps
    # every rank walks the same arc, offset so ranks do not share poses
    poses = [scenes.orbit_pose(i + rank * 7, YAW_STEP) for i in range(total_frames)]

    def fresh():
        vol = api.Volume(MAIN, EXCESS, voxel_length=VOXEL, truncation_length=TRUNC)
        frame = api.Frame(depth_np, k, poses[0])
        out = api.Frame(torch.zeros((H, W), dtype=torch.float32, device="cuda"), k, poses[0],
                        color=torch.zeros((H, W, 3), dtype=torch.float32, device="cuda"),
                        normals=torch.zeros((H, W, 3), dtype=torch.float32, device="cuda"))
        return vol, frame, out, api.DepthIntegrator(vol), api.Tracer(vol)

    vol, frame, out, integ, tracer = fresh()
    stream = api.stream()

    # The timed loop calls the C ABI directly with descriptors built once, the way a
    # C++ caller would (the api.* wrappers rebuild their ctypes structs per call).
    vdesc, fdesc, odesc = vol.desc(), frame.desc(), out.desc()
    pdesc = integ.params
    vref, fref, oref, pref = C.byref(vdesc), C.byref(fdesc), C.byref(odesc), C.byref(pdesc)
    d_ptr, c_ptr, n_ptr = (C.c_void_p(t.data_ptr()) for t in (out.depth, out.color, out.normals))
    # the record through which DepthIntegrator::Integrate hands the raycast bounds of
    # its view to Tracer::Trace (include/vk.h vk_view_bounds), as the class layer does
    vb = tracer.view_bounds
    bref = C.byref(vb)

    def step(i, ev=None):
        fdesc.depth_to_world = poses[i]
        odesc.depth_to_world = poses[i]
        vb.valid = 0                                                       # Volume::SetView: new visible list
        rc = lib.vk_volume_set_view(vref, fref, stream)                    # volume.cu:430-437
        if ev:
            lib.vk_event_record(ev[0], stream)
        rc |= lib.vk_integrate_ahead(vref, pref, fref, 0, None, None, bref, stream)   # depth_integrator.cu:89-115
        if ev:
            lib.vk_event_record(ev[1], stream)
        rc |= lib.vk_trace_ahead(vref, oref, bref, d_ptr, c_ptr, n_ptr, stream)       # tracer.cpp:41-47
        if rc:
            raise api.VkError(f"frame {i}: C ABI returned {rc}")

    def make_event():
        e = C.c_void_p()
        api.check(lib.vk_event_create(C.byref(e)), "vk_event_create")
        return e

    # HIP events (created without the system-scope fence, vk_event_create) around the
    # integrate launch of every EVENT_STRIDE-th timed frame: even so a pair of records
    # costs the stream ~1.7 us, so bracketing every frame would slow the loop being timed
    EVENT_STRIDE = 4
    sampled = list(range(0, args.steps, EVENT_STRIDE))
    events = {i: (make_event(), make_event()) for i in sampled}

    for i in range(args.warmup):
        step(i)
    vd.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i, events.get(i))
    torch.cuda.synchronize()
    vd.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = vd.max_over_ranks(elapsed, device="cuda")

    # integrate kernel time from the HIP events recorded inside the timed region
    kernel_ms = []
    for i in sampled:
        e0, e1 = events[i]
        ms = C.c_float()
        api.check(lib.vk_event_elapsed_ms(e0, e1, C.byref(ms)), "vk_event_elapsed_ms")
        kernel_ms.append(ms.value)
    ctr = vol.read_counters()

    # Allocation is deterministic, so an untimed replay of the same poses gives the
    # visible-block count the integrate launch of every timed frame saw.
    del vol, frame, out, integ, tracer
    vol, frame, out, integ, tracer = fresh()
    nvis = []
    for i in range(total_frames):
        frame.depth_to_world = poses[i]
        vol.set_view(frame)
        integ.integrate(frame)
        if i >= args.warmup:
            nvis.append(vol.visible_count)
    nvis = np.array(nvis, dtype=np.float64)
    alg_bytes = nvis * BYTES_PER_BLOCK + W * H * 4
    achieved = float(alg_bytes[sampled].sum() / (np.sum(kernel_ms) * 1e-3) / 1e9)   # the sampled launches

    # what this GPU sustains on (a) a plain float4 copy and (b) the integrate kernel's own
    # access pattern with the arithmetic removed ("peak: measured"), outside
    # the timed region; reported next to the 8 TB/s spec figure the fraction is taken of
    def probe(fn, bytes_moved, reps=10):
        e0, e1 = make_event(), make_event()
        fn()
        lib.vk_event_record(e0, stream)
        for _ in range(reps):
            fn()
        lib.vk_event_record(e1, stream)
        ms = C.c_float()
        api.check(lib.vk_event_elapsed_ms(e0, e1, C.byref(ms)), "vk_event_elapsed_ms")
        return bytes_moved * reps / (ms.value * 1e-3) / 1e9

    buf_a = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    buf_b = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    copy_gbs = probe(lambda: lib.vk_probe_stream_copy(C.c_void_p(buf_b.data_ptr()), C.c_void_p(buf_a.data_ptr()),
                                                      buf_a.numel(), stream), 2 * buf_a.numel())
    vdesc2 = vol.desc()
    rmw_gbs = probe(lambda: lib.vk_probe_block_rmw(C.byref(vdesc2), stream), float(nvis[-1]) * 2 * 10240)
    del buf_a, buf_b

    frames_all = vd.sum_over_ranks(args.steps, device="cuda")
    result = {
        "metric": "RGB-D frames/sec (integrate+raycast), 640x480 @ 5 mm voxels",
        "value": frames_all / elapsed,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE configs[1]: 640x480 depth-only sequence, SetView + DepthIntegrator + Tracer, "
                        "5 mm voxels, Volume(65024,8192), camera at the centre of a 2 m sphere yawing 0.5 deg/frame",
            "frames_per_rank": args.steps, "image": [W, H], "voxel_length": VOXEL,
            "truncation_length": TRUNC, "visible_blocks_mean": float(nvis.mean()),
            "allocated_blocks_end": int(MAIN + EXCESS - 1 - ctr[T.VK_CTR_VOXEL_PTR]),
            "dropped_requests": int(ctr[T.VK_CTR_DROPPED]), "parallelism": f"replica volume per GPU x{world}",
        },
        "roofline": {
            "kernel": "integrate_pipelined_kernel<depth> (vk_integrate_depth)",
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(),
            "algorithmic_bytes_per_launch": float(alg_bytes.mean()),
            "avg_launch_us": float(np.mean(kernel_ms) * 1e3),
            "launches_timed": len(sampled),
            "measured_copy_GBps": copy_gbs, "measured_block_rmw_GBps": rmw_gbs,
        },
    }

    if rank == 0 and world == 1 and args.cpu_frames > 0:   # the CPU leg is reported at N=1 only
        result["cpu_baseline"] = cpu_baseline(depth_np, k, poses, args.cpu_frames)

    if rank == 0:
        print(json.dumps(result), flush=True)
    vd.shutdown()


def extra_workload(args):
    """BASELINE configs[2]: 640x480 RGB-D, depth + light-colour integration (mask, shading)
    and raycast, optionally preceded by PyramidTracker<DepthTracker> against the previous
    raycast (tracker -> SetView -> Integrate -> Trace, apps/vulcan/vulcan.cu:300-325)."""
    import torch
    from vulcan_amd import api, dist as vd, vk_types as T
    import scenes

    rank, local_rank, world = vd.init()
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    k = T.Projection.make(*scenes.APP_INTRINSICS)
    depth_np = sphere_room_depth(k)
    color_np = scenes.checker_color(W, H, 0.1, 0.9)
    total = args.warmup + args.steps
    poses = [scenes.orbit_pose(i + rank * 7, YAW_STEP) for i in range(total)]
    vol = api.Volume(MAIN, EXCESS, voxel_length=VOXEL, truncation_length=TRUNC)
    frame = api.Frame(depth_np, k, poses[0], color=color_np)
    frame.compute_normals()
    key = api.Frame(torch.zeros((H, W), dtype=torch.float32, device="cuda"), k, poses[0],
                    color=torch.zeros((H, W, 3), dtype=torch.float32, device="cuda"),
                    normals=torch.zeros((H, W, 3), dtype=torch.float32, device="cuda"))
    integ = api.LightIntegrator(vol)
    integ.light = T.Light.make(2.0, (0.025, 0.08, 0.0))     # apps/vulcan/vulcan.cu:87-88
    tracer = api.Tracer(vol)
    tracker = api.PyramidTracker()
    track = args.workload == "rgbd-icp"
    if track and world > 1:
        tracker.tracker.reduce_hook = vd.allreduce_system   # rigid rig: one system for all cameras

    # SetView / Integrate / Trace go straight to the C ABI with descriptors built once
    # (the api.* wrappers rebuild their ctypes structs on every call: ~35 us per frame)
    lib, stream = api.lib(), api.stream()
    vdesc, fdesc, kdesc = vol.desc(), frame.desc(), key.desc()
    vref, fref, kref, pref, lref = (C.byref(x) for x in (vdesc, fdesc, kdesc, integ.params, integ.light))
    vb = tracer.view_bounds
    bref = C.byref(vb)
    mask = integ.compute_frame_mask(frame)
    m_ptr = C.c_void_p(mask.data_ptr())
    d_ptr, c_ptr, n_ptr = (C.c_void_p(t.data_ptr()) for t in (key.depth, key.color, key.normals))

    def step(i):
        if track and i > 0:
            frame.depth_to_world = poses[i - 1]             # previous pose as the initial guess
            tracker.keyframe = key
            tracker.track(frame)                            # one 128-byte pose readback per level
        frame.depth_to_world = poses[i]                     # ground truth keeps the map consistent
        key.depth_to_world = poses[i]
        fdesc.depth_to_world = poses[i]
        kdesc.depth_to_world = poses[i]
        vb.valid = 0
        rc = lib.vk_volume_set_view(vref, fref, stream)
        rc |= lib.vk_light_compute_frame_mask(fref, integ.depth_threshold, m_ptr, stream)   # light_integrator.cu:270-275
        rc |= lib.vk_integrate_ahead(vref, pref, fref, 2, lref, m_ptr, bref, stream)
        rc |= lib.vk_trace_ahead(vref, kref, bref, d_ptr, c_ptr, n_ptr, stream)
        if rc:
            raise api.VkError(f"frame {i}: C ABI returned {rc}")

    for i in range(args.warmup):
        step(i)
    vd.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    vd.barrier()
    elapsed = vd.max_over_ranks(time.perf_counter() - t0, device="cuda")
    frames_all = vd.sum_over_ranks(args.steps, device="cuda")
    if rank == 0:
        print(json.dumps({
            "metric": "RGB-D frames/sec (integrate+raycast), 640x480 @ 5 mm voxels", "value": frames_all / elapsed,
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: 640x480 RGB-D, LightIntegrator (mask + depth + shaded colour) + "
                                   "Tracer" + (" + PyramidTracker<DepthTracker> (15 + 20 Gauss-Newton iterations)" if track else ""),
                       "visible_blocks": vol.visible_count, "parallelism": f"replica volume per GPU x{world}"},
        }), flush=True)
    vd.shutdown()


def pmc_traffic():
    """HBM-side bytes per integrate launch from the committed rocprofv3 PMC passes
    (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate passes): counters cannot
    be collected from inside an unprofiled run, so the newest measurement on file is
    reported; None when there is none."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_integrate_traffic.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        return json.load(f)["bytes_per_launch"]


def cpu_baseline(depth_np, k, poses, frames):
    """The CPU oracle (oracle/, the restated reference kernels) on the first `frames`
    poses of the same sequence; OpenMP over blocks / pixels where the reference's
    threads are independent, allocation serial."""
    from oracle import oracle as orc
    # a 1-GPU box owns a 16-core share of the host (not all 256 hardware threads)
    cores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    orc.set_threads(cores)
    hv = orc.HostVolume(MAIN, EXCESS, voxel_length=VOXEL, truncation_length=TRUNC)
    hf = orc.HostFrame(depth_np, k, poses[0])
    for i in range(2):                       # untimed: the cold first frames allocate ~7k blocks
        hf.depth_to_world = poses[i]
        hv.set_view(hf, orc.POLICY_SERIAL)
        orc.integrate_depth(hv, hf)
    t0 = time.perf_counter()
    for i in range(2, 2 + frames):
        hf.depth_to_world = poses[i]
        hv.set_view(hf, orc.POLICY_SERIAL)
        orc.integrate_depth(hv, hf)
        orc.trace(hv, hf)
    dt = time.perf_counter() - t0
    orc.set_threads(1)
    return {"value": frames / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"frames 2..{1 + frames} of the same sequence (SetView + integrate + raycast + normals each), "
                      f"allocation serial, integrate/raycast/normals OpenMP x{cores}"}


if __name__ == "__main__":
    main()
