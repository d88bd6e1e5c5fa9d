#!/usr/bin/env python3
"""Writes tests/golden/scenes_160x120.npz: reduced-size golden vectors for the synthetic
scenes of SURVEY.md §8(d), produced by the CPU oracle (oracle/, the restated reference
kernels — the reference itself cannot be built or run here: CUDA + OpenCV).

  python tests/golden/make_fixtures.py            # regenerate (needs liboracle.so)
  python tests/golden/make_fixtures.py --step-cap-only   # only tests/golden/step_cap_64x48.npz (round 5)

Inputs are closed-form (tests/scenes.py) and are not stored. Per scene the file holds the
raycast depth / colour / normal images and bounds grid, the visible count and pool pointers,
and SHA-256 digests of the volume's buffers (the voxel pool is 50 MB). tests/test_golden.py
checks the oracle against the file on the CPU; tests/test_gpu_golden.py checks the device
against it without building or loading the oracle."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

W, H = 160, 120
MAIN, EXCESS, VOXEL, TRUNC = 4096, 1024, 0.008, 0.04
FILE = os.path.join(HERE, "scenes_160x120.npz")


def scene_inputs(name):
    """(depth, colour, projection, pose) of scene `name` — closed form, shared with the tests."""
    import scenes
    from vulcan_amd import vk_types as T
    k = T.Projection.make(136.0, 136.0, 80.0, 60.0)
    if name == "plane":            # scene A
        return scenes.plane(W, H, 1.5), scenes.constant_color(W, H), k, T.Transform.identity()
    if name == "sphere":           # scene B, scaled to the reduced image: r < 50 px
        y, x = np.mgrid[0:H, 0:W]
        u = x.astype(np.float32) + np.float32(0.5) - np.float32(0.5 * W)
        v = y.astype(np.float32) + np.float32(0.5) - np.float32(0.5 * H)
        rr = (u * u + v * v).astype(np.float64)
        depth = np.where(np.sqrt(rr) < 50, 4.0 - 1.5 * np.sqrt(np.maximum(50.0 * 50.0 - rr, 0.0)) / 50.0, 0.0)
        return depth.astype(np.float32), scenes.checker_color(W, H, 0.1, 0.9), k, scenes.tracer_test_pose()
    if name == "ripple":           # scene C
        return (scenes.ripple(W, H) * np.float32(1.3)).astype(np.float32), scenes.checker_color(W, H, 0.1, 0.9), k, \
            T.Transform.translate(0.001, -0.002, 0.003) * T.Transform.rotate(0.9998719, 0.0085884, -0.0104268, 0.0085884)
    if name == "ramp":             # scene D
        return scenes.ramp(W, H), scenes.constant_color(W, H, (0.4, 0.5, 0.6)), k, T.Transform.translate(-10.73, 2.11, -33.54)
    raise KeyError(name)


SCENES = ("plane", "sphere", "ripple", "ramp")


def digest(array):
    return hashlib.sha256(np.ascontiguousarray(array).tobytes()).hexdigest()


def run_scene(backend, name):
    """backend: an object with the oracle's or the device's operations (tests build one each)."""
    depth, color, k, pose = scene_inputs(name)
    return backend(depth, color, k, pose)


def oracle_backend(depth, color, k, pose):
    from oracle import oracle as orc
    from vulcan_amd import vk_types as T
    hv = orc.HostVolume(MAIN, EXCESS, voxel_length=VOXEL, truncation_length=TRUNC)
    hf = orc.HostFrame(depth, k, pose, color=color)
    hf.compute_normals()
    for _ in range(5):
        hv.set_view(hf, orc.POLICY_MAXKEY)
    orc.integrate_depth(hv, hf)
    orc.integrate_color(hv, hf)
    odepth, ocolor, onormals, obounds = orc.trace(hv, hf)
    points, faces, skipped = orc.extract_mesh(hv, True, True)
    key = orc.HostFrame(odepth, k, pose, normals=onormals)
    moved = orc.HostFrame(depth, k, T.Transform.translate(0.002, -0.001, 0.001) * pose, normals=hf.normals)
    residuals = orc.icp_residuals(key, moved)
    return dict(frame_normals=hf.normals, depth=odepth, color=ocolor, normals=onormals, bounds=obounds,
                counters=hv.counters[:8].copy(), visible=np.sort(hv.visible()),   # the counters the fixture was made with (vk.h grew internal ones since)
                voxels_sha256=digest(hv.voxels), entries_sha256=digest(hv.hash_entries),
                visibility_sha256=digest(hv.block_visibility),
                mesh_points_sha256=digest(points), mesh_faces_sha256=digest(faces),
                mesh_counts=np.array([len(points), len(faces), skipped], dtype=np.int32),
                icp_residuals=residuals)


# ---- the march's 500-step cap (tracer.cu:437-442; tests/scenes.py STEP_CAP_*, tests/step_cap.py): a file of its own
STEP_CAP_FILE = os.path.join(HERE, "step_cap_64x48.npz")


def step_cap_oracle():
    from oracle import oracle as orc
    import step_cap
    hv, hf = step_cap.build_host(orc)
    depth, color, normals, bounds = orc.trace(hv, hf)
    return dict(depth=depth, color=color, normals=normals, bounds=bounds, visible=np.sort(hv.visible()),
                entries_sha256=digest(hv.hash_entries), voxels_sha256=digest(hv.voxels))


# ---- the soak run's parity leg (round 6; tools/soak.py): 2 000 frames of fusion + raycast at GIVEN poses on the room
# sequence, looped (240 distinct frames), at the bench's size — Volume(65024, 8192), 640x480, 5 mm. The file holds SHA-256
# digests of the hash table, the voxel pool and the raycast images at a few frame counts: long enough for the free list,
# the visibility churn of revisited blocks and the saturated weights to matter, and still a parity statement.
SOAK_FILE = os.path.join(HERE, "soak_room_640x480.json")
SOAK_CHECKPOINTS = (20, 240, 480, 2000)
SOAK_CYCLE = 240          # tests/scenes.py room_pose: frames per swing


def soak_inputs(count=SOAK_CYCLE):
    """[(depth, colour, pose)] of the first `count` frames of the room sequence at the bench's size"""
    import scenes
    from concurrent.futures import ThreadPoolExecutor
    from vulcan_amd import vk_types as T
    k = T.Projection.make(*scenes.APP_INTRINSICS)
    poses = [scenes.room_pose(i) for i in range(count)]
    with ThreadPoolExecutor(max_workers=8) as pool:
        images = list(pool.map(lambda p: scenes.room_frame(k, p, 640, 480, light=(2.0, (0.025, 0.08, 0.0))), poses))
    return k, [(d, c, p) for (d, c), p in zip(images, poses)]


def soak_oracle(frames, checkpoints=SOAK_CHECKPOINTS, progress=None):
    """the oracle's loop: Frame::ComputeNormals, SetView x3, depth + frame mask + shaded colour, Trace (vulcan.cu:297,316-325)"""
    from oracle import oracle as orc
    from vulcan_amd import vk_types as T
    k, inputs = soak_inputs(min(frames, SOAK_CYCLE))
    light = T.Light.make(2.0, (0.025, 0.08, 0.0))
    hv = orc.HostVolume(65024, 8192, voxel_length=0.005, truncation_length=0.04)
    out = {}
    for i in range(frames):
        depth, color, pose = inputs[i % SOAK_CYCLE]
        hf = orc.HostFrame(depth, k, pose, color=color)
        hf.compute_normals()
        for _ in range(3):
            hv.set_view(hf, orc.POLICY_MAXKEY)
        orc.integrate_depth(hv, hf)
        orc.integrate_light_color(hv, hf, light, orc.light_frame_mask(hf, 0.2))
        if i + 1 in checkpoints:
            odepth, ocolor, onormals, _ = orc.trace(hv, hf)
            out[str(i + 1)] = {"entries_sha256": digest(hv.hash_entries), "voxels_sha256": digest(hv.voxels),
                               "depth_sha256": digest(odepth), "color_sha256": digest(ocolor), "normals_sha256": digest(onormals),
                               "visible": int(hv.visible_count), "voxel_pointer": int(hv.counters[T.VK_CTR_VOXEL_PTR]),
                               "dropped": int(hv.counters[T.VK_CTR_DROPPED])}
            if progress:
                progress(i + 1, out[str(i + 1)])
    return out


def main():
    from oracle import oracle as orc
    orc.build()
    orc.set_threads(8)
    if "--soak" in sys.argv:
        import json
        import time
        t0 = time.time()
        out = soak_oracle(max(SOAK_CHECKPOINTS), progress=lambda n, d: print(f"frame {n}: {time.time() - t0:.0f} s {d}", flush=True))
        with open(SOAK_FILE, "w") as f:
            json.dump({"what": "tests/golden/make_fixtures.py --soak: the CPU oracle over the looped room sequence, digests at "
                               "the named frame counts (the raycast is made only there: it changes nothing)",
                       "checkpoints": out}, f, indent=1)
        print(f"{SOAK_FILE}: written")
        return
    if "--step-cap-only" not in sys.argv:
        out = {}
        for name in SCENES:
            for key, value in run_scene(oracle_backend, name).items():
                out[f"{name}/{key}"] = np.asarray(value)
        np.savez_compressed(FILE, **out)
        print(f"{FILE}: {os.path.getsize(FILE) / 1e6:.2f} MB, {len(out)} arrays")
    out = {f"step_cap/{key}": np.asarray(value) for key, value in step_cap_oracle().items()}
    np.savez_compressed(STEP_CAP_FILE, **out)
    print(f"{STEP_CAP_FILE}: {os.path.getsize(STEP_CAP_FILE) / 1e3:.1f} kB, {len(out)} arrays")


if __name__ == "__main__":
    main()
