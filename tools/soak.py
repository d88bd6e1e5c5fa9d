#!/usr/bin/env python3
"""tools/soak.py — the long run nobody had made: every parity test is <= 6 frames from an empty volume or
<= 300 frames of one scene; nothing had run long enough for counters, tags, the free list or the saturated weights to do
anything unusual.

  leg 1 (parity): 2 000 frames of fusion + raycast at GIVEN poses on the room sequence, looped (240 distinct frames), through
         bench.FrameLoop("rgbd") — the bench's own step, request pass riding behind the raycast — against the CPU oracle's
         digests in tests/golden/soak_room_640x480.json (tests/golden/make_fixtures.py --soak) at frames 20 / 240 / 480 / 2 000:
         hash table, every voxel byte, raycast depth / colour / normals, visible count, free-slot pointer. Bit for bit.
  leg 2 (soak):   --frames (default 200 000) of `rgbd-icp` — each frame tracked by PyramidTracker<DepthTracker> against the
         previous raycast, fused and raycast at the TRACKED pose — on the same loop of 240 frames. Every --every frames: pose
         error of the window against the ground truth the loop never saw, VK_CTR_DROPPED, the free-slot pointer, the visible
         count, a digest of the hash table, Gauss-Newton steps, aborted Tracks. At the end the raycast from the first pose
         against the raycast taken from the same pose at frame 300: the map must not have degraded.

Writes one JSON document (profiles/r06_soak.json is a run of this on one MI355X through gpurun). ref: src/volume.cu:304-368
(the allocator whose state it exercises), apps/vulcan/vulcan.cu:297-325 (the loop)."""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


class Looped:
    """a list that repeats: frame i of the soak is frame i mod 240 of the room sequence"""

    def __init__(self, items, length):
        self.items, self.length = items, length

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        return self.items[i % len(self.items)]


def sha(t):
    return hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=200000)
    ap.add_argument("--every", type=int, default=10000)
    ap.add_argument("--parity-frames", type=int, default=2000)
    ap.add_argument("--excess", type=int, default=0, help="excess block count of the soak leg's volume (0: the app's 8192)")
    ap.add_argument("--max-weight", type=int, default=0, help="integrator weight caps of the soak leg (0: the default 16)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "soak.json"))
    args = ap.parse_args()

    import torch
    import bench
    import make_fixtures as mf
    import scenes
    from vulcan_amd import api, vk_types as T
    torch.cuda.set_device(0)
    api.lib()
    torch.cuda.set_stream(torch.cuda.Stream())
    t_start = time.time()
    doc = {"tool": "tools/soak.py", "device": torch.cuda.get_device_name(0), "cycle_frames": mf.SOAK_CYCLE}

    k, inputs = mf.soak_inputs(mf.SOAK_CYCLE)
    room = bench.RoomSequence.__new__(bench.RoomSequence)
    room.truth = [p for _, _, p in inputs]
    room.depth = [torch.from_numpy(d).cuda() for d, _, _ in inputs]
    room.color = [torch.from_numpy(c).cuda() for _, c, _ in inputs]
    torch.cuda.synchronize()

    def looped(n):
        seq = bench.RoomSequence.__new__(bench.RoomSequence)
        seq.truth, seq.depth, seq.color = Looped(room.truth, n), Looped(room.depth, n), Looped(room.color, n)
        return seq

    # ---------------------------------------------------------------- leg 1: parity ----
    golden = json.load(open(mf.SOAK_FILE))["checkpoints"]
    n = args.parity_frames
    if n <= 0:
        golden = {}
    seq = looped(n + 1)
    loop = bench.FrameLoop("rgbd", seq.truth, sequence=seq)
    assert loop.ahead is not None, "the bench's own step: the request pass rides behind the raycast"
    vol = loop.vols[0]["vol"]
    parity = {"frames": n, "workload": "bench.FrameLoop('rgbd') on the looped room sequence, poses given", "checkpoints": {}}
    ok = True
    t0 = time.time()
    for i in range(n):
        loop.step(i)
        if str(i + 1) in golden:
            torch.cuda.synchronize()
            ctr = vol.read_counters()
            got = {"entries_sha256": sha(vol.hash_entries), "voxels_sha256": sha(vol.voxels), "depth_sha256": sha(loop.key.depth),
                   "color_sha256": sha(loop.key.color), "normals_sha256": sha(loop.key.normals),
                   "visible": int(ctr[T.VK_CTR_VISIBLE]), "voxel_pointer": int(ctr[T.VK_CTR_VOXEL_PTR]),
                   "dropped": int(ctr[T.VK_CTR_DROPPED])}
            want = golden[str(i + 1)]
            differs = sorted(key for key in want if got[key] != want[key])
            parity["checkpoints"][str(i + 1)] = {"equal_to_the_oracle": not differs, "differs": differs, "visible": got["visible"],
                                                 "voxel_pointer": got["voxel_pointer"], "dropped": got["dropped"]}
            ok = ok and not differs
            print(f"parity leg, frame {i + 1}: {'equal to the oracle' if not differs else 'DIFFERS: ' + str(differs)}", flush=True)
    torch.cuda.synchronize()
    parity["seconds"] = time.time() - t0
    parity["ok"] = ok and (len(parity["checkpoints"]) > 0 or n <= 0)
    doc["parity_leg"] = parity
    del loop, vol
    torch.cuda.empty_cache()

    # ------------------------------------------------------------------ leg 2: soak ----
    n = args.frames
    seq = looped(n + 1)
    if args.excess:
        bench.EXCESS = args.excess
    loop = bench.FrameLoop("rgbd-icp", seq.truth, sequence=seq)
    vol = loop.vols[0]["vol"]
    if args.max_weight:
        loop.vols[0]["integ"].params.max_distance_weight = loop.vols[0]["integ"].params.max_color_weight = args.max_weight
    doc["soak_volume"] = {"main_blocks": vol.main, "excess_blocks": vol.excess, "max_weight": args.max_weight or 16}
    # the 22-bit launch tag of the Gauss-Newton exchange repeats every 2^22 - 1 loop launches (vk_runtime.hip): start the leg
    # 100 000 launches in front of a repeat so that the run crosses it (a fifth of the way through the default run)
    import ctypes as C
    period = (1 << 22) - 1
    now = api_loop_count(api)[0]
    api.check(api.lib().vk_test_hooks_loop_count(C.byref(C.c_uint64((now // period + 2) * period - 100000)), None, None),
              "vk_test_hooks_loop_count")
    count_before = api_loop_count(api)
    reports, first_view = [], None
    t0 = time.time()
    window_from = 0
    # the first 2 000 frames, every 100: what the CPU oracle's OWN tracked loop (tests/golden/make_soak_oracle_drift.py,
    # tests/golden/soak_oracle_drift.json) is compared with — is a slow creep of the pose the algorithm's or the device path's?
    fine, fine_from = [], 0
    for i in range(n):
        loop.step(i)
        if i < 2000 and (i + 1) % 100 == 0:
            errs = [bench.pose_error(loop.tracked_poses[j], seq.truth[j]) for j in range(fine_from, i + 1)]
            fine.append({"frame": i + 1, "translation_m": max(e[0] for e in errs), "rotation_deg": max(e[1] for e in errs)})
            fine_from = i + 1
        if i + 1 == 300:
            first_view = raycast_from(api, loop, room)
        if (i + 1) % args.every == 0 or i + 1 == n:
            torch.cuda.synchronize()
            ctr = vol.read_counters()
            errs = [bench.pose_error(loop.tracked_poses[j], seq.truth[j]) for j in range(window_from, i + 1)]
            steps = np.array(loop.gn_steps[max(0, window_from - 1):i], dtype=np.int64)
            reports.append({"frame": i + 1, "seconds": time.time() - t0,
                            "pose_error_max": {"translation_m": max(e[0] for e in errs), "rotation_deg": max(e[1] for e in errs)},
                            "pose_error_last": {"translation_m": errs[-1][0], "rotation_deg": errs[-1][1]},
                            "dropped_requests": int(ctr[T.VK_CTR_DROPPED]), "voxel_pointer": int(ctr[T.VK_CTR_VOXEL_PTR]),
                            "allocated_blocks": int(min(vol.max, vol.max - 1 - ctr[T.VK_CTR_VOXEL_PTR])),
                            "visible_blocks": int(ctr[T.VK_CTR_VISIBLE]), "entries_sha256": sha(vol.hash_entries)[:16],
                            "gn_steps_full_level": {"median": float(np.median(steps)), "max": int(steps.max())} if len(steps) else None,
                            "loop_state": int(loop.tracker.tracker.state.cpu()[1])})
            print(json.dumps(reports[-1]), flush=True)
            window_from = i + 1
            # the tracked poses of past windows are scored: drop them (200 000 Transforms are 100 MB of Python objects)
            loop.tracked_poses[:i + 1] = [None] * (i + 1)
    torch.cuda.synchronize()
    seconds = time.time() - t0
    last_view = raycast_from(api, loop, room)
    count_after = api_loop_count(api)
    both = (first_view > 0) & (last_view > 0) if first_view is not None else None
    soak = {"frames": n, "seconds": seconds, "frames_per_s": n / seconds, "reports": reports,
            "loop_launches": count_after[0] - count_before[0], "exchange_area_clears": count_after[1] - count_before[1],
            "epoch_wraps_crossed": (count_after[0] // ((1 << 22) - 1)) - (count_before[0] // ((1 << 22) - 1))}
    if both is not None:
        diff = np.abs(first_view - last_view)[both]
        soak["map_from_the_first_pose"] = {
            "what": "raycast depth from the first pose, at frame 300 and at the end, pixels both hit",
            "pixels_both": int(both.sum()), "pixels_hit_at_300": int((first_view > 0).sum()), "pixels_hit_at_end": int((last_view > 0).sum()),
            "within_1mm_fraction": float((diff < 1e-3).mean()), "max_abs_m": float(diff.max()), "median_abs_m": float(np.median(diff))}
        soak["map_ok"] = bool(soak["map_from_the_first_pose"]["within_1mm_fraction"] >= 0.99)
    drift_file = os.path.join(ROOT, "tests", "golden", "soak_oracle_drift.json")
    if fine and os.path.exists(drift_file) and not args.excess and not args.max_weight:
        oracle = {r["frame"]: r["pose_error_max"] for r in json.load(open(drift_file))["reports"]}
        rows = [{"frame": f["frame"], "device_mm": 1e3 * f["translation_m"], "oracle_mm": 1e3 * oracle[f["frame"]]["translation_m"]}
                for f in fine if f["frame"] in oracle]
        soak["first_2000_frames_against_the_oracles_own_tracked_loop"] = {
            "what": "largest translation error of every 100 frames, the device's loop and the CPU oracle's (restated reference kernels, "
                    "float64 sums), same input: the creep of the tracked pose is the algorithm's, not the device path's",
            "rows": rows, "largest_difference_mm": max(abs(r["device_mm"] - r["oracle_mm"]) for r in rows) if rows else None}
    doc["soak_leg"] = soak
    doc["total_seconds"] = time.time() - t_start
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(doc, f, indent=1)
    print(f"{args.out}: written; parity leg {'ok' if parity['ok'] else 'FAILED'}, map {'ok' if soak.get('map_ok') else 'DEGRADED or not checked'}")
    return 0 if parity["ok"] and soak.get("map_ok") else 1


def api_loop_count(api):
    import ctypes as C
    now, clears = C.c_uint64(), C.c_uint64()
    api.check(api.lib().vk_test_hooks_loop_count(None, C.byref(now), C.byref(clears)), "vk_test_hooks_loop_count")
    return now.value, clears.value


def raycast_from(api, loop, room, index=0):
    """depth image of the map from the room sequence's pose `index`: on a COPY of the volume (the soak's own volume sees no
    call it would not have seen), after SetView with that frame — the raycast walks the visible list of the view it renders"""
    import torch
    import bench
    src = loop.vols[0]["vol"]
    torch.cuda.synchronize()
    snap = api.Volume(src.main, src.excess, voxel_length=src.voxel_length, truncation_length=src.truncation_length)
    for name in ("voxels", "hash_entries", "free_voxel_blocks", "allocation_types", "allocation_blocks", "block_visibility",
                 "visible_blocks", "counters"):
        getattr(snap, name).copy_(getattr(src, name))
    frame = api.Frame(room.depth[index], loop.k, room.truth[index])
    snap.set_view(frame, rounds=3)
    out = api.Frame(torch.zeros((bench.H, bench.W), dtype=torch.float32, device="cuda"), loop.k, room.truth[index])
    api.Tracer(snap).trace(out)
    torch.cuda.synchronize()
    depth = out.depth.cpu().numpy()
    del snap, out, frame
    torch.cuda.empty_cache()
    return depth


if __name__ == "__main__":
    sys.exit(main())
