#!/usr/bin/env python3
"""How far floating-point contraction moves the results of the path — measured, on the CPU.

The reference is compiled by nvcc, whose default (--fmad=true) fuses a multiplication and the addition that
consumes it into one FMA wherever it likes; the oracle and the HIP kernels are compiled with contraction OFF so
that they can be compared bit for bit with each other. Neither can reproduce nvcc's own choice of contractions,
so the question north_star's tolerance ("TSDF within 1e-4 of the reference CUDA path") leaves open is how much
ANY such choice can move a result. This tool answers it with the oracle's own sources compiled twice:

    oracle/liboracle.so       -ffp-contract=off            (the checker)
    oracle/liboracle_fma.so   -ffp-contract=fast -mfma     (every product that feeds a sum is fused: gcc contracts
                                                            across statements, the most aggressive choice there is)

and the same inputs through both: BASELINE configs[0] (one frame into the dense 128^3 region), five frames of
configs[1] (depth) and of configs[2] (RGB-D, light integrator), each followed by the raycast. A difference has two
possible sources:

  * rounding: a fused multiply-add rounds once where the plain pair rounds twice — a few units in the last place
    per operation, far below 1e-4;
  * a discrete decision that falls the other way because of such a rounding: `int(uv)` picks the neighbouring
    depth pixel (depth_integrator.cu:44-52 — the reference's own test exempts these "border points",
    tests/integrator_test.cu:160-168,203-206), `distance > -truncation` admits or rejects a voxel (:58), a ray's
    march takes one step more (tracer.cu:386-427). These are jumps, not errors: both outcomes are what the
    reference's arithmetic yields under some legal choice of contractions.

The tool lists every voxel whose TSDF differs by more than 1e-4 and checks that each one sits on such a decision
boundary in some frame (projection within `--eps` pixels of a pixel boundary, or its signed distance within 1e-5 m
of the truncation band's edge), using a float64 evaluation of the projection that is independent of both builds.

    python tools/contraction_sensitivity.py [--size 640x480] [--frames 5] [--json out.json]
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

VOXEL, TRUNC, MAIN, EXCESS = 0.005, 0.04, 65024, 8192
LIGHT = (2.0, (0.025, 0.08, 0.0))


def libraries():
    from oracle import oracle as orc
    plain = orc.lib()
    so = os.path.join(ROOT, "oracle", "liboracle_fma.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in os.listdir(os.path.join(ROOT, "oracle")) if f.endswith((".c", ".h"))]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-B", "fma"], stdout=subprocess.DEVNULL)
    fused = C.CDLL(so)
    for lib in (plain, fused):
        lib.orc_icp_solve_update.restype = C.c_float
    return orc, plain, fused


def sphere_depth(k, w, h, radius=2.0):
    y, x = np.mgrid[0:h, 0:w]
    rx, ry = (x + 0.5 - k.cx) / k.fx, (y + 0.5 - k.cy) / k.fy
    return (radius / np.sqrt(rx * rx + ry * ry + 1.0)).astype(np.float32)


def run_sequence(orc, lib, kind, w, h, frames, threads):
    """configs[1] / configs[2]: the bench's sequence (camera yawing at the centre of a 2 m sphere), SetView x3,
    Integrate, Trace per frame. Returns the final table + voxels and every frame's raycast depth."""
    import scenes
    from vulcan_amd import vk_types as T
    orc._LIB = lib
    orc.set_threads(threads)
    s = w / 640.0
    k = T.Projection.make(*(np.float32(s) * np.float32(v) for v in scenes.APP_INTRINSICS))
    depth = sphere_depth(k, w, h)
    color = scenes.checker_color(w, h, 0.1, 0.9)
    light = T.Light.make(*LIGHT)
    hv = orc.HostVolume(MAIN, EXCESS, voxel_length=VOXEL, truncation_length=TRUNC)
    hf = orc.HostFrame(depth, k, T.Transform.identity(), color=color if kind == "rgbd" else None)
    raycasts, poses = [], []
    for i in range(frames):
        hf.depth_to_world = scenes.orbit_pose(i, 0.5)
        if kind == "rgbd":
            hf.compute_normals()
        for _ in range(3):
            hv.set_view(hf, orc.POLICY_MAXKEY)
        orc.integrate_depth(hv, hf)
        if kind == "rgbd":
            orc.integrate_light_color(hv, hf, light, orc.light_frame_mask(hf, 0.2))
        odepth, ocolor, onormals, _ = orc.trace(hv, hf)
        raycasts.append((odepth, ocolor))
        poses.append(hf.depth_to_world)
    return dict(entries=hv.hash_entries.copy(), voxels=hv.voxels.copy(), raycasts=raycasts, poses=poses, k=k, depth=depth,
                visible=hv.visible_count)


def run_dense(orc, lib, w, h, threads):
    """configs[0]: one frame into 4096 hand-placed blocks (the layout of bench.py's cpu_baseline and of
    tests/test_gpu_configs.py::test_configs0_dense_128_on_the_device), scaled with the image."""
    import scenes
    from vulcan_amd import vk_types as T
    orc._LIB = lib
    orc.set_threads(threads)
    s = w / 640.0
    k = T.Projection.make(*(np.float32(s) * np.float32(v) for v in scenes.APP_INTRINSICS))
    depth = sphere_depth(k, w, h)
    hv = orc.HostVolume(8192, 1024, voxel_length=VOXEL, truncation_length=TRUNC)
    origin = np.array([[x, y, z] for z in range(42, 58) for y in range(-8, 8) for x in range(-8, 8)], dtype=np.int16)
    n = len(origin)
    hv.hash_entries["block"]["origin"][:n] = origin
    hv.hash_entries["data"][:n] = np.arange(n)
    hv.hash_entries["next"][:n] = -1
    hv.visible_blocks[:n] = np.arange(n)
    hv.counters[T.VK_CTR_VISIBLE] = n
    hf = orc.HostFrame(depth, k, T.Transform.identity())
    orc.integrate_depth(hv, hf)
    return dict(entries=hv.hash_entries.copy(), voxels=hv.voxels.copy(), raycasts=[], poses=[hf.depth_to_world], k=k,
                depth=depth, visible=n)


def near_a_decision(run, slots, voxel_ids, w, h, eps_px, eps_m):
    """For the listed voxels (pool slot, index in block): does the voxel sit on a decision boundary in ANY frame?
    float64 evaluation of depth_integrator.cu:35-58, independent of either build. Returns three boolean arrays:
    pixel boundary, truncation edge, image border."""
    entries = run["entries"]
    by_slot = {int(e["data"]): tuple(int(c) for c in e["block"]["origin"]) for e in entries if e["data"] >= 0}
    origin = np.array([by_slot[int(s)] for s in slots], dtype=np.float64)
    vx, vy, vz = voxel_ids & 7, (voxel_ids >> 3) & 7, voxel_ids >> 6
    block_length = np.float64(np.float32(8) * np.float32(VOXEL))
    voxel_length = np.float64(np.float32(VOXEL))
    world = origin * block_length + (np.stack([vx, vy, vz], 1) + 0.5) * voxel_length
    k, depth = run["k"], run["depth"]
    pixel = np.zeros(len(slots), bool)
    edge = np.zeros(len(slots), bool)
    border = np.zeros(len(slots), bool)
    for pose in run["poses"]:
        inv = pose.inverse_matrix().astype(np.float64)
        cam = world @ inv[:3, :3].T + inv[:3, 3]
        u = np.float64(k.fx) * cam[:, 0] / cam[:, 2] + np.float64(k.cx)
        v = np.float64(k.fy) * cam[:, 1] / cam[:, 2] + np.float64(k.cy)
        du, dv = np.abs(u - np.round(u)), np.abs(v - np.round(v))
        close = (du < eps_px) | (dv < eps_px)
        inside = (u > -eps_px) & (u < w + eps_px) & (v > -eps_px) & (v < h + eps_px)
        pixel |= close & inside
        border |= inside & ((np.abs(u) < eps_px) | (np.abs(u - w) < eps_px) | (np.abs(v) < eps_px) | (np.abs(v - h) < eps_px))
        ui, vi = np.clip(u.astype(np.int64), 0, w - 1), np.clip(v.astype(np.int64), 0, h - 1)
        distance = depth[vi, ui].astype(np.float64) - cam[:, 2]
        edge |= inside & (np.abs(distance + np.float64(np.float32(TRUNC))) < eps_m)
    return pixel, edge, border


def compare(name, a, b, w, h, eps_px, eps_m):
    out = {"case": name, "visible_blocks": int(a["visible"])}
    same_table = np.array_equal(a["entries"], b["entries"])
    out["hash_table_identical"] = bool(same_table)
    if not same_table:
        # align by block origin: the blocks both builds allocated
        def keyed(run):
            e = run["entries"]
            live = e["data"] >= 0
            o = e["block"]["origin"][live].astype(np.int64)
            return {(int(x), int(y), int(z)): int(d) for (x, y, z), d in zip(o, e["data"][live])}
        ka, kb = keyed(a), keyed(b)
        common = sorted(set(ka) & set(kb))
        out["blocks_only_in_one_build"] = len(set(ka) ^ set(kb))
        sa = np.array([ka[c] for c in common])
        sb = np.array([kb[c] for c in common])
    else:
        sa = sb = np.sort(a["entries"]["data"][a["entries"]["data"] >= 0])
        out["blocks_only_in_one_build"] = 0
    va = a["voxels"].reshape(-1, 512)[sa]
    vb = b["voxels"].reshape(-1, 512)[sb]
    dd = np.abs(va["distance"].astype(np.float64) - vb["distance"])
    dw = np.abs(va["distance_weight"].astype(np.int64) - vb["distance_weight"])
    touched = (va["distance_weight"] > 0) | (vb["distance_weight"] > 0)
    out["voxels_integrated"] = int(touched.sum())
    out["voxels_bit_identical"] = int((va.view(np.uint8).reshape(len(sa), 512, 20) == vb.view(np.uint8).reshape(len(sb), 512, 20)).all(-1)[touched].sum())
    out["tsdf_max_abs_diff"] = float(dd.max())
    out["weight_max_abs_diff"] = int(dw.max())
    big = np.argwhere((dd > 1e-4) | (dw > 0))
    out["voxels_over_1e-4_or_weight_differs"] = int(len(big))
    small = dd[(dd <= 1e-4) & (dw == 0)]
    out["tsdf_max_abs_diff_of_the_rest"] = float(small.max()) if small.size else 0.0
    if len(big):
        pixel, edge, border = near_a_decision(a, sa[big[:, 0]], big[:, 1], w, h, eps_px, eps_m)
        out["of_those_on_a_pixel_boundary"] = int(pixel.sum())
        out["of_those_on_the_truncation_edge"] = int((edge & ~pixel).sum())
        out["of_those_on_the_image_border"] = int(border.sum())
        out["unexplained"] = int((~pixel & ~edge).sum())
    else:
        out["of_those_on_a_pixel_boundary"] = out["of_those_on_the_truncation_edge"] = out["of_those_on_the_image_border"] = 0
        out["unexplained"] = 0
    if "color" in va.dtype.names and (va["color_weight"] > 0).any():
        both = (va["color_weight"] > 0) & (vb["color_weight"] > 0) & (dd <= 1e-4) & (dw == 0) & \
               (va["color_weight"] == vb["color_weight"])
        dc = np.abs(va["color"].astype(np.float64) - vb["color"]).max(-1)
        out["color_max_abs_diff_same_weights"] = float(dc[both].max()) if both.any() else 0.0
        out["color_over_1e-4_same_weights"] = int((dc[both] > 1e-4).sum()) if both.any() else 0
    # the raycasts
    if a["raycasts"]:
        worst, flips, hits, p9999, all_d = 0.0, 0, 0, 0.0, []
        for (da, _), (db, _) in zip(a["raycasts"], b["raycasts"]):
            d = np.abs(da.astype(np.float64) - db)
            one_sided = (da > 0) != (db > 0)
            jump = (d > 1e-4) | one_sided
            flips += int(jump.sum())
            hits += int(((da > 0) | (db > 0)).sum())
            rest = d[~jump]
            worst = max(worst, float(rest.max()) if rest.size else 0.0)
            p9999 = max(p9999, float(np.quantile(d[(da > 0) & (db > 0)], 0.9999)))
            all_d.append(d[(da > 0) & (db > 0)])
        out["raycast_pixels"] = hits
        out["raycast_pixels_over_1e-4_m"] = flips
        out["raycast_depth_max_abs_diff_of_the_rest_m"] = worst
        out["raycast_depth_p99.99_abs_diff_m"] = p9999
        all_d = np.concatenate(all_d)
        out["raycast_depth_abs_diff_quantiles_m"] = {q: float(np.quantile(all_d, float(q))) for q in ("0.5", "0.99", "0.999")}
        out["raycast_pixels_bit_identical"] = int((all_d == 0).sum())
        out["raycast_depth_max_abs_diff_m"] = max(float(np.abs(da.astype(np.float64) - db).max()) for (da, _), (db, _) in zip(a["raycasts"], b["raycasts"]))
    return out


def measure(size=(640, 480), frames=5, threads=8, eps_px=2e-3, eps_m=1e-5, cases=("configs0", "configs1", "configs2")):
    orc, plain, fused = libraries()
    w, h = size
    results = []
    try:
        for case in cases:
            t0 = time.time()
            if case == "configs0":
                a, b = run_dense(orc, plain, w, h, threads), run_dense(orc, fused, w, h, threads)
            else:
                kind = "depth" if case == "configs1" else "rgbd"
                a, b = run_sequence(orc, plain, kind, w, h, frames, threads), run_sequence(orc, fused, kind, w, h, frames, threads)
            r = compare(case, a, b, w, h, eps_px, eps_m)
            r["size"], r["frames"], r["seconds"] = f"{w}x{h}", (1 if case == "configs0" else frames), round(time.time() - t0, 1)
            results.append(r)
    finally:
        orc._LIB = plain
        orc.set_threads(1)
    return results


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="640x480")
    ap.add_argument("--frames", type=int, default=5)
    ap.add_argument("--threads", type=int, default=min(16, os.cpu_count() or 1))
    ap.add_argument("--eps", type=float, default=2e-3, help="pixels: how close to a pixel boundary counts as 'on it'")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    w, h = (int(v) for v in args.size.split("x"))
    results = measure((w, h), args.frames, args.threads, args.eps)
    for r in results:
        print(json.dumps(r))
    if args.json:
        with open(args.json, "w") as f:
            json.dump({"eps_px": args.eps, "results": results}, f, indent=1)


if __name__ == "__main__":
    main()
