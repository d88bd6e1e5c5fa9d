#!/usr/bin/env python3
"""Summarise tools/traffic.sh: mean FETCH_SIZE / WRITE_SIZE per launch of the integrate and the
raycast kernel, with the gfx950 correction from MI355X_MICROARCH.md (FETCH_SIZE counts the
128-byte requests of 16 B/lane coalesced reads as 64 B: x2) — which round 6 measured to hold for the
raycast's scattered 4- and 12-byte loads as well (tools/fetch_calibration.sh). Writes <outdir>/<name>_traffic.json
(committed under profiles/ as rNN_<name>_traffic.json)."""
import csv, glob, json, os, sys

out, workload = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "rgbd")
# (the raycast is compute_points_kernel, or — bench.py's default at given poses — trace_and_request_kernel, which also makes
# the next frame's request pass: its bytes then include that pass's, i.e. the depth image read and the normals written)
KERNELS = {"integrate": ("integrate_pipelined_kernel",), "raycast": ("compute_points_kernel", "trace_and_request_kernel")}
res = {k: {} for k in KERNELS}
calib = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = {k: [] for k in KERNELS}
    fill = []
    for f in glob.glob(os.path.join(out, c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            for k, pats in KERNELS.items():
                if any(pat in r["Kernel_Name"] for pat in pats):
                    vals[k].append(float(r["Counter_Value"]))
            if "fill_voxels_kernel" in r["Kernel_Name"]:
                fill.append(float(r["Counter_Value"]))
    for k in KERNELS:
        res[k][c] = (sum(vals[k]) / len(vals[k]), len(vals[k])) if vals[k] else (None, 0)
    calib[c] = sum(fill) / len(fill) if fill else None
bench = None
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    try:
        for line in open(os.path.join(out, c + ".log")):
            if line.startswith("{") and "roofline" in line:
                bench = json.loads(line)
    except OSError:
        pass
cmd = f"tools/traffic.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 bench.py --workload {workload} --only --steps 60 --warmup 20 --cpu-seconds 0 (and a separate pass with --pmc WRITE_SIZE)"
for k in KERNELS:
    fetch, n = res[k]["FETCH_SIZE"]
    write, _ = res[k]["WRITE_SIZE"]
    doc = {"kernel": " | ".join(KERNELS[k]), "workload": workload, "command": cmd, "launches_averaged": n,
           "FETCH_SIZE_KB_raw": fetch, "WRITE_SIZE_KB_raw": write,
           "WRITE_calibration": "fill_voxels_kernel writes 749.7 MB; WRITE_SIZE reported %s KB in this run" % calib["WRITE_SIZE"],
           "note": "fabric-side request bytes; Infinity-Cache hits are counted, not excluded."}
    if fetch is None or write is None:
        continue
    if k == "integrate":
        doc["FETCH_correction"] = ("x2: on gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B tallies 128-B requests at 64 B for 16 B/lane "
                                   "coalesced reads (MI355X_MICROARCH.md, HBM)")
        doc["read_bytes_per_launch"] = fetch * 1024 * 2
        doc["note"] += " The kernel writes back only the 16-byte pieces that changed, so WRITE_SIZE is below the algorithmic write volume."
        if bench:
            doc["algorithmic_bytes_per_launch"] = bench["roofline"]["algorithmic_bytes_per_launch"]
    else:
        # Calibrated in round 6 (profiles/r06_fetch_calibration.json, tools/fetch_calibration.sh): on 4- and 12-byte loads at the
        # 20-byte voxel stride — dense, sparse, and the raycast's own 2 x 2 x 2 corner shape alike — the L2 issues exactly ONE
        # fabric request per touched 128-byte line (TCC_EA0_RDREQ = lines, no 32-byte requests) and FETCH_SIZE tallies it at
        # 64 B, as for float4 streaming reads: the bytes MOVED are 2 x FETCH_SIZE for this kernel too.
        doc["FETCH_correction"] = ("x2: one 128-byte fabric request per touched line, tallied at 64 B — measured on this kernel's own "
                                   "access shape (4- and 12-byte loads at a 20-byte stride), profiles/r06_fetch_calibration.json")
        doc["read_bytes_per_launch"] = fetch * 1024 * 2
    doc["write_bytes_per_launch"] = write * 1024
    doc["bytes_per_launch"] = doc["read_bytes_per_launch"] + doc["write_bytes_per_launch"]
    if doc.get("algorithmic_bytes_per_launch"):
        doc["traffic_over_algorithmic"] = doc["bytes_per_launch"] / doc["algorithmic_bytes_per_launch"]
    name = ("integrate" if workload == "depth" else "integrate_rgbd") if k == "integrate" else "raycast"
    with open(os.path.join(out, name + "_traffic.json"), "w") as f:
        json.dump(doc, f, indent=1)
    print(name, json.dumps(doc, indent=1))
