#!/usr/bin/env python3
"""Summarise tools/traffic.sh: mean FETCH_SIZE / WRITE_SIZE per integrate launch,
with the gfx950 correction from MI355X_MICROARCH.md (FETCH_SIZE counts 128-byte
requests of 16 B/lane coalesced reads as 64 B: x2). Prints the JSON that is
committed as profiles/r01_integrate_traffic.json."""
import csv, glob, json, os, sys

out = sys.argv[1]
res = {}
calib = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals, fill = [], []
    for f in glob.glob(os.path.join(out, c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            if "integrate_pipelined_kernel" in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
            if "fill_voxels_kernel" in r["Kernel_Name"]:
                fill.append(float(r["Counter_Value"]))
    res[c] = (sum(vals) / len(vals), len(vals)) if vals else (None, 0)
    calib[c] = sum(fill) / len(fill) if fill else None
algorithmic = None
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    try:
        for line in open(os.path.join(out, c + ".log")):
            if line.startswith("{") and "roofline" in line:
                algorithmic = json.loads(line)["roofline"]["algorithmic_bytes_per_launch"]
    except OSError:
        pass
fetch, n = res["FETCH_SIZE"]
write, _ = res["WRITE_SIZE"]
doc = {
    "kernel": "integrate_pipelined_kernel<true, 0> (vk_integrate_depth)",
    "command": "tools/traffic.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 bench.py --steps 60 --warmup 20 --cpu-frames 0 (and a separate pass with --pmc WRITE_SIZE)",
    "launches_averaged": n,
    "FETCH_SIZE_KB_raw": fetch,
    "FETCH_correction": "x2: on gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B tallies 128-B requests at 64 B for 16 B/lane coalesced reads (MI355X_MICROARCH.md, HBM)",
    "WRITE_SIZE_KB_raw": write,
    "WRITE_calibration": "fill_voxels_kernel writes 749.7 MB; WRITE_SIZE reported %s KB in this run" % calib["WRITE_SIZE"],
    "note": "fabric-side request bytes; Infinity-Cache hits are counted, not excluded. The kernel writes back only the 16-byte pieces that changed, so WRITE_SIZE is below the algorithmic write volume.",
}
if fetch is not None and write is not None:
    doc["bytes_per_launch"] = fetch * 1024 * 2 + write * 1024
    doc["read_bytes_per_launch"] = fetch * 1024 * 2
    doc["write_bytes_per_launch"] = write * 1024
    if algorithmic:
        doc["algorithmic_bytes_per_launch"] = algorithmic
        doc["traffic_over_algorithmic"] = doc["bytes_per_launch"] / algorithmic
print(json.dumps(doc, indent=1))
