#!/usr/bin/env python3
"""What rocprofv3's FETCH_SIZE tallies for the RAYCAST's access shape (the gather amplification was quoted
as 0.66 "or 1.27 if the guide's x2 applied" — a figure with a factor-of-two ambiguity is not a figure).

  python3 tools/fetch_calibration.py run            launches tools/probe's vk_probe_gather in its four modes over a buffer of
                                                    voxel blocks no line of which is touched twice (run it under rocprofv3
                                                    --pmc FETCH_SIZE: tools/fetch_calibration.sh) and writes the set of lines
                                                    each mode touches — enumerated on the host from the same addresses — to
                                                    <out>/expected.json
  python3 tools/fetch_calibration.py summary <out>  reads the counter CSVs and expected.json: bytes tallied per touched 64-byte
                                                    and 128-byte line for every mode, and the factor that turns FETCH_SIZE
                                                    into bytes moved for 4- and 12-byte loads at a 20-byte stride

MI355X_MICROARCH.md (HBM): FETCH_SIZE = TCC_EA0_RDREQ x 64 B; a 128-byte request is tallied at 64 B, which is why a wide
coalesced read shows half its bytes. Mode 3 (float4, the guide's calibrated shape) is the control."""
import csv
import ctypes as C
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BLOCKS = 100000                   # 1.02 GB: far beyond the 32 MiB of L2 and the 256 MiB Infinity Cache
MODES = {0: "dense 4-byte loads (all 512 distances of a block)", 1: "sparse 4-byte loads (one word every 160 bytes)",
         2: "corners: 4-byte distance + 12-byte rgb of 2x2x2 voxels (the raycast's trilinear sample)",
         3: "float4 (16 bytes per lane, coalesced: the guide's calibrated shape)"}


def gather_hash(block, lane):
    x = (block * np.uint32(0x9E3779B1) + lane * np.uint32(0x85EBCA77) + np.uint32(0x165667B1)).astype(np.uint32)
    x ^= x >> np.uint32(15)
    x = (x * np.uint32(0x2C1B3C6D)).astype(np.uint32)
    x ^= x >> np.uint32(12)
    x = (x * np.uint32(0x297A2D39)).astype(np.uint32)
    x ^= x >> np.uint32(15)
    return x


def touched_lines(mode, blocks):
    """(distinct 64-byte lines, distinct 128-byte lines, bytes the loads ask for) over all blocks of a launch"""
    if mode == 0:
        return blocks * 160, blocks * 80, blocks * 512 * 4
    if mode == 3:
        return blocks * 160, blocks * 80, blocks * 10240
    if mode == 1:
        p = np.arange(64) * 8 * 20
        return blocks * len(np.unique(p // 64)), blocks * len(np.unique(p // 128)), blocks * 64 * 4
    n64 = n128 = 0
    lane = np.arange(64, dtype=np.uint32)[None, :]
    for first in range(0, blocks, 10000):
        blk = np.arange(first, min(blocks, first + 10000), dtype=np.uint32)[:, None]
        with np.errstate(over="ignore"):
            h = gather_hash(blk, lane)
        x, y, z = (h & 7) % 7, ((h >> 3) & 7) % 7, ((h >> 6) & 7) % 7
        marks64 = np.zeros((len(blk), 160), dtype=bool)
        rows = np.arange(len(blk))[:, None]
        for c in range(8):
            v = (z + (c >> 2)) * 64 + (y + ((c >> 1) & 1)) * 8 + x + (c & 1)
            p = v.astype(np.int64) * 20
            for line in (p // 64, (p + 15) // 64):              # a voxel's first 16 bytes: distance + rgb
                marks64[rows, line] = True
        n64 += int(marks64.sum())
        n128 += int((marks64[:, 0::2] | marks64[:, 1::2]).sum())
    return n64, n128, blocks * 64 * 8 * 16


def run(out):
    import torch
    from vulcan_amd import api
    api.lib()
    pl = C.CDLL(os.path.join(ROOT, "vulcan_amd", "lib", "libvk_probe.so"))
    pl.vk_probe_gather.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    voxels = torch.empty(BLOCKS * 10240, dtype=torch.uint8, device="cuda")
    voxels.zero_()
    sink = torch.zeros(BLOCKS, dtype=torch.float32, device="cuda")
    flush = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for mode in MODES:
        for _ in range(3):
            flush.fill_(1)                                       # 1 GiB through the caches between two launches
            rc = pl.vk_probe_gather(voxels.data_ptr(), BLOCKS, mode, sink.data_ptr(), api.stream())
            assert rc == 0, rc
        torch.cuda.synchronize()
    os.makedirs(out, exist_ok=True)
    expected = {}
    for mode, what in MODES.items():
        n64, n128, asked = touched_lines(mode, BLOCKS)
        expected[str(mode)] = {"what": what, "blocks": BLOCKS, "lines_64B": n64, "lines_128B": n128, "bytes_asked_for": asked}
    with open(os.path.join(out, "expected.json"), "w") as f:
        json.dump(expected, f, indent=1)
    print(json.dumps(expected, indent=1))


def summary(out):
    expected = json.load(open(os.path.join(out, "expected.json")))
    counters = {}
    for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "gather_kernel" not in name:
                continue
            mode = name[name.index("<") + 1]
            counters.setdefault(r["Counter_Name"], {}).setdefault(mode, []).append(float(r["Counter_Value"]))
    doc = {"tool": "tools/fetch_calibration.sh", "modes": {}, "raw_counters": {c: {m: float(np.mean(v)) for m, v in d.items()} for c, d in counters.items()}}
    fetch = counters.get("FETCH_SIZE", {})
    for mode, e in expected.items():
        if mode not in fetch:
            continue
        kb = float(np.mean(fetch[mode]))
        b = kb * 1024
        doc["modes"][mode] = {"what": e["what"], "launches": len(fetch[mode]), "FETCH_SIZE_KB": kb,
                              "bytes_tallied_per_touched_64B_line": b / e["lines_64B"],
                              "bytes_tallied_per_touched_128B_line": b / e["lines_128B"],
                              "lines_64B": e["lines_64B"], "lines_128B": e["lines_128B"], "bytes_asked_for": e["bytes_asked_for"]}
    m = doc["modes"]
    if "0" in m and "3" in m and "2" in m:
        control = m["3"]["bytes_tallied_per_touched_128B_line"]           # the guide: 64 (a 128-byte request tallied at 64 B)
        dense = m["0"]["bytes_tallied_per_touched_128B_line"]
        corners = m["2"]["bytes_tallied_per_touched_128B_line"]
        doc["reading"] = {
            "control_float4_bytes_per_128B_line": control,
            "dense_4B_loads_bytes_per_128B_line": dense,
            "corner_loads_bytes_per_128B_line": corners,
            "corner_loads_bytes_per_64B_line": m["2"]["bytes_tallied_per_touched_64B_line"],
        }
        # the factor that turns FETCH_SIZE into bytes MOVED for the raycast's shape: the L2 fetches whole lines; what matters
        # is how many bytes the counter tallies for a line that was moved once
        doc["reading"]["verdict"] = (
            "FETCH_SIZE tallies %.1f B per touched 128-byte line for the raycast's corner loads (float4 control: %.1f B, "
            "dense 4-byte loads: %.1f B)." % (corners, control, dense))
    with open(os.path.join(out, "fetch_calibration.json"), "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "fetch_calibration"))
    else:
        summary(sys.argv[2])
