#!/usr/bin/env python3
"""Instruction histogram of a kernel's ISA by what the instructions are FOR ("the histogram first").

  hipcc ... --cuda-device-only -gline-tables-only -S vk_trace.hip -o /tmp/vk_trace_g.s        (tools/isa_bins.sh does it)
  python tools/isa_bins.py /tmp/vk_trace_g.s compute_points_kernelILb1E [--blocks] [--path L1,L2,...]

Every instruction is attributed to the source line its .loc names; helper lines (vk_common.hpp: add3, f2i, xform_point
...; the HIP headers) inherit the last vk_raycast.hpp / vk_trace.hip line seen before them in the same basic block, or
the next one. Lines are mapped to bins by the table BINS below (vk_raycast.hpp line ranges). --blocks lists the basic
blocks (label, instructions, bins, terminator) so that the blocks of ONE path through the march loop can be named;
--path sums the histogram over exactly those blocks."""
import collections
import re
import sys

# (file, first line, last line) -> bin. vk_raycast.hpp as of this commit; tools/isa_bins.py --check verifies the anchors.
BINS = [
    ("vk_raycast.hpp", 99, 121, "table probe (global hash walk: directory miss only)"),
    ("vk_raycast.hpp", 135, 160, "file_blocks (directory miss only)"),
    ("vk_raycast.hpp", 165, 191, "directory lookup: march block (find_block / lookup_block)"),
    ("vk_raycast.hpp", 211, 232, "corner loads (Corners::word / tail)"),
    ("vk_raycast.hpp", 249, 258, "corner index arithmetic (floor, & 7, crossing flags)"),
    ("vk_raycast.hpp", 268, 296, "directory lookup: neighbour blocks (8 guarded LDS reads + compares)"),
    ("vk_raycast.hpp", 297, 308, "neighbour miss loop"),
    ("vk_raycast.hpp", 310, 322, "slot selection network (cndmask tree)"),
    ("vk_raycast.hpp", 324, 345, "corner address arithmetic (slot * 10240 + voxel * 20, absent bits, fractions)"),
    ("vk_raycast.hpp", 350, 366, "corner distance: loads, absent override, trilinear"),
    ("vk_raycast.hpp", 373, 413, "corner colour (after the march)"),
    ("vk_raycast.hpp", 429, 450, "ray set-up"),
    ("vk_raycast.hpp", 454, 456, "block coordinates of p (3 x double mul + floor + cvt)"),
    ("vk_raycast.hpp", 457, 473, "loop head / find_block call"),
    ("vk_raycast.hpp", 474, 499, "voxel coordinates + nearest-voxel load"),
    ("vk_raycast.hpp", 518, 532, "sample decision + bookkeeping of the last sample"),
    ("vk_raycast.hpp", 534, 557, "step (p += dir * ...)"),
    ("vk_raycast.hpp", 559, 574, "depth of p, exit tests"),
    ("vk_raycast.hpp", 579, 597, "colour of last sample + stores"),
]

CLASSES = [
    (re.compile(r"^s_waitcnt"), "s_waitcnt"),
    (re.compile(r"^s_nop"), "s_nop"),
    (re.compile(r"^s_(c?branch|cbranch|setpc|endpgm|barrier)"), "branch"),
    (re.compile(r"^(global|flat|buffer|scratch)_load"), "vmem load"),
    (re.compile(r"^(global|flat|buffer|scratch)_(store|atomic)"), "vmem store/atomic"),
    (re.compile(r"^s_load|^s_buffer_load"), "smem load"),
    (re.compile(r"^ds_"), "lds"),
    (re.compile(r"^v_cndmask"), "v_cndmask"),
    (re.compile(r"^v_cmp|^v_cmpx"), "v_cmp"),
    (re.compile(r"^v_(readlane|readfirstlane|writelane)"), "v_readlane"),
    (re.compile(r"^v_.*f64"), "valu f64"),
    (re.compile(r"^v_"), "valu other"),
    (re.compile(r"^s_"), "salu"),
]


def classify(mnemonic):
    for rx, name in CLASSES:
        if rx.match(mnemonic):
            return name
    return "other"


def parse(path, kernel):
    files, out, inside = {}, [], False
    cur = (None, 0)
    block = "entry"
    for raw in open(path):
        line = raw.rstrip("\n")
        m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', line)
        if m:
            files[int(m.group(1))] = m.group(2)
            continue
        if not inside:
            if re.match(r"^_Z\w*" + re.escape(kernel) + r"\w*:", line):
                inside = True
            continue
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
        if m:
            cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            block = m.group(1)
            continue
        s = line.strip()
        if not s or s.startswith(".") or s.startswith(";"):
            continue
        mnemonic = s.split()[0]
        out.append(dict(block=block, op=mnemonic, text=s.split(";")[0].strip(), file=cur[0], line=cur[1]))
        if mnemonic == "s_endpgm":
            break
    return out


def bin_of(f, l):
    for bf, lo, hi, name in BINS:
        if f.endswith(bf) and lo <= l <= hi:
            return name
    return None


def attribute(instrs):
    """helper lines inherit the nearest raycast line of their basic block (previous, else next)"""
    by_block = collections.OrderedDict()
    for i in instrs:
        by_block.setdefault(i["block"], []).append(i)
    for block in by_block.values():
        last = None
        for i in block:
            b = bin_of(i["file"], i["line"])
            if b is None:
                i["bin"] = last
            else:
                i["bin"] = last = b
        nxt = None
        for i in reversed(block):
            if i["bin"] is None:
                i["bin"] = nxt
            else:
                nxt = i["bin"]
        for i in block:
            if i["bin"] is None:
                i["bin"] = "outside the march (tile set-up, bounds merge, normals, request pass)" if not i["file"].endswith("vk_raycast.hpp") else f"{i['file']}:{i['line']}"
    return by_block


def histogram(instrs, title):
    bins = collections.OrderedDict()
    for i in instrs:
        bins.setdefault(i["bin"], collections.Counter())[classify(i["op"])] += 1
    total = sum(sum(c.values()) for c in bins.values())
    print(f"== {title}: {total} instructions")
    cols = ["valu other", "valu f64", "v_cndmask", "v_cmp", "v_readlane", "salu", "lds", "vmem load", "vmem store/atomic", "smem load", "s_waitcnt", "s_nop", "branch", "other"]
    print("%-92s %5s  %s" % ("bin", "all", " ".join("%6s" % c.split()[-1][:6] for c in cols)))
    for name, c in sorted(bins.items(), key=lambda kv: -sum(kv[1].values())):
        print("%-92s %5d  %s" % (name[:92], sum(c.values()), " ".join("%6d" % c.get(col, 0) for col in cols)))
    tot = collections.Counter()
    for c in bins.values():
        tot.update(c)
    print("%-92s %5d  %s" % ("total", total, " ".join("%6d" % tot.get(col, 0) for col in cols)))


def main():
    path, kernel = sys.argv[1], sys.argv[2]
    instrs = parse(path, kernel)
    blocks = attribute(instrs)
    if "--blocks" in sys.argv:
        for label, block in blocks.items():
            c = collections.Counter(i["bin"] for i in block)
            lines = sorted({i["line"] for i in block if i["file"].endswith("vk_raycast.hpp")})
            term = block[-1]["text"] if block[-1]["op"].startswith("s_c") or block[-1]["op"].startswith("s_b") else "(falls through)"
            print(f"{label:12s} {len(block):4d}  lines {lines[0] if lines else '-'}..{lines[-1] if lines else '-'}  {term}")
            for name, n in c.most_common(3):
                print(f"{'':18s}{n:4d}  {name}")
        return
    if "--path" in sys.argv:
        names = sys.argv[sys.argv.index("--path") + 1].split(",")
        chosen = [i for label in names for i in blocks[label]]
        histogram(chosen, f"{kernel}: blocks {','.join(names)}")
        return
    histogram(instrs, f"{kernel}: whole kernel (static)")


if __name__ == "__main__":
    main()
