#!/usr/bin/env python3
"""Instruction histogram of a kernel's ISA by what the instructions are FOR ("the histogram first").

  hipcc ... --cuda-device-only -gline-tables-only -S vk_trace.hip -o /tmp/vk_trace_g.s        (tools/isa_bins.sh does it)
  python tools/isa_bins.py /tmp/vk_trace_g.s compute_points_kernelILb1E [--blocks] [--path L1,L2,...] [--source old_vk_raycast.hpp]

Every instruction is attributed to the source line its .loc names; helper lines (vk_common.hpp: add3, f2i, xform_point
...; the HIP headers) inherit the last vk_raycast.hpp / vk_trace.hip line seen before them in the same basic block, or
the next one. Lines are mapped to bins by the ANCHORS below (regular expressions on vk_raycast.hpp's own text). --blocks lists the basic
blocks (label, instructions, bins, terminator) so that the blocks of ONE path through the march loop can be named;
--path sums the histogram over exactly those blocks."""
import collections
import re
import sys

# Bins by ANCHOR: a source line of vulcan_amd/csrc/vk_raycast.hpp belongs to the bin of the last anchor (a regular expression
# matched against the file's own text, so that the table follows the source) at or above it.
SOURCE = "vulcan_amd/csrc/vk_raycast.hpp"
ANCHORS = [
    (r"^struct PointParams", "outside the march"),
    (r"^// a / b, correctly rounded, for a divisor known on the host", None),       # div_uniform: a helper, inherits its caller's bin
    (r"^struct BlockCache", "outside the march"),
    (r"^__device__ __forceinline__ Entry load_whole_entry", "table probe (global hash walk: directory miss only)"),
    (r"^__device__ __forceinline__ void file_blocks", "file_blocks (directory miss only)"),
    (r"^__device__ __forceinline__ int lookup_block", "block coordinates of p (3 x double mul, floor, cvt) + directory lookup of that block (find_block)"),
    (r"^struct Corners", "corner loads (Corners::word / tail)"),
    (r"^__device__ __forceinline__ uint32_t lds_address", "directory lookup: the corners' blocks (reads, wait, tag test)"),
    (r"^\s+const int ix = f2i\(floorf\(wx", "corner index arithmetic (floor, & 7, crossing flags, base block)"),
    (r"^\s+// Pool slots of the", "directory lookup: the corners' blocks (reads, wait, tag test)"),
    (r"^\s+if \(__any\(moved\)\)", "directory lookup: the corners' blocks (reads, wait, tag test)"),
    (r"^\s+while \(__any\(missing != 0\)\)", "corner-block miss loop (rare)"),
    (r"^\s+// corner c is in block", "slot selection network (cndmask tree)"),
    (r"^\s+Corners<POOL32> C;", "corner address arithmetic (slot * 10240 + voxel * 20, absent bits, fractions)"),
    (r"^__device__ __forceinline__ float corner_distance", "corner distance: absent override, trilinear"),
    (r"^__device__ __forceinline__ f3 corner_color", "corner colour (after the march)"),
    (r"^__device__ __forceinline__ void march_ray_nested", "ray set-up"),
    (r"^\s+for \(;;\)", "block coordinates of p (3 x double mul + floor + cvt) + loop head"),
    (r"^\s+const int data = find_block", "block coordinates of p (3 x double mul, floor, cvt) + directory lookup of that block (find_block)"),
    (r"^\s+if \(data >= 0\)", "voxel coordinates + nearest-voxel load"),
    (r"^\s+if \(!refine\) sample = ", "sample decision + bookkeeping of the last sample"),
    (r"^\s+if \(refine\)$", "step (p += dir * ...)"),
    (r"^\s+const float depth = xform_point", "depth of p, exit tests"),
    (r"^\s+if \(__any\(sampled\)\)", "colour of last sample + stores"),
]


def load_bins(root):
    import os
    path = sys.argv[sys.argv.index("--source") + 1] if "--source" in sys.argv else os.path.join(root, SOURCE)
    lines = open(path).read().split("\n")
    table, current, used = {}, "outside the march", set()
    for number, text in enumerate(lines, 1):
        for rx, name in ANCHORS:
            if re.search(rx, text):
                current = name
                used.add(rx)
        table[number] = current
    return table


LINE_BINS = None


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
    global LINE_BINS
    if LINE_BINS is None:
        import os
        LINE_BINS = load_bins(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if f.endswith("vk_raycast.hpp") and l in LINE_BINS:
        return LINE_BINS[l]          # (None for a helper's lines)
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
                i["bin"] = "outside the march"
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
    if "--trip" in sys.argv:
        # the march loop's body: from the first instruction of its head to the last of its exit tests (the second, inlined
        # copy of resolve_corners — the colour of the last sample — lies behind it and is left out)
        flat = [i for block in blocks.values() for i in block]
        head = next(n for n, i in enumerate(flat) if i["bin"].startswith("block coordinates of p") and i["block"] != "entry")
        # ... up to the basic block in which the code behind the loop (the colour of the last sample) begins
        after = next(n for n, i in enumerate(flat) if n > head and i["bin"].startswith("colour of last sample"))
        while after > head and flat[after - 1]["block"] == flat[after]["block"]:
            after -= 1
        body = flat[head:after]
        rare = ("table probe", "file_blocks", "corner-block miss loop", "slot selection network", "outside the march", "ray set-up",
                "corner colour", "colour of last sample")
        absent_bins = ("block coordinates of p", "step (p +=", "depth of p")
        histogram([i for i in body if not i["bin"].startswith(rare)],
                  f"{kernel}: ONE SAMPLED TRIP through the march loop, every block of the directory met before (static upper bound: "
                  "all instructions of the loop body outside the miss handling)")
        print()
        histogram([i for i in body if i["bin"].startswith(absent_bins)],
                  f"{kernel}: ONE PASS THROUGH A BLOCK THAT IS NOT THERE (loop head, block lookup in the directory, step, exit tests)")
        print()
        histogram([i for i in body if i["bin"].startswith(rare[:4])], f"{kernel}: miss handling inside the loop (not on the two paths above)")
        return
    histogram(instrs, f"{kernel}: whole kernel (static)")


if __name__ == "__main__":
    main()
