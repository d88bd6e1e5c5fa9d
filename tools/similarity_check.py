#!/usr/bin/env python3
"""Self-check: line-level similarity of every source file in this repository
against every file of the upstream checkout (when it is mounted). Prints the
closest upstream file for each of ours, highest first. Development aid only."""
import difflib, os, sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXT = (".h", ".hpp", ".cuh", ".cpp", ".c", ".hip", ".cu", ".py", ".in", ".txt", ".cmake")

def lines(path):
    with open(path, errors="ignore") as f:
        return [l.strip() for l in f if l.strip()]

def walk(root, skip=()):
    for d, dirs, files in os.walk(root):
        dirs[:] = [x for x in dirs if x not in skip]
        for f in files:
            if f.endswith(EXT):
                yield os.path.join(d, f)

ref = [(p, lines(p)) for p in walk(REF, skip=(".git",))]
rows = []
for m in walk(ROOT, skip=(".git", "gpurun_out", "obj", "bin", "__pycache__")):
    a = lines(m)
    if len(a) < 8:
        continue
    best = (0.0, "")
    for p, b in ref:
        if not b or not (0.4 < len(a) / max(len(b), 1) < 2.5):
            continue
        sm = difflib.SequenceMatcher(None, a, b, autojunk=False)
        if sm.real_quick_ratio() < best[0] or sm.quick_ratio() < best[0]:
            continue
        r = sm.ratio()
        if r > best[0]:
            best = (r, p)
    rows.append((best[0], os.path.relpath(m, ROOT), os.path.relpath(best[1], REF) if best[1] else "-"))
for r in sorted(rows, reverse=True)[:20]:
    print("%.2f  %-50s %s" % r)
