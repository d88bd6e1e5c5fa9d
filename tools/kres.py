#!/usr/bin/env python3
"""usage: tools/kres.py vk_trace.hip [extra hipcc flags] — per-kernel VGPR / SGPR / occupancy /
scratch / LDS as hipcc reports them (-Rpass-analysis=kernel-resource-usage, the library's flags)."""
import os
import re
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
       "-fno-fast-math", "-fno-slp-vectorize", "-fvisibility=hidden", "-Rpass-analysis=kernel-resource-usage",
       *sys.argv[2:], "-c", src, "-o", "/tmp/kres.o"]
out = subprocess.run(cmd, cwd=os.path.join(here, "..", "vulcan_amd", "csrc"), stderr=subprocess.PIPE, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    if "error" in line:
        print(line)
    m = re.search(r"remark: (?:[^:]+:\d+:\d+: )?\s*(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith(("Function Name", "Name")):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    n = subprocess.run(["c++filt", r["name"]], stdout=subprocess.PIPE, text=True).stdout.strip()
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    g = lambda k: r.get(k, "?")
    print(f"{n[:64]:64s} vgpr={g('VGPRs'):>4s} agpr={g('AGPRs'):>3s} sgpr={g('TotalSGPRs'):>4s} "
          f"scratch={g('ScratchSize [bytes/lane]'):>4s} occ={g('Occupancy [waves/SIMD]'):>2s} lds={g('LDS Size [bytes/block]')}")
