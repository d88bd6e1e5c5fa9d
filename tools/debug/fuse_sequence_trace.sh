#!/bin/bash
# which launches the C++ class layer's loop makes per frame (rocprofv3 kernel stats of fuse_sequence <args>)
export TMPDIR=/tmp
out=gpurun_out/fs_trace; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- vulcan_amd/host/bin/fuse_sequence "$@" > $out/run.txt 2>&1
grep "^frames" $out/run.txt
python3 - <<PY
import csv,glob
f=glob.glob("$out/**/p_kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print("   %-70s calls=%s avg_us=%.2f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
