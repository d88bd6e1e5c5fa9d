#!/bin/bash
# usage: tools/variants.sh <outdir> <workload> lib1.so lib2.so ...   — bench.py --only per library build
# (VK_HIP_LIBRARY) under rocprofv3 --kernel-trace --stats; prints the top kernels of each
out=$1; wl=$2; shift 2
export TMPDIR=/tmp
mkdir -p $out
for lib in "$@"; do
  name=$(basename $lib .so)
  export VK_HIP_LIBRARY=$PWD/$lib
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -o p -- python3 bench.py --workload $wl --only --cpu-seconds 0 --steps 100 --warmup 20 > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; }
  echo "== $name: $(python3 -c "import json;d=json.load(open('$out/$name.json'));print(round(d['value']),'fps integrate_us',round(d['roofline']['avg_launch_us'],2),'raycast_us',round(d['roofline']['raycast']['avg_us'],2))")"
  python3 - <<PY
import csv
for r in list(csv.DictReader(open("$out/$name/p_kernel_stats.csv")))[:4]:
    print("   %-60s calls=%s avg_us=%.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
