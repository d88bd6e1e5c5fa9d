set -o pipefail
out=gpurun_out/r02_e; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1; tail -2 $out/pytest_gpu.txt
timeout -k 10 400 python bench.py > $out/bench.json 2> $out/bench.err; python3 -c "
import json; d=json.load(open('$out/bench.json')); print('rgbd', round(d['value']), 'frac', round(d['roofline']['frac'],3), 'depth', round(d['other_workloads']['depth']['value']), 'icp', round(d['other_workloads']['rgbd-icp']['value']))"
for wl in rgbd depth rgbd-icp; do
  n=$(echo $wl | tr - _)
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$n -o p -- python3 bench.py --workload $wl --only --steps 100 --warmup 20 --cpu-seconds 0 > $out/bench_${n}_under_rocprof.json 2> $out/prof_$n.err
  echo "== $wl"; python3 - <<PY
import csv,glob
f=glob.glob("$out/prof_$n/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:9]: print("  ", r["Name"].replace("(anonymous namespace)::","")[:70].ljust(70), r["Calls"], round(float(r["AverageNs"])/1e3,2))
PY
done
for m in 0 1 2; do vulcan_amd/host/bin/fuse_sequence 300 $m | tail -2; done > $out/fuse_sequence.txt 2>&1; grep frames $out/fuse_sequence.txt
timeout -k 10 120 python tools/icp_bench.py > $out/tracker_steps.txt 2>&1; tail -3 $out/tracker_steps.txt
timeout -k 10 120 python tools/gn_steps.py > $out/gn_steps.txt 2>&1; tail -3 $out/gn_steps.txt
