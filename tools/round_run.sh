# The round's standard GPU pass (run through gpurun from the repo root):
#   bash tools/round_run.sh <tag> [tests|notests]
# tests -> demo loop (3 modes) -> bench.py -> rocprofv3 kernel traces of the three workloads.
# Every step is joined with &&-semantics (set -e): a failing or hanging GPU step ends the pass.
set -eo pipefail
tag=${1:-r03_x}
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
if [ "${2:-tests}" = "tests" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1 || { tail -30 $out/pytest_gpu.txt; exit 1; }
  tail -2 $out/pytest_gpu.txt
fi
: > $out/fuse_sequence.txt
for m in 0 1 2 3; do timeout -k 10 120 vulcan_amd/host/bin/fuse_sequence 300 $m >> $out/fuse_sequence.txt 2>&1; done
# mode 0 with every raycast announcing the next frame (Tracer::Trace(keyframe, next_frame)): the class layer's form of bench.py's step
timeout -k 10 120 vulcan_amd/host/bin/fuse_sequence 300 0 0 0 1 >> $out/fuse_sequence.txt 2>&1
# the photometric loops once more over half a cycle of the camera's path: what they allocate then fits the app's pool
for m in 2 3; do timeout -k 10 120 vulcan_amd/host/bin/fuse_sequence 120 $m >> $out/fuse_sequence.txt 2>&1; done
grep "^frames\|^steady" $out/fuse_sequence.txt
timeout -k 10 400 python bench.py > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
python3 -c "
import json; d=json.load(open('$out/bench.json')); o=d['other_workloads']
print('rgbd', round(d['value']), 'us', round(1e3*d['ms_per_step'],1), 'frac', round(d['roofline']['frac'],3), 'past_l3', round(d['roofline'].get('frac_past_l3',0),3), '| depth', round(o['depth']['value']), '| icp', round(o['rgbd-icp']['value']), o['rgbd-icp'].get('pose_error_max'), 'gn', o['rgbd-icp'].get('gn_steps_median'))"
for wl in rgbd depth rgbd-icp; do
  n=$(echo $wl | tr - _)
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$n -o p -- python3 bench.py --workload $wl --only --steps 100 --warmup 20 --cpu-seconds 0 > $out/bench_${n}_under_rocprof.json 2> $out/prof_$n.err
  echo "== $wl"; python3 - <<PY
import csv,glob,shutil
f=glob.glob("$out/prof_$n/**/*kernel_stats.csv",recursive=True)[0]
shutil.copy(f, "$out/${n}_kernel_stats.csv")
for r in list(csv.DictReader(open(f)))[:10]: print("  ", r["Name"].replace("(anonymous namespace)::","")[:70].ljust(70), r["Calls"], round(float(r["AverageNs"])/1e3,2))
PY
  rm -rf $out/prof_$n
done
