# bash tools/kernel_avg.sh <tag> <workload> [pattern]: rocprofv3 kernel averages of one bench workload (development aid)
set -eo pipefail
tag=${1:-x}; wl=${2:-rgbd}; pat=${3:-.}
out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_k -o p -- python3 bench.py --workload $wl --only --steps 100 --warmup 20 --cpu-seconds 0 > $out/bench_k.json 2> $out/prof_k.err
python3 - <<PY
import csv,glob,json
f=glob.glob("$out/prof_k/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:9]:
    print("  ", r["Name"].replace("(anonymous namespace)::","")[:64].ljust(64), r["Calls"], "avg", round(float(r["AverageNs"])/1e3,2), "min", round(float(r["MinNs"])/1e3,2))
# medians from the trace (an average hides the first frame, which allocates everything at once)
import statistics
t=glob.glob("$out/prof_k/**/*kernel_trace.csv",recursive=True)
if t:
    by={}
    for r in csv.DictReader(open(t[0])):
        by.setdefault(r["Kernel_Name"],[]).append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
    for name,v in sorted(by.items(), key=lambda kv:-sum(kv[1]))[:7]:
        v.sort()
        print("   median", name.replace("(anonymous namespace)::","")[:56].ljust(56), round(statistics.median(v),2), "p90", round(v[int(0.9*len(v))],2), "max", round(v[-1],1))
d=json.load(open("$out/bench_k.json")); print("fps", round(d["value"]), "us", round(1e3*d["ms_per_step"],1))
PY
rm -rf $out/prof_k
