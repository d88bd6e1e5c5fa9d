# The round's measurement passes that tools/round_run.sh does not make (run through gpurun from the repo root, after it):
#   bash tools/round_profiles.sh <tag>
# the driver's own command line, the PMC traffic passes (FETCH_SIZE / WRITE_SIZE, one pass each), the past-L3 integrate
# launches under the kernel trace, the two-rank rehearsal over gloo on one GPU, the switches' parity.
tag=${1:-r06_h}
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_like.json 2> $out/bench_driver_like.err; echo "driver-like bench rc $?"
python3 -c "
import json; d=json.load(open('$out/bench_driver_like.json')); print('value', round(d['value']), 'windows', {k: round(v) for k, v in d['windows'].items() if k in ('median','min','max')}, 'frac', round(d['roofline']['frac'],3), 'frac_hbm', round(d['roofline']['frac_hbm'],3), 'error', d.get('error'))"
for wl in rgbd depth; do
  bash tools/traffic.sh $out/traffic_$wl $wl > $out/traffic_$wl.log 2>&1; echo "traffic $wl rc $?"
done
for wl in rgbd depth; do
  rm -rf $out/pl3_$wl; mkdir -p $out/pl3_$wl
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/pl3_$wl -o p -- python3 tools/past_l3_profile.py $wl > $out/pl3_$wl/run.json 2> $out/pl3_$wl.err; echo "past-L3 $wl rc $?"
  python3 tools/past_l3_profile.py --parse $out/pl3_$wl > $out/past_l3_$wl.json; cat $out/past_l3_$wl.json
done
VK_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 5 > $out/two_rank_gloo_rehearsal.json 2> $out/two_rank.err; echo "two-rank rehearsal rc $?"
bash tools/switch_parity.sh run $out/switch_parity
bash tools/epoch_wrap_proof.sh run $out/epoch_wrap
