#!/bin/bash
# the fuse_sequence runs of tools/round_run.sh alone (after a change to the app): $1 = output file
out=${1:-gpurun_out/fuse_sequence.txt}; : > $out
for m in 0 1 2 3; do timeout -k 10 120 vulcan_amd/host/bin/fuse_sequence 300 $m >> $out 2>&1; done
timeout -k 10 120 vulcan_amd/host/bin/fuse_sequence 300 0 0 0 1 >> $out 2>&1
for m in 2 3; do timeout -k 10 120 vulcan_amd/host/bin/fuse_sequence 120 $m >> $out 2>&1; done
grep "^frames\|^steady" $out
