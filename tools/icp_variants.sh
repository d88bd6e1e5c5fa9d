#!/bin/bash
# Development aid for the loop kernels of the trackers.
# in the container:  tools/icp_variants.sh build "<name> <-D flags>" ...   — links one libvk_hip_var_<name>.so per
#                    flag set (with in-kernel phase timing, -DVK_LOOP_TIMING) into vulcan_amd/lib/
# on the GPU box:    tools/icp_variants.sh run <outdir> [depth|light]      — tools/gn_steps.py (depth tracker) or
#                    fuse_sequence 300 2 (light tracker) with each of them, phase means per step
set -e
root=$(cd $(dirname $0)/.. && pwd)
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fvisibility=hidden"
if [ "$1" = build ]; then
  shift
  for spec in "$@"; do
    set -- $spec; name=$1; shift
    for f in vk_icp vk_color_tracker; do
      (cd $root/vulcan_amd/csrc && /opt/rocm/bin/hipcc $flags -DVK_LOOP_TIMING "$@" -c $f.hip -o /tmp/${f}_$name.o)
    done
    objs=$(ls $root/vulcan_amd/lib/obj/*.o | grep -v "vk_icp.o\|vk_color_tracker.o\|vk_probe.o")
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/vulcan_amd/lib/libvk_hip_var_$name.so /tmp/vk_icp_$name.o /tmp/vk_color_tracker_$name.o $objs
    echo built libvk_hip_var_$name.so
  done
else
  out=${2:-gpurun_out/icp_variants}; what=${3:-depth}; mkdir -p $out
  for lib in $root/vulcan_amd/lib/libvk_hip_var_*.so; do
    name=$(basename $lib .so)
    if [ $what = depth ]; then
      VK_HIP_LIBRARY=$lib VK_LOOP_TIMING_DUMP=1 timeout -k 10 200 python3 $root/tools/gn_steps.py > $out/$name.txt 2>&1 || echo "$name failed"
      echo "== $name"; grep "^track\|^per frame" $out/$name.txt
    else
      # the host layer links libvk_hip.so by name: put the variant first on the loader's path under that name
      mkdir -p /tmp/var_$name && cp $lib /tmp/var_$name/libvk_hip.so
      LD_LIBRARY_PATH=/tmp/var_$name:$LD_LIBRARY_PATH timeout -k 10 100 $root/vulcan_amd/host/bin/fuse_sequence 300 2 | tail -2 | head -1 > $out/$name.fps
      LD_LIBRARY_PATH=/tmp/var_$name:$LD_LIBRARY_PATH VK_LOOP_TIMING_DUMP=1 timeout -k 10 100 $root/vulcan_amd/host/bin/fuse_sequence 60 2 > $out/$name.txt 2>&1 || echo "$name failed"
      echo "== $name: $(cat $out/$name.fps)"
    fi
    grep "^step" $out/$name.txt | awk '{n++; px+=$4; pub+=$6; sm+=$10; sv+=$12; tot+=$15} END {printf "   all steps (both levels) mean: pixels %.2f publish %.2f wait+sum %.2f solve %.2f total %.2f us (n=%d)\n", px/n, pub/n, sm/n, sv/n, tot/n, n}'
  done
fi
