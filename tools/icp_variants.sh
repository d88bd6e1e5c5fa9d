#!/bin/bash
# usage (in the container): tools/icp_variants.sh build "T P" "T P" ...  — links one libvk_hip_icp_TxP.so per
# depth-tracker workgroup shape (VK_ICP_THREADS x VK_ICP_PIXELS, with in-kernel phase timing) into vulcan_amd/lib/
# usage (on the GPU box):     tools/icp_variants.sh run <outdir>         — tools/gn_steps.py with each of them
set -e
root=$(cd $(dirname $0)/.. && pwd)
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fvisibility=hidden"
if [ "$1" = build ]; then
  shift
  for shape in "$@"; do
    set -- $shape; t=$1; p=$2
    o=/tmp/vk_icp_${t}x${p}.o
    (cd $root/vulcan_amd/csrc && /opt/rocm/bin/hipcc $flags -DVK_LOOP_TIMING -DVK_ICP_THREADS=$t -DVK_ICP_PIXELS=$p -c vk_icp.hip -o $o)
    objs=$(ls $root/vulcan_amd/lib/obj/*.o | grep -v "vk_icp.o\|vk_probe.o")
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/vulcan_amd/lib/libvk_hip_icp_${t}x${p}.so $o $objs
    echo built libvk_hip_icp_${t}x${p}.so
  done
else
  out=${2:-gpurun_out/icp_variants}; mkdir -p $out
  for lib in $root/vulcan_amd/lib/libvk_hip_icp_*.so; do
    name=$(basename $lib .so)
    VK_HIP_LIBRARY=$lib VK_LOOP_TIMING_DUMP=1 timeout -k 10 200 python3 $root/tools/gn_steps.py > $out/$name.txt 2>&1 || echo "$name failed"
    echo "== $name"; grep "^track\|^per frame" $out/$name.txt
    grep "^step" $out/$name.txt | awk '{n++; px+=$4; pub+=$6; fl+=$8; sm+=$10; sv+=$12; tot+=$15} END {printf "   all steps (both levels) mean: pixels %.2f publish %.2f wait %.2f sum %.2f solve %.2f total %.2f us (n=%d)\n", px/n, pub/n, fl/n, sm/n, sv/n, tot/n, n}'
  done
fi
