#!/bin/bash
# tools/build_variant.sh <name> <file.hip> <-D flags ...>: links vulcan_amd/lib/libvk_hip_var_<name>.so — the
# product library with ONE source rebuilt with extra flags (A/B measurements: tools/variants.sh runs them)
set -e
root=$(cd $(dirname $0)/.. && pwd)
name=$1; src=$2; shift 2
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fvisibility=hidden"
base=$(basename $src .hip)
(cd $root/vulcan_amd/csrc && /opt/rocm/bin/hipcc $flags "$@" -c $base.hip -o /tmp/${base}_$name.o)
objs=$(ls $root/vulcan_amd/lib/obj/*.o | grep -v "/$base.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/vulcan_amd/lib/libvk_hip_var_$name.so /tmp/${base}_$name.o $objs
echo built libvk_hip_var_$name.so
