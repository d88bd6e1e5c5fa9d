#!/bin/bash
# The experiments that lost stay in the tree behind compile-time switches (docs/rounds/r05.md). This keeps them honest:
#   in the container:  tools/switch_parity.sh build          — one libvk_hip_var_sw_<name>.so per switch
#   on the GPU box:    tools/switch_parity.sh run <outdir>   — the parity tests that cover the switched code, per build
# A switch whose build no longer passes is a bug in the record, not a variant to be quoted.
set -e
root=$(cd $(dirname $0)/.. && pwd)
# name | source | flags | tests
table="
light_packed|vk_integrate.hip|-DVK_LIGHT_PACKED=1|tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_weights.py tests/test_gpu_fuzz.py tests/test_gpu_round5.py
integrate_ring|vk_integrate.hip|-DVK_INTEGRATE_RING=1|tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_weights.py tests/test_gpu_fuzz.py tests/test_gpu_round5.py tests/test_gpu_banded_lists.py
handle_fenced|vk_volume.hip|-DVK_HANDLE_ALWAYS_FENCED=1|tests/test_gpu_parity.py tests/test_gpu_set_view_rounds.py tests/test_gpu_banded_lists.py tests/test_gpu_edge_cases.py tests/test_gpu_round5.py
atomic_exchange|vk_icp.hip|-DVK_LOOP_ATOMIC_EXCHANGE|tests/test_gpu_closed_loop.py tests/test_gpu_loop_abort.py
march_ahead|vk_trace.hip|-DVK_MARCH_AHEAD=4 -DVK_POINTS_WAVES_PER_EU=6 -DVK_MARCH_AHEAD_AFTER=6|tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_round5.py tests/test_gpu_fuzz.py tests/test_gpu_closed_loop.py
"
if [ "$1" = build ]; then
  echo "$table" | while IFS='|' read name src flags tests; do
    [ -z "$name" ] && continue
    bash $root/tools/build_variant.sh sw_$name $root/vulcan_amd/csrc/$src $flags
  done
else
  out=${2:-gpurun_out/switch_parity}; mkdir -p $out
  echo "$table" | while IFS='|' read name src flags tests; do
    [ -z "$name" ] && continue
    lib=$root/vulcan_amd/lib/libvk_hip_var_sw_$name.so
    [ -f $lib ] || { echo "== $name: not built"; continue; }
    if VK_HIP_LIBRARY=$lib timeout -k 10 600 python -m pytest $tests -m gpu -x -q > $out/$name.txt 2>&1; then verdict=ok; else verdict=FAILED; fi
    echo "== $name ($src $flags): $verdict: $(tail -1 $out/$name.txt)"
  done
fi
