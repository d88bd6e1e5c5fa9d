#!/bin/bash
# The proof that tests/test_gpu_epoch_wrap.py bites:
#   in the container:  tools/epoch_wrap_proof.sh build        — libvk_hip_var_epoch_unguarded.so: the product with
#                                                              -DVK_LOOP_EPOCH_UNGUARDED (vk_runtime.hip: no area is ever cleared)
#   on the GPU box:    tools/epoch_wrap_proof.sh run <outdir> — the wrap tests on the product (must pass) and on that build
#                                                              (the adversarial cases must FAIL: stale words with the
#                                                              launch's own tag are taken for its sums)
root=$(cd $(dirname $0)/.. && pwd)
if [ "$1" = build ]; then
  exec bash $root/tools/build_variant.sh epoch_unguarded $root/vulcan_amd/csrc/vk_runtime.hip -DVK_LOOP_EPOCH_UNGUARDED
fi
out=${2:-gpurun_out/epoch_wrap}; mkdir -p $out
cd $root
timeout -k 10 300 python -m pytest tests/test_gpu_epoch_wrap.py -m gpu -q > $out/guarded.txt 2>&1
echo "product: rc $? : $(tail -1 $out/guarded.txt)"
VK_HIP_LIBRARY=$root/vulcan_amd/lib/libvk_hip_var_epoch_unguarded.so timeout -k 10 300 python -m pytest tests/test_gpu_epoch_wrap.py -m gpu -q > $out/unguarded.txt 2>&1
echo "-DVK_LOOP_EPOCH_UNGUARDED: rc $? : $(tail -1 $out/unguarded.txt)"
grep -E "^(FAILED|PASSED|ERROR)|passed|failed" $out/unguarded.txt | tail -8
