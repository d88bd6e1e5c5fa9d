mkdir -p gpurun_out/r06_i
L=$PWD/vulcan_amd/lib
VK_HIP_LIBRARY=$L/libvk_hip_var_a4w6n6.so timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_round5.py tests/test_gpu_fuzz.py tests/test_gpu_closed_loop.py -m gpu -q > gpurun_out/r06_i/parity_a4w6n6.txt 2>&1; echo "parity a4w6n6 rc $?: $(tail -1 gpurun_out/r06_i/parity_a4w6n6.txt)"
for lib in libvk_hip.so libvk_hip_var_a4.so libvk_hip_var_a4w6.so libvk_hip_var_a4w6n6.so libvk_hip_var_a4w6n24.so libvk_hip.so; do
  VK_HIP_LIBRARY=$L/$lib timeout -k 10 120 python bench.py --workload rgbd-icp --only --steps 300 --warmup 20 --cpu-seconds 0 > gpurun_out/r06_i/icp_$lib.json 2>gpurun_out/r06_i/err.txt
  VK_HIP_LIBRARY=$L/$lib timeout -k 10 120 python bench.py --workload rgbd --only --steps 200 --warmup 20 --cpu-seconds 0 > gpurun_out/r06_i/rgbd_$lib.json 2>>gpurun_out/r06_i/err.txt
  echo "$lib: $(python3 -c "
import json
a=json.load(open('gpurun_out/r06_i/icp_$lib.json')); b=json.load(open('gpurun_out/r06_i/rgbd_$lib.json'))
print('icp us/frame', round(a['ms_per_step']*1e3,1), 'raycast', round(a['roofline']['raycast']['avg_us'],1), '| rgbd us/frame', round(b['ms_per_step']*1e3,1), 'raycast+req', round(b['roofline']['raycast']['avg_us'],1))")"
done
