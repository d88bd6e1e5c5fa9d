// What a dependency between two streams costs on gfx950: kernel A (40 us) on stream a, then kernel B (40 us) on
// stream b that must follow A — through an event (hipEventRecord + hipStreamWaitEvent, three kinds of event), through
// hipStreamWriteValue32 / hipStreamWaitValue32 on signal memory, and through a flag in device memory that B's waves
// poll themselves. Reported: time of the pair minus 80 us.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin(unsigned long long ticks, int* out, volatile int* wait_flag, int wait_for, int* set_flag, int set_to)
{
  if (wait_flag) { while (__hip_atomic_load((int*)wait_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < wait_for) __builtin_amdgcn_s_sleep(2); }
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (threadIdx.x == 0 && out) out[blockIdx.x] = 1;
  if (set_flag && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(set_flag, set_to, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
int main()
{
  int* out; hipMalloc(&out, 4096);
  int* flag; hipMalloc(&flag, 64); hipMemset(flag, 0, 64);
  uint32_t* sig = nullptr;
  const hipError_t se = hipExtMallocWithFlags((void**)&sig, 64, hipMallocSignalMemory);
  int can = 0; hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
  hipStream_t a, b; hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
  hipEvent_t e[3];
  hipEventCreateWithFlags(&e[0], hipEventDefault);
  hipEventCreateWithFlags(&e[1], hipEventDisableTiming);
  hipEventCreateWithFlags(&e[2], hipEventDisableTiming | hipEventDisableSystemFence);
  const unsigned long long ticks = 4000;
  int epoch = 0;
  auto run = [&](int mode) {
    hipDeviceSynchronize();
    const int reps = 20;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i)
    {
      ++epoch;
      if (mode <= 2)
      {
        hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, a, ticks, out, nullptr, 0, nullptr, 0);
        hipEventRecord(e[mode], a);
        hipStreamWaitEvent(b, e[mode], 0);
        hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, b, ticks, out, nullptr, 0, nullptr, 0);
      }
      else if (mode == 3)
      {
        hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, a, ticks, out, nullptr, 0, nullptr, 0);
        hipStreamWriteValue32(a, sig, (uint32_t)epoch, 0);
        hipStreamWaitValue32(b, sig, (uint32_t)epoch, hipStreamWaitValueGte, 0xffffffffu);
        hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, b, ticks, out, nullptr, 0, nullptr, 0);
      }
      else if (mode == 4)
      {
        // A sets a flag with its last instructions; B is launched at once on the other stream and its waves poll the flag
        hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, a, ticks, out, nullptr, 0, flag, epoch);
        hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, b, ticks, out, flag, epoch, nullptr, 0);
      }
      else
      {
        hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, a, ticks, out, nullptr, 0, nullptr, 0);
        hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, a, ticks, out, nullptr, 0, nullptr, 0);
      }
      // the next pair must follow this one: b -> a
      hipEventRecord(e[0], b); hipStreamWaitEvent(a, e[0], 0);
    }
    hipDeviceSynchronize();
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6 / reps;
  };
  std::printf("signal memory: %s, hipDeviceAttributeCanUseStreamWaitValue = %d\n", hipGetErrorString(se), can);
  const char* names[] = {"event (default)", "event (no timing)", "event (no timing, no system fence)", "stream write / wait value", "flag polled by the waves", "same stream"};
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 6; ++mode)
    {
      if (mode == 3 && (se != hipSuccess || !can)) continue;
      std::printf("%-36s %.1f us per pair\n", names[mode], run(mode));
    }
  return 0;
}
