// Does hipExtAnyOrderLaunch let a kernel start beside its predecessor in the SAME stream on gfx950?
// Two kernels of 64 workgroups that spin for ~40 us each: back to back (80 us), the second with the flag, and on two streams.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void spin(unsigned long long ticks, int* out)
{
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (threadIdx.x == 0 && out) out[blockIdx.x] = 1;
}
int main()
{
  int* out; hipMalloc(&out, 4096);
  hipStream_t a, b; hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
  const unsigned long long ticks = 4000;   // 100 MHz -> 40 us
  auto run = [&](int mode) {
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 20; ++i)
    {
      hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, a, ticks, out);
      if (mode == 0) hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, a, ticks, out);
      if (mode == 1) hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, a, nullptr, nullptr, hipExtAnyOrderLaunch, ticks, out);
      if (mode == 2) hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, b, ticks, out);
    }
    hipDeviceSynchronize();
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6 / 20;
  };
  for (int rep = 0; rep < 2; ++rep)
    std::printf("pair of 40 us kernels: same stream %.1f us, second any-order %.1f us, two streams %.1f us\n", run(0), run(1), run(2));
  return 0;
}
