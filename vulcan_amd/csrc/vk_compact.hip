// vk_compact.hip — the stream-compaction primitive of the reference as an entry point of its
// own (ref: include/vulcan/util.cuh:52-140 PrefixSum<N>, both overloads).
//
// PrefixSum gives every thread that holds `value` > 0 items the first index of a private
// range of `value` slots in a packed output, returns -1 to threads with nothing, and adds
// the workgroup's total to a global counter with one atomicAdd (so the order of the
// workgroups' ranges is whatever order the atomics land in). It is a 2 * log2(512)-barrier
// up/down-sweep over an LDS array. Here: a 64-lane inclusive scan by six DPP-free shuffles,
// one LDS hop across the waves of a workgroup, and the workgroup bases from a second,
// single-workgroup pass over the per-workgroup totals instead of an atomic — so the ranges
// are in input order (a stable compaction: one of the outcomes the reference allows, and the
// same one every time). No inter-workgroup hand-off inside a launch.
//
// The kernels of the hot path that compact (update_visibility_kernel, compute_patches_kernel,
// the detector) carry the same wave scan inline; this entry point exists so that the
// primitive can be called and tested by itself (tests/util_test.cu:63-98).
#include "vk_common.hpp"

using namespace vk;

namespace
{

constexpr int kCompactThreads = 1024;

__device__ __forceinline__ int wave_inclusive_scan(int v)
{
  const int lane = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1)
  {
    const int t = __shfl_up(v, d);
    if (lane >= d) v += t;
  }
  return v;
}

// inclusive scan over the workgroup; returns the value for this thread, *total for all
__device__ __forceinline__ int group_inclusive_scan(int v, int* wave_totals /* kCompactThreads / 64 */, int& total)
{
  const int wave = threadIdx.x >> 6;
  const int incl = wave_inclusive_scan(v);
  if (lane_id() == 63) wave_totals[wave] = incl;
  __syncthreads();
  int before = 0;
  total = 0;
  for (int w = 0; w < kCompactThreads / 64; ++w)
  {
    if (w < wave) before += wave_totals[w];
    total += wave_totals[w];
  }
  __syncthreads();
  return before + incl;
}

__global__ __launch_bounds__(kCompactThreads) void group_totals_kernel(const int32_t* __restrict__ counts, int n,
    int32_t* __restrict__ group_totals)
{
  __shared__ int wave_totals[kCompactThreads / 64];
  const int i = blockIdx.x * kCompactThreads + (int)threadIdx.x;
  int total;
  group_inclusive_scan(i < n ? counts[i] : 0, wave_totals, total);
  if (threadIdx.x == 0) group_totals[blockIdx.x] = total;
}

// exclusive scan of the workgroup totals in place (one workgroup); adds the grand total to *total
__global__ __launch_bounds__(kCompactThreads) void scan_totals_kernel(int32_t* __restrict__ group_totals, int groups,
    int32_t* __restrict__ total)
{
  __shared__ int wave_totals[kCompactThreads / 64];
  __shared__ int running;
  if (threadIdx.x == 0) running = *total;       // the reference accumulates into `total` (util.cuh:88-91)
  __syncthreads();
  for (int base = 0; base < groups; base += kCompactThreads)
  {
    const int i = base + (int)threadIdx.x;
    const int v = i < groups ? group_totals[i] : 0;
    int chunk;
    const int incl = group_inclusive_scan(v, wave_totals, chunk);
    if (i < groups) group_totals[i] = running + incl - v;
    __syncthreads();
    if (threadIdx.x == 0) running += chunk;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = running;
}

__global__ __launch_bounds__(kCompactThreads) void offsets_kernel(const int32_t* __restrict__ counts, int n,
    const int32_t* __restrict__ group_bases, int32_t* __restrict__ offsets)
{
  __shared__ int wave_totals[kCompactThreads / 64];
  const int i = blockIdx.x * kCompactThreads + (int)threadIdx.x;
  const int v = i < n ? counts[i] : 0;
  int total;
  const int incl = group_inclusive_scan(v, wave_totals, total);
  // util.cuh:93-94: threads with nothing get -1
  if (i < n) offsets[i] = v > 0 ? group_bases[blockIdx.x] + incl - v : -1;
}

}  // namespace

extern "C" {

size_t vk_compact_workspace_bytes(int32_t count)
{
  if (count <= 0) return 0;
  return sizeof(int32_t) * (size_t)((count + kCompactThreads - 1) / kCompactThreads);
}

int vk_compact_offsets(const int32_t* counts, int32_t count, int32_t* offsets, int32_t* total_dev, void* workspace,
    void* stream)
{
  VK_REQUIRE(count >= 0);
  if (count == 0) return VK_OK;
  VK_REQUIRE(counts && offsets && total_dev && workspace);
  hipStream_t s = vk_s(stream);
  const int groups = (count + kCompactThreads - 1) / kCompactThreads;
  int32_t* group_totals = static_cast<int32_t*>(workspace);
  hipLaunchKernelGGL(group_totals_kernel, dim3(groups), dim3(kCompactThreads), 0, s, counts, count, group_totals);
  VK_LAUNCH_CHECK();
  hipLaunchKernelGGL(scan_totals_kernel, dim3(1), dim3(kCompactThreads), 0, s, group_totals, groups, total_dev);
  VK_LAUNCH_CHECK();
  hipLaunchKernelGGL(offsets_kernel, dim3(groups), dim3(kCompactThreads), 0, s, counts, count, group_totals, offsets);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

}  // extern "C"
