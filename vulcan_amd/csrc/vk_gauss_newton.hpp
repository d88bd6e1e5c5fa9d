// vk_gauss_newton.hpp — what the trackers share: the two-stage reduction of the
// 27 sums of a 6-parameter normal system and the small dense algebra of a
// Gauss-Newton step, all of it sized at compile time so that it stays in
// registers (ref: src/tracker.cpp:124-163 and the ComputeSystemKernel of
// src/depth_tracker.cu:144-268 / src/color_tracker.cu:206-343).
#pragma once

#include "vk_common.hpp"

#include <cstdlib>
#include <mutex>

#ifdef VK_LOOP_TIMING
#include <cstdio>
#include <cstdlib>
#include <cstring>
#endif

namespace vk
{

constexpr int kSysStride = 32;     // floats per workgroup partial: 6 gradient + 21 hessian + pad

// Wave64 sum without LDS: four DPP steps fold each row of 16 lanes (two quad
// permutes, half-mirror, mirror), row_bcast:15 / row_bcast:31 carry the row
// totals upwards; the wave's sum ends in lane 63. Six v_add_f32 with a DPP
// operand per value, against six ds_bpermute round trips for __shfl_xor.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_term(float v)
{
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}

__device__ __forceinline__ float wave_sum_lane63(float v)
{
  v += dpp_term<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
  v += dpp_term<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
  v += dpp_term<0x141, 0xf>(v);   // row_half_mirror
  v += dpp_term<0x140, 0xf>(v);   // row_mirror
  v += dpp_term<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
  v += dpp_term<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3
  return v;
}

// One pixel's contribution: slots [0,6) J^T r, [6,27) the packed lower triangle of
// J^T J in (r, c<=r) row-major order (depth_tracker.cu:163-165,208-214).
__device__ __forceinline__ void outer_products(const float (&J)[6], float r, float (&acc)[27])
{
#pragma unroll
  for (int i = 0; i < 6; ++i) acc[i] = J[i] * r;
  int counter = 6;
#pragma unroll
  for (int rr = 0; rr < 6; ++rr)
#pragma unroll
    for (int c = 0; c <= rr; ++c, ++counter) acc[counter] = J[rr] * J[c];
}

// First stage: the workgroup's 27 sums -> workspace[blockIdx.x]. Called by every
// thread of a WAVES * 64 wide workgroup.
template <int WAVES>
__device__ __forceinline__ void store_partial(const float (&acc)[27], float (*lds)[kSysStride], float* workspace)
{
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 27; ++i)
  {
    const float v = wave_sum_lane63(acc[i]);
    if (lane == 63) lds[wave][i] = v;
  }
  __syncthreads();

  if (threadIdx.x < kSysStride)
  {
    float v = 0.0f;
    if (threadIdx.x < 27)
    {
#pragma unroll
      for (int w = 0; w < WAVES; ++w) v += lds[w][threadIdx.x];
    }
    workspace[(size_t)blockIdx.x * kSysStride + threadIdx.x] = v;
  }
}

// number of interleaved slices of the second stage (see sum_partials)
constexpr int kSysSlices = 32;

// The slices added in order into hessian[36] / gradient[6] (either may be null) and sums[48]
// (LDS). Every thread of the workgroup must call it, after a barrier that follows the
// writes of `slices`; ends with a barrier.
__device__ __forceinline__ void finish_sums(float (*slices)[kSysStride], int translation_enabled,
    float* hessian, float* gradient, float* sums)
{
  if (threadIdx.x < 36 + 6)
  {
    if (threadIdx.x < 6)
    {
      float g = 0.0f;
      const int n = translation_enabled ? 6 : 3;
      if ((int)threadIdx.x < n)
        for (int k = 0; k < kSysSlices; ++k) g += slices[k][threadIdx.x];
      if (gradient) gradient[threadIdx.x] = g;
      sums[36 + threadIdx.x] = g;
    }
    else
    {
      // depth_tracker.cu:199-214: with translation disabled the packed triangle
      // is that of the 3x3 rotation block (6 values)
      const int out = threadIdx.x - 6;
      const int n = translation_enabled ? 21 : 6;
      float h = 0.0f;
      if (out < n)
        for (int k = 0; k < kSysSlices; ++k) h += slices[k][6 + out];
      if (hessian) hessian[out] = h;
      sums[out] = h;
    }
  }
  __syncthreads();
}

// Second stage: fixed-order sum of the per-workgroup partials into hessian[36] (packed
// lower triangle first, rest 0) and gradient[6] (either may be null), and into sums[48]
// in LDS for a solve that follows. Slice s of kSysSlices adds partials s, s + 32, s + 64,
// ... in that order, then the slices are added in order: the result depends on the
// partials alone, not on how many threads call (256 or 1024: both are used), so every
// caller gets the same bits. Every thread of the workgroup must call it; ends with a
// barrier. The loads of a slice are issued sixteen at a time: with one L2 round trip per
// term this was the longest part of a Gauss-Newton step (a 640x480 image has 300
// partials, 1280x960 has 1200).
__device__ __forceinline__ void sum_partials(const float* workspace, int partials, int translation_enabled,
    float* hessian, float* gradient, float (*slices)[kSysStride], float* sums)
{
  const int c = threadIdx.x & 31;
  for (int s = threadIdx.x >> 5; s < kSysSlices; s += blockDim.x >> 5)
  {
    float v = 0.0f;
    int j = s;
    for (; j + 15 * kSysSlices < partials; j += 16 * kSysSlices)
    {
      float t[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) t[u] = workspace[(size_t)(j + kSysSlices * u) * kSysStride + c];
#pragma unroll
      for (int u = 0; u < 16; ++u) v += t[u];
    }
    {
      // the tail, still issued together: terms past the end read partial 0 and add 0
      float t[16];
#pragma unroll
      for (int u = 0; u < 16; ++u)
      {
        const int k = j + kSysSlices * u;
        t[u] = workspace[(size_t)(k < partials ? k : 0) * kSysStride + c];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) v += (j + kSysSlices * u < partials) ? t[u] : 0.0f;
    }
    slices[s][c] = v;
  }
  __syncthreads();
  finish_sums(slices, translation_enabled, hessian, gradient, sums);
}

// Development aid (-DVK_LOOP_TIMING): workgroup 0 of a loop kernel stamps the wall clock at six
// points of every step; with VK_LOOP_TIMING_DUMP set the host prints the previous launch's
// stamps before the next launch. Compiled out of the product.
#ifdef VK_LOOP_TIMING
#define VK_STAMP(k) if (blockIdx.x == 0 && threadIdx.x == 0 && L.timing) L.timing[it * 8 + (k)] = wall_clock64()
inline unsigned long long* loop_timing_attach(hipStream_t s)
{
  static unsigned long long* timing = nullptr;
  if (!timing) { (void)hipHostMalloc((void**)&timing, 64 * 8 * 8, hipHostMallocMapped); memset(timing, 0, 64 * 8 * 8); }
  if (getenv("VK_LOOP_TIMING_DUMP"))
  {
    (void)hipStreamSynchronize(s);
    for (int k = 0; k < 32 && timing[k * 8]; ++k)
      fprintf(stderr, "step %2d: pixels %5.2f publish %5.2f flags %5.2f sum %5.2f solve %5.2f | total %5.2f us\n", k,
          (timing[k * 8 + 1] - timing[k * 8 + 0]) / 100.0, (timing[k * 8 + 2] - timing[k * 8 + 1]) / 100.0,
          (timing[k * 8 + 3] - timing[k * 8 + 2]) / 100.0, (timing[k * 8 + 4] - timing[k * 8 + 3]) / 100.0,
          (timing[k * 8 + 5] - timing[k * 8 + 4]) / 100.0, (timing[k * 8 + 5] - timing[k * 8 + 0]) / 100.0);
    memset(timing, 0, 64 * 8 * 8);
  }
  return timing;
}
#define VK_LOOP_TIMING_ATTACH(L, s) (L).timing = loop_timing_attach(s)
#define VK_LOOP_TIMING_FIELD unsigned long long* timing;
#else
#define VK_STAMP(k)
#define VK_LOOP_TIMING_ATTACH(L, s)
#define VK_LOOP_TIMING_FIELD
#endif

// ---- partials exchanged INSIDE a launch (the whole Gauss-Newton loop as one kernel) ----
//
// A step of the loop is: every workgroup evaluates its pixels at the current pose, all
// workgroups' sums are added, the 6x6 system is solved, the pose moves. With one launch per
// step the kernel boundary is the exchange, and on this part a kernel boundary costs
// ~4.5 us plus cold caches for the images every step. Here the workgroups of ONE launch
// exchange their sums through memory and every workgroup then adds all of them and solves,
// redundantly and identically — so after the exchange nobody waits for anybody.
//
// The exchange uses no fences (an agent-scope release / acquire writes back and
// invalidates a whole L2 per workgroup on this multi-XCD part: measured 2.5x slower than
// ending the kernel, r01) and no atomic read-modify-write (hundreds of workgroups on one
// counter serialise). Each of a workgroup's 27 sums travels as ONE 64-bit word
// {tag, value}, written and read with relaxed agent-scope atomics — which are coherent
// across the XCDs by themselves (they bypass the non-coherent L2 path). A reader takes a
// value only if its tag names this launch and step, and otherwise asks again; value and
// tag cannot be torn apart because they are one atomic object, and no second round trip
// ("data, wait, then flag") is needed: the chain per step is one store and one load
// across the fabric (~1 us each on this part; measured alternative with separate flags
// and L2-cached slots: one fabric trip more per step, 1 us slower). Nothing is ever
// invalidated, so the images stay in L2 across steps.
//
// Tag = epoch (22 bits, never 0) << 10 | step + 1. Slots are double buffered by step
// parity: a workgroup writes step i + 2 only after it has read every workgroup's step
// i + 1, which they wrote after reading all of step i. The epoch names the launch among
// the last 2^22 - 2 of the process, and no word older than 2^21 launches is ever left in
// an area (vk_loop_epoch_begin, vk_runtime.hip, clears it first): what a reader of
// {epoch, step} can find in a slot is cleared memory, a word of a younger launch with
// another epoch, an earlier step of its own launch, or the word it waits for — the
// counter's wrap (every ~10 minutes of tracking) is not an event
// (tests/test_gpu_epoch_wrap.py crosses it with a tracker that sat idle for a whole period).
//
// Every workgroup of the launch must be resident at the same time (the host sizes the grid
// from the occupancy query); should that ever not hold — or a workgroup die — the readers
// give up after kExchangeTimeout and the launch ends with state[1] = VK_TRACK_ABORTED
// instead of spinning for ever.
constexpr unsigned long long kExchangeTimeout = 200000000ull;   // wall_clock64 ticks (100 MHz): 2 s
#ifndef VK_POLL_GAP
#define VK_POLL_GAP 2        // s_sleep units of 64 clocks between two rounds of questions
#endif
constexpr int kExchangeSteps = 1023;                              // ten tag bits name the step

__device__ __forceinline__ uint32_t exchange_tag(uint32_t epoch, int step) { return (epoch << 10) | (uint32_t)(step + 1); }

// floats of an exchange area for `count` workgroups: two parities x count x kSysStride words of 64 bits
inline size_t exchange_floats(int count) { return 4 * (size_t)count * kSysStride; }

struct Exchange
{
  unsigned long long* words;   // [2][count][kSysStride]
  int count;                   // pixel groups of the image = slots per parity (a workgroup fills one or more)
  uint32_t epoch;
};

}  // namespace vk

// The tag of the next loop launch on the exchange area `area` (vk_runtime.hip: a process-wide count of loop
// launches, 22 bits of it, never 0), and — enqueued on `s` in front of that launch — the area's clear when it is due
// (first use, grown, written by a launch-per-stage loop, or 2^21 launches old). VK_OK or the runtime's error.
int vk_loop_epoch_begin(void* area, size_t bytes, hipStream_t s, uint32_t* epoch);
// something other than a loop launch is about to write `area` (the launch-per-stage loops' float partials)
void vk_loop_area_written(const void* area);

// test aid (vk_test_hooks.force_loop_abort): the next loop launches behave as if a workgroup's sums had
// never arrived — they end with VK_TRACK_ABORTED at once — so that the hosts' way out of an
// aborted Track (the launch-per-stage loop) can be tested without starving a real launch
inline int vk_forced_loop_abort()
{
  return vk_hook(VK_HOOK_FORCE_LOOP_ABORT) == 1 ? 1 : 0;
}

// vk_test_hooks.loop_cooperative: loop kernels go through hipLaunchCooperativeKernel, which refuses a grid
// that cannot be resident at once instead of letting the kernel find out (measured: see DESIGN.md)
inline bool vk_loop_cooperative()
{
  return vk_hook(VK_HOOK_LOOP_COOPERATIVE) == 1;
}

template <typename Kernel, typename A, typename B>
inline hipError_t launch_loop_kernel(Kernel kernel, int grid, int threads, hipStream_t s, A& a, B& b)
{
  if (vk_loop_cooperative())
  {
    void* args[2] = {&a, &b};
    return hipLaunchCooperativeKernel(reinterpret_cast<const void*>(kernel), dim3(grid), dim3(threads), args, 0, s);
  }
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), 0, s, a, b);
  return hipGetLastError();
}

// bracket every loop-kernel launch: loop kernels of different streams of one device never overlap
void vk_loop_launch_begin(hipStream_t stream);
void vk_loop_launch_end(hipStream_t stream);

namespace vk
{

// How many workgroups of `kernel` (THREADS wide, static LDS only) the current device holds
// at once: the grid of a loop kernel must not exceed it. 0 on error.
template <typename Kernel>
inline int resident_workgroups(Kernel kernel, int threads)
{
  struct Entry { const void* kernel; int device; int capacity; };
  static Entry cache[32];
  static int used = 0;
  static std::mutex guard;       // Tracks may be issued from several host threads (one per stream)
  std::lock_guard<std::mutex> hold(guard);
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return 0;
  int capacity = -1;
  for (int i = 0; i < used; ++i)
    if (cache[i].kernel == reinterpret_cast<const void*>(kernel) && cache[i].device == device) capacity = cache[i].capacity;
  if (capacity < 0)
  {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess)
      return 0;
    capacity = per_cu * cus > 0 ? per_cu * cus : 0;
    if (used < 32) cache[used++] = Entry{reinterpret_cast<const void*>(kernel), device, capacity};
  }
  // test aid: a smaller grid than the device could hold (workgroups then take several pixel
  // groups each — the path a device with fewer CUs, or larger images, would take)
  const int cap = vk_hook(VK_HOOK_LOOP_GRID_CAP);
  if (cap > 0 && cap < capacity) capacity = cap;
  return capacity;
}

template <int WAVES>
__device__ __forceinline__ void publish_partial(const float (&acc)[27], float (*lds)[kSysStride],
    const Exchange& E, int step, int group)
{
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 27; ++i)
  {
    const float v = wave_sum_lane63(acc[i]);
    if (lane == 63) lds[wave][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < 27)
  {
    float v = 0.0f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) v += lds[w][threadIdx.x];
    __hip_atomic_store(E.words + ((size_t)(step & 1) * E.count + group) * kSysStride + threadIdx.x,
        ((unsigned long long)exchange_tag(E.epoch, step) << 32) | (unsigned long long)__float_as_uint(v),
        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// The counterpart of sum_partials for exchanged sums: same slices, same order within a
// slice, so the result depends on the published values alone. THREADS = workgroup size.
// A thread owns 32 / (THREADS / 32) slices of one component and keeps sixteen requests in
// flight; a word whose tag is not yet this step's is asked for again. Returns false (for
// every thread of the workgroup) if some workgroup's sums did not arrive in time.
template <int THREADS>
__device__ __forceinline__ bool gather_partials(const Exchange& E, int step, int translation_enabled,
    float* hessian, float* gradient, float (*slices)[kSysStride], float* sums, int* failed)
{
  constexpr int PASS = THREADS / 32;            // slices covered by the workgroup at once
  constexpr int M = kSysSlices / PASS;          // slices per thread
  static_assert(M >= 1 && 16 % M == 0, "workgroup of 256, 512 or 1024 threads");
  const uint32_t tag = exchange_tag(E.epoch, step);
  const unsigned long long* words = E.words + (size_t)(step & 1) * E.count * kSysStride;
  const int c = threadIdx.x & 31;
  const int s0 = threadIdx.x >> 5;
  const unsigned long long deadline = (unsigned long long)wall_clock64() + kExchangeTimeout;
  float v[M];
#pragma unroll
  for (int m = 0; m < M; ++m) v[m] = 0.0f;
  bool ok = true;
  if (c < 27)
  {
    const int items = M * ((E.count + kSysSlices - 1) / kSysSlices);
    for (int i0 = 0; i0 < items && ok; i0 += 16)
    {
      // sixteen requests in flight; whatever came back with an older tag is asked for again,
      // all of those together (one round trip per attempt, however many are missing), and
      // the values are added only once all sixteen are there — in their fixed order
      unsigned long long w[16];
      unsigned missing = 0;
#pragma unroll
      for (int q = 0; q < 16; ++q)
      {
        const int k = s0 + (q % M) * PASS + kSysSlices * ((i0 + q) / M);     // slice s0 + m * PASS, term u
        if (k < E.count) missing |= 1u << q;
      }
      // (Measured and rejected: letting only the lane of a slot's first word poll while the
      // slot is missing, the others asking again once it has arrived — a quarter of the polling
      // requests, but one more round trip for most words: 256 -> 278 us per depth Track. Two
      // rounds of questions in flight, so that a word is seen at most one gap + one trip after
      // it has landed: 256 -> 324 us. A longer pause between rounds: 0-8 x 64 clocks make no
      // difference, 24 and 64 cost 5 % and 14 %.)
      for (;;)
      {
#pragma unroll
        for (int q = 0; q < 16; ++q)
        {
          const int k = s0 + (q % M) * PASS + kSysSlices * ((i0 + q) / M);
          if (missing & (1u << q))
            w[q] = __hip_atomic_load(words + (size_t)k * kSysStride + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q)
          if ((missing & (1u << q)) && (uint32_t)(w[q] >> 32) == tag) missing &= ~(1u << q);
        if (!missing) break;
        if ((unsigned long long)wall_clock64() > deadline) { ok = false; break; }
        __builtin_amdgcn_s_sleep(VK_POLL_GAP);
      }
      if (ok)
      {
#pragma unroll
        for (int q = 0; q < 16; ++q)
        {
          const int k = s0 + (q % M) * PASS + kSysSlices * ((i0 + q) / M);
          if (k < E.count) v[q % M] += __uint_as_float((uint32_t)w[q]);
        }
      }
    }
  }
#pragma unroll
  for (int m = 0; m < M; ++m) slices[s0 + m * PASS][c] = v[m];
  if (!ok) *failed = 1;      // LDS flag, zeroed by the caller before the loop; any writer writes 1
  __syncthreads();
  if (*failed) return false;
  finish_sums(slices, translation_enabled, hessian, gradient, sums);
  return true;
}

// ---- the same exchange by float atomics (an A/B experiment, -DVK_LOOP_ATOMIC_EXCHANGE; VERDICT r4 #6a) -----------------
// The reference itself sums with atomicAdd in arrival order (depth_tracker.cu:207-209,261-263). Here: three accumulators
// of 32 floats (step % 3) and one arrival counter at the head of the exchange area, zeroed by the host before the launch.
// A workgroup adds its 27 sums with returning atomics (their return is their completion), then — behind a workgroup
// barrier — bumps the counter; a reader waits for count x (step + 1) arrivals and reads 27 words instead of 27 x count.
// Buffer (step - 1) % 3 is zeroed by workgroup 0 once step `step` is complete (nobody reads it any more; it is added to
// again at step + 2, after arrivals that follow workgroup 0's own). The order of the additions is the order of arrival:
// the sums are no longer the same bits from run to run — the property the product's fixed-order exchange exists for.
#ifdef VK_LOOP_ATOMIC_EXCHANGE
template <int WAVES>
__device__ __forceinline__ void publish_atomic(const float (&acc)[27], float (*lds)[kSysStride], const Exchange& E, int step)
{
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 27; ++i)
  {
    const float v = wave_sum_lane63(acc[i]);
    if (lane == 63) lds[wave][i] = v;
  }
  __syncthreads();
  float* accumulators = reinterpret_cast<float*>(E.words) + 16;      // behind the counter's line
  if (threadIdx.x < 27)
  {
    float v = 0.0f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) v += lds[w][threadIdx.x];
    const float before = __hip_atomic_fetch_add(accumulators + (step % 3) * kSysStride + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" :: "v"(before));                                    // returned = performed
  }
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(E.words, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ bool gather_atomic(const Exchange& E, int step, int translation_enabled, float* hessian,
    float* gradient, float* sums, int* failed)
{
  float* accumulators = reinterpret_cast<float*>(E.words) + 16;
  if (threadIdx.x == 0)
  {
    const unsigned long long target = (unsigned long long)E.count * (unsigned long long)(step + 1);
    const unsigned long long deadline = (unsigned long long)wall_clock64() + kExchangeTimeout;
    while (__hip_atomic_load(E.words, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target)
    {
      if ((unsigned long long)wall_clock64() > deadline) { *failed = 1; break; }
      __builtin_amdgcn_s_sleep(VK_POLL_GAP);
    }
  }
  __syncthreads();
  if (*failed) return false;
  if (threadIdx.x < 27)
  {
    const float v = __hip_atomic_load(accumulators + (step % 3) * kSysStride + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int n = translation_enabled ? 6 : 3, nh = translation_enabled ? 21 : 6;
    if (threadIdx.x < 6)
    {
      const float g = (int)threadIdx.x < n ? v : 0.0f;
      if (gradient) gradient[threadIdx.x] = g;
      sums[36 + threadIdx.x] = g;
    }
    else
    {
      const int out = (int)threadIdx.x - 6;
      const float h = out < nh ? v : 0.0f;
      if (hessian) hessian[out] = h;
      sums[out] = h;
    }
    // the buffer of the step before has no reader left: ready for step + 2
    if (blockIdx.x == 0 && step >= 1)
      __hip_atomic_store(accumulators + ((step - 1) % 3) * kSysStride + threadIdx.x, 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x >= 27 && threadIdx.x < 42)     // hessian[21 .. 36) as finish_sums leaves them
  {
    sums[threadIdx.x - 6] = 0.0f;
    if (hessian) hessian[threadIdx.x - 6] = 0.0f;
  }
  __syncthreads();
  return true;
}
#endif

// LDL^T, no pivoting, float32 (the reference calls Eigen::LDLT — unpinned,
// not vendored; agreement is to rounding). A zero pivot is handled the way
// Eigen's LDLT handles it (the column of L is left at 0, and the solve sets
// y[j] = 0 where |D[j]| <= FLT_MIN): an empty or rank-deficient system — no valid
// correspondence, e.g. an all-hole frame — then yields update = 0, the step norm
// is below 1e-6 and the pose is left untouched, instead of 0/0 = NaN poisoning
// the pose (tracker.cpp:153-162). N is a template parameter and every
// loop is unrolled so that L, D, y live in registers: with a run-time size the
// arrays are indexed dynamically, land in scratch memory, and the one lane that
// solves spends ~10 us waiting on it.
template <int N>
__device__ __forceinline__ void ldlt_solve(const float (&A)[N * N], const float (&b)[N], float (&x)[N])
{
  float L[N * N], D[N], y[N];
#pragma unroll
  for (int i = 0; i < N * N; ++i) L[i] = 0.0f;

#pragma unroll
  for (int j = 0; j < N; ++j)
  {
    float d = A[j * N + j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= L[j * N + k] * L[j * N + k] * D[k];
    D[j] = d;
    L[j * N + j] = 1.0f;

#pragma unroll
    for (int i = j + 1; i < N; ++i)
    {
      float s = A[i * N + j];
#pragma unroll
      for (int k = 0; k < j; ++k) s -= L[i * N + k] * L[j * N + k] * D[k];
      L[i * N + j] = (fabsf(d) > 0.0f) ? s / d : 0.0f;
    }
  }

#pragma unroll
  for (int i = 0; i < N; ++i)
  {
    float s = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) s -= L[i * N + k] * y[k];
    y[i] = s;
  }

#pragma unroll
  for (int i = 0; i < N; ++i) y[i] = (fabsf(D[i]) > FLT_MIN) ? y[i] / D[i] : 0.0f;

#pragma unroll
  for (int i = N - 1; i >= 0; --i)
  {
    float s = y[i];
#pragma unroll
    for (int k = i + 1; k < N; ++k) s -= L[k * N + i] * x[k];
    x[i] = s;
  }
}

__device__ __forceinline__ void matmul4(const float (&A)[16], const float (&B)[16], float (&C)[16])  // matrix.h:297-318
{
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int m = 0; m < 4; ++m)
    {
      float r = 0.0f;
#pragma unroll
      for (int n = 0; n < 4; ++n) r += A[n * 4 + m] * B[p * 4 + n];
      C[p * 4 + m] = r;
    }
}

// tracker.cpp:142-159: unpack the packed lower triangle, solve H x = g, update = -x
// (entries beyond N stay 0)
template <int N>
__device__ __forceinline__ void solve_step(const float* hessian, const float* gradient, float (&update)[6])
{
  float H[N * N], g[N], x[N];
  int index = 0;
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j)
    {
      H[i * N + j] = hessian[index];
      H[j * N + i] = hessian[index];
      ++index;
    }
#pragma unroll
  for (int i = 0; i < N; ++i) g[i] = gradient[i];
  ldlt_solve<N>(H, g, x);
#pragma unroll
  for (int i = 0; i < 6; ++i) update[i] = 0.0f;
#pragma unroll
  for (int i = 0; i < N; ++i) update[i] = -x[i];
}

// depth_tracker.cpp:57-84 / color_tracker.cpp:67-95: re-orthonormalise the
// rotation columns of M and rebuild the rigid transform Translate(t) * Rotate(R)
// together with its inverse (transform.h:62-66,74-99,146-159).
__device__ __forceinline__ void rigid_from(const float (&M)[16], float (&out_m)[16], float (&out_i)[16])
{
  f3 x_axis = normalized3(make3(M[0], M[1], M[2]));
  f3 y_axis = normalized3(make3(M[4], M[5], M[6]));
  const f3 z_axis = cross3(x_axis, y_axis);
  y_axis = cross3(z_axis, x_axis);

  float Tm[16], Ti[16], Rm[16], Ri[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { Tm[i] = Ti[i] = Rm[i] = 0.0f; }
#pragma unroll
  for (int i = 0; i < 4; ++i) { Tm[5 * i] = Ti[5 * i] = Rm[5 * i] = 1.0f; }
  Tm[12] = M[12];  Tm[13] = M[13];  Tm[14] = M[14];
  Ti[12] = -M[12]; Ti[13] = -M[13]; Ti[14] = -M[14];
  Rm[0] = x_axis.x; Rm[1] = x_axis.y; Rm[2] = x_axis.z;
  Rm[4] = y_axis.x; Rm[5] = y_axis.y; Rm[6] = y_axis.z;
  Rm[8] = z_axis.x; Rm[9] = z_axis.y; Rm[10] = z_axis.z;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) Ri[c * 4 + r] = Rm[r * 4 + c];

  matmul4(Tm, Rm, out_m);
  matmul4(Ri, Ti, out_i);
}

// ---- the solve + pose update of one Gauss-Newton step, spread over the lanes of ONE wave ----
//
// solve_step / matmul4 / rigid_from above run on one lane: ~1000 dependent instructions of four
// cycles each, 1.7 us of every step of the loop kernels, on the critical path of every workgroup.
// Here the same operations — each element computed by exactly the same expression in the same
// order, so the bits are those of the one-lane code — are laid across lanes: row i of the LDL^T
// factorisation in lane i (the k-loop of an element stays sequential, the elements of a column
// are independent), a 4x4 product with one output element per lane. Values cross lanes through
// v_readlane (to scalar registers) and, for the transposed accesses, a few LDS words.
__device__ __forceinline__ float lane_value(float v, int from)
{
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), from));
}

// element (lane & 15) of A * B for column-major 4x4 matrices in LDS (matmul4: r = 0, r += A[n*4+m] * B[p*4+n])
__device__ __forceinline__ float matmul4_lane(const float* A, const float* B, int lane)
{
  const int p = (lane >> 2) & 3, m = lane & 3;
  float r = 0.0f;
#pragma unroll
  for (int n = 0; n < 4; ++n) r += A[n * 4 + m] * B[p * 4 + n];
  return r;
}

// x with H x = g for the packed system in `sums` (LDS: hessian [0, 21), gradient [36, 42)); returns
// update = -x in every lane (entries beyond N are 0). `scratch`: LDS, 64 floats. One whole wave.
template <int N>
__device__ __forceinline__ void wave_solve_step(const float* sums, float* scratch, float (&update)[6])
{
  const int lane = lane_id();
  const int row = lane < N ? lane : 0;          // lanes >= N repeat row 0; nothing of theirs is used
  float A[N], L[N], D[N];
#pragma unroll
  for (int j = 0; j < N; ++j)
  {
    const int r = row > j ? row : j, c = row > j ? j : row;
    A[j] = sums[r * (r + 1) / 2 + c];
  }
  const float b = sums[36 + row];
  float Dv = 0.0f;

#pragma unroll
  for (int j = 0; j < N; ++j)
  {
    float s = A[j];
#pragma unroll
    for (int k = 0; k < j; ++k) s -= L[k] * lane_value(L[k], j) * D[k];
    const float d = lane_value(s, j);
    D[j] = d;
    const float q = s / d;
    L[j] = (lane > j) ? ((fabsf(d) > 0.0f) ? q : 0.0f) : ((lane == j) ? 1.0f : 0.0f);
    Dv = (lane == j) ? d : Dv;
  }

  float y = b;
#pragma unroll
  for (int k = 0; k < N; ++k)
  {
    const float yk = lane_value(y, k);
    if (lane > k) y -= L[k] * yk;
  }
  y = (fabsf(Dv) > FLT_MIN) ? y / Dv : 0.0f;

  // column `lane` of L for the back substitution
  if (lane < N)
  {
#pragma unroll
    for (int j = 0; j < N; ++j) scratch[lane * 8 + j] = L[j];
  }
  wave_lds_fence();
  float Lt[N];
#pragma unroll
  for (int k = 0; k < N; ++k) Lt[k] = scratch[k * 8 + row];
  wave_lds_fence();

  float x[N];
#pragma unroll
  for (int i = N - 1; i >= 0; --i)
  {
    float s = y;
#pragma unroll
    for (int k = i + 1; k < N; ++k) s -= Lt[k] * x[k];
    x[i] = lane_value(s, i);
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) update[i] = 0.0f;
#pragma unroll
  for (int i = 0; i < N; ++i) update[i] = -x[i];
}

// rigid_from() with one element of the result per lane: `M_lane` = element (lane & 15) of M in the
// lanes 0..15; returns element (lane & 15) of Translate(t) * Rotate(R) (out_m of rigid_from).
// `scratch`: LDS, 64 floats.
__device__ __forceinline__ float wave_rigid_from(float M_lane, float* scratch)
{
  const int lane = lane_id();
  const int l = lane & 15;
  f3 x_axis = normalized3(make3(lane_value(M_lane, 0), lane_value(M_lane, 1), lane_value(M_lane, 2)));
  f3 y_axis = normalized3(make3(lane_value(M_lane, 4), lane_value(M_lane, 5), lane_value(M_lane, 6)));
  const f3 z_axis = cross3(x_axis, y_axis);
  y_axis = cross3(z_axis, x_axis);
  const float tx = lane_value(M_lane, 12), ty = lane_value(M_lane, 13), tz = lane_value(M_lane, 14);

  float Tm = (l % 5 == 0) ? 1.0f : 0.0f;
  Tm = (l == 12) ? tx : Tm;
  Tm = (l == 13) ? ty : Tm;
  Tm = (l == 14) ? tz : Tm;
  float Rm = (l == 15) ? 1.0f : 0.0f;
  Rm = (l == 0) ? x_axis.x : Rm;  Rm = (l == 1) ? x_axis.y : Rm;  Rm = (l == 2) ? x_axis.z : Rm;
  Rm = (l == 4) ? y_axis.x : Rm;  Rm = (l == 5) ? y_axis.y : Rm;  Rm = (l == 6) ? y_axis.z : Rm;
  Rm = (l == 8) ? z_axis.x : Rm;  Rm = (l == 9) ? z_axis.y : Rm;  Rm = (l == 10) ? z_axis.z : Rm;
  if (lane < 16) { scratch[lane] = Tm; scratch[16 + lane] = Rm; }
  wave_lds_fence();
  const float out = matmul4_lane(scratch, scratch + 16, lane);
  wave_lds_fence();
  return out;
}

// tracker.cpp:160-162: record the step and stop once it is shorter than 1e-6.
// `mirror` (optional): pinned host memory that receives {iterations, converged, epoch} as
// ONE 64-bit system-scope store after every step, so a host that enqueues the loop
// in chunks can stop enqueuing once it has converged (vk_track_poll, vk.h). The epoch
// (bits 48..63) names the Track call the word belongs to: the launches of one call may
// still be running when the next call starts to look at the mirror.
struct Mirror
{
  unsigned long long* word;
  uint32_t epoch;
  vk_transform* host_pose;   // optional (vk_track_poll::host_pose): where the final pose is left for vk_track_wait
};

// The last thing a Track does when the caller gave a host_pose: the pose goes to pinned host
// memory and then host_state[3] takes the call's tag (system-scope release), so that the
// host can pick the pose up by watching one word instead of a copy + stream synchronisation.
// Called by the first 32 lanes of ONE wave (the release of lane 0 then covers all 32 stores).
__device__ __forceinline__ void publish_host_pose(const Mirror& mirror, const vk_transform* pose)
{
  if (!mirror.word || !mirror.host_pose) return;
  if (threadIdx.x < 32)
  {
    const float v = threadIdx.x < 16 ? pose->m[threadIdx.x] : pose->inv[threadIdx.x - 16];
    if (threadIdx.x < 16) mirror.host_pose->m[threadIdx.x] = v; else mirror.host_pose->inv[threadIdx.x - 16] = v;
  }
  if (threadIdx.x == 0)
    __hip_atomic_store(reinterpret_cast<uint32_t*>(mirror.word) + 3, mirror.epoch & 0xffffu, __ATOMIC_RELEASE,
        __HIP_MEMORY_SCOPE_SYSTEM);
}

template <int N>
__device__ __forceinline__ void finish_step(const float (&update)[6], int32_t* state, float* update_out,
    Mirror mirror = Mirror{nullptr, 0, nullptr})
{
  float sq = 0.0f;
#pragma unroll
  for (int i = 0; i < N; ++i) sq += update[i] * update[i];
  if (update_out)
  {
#pragma unroll
    for (int i = 0; i < 6; ++i) update_out[i] = update[i];
  }
  if (state)
  {
    const int iterations = state[0] + 1;
    const int converged = (sqrtf(sq) < 1E-6f) ? 1 : state[1];
    state[0] = iterations;
    state[1] = converged;
    if (mirror.word)
      __hip_atomic_store(mirror.word, ((unsigned long long)(mirror.epoch & 0xffffu) << 48) |
          ((unsigned long long)(uint32_t)(converged & 1) << 32) | (uint32_t)iterations,
          __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ---- host side of the chunked loop ------------------------------------------------

// After the steps up to `target` have been enqueued on `s`: wait until the mirror shows that
// many steps or a converged loop, and say whether the loop had converged BY step `target`
// (stop enqueuing). The answer must not depend on how far ahead this GPU happens to be: in a
// multi-GPU rig every rank has to enqueue the same number of steps, because each step carries
// an all-reduce. A loop that converges stops writing the mirror, so a converged word holds the
// step at which it converged — the same number on every rank — and only that is compared.
// The spin is bounded by the stream itself: once the stream has drained nothing more will be
// written.
inline bool wait_for_steps(const Mirror& mirror, int target, hipStream_t s)
{
  const volatile unsigned long long* word = reinterpret_cast<const volatile unsigned long long*>(mirror.word);
  const unsigned long long tag = (unsigned long long)(mirror.epoch & 0xffffu);
  for (unsigned spin = 0;; ++spin)
  {
    const unsigned long long v = *word;
    if ((v >> 48) == tag)                          // a word of THIS call (earlier calls' launches may still write theirs)
    {
      const int steps = (int)(uint32_t)v;
      if ((v >> 32) & 1u) return steps <= target;  // converged at step `steps`
      if (steps >= target) return false;
    }
    if ((spin & 1023u) == 1023u && hipStreamQuery(s) != hipErrorNotReady)
    {
      const unsigned long long last = *word;      // drained: the mirror is final
      return (last >> 48) == tag && ((last >> 32) & 1u) && (int)(uint32_t)last <= target;
    }
  }
}

inline bool polling(const vk_track_poll* poll) { return poll && poll->host_state && poll->chunk > 0; }

// the mirror of one Track call: a fresh epoch, kept in the pinned block itself (word 2)
inline Mirror begin_mirror(const vk_track_poll* poll)
{
  Mirror m{nullptr, 0, nullptr};
  if (!poll || !poll->host_state) return m;
  volatile int32_t* host = poll->host_state;
  const uint32_t epoch = ((uint32_t)host[2] + 1u) & 0xffffu;
  host[2] = (int32_t)(epoch ? epoch : 1u);        // never 0: a zeroed block matches no call
  m.word = reinterpret_cast<unsigned long long*>(poll->host_state);
  m.epoch = (uint32_t)host[2];
  m.host_pose = poll->host_pose;
  return m;
}

// ---- how an image's pixels are grouped into partial sums --------------------------
//
// One workgroup adds up one GROUP of consecutive pixels (a lane takes pixels lane, lane +
// width of the workgroup, ... of the group, in that order). The group size depends on the
// image alone — never on the device — so the partial sums, and with them every bit of the
// result, are the same on any GPU and in both loops (launch per stage with a reduce hook, or
// everything in one launch): the smallest multiple of kGroupQuantum pixels that gives at
// most kTargetGroups groups. At most that many workgroups exchange their sums in the
// one-launch loops (the cost of the exchange grows with the square of their number), and
// there are enough of them to use the whole device at every pyramid level:
// 640x480 -> 240 groups of 1280 pixels, 320x240 -> 150 of 512, 1280x960 -> 253 of 4864.
// (Measured on the light tracker, shipped-app loop: groups of 1024 pixels — 75 / 300 of
// them — 1290 frames/s; this rule 1640.)
constexpr int kTargetGroups = 256;
constexpr int kGroupQuantum = 256;

inline int group_pixels_for(int total)
{
  int target = kTargetGroups, quantum = kGroupQuantum;
#ifdef VK_LOOP_TIMING
  if (const char* e = getenv("VK_TARGET_GROUPS")) target = atoi(e);     // development sweeps only
  if (const char* e = getenv("VK_GROUP_QUANTUM")) quantum = atoi(e);
#endif
  const int per_group = (total + target - 1) / target;
  const int rounded = (per_group + quantum - 1) / quantum * quantum;
  return rounded < quantum ? quantum : rounded;
}

inline int group_count_for(int total, int group_pixels) { return (total + group_pixels - 1) / group_pixels; }

// The depth tracker's workgroups are kIcpThreads lanes wide; a lane works on up to
// kIcpPixels of its pixels at a time (their keyframe loads are in flight together).
#ifndef VK_ICP_THREADS
#define VK_ICP_THREADS 512
#endif
#ifndef VK_ICP_PIXELS
#define VK_ICP_PIXELS 4
#endif
constexpr int kIcpThreads = VK_ICP_THREADS;
constexpr int kIcpPixels = VK_ICP_PIXELS;

}  // namespace vk
