// vk_probe.hip — bandwidth probes (libvk_probe.so, see vk_probe.h). Measurement
// tooling only: nothing in libvk_hip.so or vulcan_amd/ depends on it.
#include "../../vulcan_amd/csrc/vk_common.hpp"
#include "../../vulcan_amd/csrc/vk_raycast.hpp"
#include "vk_probe.h"

using namespace vk;

namespace
{

typedef float nf4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void copy_one_kernel(float4* __restrict__ dst, const float4* __restrict__ src, size_t n4)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) dst[i] = src[i];
}

// a workgroup moves 16 KiB: four 4-KiB rows, every wave-instruction a contiguous 1 KiB
template <bool NT>
__global__ __launch_bounds__(256) void copy_four_kernel(float4* __restrict__ dst, const float4* __restrict__ src, size_t n4)
{
  const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
  float4 r[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
  {
    const size_t i = base + (size_t)k * 256;
    if (i < n4)
    {
      if (NT) { const nf4 t = __builtin_nontemporal_load(reinterpret_cast<const nf4*>(src + i)); r[k] = make_float4(t.x, t.y, t.z, t.w); }
      else r[k] = src[i];
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
  {
    const size_t i = base + (size_t)k * 256;
    if (i < n4)
    {
      if (NT) { nf4 t; t.x = r[k].x; t.y = r[k].y; t.z = r[k].z; t.w = r[k].w; __builtin_nontemporal_store(t, reinterpret_cast<nf4*>(dst + i)); }
      else dst[i] = r[k];
    }
  }
}

__global__ __launch_bounds__(256) void copy_persistent_kernel(float4* __restrict__ dst, const float4* __restrict__ src, size_t n4)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride)
  {
    const float4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
  }
  for (; i < n4; i += stride) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void read_kernel(const float4* __restrict__ src, size_t n4, float* __restrict__ sink)
{
  const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
  float acc = 0.0f;
#pragma unroll
  for (int k = 0; k < 4; ++k)
  {
    const size_t i = base + (size_t)k * 256;
    if (i < n4) { const float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
  }
  if (acc == 123.456f) sink[0] = acc;   // keeps the loads; practically never true
}

// Reads every visible block (10 240 B) and writes it back unchanged: one wave per
// block, ten float4 per lane, the integrate kernel's persistent grid.
template <int MODE>
__global__ __launch_bounds__(256) void block_rmw_kernel(float4* __restrict__ voxels4,
    const vk_hash_entry* __restrict__ entries, const int32_t* __restrict__ visible,
    const int32_t* __restrict__ counters)
{
  const int lane = lane_id();
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int total_waves = gridDim.x * 4;
  const int count = counters[VK_CTR_VISIBLE];

  for (int i = wave; i < count; i += total_waves)
  {
    const Entry entry = load_entry(entries, (uint32_t)__builtin_amdgcn_readfirstlane(visible[i]));
    if (entry.data < 0) continue;
    float4* block4 = voxels4 + (size_t)entry.data * 640;
    float4 r[10];
#pragma unroll
    for (int k = 0; k < 10; ++k)
    {
      if (MODE == 2)
      {
        const nf4 t = __builtin_nontemporal_load(reinterpret_cast<const nf4*>(&block4[k * 64 + lane]));
        r[k] = make_float4(t.x, t.y, t.z, t.w);
      }
      else r[k] = block4[k * 64 + lane];
    }
#pragma unroll
    for (int k = 0; k < 10; ++k)
    {
      r[k].x += 0.0f;   // keeps the store: x + 0.0f is not an identity for -0.0f
      if (MODE == 1 || MODE == 2)
      {
        nf4 t; t.x = r[k].x; t.y = r[k].y; t.z = r[k].z; t.w = r[k].w;
        __builtin_nontemporal_store(t, reinterpret_cast<nf4*>(&block4[k * 64 + lane]));
      }
      else block4[k * 64 + lane] = r[k];
    }
  }
}

// the product's ray march with the counting hook compiled in (one wave per 8x8 tile)
__global__ __launch_bounds__(256) void count_points_kernel(PointParams P, unsigned long long* wave_clocks)
{
  const unsigned long long t0 = wall_clock64();
  __shared__ int4 directories[4][kDirWords];
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  int4* bdir = directories[wave];
  bdir[lane] = make_int4(INT32_MIN, INT32_MIN, INT32_MIN, -1);
  wave_lds_fence();
  const int tiles_x = (P.image_width + 15) / 16;
  const int tile_x = blockIdx.x % tiles_x, tile_y = blockIdx.x / tiles_x;
  const int x = tile_x * 16 + (wave & 1) * 8 + (lane & 7);
  const int y = tile_y * 16 + (wave >> 1) * 8 + (lane >> 3);
  if (x < P.image_width && y < P.image_height)
  {
    const int px = P.bounds_width * x / P.image_width;
    const int py = P.bounds_height * y / P.image_height;
    const float2 bound = reinterpret_cast<const float2*>(P.bounds)[py * P.bounds_width + px];
    if (P.touched) march_ray<true, false>(P, bdir, x, y, bound, (int)blockIdx.x * 4 + wave);
    else march_ray<false, true>(P, bdir, x, y, bound);
  }
  // per-wave start / end on the 100 MHz wall clock (distribution of wave lifetimes)
  if (wave_clocks && lane == 0)
  {
    wave_clocks[2 * (blockIdx.x * 4 + wave) + 0] = t0;
    wave_clocks[2 * (blockIdx.x * 4 + wave) + 1] = wall_clock64();
  }
}

// launch-floor probe: a kernel whose every wave reads one counter and leaves
__global__ __launch_bounds__(256) void early_exit_kernel(const int32_t* __restrict__ counters, float* __restrict__ sink)
{
  if (counters[0] == 0) return;
  sink[blockIdx.x * 256 + threadIdx.x] = 1.0f;
}

}  // namespace

// ---- correctly rounded 1 / d and sqrt(x) for operands in the middle of the range: an experiment of round 5 ----------------
// `1.0f / d` and `sqrtf(x)` are correctly rounded (IEEE-754 RN), so any sequence that is correctly rounded gives their bits.
// hipcc's own expansions spend most of their instructions on the ends of the range: 1 / d = v_div_scale x2, v_rcp, five
// multiply-adds, v_div_fmas, v_div_fixup (11 vector instructions); sqrt = scale by 2^32 below 2^-96 (3), v_sqrt, the two
// neighbours tried by their residuals (8), unscale (2), class fix-up (3) (17). For |d|, x in [2^-60, 2^60] none of that is
// needed: v_rcp + ONE Newton step (3), and v_sqrt + the neighbours' test (9), give the same bits — for EVERY float of the
// range, which is what rounding_kernel checks (2.0e9 / 1.0e9 values, 0 differences; the bare v_rcp differs for 10.7 %).
// In the integrate kernel (projection, light direction, shading: three reciprocals and a square root per shaded voxel, 32
// of its ~200 vector instructions) the guarded form — fast path + hipcc's expansion behind a range test — measures 33.96 us
// against 33.85, and even WITHOUT the guard only 33.4: since the wave priorities the light model's arithmetic runs in the
// gaps of the memory phases, and fewer instructions there buy nothing. Not adopted; docs/rounds/r05.md section 8.
__device__ __forceinline__ bool mid_range(float x)
{
  const float a = __builtin_fabsf(x);
  return a >= 0x1p-60f && a <= 0x1p60f;      // (a NaN fails both)
}
#ifndef VK_RCP_STEPS
#define VK_RCP_STEPS 1
#endif
__device__ __forceinline__ float rcp_rn_mid(float d)
{
  float r = __builtin_amdgcn_rcpf(d);
#pragma unroll
  for (int i = 0; i < VK_RCP_STEPS; ++i) r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
  return r;
}
__device__ __forceinline__ float sqrt_rn_mid(float x)
{
  const float s = __builtin_amdgcn_sqrtf(x);
  // the neighbours of s, as hipcc's expansion tries them: s - 1 ulp if its residual says s is too large, s + 1 ulp if too small
  const float below = __int_as_float(__float_as_int(s) - 1), above = __int_as_float(__float_as_int(s) + 1);
  const float r_below = __builtin_fmaf(-below, s, x), r_above = __builtin_fmaf(-above, s, x);
  float out = r_below <= 0.0f ? below : s;
  out = r_above > 0.0f ? above : out;
  return out;
}

// Every float: rcp_rn_mid / sqrt_rn_mid against the compiler's own 1.0f / x and sqrtf(x), bit for bit, over the range a
// caller would use them in. out[0], out[1]: values tested, values that differ (reciprocal); out[2], out[3]: the same
// for the square root; out[4], out[5]: the first differing input of each (bit pattern + 1; 0 = none).
__global__ __launch_bounds__(256) void rounding_kernel(unsigned long long* out)
{
  unsigned long long tested_r = 0, bad_r = 0, tested_s = 0, bad_s = 0;
  const unsigned long long threads = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long b = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b < (1ull << 32); b += threads)
  {
    const float x = __uint_as_float((uint32_t)b);
    if (!mid_range(x)) continue;
    ++tested_r;
    if (__float_as_uint(rcp_rn_mid(x)) != __float_as_uint(1.0f / x))
    {
      ++bad_r;
      atomicCAS(&out[4], 0ull, b + 1);
    }
    if (x > 0.0f)
    {
      ++tested_s;
      if (__float_as_uint(sqrt_rn_mid(x)) != __float_as_uint(sqrtf(x)))
      {
        ++bad_s;
        atomicCAS(&out[5], 0ull, b + 1);
      }
    }
  }
  atomicAdd(&out[0], tested_r);
  if (bad_r) atomicAdd(&out[1], bad_r);
  atomicAdd(&out[2], tested_s);
  if (bad_s) atomicAdd(&out[3], bad_s);
}


// ---- FETCH_SIZE calibration: the raycast's gather shape over a known set of lines (vk_probe.h vk_probe_gather) ----
__host__ __device__ inline uint32_t gather_hash(uint32_t block, uint32_t lane)
{
  uint32_t x = block * 0x9E3779B1u + lane * 0x85EBCA77u + 0x165667B1u;
  x ^= x >> 15;  x *= 0x2C1B3C6Du;  x ^= x >> 12;  x *= 0x297A2D39u;  x ^= x >> 15;
  return x;
}

template <int MODE>
__global__ __launch_bounds__(256) void gather_kernel(const char* __restrict__ voxels, int blocks, float* __restrict__ sink)
{
  const int wave = (int)(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int lane = (int)(threadIdx.x & 63);
  if (wave >= blocks) return;
  const char* block = voxels + (size_t)wave * 10240;
  float acc = 0.0f;
  if (MODE == 0)
  {
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += *reinterpret_cast<const float*>(block + (size_t)(k * 64 + lane) * 20);
  }
  else if (MODE == 1)
  {
    acc = *reinterpret_cast<const float*>(block + (size_t)(8 * lane) * 20);
  }
  else if (MODE == 2)
  {
    const uint32_t h = gather_hash((uint32_t)wave, (uint32_t)lane);
    const int x = (int)(h & 7) % 7, y = (int)((h >> 3) & 7) % 7, z = (int)((h >> 6) & 7) % 7;     // low corner in [0, 6]
#pragma unroll
    for (int c = 0; c < 8; ++c)
    {
      const int v = (z + (c >> 2)) * 64 + (y + ((c >> 1) & 1)) * 8 + x + (c & 1);
      const char* p = block + (size_t)v * 20;
      typedef float vf3 __attribute__((ext_vector_type(3)));
      typedef vf3 __attribute__((aligned(4))) vf3u;
      const vf3 rgb = *reinterpret_cast<const vf3u*>(p + 4);
      acc += *reinterpret_cast<const float*>(p) + rgb.x + rgb.y + rgb.z;
    }
  }
  else
  {
#pragma unroll
    for (int k = 0; k < 10; ++k)
    {
      const float4 v = *reinterpret_cast<const float4*>(block + (size_t)(k * 64 + lane) * 16);
      acc += v.x + v.y + v.z + v.w;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) sink[wave] = acc;
}

extern "C" {

int vk_probe_rounding(unsigned long long* out_dev6, void* stream)
{
  VK_REQUIRE(out_dev6);
  VK_CHECK(hipMemsetAsync(out_dev6, 0, 6 * sizeof(unsigned long long), vk_s(stream)));
  hipLaunchKernelGGL(rounding_kernel, dim3(16384), dim3(256), 0, vk_s(stream), out_dev6);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

int vk_probe_stream_copy(void* dst, const void* src, size_t bytes, int shape, void* stream)
{
  VK_REQUIRE(dst && src && bytes >= 16 && (bytes % 16) == 0 && shape >= 0 && shape <= 3);
  const size_t n4 = bytes / 16;
  float4* d = reinterpret_cast<float4*>(dst);
  const float4* s = reinterpret_cast<const float4*>(src);
  VK_REQUIRE((n4 + 255) / 256 < (size_t)1 << 31);
  switch (shape)
  {
    case 0: hipLaunchKernelGGL(copy_one_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, vk_s(stream), d, s, n4); break;
    case 1: hipLaunchKernelGGL(copy_four_kernel<false>, dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, vk_s(stream), d, s, n4); break;
    case 2: hipLaunchKernelGGL(copy_persistent_kernel, dim3(kCUs * 8), dim3(256), 0, vk_s(stream), d, s, n4); break;
    default: hipLaunchKernelGGL(copy_four_kernel<true>, dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, vk_s(stream), d, s, n4); break;
  }
  VK_LAUNCH_CHECK();
  return VK_OK;
}

int vk_probe_stream_read(const void* src, size_t bytes, float* sink, void* stream)
{
  VK_REQUIRE(src && sink && bytes >= 16 && (bytes % 16) == 0);
  const size_t n4 = bytes / 16;
  hipLaunchKernelGGL(read_kernel, dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, vk_s(stream),
      reinterpret_cast<const float4*>(src), n4, sink);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

int vk_probe_trace_touched(const vk_hash_entry* entries, const vk_voxel* voxels, const float* bounds,
    int block_count, float block_length, float voxel_length, float trunc_length, const vk_transform* Twc,
    const vk_projection* projection, float* depths, float* colors, int image_width, int image_height,
    int bounds_width, int bounds_height, uint8_t* touched, unsigned long long* wave_clocks, void* stream)
{
  return vk_probe_trace_steps(entries, voxels, bounds, block_count, block_length, voxel_length, trunc_length, Twc, projection,
      depths, colors, image_width, image_height, bounds_width, bounds_height, touched, wave_clocks, nullptr, stream);
}

int vk_probe_trace_steps(const vk_hash_entry* entries, const vk_voxel* voxels, const float* bounds,
    int block_count, float block_length, float voxel_length, float trunc_length, const vk_transform* Twc,
    const vk_projection* projection, float* depths, float* colors, int image_width, int image_height,
    int bounds_width, int bounds_height, uint8_t* touched, unsigned long long* wave_clocks, int* march_steps, void* stream)
{
  return vk_probe_trace_log(entries, voxels, bounds, block_count, block_length, voxel_length, trunc_length, Twc, projection,
      depths, colors, image_width, image_height, bounds_width, bounds_height, touched, wave_clocks, march_steps, nullptr, 0, stream);
}

int vk_probe_trace_log(const vk_hash_entry* entries, const vk_voxel* voxels, const float* bounds,
    int block_count, float block_length, float voxel_length, float trunc_length, const vk_transform* Twc,
    const vk_projection* projection, float* depths, float* colors, int image_width, int image_height,
    int bounds_width, int bounds_height, uint8_t* touched, unsigned long long* wave_clocks, int* march_steps,
    unsigned long long* trip_log, int trip_log_passes, void* stream)
{
  VK_REQUIRE(entries && voxels && bounds && Twc && projection && depths && colors && (touched || wave_clocks));
  VK_REQUIRE(block_count > 0 && image_width > 0 && image_height > 0 && bounds_width > 0 && bounds_height > 0);
  PointParams P;
  P.entries = entries;
  P.voxels = voxels;
  P.bounds = bounds;
  P.partials = nullptr;
  P.bounds_out = nullptr;
  P.K = (uint32_t)block_count;
  P.block_length = block_length;
  P.voxel_length = voxel_length;
  P.trunc_length = trunc_length;
  P.inv_block_length = 1.0 / (double)block_length;
  P.inv_voxel_length = 1.0 / (double)voxel_length;
  P.Twc = make_rt(Twc->m);
  P.Tcw = make_rt(Twc->inv);
  P.k = make_projection(*projection);
  P.depths = depths;
  P.colors = colors;
  P.image_width = image_width;
  P.image_height = image_height;
  P.bounds_width = bounds_width;
  P.bounds_height = bounds_height;
  P.touched = touched;
  P.march_steps = march_steps;
  P.trip_log = trip_log;
  P.trip_log_passes = trip_log_passes;
  P.rows_done = nullptr;
  P.rows_target = 0;
  P.late_dev = nullptr;
  P.late_host = nullptr;
  P.normal_polls = 0;
  const int tiles = ((image_width + 15) / 16) * ((image_height + 15) / 16);
  hipLaunchKernelGGL(count_points_kernel, dim3(tiles), dim3(256), 0, vk_s(stream), P, wave_clocks);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

int vk_probe_block_rmw(const vk_volume* v, int mode, void* stream)
{
  VK_REQUIRE(v && v->voxels && v->hash_entries && v->visible_blocks && v->counters);
  const int max_count = v->main_block_count + v->excess_block_count;
  int grid = (max_count + 3) / 4;
  if (grid > kCUs * 4) grid = kCUs * 4;
  float4* vox = reinterpret_cast<float4*>(v->voxels);
  switch (mode)
  {
    case 1: hipLaunchKernelGGL(block_rmw_kernel<1>, dim3(grid), dim3(256), 0, vk_s(stream), vox, v->hash_entries, v->visible_blocks, v->counters); break;
    case 2: hipLaunchKernelGGL(block_rmw_kernel<2>, dim3(grid), dim3(256), 0, vk_s(stream), vox, v->hash_entries, v->visible_blocks, v->counters); break;
    default: hipLaunchKernelGGL(block_rmw_kernel<0>, dim3(grid), dim3(256), 0, vk_s(stream), vox, v->hash_entries, v->visible_blocks, v->counters); break;
  }
  VK_LAUNCH_CHECK();
  return VK_OK;
}

int vk_probe_launch_floor(const int32_t* counters, float* sink, int workgroups, int launches, int replays, void* stream)
{
  VK_REQUIRE(counters && sink && workgroups > 0 && launches > 0 && replays > 0);
  hipStream_t s = vk_s(stream);
  for (int r = 0; r < replays; ++r)
    for (int i = 0; i < launches; ++i)
      hipLaunchKernelGGL(early_exit_kernel, dim3(workgroups), dim3(256), 0, s, counters, sink);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

int vk_probe_launch_floor_graph(const int32_t* counters, float* sink, int workgroups, int launches, int replays,
    float* us_per_launch)
{
  VK_REQUIRE(counters && sink && workgroups > 0 && launches > 0 && replays > 0 && us_per_launch);
#define VK_PROBE_TRY(expr) { const hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; }
  hipStream_t s;
  hipGraph_t graph;
  hipGraphExec_t exec;
  hipEvent_t e0, e1;
  VK_PROBE_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  VK_PROBE_TRY(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < launches; ++i)
    hipLaunchKernelGGL(early_exit_kernel, dim3(workgroups), dim3(256), 0, s, counters, sink);
  VK_PROBE_TRY(hipStreamEndCapture(s, &graph));
  VK_PROBE_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  VK_PROBE_TRY(hipEventCreate(&e0));
  VK_PROBE_TRY(hipEventCreate(&e1));
  for (int r = 0; r < 20; ++r) VK_PROBE_TRY(hipGraphLaunch(exec, s));
  VK_PROBE_TRY(hipStreamSynchronize(s));
  VK_PROBE_TRY(hipEventRecord(e0, s));
  for (int r = 0; r < replays; ++r) VK_PROBE_TRY(hipGraphLaunch(exec, s));
  VK_PROBE_TRY(hipEventRecord(e1, s));
  VK_PROBE_TRY(hipEventSynchronize(e1));
  float ms = 0.0f;
  VK_PROBE_TRY(hipEventElapsedTime(&ms, e0, e1));
  *us_per_launch = ms * 1e3f / (float)(replays * launches);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipGraphExecDestroy(exec);
  (void)hipGraphDestroy(graph);
  (void)hipStreamDestroy(s);
#undef VK_PROBE_TRY
  return 0;
}


int vk_probe_gather(const void* voxels, int blocks, int mode, float* sink, void* stream)
{
  VK_REQUIRE(voxels && sink && blocks > 0 && mode >= 0 && mode <= 3);
  const dim3 grid((blocks + 3) / 4);
  const char* v = static_cast<const char*>(voxels);
  if (mode == 0) hipLaunchKernelGGL(gather_kernel<0>, grid, dim3(256), 0, vk_s(stream), v, blocks, sink);
  else if (mode == 1) hipLaunchKernelGGL(gather_kernel<1>, grid, dim3(256), 0, vk_s(stream), v, blocks, sink);
  else if (mode == 2) hipLaunchKernelGGL(gather_kernel<2>, grid, dim3(256), 0, vk_s(stream), v, blocks, sink);
  else hipLaunchKernelGGL(gather_kernel<3>, grid, dim3(256), 0, vk_s(stream), v, blocks, sink);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

}  // extern "C"
