// vk_probe.hip — bandwidth probes used by bench.py / tools to put the integrate
// kernel's GB/s next to what the same GPU sustains on (a) a plain float4 copy
// and (b) the integrate kernel's own access pattern with the arithmetic removed
// (SURVEY.md §8d "Peak to divide by: measured").
#include "vk_common.hpp"

using namespace vk;

namespace
{

typedef float nf4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void stream_copy_kernel(float4* __restrict__ dst,
    const float4* __restrict__ src, size_t n4)
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

// Reads every visible block (10 240 B) and writes it back unchanged: one wave per
// block, ten float4 per lane, same persistent grid as integrate_kernel.
template <int MODE>   // 0 plain, 1 nontemporal stores, 2 nontemporal loads + stores, 3 delayed plain stores
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
    if (MODE == 3)
    {
      // ~2000 cycles of dependent ALU work between the loads and the stores
      float acc = r[0].x;
      for (int t = 0; t < 500; ++t) acc = acc * 1.0000001f + 1e-9f;
      r[9].w += acc * 0.0f;
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

}  // namespace

extern "C" {

int vk_probe_stream_copy(void* dst, const void* src, size_t bytes, void* stream)
{
  VK_REQUIRE(dst && src && bytes >= 16 && (bytes % 16) == 0);
  hipLaunchKernelGGL(stream_copy_kernel, dim3(kCUs * 8), dim3(256), 0, vk_s(stream),
      reinterpret_cast<float4*>(dst), reinterpret_cast<const float4*>(src), bytes / 16);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

int g_rmw_mode = 0;
int vk_probe_block_rmw_mode(int mode) { g_rmw_mode = mode; return VK_OK; }

int vk_probe_block_rmw(const vk_volume* v, void* stream)
{
  VK_REQUIRE(v && v->voxels && v->hash_entries && v->visible_blocks && v->counters);
  const int max_count = v->main_block_count + v->excess_block_count;
  int grid = (max_count + 3) / 4;
  if (grid > kCUs * 4) grid = kCUs * 4;
  float4* vox = reinterpret_cast<float4*>(v->voxels);
  switch (g_rmw_mode)
  {
    case 1: hipLaunchKernelGGL(block_rmw_kernel<1>, dim3(grid), dim3(256), 0, vk_s(stream), vox, v->hash_entries, v->visible_blocks, v->counters); break;
    case 2: hipLaunchKernelGGL(block_rmw_kernel<2>, dim3(grid), dim3(256), 0, vk_s(stream), vox, v->hash_entries, v->visible_blocks, v->counters); break;
    case 3: hipLaunchKernelGGL(block_rmw_kernel<3>, dim3(grid), dim3(256), 0, vk_s(stream), vox, v->hash_entries, v->visible_blocks, v->counters); break;
    default: hipLaunchKernelGGL(block_rmw_kernel<0>, dim3(grid), dim3(256), 0, vk_s(stream), vox, v->hash_entries, v->visible_blocks, v->counters); break;
  }
  VK_LAUNCH_CHECK();
  return VK_OK;
}

}  // extern "C"
