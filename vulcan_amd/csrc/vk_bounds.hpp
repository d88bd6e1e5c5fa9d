// vk_bounds.hpp — screen-space depth bounds of the visible blocks (ref:
// src/tracer.cu:13-112 ComputePatches + ComputeBounds), shared by vk_trace.hip and
// by the integrate launch that can compute them ahead of the raycast.
#pragma once

#include <cstring>

#include "vk_common.hpp"

namespace vk
{

struct BlockRect
{
  int bmin_x, bmin_y, bmax_x, bmax_y;
  float near_, far_;
  int gx, gy, count;
};

// tracer.cu:23-71: project the 8 corners, running clamp of the cell rectangle
// and of the depth interval, then the patch grid size.
__device__ __forceinline__ BlockRect block_rect(const Entry& entry, const Rt& Tcw,
    const vk_projection& k, float block_length, float min_depth, float max_depth,
    int image_width, int image_height, int bounds_width, int bounds_height)
{
  int bmax_x = -1, bmax_y = -1;
  int bmin_x = (int16_t)bounds_width, bmin_y = (int16_t)bounds_height;
  float d0 = +FLT_MAX, d1 = -FLT_MAX;

#pragma unroll
  for (int z = 0; z <= 1; ++z)
  {
    const float wz = block_length * (z + entry.oz);
#pragma unroll
    for (int y = 0; y <= 1; ++y)
    {
      const float wy = block_length * (y + entry.oy);
#pragma unroll
      for (int x = 0; x <= 1; ++x)
      {
        const float wx = block_length * (x + entry.ox);
        const f3 Xcp = xform_point(Tcw, make3(wx, wy, wz));
        float u, v;
        project(k, Xcp, u, v);
        u = bounds_width * u / image_width;
        v = bounds_height * v / image_height;

        bmin_x = vclampi(vmini(f2s(floorf(u)), bmin_x), 0, bounds_width - 1);
        bmin_y = vclampi(vmini(f2s(floorf(v)), bmin_y), 0, bounds_height - 1);
        bmax_x = vclampi(vmaxi(f2s(ceilf(u)), bmax_x), 0, bounds_width - 1);
        bmax_y = vclampi(vmaxi(f2s(ceilf(v)), bmax_y), 0, bounds_height - 1);

        d0 = vclamp(vmin(Xcp.z, d0), min_depth, max_depth);
        d1 = vclamp(vmax(Xcp.z, d1), min_depth, max_depth);
      }
    }
  }

  BlockRect r;
  r.bmin_x = bmin_x; r.bmin_y = bmin_y; r.bmax_x = bmax_x; r.bmax_y = bmax_y;
  r.near_ = d0; r.far_ = d1;
  const int rx = bmax_x - bmin_x;
  const int ry = bmax_y - bmin_y;
  r.gx = (rx + VK_PATCH_MAX_SIZE - 1) / VK_PATCH_MAX_SIZE;
  r.gy = (ry + VK_PATCH_MAX_SIZE - 1) / VK_PATCH_MAX_SIZE;
  r.count = (d1 > d0) ? r.gx * r.gy : 0;
  if (r.count < 0) r.count = 0;
  return r;
}

struct PatchParams
{
  const int32_t* indices;
  const vk_hash_entry* entries;
  Rt Tcw;
  vk_projection k;
  float block_length, min_depth, max_depth;
  int block_count;
  const int32_t* block_count_dev;
  int image_width, image_height, bounds_width, bounds_height;
  vk_patch* patches;
  int patch_capacity;
  int32_t* patch_count;
  float* bounds;
};

// The fused bounds pass. Global float atomics on 9600 words from ~7k blocks x ~10
// cells ran at ~1.5 atomics/ns (77 us per frame, r01 profile): instead each of
// kBoundsGroups workgroups folds its share of the visible blocks into a PRIVATE
// copy of the grid in LDS (ds_min/ds_max_i32), then stores the copy with plain
// coalesced writes. The consumer takes min/max over the kBoundsGroups copies
// (compute_points_kernel, or merge_bounds_kernel for the stand-alone API). No
// global atomics, no reset pass, no inter-workgroup order.
constexpr int kBoundsGroups = 32;
constexpr int kBoundsMaxCells = 8192;  // 64 KiB of LDS

// Body of one of the kBoundsGroups workgroups (any width): `grid` is 2 * cells
// ints of LDS, `group` selects the share of the visible list and the output copy.
__device__ __forceinline__ void bounds_group(const PatchParams& P, float2* __restrict__ partials, int* grid,
    int group, int threads)
{
  const int cells = P.bounds_width * P.bounds_height;
  const int block_count = P.block_count_dev ? min(*P.block_count_dev, P.block_count) : P.block_count;

  for (int c = threadIdx.x; c < cells; c += threads)
  {
    grid[2 * c + 0] = __float_as_int(+FLT_MAX);
    grid[2 * c + 1] = __float_as_int(-FLT_MAX);
  }
  __syncthreads();

  // visible block i goes to group i % kBoundsGroups: every group gets the same share
  // whatever the count
  for (int index = group + kBoundsGroups * (int)threadIdx.x; index < block_count; index += kBoundsGroups * threads)
  {
    const Entry entry = load_entry(P.entries, (uint32_t)P.indices[index]);
    const BlockRect r = block_rect(entry, P.Tcw, P.k, P.block_length, P.min_depth, P.max_depth,
        P.image_width, P.image_height, P.bounds_width, P.bounds_height);
    if (r.count <= 0) continue;

    // The patches tile [bmin, bmin + 16*g) clipped to bmax (tracer.cu:67-82). With
    // g = (bmax - bmin + 15) / 16 a span that is an exact multiple of 16 leaves
    // its last column / row uncovered in the reference; reproduced here.
    const int x_end = vmini(r.bmax_x, r.bmin_x + VK_PATCH_MAX_SIZE * r.gx - 1);
    const int y_end = vmini(r.bmax_y, r.bmin_y + VK_PATCH_MAX_SIZE * r.gy - 1);
    const int n = __float_as_int(r.near_), f = __float_as_int(r.far_);
    for (int y = r.bmin_y; y <= y_end; ++y)
      for (int x = r.bmin_x; x <= x_end; ++x)
      {
        // the values only ever move one way, so a plain read that already beats
        // ours makes the atomic unnecessary; neighbouring blocks cover the same
        // cells and same-address LDS atomics serialise
        const int c = y * P.bounds_width + x;
        if (n < grid[2 * c + 0]) atomicMin(&grid[2 * c + 0], n);
        if (f > grid[2 * c + 1]) atomicMax(&grid[2 * c + 1], f);
      }
  }
  __syncthreads();

  float2* out = partials + (size_t)group * cells;
  for (int c = threadIdx.x; c < cells; c += threads)
    out[c] = make_float2(__int_as_float(grid[2 * c + 0]), __int_as_float(grid[2 * c + 1]));
}

// Cells the bounds workgroups inside an integrate launch can hold (their LDS is
// shared with the integrate workgroups' tiles): the default 80 x 60 grid.
constexpr int kAheadMaxCells = 4800;

inline int fill_patch_params(PatchParams& P, const int32_t* indices, const vk_hash_entry* entries,
    const vk_transform* Tcw, const vk_projection* projection, float block_length, float min_depth,
    float max_depth, int block_count, const int32_t* block_count_dev, int image_width,
    int image_height, int bounds_width, int bounds_height)
{
  if (!indices || !entries || !Tcw || !projection) return VK_ERR_ARGUMENT;
  if (block_count < 0 || image_width <= 0 || image_height <= 0 || bounds_width <= 0 ||
      bounds_height <= 0 || bounds_width > 32767 || bounds_height > 32767)
    return VK_ERR_ARGUMENT;
  P.indices = indices;
  P.entries = entries;
  P.Tcw = make_rt(Tcw->m);
  P.k = *projection;
  P.block_length = block_length;
  P.min_depth = min_depth;
  P.max_depth = max_depth;
  P.block_count = block_count;
  P.block_count_dev = block_count_dev;
  P.image_width = image_width;
  P.image_height = image_height;
  P.bounds_width = bounds_width;
  P.bounds_height = bounds_height;
  P.patches = nullptr;
  P.patch_capacity = 0;
  P.patch_count = nullptr;
  P.bounds = nullptr;
  return VK_OK;
}

// PatchParams of the view `frame` of volume `v` with the tracer settings in `ahead`
inline int view_patch_params(PatchParams& P, const vk_volume* v, const vk_frame* frame, const vk_view_bounds* ahead)
{
  vk_transform Tcw;
  for (int i = 0; i < 16; ++i) { Tcw.m[i] = frame->depth_to_world.inv[i]; Tcw.inv[i] = frame->depth_to_world.m[i]; }
  return fill_patch_params(P, v->visible_blocks, v->hash_entries, &Tcw, &frame->depth_projection,
      VK_BLOCK_RESOLUTION * v->voxel_length, ahead->min_depth, ahead->max_depth,
      v->main_block_count + v->excess_block_count, v->counters + VK_CTR_VISIBLE, frame->width,
      frame->height, ahead->bounds_width, ahead->bounds_height);
}

inline bool view_matches(const vk_view_bounds* ahead, const vk_volume* v, const vk_frame* frame)
{
  return ahead->valid && ahead->width == frame->width && ahead->height == frame->height &&
         ahead->block_length == VK_BLOCK_RESOLUTION * v->voxel_length &&
         ahead->visible_blocks == (const void*)v->visible_blocks &&
         std::memcmp(&ahead->projection, &frame->depth_projection, sizeof(vk_projection)) == 0 &&
         std::memcmp(&ahead->depth_to_world, &frame->depth_to_world, sizeof(vk_transform)) == 0;
}

inline void view_record(vk_view_bounds* ahead, const vk_volume* v, const vk_frame* frame)
{
  ahead->width = frame->width;
  ahead->height = frame->height;
  ahead->block_length = VK_BLOCK_RESOLUTION * v->voxel_length;
  ahead->visible_blocks = v->visible_blocks;
  ahead->projection = frame->depth_projection;
  ahead->depth_to_world = frame->depth_to_world;
  ahead->valid = 1;
}

}  // namespace vk
