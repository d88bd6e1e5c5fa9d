// vk_integrate.hip — TSDF / colour / light integration for gfx950
// (ref: src/depth_integrator.cu, src/color_integrator.cu, src/light_integrator.cu).
//
// Layout: the 20-byte AoS Voxel is pinned by the API (Volume::GetVoxels), so a
// block is 512*20 = 10 240 contiguous bytes = 640 float4. ONE WAVEFRONT owns one
// visible block: it pulls the block into LDS with ten coalesced 1-KiB float4
// loads, updates its eight z-slices (lane = y*8+x) out of LDS — a 5-dword
// voxel stride is conflict-free on the 32-bank ds_read_b32 path — and streams
// the tile back with ten float4 stores. The reference runs one 512-thread CUDA
// block per voxel block with 5-dword-strided per-thread global accesses, and a
// second full pass for colour; here depth+colour are one pass.
//
// HBM-bound: algorithmic bytes per visible block = 4 (index) + 16 (entry) +
// 2*10240 (voxel read + write) = 20 500 B, plus the images once per frame
// (SURVEY.md §8d).
#include "vk_bounds.hpp"

#ifndef VK_INTEGRATE_NT_STORES
#define VK_INTEGRATE_NT_STORES 0
#endif

using namespace vk;

namespace
{

enum { COLOR_NONE = 0, COLOR_PLAIN = 1, COLOR_LIGHT = 2 };

struct IntegrateParams
{
  float4* voxels4;               // pool viewed as float4
  const vk_hash_entry* entries;
  const int32_t* visible;
  const int32_t* counters;
  const float* depth;
  const float* color;
  const float* normals;
  const float* mask;
  int width, height;      // depth image
  int cwidth, cheight;    // colour image (color_integrator.cu:183-184)
  vk_projection kd, kc;
  Rt Tdw, Tcw, Tcd;
  vk_light light;
  float voxel_length, block_length, truncation_length;
  float min_depth, max_depth;
  float max_distance_weight, max_color_weight;
};

constexpr int kTileF4 = 640;  // float4 per voxel block

// light.h:53-60
__device__ __forceinline__ float light_shading(const vk_light& l, f3 point, f3 normal)
{
  const f3 delta = sub3(make3(l.position[0], l.position[1], l.position[2]), point);
  const f3 direction = normalized3(delta);
  const float distance_squared = sqnorm3(delta);
  const float cos_theta = dot3(normal, direction);
  return l.intensity * cos_theta / distance_squared;
}

// ---------------------------------------------------------------------------
// Software pipeline. Moving a whole block per step makes every wave of a CU load,
// then compute, then store at about the same time, so the ~10 us of per-voxel
// arithmetic does not overlap with the ~15 us of data movement (r01 ablations,
// DESIGN.md section 4). The unit of work is therefore HALF a block (four z
// slices = 5 KiB = 320 float4, five float4 per lane) and a wave keeps two units
// in flight: while unit s is updated out of LDS, the tile loads and the depth
// gathers of unit s+1 are already on their way into the other register set.
// ---------------------------------------------------------------------------

constexpr int kHalfF4 = 320;          // float4 per half block
constexpr bool g_nt_stores = VK_INTEGRATE_NT_STORES;
typedef float nf4 __attribute__((ext_vector_type(4)));
constexpr int kPipeWavesPerGroup = 4;

// Per-unit state. It is deliberately a bundle of separate locals passed by
// reference (not a struct with array members): hipcc's SROA left the struct form
// in scratch memory, which serialised every tile load behind a scratch store.
#define UNIT_DECL(U)                                                                      \
  float4 U##_r0, U##_r1, U##_r2, U##_r3, U##_r4;   /* the half tile, five float4 per lane */ \
  float U##_depth[4], U##_z[4], U##_du[4], U##_dv[4];                                      \
  uint32_t U##_valid = 0;                                                                  \
  float4* U##_base = nullptr;                                                              \
  bool U##_skip = true;                                                                    \
  f3 U##_off = make3(0, 0, 0);                                                             \
  int U##_half = 0
#define UNIT_ARGS(U) U##_r0, U##_r1, U##_r2, U##_r3, U##_r4, U##_depth, U##_z, U##_du, U##_dv, U##_valid, \
  U##_base, U##_skip, U##_off, U##_half
#define UNIT_PARAMS float4& r0, float4& r1, float4& r2, float4& r3, float4& r4, float (&u_depth)[4],       \
  float (&u_z)[4], float (&u_du)[4], float (&u_dv)[4], uint32_t& u_valid, float4*& u_base, bool& u_skip,  \
  f3& u_off, int& u_half

// unit s of the wave = half (s & 1) of its (s >> 1)-th block, whose hash entry sits
// in lane (s >> 1) of my_entry
template <bool DEPTH, int COLOR>
__device__ __forceinline__ void unit_issue(const IntegrateParams& P, int4 my_entry, int s, int lane, UNIT_PARAMS)
{
  const int j = s >> 1, half = s & 1;
  const int e0 = __builtin_amdgcn_readlane(my_entry.x, j);
  const int e1 = __builtin_amdgcn_readlane(my_entry.y, j);
  const int data = __builtin_amdgcn_readlane(my_entry.z, j);
  const int ox = (int16_t)(e0 & 0xffff), oy = (int16_t)((uint32_t)e0 >> 16), oz = (int16_t)(e1 & 0xffff);
  const int vx = lane & 7, vy = lane >> 3;
  u_half = half;
  u_off = scale3(make3((float)ox, (float)oy, (float)oz), P.block_length);
  // the never-allocated origin block (data == -1) reads slot 0 and is
  // flagged so that unit_update writes nothing
  u_skip = data < 0;
  u_base = P.voxels4 + (size_t)(data < 0 ? 0 : data) * kTileF4 + half * kHalfF4;
  u_valid = 0;

  if (DEPTH || COLOR == COLOR_LIGHT)
  {
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
      const int vz = half * 4 + k;
      const f3 voxel_offset = scale3(make3(vx + 0.5f, vy + 0.5f, vz + 0.5f), P.voxel_length);
      const f3 Xdp = xform_point(P.Tdw, add3(u_off, voxel_offset));
      project(P.kd, Xdp, u_du[k], u_dv[k]);
      u_z[k] = Xdp.z;
      const bool valid = (u_du[k] >= 0) & (u_du[k] < P.width) & (u_dv[k] >= 0) & (u_dv[k] < P.height);
      u_valid |= (valid ? 1u : 0u) << k;
      u_depth[k] = 0.0f;
      if (DEPTH)
      {
        // 32-bit unsigned index: a signed 64-bit mad here made hipcc read a register
        // pair whose upper half was a pending gather result (forced s_waitcnt vmcnt(0))
        // (v_mad_u32_u24: row and width are < 2^24, checked on the host)
        const uint32_t pixel = valid ? __umul24((uint32_t)f2i(u_dv[k]), (uint32_t)P.width) + (uint32_t)f2i(u_du[k]) : 0u;
        u_depth[k] = P.depth[pixel];
      }
    }
  }

  r0 = u_base[0 * 64 + lane];
  r1 = u_base[1 * 64 + lane];
  r2 = u_base[2 * 64 + lane];
  r3 = u_base[3 * 64 + lane];
  r4 = u_base[4 * 64 + lane];
}

template <bool DEPTH, int COLOR>
__device__ __forceinline__ void unit_update(const IntegrateParams& P, int lane, float4* tile4, UNIT_PARAMS)
{
  float* tile = reinterpret_cast<float*>(tile4);
  const int vx = lane & 7, vy = lane >> 3;
  if (u_skip) return;

  tile4[0 * 64 + lane] = r0;
  tile4[1 * 64 + lane] = r1;
  tile4[2 * 64 + lane] = r2;
  tile4[3 * 64 + lane] = r3;
  tile4[4 * 64 + lane] = r4;
  wave_lds_fence();   // float4-per-lane layout written, voxel-per-lane layout read

  float old_d[4];
  uint32_t old_w[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
  {
    const float* vox = tile + (k * 64 + lane) * 5;
    old_d[k] = vox[0];
    old_w[k] = __float_as_uint(vox[4]);
  }

  bool dirty = false;
  float dist[4];   // the voxel's current distance after the depth pass (or as stored)

#pragma unroll
  for (int k = 0; k < 4; ++k)
  {
    float* vox = tile + (k * 64 + lane) * 5;
    dist[k] = old_d[k];
    const bool valid = (u_valid >> k) & 1u;

    if (DEPTH)
    {
      // depth_integrator.cu:54-78, evaluated branch-free: the new value is computed
      // for every voxel and selected, so the four division chains of a lane's
      // voxels interleave instead of running one after another behind branches.
      const float depth = u_depth[k];
      const float distance = depth - u_z[k];
      const bool update = valid & !((depth < P.min_depth) | (depth > P.max_depth)) & (distance > -P.truncation_length);
      const uint32_t weights = old_w[k];
      const int16_t dw = (int16_t)(weights & 0xffff);
      const float prev_dist = dw * old_d[k];
      const float curr_dist = vmin(1.0f, distance / P.truncation_length);
      const float dist_weight = dw + 1;
      const int16_t new_dw = (int16_t)vmin(P.max_distance_weight, dist_weight);
      const float new_distance = (prev_dist + curr_dist) / dist_weight;
      dist[k] = update ? new_distance : old_d[k];
      old_w[k] = update ? ((weights & 0xffff0000u) | (uint16_t)new_dw) : weights;
      if (update)
      {
        vox[0] = dist[k];
        vox[4] = __uint_as_float(old_w[k]);
      }
      dirty |= update;
    }
  }

  if (COLOR != COLOR_NONE)
  {
    // Colour pass in three sweeps over the lane's four voxels, so that the image
    // gathers of all four are in flight together instead of one dependent chain per
    // voxel: (1) which voxels take colour and from which pixel, (2) the gathers,
    // (3) the running averages. Conditions are pure, so testing |dist| < 1 before the
    // mask (the reference tests the mask first) selects the same voxels.
    f3 Xcp[4];
    int color_index[4], depth_index[4];
    bool want[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
      const int vz = u_half * 4 + k;
      const f3 voxel_offset = scale3(make3(vx + 0.5f, vy + 0.5f, vz + 0.5f), P.voxel_length);
      Xcp[k] = xform_point(P.Tcw, add3(u_off, voxel_offset));
      float cu, cv;
      project(P.kc, Xcp[k], cu, cv);
      const bool color_valid = cu >= 0 && cu < P.cwidth && cv >= 0 && cv < P.cheight;
      const bool valid = (u_valid >> k) & 1u;
      want[k] = color_valid && fabsf(dist[k]) < 1.0f && (COLOR == COLOR_PLAIN || valid);
      color_index[k] = want[k] ? (int)cv * P.cwidth + (int)cu : 0;
      depth_index[k] = (COLOR == COLOR_LIGHT && want[k]) ? (int)u_dv[k] * P.width + (int)u_du[k] : 0;
    }

    if (COLOR == COLOR_LIGHT)
    {
      float m[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) m[k] = want[k] ? P.mask[depth_index[k]] : 0.0f;
#pragma unroll
      for (int k = 0; k < 4; ++k) want[k] = want[k] && m[k] > 0.5f;
    }

    f3 pixel_color[4], pixel_normal[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
      pixel_color[k] = make3(0, 0, 0);
      pixel_normal[k] = make3(0, 0, 0);
      if (want[k])
      {
        pixel_color[k] = make3(P.color[3 * color_index[k] + 0], P.color[3 * color_index[k] + 1],
            P.color[3 * color_index[k] + 2]);
        if (COLOR == COLOR_LIGHT)
          pixel_normal[k] = make3(P.normals[3 * depth_index[k] + 0], P.normals[3 * depth_index[k] + 1],
              P.normals[3 * depth_index[k] + 2]);
      }
    }

#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
      if (!want[k]) continue;
      float* vox = tile + (k * 64 + lane) * 5;
      f3 curr_color = pixel_color[k];

      if (COLOR == COLOR_LIGHT)
      {
        // light_integrator.cu:197-248
        const f3 Xcn = xform_dir(P.Tcd, pixel_normal[k]);
        const float shading = light_shading(P.light, Xcp[k], Xcn);
        if (!(shading > 0.05f)) continue;
        curr_color = div3(curr_color, shading);
      }

      // color_integrator.cu:100-134 / light_integrator.cu:233-246
      const uint32_t weights = old_w[k];
      const int16_t cw = (int16_t)(weights >> 16);
      const float cwf = cw;
      const f3 prev_color = scale3(make3(vox[1], vox[2], vox[3]), cwf);
      const float color_weight = cw + 1;
      const int16_t new_cw = (int16_t)vmin(P.max_color_weight, color_weight);
      const f3 c = div3(add3(prev_color, curr_color), color_weight);
      vox[1] = c.x;
      vox[2] = c.y;
      vox[3] = c.z;
      vox[4] = __uint_as_float((weights & 0x0000ffffu) | ((uint32_t)(uint16_t)new_cw << 16));
      dirty = true;
    }
  }

  if (__any(dirty))
  {
    // Write back only what differs from what was read: in steady state about a
    // third of a visible block is bit-for-bit unchanged (voxels behind the band are
    // never touched; voxels in front of it sit at distance 1 with a saturated
    // weight and are re-written with the same value), and a 64-byte line that no
    // lane stores to stays clean in L2 and is never written to HBM.
    float4 out[5];
    wave_lds_fence();   // other lanes' voxels make up this lane's float4s
#pragma unroll
    for (int k = 0; k < 5; ++k) out[k] = tile4[k * 64 + lane];
    const float4 was[5] = {r0, r1, r2, r3, r4};
#pragma unroll
    for (int k = 0; k < 5; ++k)
    {
      const uint32_t differs = (__float_as_uint(out[k].x) ^ __float_as_uint(was[k].x)) |
                               (__float_as_uint(out[k].y) ^ __float_as_uint(was[k].y)) |
                               (__float_as_uint(out[k].z) ^ __float_as_uint(was[k].z)) |
                               (__float_as_uint(out[k].w) ^ __float_as_uint(was[k].w));
      if (differs == 0) continue;
      if (g_nt_stores)
      {
        nf4 t; t.x = out[k].x; t.y = out[k].y; t.z = out[k].z; t.w = out[k].w;
        __builtin_nontemporal_store(t, reinterpret_cast<nf4*>(&u_base[k * 64 + lane]));
      }
      else u_base[k * 64 + lane] = out[k];
    }
  }
}

// What the first kBoundsGroups workgroups of an AHEAD launch do instead of
// integrating: the raycast bounds of the same view (vk_bounds.hpp). They only read
// the visible list and the hash entries, so they run alongside the integrate
// workgroups of the same launch for free.
struct AheadParams
{
  PatchParams patch;
  float2* partials;
};

template <bool DEPTH, int COLOR, bool AHEAD>
__global__ __launch_bounds__(kPipeWavesPerGroup * 64) void integrate_pipelined_kernel(IntegrateParams P, AheadParams A)
{
  // one LDS pool: four half-block tiles (20 KiB), or one bounds grid (37.5 KiB)
  constexpr int kTileInts = kPipeWavesPerGroup * kHalfF4 * 4;
  constexpr int kPoolInts = AHEAD ? (2 * kAheadMaxCells > kTileInts ? 2 * kAheadMaxCells : kTileInts) : kTileInts;
  __shared__ __attribute__((aligned(16))) int pool[kPoolInts];

  int group = (int)blockIdx.x, groups = (int)gridDim.x;
  if (AHEAD)
  {
    if (group < kBoundsGroups)
    {
      bounds_group(A.patch, A.partials, pool, group, kPipeWavesPerGroup * 64);
      return;
    }
    group -= kBoundsGroups;
    groups -= kBoundsGroups;
  }

  const int lane = lane_id();
  const int wave_in_group = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wave = group * kPipeWavesPerGroup + wave_in_group;
  const int total_waves = groups * kPipeWavesPerGroup;
  const int count = P.counters[VK_CTR_VISIBLE];
  float4* tile4 = reinterpret_cast<float4*>(pool) + wave_in_group * kHalfF4;

  for (int first = wave; first < count; first += 64 * total_waves)
  {
    // lane j holds the hash entry of this wave's j-th block of the group
    const int mine = first + lane * total_waves;
    int4 my_entry = make_int4(0, 0, -1, -1);
    if (mine < count) my_entry = reinterpret_cast<const int4*>(P.entries)[P.visible[mine]];
    int blocks = (count - first + total_waves - 1) / total_waves;
    if (blocks > 64) blocks = 64;
    const int units = 2 * blocks;   // unit s = (block s >> 1, half s & 1)

    UNIT_DECL(A);
    UNIT_DECL(B);
    unit_issue<DEPTH, COLOR>(P, my_entry, 0, lane, UNIT_ARGS(A));
    int s = 0;
    for (; s + 2 < units; s += 2)   // steady state: two units per trip, next one always in flight
    {
      unit_issue<DEPTH, COLOR>(P, my_entry, s + 1, lane, UNIT_ARGS(B));
      unit_update<DEPTH, COLOR>(P, lane, tile4, UNIT_ARGS(A));
      unit_issue<DEPTH, COLOR>(P, my_entry, s + 2, lane, UNIT_ARGS(A));
      unit_update<DEPTH, COLOR>(P, lane, tile4, UNIT_ARGS(B));
    }
    // units is even and >= 2: exactly two remain (s, s + 1)
    unit_issue<DEPTH, COLOR>(P, my_entry, s + 1, lane, UNIT_ARGS(B));
    unit_update<DEPTH, COLOR>(P, lane, tile4, UNIT_ARGS(A));
    unit_update<DEPTH, COLOR>(P, lane, tile4, UNIT_ARGS(B));
  }
}

int fill_params(IntegrateParams& P, const vk_volume* v, const vk_integrator* p, const vk_frame* f,
    const vk_light* light, const float* mask, bool need_depth, bool need_color, bool need_light)
{
  if (!v || !p || !f) return VK_ERR_ARGUMENT;
  if (!v->voxels || !v->hash_entries || !v->visible_blocks || !v->counters) return VK_ERR_ARGUMENT;
  if (reinterpret_cast<uintptr_t>(v->voxels) & 15) return VK_ERR_ARGUMENT;
  if (f->width <= 0 || f->height <= 0 || f->width >= (1 << 24) || f->height >= (1 << 24)) return VK_ERR_ARGUMENT;
  if (need_depth && !f->depth) return VK_ERR_ARGUMENT;
  if (need_color && !f->color) return VK_ERR_ARGUMENT;
  if (need_light && (!f->normals || !mask || !light)) return VK_ERR_ARGUMENT;
  const int cwidth = f->color_width > 0 ? f->color_width : f->width;
  const int cheight = f->color_height > 0 ? f->color_height : f->height;
  if (cwidth >= (1 << 24) || cheight >= (1 << 24)) return VK_ERR_ARGUMENT;
  // light_integrator.cu:333-334 indexes the colour image with the depth image's size
  if (need_light && (cwidth != f->width || cheight != f->height)) return VK_ERR_ARGUMENT;

  P.voxels4 = reinterpret_cast<float4*>(v->voxels);
  P.entries = v->hash_entries;
  P.visible = v->visible_blocks;
  P.counters = v->counters;
  P.depth = f->depth;
  P.color = f->color;
  P.normals = f->normals;
  P.mask = mask;
  P.width = f->width;
  P.height = f->height;
  P.cwidth = cwidth;
  P.cheight = cheight;
  P.kd = f->depth_projection;
  P.kc = f->color_projection;
  P.Tdw = make_rt(f->depth_to_world.inv);  // Twd.Inverse(), depth_integrator.cu:104

  // Tcw = Tcd * Tdw (color_integrator.cu:192, light_integrator.cu:337-338),
  // matrix.h:297-318 product order
  float Tcw[16];
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r)
    {
      float acc = 0.0f;
      for (int n = 0; n < 4; ++n) acc += f->depth_to_color.m[n * 4 + r] * f->depth_to_world.inv[c * 4 + n];
      Tcw[c * 4 + r] = acc;
    }
  P.Tcw = make_rt(Tcw);
  P.Tcd = make_rt(f->depth_to_color.m);
  if (light) P.light = *light; else { P.light.intensity = 1.0f; P.light.position[0] = P.light.position[1] = P.light.position[2] = 0.0f; }
  P.voxel_length = v->voxel_length;
  P.block_length = VK_BLOCK_RESOLUTION * v->voxel_length;
  P.truncation_length = v->truncation_length;
  P.min_depth = p->min_depth;
  P.max_depth = p->max_depth;
  P.max_distance_weight = p->max_distance_weight;
  P.max_color_weight = p->max_color_weight;
  return VK_OK;
}

// Persistent grid. 20 KiB of LDS per workgroup: up to 8 workgroups (32 waves) per CU
// by LDS, the register budget decides; sized for 5 workgroups per CU (r01 sweep of
// 2..8: profiles/r01_h_stage_timings_and_ablations.txt) and capped by the largest
// possible visible count so small volumes do not launch idle workgroups.
int pipe_grid_for(const vk_volume* v, int groups_per_cu)
{
  const int max_count = v->main_block_count + v->excess_block_count;
  const int want = (max_count + kPipeWavesPerGroup - 1) / kPipeWavesPerGroup;
  const int cap = kCUs * groups_per_cu;
  return want < cap ? (want > 0 ? want : 1) : cap;
}

// `ahead` (optional): also compute the raycast bounds of the frame's own view
template <bool DEPTH, int COLOR>
int launch(const IntegrateParams& P, const vk_volume* v, const vk_frame* frame, vk_view_bounds* ahead, hipStream_t s)
{
  AheadParams A;
  A.partials = nullptr;
  bool with_bounds = false;
  if (ahead && ahead->scratch && ahead->bounds_width > 0 && ahead->bounds_height > 0 &&
      ahead->bounds_width * ahead->bounds_height <= kAheadMaxCells)
  {
    ahead->valid = 0;
    if (view_patch_params(A.patch, v, frame, ahead) != VK_OK) return VK_ERR_ARGUMENT;
    A.partials = reinterpret_cast<float2*>(ahead->scratch) + ahead->bounds_width * ahead->bounds_height;
    with_bounds = true;
  }

  if (with_bounds)
  {
    // 37.5 KiB of LDS per workgroup: four per CU
    const int grid = pipe_grid_for(v, 4) + kBoundsGroups;
    hipLaunchKernelGGL((integrate_pipelined_kernel<DEPTH, COLOR, true>), dim3(grid),
        dim3(kPipeWavesPerGroup * 64), 0, s, P, A);
  }
  else
    hipLaunchKernelGGL((integrate_pipelined_kernel<DEPTH, COLOR, false>), dim3(pipe_grid_for(v, 5)),
        dim3(kPipeWavesPerGroup * 64), 0, s, P, A);
  VK_LAUNCH_CHECK();
  if (with_bounds) view_record(ahead, v, frame);
  return VK_OK;
}

// ---------------------------------------------------------------- frame mask ----

// ref: light_integrator.cu:17-103 ComputeFrameMaskKernel<16,3>. The reference
// stages a 22x22 tile with a -1 halo offset and reads it with a +3 centre, so
// pixel (x,y) looks at [x-1,x+5] x [y-1,y+5]; kept as is.
__global__ __launch_bounds__(256) void frame_mask_kernel(int width, int height,
    const float* __restrict__ depths, const float* __restrict__ colors, float depth_threshold,
    float* __restrict__ mask)
{
  constexpr int BD = 16, KS = 3, DIM = BD + 2 * KS;
  __shared__ float buffer[DIM * DIM];

  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int x = blockIdx.x * BD + tx;
  const int y = blockIdx.y * BD + ty;

  for (int sindex = threadIdx.x; sindex < DIM * DIM; sindex += 256)
  {
    float depth = 0;
    const int vx = (blockIdx.x * BD - 1) + (sindex % DIM);
    const int vy = (blockIdx.y * BD - 1) + (sindex / DIM);
    if (vx >= 0 && vx < width && vy >= 0 && vy < height) depth = depths[vy * width + vx];
    buffer[sindex] = depth;
  }

  __syncthreads();

  if (x < width && y < height)
  {
    const int index = y * width + x;
    const float c0 = colors[3 * index + 0], c1 = colors[3 * index + 1], c2 = colors[3 * index + 2];

    if (c0 < 0.02f || c0 > 0.98f || c1 < 0.02f || c1 > 0.98f || c2 < 0.02f || c2 > 0.98f)
    {
      mask[index] = 0.0f;
      return;
    }

    float dmin = +FLT_MAX;
    float dmax = -FLT_MAX;
    const int cx = tx + KS;
    const int cy = ty + KS;

    for (int i = -KS; i <= KS; ++i)
      for (int j = -KS; j <= KS; ++j)
      {
        const float depth = buffer[(cy + i) * DIM + (cx + j)];
        dmin = fminf(depth, dmin);
        dmax = fmaxf(depth, dmax);
      }

    mask[index] = (dmax - dmin <= depth_threshold) ? 1.0f : 0.0f;
  }
}

}  // namespace

extern "C" {

int vk_integrate_depth(const vk_volume* v, const vk_integrator* p, const vk_frame* frame, void* stream)
{
  IntegrateParams P;
  const int rc = fill_params(P, v, p, frame, nullptr, nullptr, true, false, false);
  if (rc != VK_OK) return rc;
  return launch<true, COLOR_NONE>(P, v, frame, nullptr, vk_s(stream));
}

int vk_integrate_color(const vk_volume* v, const vk_integrator* p, const vk_frame* frame, void* stream)
{
  IntegrateParams P;
  const int rc = fill_params(P, v, p, frame, nullptr, nullptr, false, true, false);
  if (rc != VK_OK) return rc;
  return launch<false, COLOR_PLAIN>(P, v, frame, nullptr, vk_s(stream));
}

int vk_integrate_depth_color(const vk_volume* v, const vk_integrator* p, const vk_frame* frame, void* stream)
{
  IntegrateParams P;
  const int rc = fill_params(P, v, p, frame, nullptr, nullptr, true, true, false);
  if (rc != VK_OK) return rc;
  return launch<true, COLOR_PLAIN>(P, v, frame, nullptr, vk_s(stream));
}

int vk_light_compute_frame_mask(const vk_frame* frame, float depth_threshold, float* mask, void* stream)
{
  VK_REQUIRE(frame && frame->depth && frame->color && mask && frame->width > 0 && frame->height > 0);
  // light_integrator.cu:277-293 walks the colour image with the depth image's size
  VK_REQUIRE((frame->color_width <= 0 || frame->color_width == frame->width) &&
             (frame->color_height <= 0 || frame->color_height == frame->height));
  const dim3 grid((frame->width + 15) / 16, (frame->height + 15) / 16);
  hipLaunchKernelGGL(frame_mask_kernel, grid, dim3(256), 0, vk_s(stream), frame->width, frame->height,
      frame->depth, frame->color, depth_threshold, mask);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

int vk_integrate_light_color(const vk_volume* v, const vk_integrator* p, const vk_light* light,
    const float* mask, const vk_frame* frame, void* stream)
{
  IntegrateParams P;
  const int rc = fill_params(P, v, p, frame, light, mask, false, true, true);
  if (rc != VK_OK) return rc;
  return launch<false, COLOR_LIGHT>(P, v, frame, nullptr, vk_s(stream));
}

int vk_integrate_depth_light(const vk_volume* v, const vk_integrator* p, const vk_light* light,
    const float* mask, const vk_frame* frame, void* stream)
{
  IntegrateParams P;
  const int rc = fill_params(P, v, p, frame, light, mask, true, true, true);
  if (rc != VK_OK) return rc;
  return launch<true, COLOR_LIGHT>(P, v, frame, nullptr, vk_s(stream));
}

int vk_integrate_ahead(const vk_volume* v, const vk_integrator* p, const vk_frame* frame, int color_mode,
    const vk_light* light, const float* mask, vk_view_bounds* ahead, void* stream)
{
  IntegrateParams P;
  VK_REQUIRE(color_mode >= 0 && color_mode <= 2);
  const int rc = fill_params(P, v, p, frame, color_mode == 2 ? light : nullptr, color_mode == 2 ? mask : nullptr, true,
      color_mode != 0, color_mode == 2);
  if (rc != VK_OK) return rc;
  if (color_mode == 0) return launch<true, COLOR_NONE>(P, v, frame, ahead, vk_s(stream));
  if (color_mode == 1) return launch<true, COLOR_PLAIN>(P, v, frame, ahead, vk_s(stream));
  return launch<true, COLOR_LIGHT>(P, v, frame, ahead, vk_s(stream));
}

}  // extern "C"
