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

#include <cstring>

#ifndef VK_INTEGRATE_NT_STORES
#define VK_INTEGRATE_NT_STORES 0
#endif
#ifndef VK_INTEGRATE_PLAIN_LIST
#define VK_INTEGRATE_PLAIN_LIST 0   // 1: never use the banded visible lists (A/B measurements, tools/build_variant.sh)
#endif

// Four waves per SIMD (<= 128 VGPRs) for every variant: the RGB-D kernel is bound by the
// latency of its image gathers and division chains, and ran 42 us at three waves
// (155 VGPRs), 39 us at four (r02 variants D/E/F, profiles/r02_b_light_variants.txt).
// The register allocator reaches 128 by spilling two dwords outside the voxel loop.
#define VK_INTEGRATE_WAVES __attribute__((amdgpu_waves_per_eu(4)))
// voxels whose light-model colour update is in flight together (see unit_update)
constexpr int kLightGroup = 2;

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
  const float* records;   // optional: per depth pixel {Tcd * normal, mask} (vk_light_prepare)
  bool same_camera;       // Tcw == Tdw, kc == kd, colour size == depth size
  int width, height;      // depth image
  int cwidth, cheight;    // colour image (color_integrator.cu:183-184)
  vk_projection kd, kc;
  Rt Tdw, Tcw, Tcd;
  vk_light light;
  float voxel_length, block_length, truncation_length;
  double inv_truncation_length;   // RN64(1 / truncation_length), see exact_quotient
  float min_depth, max_depth;
  float max_distance_weight, max_color_weight;
};

constexpr int kTileF4 = 640;  // float4 per voxel block

// a / b for binary32 a, b, correctly rounded, from inv_b = RN64(1 / b): the double product is
// within 2^-52 (relative) of a / b and a binary32 quotient is never within 2^-49 of the midpoint
// of two binary32 neighbours, so rounding it gives exactly RN32(a / b) — the value the
// reference's `a / b` has (vk_raycast.hpp div_uniform has the argument in full). Three
// instructions instead of the ten of the division expansion. The same argument makes
// (float)RN64(1 / b) equal to RN32(1 / b), the reference's `1.0f / b`.
__device__ __forceinline__ float exact_quotient(float a, double inv_b) { return (float)((double)a * inv_b); }

// The running averages divide by a weight: a small integer (the weights are capped at 16 by
// default, integrator.cu:7-13; the shipped app caps the distance weight at 100,
// apps/vulcan/vulcan.cu:92). RN64(1 / n) for n < kReciprocals sits in LDS, filled once per
// workgroup with the correctly rounded double division; a wave that meets a larger weight takes
// the plain division (tests/test_gpu_weights.py runs both sides of that test).
constexpr int kReciprocals = 129;   // weights 0 .. 127 divide by 1 .. 128
constexpr uint32_t kLargeWeightBits = 0xff80ff80u;   // set in a weight pair when either is negative or >= 128
// one byte per 16-byte piece of a half tile
constexpr int kChangedBytes = 320;

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
  float U##_depth[4], U##_z[4];                                                            \
  uint32_t U##_pix[4];   /* the voxel's pixel in the depth image, 0 when it has none */     \
  uint32_t U##_valid = 0;                                                                  \
  float4* U##_base = nullptr;                                                              \
  bool U##_skip = true;                                                                    \
  f3 U##_off = make3(0, 0, 0);                                                             \
  int U##_half = 0
#define UNIT_ARGS(U) U##_r0, U##_r1, U##_r2, U##_r3, U##_r4, U##_depth, U##_z, U##_pix, U##_valid, \
  U##_base, U##_skip, U##_off, U##_half
#define UNIT_PARAMS float4& r0, float4& r1, float4& r2, float4& r3, float4& r4, float (&u_depth)[4],       \
  float (&u_z)[4], uint32_t (&u_pix)[4], uint32_t& u_valid, float4*& u_base, bool& u_skip,  \
  f3& u_off, int& u_half

// unit s of the wave = half (s & 1) of its (s >> 1)-th block, whose hash entry sits
// in lane (s >> 1) of my_entry
template <bool DEPTH, int COLOR, bool SAME_CAM>
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

  if (DEPTH || COLOR == COLOR_LIGHT || (SAME_CAM && COLOR != COLOR_NONE))
  {
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
      const int vz = half * 4 + k;
      const f3 voxel_offset = scale3(make3(vx + 0.5f, vy + 0.5f, vz + 0.5f), P.voxel_length);
      const f3 Xdp = xform_point(P.Tdw, add3(u_off, voxel_offset));
      float du, dv;
      project(P.kd, Xdp, du, dv);
      u_z[k] = Xdp.z;
      const bool valid = (du >= 0) & (du < P.width) & (dv >= 0) & (dv < P.height);
      u_valid |= (valid ? 1u : 0u) << k;
      // 32-bit unsigned index: a signed 64-bit mad here made hipcc read a register
      // pair whose upper half was a pending gather result (forced s_waitcnt vmcnt(0))
      // (v_mad_u32_u24: row and width are < 2^24, checked on the host)
      u_pix[k] = valid ? __umul24((uint32_t)f2i(dv), (uint32_t)P.width) + (uint32_t)f2i(du) : 0u;
      u_depth[k] = 0.0f;
      if (DEPTH) u_depth[k] = P.depth[u_pix[k]];
    }
  }

  r0 = u_base[0 * 64 + lane];
  r1 = u_base[1 * 64 + lane];
  r2 = u_base[2 * 64 + lane];
  r3 = u_base[3 * 64 + lane];
  r4 = u_base[4 * 64 + lane];
}

template <bool DEPTH, int COLOR, bool SAME_CAM, bool RECORDS>
__device__ __forceinline__ void unit_update(const IntegrateParams& P, int lane, float4* tile4, uint8_t* changed,
    const double* reciprocal, UNIT_PARAMS)
{
  float* tile = reinterpret_cast<float*>(tile4);
  const int vx = lane & 7, vy = lane >> 3;
  if (u_skip) return;

  tile4[0 * 64 + lane] = r0;
  tile4[1 * 64 + lane] = r1;
  tile4[2 * 64 + lane] = r2;
  tile4[3 * 64 + lane] = r3;
  tile4[4 * 64 + lane] = r4;
  // one byte per 16-byte piece of the half tile: set by the lane that changes a dword
  // of the piece, read at write-back by the lane that owns the piece
  if (lane < kChangedBytes / 8) reinterpret_cast<uint2*>(changed)[lane] = make_uint2(0u, 0u);
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

  // every weight of the wave's voxels below kReciprocals - 1 (bits 7..14 of both 16-bit halves clear;
  // a negative weight has bit 15 set): the divisions by weight + 1 use the reciprocal table
  const bool small_weights = !__any(((old_w[0] | old_w[1] | old_w[2] | old_w[3]) & kLargeWeightBits) != 0u);

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
      const float curr_dist = vmin(1.0f, exact_quotient(distance, P.inv_truncation_length));
      const float dist_weight = dw + 1;
      const int16_t new_dw = (int16_t)vmin(P.max_distance_weight, dist_weight);
      float new_distance;
      if (small_weights) new_distance = exact_quotient(prev_dist + curr_dist, reciprocal[dw + 1]);
      else new_distance = (prev_dist + curr_dist) / dist_weight;
      dist[k] = update ? new_distance : old_d[k];
      old_w[k] = update ? ((weights & 0xffff0000u) | (uint16_t)new_dw) : weights;
      if (update)
      {
        vox[0] = dist[k];
        vox[4] = __uint_as_float(old_w[k]);
        // which 16-byte pieces now differ from what was read (dword j of voxel v sits in
        // piece (5 v + j) / 4): a voxel in front of the band sits at distance 1 with a
        // saturated weight and is "updated" to the very same bits
        const int dword = (k * 64 + lane) * 5;
        if (__float_as_uint(dist[k]) != __float_as_uint(old_d[k])) { changed[dword >> 2] = 1; dirty = true; }
        if (old_w[k] != weights) { changed[(dword + 4) >> 2] = 1; dirty = true; }
      }
    }
  }

  if (COLOR != COLOR_NONE)
  {
    // Colour pass in three sweeps over the lane's four voxels, so that the image
    // gathers of all four are in flight together instead of one dependent chain per
    // voxel: (1) which voxels take colour and from which pixel, (2) the gathers,
    // (3) the running averages. Conditions are pure, so testing |dist| < 1 before the
    // mask (the reference tests the mask first) selects the same voxels.
    // G voxels at a time: per voxel the light model needs a 16-byte record, a colour, a
    // normalisation and three divisions, ~35 registers; two at a time keep the kernel at
    // 128 VGPRs = four waves per SIMD. The plain colour pass takes all four together.
    constexpr int G = (COLOR == COLOR_LIGHT) ? kLightGroup : 4;
#pragma unroll
    for (int g0 = 0; g0 < 4; g0 += G)
    {
      if (g0 > 0) __builtin_amdgcn_sched_barrier(0);   // keep the next group's gathers below this group's work

      f3 Xcp[G];
      uint32_t color_index[G], depth_index[G];
      bool want[G];
#pragma unroll
      for (int i = 0; i < G; ++i)
      {
        const int k = g0 + i;
        const int vz = u_half * 4 + k;
        const bool valid = (u_valid >> k) & 1u;
        const f3 voxel_offset = scale3(make3(vx + 0.5f, vy + 0.5f, vz + 0.5f), P.voxel_length);
        if (SAME_CAM)
        {
          // One camera (Tcd = identity, equal intrinsics and image size: fill_params checks
          // Tcw == Tdw and kc == kd bit for bit): Tcw * Xwp and its projection are exactly
          // what unit_issue computed for the depth image, so the colour pixel IS u_pix.
          want[i] = valid && fabsf(dist[k]) < 1.0f;
          color_index[i] = want[i] ? u_pix[k] : 0u;
          if (COLOR == COLOR_LIGHT) Xcp[i] = xform_point(P.Tdw, add3(u_off, voxel_offset));
        }
        else
        {
          float cu, cv;
          Xcp[i] = xform_point(P.Tcw, add3(u_off, voxel_offset));
          project(P.kc, Xcp[i], cu, cv);
          const bool color_valid = cu >= 0 && cu < P.cwidth && cv >= 0 && cv < P.cheight;
          want[i] = color_valid && fabsf(dist[k]) < 1.0f && (COLOR == COLOR_PLAIN || valid);
          color_index[i] = want[i] ? __umul24((uint32_t)f2i(cv), (uint32_t)P.cwidth) + (uint32_t)f2i(cu) : 0u;
        }
        depth_index[i] = SAME_CAM ? color_index[i] : ((COLOR == COLOR_LIGHT && want[i]) ? u_pix[k] : 0u);
      }

      // light_integrator.cu:215-225: the frame mask and the pixel's normal in the colour
      // camera's frame (Tcd * n). Neither depends on the voxel, so vk_light_prepare leaves
      // both in one 16-byte record per pixel: one gather here instead of four.
      f3 pixel_normal[G];   // already rotated by Tcd
      if (COLOR == COLOR_LIGHT)
      {
        if (RECORDS)
        {
          float4 rec[G];
#pragma unroll
          for (int i = 0; i < G; ++i) rec[i] = reinterpret_cast<const float4*>(P.records)[depth_index[i]];
#pragma unroll
          for (int i = 0; i < G; ++i)
          {
            want[i] = want[i] && rec[i].w > 0.5f;
            pixel_normal[i] = make3(rec[i].x, rec[i].y, rec[i].z);
          }
        }
        else
        {
          float m[G];
#pragma unroll
          for (int i = 0; i < G; ++i) m[i] = want[i] ? P.mask[depth_index[i]] : 0.0f;
#pragma unroll
          for (int i = 0; i < G; ++i) want[i] = want[i] && m[i] > 0.5f;
#pragma unroll
          for (int i = 0; i < G; ++i)
          {
            const vf3 n = *reinterpret_cast<const vf3*>(P.normals + 3 * (want[i] ? depth_index[i] : 0u));
            pixel_normal[i] = xform_dir(P.Tcd, make3(n.x, n.y, n.z));
          }
        }
      }

      // a 12-byte packed colour is one dwordx3 load (4-byte alignment suffices on gfx950)
      vf3 pixel_color[G];
#pragma unroll
      for (int i = 0; i < G; ++i)
        pixel_color[i] = *reinterpret_cast<const vf3*>(P.color + 3 * (want[i] ? color_index[i] : 0u));

#pragma unroll
      for (int i = 0; i < G; ++i)
      {
        const int k = g0 + i;
        if (!want[i]) continue;
        float* vox = tile + (k * 64 + lane) * 5;
        f3 curr_color = make3(pixel_color[i].x, pixel_color[i].y, pixel_color[i].z);

        if (COLOR == COLOR_LIGHT)
        {
          // light_integrator.cu:197-248
          const float shading = light_shading(P.light, Xcp[i], pixel_normal[i]);
          if (!(shading > 0.05f)) continue;
          curr_color = div3(curr_color, shading);
        }

        // color_integrator.cu:100-134 / light_integrator.cu:233-246
        const uint32_t weights = old_w[k];
        const int16_t cw = (int16_t)(weights >> 16);
        const float cwf = cw;
        const f3 stored = make3(vox[1], vox[2], vox[3]);
        const f3 prev_color = scale3(stored, cwf);
        const float color_weight = cw + 1;
        const int16_t new_cw = (int16_t)vmin(P.max_color_weight, color_weight);
        // matrix.h:279-295: operator/ multiplies by 1.0f / s
        const float inv_weight = small_weights ? (float)reciprocal[cw + 1] : 1.0f / color_weight;
        const f3 c = scale3(add3(prev_color, curr_color), inv_weight);
        const uint32_t new_weights = (weights & 0x0000ffffu) | ((uint32_t)(uint16_t)new_cw << 16);
        vox[1] = c.x;
        vox[2] = c.y;
        vox[3] = c.z;
        vox[4] = __uint_as_float(new_weights);
        const int dword = (k * 64 + lane) * 5;
        if ((__float_as_uint(c.x) ^ __float_as_uint(stored.x)) | (__float_as_uint(c.y) ^ __float_as_uint(stored.y)) |
            (__float_as_uint(c.z) ^ __float_as_uint(stored.z)))
        {
          changed[(dword + 1) >> 2] = 1;   // dwords 1..3 touch at most these two pieces
          changed[(dword + 3) >> 2] = 1;
          dirty = true;
        }
        if (new_weights != weights) { changed[(dword + 4) >> 2] = 1; dirty = true; }
      }
    }
  }

  if (__any(dirty))
  {
    // Write back only what differs from what was read: in steady state about a
    // third of a visible block is bit-for-bit unchanged (voxels behind the band are
    // never touched; voxels in front of it sit at distance 1 with a saturated
    // weight and are re-written with the same value), and a 64-byte line that no
    // lane stores to stays clean in L2 and is never written to HBM. Which pieces
    // changed comes from the `changed` bytes, not from a register copy of the tile as
    // read: those 20 VGPRs are what kept the RGB-D kernel at three waves per SIMD.
    wave_lds_fence();   // other lanes' voxels make up this lane's float4s
#pragma unroll
    for (int k = 0; k < 5; ++k)
    {
      if (changed[k * 64 + lane] == 0) continue;
      const float4 out = tile4[k * 64 + lane];
      if (g_nt_stores)
      {
        nf4 t; t.x = out.x; t.y = out.y; t.z = out.z; t.w = out.w;
        __builtin_nontemporal_store(t, reinterpret_cast<nf4*>(&u_base[k * 64 + lane]));
      }
      else u_base[k * 64 + lane] = out;
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

template <bool DEPTH, int COLOR, bool AHEAD, bool SAME_CAM, bool RECORDS>
__global__ __launch_bounds__(kPipeWavesPerGroup * 64) VK_INTEGRATE_WAVES void integrate_pipelined_kernel(IntegrateParams P, AheadParams A)
{
  // one LDS pool: four half-block tiles (20 KiB), or one bounds grid (37.5 KiB)
  constexpr int kTileInts = kPipeWavesPerGroup * kHalfF4 * 4;
  constexpr int kPoolInts = AHEAD ? (2 * kAheadMaxCells > kTileInts ? 2 * kAheadMaxCells : kTileInts) : kTileInts;
  __shared__ __attribute__((aligned(16))) int pool[kPoolInts];
  __shared__ __attribute__((aligned(8))) uint8_t changed_bytes[kPipeWavesPerGroup][kChangedBytes];
  __shared__ double reciprocal[kReciprocals];

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
  if (threadIdx.x < kReciprocals) reciprocal[threadIdx.x] = threadIdx.x ? 1.0 / (double)(int)threadIdx.x : 0.0;
  __syncthreads();

  const int lane = lane_id();
  const int wave_in_group = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int count = P.counters[VK_CTR_VISIBLE];
  float4* tile4 = reinterpret_cast<float4*>(pool) + wave_in_group * kHalfF4;
  uint8_t* changed = changed_bytes[wave_in_group];

  // Which blocks this wave takes. Plainly: item wave, wave + total_waves, ... of the visible list. With the banded
  // lists of the visibility pass (vk.h VK_BANDS; valid when VK_CTR_BANDED equals the count, every band within its
  // slots, the counts adding up): the lists are read as ONE list in band order and XCD x (workgroup g runs on XCD
  // g % 8) takes the x-th eighth of it — equal shares, each a contiguous range of image rows, so that the depth
  // pixels, records and colours an XCD's blocks gather from are one eighth of the images (plus the rows of the bands'
  // overlap) and stay in that XCD's 4 MB L2. Dealt round robin, every L2 streamed all 9.8 MB of images for itself:
  // 3.6 fetches of every image byte per launch (profiles/r03_integrate_rgbd_traffic.json).
  int lo = 0, hi = count, stride = groups * kPipeWavesPerGroup, first_item = group * kPipeWavesPerGroup + wave_in_group;
  int band_end[VK_BANDS];
  bool banded = !VK_INTEGRATE_PLAIN_LIST && groups >= kXCDs && count > 0 && P.counters[VK_CTR_BANDED] == count;
  if (banded)
  {
    const int32_t* sizes = band_counts(P.counters);
    int total = 0;
#pragma unroll
    for (int b = 0; b < VK_BANDS; ++b)
    {
      const int n = sizes[b];
      banded = banded && n >= 0 && n <= VK_BAND_SLOTS;
      total += n;
      band_end[b] = total;
    }
    banded = banded && total == count;
  }
  if (banded)
  {
    const int xcd = group % kXCDs, local_group = group / kXCDs;
    const int local_groups = (groups - xcd + kXCDs - 1) / kXCDs;
    lo = (int)(((long long)count * xcd) / kXCDs);
    hi = (int)(((long long)count * (xcd + 1)) / kXCDs);
    stride = local_groups * kPipeWavesPerGroup;
    first_item = lo + local_group * kPipeWavesPerGroup + wave_in_group;
  }
  const int32_t* lists = band_lists(P.counters);

  for (int first = first_item; first < hi; first += 64 * stride)
  {
    // lane j holds the hash entry of this wave's j-th block of the group
    const int mine = first + lane * stride;
    int4 my_entry = make_int4(0, 0, -1, -1);
    if (mine < hi)
    {
      int entry_index;
      if (banded)
      {
        int b = 0;
#pragma unroll
        for (int i = 0; i < VK_BANDS - 1; ++i) b += mine >= band_end[i] ? 1 : 0;
        int before = 0;
#pragma unroll
        for (int i = 0; i < VK_BANDS - 1; ++i) before = (b == i + 1) ? band_end[i] : before;
        entry_index = lists[b * VK_BAND_SLOTS + (mine - before)];
      }
      else entry_index = P.visible[mine];
      my_entry = reinterpret_cast<const int4*>(P.entries)[entry_index];
    }
    int blocks = (hi - first + stride - 1) / stride;
    if (blocks > 64) blocks = 64;
    const int units = 2 * blocks;   // unit s = (block s >> 1, half s & 1)

    UNIT_DECL(A);
    UNIT_DECL(B);
    unit_issue<DEPTH, COLOR, SAME_CAM>(P, my_entry, 0, lane, UNIT_ARGS(A));
    int s = 0;
    for (; s + 2 < units; s += 2)   // steady state: two units per trip, next one always in flight
    {
      unit_issue<DEPTH, COLOR, SAME_CAM>(P, my_entry, s + 1, lane, UNIT_ARGS(B));
      unit_update<DEPTH, COLOR, SAME_CAM, RECORDS>(P, lane, tile4, changed, reciprocal, UNIT_ARGS(A));
      unit_issue<DEPTH, COLOR, SAME_CAM>(P, my_entry, s + 2, lane, UNIT_ARGS(A));
      unit_update<DEPTH, COLOR, SAME_CAM, RECORDS>(P, lane, tile4, changed, reciprocal, UNIT_ARGS(B));
    }
    // units is even and >= 2: exactly two remain (s, s + 1)
    unit_issue<DEPTH, COLOR, SAME_CAM>(P, my_entry, s + 1, lane, UNIT_ARGS(B));
    unit_update<DEPTH, COLOR, SAME_CAM, RECORDS>(P, lane, tile4, changed, reciprocal, UNIT_ARGS(A));
    unit_update<DEPTH, COLOR, SAME_CAM, RECORDS>(P, lane, tile4, changed, reciprocal, UNIT_ARGS(B));
  }
}

int fill_params(IntegrateParams& P, const vk_volume* v, const vk_integrator* p, const vk_frame* f,
    const vk_light* light, const float* mask, bool need_depth, bool need_color, bool need_light,
    const float* records = nullptr)
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
  P.records = records;
  if (records && (reinterpret_cast<uintptr_t>(records) & 15)) return VK_ERR_ARGUMENT;
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
  // one camera for depth and colour: the colour projection of a voxel IS its depth projection
  P.same_camera = cwidth == f->width && cheight == f->height &&
      std::memcmp(Tcw, f->depth_to_world.inv, sizeof(Tcw)) == 0 &&
      std::memcmp(&f->color_projection, &f->depth_projection, sizeof(vk_projection)) == 0;
  P.Tcd = make_rt(f->depth_to_color.m);
  if (light) P.light = *light; else { P.light.intensity = 1.0f; P.light.position[0] = P.light.position[1] = P.light.position[2] = 0.0f; }
  P.voxel_length = v->voxel_length;
  P.block_length = VK_BLOCK_RESOLUTION * v->voxel_length;
  P.truncation_length = v->truncation_length;
  P.inv_truncation_length = 1.0 / (double)v->truncation_length;
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
template <bool DEPTH, int COLOR, bool SAME_CAM, bool RECORDS>
int launch_as(const IntegrateParams& P, const vk_volume* v, const vk_frame* frame, vk_view_bounds* ahead, hipStream_t s)
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
    hipLaunchKernelGGL((integrate_pipelined_kernel<DEPTH, COLOR, true, SAME_CAM, RECORDS>), dim3(grid),
        dim3(kPipeWavesPerGroup * 64), 0, s, P, A);
  }
  else
    hipLaunchKernelGGL((integrate_pipelined_kernel<DEPTH, COLOR, false, SAME_CAM, RECORDS>), dim3(pipe_grid_for(v, 5)),
        dim3(kPipeWavesPerGroup * 64), 0, s, P, A);
  VK_LAUNCH_CHECK();
  if (with_bounds) view_record(ahead, v, frame);
  return VK_OK;
}

template <bool DEPTH, int COLOR>
int launch(const IntegrateParams& P, const vk_volume* v, const vk_frame* frame, vk_view_bounds* ahead, hipStream_t s)
{
  if (COLOR == COLOR_NONE) return launch_as<DEPTH, COLOR, false, false>(P, v, frame, ahead, s);
  const bool records = COLOR == COLOR_LIGHT && P.records;
  if (P.same_camera)
    return records ? launch_as<DEPTH, COLOR, true, COLOR == COLOR_LIGHT>(P, v, frame, ahead, s)
                   : launch_as<DEPTH, COLOR, true, false>(P, v, frame, ahead, s);
  return records ? launch_as<DEPTH, COLOR, false, COLOR == COLOR_LIGHT>(P, v, frame, ahead, s)
                 : launch_as<DEPTH, COLOR, false, false>(P, v, frame, ahead, s);
}

// ---------------------------------------------------------------- frame mask ----

// ref: light_integrator.cu:17-103 ComputeFrameMaskKernel<16,3>. The reference
// stages a 22x22 tile with a -1 halo offset and reads it with a +3 centre, so
// pixel (x,y) looks at [x-1,x+5] x [y-1,y+5]; kept as is.
// With `records` it also leaves, per pixel, what LightIntegrator's colour kernel would
// otherwise gather and compute once per VOXEL that projects to the pixel
// (light_integrator.cu:215-225): {Tcd * normal, mask} as one 16-byte record.
__global__ __launch_bounds__(256) void frame_mask_kernel(int width, int height,
    const float* __restrict__ depths, const float* __restrict__ colors, float depth_threshold,
    float* __restrict__ mask, const float* __restrict__ normals, Rt Tcd, float4* __restrict__ records)
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

    f3 Xcn = make3(0, 0, 0);
    if (records)
    {
      const vf3 n = *reinterpret_cast<const vf3*>(normals + 3 * index);
      Xcn = xform_dir(Tcd, make3(n.x, n.y, n.z));                     // light_integrator.cu:223
    }

    if (!light_color_usable(c0, c1, c2))
    {
      mask[index] = 0.0f;
      if (records) records[index] = make_float4(Xcn.x, Xcn.y, Xcn.z, 0.0f);
      return;
    }

    const float m = light_window_mask(buffer, DIM, tx + KS, ty + KS, depth_threshold);
    mask[index] = m;
    if (records) records[index] = make_float4(Xcn.x, Xcn.y, Xcn.z, m);
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

static int light_prepare(const vk_frame* frame, float depth_threshold, float* mask, float* records, void* stream)
{
  VK_REQUIRE(frame && frame->depth && frame->color && mask && frame->width > 0 && frame->height > 0);
  VK_REQUIRE(!records || (frame->normals && (reinterpret_cast<uintptr_t>(records) & 15) == 0));
  // light_integrator.cu:277-293 walks the colour image with the depth image's size
  VK_REQUIRE((frame->color_width <= 0 || frame->color_width == frame->width) &&
             (frame->color_height <= 0 || frame->color_height == frame->height));
  const dim3 grid((frame->width + 15) / 16, (frame->height + 15) / 16);
  hipLaunchKernelGGL(frame_mask_kernel, grid, dim3(256), 0, vk_s(stream), frame->width, frame->height,
      frame->depth, frame->color, depth_threshold, mask, frame->normals, make_rt(frame->depth_to_color.m),
      reinterpret_cast<float4*>(records));
  VK_LAUNCH_CHECK();
  return VK_OK;
}

int vk_light_compute_frame_mask(const vk_frame* frame, float depth_threshold, float* mask, void* stream)
{
  return light_prepare(frame, depth_threshold, mask, nullptr, stream);
}

int vk_light_prepare(const vk_frame* frame, float depth_threshold, float* mask, float* records, void* stream)
{
  VK_REQUIRE(records);
  return light_prepare(frame, depth_threshold, mask, records, stream);
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
    const vk_light* light, const float* mask, const float* light_records, vk_view_bounds* ahead, void* stream)
{
  IntegrateParams P;
  VK_REQUIRE(color_mode >= 0 && color_mode <= 2);
  const int rc = fill_params(P, v, p, frame, color_mode == 2 ? light : nullptr, color_mode == 2 ? mask : nullptr, true,
      color_mode != 0, color_mode == 2, color_mode == 2 ? light_records : nullptr);
  if (rc != VK_OK) return rc;
  if (color_mode == 0) return launch<true, COLOR_NONE>(P, v, frame, ahead, vk_s(stream));
  if (color_mode == 1) return launch<true, COLOR_PLAIN>(P, v, frame, ahead, vk_s(stream));
  return launch<true, COLOR_LIGHT>(P, v, frame, ahead, vk_s(stream));
}

}  // extern "C"
