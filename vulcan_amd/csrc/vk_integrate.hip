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
#include <hip/hip_ext.h>

#ifndef VK_INTEGRATE_NT_STORES
#define VK_INTEGRATE_NT_STORES 0
#endif
#ifndef VK_INTEGRATE_PLAIN_LIST
#define VK_INTEGRATE_PLAIN_LIST 0   // 1: never use the banded visible lists (A/B measurements, tools/build_variant.sh)
#endif
#ifndef VK_LIGHT_PACKED
// 1: the light model of two voxels in hand-packed f32 (divide2 below; VERDICT r4 #4's experiment). Bit-exact (the whole GPU
// suite passes with it) and SLOWER: profiles/r05_light_packed_f32.txt — 36.7 us against 34.8 (35.7 at the same number of
// gathers ahead). The product is the scalar form.
#define VK_LIGHT_PACKED 0
#endif
// Wave priority (s_setprio, round 5). The kernel is bound by instruction issue; which of a SIMD's four waves issues next is
// the hardware's choice (oldest first) unless the waves say otherwise. A unit's update has one long stretch of pure
// arithmetic — step 4, the light model — between two stretches that END IN MEMORY REQUESTS (steps 1-3: tile to LDS, depth
// pass, the gathers and the next unit's tile; step 5: the write-back). With the arithmetic at priority 0 and the rest at 1
// a wave whose data has just landed gets its next requests out before a neighbour's light model takes the issue slots:
// 34.9 -> 34.0 us (two alternating pairs of runs; steps 1-3 alone 34.2, steps 2-3 alone 34.8, the light model ALONE raised
// 34.6; levels 1, 2, 3 the same). Same instructions, same bits. -DVK_INTEGRATE_PRIO=0: without.
#ifndef VK_INTEGRATE_PRIO
#define VK_INTEGRATE_PRIO 1
#endif
#ifndef VK_INTEGRATE_GATHER_AHEAD
#define VK_INTEGRATE_GATHER_AHEAD (VK_LIGHT_PACKED ? 2 : 4)   // packed: three / four voxels' gathers ahead cost 12 / 28 bytes of scratch
#endif
#ifndef VK_INTEGRATE_WAVES_PER_EU
#define VK_INTEGRATE_WAVES_PER_EU 4
#endif

// Four waves per SIMD (<= 128 VGPRs) for every variant: the RGB-D kernel is bound by the
// latency of its image gathers and division chains, and ran 42 us at three waves
// (155 VGPRs), 39 us at four (r02 variants D/E/F, profiles/r02_b_light_variants.txt).
// The register allocator reaches 128 by spilling two dwords outside the voxel loop.
#define VK_INTEGRATE_WAVES __attribute__((amdgpu_waves_per_eu(VK_INTEGRATE_WAVES_PER_EU)))
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

// ---- the light model for TWO voxels at a time, in packed f32 (round 5; an experiment that lost: VK_LIGHT_PACKED) -----
// The RGB-D kernel issues 817 vector instructions per half block and wave, and most of them are the light model's: per
// voxel four correctly rounded divisions, a square root, and some sixty multiplies and adds. gfx950 issues v_pk_mul_f32 /
// v_pk_add_f32 / v_pk_fma_f32 — two binary32 operations per lane — in the slot of one: the two voxels a lane updates side
// by side (kLightGroup) become the two halves of 64-bit register pairs. It removes 8 % of the loop's vector instructions
// (1881 -> 1733 static, 260 of them packed) and adds 60 v_mov (pairs put together from separately loaded registers), 100
// scalar instructions and 88 s_nop (more VALU-writes-SGPR hazards), and the aligned pairs cost registers: with all four
// voxels' gathers ahead 28 bytes of scratch, with three 12, none with two. Measured: slower in every form. Every operation
// is the same IEEE operation on the same operands in the same order as in the scalar form (contraction stays off: a packed
// multiply followed by a packed add is two roundings), so the bits are the scalar form's — tests/test_gpu_*.py compare
// them with the oracle as before. The division is written out: it is hipcc's own expansion of `a / b` (LowerFDIV32:
// v_div_scale x2, v_rcp, FMA x3 + MUL + FMA x2, v_div_fmas, v_div_fixup — correctly rounded, so ANY correct sequence gives
// these bits) with its six multiply-adds done for both quotients at once.
typedef float f2v __attribute__((ext_vector_type(2)));
struct p3 { f2v x, y, z; };          // three coordinates of a PAIR of points / vectors

__device__ __forceinline__ f2v splat2(float a) { return f2v{a, a}; }
__device__ __forceinline__ f2v fma2(f2v a, f2v b, f2v c) { return __builtin_elementwise_fma(a, b, c); }

__device__ __forceinline__ f2v divide2(f2v n, f2v d)
{
  bool unused, flag0, flag1;
  const f2v ds = {__builtin_amdgcn_div_scalef(n.x, d.x, false, &unused), __builtin_amdgcn_div_scalef(n.y, d.y, false, &unused)};
  const f2v ns = {__builtin_amdgcn_div_scalef(n.x, d.x, true, &flag0), __builtin_amdgcn_div_scalef(n.y, d.y, true, &flag1)};
  const f2v r = {__builtin_amdgcn_rcpf(ds.x), __builtin_amdgcn_rcpf(ds.y)};
  const f2v nd = -ds;
  const f2v e0 = fma2(nd, r, splat2(1.0f));
  const f2v r1 = fma2(e0, r, r);
  const f2v q = ns * r1;
  const f2v e1 = fma2(nd, q, ns);
  const f2v q1 = fma2(e1, r1, q);
  const f2v e2 = fma2(nd, q1, ns);
  return f2v{__builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(e2.x, r1.x, q1.x, flag0), d.x, n.x),
             __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(e2.y, r1.y, q1.y, flag1), d.y, n.y)};
}

// matrix.h:157-169 Dot for a pair: starts from 0, adds in index order
__device__ __forceinline__ f2v dot3(const p3& a, const p3& b)
{
  f2v r = splat2(0.0f);
  r += a.x * b.x;
  r += a.y * b.y;
  r += a.z * b.z;
  return r;
}

// light.h:53-60 for two points and their normals
__device__ __forceinline__ f2v light_shading(const vk_light& l, const p3& point, const p3& normal)
{
  const p3 delta = {splat2(l.position[0]) - point.x, splat2(l.position[1]) - point.y, splat2(l.position[2]) - point.z};
  const f2v squared = dot3(delta, delta);                                     // sqnorm3, also distance_squared below
  const f2v inv = divide2(splat2(1.0f), f2v{sqrtf(squared.x), sqrtf(squared.y)});   // normalized3: 1 / sqrt(dot)
  const p3 direction = {delta.x * inv, delta.y * inv, delta.z * inv};
  const f2v cos_theta = dot3(normal, direction);
  return divide2(splat2(l.intensity) * cos_theta, squared);
}

// ---------------------------------------------------------------------------
// Software pipeline. Moving a whole block per step makes every wave of a CU load,
// then compute, then store at about the same time, so the ~10 us of per-voxel
// arithmetic does not overlap with the ~15 us of data movement (r01 ablations,
// DESIGN.md section 4). The unit of work is therefore HALF a block (four z
// slices = 5 KiB = 320 float4, five float4 per lane), and the tile loads and depth
// gathers of unit s+1 leave in the middle of unit s's update — after its own
// gathers, before its light model — into the registers unit s's tile has just
// left for LDS (unit_step below has the reasons).
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

// One unit's update, and — in the middle of it — the issue of the unit that takes over its registers.
//
// Vector memory operations retire in order as far as s_waitcnt can tell (vmcnt counts the operations issued
// AFTER the one waited for), so the order of issue decides what a wait drains. Until round 3 a wave held two
// register sets and issued unit s+1's tile loads BEFORE unit s's update; the record and colour gathers of that
// update came after them in program order, and the wait for those gathers (`s_waitcnt vmcnt(0)` in the ISA)
// drained the tile loads too: they were on their way during the depth pass only, and the light model — most of
// a unit's arithmetic — overlapped with no load of its own wave. Now (round 4):
//   1. the half tile moves from its registers into LDS; the depth pass runs; every voxel's gather addresses
//      are known (they need the distances the depth pass has just made);
//   2. the record / colour gathers of all four voxels are issued (kGatherAhead);
//   3. the tile registers are free: the tile loads and depth gathers of unit s+1 are issued INTO THEM (NEXT);
//   4. the wait for the gathers leaves those nine younger operations in flight (vmcnt(9) ... vmcnt(16) in the
//      ISA); the light model and the write-back run while unit s+1 is on its way.
// One register set instead of two: 127 VGPRs without a spill for the RGB-D kernel (128 + 8 bytes of scratch
// before), 66 for the depth kernel. Measured on the banded lists (profiles/r04_a_integrate_schedule_variants.txt):
// 35.9 -> 34.6 us (RGB-D), 27.7 -> 27.1 us (depth). Two sets with unit s+2 leaving in the middle of unit s — two
// units in flight per wave — need 167 VGPRs = three waves per SIMD: 40.4 us; with only two voxels' gathers ahead
// the second pair's wait drains the next tile again (35.5 us); the tile loads ahead of the next unit's projections
// 35.2-35.6 us; dealing the two-block waves evenly over the CUs of an XCD 35.6 us (tools/patches/ keeps the first).
constexpr int kGatherAhead = VK_INTEGRATE_GATHER_AHEAD;   // voxels whose light gathers leave before step 3

template <bool DEPTH, int COLOR, bool SAME_CAM, bool RECORDS, bool NEXT>
__device__ __forceinline__ void unit_step(const IntegrateParams& P, int4 my_entry, int next_s, int lane, float4* tile4,
    uint8_t* changed, const double* reciprocal, UNIT_PARAMS)
{
  float* tile = reinterpret_cast<float*>(tile4);
  const int vx = lane & 7, vy = lane >> 3;
#if VK_INTEGRATE_PRIO
  __builtin_amdgcn_s_setprio(VK_INTEGRATE_PRIO);   // steps 1-3 end in memory requests: ahead of the other waves' arithmetic
#endif
  // The never-allocated origin block (data == -1) has read slot 0 and goes through the update like
  // any other unit — one straight path keeps the wait counts of the steady state exact — but writes nothing back.
  const bool skip = u_skip;
  // what the rest of this update needs after the register set has gone to the next unit
  float4* const base = u_base;
  const f3 off = u_off;
  const int half = u_half;
  const uint32_t valid_bits = u_valid;

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
    const bool valid = (valid_bits >> k) & 1u;

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

  // ---- which voxels take colour and from which pixels (pure conditions, so testing |dist| < 1 before the mask —
  // the reference tests the mask first — selects the same voxels) -------------------------------------------------
  bool want[4] = {false, false, false, false};
  uint32_t color_index[4] = {0u, 0u, 0u, 0u}, depth_index[4] = {0u, 0u, 0u, 0u};
  if (COLOR != COLOR_NONE)
  {
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
      const bool valid = (valid_bits >> k) & 1u;
      if (SAME_CAM)
      {
        // One camera (Tcd = identity, equal intrinsics and image size: fill_params checks
        // Tcw == Tdw and kc == kd bit for bit): Tcw * Xwp and its projection are exactly
        // what unit_issue computed for the depth image, so the colour pixel IS u_pix.
        want[k] = valid && fabsf(dist[k]) < 1.0f;
        color_index[k] = want[k] ? u_pix[k] : 0u;
        depth_index[k] = color_index[k];
      }
      else
      {
        const int vz = half * 4 + k;
        const f3 voxel_offset = scale3(make3(vx + 0.5f, vy + 0.5f, vz + 0.5f), P.voxel_length);
        float cu, cv;
        const f3 Xcp = xform_point(P.Tcw, add3(off, voxel_offset));
        project(P.kc, Xcp, cu, cv);
        const bool color_valid = cu >= 0 && cu < P.cwidth && cv >= 0 && cv < P.cheight;
        want[k] = color_valid && fabsf(dist[k]) < 1.0f && (COLOR == COLOR_PLAIN || valid);
        color_index[k] = want[k] ? __umul24((uint32_t)f2i(cv), (uint32_t)P.cwidth) + (uint32_t)f2i(cu) : 0u;
        depth_index[k] = (COLOR == COLOR_LIGHT && want[k]) ? u_pix[k] : 0u;
      }
    }
  }

  // ---- step 2: the gathers that can leave now ---------------------------------------------------------------------
  // light_integrator.cu:215-225: the frame mask and the pixel's normal in the colour camera's frame (Tcd * n). Neither
  // depends on the voxel, so vk_light_prepare leaves both in one 16-byte record per pixel: one gather instead of four.
  // A 12-byte packed colour is one dwordx3 load (4-byte alignment suffices on gfx950).
  constexpr int AHEAD = (COLOR == COLOR_LIGHT) ? (RECORDS ? kGatherAhead : 0) : 4;
  float4 rec[4];
  vf3 pixel_color[4];
  float mask_value[4];
  if (COLOR == COLOR_LIGHT && RECORDS)
  {
#pragma unroll
    for (int k = 0; k < AHEAD; ++k) rec[k] = reinterpret_cast<const float4*>(P.records)[depth_index[k]];
  }
  if (COLOR == COLOR_LIGHT && !RECORDS)
  {
#pragma unroll
    for (int k = 0; k < 4; ++k) mask_value[k] = want[k] ? P.mask[depth_index[k]] : 0.0f;
  }
  if (COLOR != COLOR_NONE)
  {
#pragma unroll
    for (int k = 0; k < AHEAD; ++k) pixel_color[k] = *reinterpret_cast<const vf3*>(P.color + 3 * color_index[k]);
  }

  // ---- step 3: the next unit takes the register set ---------------------------------------------------------------
  __builtin_amdgcn_sched_barrier(0);
  if (NEXT) unit_issue<DEPTH, COLOR, SAME_CAM>(P, my_entry, next_s, lane, r0, r1, r2, r3, r4, u_depth, u_z, u_pix, u_valid,
      u_base, u_skip, u_off, u_half);
  __builtin_amdgcn_sched_barrier(0);
#if VK_INTEGRATE_PRIO
  __builtin_amdgcn_s_setprio(0);                   // the light model: pure arithmetic, behind the others' requests
#endif

  // ---- step 4: the colour running averages -------------------------------------------------------------------------
  if (COLOR != COLOR_NONE)
  {
    // G voxels at a time: per voxel the light model needs a normalisation and three divisions, ~35 registers; two
    // at a time keep the kernel at 128 VGPRs = four waves per SIMD. The plain colour pass takes all four together.
    constexpr int G = (COLOR == COLOR_LIGHT) ? kLightGroup : 4;
#pragma unroll
    for (int g0 = 0; g0 < 4; g0 += G)
    {
      if (g0 > 0) __builtin_amdgcn_sched_barrier(0);   // keep the next group's work below this group's

      f3 pixel_normal[G];   // already rotated by Tcd
      if (COLOR == COLOR_LIGHT)
      {
        if (RECORDS)
        {
#pragma unroll
          for (int i = 0; i < G; ++i)
            if (g0 + i >= AHEAD) rec[g0 + i] = reinterpret_cast<const float4*>(P.records)[depth_index[g0 + i]];
#pragma unroll
          for (int i = 0; i < G; ++i)
          {
            want[g0 + i] = want[g0 + i] && rec[g0 + i].w > 0.5f;
            pixel_normal[i] = make3(rec[g0 + i].x, rec[g0 + i].y, rec[g0 + i].z);
          }
        }
        else
        {
#pragma unroll
          for (int i = 0; i < G; ++i) want[g0 + i] = want[g0 + i] && mask_value[g0 + i] > 0.5f;
#pragma unroll
          for (int i = 0; i < G; ++i)
          {
            const vf3 n = *reinterpret_cast<const vf3*>(P.normals + 3 * (want[g0 + i] ? depth_index[g0 + i] : 0u));
            pixel_normal[i] = xform_dir(P.Tcd, make3(n.x, n.y, n.z));
          }
        }
      }
#pragma unroll
      for (int i = 0; i < G; ++i)
        if (g0 + i >= AHEAD)
          pixel_color[g0 + i] = *reinterpret_cast<const vf3*>(P.color + 3 * (want[g0 + i] ? color_index[g0 + i] : 0u));

      if (COLOR == COLOR_LIGHT && VK_LIGHT_PACKED && G == 2)
      {
        // ---- both voxels of the group in packed f32 (see divide2): light_integrator.cu:197-248, twice at once -----------
        const int k0 = g0, k1 = g0 + 1;
        float* vox0 = tile + (k0 * 64 + lane) * 5;
        float* vox1 = tile + (k1 * 64 + lane) * 5;
        // the voxels' positions in the colour camera: x and y of the two voxel offsets are the same numbers, so the part
        // of xform_point's sum that they make up — (r0 * px + r1 * py), the first two terms in its order — is the same
        // for both and computed once
        const Rt& T = SAME_CAM ? P.Tdw : P.Tcw;
        const f3 offset_xy = scale3(make3(vx + 0.5f, vy + 0.5f, 0.0f), P.voxel_length);
        const float px = off.x + offset_xy.x, py = off.y + offset_xy.y;
        const f2v pz = splat2(off.z) + f2v{(half * 4 + k0) + 0.5f, (half * 4 + k1) + 0.5f} * splat2(P.voxel_length);
        p3 Xcp;
        Xcp.x = splat2(T.r[0] * px + T.r[1] * py) + splat2(T.r[2]) * pz + splat2(T.r[3] * 1.0f);
        Xcp.y = splat2(T.r[4] * px + T.r[5] * py) + splat2(T.r[6]) * pz + splat2(T.r[7] * 1.0f);
        Xcp.z = splat2(T.r[8] * px + T.r[9] * py) + splat2(T.r[10]) * pz + splat2(T.r[11] * 1.0f);
        const p3 normal = {f2v{pixel_normal[0].x, pixel_normal[1].x}, f2v{pixel_normal[0].y, pixel_normal[1].y},
                           f2v{pixel_normal[0].z, pixel_normal[1].z}};
        const f2v shading = light_shading(P.light, Xcp, normal);
        const bool ok0 = want[k0] && shading.x > 0.05f, ok1 = want[k1] && shading.y > 0.05f;
        if (ok0 || ok1)
        {
          const f2v inv_shading = divide2(splat2(1.0f), shading);                   // div3: multiply by 1.0f / s
          const p3 curr = {f2v{pixel_color[k0].x, pixel_color[k1].x} * inv_shading, f2v{pixel_color[k0].y, pixel_color[k1].y} * inv_shading,
                           f2v{pixel_color[k0].z, pixel_color[k1].z} * inv_shading};
          // color_integrator.cu:100-134 / light_integrator.cu:233-246
          // (the weights as the depth pass left them, from the tile: four registers less across the wait for the gathers)
          const uint32_t w0 = __float_as_uint(vox0[4]), w1 = __float_as_uint(vox1[4]);
          const int16_t cw0 = (int16_t)(w0 >> 16), cw1 = (int16_t)(w1 >> 16);
          const f2v cwf = {(float)cw0, (float)cw1};
          const p3 stored = {f2v{vox0[1], vox1[1]}, f2v{vox0[2], vox1[2]}, f2v{vox0[3], vox1[3]}};
          const f2v color_weight = {(float)(cw0 + 1), (float)(cw1 + 1)};
          const int16_t new_cw0 = (int16_t)vmin(P.max_color_weight, color_weight.x), new_cw1 = (int16_t)vmin(P.max_color_weight, color_weight.y);
          // matrix.h:279-295: operator/ multiplies by 1.0f / s
          const f2v inv_weight = small_weights ? f2v{(float)reciprocal[cw0 + 1], (float)reciprocal[cw1 + 1]}
                                               : divide2(splat2(1.0f), color_weight);
          const p3 c = {(stored.x * cwf + curr.x) * inv_weight, (stored.y * cwf + curr.y) * inv_weight, (stored.z * cwf + curr.z) * inv_weight};
          const uint32_t nw0 = (w0 & 0x0000ffffu) | ((uint32_t)(uint16_t)new_cw0 << 16);
          const uint32_t nw1 = (w1 & 0x0000ffffu) | ((uint32_t)(uint16_t)new_cw1 << 16);
          if (ok0)
          {
            vox0[1] = c.x.x; vox0[2] = c.y.x; vox0[3] = c.z.x; vox0[4] = __uint_as_float(nw0);
            const int dword = (k0 * 64 + lane) * 5;
            if ((__float_as_uint(c.x.x) ^ __float_as_uint(stored.x.x)) | (__float_as_uint(c.y.x) ^ __float_as_uint(stored.y.x)) |
                (__float_as_uint(c.z.x) ^ __float_as_uint(stored.z.x)))
            {
              changed[(dword + 1) >> 2] = 1;   // dwords 1..3 touch at most these two pieces
              changed[(dword + 3) >> 2] = 1;
              dirty = true;
            }
            if (nw0 != w0) { changed[(dword + 4) >> 2] = 1; dirty = true; }
          }
          if (ok1)
          {
            vox1[1] = c.x.y; vox1[2] = c.y.y; vox1[3] = c.z.y; vox1[4] = __uint_as_float(nw1);
            const int dword = (k1 * 64 + lane) * 5;
            if ((__float_as_uint(c.x.y) ^ __float_as_uint(stored.x.y)) | (__float_as_uint(c.y.y) ^ __float_as_uint(stored.y.y)) |
                (__float_as_uint(c.z.y) ^ __float_as_uint(stored.z.y)))
            {
              changed[(dword + 1) >> 2] = 1;
              changed[(dword + 3) >> 2] = 1;
              dirty = true;
            }
            if (nw1 != w1) { changed[(dword + 4) >> 2] = 1; dirty = true; }
          }
        }
      }
      else
#pragma unroll
      for (int i = 0; i < G; ++i)
      {
        const int k = g0 + i;
        if (!want[k]) continue;
        float* vox = tile + (k * 64 + lane) * 5;
        f3 curr_color = make3(pixel_color[k].x, pixel_color[k].y, pixel_color[k].z);

        if (COLOR == COLOR_LIGHT)
        {
          // light_integrator.cu:197-248
          const int vz = half * 4 + k;
          const f3 voxel_offset = scale3(make3(vx + 0.5f, vy + 0.5f, vz + 0.5f), P.voxel_length);
          const f3 Xcp = xform_point(SAME_CAM ? P.Tdw : P.Tcw, add3(off, voxel_offset));
          const float shading = light_shading(P.light, Xcp, pixel_normal[i]);
          if (!(shading > 0.05f)) continue;
          curr_color = div3(curr_color, shading);
        }

        // color_integrator.cu:100-134 / light_integrator.cu:233-246
        const uint32_t weights = (COLOR == COLOR_LIGHT) ? __float_as_uint(vox[4]) : old_w[k];   // (light: from the tile, see the packed form)
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

#if VK_INTEGRATE_PRIO
  __builtin_amdgcn_s_setprio(VK_INTEGRATE_PRIO);   // the write-back
#endif
  if (__any(dirty) && !skip)
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
        __builtin_nontemporal_store(t, reinterpret_cast<nf4*>(&base[k * 64 + lane]));
      }
      else base[k * 64 + lane] = out;
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
    // one register set: unit s + 1 leaves in the middle of unit s's update, into the registers unit s has just left
    unit_issue<DEPTH, COLOR, SAME_CAM>(P, my_entry, 0, lane, UNIT_ARGS(A));
    for (int s = 0; s + 1 < units; ++s)
      unit_step<DEPTH, COLOR, SAME_CAM, RECORDS, true>(P, my_entry, s + 1, lane, tile4, changed, reciprocal, UNIT_ARGS(A));
    unit_step<DEPTH, COLOR, SAME_CAM, RECORDS, false>(P, my_entry, 0, lane, tile4, changed, reciprocal, UNIT_ARGS(A));
  }
}

#ifndef VK_INTEGRATE_RING
#define VK_INTEGRATE_RING 0     // 1: quarter-block units through a three-deep LDS-DMA ring (vk_integrate_ring.inc; an experiment)
#endif
#include "vk_integrate_ring.inc"

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

// vk_integrate_time_next: the events the next pipelined launch of this host thread records as its own begin and end
thread_local hipEvent_t g_time_start = nullptr, g_time_stop = nullptr;

// the pair is used up by the next vk_integrate_* CALL, whether or not that call gets as far as its launch (an argument error,
// a failed fill_params): armed events must never be recorded by a later, unrelated launch
struct TimedLaunchScope
{
  ~TimedLaunchScope() { g_time_start = g_time_stop = nullptr; }
};

template <typename Kernel>
void launch_pipelined(Kernel kernel, int grid, hipStream_t s, const IntegrateParams& P, const AheadParams& A)
{
  if (g_time_start && g_time_stop)
  {
    hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(kPipeWavesPerGroup * 64), 0, s, g_time_start, g_time_stop, 0, P, A);
    g_time_start = g_time_stop = nullptr;
  }
  else hipLaunchKernelGGL(kernel, dim3(grid), dim3(kPipeWavesPerGroup * 64), 0, s, P, A);
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

#if VK_INTEGRATE_RING
  // the experiment covers the two instantiations the bench times: depth only, and depth + light colour from one camera with records
  if (DEPTH && (COLOR == COLOR_NONE || (COLOR == COLOR_LIGHT && SAME_CAM && RECORDS)))
  {
    constexpr int RING_COLOR = COLOR == COLOR_NONE ? COLOR_NONE : COLOR_LIGHT;
    if (with_bounds)
      launch_pipelined(ring::integrate_ring_kernel<RING_COLOR, true>, pipe_grid_for(v, 4) + kBoundsGroups, s, P, A);
    else   // (without the bounds groups' 37.5 KiB the ring's 32.4 KiB fit five times into a CU: VK_INTEGRATE_WAVES_PER_EU=5)
      launch_pipelined(ring::integrate_ring_kernel<RING_COLOR, false>, pipe_grid_for(v, VK_INTEGRATE_WAVES_PER_EU), s, P, A);
    VK_LAUNCH_CHECK();
    if (with_bounds) view_record(ahead, v, frame);
    return VK_OK;
  }
#endif
  if (with_bounds)
  {
    // 37.5 KiB of LDS per workgroup: four per CU
    launch_pipelined(integrate_pipelined_kernel<DEPTH, COLOR, true, SAME_CAM, RECORDS>, pipe_grid_for(v, 4) + kBoundsGroups, s, P, A);
  }
  else
    launch_pipelined(integrate_pipelined_kernel<DEPTH, COLOR, false, SAME_CAM, RECORDS>, pipe_grid_for(v, 5), s, P, A);
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
  TimedLaunchScope used_up;
  IntegrateParams P;
  const int rc = fill_params(P, v, p, frame, nullptr, nullptr, true, false, false);
  if (rc != VK_OK) return rc;
  return launch<true, COLOR_NONE>(P, v, frame, nullptr, vk_s(stream));
}

int vk_integrate_color(const vk_volume* v, const vk_integrator* p, const vk_frame* frame, void* stream)
{
  TimedLaunchScope used_up;
  IntegrateParams P;
  const int rc = fill_params(P, v, p, frame, nullptr, nullptr, false, true, false);
  if (rc != VK_OK) return rc;
  return launch<false, COLOR_PLAIN>(P, v, frame, nullptr, vk_s(stream));
}

int vk_integrate_depth_color(const vk_volume* v, const vk_integrator* p, const vk_frame* frame, void* stream)
{
  TimedLaunchScope used_up;
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
  TimedLaunchScope used_up;
  IntegrateParams P;
  const int rc = fill_params(P, v, p, frame, light, mask, false, true, true);
  if (rc != VK_OK) return rc;
  return launch<false, COLOR_LIGHT>(P, v, frame, nullptr, vk_s(stream));
}

int vk_integrate_depth_light(const vk_volume* v, const vk_integrator* p, const vk_light* light,
    const float* mask, const vk_frame* frame, void* stream)
{
  TimedLaunchScope used_up;
  IntegrateParams P;
  const int rc = fill_params(P, v, p, frame, light, mask, true, true, true);
  if (rc != VK_OK) return rc;
  return launch<true, COLOR_LIGHT>(P, v, frame, nullptr, vk_s(stream));
}

int vk_integrate_time_next(void* start_event, void* stop_event)
{
  VK_REQUIRE((start_event == nullptr) == (stop_event == nullptr));
  g_time_start = static_cast<hipEvent_t>(start_event);
  g_time_stop = static_cast<hipEvent_t>(stop_event);
  return VK_OK;
}

int vk_integrate_ahead(const vk_volume* v, const vk_integrator* p, const vk_frame* frame, int color_mode,
    const vk_light* light, const float* mask, const float* light_records, vk_view_bounds* ahead, void* stream)
{
  TimedLaunchScope used_up;
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
