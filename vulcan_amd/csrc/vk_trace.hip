// vk_trace.hip — raycasting the hashed volume back to a depth/colour/normal
// frame on gfx950 (ref: src/tracer.cu, src/tracer.cpp, src/frame.cu).
//
// The reference runs ComputePatches -> (host readback of the patch count) ->
// ComputeBounds with float CAS atomics -> ComputePoints -> ComputeNormals. Here
// the counts stay on the device, and the fused path (vk_trace /
// vk_trace_compute_block_bounds) rasterises each visible block's cell rectangle
// straight into the 80x60 bounds grid with native integer atomic min/max
// (positive floats order like their bit patterns), skipping the patch list.
#include "vk_bounds.hpp"
#include "vk_raycast.hpp"
#include "vk_requests.hpp"

using namespace vk;

namespace
{

// ------------------------------------------------------------------ patches ----

constexpr int kPatchThreads = 256;

// ref: tracer.cu:13-87. Offsets: wave prefix (ballot-free integer scan with
// shuffles) + one atomicAdd per workgroup; util.cuh:52-95 PrefixSum used a
// 2*log2(512)-barrier LDS scan.
__global__ __launch_bounds__(kPatchThreads) void compute_patches_kernel(PatchParams P)
{
  __shared__ int wave_total[kPatchThreads / 64];
  __shared__ int block_base;

  const int block_count = P.block_count_dev ? min(*P.block_count_dev, P.block_count) : P.block_count;
  const int index = blockIdx.x * kPatchThreads + threadIdx.x;
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;

  BlockRect r;
  r.count = 0;
  r.gx = r.gy = 0;

  if (index < block_count)
  {
    const Entry entry = load_entry(P.entries, (uint32_t)P.indices[index]);
    r = block_rect(entry, P.Tcw, P.k, P.block_length, P.min_depth, P.max_depth, P.image_width,
        P.image_height, P.bounds_width, P.bounds_height);
  }

  int incl = r.count;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1)
  {
    const int t = __shfl_up(incl, d);
    if (lane >= d) incl += t;
  }
  if (lane == 63) wave_total[wave] = incl;
  __syncthreads();

  if (threadIdx.x == 0)
  {
    int total = 0;
    for (int w = 0; w < kPatchThreads / 64; ++w)
    {
      const int c = wave_total[w];
      wave_total[w] = total;
      total += c;
    }
    block_base = (total > 0) ? atomicAdd(P.patch_count, total) : 0;
  }
  __syncthreads();

  if (r.count > 0)
  {
    const int offset = block_base + wave_total[wave] + incl - r.count;

    for (int i = 0; i < r.gy; ++i)
      for (int j = 0; j < r.gx; ++j)
      {
        const int output = offset + i * r.gx + j;
        if (output >= P.patch_capacity) continue;
        vk_patch patch;
        patch.origin[0] = (int16_t)(r.bmin_x + VK_PATCH_MAX_SIZE * j);
        patch.origin[1] = (int16_t)(r.bmin_y + VK_PATCH_MAX_SIZE * i);
        patch.size[0] = (int16_t)vmini(VK_PATCH_MAX_SIZE, r.bmax_x - patch.origin[0] + 1);
        patch.size[1] = (int16_t)vmini(VK_PATCH_MAX_SIZE, r.bmax_y - patch.origin[1] + 1);
        patch.bounds[0] = r.near_;
        patch.bounds[1] = r.far_;
        P.patches[output] = patch;
      }
  }
}

// ------------------------------------------------------------------- bounds ----

__global__ __launch_bounds__(256) void reset_bounds_kernel(float2* __restrict__ bounds, int count)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) bounds[i] = make_float2(+FLT_MAX, -FLT_MAX);
}

// Every stored bound is a positive float (clamped to [min_depth, max_depth],
// min_depth > 0) or the +-FLT_MAX initialiser, so signed-integer min/max on the
// bit pattern equals float min/max: -FLT_MAX is a negative int, below every
// positive pattern. One native atomic instead of util.cuh:20-50's CAS loop.
__device__ __forceinline__ void bound_cell(float* bounds, int pixel, float near_, float far_)
{
  int* cell = reinterpret_cast<int*>(bounds) + 2 * pixel;
  const int n = __float_as_int(near_), f = __float_as_int(far_);
  if (n < __hip_atomic_load(cell + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(cell + 0, n);
  if (f > __hip_atomic_load(cell + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(cell + 1, f);
}

// ref: tracer.cu:89-112
__global__ __launch_bounds__(256) void compute_bounds_kernel(const vk_patch* __restrict__ patches,
    float* bounds, int bounds_width, int patch_count, const int32_t* patch_count_dev)
{
  const int count = patch_count_dev ? min(*patch_count_dev, patch_count) : patch_count;
  const int index = blockIdx.x * blockDim.x + threadIdx.x;
  if (index >= count) return;

  const vk_patch patch = patches[index];

  for (int i = 0; i < patch.size[1]; ++i)
  {
    const int y = patch.origin[1] + i;
    for (int j = 0; j < patch.size[0]; ++j)
    {
      const int x = patch.origin[0] + j;
      bound_cell(bounds, y * bounds_width + x, patch.bounds[0], patch.bounds[1]);
    }
  }
}

// Fused ComputePatches + ComputeBounds: the union of a block's patches is a
// rectangle anchored at bmin, so rasterising that rectangle gives bit-identical
// bounds (min/max commute) without materialising the patch list.
__global__ __launch_bounds__(256) void block_bounds_kernel(PatchParams P)
{
  const int block_count = P.block_count_dev ? min(*P.block_count_dev, P.block_count) : P.block_count;
  const int stride = gridDim.x * blockDim.x;

  for (int index = blockIdx.x * blockDim.x + threadIdx.x; index < block_count; index += stride)
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
    for (int y = r.bmin_y; y <= y_end; ++y)
      for (int x = r.bmin_x; x <= x_end; ++x)
        bound_cell(P.bounds, y * P.bounds_width + x, r.near_, r.far_);
  }
}

constexpr int kBoundsThreads = 512;

__global__ __launch_bounds__(kBoundsThreads) void block_bounds_partial_kernel(PatchParams P, float2* __restrict__ partials)
{
  __shared__ int grid[2 * kBoundsMaxCells];
  bounds_group(P, partials, grid, (int)blockIdx.x, kBoundsThreads);
}

__device__ __forceinline__ float2 merged_bound(const float2* __restrict__ partials, int cells, int cell)
{
  float2 b = partials[cell];
#pragma unroll
  for (int g = 1; g < kBoundsGroups; ++g)
  {
    const float2 o = partials[(size_t)g * cells + cell];
    b.x = vmin(o.x, b.x);
    b.y = vmax(o.y, b.y);
  }
  return b;
}

__global__ __launch_bounds__(256) void merge_bounds_kernel(const float2* __restrict__ partials,
    float2* __restrict__ bounds, int cells)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < cells) bounds[c] = merged_bound(partials, cells, c);
}

// ------------------------------------------------------------------- points ----

// Wave-wide integer min / max ending in lane 63 (DPP row operations, no LDS): the
// bounds are positive floats or the +-FLT_MAX initialisers, which order like their bit
// patterns read as signed integers (see bound_cell).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_int(int v)
{
  return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false);   // lanes outside the mask keep v
}

template <bool MAX>
__device__ __forceinline__ int wave_minmax_lane63(int v)
{
#define VK_STEP(CTRL, MASK) { const int o = dpp_int<CTRL, MASK>(v); v = MAX ? (o > v ? o : v) : (o < v ? o : v); }
  VK_STEP(0xB1, 0xf)    // quad_perm [1,0,3,2]
  VK_STEP(0x4E, 0xf)    // quad_perm [2,3,0,1]
  VK_STEP(0x141, 0xf)   // row_half_mirror
  VK_STEP(0x140, 0xf)   // row_mirror
  VK_STEP(0x142, 0xa)   // row_bcast:15 into rows 1 and 3
  VK_STEP(0x143, 0xc)   // row_bcast:31 into rows 2 and 3
#undef VK_STEP
  return v;
}

// min / max of one cell over the kBoundsGroups private grids, for a wave-uniform cell:
// lane g reads group g's copy (one 8-byte load for the whole wave), twelve DPP steps fold
// the 32 values. (64 s_loads + 64 compare/select pairs when each lane did all 32.)
static_assert(kBoundsGroups == 32, "lane g of a wave reads group g & 31");
__device__ __forceinline__ float2 merged_bound_wave(const float2* __restrict__ partials, int cells, int cell)
{
  const float2 mine = partials[(size_t)(lane_id() & 31) * cells + cell];
  const int lo = wave_minmax_lane63<false>(__float_as_int(mine.x));
  const int hi = wave_minmax_lane63<true>(__float_as_int(mine.y));
  return make_float2(__int_as_float(__builtin_amdgcn_readlane(lo, 63)), __int_as_float(__builtin_amdgcn_readlane(hi, 63)));
}

// ref: tracer.cu:317-451. One lane per pixel; a wave covers an 8x8 pixel tile
// (one bounds cell at 640x480 / 80x60) so its rays start at the same depth, run
// a similar number of steps and walk the same few blocks. WAVES waves per workgroup:
// 2x2 tiles (WAVES = 4) or a single tile (WAVES = 1).
// (r02 variants, rocprofv3 averages of 120 launches: 2x2 tiles per workgroup in one band per
// XCD 31.25 us; single tiles 31.46; two half bands per XCD 31.06 / 30.71 with single tiles;
// plain round robin 32.21.)
constexpr int kNormalRows = 128;          // counters behind the bounds in the scratch: 8-pixel rows of tiles, up to 1024 rows
constexpr int kNormalRowStride = 16;      // words: a 64-byte line per counter (the waves' increments and the waiting groups'
                                          // polls of ONE row meet on a line; with 16 counters to a line the polls of a few
                                          // hundred waiting groups held up the increments they were waiting for)
constexpr int kNormalWaitPolls = 1 << 16; // x s_sleep(32): some 60 ms
constexpr int kNormalLateWords = 16;      // one more line behind the counters: word 0 = a group's wait has expired

// The raycast's waves above the request pass's in the launch they share (s_setprio, round 5): the launch is as long as its
// slowest raycast wave, the request pass fills in. 37.75 -> 37.53 us in two alternating pairs of runs (the request pass's waves
// raised instead: 38.2; a march's waves raised after 8 / 16 / 32 passes through the loop: nothing, either scene). -DVK_TR_PRIO=0: without.
#ifndef VK_TR_PRIO
#define VK_TR_PRIO 1
#endif
#ifndef VK_TR_WAVES
#define VK_TR_WAVES 5
#endif
#ifndef VK_TP_KEY_SIDE
#define VK_TP_KEY_SIDE 1     // 0 (experiment): only the FRAME side of the next pyramid rides behind the raycast — nothing waits
#endif
#ifndef VK_TP_WAVES
#define VK_TP_WAVES 6        // trace_and_pyramid_kernel: the raycast alone (compute_points_kernel) runs six waves per SIMD
#endif
#ifndef VK_TRACE_NORMALS_RIDE
#define VK_TRACE_NORMALS_RIDE 1
#endif
#define VK_POINTS_WAVES 1
#define VK_POINTS_ORDER 1
constexpr int kPointsWaves = VK_POINTS_WAVES;
constexpr int kPointsTile = (kPointsWaves == 4) ? 16 : 8;     // pixels per workgroup edge

// The work of ONE workgroup of the raycast: WAVES = 1: an 8 x 8 tile; WAVES = 4: 2 x 2 of them. `group` of `groups`
// workgroups (the launch's own blockIdx / gridDim, or the leading part of a launch that has other work behind it).
template <bool POOL32, int WAVES>
__device__ __forceinline__ void points_group(const PointParams& P, const int group, const int groups, int4 (*directories)[kDirWords])
{
  constexpr int TILE = (WAVES == 4) ? 16 : 8;     // pixels per workgroup edge
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  int4* bdir = directories[wave];
  bdir[lane] = make_int4(INT32_MIN, INT32_MIN, INT32_MIN, -1);   // no block has these coordinates after f2i
  wave_lds_fence();

  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, each
  // with its own L2, and neighbouring tiles march through the same voxel blocks.
  // Workgroup w therefore takes tile (w % 8) * chunk + w / 8, which gives every XCD
  // one contiguous band of the image (placement only affects speed).
  const int tiles_x = (P.image_width + TILE - 1) / TILE, tiles_y = (P.image_height + TILE - 1) / TILE;
  const int tiles = tiles_x * tiles_y;
  const int chunk = (tiles + 7) / 8;
  int tile;
  if (VK_POINTS_ORDER == 0) tile = (group & 7) * chunk + (group >> 3);
  else if (VK_POINTS_ORDER == 1)
  {
    // two half bands per XCD, k and k + 8 of 16: rays near the image border graze the
    // surface and take more steps, so every XCD gets one outer and one inner strip
    const int half = (chunk + 1) / 2;
    const int i = group >> 3, k = group & 7;
    tile = (i < half) ? k * half + i : (8 + k) * half + (i - half);
  }
  else tile = group;                                        // plain round robin
  const int tile_x = tile % tiles_x, tile_y = tile / tiles_x;
  const int x = tile_x * TILE + (WAVES == 4 ? (wave & 1) * 8 : 0) + (lane & 7);
  const int y = tile_y * TILE + (WAVES == 4 ? (wave >> 1) * 8 : 0) + (lane >> 3);

  // This wave's bound first: with a wave-uniform cell (always, when a bounds cell is
  // 8x8 pixels) the kBoundsGroups partial grids are merged by the wave as a whole.
  const bool inside = tile < tiles && x < P.image_width && y < P.image_height;
  const int px = P.bounds_width * vmini(x, P.image_width - 1) / P.image_width;
  const int py = P.bounds_height * vmini(y, P.image_height - 1) / P.image_height;
  const int cell = (tile < tiles) ? py * P.bounds_width + px : 0;
  float2 bound;

  if (P.partials)
  {
    const int cells = P.bounds_width * P.bounds_height;
    const int first = __builtin_amdgcn_readfirstlane(cell);
    if (__all(cell == first)) bound = merged_bound_wave(P.partials, cells, first);
    else bound = merged_bound(P.partials, cells, cell);

    // publish the merged grid (Tracer::bounds_) — every cell, whether or not a
    // pixel maps to it
    const int threads = groups * WAVES * 64;
    for (int c = group * WAVES * 64 + threadIdx.x; c < cells; c += threads)
      P.bounds_out[c] = merged_bound(P.partials, cells, c);
  }
  else
  {
    bound = reinterpret_cast<const float2*>(P.bounds)[cell];
  }

  if (inside) march_ray<false, POOL32>(P, bdir, x, y, bound);
  if (P.rows_done && tile < tiles)
  {
    // this wave's tile is written (its depths through to memory, march_ray): one more of its row of tiles
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0)
      __hip_atomic_fetch_add(&P.rows_done[(tile_y * (TILE / 8) + (WAVES == 4 ? (wave >> 1) : 0)) * kNormalRowStride], 1u,
          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

#ifdef VK_POINTS_WAVES_PER_EU      // experiments only: the kernel held to this many waves per SIMD (spills instead of registers)
#define VK_POINTS_OCCUPANCY __attribute__((amdgpu_waves_per_eu(VK_POINTS_WAVES_PER_EU)))
#else
#define VK_POINTS_OCCUPANCY
#endif
template <bool POOL32>
__global__ __launch_bounds__(kPointsWaves * 64) VK_POINTS_OCCUPANCY void compute_points_kernel(PointParams P)
{
  __shared__ int4 directories[kPointsWaves][kDirWords];
  points_group<POOL32, kPointsWaves>(P, (int)blockIdx.x, (int)gridDim.x, directories);
}

// The raycast's normals (Frame::ComputeNormals of the traced frame, frame.cu:9-122) by TRAILING WORKGROUPS OF THE SAME
// LAUNCH: 64 x 4 pixels each, like compute_normals_kernel. A pixel's taps lie two pixels up, down, left and right, so a
// group needs two 8-pixel rows of tiles complete; it waits for their counters (PointParams::rows_done, bumped by every
// raycast wave behind its tile's depths), then reads the depths where the raycast's waves — on any XCD — wrote them
// through to. The groups come last in the grid, i.e. they are dispatched when every raycast workgroup already has its
// place on the device, so whatever they wait for is running or done; the wait is bounded all the same (a launch whose
// counters were left in a bad state by an aborted one writes wrong normals for a frame instead of hanging the device).
// As a launch of its own the pass costs 5 us of launch floor behind a raycast whose last third leaves the device idle.
// Used when the next frame's request pass sits between the raycast's workgroups and these (vk_trace_ahead_requests): the
// groups are then dispatched late and find most rows complete. Dispatched right behind the raycast's workgroups they wait
// from the first microsecond on, and the launch lasted 46.3 us instead of 29 + 5 (sphere) and 84.4 instead of 69.7 + 5
// (tracking scene): vk_trace_ahead keeps its two launches.
__device__ __forceinline__ float depth_through(const float* depths, int w, int h, int x, int y)
{
  return (x >= 0 && x < w && y >= 0 && y < h) ? __hip_atomic_load(&depths[y * w + x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
}

// The wait is bounded, and its expiry has an OUTCOME (round 5): the group stores nothing — normals from depths that are
// not all there would be wrong without anyone knowing — and says so in the word behind the counters and in the caller's
// pinned word; the host side repairs and reports it (vk_trace_normals_settle, and the check at the start of trace_ahead).
__device__ __forceinline__ void normals_group(const PointParams& P, float* __restrict__ normals, int group_x, int group_y, int* expired)
{
  if (threadIdx.x == 0)
  {
    const int first = vmaxi(group_y * 4 - 2, 0) >> 3, last = vmini(group_y * 4 + 5, P.image_height - 1) >> 3;
    int polls = 0;
    bool late = false;
    for (int row = first; row <= last && !late; ++row)
      while ((int)(__hip_atomic_load(&P.rows_done[row * kNormalRowStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - P.rows_target) < 0)
      {
        if (++polls > P.normal_polls) { late = true; break; }
        __builtin_amdgcn_s_sleep(32);
      }
    *expired = late ? 1 : 0;
    if (late)
    {
      // (the word names the LAUNCH whose wait expired — the record's launch count, never 0 — so that a host that is several
      // launches ahead when it sees the word knows whether the image it can still repair is the one that expired: ADVICE r5)
      __hip_atomic_store(P.late_dev, P.late_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (P.late_host) __hip_atomic_store(P.late_host, (int32_t)P.late_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  __syncthreads();
  if (*expired) return;
  const int x = group_x * 64 + (threadIdx.x & 63);
  const int y = group_y * 4 + (threadIdx.x >> 6);
  if (x >= P.image_width || y >= P.image_height) return;
  const int pad = 2;
  const float depth = depth_through(P.depths, P.image_width, P.image_height, x, y);
  f3 normal = make3(0, 0, 0);
  if (depth > 0)
    normal = normal_from_taps(P.k, x, y, depth,
        depth_through(P.depths, P.image_width, P.image_height, x - pad, y), depth_through(P.depths, P.image_width, P.image_height, x + pad, y),
        depth_through(P.depths, P.image_width, P.image_height, x, y - pad), depth_through(P.depths, P.image_width, P.image_height, x, y + pad));
  const int output = y * P.image_width + x;
  normals[3 * output + 0] = normal.x;
  normals[3 * output + 1] = normal.y;
  normals[3 * output + 2] = normal.z;
}

// The raycast of one frame with the REQUEST PASS OF THE NEXT FRAME'S SetView behind it in the same launch
// (vk_trace_ahead_requests): the first `trace_groups` workgroups are the raycast's (16 x 16 pixels each), the next are the
// request pass's (64 x 4 pixels each, vk_requests.hpp), dispatched as the raycast's waves retire, the last the normals'
// (`normals`, or nullptr). The raycast is as long as
// its slowest wave (mean wave life 17 us, launch 31 us) and leaves most of the device idle for its
// last third; as a launch of its own the request pass (17 us) would start only after that. The two touch disjoint state:
// the raycast reads the table, the voxels and its bounds; the request pass reads the table and the visibility bytes and
// writes visibility bytes, request flags and the light preparation's buffers.
template <bool POOL32, int PREP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(VK_TR_WAVES))) void trace_and_request_kernel(PointParams P,
    RequestParams R, Retry retry, int trace_groups, int request_groups_x, int request_groups, float* normals)
{
  __shared__ int4 directories[4][kDirWords];
  __shared__ int normals_expired;
#ifdef VK_TR_EXTRA_LDS
  // experiment (profiles/r05_two_launch_frame.txt): the static LDS SetView's handle + visibility pass would bring into this
  // launch if it rode here too (handle_listed 16 KiB, later_rounds 20 KiB) — what that does to the raycast's occupancy
  __shared__ int extra_lds[VK_TR_EXTRA_LDS / 4];
  asm volatile("" :: "v"(&extra_lds[threadIdx.x]));
#endif
  if ((int)blockIdx.x < trace_groups)
  {
#if VK_TR_PRIO
    __builtin_amdgcn_s_setprio(2);
#endif
    points_group<POOL32, 4>(P, (int)blockIdx.x, trace_groups, directories);
    return;
  }
  const int g = (int)blockIdx.x - trace_groups;
  if (g < request_groups)
  {
    requests_group<true, PREP>(R, retry, g % request_groups_x, g / request_groups_x);
    return;
  }
  const int n = g - request_groups, normal_groups_x = (P.image_width + 63) / 64;
  normals_group(P, normals, n % normal_groups_x, n / normal_groups_x, &normals_expired);
}

// ---- the NEXT Track's pyramid behind the raycast (round 6, vk_trace_ahead_pyramid) ------------------------------------------
//
// The tracking loop's frame is raycast(i) -> pyramid launch(i + 1) -> two Gauss-Newton loop launches -> ... The pyramid launch
// (vk_icp.hip pyramid_level_kernel: the input frame's normal image and half-resolution level, the key frame's normal image and
// half-resolution level) is 7 us of launch-floor work in a strict chain, and most of it needs nothing but the INPUT frame —
// which is known while frame i is still being raycast. Here it rides behind the raycast's workgroups in the raycast's launch:
//   [raycast: 16 x 16 pixel groups] [frame side: normals, then half level — plain loads, they wait for nobody]
//   [key side: the raycast's normal image (normals_group) and its half level — per row of tiles behind the row counters]
// The key side waits exactly as normals_group does, with the same bounded wait and the same outcome (VK_ERR_TIMEOUT through
// the record's pinned word; the Track that consumed the level is then not to be trusted, vk.h). The bits are those of
// pyramid_level_kernel: the same normal_from_taps on the same taps, nearest sampling at (2x, 2y).
struct PyramidRide
{
  const float* frame_depths;       // the NEXT input frame
  float* frame_normals;            // written: Frame::ComputeNormals
  float* frame_half_depth;         // written: Frame::Downsample (nearest)
  float* frame_half_normals;
  int frame_w, frame_h;
  vk_projection frame_k;
  float* key_half_depth;           // written behind the row counters: the raycast image's half level
  float* key_half_normals;
  int groups_frame_full, groups_frame_half, groups_key_full, groups_key_half;    // workgroups of 64 x 4 pixels each
};

__device__ __forceinline__ float ride_depth_at(const float* depths, int w, int h, int x, int y)
{
  return (x >= 0 && x < w && y >= 0 && y < h) ? depths[y * w + x] : 0.0f;
}

// one side's normal at full-resolution pixel (x, y): compute_normals_kernel's expressions (ref: frame.cu:9-122)
template <typename K>
__device__ __forceinline__ f3 ride_normal(const float* depths, const K& k, int w, int h, int x, int y)
{
  const int pad = 2;
  const float depth = depths[y * w + x];
  f3 normal = make3(0, 0, 0);
  if (depth > 0)
    normal = normal_from_taps(k, x, y, depth, ride_depth_at(depths, w, h, x - pad, y), ride_depth_at(depths, w, h, x + pad, y),
        ride_depth_at(depths, w, h, x, y - pad), ride_depth_at(depths, w, h, x, y + pad));
  return normal;
}

__device__ __forceinline__ void ride_frame_group(const PyramidRide& Y, int g)
{
  if (g < Y.groups_frame_full)
  {
    const int gx = (Y.frame_w + 63) / 64;
    const int x = (g % gx) * 64 + (threadIdx.x & 63), y = (g / gx) * 4 + (threadIdx.x >> 6);
    if (x >= Y.frame_w || y >= Y.frame_h) return;
    const f3 n = ride_normal(Y.frame_depths, Y.frame_k, Y.frame_w, Y.frame_h, x, y);
    float* out = Y.frame_normals + 3 * ((size_t)y * Y.frame_w + x);
    out[0] = n.x;  out[1] = n.y;  out[2] = n.z;
    return;
  }
  g -= Y.groups_frame_full;
  const int hw = Y.frame_w / 2, hh = Y.frame_h / 2, gx = (hw + 63) / 64;
  const int x = (g % gx) * 64 + (threadIdx.x & 63), y = (g / gx) * 4 + (threadIdx.x >> 6);
  if (x >= hw || y >= hh) return;
  Y.frame_half_depth[y * hw + x] = Y.frame_depths[(2 * y) * Y.frame_w + 2 * x];
  // (the half-resolution normal is computed at the pixel it is sampled from: the same normal, bit for bit — the full image is
  // being written by other workgroups of this launch)
  const f3 n = ride_normal(Y.frame_depths, Y.frame_k, Y.frame_w, Y.frame_h, 2 * x, 2 * y);
  float* out = Y.frame_half_normals + 3 * ((size_t)y * hw + x);
  out[0] = n.x;  out[1] = n.y;  out[2] = n.z;
}

// the wait of a key-side group for the 8-pixel rows of tiles [first, last] (normals_group's, factored out)
__device__ __forceinline__ bool ride_rows_complete(const PointParams& P, int first, int last, int* expired)
{
  if (threadIdx.x == 0)
  {
    int polls = 0;
    bool late = false;
    for (int row = first; row <= last && !late; ++row)
      while ((int)(__hip_atomic_load(&P.rows_done[row * kNormalRowStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - P.rows_target) < 0)
      {
        if (++polls > P.normal_polls) { late = true; break; }
        __builtin_amdgcn_s_sleep(32);
      }
    *expired = late ? 1 : 0;
    if (late)
    {
      // (the word names the LAUNCH whose wait expired — the record's launch count, never 0 — so that a host that is several
      // launches ahead when it sees the word knows whether the image it can still repair is the one that expired: ADVICE r5)
      __hip_atomic_store(P.late_dev, P.late_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (P.late_host) __hip_atomic_store(P.late_host, (int32_t)P.late_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  __syncthreads();
  return *expired == 0;
}

__device__ __forceinline__ void ride_key_half_group(const PointParams& P, const PyramidRide& Y, int g, int* expired)
{
  const int hw = P.image_width / 2, hh = P.image_height / 2, gx = (hw + 63) / 64;
  const int group_y = g / gx;
  // half rows [4 gy, 4 gy + 3] sample full rows 8 gy .. 8 gy + 6, their normals' taps two rows further
  const int first = vmaxi(group_y * 8 - 2, 0) >> 3, last = vmini(group_y * 8 + 8, P.image_height - 1) >> 3;
  if (!ride_rows_complete(P, first, last, expired)) return;
  const int x = (g % gx) * 64 + (threadIdx.x & 63), y = group_y * 4 + (threadIdx.x >> 6);
  if (x >= hw || y >= hh) return;
  const int w = P.image_width, h = P.image_height, sx = 2 * x, sy = 2 * y, pad = 2;
  const float depth = depth_through(P.depths, w, h, sx, sy);
  Y.key_half_depth[y * hw + x] = depth;
  f3 normal = make3(0, 0, 0);
  if (depth > 0)
    normal = normal_from_taps(P.k, sx, sy, depth, depth_through(P.depths, w, h, sx - pad, sy), depth_through(P.depths, w, h, sx + pad, sy),
        depth_through(P.depths, w, h, sx, sy - pad), depth_through(P.depths, w, h, sx, sy + pad));
  float* out = Y.key_half_normals + 3 * ((size_t)y * hw + x);
  out[0] = normal.x;  out[1] = normal.y;  out[2] = normal.z;
}

template <bool POOL32>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(VK_TP_WAVES))) void trace_and_pyramid_kernel(PointParams P,
    PyramidRide Y, int trace_groups, float* normals)
{
  __shared__ int4 directories[4][kDirWords];
  __shared__ int normals_expired;
  if ((int)blockIdx.x < trace_groups)
  {
#if VK_TR_PRIO
    __builtin_amdgcn_s_setprio(2);
#endif
    points_group<POOL32, 4>(P, (int)blockIdx.x, trace_groups, directories);
    return;
  }
  int g = (int)blockIdx.x - trace_groups;
  if (g < Y.groups_frame_full + Y.groups_frame_half) { ride_frame_group(Y, g);  return; }
  g -= Y.groups_frame_full + Y.groups_frame_half;
  if (g < Y.groups_key_full)
  {
    const int normal_groups_x = (P.image_width + 63) / 64;
    normals_group(P, normals, g % normal_groups_x, g / normal_groups_x, &normals_expired);
    return;
  }
  ride_key_half_group(P, Y, g - Y.groups_key_full, &normals_expired);
}

// ------------------------------------------------------------------ normals ----

__device__ __forceinline__ float depth_at(const float* depths, int w, int h, int x, int y)
{
  return (x >= 0 && x < w && y >= 0 && y < h) ? depths[y * w + x] : 0.0f;
}

// ref: frame.cu:9-122. The +-2 px taps are read through L1/L2 (each depth value
// is used by 5 pixels of neighbouring rows/columns) instead of a 20x20 LDS tile.
__global__ __launch_bounds__(256) void compute_normals_kernel(const float* __restrict__ depths,
    vk_projection k, float* __restrict__ normals, int image_width, int image_height)
{
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= image_width || y >= image_height) return;

  const int pad = 2;
  const float depth = depths[y * image_width + x];
  f3 normal = make3(0, 0, 0);
  if (depth > 0)
    normal = normal_from_taps(k, x, y, depth,
        depth_at(depths, image_width, image_height, x - pad, y), depth_at(depths, image_width, image_height, x + pad, y),
        depth_at(depths, image_width, image_height, x, y - pad), depth_at(depths, image_width, image_height, x, y + pad));

  const int output = y * image_width + x;
  normals[3 * output + 0] = normal.x;
  normals[3 * output + 1] = normal.y;
  normals[3 * output + 2] = normal.z;
}

// ref: frame.cu:126-181 (7x7 bilateral; expf is not bit-reproducible across
// libm implementations, parity is to 1e-6 relative)
__global__ __launch_bounds__(256) void filter_depths_kernel(int image_width, int image_height,
    const float* __restrict__ src, float* __restrict__ dst)
{
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= image_width || y >= image_height) return;

  const int pad = 3;
  const float d0 = src[y * image_width + x];
  float dn = 0;
  float w = 0;

  for (int i = -pad; i <= pad; ++i)
    for (int j = -pad; j <= pad; ++j)
    {
      const float dk = depth_at(src, image_width, image_height, x + j, y + i);
      const float delta = d0 - dk;
      float sq = 0;
      sq += (float)i * (float)i;
      sq += (float)j * (float)j;
      const float wr = expf(-sq / (pad * pad));
      const float ws = expf(-(delta * delta) / 0.0004f);
      const float ww = wr * ws;
      dn += ww * dk;
      w += ww;
    }

  dst[y * image_width + x] = dn / w;
}

// Fused patches+bounds. `partials` (device, kBoundsGroups grids) selects the
// LDS-privatised path; with merge == true the merged grid is also written to
// `bounds` by a small kernel (stand-alone API), otherwise the consumer merges.
int launch_block_bounds(PatchParams& P, float* bounds, float2* partials, bool merge, hipStream_t s)
{
  const int cells = P.bounds_width * P.bounds_height;

  if (partials && cells <= kBoundsMaxCells)
  {
    hipLaunchKernelGGL(block_bounds_partial_kernel, dim3(kBoundsGroups), dim3(kBoundsThreads), 0, s, P, partials);
    VK_LAUNCH_CHECK();
    if (merge)
    {
      hipLaunchKernelGGL(merge_bounds_kernel, dim3((cells + 255) / 256), dim3(256), 0, s, partials,
          reinterpret_cast<float2*>(bounds), cells);
      VK_LAUNCH_CHECK();
    }
    return VK_OK;
  }

  hipLaunchKernelGGL(reset_bounds_kernel, dim3((cells + 255) / 256), dim3(256), 0, s,
      reinterpret_cast<float2*>(bounds), cells);
  VK_LAUNCH_CHECK();
  if (P.block_count == 0) return VK_OK;
  P.bounds = bounds;
  int blocks = (P.block_count + 255) / 256;
  if (blocks > kCUs * 4) blocks = kCUs * 4;
  hipLaunchKernelGGL(block_bounds_kernel, dim3(blocks), dim3(256), 0, s, P);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

// `normals` with `rows_done`: the normals of the traced frame by trailing workgroups of the raycast's launch (normals_group);
// the caller has set the counters' target. `next_requests`: the next frame's request pass in the same launch as well.
int launch_points(const vk_hash_entry* entries, const vk_voxel* voxels, const float* bounds,
    const float2* partials, int block_count, float block_length, float voxel_length, float trunc_length,
    const vk_transform* Twc, const vk_projection* projection, float* depths, float* colors,
    int image_width, int image_height, int bounds_width, int bounds_height, unsigned long long pool_bytes,
    hipStream_t s, float* normals = nullptr, uint32_t* rows_done = nullptr, uint32_t rows_target = 0,
    const RequestParams* next_requests = nullptr, const Retry* next_retry = nullptr, int next_prep = 0,
    int32_t* late_host = nullptr, int normal_polls = kNormalWaitPolls, const PyramidRide* ride = nullptr, uint32_t late_tag = 1)
{
  PointParams P;
  P.late_tag = late_tag ? late_tag : 1u;
  P.entries = entries;
  P.voxels = voxels;
  P.bounds = bounds;
  P.partials = partials;
  P.bounds_out = reinterpret_cast<float2*>(const_cast<float*>(bounds));
  P.K = (uint32_t)block_count;
  P.block_length = block_length;
  P.voxel_length = voxel_length;
  P.trunc_length = trunc_length;
  P.inv_block_length = 1.0 / (double)block_length;   // correctly rounded: vk_raycast.hpp div_uniform
  P.inv_voxel_length = 1.0 / (double)voxel_length;
  P.touched = nullptr;
  P.march_steps = nullptr;
  P.trip_log = nullptr;
  P.trip_log_passes = 0;
  P.rows_done = (normals && rows_done && (next_requests || ride)) ? rows_done : nullptr;   // (a ride of the frame side only: no counters)
  P.rows_target = rows_target;
  P.late_dev = P.rows_done ? P.rows_done + kNormalRows * kNormalRowStride : nullptr;
  P.late_host = late_host;
  P.normal_polls = normal_polls;
  P.Twc = make_rt(Twc->m);
  P.Tcw = make_rt(Twc->inv);  // tracer.cu:350 Twc.Inverse()
  P.k = make_projection(*projection);
  P.depths = depths;
  P.colors = colors;
  P.image_width = image_width;
  P.image_height = image_height;
  P.bounds_width = bounds_width;
  P.bounds_height = bounds_height;
  const bool pool32 = pool_bytes <= 0xffffffffull;   // a pool under 4 GiB is addressed with 32-bit offsets from a scalar base
  if (ride)
  {
    // workgroups of 256: 16 x 16 pixel tiles of the raycast, then the next Track's pyramid (frame side, key side)
    if (!P.rows_done && ride->groups_key_full + ride->groups_key_half > 0) return VK_ERR_ARGUMENT;
    const int tiles16 = ((image_width + 15) / 16) * ((image_height + 15) / 16);
    int chunk16 = (tiles16 + 7) / 8;
    if (VK_POINTS_ORDER == 1) chunk16 = 2 * ((chunk16 + 1) / 2);
    const int trace_groups = 8 * chunk16;
    const dim3 grid2(trace_groups + ride->groups_frame_full + ride->groups_frame_half + ride->groups_key_full + ride->groups_key_half);
    if (pool32) hipLaunchKernelGGL(trace_and_pyramid_kernel<true>, grid2, dim3(256), 0, s, P, *ride, trace_groups, normals);
    else hipLaunchKernelGGL(trace_and_pyramid_kernel<false>, grid2, dim3(256), 0, s, P, *ride, trace_groups, normals);
    VK_LAUNCH_CHECK();
    return VK_OK;
  }
  if (next_requests)
  {
    // workgroups of 256: 16 x 16 pixel tiles of the raycast, then 64 x 4 pixel groups of the request pass and the normals
    const int tiles16 = ((image_width + 15) / 16) * ((image_height + 15) / 16);
    int chunk16 = (tiles16 + 7) / 8;
    if (VK_POINTS_ORDER == 1) chunk16 = 2 * ((chunk16 + 1) / 2);
    const int trace_groups = 8 * chunk16;
    const int normal_groups = P.rows_done ? ((image_width + 63) / 64) * ((image_height + 3) / 4) : 0;
    const int gx = (next_requests->width + 63) / 64, gy = (next_requests->height + 3) / 4;
    const dim3 grid2(trace_groups + gx * gy + normal_groups);
    float* fused_normals = P.rows_done ? normals : nullptr;
#define VK_LAUNCH_TR(POOL, PREP) hipLaunchKernelGGL((trace_and_request_kernel<POOL, PREP>), grid2, dim3(256), 0, s, P, *next_requests, *next_retry, trace_groups, gx, gx * gy, fused_normals)
    if (pool32) { if (next_prep == 2) VK_LAUNCH_TR(true, 2); else if (next_prep == 1) VK_LAUNCH_TR(true, 1); else VK_LAUNCH_TR(true, 0); }
    else { if (next_prep == 2) VK_LAUNCH_TR(false, 2); else if (next_prep == 1) VK_LAUNCH_TR(false, 1); else VK_LAUNCH_TR(false, 0); }
#undef VK_LAUNCH_TR
    VK_LAUNCH_CHECK();
    return VK_OK;
  }
  const int tiles = ((image_width + kPointsTile - 1) / kPointsTile) * ((image_height + kPointsTile - 1) / kPointsTile);
  int chunk = (tiles + 7) / 8;
  if (VK_POINTS_ORDER == 1) chunk = 2 * ((chunk + 1) / 2);
  const dim3 grid(8 * chunk);   // padded so every XCD gets an equal share
  if (pool32)
    hipLaunchKernelGGL(compute_points_kernel<true>, grid, dim3(kPointsWaves * 64), 0, s, P);
  else
    hipLaunchKernelGGL(compute_points_kernel<false>, grid, dim3(kPointsWaves * 64), 0, s, P);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

int launch_normals(const float* depths, const vk_projection* projection, float* normals,
    int image_width, int image_height, hipStream_t s)
{
  const dim3 grid((image_width + 63) / 64, (image_height + 3) / 4);
  hipLaunchKernelGGL(compute_normals_kernel, grid, dim3(256), 0, s, depths, *projection, normals,
      image_width, image_height);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

}  // namespace

extern "C" {

int vk_trace_compute_patches(const int32_t* indices, const vk_hash_entry* entries,
    const vk_transform* Tcw, const vk_projection* projection, float block_length, float min_depth,
    float max_depth, int block_count, const int32_t* block_count_dev, int image_width,
    int image_height, int bounds_width, int bounds_height, vk_patch* patches, int patch_capacity,
    int32_t* patch_count, void* stream)
{
  PatchParams P;
  const int rc = fill_patch_params(P, indices, entries, Tcw, projection, block_length, min_depth,
      max_depth, block_count, block_count_dev, image_width, image_height, bounds_width, bounds_height);
  if (rc != VK_OK) return rc;
  VK_REQUIRE(patches && patch_count && patch_capacity > 0);
  if (block_count == 0) return VK_OK;
  P.patches = patches;
  P.patch_capacity = patch_capacity;
  P.patch_count = patch_count;
  hipLaunchKernelGGL(compute_patches_kernel, dim3((block_count + kPatchThreads - 1) / kPatchThreads),
      dim3(kPatchThreads), 0, vk_s(stream), P);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

int vk_trace_compute_bounds(const vk_patch* patches, float* bounds, int bounds_width,
    int patch_count, const int32_t* patch_count_dev, void* stream)
{
  VK_REQUIRE(patches && bounds && bounds_width > 0 && patch_count >= 0);
  if (patch_count == 0) return VK_OK;
  hipLaunchKernelGGL(compute_bounds_kernel, dim3((patch_count + 255) / 256), dim3(256), 0,
      vk_s(stream), patches, bounds, bounds_width, patch_count, patch_count_dev);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

int vk_trace_reset_bounds(float* bounds, int count, void* stream)
{
  VK_REQUIRE(bounds && count > 0);
  hipLaunchKernelGGL(reset_bounds_kernel, dim3((count + 255) / 256), dim3(256), 0, vk_s(stream),
      reinterpret_cast<float2*>(bounds), count);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

int vk_trace_compute_block_bounds(const int32_t* indices, const vk_hash_entry* entries,
    const vk_transform* Tcw, const vk_projection* projection, float block_length, float min_depth,
    float max_depth, int block_count, const int32_t* block_count_dev, int image_width,
    int image_height, int bounds_width, int bounds_height, float* bounds, void* stream)
{
  PatchParams P;
  const int rc = fill_patch_params(P, indices, entries, Tcw, projection, block_length, min_depth,
      max_depth, block_count, block_count_dev, image_width, image_height, bounds_width, bounds_height);
  if (rc != VK_OK) return rc;
  VK_REQUIRE(bounds);
  return launch_block_bounds(P, bounds, nullptr, false, vk_s(stream));
}

int vk_trace_compute_points(const vk_hash_entry* entries, const vk_voxel* voxels, const float* bounds,
    int block_count, float block_length, float voxel_length, float trunc_length,
    const vk_transform* Twc, const vk_projection* projection, float* depths, float* colors,
    int image_width, int image_height, int bounds_width, int bounds_height, void* stream)
{
  VK_REQUIRE(entries && voxels && bounds && Twc && projection && depths && colors);
  VK_REQUIRE(block_count > 0 && image_width > 0 && image_height > 0 && bounds_width > 0 && bounds_height > 0);
  VK_REQUIRE(block_length > 0 && voxel_length > 0);
  return launch_points(entries, voxels, bounds, nullptr, block_count, block_length, voxel_length, trunc_length,
      Twc, projection, depths, colors, image_width, image_height, bounds_width, bounds_height,
      ~0ull /* the free function is not told how large the pool is */, vk_s(stream));
}

int vk_frame_compute_normals(const float* depths, const vk_projection* projection, float* normals,
    int image_width, int image_height, void* stream)
{
  VK_REQUIRE(depths && projection && normals && image_width > 0 && image_height > 0);
  return launch_normals(depths, projection, normals, image_width, image_height, vk_s(stream));
}

int vk_frame_filter_depths(int image_width, int image_height, const float* src, float* dst, void* stream)
{
  VK_REQUIRE(src && dst && src != dst && image_width > 0 && image_height > 0);
  const dim3 grid((image_width + 63) / 64, (image_height + 3) / 4);
  hipLaunchKernelGGL(filter_depths_kernel, grid, dim3(256), 0, vk_s(stream), image_width, image_height, src, dst);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

size_t vk_trace_bounds_floats(int bounds_width, int bounds_height)
{
  if (bounds_width <= 0 || bounds_height <= 0) return 0;
  const size_t cells = (size_t)bounds_width * bounds_height;
  // the merged grid, the private grids of the fused bounds pass, and the row counters of the raycast's normals (normals_group)
  // ... and the line that says a normals workgroup's wait has expired
  return 2 * cells * (cells <= (size_t)kBoundsMaxCells ? 1 + kBoundsGroups : 1) + kNormalRows * kNormalRowStride + kNormalLateWords;
}

int vk_trace(const vk_volume* v, const vk_frame* frame, float min_depth, float max_depth,
    float* bounds, int bounds_width, int bounds_height, float* out_depth, float* out_color,
    float* out_normals, void* stream)
{
  VK_REQUIRE(v && frame && bounds && out_depth && out_color && out_normals);
  VK_REQUIRE(v->hash_entries && v->voxels && v->visible_blocks && v->counters);
  hipStream_t s = vk_s(stream);
  const float block_length = VK_BLOCK_RESOLUTION * v->voxel_length;
  const int max_count = v->main_block_count + v->excess_block_count;

  // tracer.cpp:49-76 ComputePatches + ComputeBounds, fused, count read on device
  vk_transform Tcw;
  for (int i = 0; i < 16; ++i) { Tcw.m[i] = frame->depth_to_world.inv[i]; Tcw.inv[i] = frame->depth_to_world.m[i]; }
  PatchParams P;
  int rc = fill_patch_params(P, v->visible_blocks, v->hash_entries, &Tcw, &frame->depth_projection,
      block_length, min_depth, max_depth, max_count, v->counters + VK_CTR_VISIBLE, frame->width,
      frame->height, bounds_width, bounds_height);
  if (rc != VK_OK) return rc;
  // `bounds` = merged grid followed by kBoundsGroups private grids (vk_trace_bounds_floats)
  const int cells = bounds_width * bounds_height;
  float2* partials = (cells <= kBoundsMaxCells) ? reinterpret_cast<float2*>(bounds) + cells : nullptr;
  if ((rc = launch_block_bounds(P, bounds, partials, false, s)) != VK_OK) return rc;

  // tracer.cpp:78-95 ComputePoints (merges the private grids on the fly)
  if ((rc = launch_points(v->hash_entries, v->voxels, bounds, partials, v->main_block_count, block_length,
           v->voxel_length, v->truncation_length, &frame->depth_to_world, &frame->depth_projection,
           out_depth, out_color, frame->width, frame->height, bounds_width, bounds_height,
           (unsigned long long)max_count * VK_BLOCK_VOXELS * sizeof(vk_voxel), s)) != VK_OK)
    return rc;

  // tracer.cpp:97-100 ComputeNormals
  return launch_normals(out_depth, &frame->depth_projection, out_normals, frame->width, frame->height, s);
}

// the row counters and the expiry line behind the grids of a tracer's scratch (nullptr: a grid too large for private copies)
static uint32_t* normals_counters(const vk_view_bounds* ahead)
{
  const int cells = ahead->bounds_width * ahead->bounds_height;
  if (cells > kBoundsMaxCells) return nullptr;
  return reinterpret_cast<uint32_t*>(ahead->scratch + 2 * (size_t)cells * (1 + kBoundsGroups));
}

// A normals workgroup's wait has expired (the stream has been synchronised by the caller): counters, expiry words and
// launch count start over, and the normals of the last traced image are made by a launch of their own.
static int normals_repair(vk_view_bounds* ahead, hipStream_t s)
{
  // (the launches the counters count ran on counted_stream: it is drained before the counters are zeroed under them)
  if (ahead->counted_stream && ahead->counted_stream != (const void*)s)
    VK_CHECK(hipStreamSynchronize(reinterpret_cast<hipStream_t>(const_cast<void*>(ahead->counted_stream))));
  uint32_t* counters = normals_counters(ahead);
  if (counters) VK_CHECK(hipMemsetAsync(counters, 0, (kNormalRows * kNormalRowStride + kNormalLateWords) * sizeof(uint32_t), s));
  if (ahead->late_host) __atomic_store_n(ahead->late_host, 0, __ATOMIC_RELAXED);
  ahead->trace_launches = 0;
  // What is guaranteed is the ERROR. The image that can still be repaired is the LAST launch's (last_*): in a pipelined loop
  // the host may be several launches ahead when it sees the word, and the image whose groups expired (the word names its
  // launch, vk.h) may be an earlier one — its riding normals are then incomplete and stay so; the caller that kept that
  // image recomputes them (vk_frame_compute_normals).
  if (ahead->last_depths && ahead->last_normals && ahead->last_width > 0 && ahead->last_height > 0)
  {
    const int rc = launch_normals(ahead->last_depths, &ahead->last_projection, ahead->last_normals, ahead->last_width,
        ahead->last_height, s);
    if (rc != VK_OK) return rc;
  }
  return VK_ERR_TIMEOUT;
}

static int trace_ahead(const vk_volume* v, const vk_frame* frame, vk_view_bounds* ahead, float* out_depth,
    float* out_color, float* out_normals, void* stream, const RequestParams* next_requests, const Retry* next_retry, int next_prep,
    const PyramidRide* ride = nullptr, bool* rode = nullptr)
{
  // (out_normals == nullptr: the caller takes care of the normal image itself — vk_trace_ahead only)
  VK_REQUIRE(v && frame && ahead && ahead->scratch && out_depth && out_color && (out_normals || !next_requests));
  VK_REQUIRE(v->hash_entries && v->voxels && v->visible_blocks && v->counters);
  VK_REQUIRE(ahead->bounds_width > 0 && ahead->bounds_height > 0);
  hipStream_t s = vk_s(stream);
  // the last launch's riding normals: did a group give up? (the pinned word, no synchronisation unless it is set)
  if (ahead->late_host && __atomic_load_n(ahead->late_host, __ATOMIC_RELAXED) != 0)
  {
    VK_CHECK(hipStreamSynchronize(s));
    return normals_repair(ahead, s);      // VK_ERR_TIMEOUT, nothing of this call launched: the caller repeats it
  }
  const float block_length = VK_BLOCK_RESOLUTION * v->voxel_length;
  const int cells = ahead->bounds_width * ahead->bounds_height;
  float* bounds = ahead->scratch;
  float2* partials = (cells <= kBoundsMaxCells) ? reinterpret_cast<float2*>(bounds) + cells : nullptr;
  int rc;

  if (!(partials && view_matches(ahead, v, frame)))
  {
    // not computed ahead (or for another view): tracer.cpp:49-76 now
    ahead->valid = 0;
    PatchParams P;
    if ((rc = view_patch_params(P, v, frame, ahead)) != VK_OK) return rc;
    if ((rc = launch_block_bounds(P, bounds, partials, false, s)) != VK_OK) return rc;
    if (partials) view_record(ahead, v, frame);
  }

  // the normals in the raycast's own launch (normals_group): counters behind the grids in the scratch, one per 8-pixel row
  // of tiles, never reset — launch n of this scratch and image size waits for n times the waves of a row
  const int rows = 2 * ((frame->height + 15) / 16), waves_per_row = 2 * ((frame->width + 15) / 16);
  // (only with the next frame's request pass between the raycast's workgroups and the normals': dispatched right behind
  // the raycast's, the waiting groups cost it 12 to 17 us — profiles/r04_trace_normals_ride.txt)
  const bool ride_waits = ride && ride->groups_key_full + ride->groups_key_half > 0;
  const bool normals_ride = VK_TRACE_NORMALS_RIDE && (next_requests || ride_waits) && out_normals && partials && rows <= kNormalRows;
  if (ride_waits && !normals_ride) ride = nullptr;       // (a grid too large for the counters: the caller's pyramid launch stays)
  if (rode) *rode = ride != nullptr;
  uint32_t* rows_done = nullptr;
  uint32_t rows_target = 0;
  if (normals_ride)
  {
    rows_done = reinterpret_cast<uint32_t*>(bounds + 2 * (size_t)cells * (1 + kBoundsGroups));
    // (another stream than the last launch's: the counters count launches in STREAM order — they start over on the new
    // one; vk.h says the caller has let the old stream's launch complete)
    if (ahead->counted_scratch != bounds || ahead->counted_width != frame->width || ahead->counted_height != frame->height ||
        ahead->counted_stream != stream)
    {
      VK_CHECK(hipMemsetAsync(rows_done, 0, (kNormalRows * kNormalRowStride + kNormalLateWords) * sizeof(uint32_t), s));
      ahead->counted_scratch = bounds;
      ahead->counted_width = frame->width;
      ahead->counted_height = frame->height;
      ahead->counted_stream = stream;
      ahead->trace_launches = 0;
    }
    rows_target = (ahead->trace_launches + 1u) * (uint32_t)waves_per_row;   // (wraps with the counters)
  }
  // vk_test_hooks.force_normals_expiry, once: a target no counter reaches and no patience
  int normal_polls = kNormalWaitPolls;
  if (normals_ride && vk_hook_take(VK_HOOK_FORCE_NORMALS_EXPIRY) == 1)
  {
    rows_target += 0x40000000u;
    normal_polls = 0;
  }
  if ((rc = launch_points(v->hash_entries, v->voxels, bounds, partials, v->main_block_count, block_length,
           v->voxel_length, v->truncation_length, &frame->depth_to_world, &frame->depth_projection,
           out_depth, out_color, frame->width, frame->height, ahead->bounds_width, ahead->bounds_height,
           (unsigned long long)(v->main_block_count + v->excess_block_count) * VK_BLOCK_VOXELS * sizeof(vk_voxel), s,
           normals_ride ? out_normals : nullptr, rows_done, rows_target, next_requests, next_retry, next_prep,
           ahead->late_host, normal_polls, ride, (ahead->trace_launches + 1u) & 0x7fffffffu)) != VK_OK)
    return rc;
  if (normals_ride)
  {
    ++ahead->trace_launches;
    // which images the riding normals belong to: what a repair recomputes
    ahead->last_depths = out_depth;
    ahead->last_normals = out_normals;
    ahead->last_width = frame->width;
    ahead->last_height = frame->height;
    ahead->last_projection = frame->depth_projection;
    return VK_OK;
  }
  ahead->last_depths = nullptr;             // the normals were a launch of their own: nothing rides, nothing to repair
  ahead->last_normals = nullptr;
  if (!out_normals) return VK_OK;
  return launch_normals(out_depth, &frame->depth_projection, out_normals, frame->width, frame->height, s);
}

int vk_trace_ahead(const vk_volume* v, const vk_frame* frame, vk_view_bounds* ahead, float* out_depth,
    float* out_color, float* out_normals, void* stream)
{
  return trace_ahead(v, frame, ahead, out_depth, out_color, out_normals, stream, nullptr, nullptr, 0);
}

int vk_trace_ahead_requests(const vk_volume* v, const vk_frame* view, vk_view_bounds* ahead, float* out_depth,
    float* out_color, float* out_normals, const vk_frame* next, vk_light_prep* next_prep, vk_requests_ahead* requests,
    void* stream)
{
  VK_REQUIRE(v && next && requests);
  // a record that still announces a frame: that frame's requests are in the volume and its SetView has not run — a second
  // pass on top would mix two frames' requests (vk.h vk_requests_ahead_cancel is the way out). Nothing is launched.
  if (requests->valid == 1) return VK_ERR_ARGUMENT;
  requests->valid = 0;
  // the pass is made ahead only in the form SetView itself would give it without further launches: a frame that names
  // its content, and — when its normals are still to be computed — a preparation that rides (the normals come with it)
  const bool ride = prep_rides(next_prep, next);
  const bool possible = next->depth && next->width > 0 && next->height > 0 && next->content_id != 0 && v->counters &&
      v->allocation_types && v->allocation_blocks && v->block_visibility &&
      !(next_prep && next_prep->normals_out && (!ride || next_prep->normals_out != next->normals));
  if (!possible) return vk_trace_ahead(v, view, ahead, out_depth, out_color, out_normals, stream);
  RequestParams R;
  Retry retry;
  const int with_prep = build_request_pass(R, retry, v, next->depth, next->width, next->height, &next->depth_projection,
      &next->depth_to_world, ride ? next : nullptr, ride ? next_prep : nullptr, true);
  if (next_prep) next_prep->valid = 0;
  const int rc = trace_ahead(v, view, ahead, out_depth, out_color, out_normals, stream, &R, &retry, with_prep);
  if (rc != VK_OK) return rc;
  if (ride) prep_note_made(next_prep, next);
  requests->counters = v->counters;
  requests->depth = next->depth;
  requests->prep = ride ? next_prep : nullptr;
  requests->width = next->width;
  requests->height = next->height;
  requests->depth_projection = next->depth_projection;
  requests->depth_to_world = next->depth_to_world;
  requests->content_id = next->content_id;
  requests->normals_made = with_prep == 2 ? 1 : 0;      // the pass wrote next->normals (prep->normals_out rode)
  requests->pose_on_device = 0;
  requests->valid = 1;
  return VK_OK;
}

int vk_trace_ahead_pyramid(const vk_volume* v, const vk_frame* frame, vk_view_bounds* ahead, float* out_depth, float* out_color,
    float* out_normals, const vk_icp_view* next, float* pyramid, vk_pyramid_ahead* built, void* stream)
{
  VK_REQUIRE(v && frame && next && pyramid && built && out_normals);
  built->valid = 0;
  VK_REQUIRE(next->depths && next->normals && next->width > 0 && next->height > 0);
  VK_REQUIRE(((frame->width | frame->height | next->width | next->height) & 1) == 0);
  // vk_icp_pyramid_track*'s layout of `pyramid`: keyframe level (depth, normals), then the frame's
  const int kw = frame->width / 2, kh = frame->height / 2, fw = next->width / 2, fh = next->height / 2;
  PyramidRide Y;
  Y.key_half_depth = pyramid;
  Y.key_half_normals = pyramid + (size_t)kw * kh;
  Y.frame_half_depth = pyramid + 4 * (size_t)kw * kh;
  Y.frame_half_normals = Y.frame_half_depth + (size_t)fw * fh;
  Y.frame_depths = next->depths;
  Y.frame_normals = const_cast<float*>(next->normals);
  Y.frame_w = next->width;
  Y.frame_h = next->height;
  Y.frame_k = next->projection;
  Y.groups_frame_full = ((next->width + 63) / 64) * ((next->height + 3) / 4);
  Y.groups_frame_half = ((fw + 63) / 64) * ((fh + 3) / 4);
  Y.groups_key_full = ((frame->width + 63) / 64) * ((frame->height + 3) / 4);
  Y.groups_key_half = ((kw + 63) / 64) * ((kh + 3) / 4);
  bool rode = false;
#if !VK_TP_KEY_SIDE
  // experiment: the frame side only; the raycast's normal image and half level stay with the next Track's pyramid launch
  Y.groups_key_full = Y.groups_key_half = 0;
  const int rc = trace_ahead(v, frame, ahead, out_depth, out_color, nullptr, stream, nullptr, nullptr, 0, &Y, &rode);
  built->pad_ = 1;
#else
  const int rc = trace_ahead(v, frame, ahead, out_depth, out_color, out_normals, stream, nullptr, nullptr, 0, &Y, &rode);
  built->pad_ = 0;
#endif
  if (rc != VK_OK) return rc;
  if (!rode)
  {
    // not ridden (vk_trace_ahead's launches; the record stays invalid) — the experiment's form owes the normal image then
    if (!VK_TP_KEY_SIDE) return launch_normals(out_depth, &frame->depth_projection, out_normals, frame->width, frame->height, vk_s(stream));
    return VK_OK;
  }
  built->key_depths = out_depth;
  built->key_normals = out_normals;
  built->frame_depths = next->depths;
  built->frame_normals = next->normals;
  built->pyramid = pyramid;
  built->key_width = frame->width;
  built->key_height = frame->height;
  built->frame_width = next->width;
  built->frame_height = next->height;
  built->valid = 1;
  return VK_OK;
}

int vk_trace_normals_settle(vk_view_bounds* ahead, void* stream)
{
  VK_REQUIRE(ahead && ahead->scratch && ahead->bounds_width > 0 && ahead->bounds_height > 0);
  hipStream_t s = vk_s(stream);
  VK_CHECK(hipStreamSynchronize(s));
  bool late = false;
  if (ahead->late_host) late = __atomic_load_n(ahead->late_host, __ATOMIC_RELAXED) != 0;
  else if (ahead->counted_scratch == ahead->scratch && normals_counters(ahead))
  {
    uint32_t word = 0;
    VK_CHECK(hipMemcpy(&word, normals_counters(ahead) + kNormalRows * kNormalRowStride, sizeof(word), hipMemcpyDeviceToHost));
    late = word != 0;
  }
  return late ? normals_repair(ahead, s) : VK_OK;
}

}  // extern "C"
