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

// The same merge for a wave-uniform cell, read through the scalar cache: the
// partial grids were written by the previous kernel and are read-only here, so
// they can be addressed as constant memory (s_load instead of 64-lane loads).
typedef const float __attribute__((address_space(4)))* scalar_floats;

__device__ __forceinline__ float2 merged_bound_uniform(const float2* partials, int cells, int cell)
{
  const scalar_floats base = (scalar_floats)reinterpret_cast<const float*>(partials);
  float2 b = make_float2(base[2 * cell + 0], base[2 * cell + 1]);
#pragma unroll
  for (int g = 1; g < kBoundsGroups; ++g)
  {
    const size_t at = 2 * ((size_t)g * cells + cell);
    b.x = vmin(base[at + 0], b.x);
    b.y = vmax(base[at + 1], b.y);
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

struct PointParams
{
  const vk_hash_entry* entries;
  const vk_voxel* voxels;
  const float* bounds;        // merged grid (read when partials == nullptr)
  const float2* partials;     // kBoundsGroups private grids (fused path), or nullptr
  float2* bounds_out;         // fused path: the merged grid is written back here
  uint32_t K;
  float block_length, voxel_length, trunc_length;
  Rt Twc, Tcw;
  vk_projection k;
  float* depths;
  float* colors;
  int image_width, image_height, bounds_width, bounds_height;
};

// A voxel is 5 dwords {distance, r, g, b, (cw << 16 | dw)} at a 4-byte aligned
// address; gfx950 global loads only need dword alignment, so a voxel is read as
// dwordx4 + dword, and the x-adjacent pair of a trilinear row as 10 contiguous
// dwords, instead of the reference's five scalar loads per voxel.

struct Corner
{
  float distance;
  float r, g, b;
  int color_weight;
};

__device__ __forceinline__ Corner empty_corner()   // Voxel::Empty(), voxel.h:31-39
{
  Corner c;
  c.distance = 1.0f;
  c.r = c.g = c.b = 0.0f;
  c.color_weight = 0;
  return c;
}

__device__ __forceinline__ int color_weight_of(float packed)
{
  return (int)(int16_t)(__float_as_uint(packed) >> 16);
}

// A voxel whose colour weight is 0 has never had its colour written (every
// colour update increments the weight), so its colour is the initial (0,0,0):
// the 12 colour bytes are only fetched when the weight is positive.

// Address of voxel `voxel` (0..511) of pool slot `slot`; 64-bit throughout, a pool
// may hold tens of millions of blocks (288 GB of HBM). Measured: 32-bit offsets for
// pools under 4 GiB change nothing (the kernel is not bound by address arithmetic).
__device__ __forceinline__ const float* voxel_address(const float* pool, int slot, int voxel)
{
  return pool + ((size_t)slot * VK_BLOCK_VOXELS + (size_t)voxel) * 5;
}

// One-entry cache of the last hash lookup: consecutive march steps and the
// eight trilinear corners mostly stay in one block, and the table is read-only
// during the kernel, so the cached answer is the answer a fresh walk would give.
struct BlockCache
{
  int bx, by, bz;
  int data;    // pool slot of the block, -1 when absent / unallocated
  bool valid;
};

// Per-wave block directory in LDS: 64 direct-mapped entries {bx, by, bz, slot}.
// The reference walks the global hash table once per ray per step (and once per
// trilinear corner near block faces). The 64 rays of a wave cross the same
// handful of blocks over and over, so here a block is resolved against the
// global table about once per wave and every later use, by any lane, is one LDS read. The
// table is read-only during the kernel, so a cached answer is the answer a fresh
// walk would give.
constexpr int kDirEntries = 64;
constexpr int kMaxChain = 1 << 24;

// Entry = the block's coordinates modulo 4 per axis: any 4x4x4 neighbourhood of
// blocks (16 cm at 5 mm voxels, far more than one wave's 8x8 pixels see in a
// step) maps to 64 different entries, so the blocks a wave works on never evict
// one another.
__device__ __forceinline__ int dir_index(int bx, int by, int bz)
{
  return (bx & 3) | ((by & 3) << 2) | ((bz & 3) << 4);
}

// tracer.cu:364-371: walk the chain until the block matches or the chain ends;
// a hit needs the match AND IsAllocated().
__device__ __forceinline__ int probe_table(const PointParams& P, int bx, int by, int bz)
{
  Entry entry = load_entry(P.entries, block_hash(bx, by, bz, P.K));
  // a chain is at most the excess region long; the cap only guarantees that a wave
  // leaves the loop if it is handed a corrupt table (a cycle would otherwise hang the GPU)
  for (int guard = 0; !entry_is(entry, bx, by, bz) && entry.next != -1 && guard < kMaxChain; ++guard)
    entry = load_entry(P.entries, (uint32_t)entry.next);
  return (entry_is(entry, bx, by, bz) && entry.data != -1) ? entry.data : -1;
}

// files a resolved block in the directory: one distinct block per trip, written
// by a single lane so that an entry is never a mix of two lanes' stores
__device__ __forceinline__ void file_blocks(int4* dir, bool pending, int bx, int by, int bz, int data)
{
  while (__any(pending))
  {
    const unsigned long long mask = __ballot(pending);
    const int leader = __ffsll((long long)mask) - 1;
    const int ubx = __builtin_amdgcn_readlane(bx, leader);
    const int uby = __builtin_amdgcn_readlane(by, leader);
    const int ubz = __builtin_amdgcn_readlane(bz, leader);
    const int udata = __builtin_amdgcn_readlane(data, leader);
    if (lane_id() == leader) dir[dir_index(ubx, uby, ubz)] = make_int4(ubx, uby, ubz, udata);
    if (bx == ubx && by == uby && bz == ubz) pending = false;
  }
  wave_lds_fence();   // later reads of the directory, by any lane, see these entries
}

__device__ __forceinline__ int find_block(const PointParams& P, BlockCache& cache, int4* dir, int bx, int by, int bz)
{
  if (cache.valid && cache.bx == bx && cache.by == by && cache.bz == bz) return cache.data;

  const int4 e = dir[dir_index(bx, by, bz)];
  int data = e.w;
  const bool missed = !(e.x == bx && e.y == by && e.z == bz);

  // directory misses probe the global table, all lanes in parallel, then file
  // their answers
  if (missed) data = probe_table(P, bx, by, bz);
  file_blocks(dir, missed, bx, by, bz, data);

  cache.bx = bx; cache.by = by; cache.bz = bz; cache.data = data; cache.valid = true;
  return data;
}

// tracer.cu:114-188 GetVoxel: a corner index outside [0,8) moves one block over
// along that axis (a single wrap, as in the reference)
__device__ __forceinline__ void wrap_axis(int v, int& local, int& shift)
{
  shift = (v < 0) ? -1 : ((v >= VK_BLOCK_RESOLUTION) ? 1 : 0);
  local = v - VK_BLOCK_RESOLUTION * shift;
}

// tracer.cu:190-315 GetInterpolatedDistance -> (sdf, colour)
// (wx, wy, wz): the sample position in voxel units relative to block (bx, by, bz),
// i.e. (p - b * block_length) / voxel_length as computed by the caller
__device__ __forceinline__ void interpolate(const PointParams& P, BlockCache& cache, int4* dir, int bx, int by,
    int bz, int data, float wx, float wy, float wz, float& sdf, f3& color)
{
  const int i0x = f2i(floorf(wx - 0.5f));
  const int i0y = f2i(floorf(wy - 0.5f));
  const int i0z = f2i(floorf(wz - 0.5f));

  Corner vv[8];  // index dz*4 + dy*2 + dx
  const float* voxf = reinterpret_cast<const float*>(P.voxels);

  // The eight corners touch at most 2x2x2 blocks. The reference has two code
  // paths (all corners in this block: eight direct reads, tracer.cu:219-243;
  // otherwise GetVoxel x8, each walking the hash table, :244-256). Here there is
  // ONE path for every lane of the wave: per axis, sx/sy/sz say which way each of
  // the two corners leaves the block (0 = stays; always 0 for interior samples)
  // and lx/ly/lz are the wrapped voxel indices; a block is looked up only if it
  // differs from this one AND from the block of a corner already resolved, so an
  // interior sample costs no lookup and a face-crossing sample costs one.
  int lx[2], ly[2], lz[2], sx[2], sy[2], sz[2];
  wrap_axis(i0x, lx[0], sx[0]); wrap_axis(i0x + 1, lx[1], sx[1]);
  wrap_axis(i0y, ly[0], sy[0]); wrap_axis(i0y + 1, ly[1], sy[1]);
  wrap_axis(i0z, lz[0], sz[0]); wrap_axis(i0z + 1, lz[1], sz[1]);

  int slot[8];  // pool slot of the block holding corner c, -1 = absent
#pragma unroll
  for (int c = 0; c < 8; ++c)
  {
    const int dx = c & 1, dy = (c >> 1) & 1, dz = (c >> 2) & 1;
    if (dx && sx[1] == sx[0]) { slot[c] = slot[c ^ 1]; continue; }
    if (dy && sy[1] == sy[0]) { slot[c] = slot[c ^ 2]; continue; }
    if (dz && sz[1] == sz[0]) { slot[c] = slot[c ^ 4]; continue; }
    if ((sx[dx] | sy[dy] | sz[dz]) == 0) { slot[c] = data; continue; }
    BlockCache scratch;
    scratch.valid = false;
    slot[c] = find_block(P, scratch, dir, bx + sx[dx], by + sy[dy], bz + sz[dz]);
  }

  // Four rows of two x-neighbours. When both voxels of a row sit in the same
  // block they are 10 contiguous dwords {d0 rgb0 w0 | d1 rgb1 w1}: d0, (w0,d1), w1
  // = 3 loads for the row. Only lanes whose sample straddles a block face in x
  // (1 in 8) fetch d1 separately (4th, exec-masked load).
#pragma unroll
  for (int row = 0; row < 4; ++row)
  {
    const int dy = row & 1, dz = row >> 1;
    const int c0 = dz * 4 + dy * 2, c1 = c0 + 1;
    const int row_voxel = lz[dz] * 64 + ly[dy] * 8;
    const bool has0 = slot[c0] >= 0, has1 = slot[c1] >= 0;
    const bool split = sx[0] != sx[1];   // x-neighbours in different blocks
    // absent blocks read voxel 0 of the pool and are overridden with Voxel::Empty() below
    const float* a0 = voxel_address(voxf, has0 ? slot[c0] : 0, has0 ? row_voxel + lx[0] : 0);
    const float* a1 = split ? voxel_address(voxf, has1 ? slot[c1] : 0, has1 ? row_voxel + lx[1] : 0) : a0 + 5;

    const float d0 = a0[0];
    // (w0, d1) when contiguous; a split lane reads (b0, w0) instead so the 8-byte
    // load never leaves its own voxel (the pool may end right after it)
    const vf2 mid = *reinterpret_cast<const vf2*>(a0 + (split ? 3 : 4));
    const float w0 = split ? mid.y : mid.x;
    const float w1 = a1[4];
    float d1 = mid.y;
    if (split) d1 = a1[0];

    vv[c0] = empty_corner();
    vv[c1] = empty_corner();
    if (has0)
    {
      vv[c0].distance = d0;
      vv[c0].color_weight = color_weight_of(w0);
      if (vv[c0].color_weight > 0)
      {
        const vf3 rgb = *reinterpret_cast<const vf3*>(a0 + 1);
        vv[c0].r = rgb.x; vv[c0].g = rgb.y; vv[c0].b = rgb.z;
      }
    }
    if (has1)
    {
      vv[c1].distance = d1;
      vv[c1].color_weight = color_weight_of(w1);
      if (vv[c1].color_weight > 0)
      {
        const vf3 rgb = *reinterpret_cast<const vf3*>(a1 + 1);
        vv[c1].r = rgb.x; vv[c1].g = rgb.y; vv[c1].b = rgb.z;
      }
    }
  }

  const float w1x = wx - (i0x + 0.5f), w1y = wy - (i0y + 0.5f), w1z = wz - (i0z + 0.5f);
  const float w0x = 1.0f - w1x, w0y = 1.0f - w1y, w0z = 1.0f - w1z;

  const float n00 = vv[0].distance * w0x + vv[1].distance * w1x;
  const float n01 = vv[2].distance * w0x + vv[3].distance * w1x;
  const float n10 = vv[4].distance * w0x + vv[5].distance * w1x;
  const float n11 = vv[6].distance * w0x + vv[7].distance * w1x;
  const float n0 = n00 * w0y + n01 * w1y;
  const float n1 = n10 * w0y + n11 * w1y;

  // tracer.cu:282-289: `a*b*c*(cw>0) ? 1 : 0` == `(a*b*c*(cw>0)) ? 1 : 0`, i.e. the
  // colour is the plain mean of the corners that carry colour. When no corner of
  // any lane in the wave has a colour weight (depth-only volumes) every weight is
  // 0 and the result is (0,0,0): the whole block is skipped wave-uniformly.
  float total = 0.0f;
  f3 acc = make3(0.0f, 0.0f, 0.0f);
  int any_weight = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) any_weight |= (vv[c].color_weight > 0) ? 1 : 0;

  if (__any(any_weight))
  {
    float cwt[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
    {
      const float fz = ((c >> 2) & 1) ? w1z : w0z;
      const float fy = ((c >> 1) & 1) ? w1y : w0y;
      const float fx = (c & 1) ? w1x : w0x;
      const float prod = fz * fy * fx * (float)(vv[c].color_weight > 0 ? 1 : 0);
      cwt[c] = (prod != 0.0f) ? 1.0f : 0.0f;   // NaN counts as true, as in C
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) total += cwt[c];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc = add3(acc, scale3(make3(vv[c].r, vv[c].g, vv[c].b), cwt[c]));
    if (total > 0) acc = div3(acc, total);
  }

  sdf = n0 * w0z + n1 * w1z;
  color = acc;
}

// ref: tracer.cu:317-451. One lane per pixel; a wave covers an 8x8 pixel tile
// (one bounds cell at 640x480 / 80x60) so its rays start at the same depth, run
// a similar number of steps and walk the same few blocks; a workgroup is 2x2
// such tiles.
__global__ __launch_bounds__(256) void compute_points_kernel(PointParams P)
{
  __shared__ int4 directories[4][kDirEntries];

  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  int4* bdir = directories[wave];
  bdir[lane] = make_int4(INT32_MIN, INT32_MIN, INT32_MIN, -1);   // no block has these coordinates after f2i
  wave_lds_fence();

  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, each
  // with its own L2, and neighbouring 16x16 tiles march through the same voxel
  // blocks. Workgroup w therefore takes tile (w % 8) * chunk + w / 8, which gives
  // every XCD one contiguous band of the image (placement only affects speed).
  const int tiles_x = (P.image_width + 15) / 16, tiles_y = (P.image_height + 15) / 16;
  const int tiles = tiles_x * tiles_y;
  const int chunk = (tiles + 7) / 8;
  const int tile = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  const int tile_x = tile % tiles_x, tile_y = tile / tiles_x;
  const int x = tile_x * 16 + (wave & 1) * 8 + (lane & 7);
  const int y = tile_y * 16 + (wave >> 1) * 8 + (lane >> 3);

  // This wave's bound first, while no store has been issued yet: with a
  // wave-uniform cell (always, when a bounds cell is 8x8 pixels) the kBoundsGroups
  // partial grids are then read through the scalar cache instead of by 64 lanes.
  const bool inside = tile < tiles && x < P.image_width && y < P.image_height;
  const int px = P.bounds_width * vmini(x, P.image_width - 1) / P.image_width;
  const int py = P.bounds_height * vmini(y, P.image_height - 1) / P.image_height;
  const int cell = (tile < tiles) ? py * P.bounds_width + px : 0;
  float2 bound;

  if (P.partials)
  {
    const int cells = P.bounds_width * P.bounds_height;
    const int first = __builtin_amdgcn_readfirstlane(cell);
    if (__all(cell == first)) bound = merged_bound_uniform(P.partials, cells, first);
    else bound = merged_bound(P.partials, cells, cell);

    // publish the merged grid (Tracer::bounds_) — every cell, whether or not a
    // pixel maps to it
    const int threads = gridDim.x * 256;
    for (int c = blockIdx.x * 256 + threadIdx.x; c < cells; c += threads)
      P.bounds_out[c] = merged_bound(P.partials, cells, c);
  }
  else
  {
    bound = reinterpret_cast<const float2*>(P.bounds)[cell];
  }

  if (!inside) return;

  float final_depth = 0;
  f3 color = make3(0, 0, 0);

  if (bound.x < bound.y)
  {
    const f3 Xcp = unproject_d(P.k, x + 0.5f, y + 0.5f, bound.x);
    const f3 Xwp = xform_point(P.Twc, Xcp);
    const f3 dir = normalized3(xform_dir(P.Twc, Xcp));

    f3 p = Xwp;
    int iters = 0;
    BlockCache cache;
    cache.valid = false;
    cache.bx = cache.by = cache.bz = 0;
    cache.data = -1;

    // The reference's loop body (tracer.cu:358-444) contains a second, nested
    // lookup + interpolation for the step that follows the first sample behind
    // the surface (:395-423). Here that step is one more trip through the same
    // loop body with `refine` set, so the lanes of a wave share ONE lookup site
    // and ONE interpolation site whatever phase each ray is in.
    bool refine = false;

    for (;;)
    {
      const int bx = f2i(floorf(p.x / P.block_length));
      const int by = f2i(floorf(p.y / P.block_length));
      const int bz = f2i(floorf(p.z / P.block_length));
      const int data = find_block(P, cache, bdir, bx, by, bz);
      bool done = false;

      if (data >= 0)
      {
        float sdf;
        bool sample = refine;

        // position in voxel units inside the block: tracer.cu:373-375 for the
        // nearest-voxel read and, with the same expression, :193-195 for the sample
        const float wx = (p.x - bx * P.block_length) / P.voxel_length;
        const float wy = (p.y - by * P.block_length) / P.voxel_length;
        const float wz = (p.z - bz * P.block_length) / P.voxel_length;

        if (!refine)
        {
          // int(w) can reach 8 on a block face: clamped, see DESIGN.md
          const int vx = vmini(f2i(wx), 7);
          const int vy = vmini(f2i(wy), 7);
          const int vz = vmini(f2i(wz), 7);

          sdf = voxel_address(reinterpret_cast<const float*>(P.voxels), data, vz * 64 + vy * 8 + vx)[0];
          sample = (sdf <= 0.1f && sdf >= -0.5f);
        }

        if (sample) interpolate(P, cache, bdir, bx, by, bz, data, wx, wy, wz, sdf, color);

        if (refine)
        {
          p = add3(p, scale3(dir, P.trunc_length * sdf));       // :417
          done = true;
        }
        else if (sdf <= 0.0f)
        {
          p = add3(p, scale3(dir, P.trunc_length * sdf));       // :397
          refine = true;
          continue;                                             // :399-418 happen next trip
        }
        else
        {
          p = add3(p, scale3(dir, vmax(P.voxel_length, P.trunc_length * sdf)));
        }
      }
      else if (refine)
      {
        done = true;                                            // :410 false: no second sample
      }
      else
      {
        p = add3(p, scale3(dir, P.block_length));
      }

      const float depth = xform_point(P.Tcw, p).z;

      if (done)
      {
        final_depth = depth;                                    // :420-422
        break;
      }

      if (++iters >= 500)
      {
        color = make3(1, 0, 0);
        break;
      }

      if (!(depth < bound.y)) break;
    }
  }

  const int pixel = y * P.image_width + x;
  P.depths[pixel] = final_depth;
  P.colors[3 * pixel + 0] = color.x;
  P.colors[3 * pixel + 1] = color.y;
  P.colors[3 * pixel + 2] = color.z;
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
  {
    const f3 z0 = unproject_d(k, (x + 0) + 0.5f, (y + 0) + 0.5f, depth);
    float d;

    d = depth_at(depths, image_width, image_height, x - pad, y);
    const f3 x0 = (d == 0) ? z0 : scale3(unproject(k, (x - pad) + 0.5f, (y + 0) + 0.5f), d);
    d = depth_at(depths, image_width, image_height, x + pad, y);
    const f3 x1 = (d == 0) ? z0 : scale3(unproject(k, (x + pad) + 0.5f, (y + 0) + 0.5f), d);
    d = depth_at(depths, image_width, image_height, x, y - pad);
    const f3 y0 = (d == 0) ? z0 : scale3(unproject(k, (x + 0) + 0.5f, (y - pad) + 0.5f), d);
    d = depth_at(depths, image_width, image_height, x, y + pad);
    const f3 y1 = (d == 0) ? z0 : scale3(unproject(k, (x + 0) + 0.5f, (y + pad) + 0.5f), d);

    const f3 dx = sub3(x0, x1);
    const f3 dy = sub3(y0, y1);

    if (sqnorm3(dx) > 0 && sqnorm3(dy) > 0) normal = normalized3(cross3(dy, dx));
  }

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

int launch_points(const vk_hash_entry* entries, const vk_voxel* voxels, const float* bounds,
    const float2* partials, int block_count, float block_length, float voxel_length, float trunc_length,
    const vk_transform* Twc, const vk_projection* projection, float* depths, float* colors,
    int image_width, int image_height, int bounds_width, int bounds_height, hipStream_t s)
{
  PointParams P;
  P.entries = entries;
  P.voxels = voxels;
  P.bounds = bounds;
  P.partials = partials;
  P.bounds_out = reinterpret_cast<float2*>(const_cast<float*>(bounds));
  P.K = (uint32_t)block_count;
  P.block_length = block_length;
  P.voxel_length = voxel_length;
  P.trunc_length = trunc_length;
  P.Twc = make_rt(Twc->m);
  P.Tcw = make_rt(Twc->inv);  // tracer.cu:350 Twc.Inverse()
  P.k = *projection;
  P.depths = depths;
  P.colors = colors;
  P.image_width = image_width;
  P.image_height = image_height;
  P.bounds_width = bounds_width;
  P.bounds_height = bounds_height;
  const int tiles = ((image_width + 15) / 16) * ((image_height + 15) / 16);
  const dim3 grid(8 * ((tiles + 7) / 8));   // padded so every XCD gets an equal band
  hipLaunchKernelGGL(compute_points_kernel, grid, dim3(256), 0, s, P);
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
      Twc, projection, depths, colors, image_width, image_height, bounds_width, bounds_height, vk_s(stream));
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
  return 2 * cells * (cells <= (size_t)kBoundsMaxCells ? 1 + kBoundsGroups : 1);
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
           out_depth, out_color, frame->width, frame->height, bounds_width, bounds_height, s)) != VK_OK)
    return rc;

  // tracer.cpp:97-100 ComputeNormals
  return launch_normals(out_depth, &frame->depth_projection, out_normals, frame->width, frame->height, s);
}

int vk_trace_ahead(const vk_volume* v, const vk_frame* frame, vk_view_bounds* ahead, float* out_depth,
    float* out_color, float* out_normals, void* stream)
{
  VK_REQUIRE(v && frame && ahead && ahead->scratch && out_depth && out_color && out_normals);
  VK_REQUIRE(v->hash_entries && v->voxels && v->visible_blocks && v->counters);
  VK_REQUIRE(ahead->bounds_width > 0 && ahead->bounds_height > 0);
  hipStream_t s = vk_s(stream);
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

  if ((rc = launch_points(v->hash_entries, v->voxels, bounds, partials, v->main_block_count, block_length,
           v->voxel_length, v->truncation_length, &frame->depth_to_world, &frame->depth_projection,
           out_depth, out_color, frame->width, frame->height, ahead->bounds_width, ahead->bounds_height, s)) != VK_OK)
    return rc;
  return launch_normals(out_depth, &frame->depth_projection, out_normals, frame->width, frame->height, s);
}

}  // extern "C"
