// vk_raycast.hpp — the per-pixel ray march through the hashed volume (ref:
// src/tracer.cu:114-451 GetVoxel, GetInterpolatedDistance, ComputePointsKernel).
//
// One lane per pixel, one wave per 8x8 pixel tile. What differs from the reference's
// thread-per-pixel kernel, with identical results:
//  * a block is resolved against the global hash table about once per WAVE (per-wave
//    block directory in LDS) instead of once per ray per step per corner;
//  * the march only needs distances: a trilinear sample reads the eight corner
//    distances and nothing else. The reference also assembles a colour at every
//    sample and overwrites it at the next one (tracer.cu:392-393,412-413): only the
//    LAST sample's colour reaches the image, so weights and colours are fetched once
//    per ray, after the march, at that sample's position;
//  * the eight corners are addressed from the block that holds the low corner, so a
//    sample only ever leaves its block in the +x/+y/+z direction and corner c lives in
//    neighbour (c & crossing mask) — no per-corner case analysis;
//  * pools under 4 GiB (POOL32) address voxels by 32-bit offsets from a scalar base;
//  * divisions by the two launch constants (block and voxel length) are exact
//    without the division expansion (div_uniform below).
// COUNT = true (tools/probe only) additionally marks every pool slot a ray reads.
#pragma once

#ifndef VK_MARCH_AHEAD
#define VK_MARCH_AHEAD 0           // 4: the gated run-ahead through absent blocks (an experiment of round 6; see march_ray_nested)
#endif
#ifndef VK_MARCH_AHEAD_AFTER
#define VK_MARCH_AHEAD_AFTER 12    // a wave takes the run-ahead path only past this many passes through its march loop
#endif

#include "vk_common.hpp"


namespace vk
{

struct PointParams
{
  const vk_hash_entry* entries;
  const vk_voxel* voxels;
  const float* bounds;        // merged grid (read when partials == nullptr)
  const float2* partials;     // kBoundsGroups private grids (fused path), or nullptr
  float2* bounds_out;         // fused path: the merged grid is written back here
  uint32_t K;
  float block_length, voxel_length, trunc_length;
  double inv_block_length, inv_voxel_length;   // RN64(1 / length), see div_uniform
  Rt Twc, Tcw;
  Projection k;          // with 1 / fx, 1 / fy (vk_common.hpp)
  float* depths;
  float* colors;
  int image_width, image_height, bounds_width, bounds_height;
  uint8_t* touched;           // COUNT only: one byte per pool slot
  int* march_steps;           // COUNT only (optional): trips through the march loop, per pixel
  // COUNT only (optional): one word per wave and pass through the march loop, trip_log[wave * trip_log_passes + pass] =
  // wall clock (32 bits, 10 ns) << 32 | lanes still marching << 24 | of those: whose last trip found no block << 16 |
  // whose last trip took a sample << 8
  unsigned long long* trip_log;
  int trip_log_passes;
  // The normals of the raycast by trailing workgroups of the same launch (vk_trace.hip normals_group): one counter per
  // 8-pixel row of tiles, bumped by every wave that has written its tile's depths; nullptr = nobody waits for them.
  uint32_t* rows_done;
  uint32_t rows_target;        // a row is complete once (int)(rows_done[row] - rows_target) >= 0
  // the outcome of that wait when it expires (vk.h vk_view_bounds.late_host): a word behind the counters, and the
  // caller's pinned word or nullptr; normal_polls: how often a group looks before it gives up
  uint32_t* late_dev;
  int32_t* late_host;
  uint32_t late_tag;           // what an expired group stores there: the record's launch number (never 0)
  int normal_polls;
};

// a / b, correctly rounded, for a divisor known on the host: inv_b = RN64(1 / b).
// (double)a * inv_b is within 2^-52 (relative) of a / b; the quotient of two binary32
// numbers is never closer than 2^-49 (relative) to the midpoint of two neighbouring
// binary32 numbers [for 24-bit significands A, B and an odd 25-bit N, |A 2^j - B N| >= 1],
// so rounding the double product to binary32 gives exactly RN32(a / b) — the value the
// reference's `a / b` has. Three instructions instead of the ten of v_div_scale /
// v_rcp / v_fma x4 / v_div_fmas / v_div_fixup.
__device__ __forceinline__ float div_uniform(float a, double inv_b)
{
  return (float)((double)a * inv_b);
}

// One-entry cache of the last hash lookup: consecutive march steps mostly stay in one
// block, and the table is read-only during the kernel, so the cached answer is the
// answer a fresh walk would give.
struct BlockCache
{
  int bx, by, bz;
  int data;    // pool slot of the block, -1 when absent / unallocated
  bool valid;
};

// Per-wave block directory in LDS: 64 direct-mapped entries {bx, by, bz, slot}. The
// reference walks the global hash table once per ray per step (and once per trilinear
// corner near block faces). The 64 rays of a wave cross the same handful of blocks over
// and over, so here a block is resolved against the global table about once per wave
// and every later use, by any lane, is one LDS read.
constexpr int kDirEntries = 64;
constexpr int kDirWords = kDirEntries + 16;   // in int4: the entries, then one tag word per entry (file_blocks)
constexpr int kMaxChain = 1 << 24;

// Entry = the block's coordinates modulo 4 per axis: any 4x4x4 neighbourhood of blocks
// (16 cm at 5 mm voxels, far more than one wave's 8x8 pixels see in a step) maps to 64
// different entries, so the blocks a wave works on never evict one another.
__device__ __forceinline__ int dir_index(int bx, int by, int bz)
{
  return (bx & 3) | ((by & 3) << 2) | ((bz & 3) << 4);
}

// tracer.cu:364-371: walk the chain until the block matches or the chain ends; a hit
// needs the match AND IsAllocated().
// One 16-byte load per entry, all of it at once: left alone, the compiler reads the coordinates and `next` inside the walk
// and comes back for `data` once the block has matched — a second, dependent read on every lookup (the ISA of round 3).
__device__ __forceinline__ Entry load_whole_entry(const vk_hash_entry* entries, uint32_t index)
{
  int4 raw = reinterpret_cast<const int4*>(entries)[index];
  asm volatile("" : "+v"(raw.x), "+v"(raw.y), "+v"(raw.z), "+v"(raw.w));
  Entry e;
  e.ox = (int16_t)(raw.x & 0xffff);
  e.oy = (int16_t)((uint32_t)raw.x >> 16);
  e.oz = (int16_t)(raw.y & 0xffff);
  e.pad = (int16_t)((uint32_t)raw.y >> 16);
  e.data = raw.z;
  e.next = raw.w;
  return e;
}

__device__ __forceinline__ int probe_table(const PointParams& P, int bx, int by, int bz)
{
  Entry entry = load_whole_entry(P.entries, block_hash(bx, by, bz, P.K));
  // a chain is at most the excess region long; the cap only guarantees that a wave
  // leaves the loop if it is handed a corrupt table (a cycle would otherwise hang the GPU)
  for (int guard = 0; !entry_is(entry, bx, by, bz) && entry.next != -1 && guard < kMaxChain; ++guard)
    entry = load_whole_entry(P.entries, (uint32_t)entry.next);
  return (entry_is(entry, bx, by, bz) && entry.data != -1) ? entry.data : -1;
}

// files resolved blocks in the directory, every lane its own at once. Two lanes may want the same entry (the same block, or
// two blocks that share it): they first write their lane number into the entry's TAG word, read it back — a wave's LDS
// operations execute in order, so all of them read the one that stayed — and only that lane writes the entry: an entry is
// never a mix of two lanes' stores, and a block that lost is simply looked up again when it is next needed.
// (Until round 4 the lanes filed one DISTINCT block per trip through a ballot / readlane loop, ~15 instructions per block:
// a wave that meets a surface files 20 to 60 blocks in a pass, 1 to 2 us of its 4 to 6 — tools/trip_log.py. r03 on that
// loop, tracking scene / fusion benchmark: a limit of N blocks per call, the allocated ones first — 8: 92.1 / 31.3 us,
// 4: 90.9 / 31.4, 2: 87.6 / 32.2, 1: 83.5 / 33.2 against 90.9 / 31.5 without a limit; not filing the absent blocks the
// MARCH runs through, find_block: 77.8 / 30.9.)
#ifndef VK_DIR_SERIAL_FILING
#define VK_DIR_SERIAL_FILING 0
#endif
__device__ __forceinline__ void file_blocks(int4* dir, bool pending, int bx, int by, int bz, int data)
{
  if (VK_DIR_SERIAL_FILING)
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
  }
  else if (__any(pending))
  {
    int* tags = reinterpret_cast<int*>(dir + kDirEntries);
    const int entry = dir_index(bx, by, bz);
    if (pending) tags[entry] = lane_id();
    wave_lds_fence();
    if (pending && tags[entry] == lane_id()) dir[entry] = make_int4(bx, by, bz, data);
  }
  wave_lds_fence();   // later reads of the directory, by any lane, see these entries
}

// slot of block (bx, by, bz) for the lanes with `active` set; -1 = absent
// `file_absent`: whether a block that turns out not to be allocated is filed as well (a later
// request for it is then answered from the directory)
__device__ __forceinline__ int lookup_block(const PointParams& P, int4* dir, bool active, int bx, int by, int bz,
    bool file_absent = true)
{
  const int4 e = dir[dir_index(bx, by, bz)];
  int data = e.w;
  const bool missed = active && !(e.x == bx && e.y == by && e.z == bz);
  if (__any(missed))
  {
    // directory misses probe the global table, all lanes in parallel, then file their answers
    if (missed) data = probe_table(P, bx, by, bz);
    file_blocks(dir, missed && (file_absent || data >= 0), bx, by, bz, data);
  }
  return data;
}

__device__ __forceinline__ int find_block(const PointParams& P, BlockCache& cache, int4* dir, int bx, int by, int bz)
{
  if (cache.valid && cache.bx == bx && cache.by == by && cache.bz == bz) return cache.data;
  // The block a ray is IN, when it is not allocated, is not filed: filing is serial in the number of
  // distinct blocks, the ray leaves an absent block with its next trip, and rays that run through
  // empty space (past an object's silhouette towards a far wall: dozens of trips, every lane its own
  // block) would file a dozen blocks per trip that nobody asks for again. rocprofv3 averages, r03,
  // tracking scene / fusion benchmark: 90.9 / 31.5 us with those filed, 77.8 / 30.9 without.
  const int data = lookup_block(P, dir, true, bx, by, bz, false);
  cache.bx = bx; cache.by = by; cache.bz = bz; cache.data = data; cache.valid = true;
  return data;
}

// explicitly global pointers: the corner loads must not become flat loads
typedef const float __attribute__((address_space(1)))* global_floats;
typedef const vf3 __attribute__((address_space(1)))* global_vf3;
typedef float vf4u __attribute__((ext_vector_type(4), aligned(4)));
typedef const vf4u __attribute__((address_space(1)))* global_vf4u;

// The eight voxels around a sample (tracer.cu:190-256), index dz*4 + dy*2 + dx: where
// each one is, and the fractional position between them. POOL32: the pool is smaller
// than 4 GiB, so a voxel is a 32-bit byte offset from the (scalar) pool base — one
// 24-bit multiply-add per corner instead of a 64-bit one and a 64-bit add.
template <bool POOL32>
struct Corners
{
  uint32_t offset[8];      // POOL32: byte offset of the voxel in the pool
  global_floats voxel[8];  // !POOL32: its address
  uint32_t absent;         // bit c: corner c's block is not allocated -> Voxel::Empty()
  float fx, fy, fz;        // w1 of tracer.cu:266-272

  __device__ __forceinline__ float word(const PointParams& P, int c, int dword) const
  {
    if (POOL32)
      return *(global_floats)((const char __attribute__((address_space(1)))*)(global_floats)reinterpret_cast<const float*>(P.voxels) +
                              offset[c] + 4 * dword);
    return voxel[c][dword];
  }
  // colour and weights of corner c: the voxel's bytes 4..19 as ONE 16-byte load (4-byte aligned)
  __device__ __forceinline__ vf4u tail(const PointParams& P, int c) const
  {
    if (POOL32)
      return *(global_vf4u)((const char __attribute__((address_space(1)))*)(global_floats)reinterpret_cast<const float*>(P.voxels) +
                            offset[c] + 4);
    return *(global_vf4u)(voxel[c] + 1);
  }
  __device__ __forceinline__ vf3 rgb(const PointParams& P, int c) const
  {
    if (POOL32)
      return *(global_vf3)((const char __attribute__((address_space(1)))*)(global_floats)reinterpret_cast<const float*>(P.voxels) +
                           offset[c] + 4);
    return *(global_vf3)(voxel[c] + 1);
  }
};

// ---- the directory reads of a sample (resolve_corners) ------------------------------------------------------------
typedef int v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char;

// the LDS byte address of a __shared__ object (what ds_* instructions take)
__device__ __forceinline__ uint32_t lds_address(const void* shared)
{
  return (uint32_t)(uintptr_t)(lds_char*)(char*)const_cast<void*>(shared);
}

// Eight 16-byte LDS reads that leave together and ONE wait, as a single asm statement (round 6, ADVICE r5: until then
// nine statements — eight reads and a wait — whose outstanding data hipcc's own s_waitcnt bookkeeping could not see, so
// that correctness rested on the register allocator never placing a copy or a spill of a destination between a read and
// the wait). Inside one statement nothing can be scheduled, copied or spilled between the reads and the wait; the
// destinations are early-clobber (they are written while the addresses are still being read), and the "memory" clobber
// tells the compiler that the statement reads LDS the surrounding code has written with ordinary stores (file_blocks).
// The eight addresses are the corners of a box: entry m sits at a + (m & 1 ? sx : 0) + (m & 2 ? sy : 0) + (m & 4 ? sz : 0).
// Walking them in Gray-code order (0, 1, 3, 2, 6, 7, 5, 4) takes ONE running address and one add or subtract per read, so
// the statement holds 32 + 4 registers instead of 32 + 8 (an LDS instruction reads its address register when it issues:
// the add behind it may overwrite it).
__device__ __forceinline__ void lds_read16x8(v4i& e0, v4i& e1, v4i& e2, v4i& e3, v4i& e4, v4i& e5, v4i& e6, v4i& e7,
    uint32_t a, uint32_t sx, uint32_t sy, uint32_t sz)
{
  asm volatile(
      "ds_read_b128 %0, %8\n\t"
      "v_add_u32 %8, %8, %9\n\t"
      "ds_read_b128 %1, %8\n\t"
      "v_add_u32 %8, %8, %10\n\t"
      "ds_read_b128 %3, %8\n\t"
      "v_sub_u32 %8, %8, %9\n\t"
      "ds_read_b128 %2, %8\n\t"
      "v_add_u32 %8, %8, %11\n\t"
      "ds_read_b128 %6, %8\n\t"
      "v_add_u32 %8, %8, %9\n\t"
      "ds_read_b128 %7, %8\n\t"
      "v_sub_u32 %8, %8, %10\n\t"
      "ds_read_b128 %5, %8\n\t"
      "v_sub_u32 %8, %8, %9\n\t"
      "ds_read_b128 %4, %8\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3), "=&v"(e4), "=&v"(e5), "=&v"(e6), "=&v"(e7), "+v"(a)
      : "v"(sx), "v"(sy), "v"(sz)
      : "memory");
}

// zero exactly when directory entry `e` holds block (bx, by, bz)
__device__ __forceinline__ uint32_t tag_difference(const v4i& e, int bx, int by, int bz)
{
  return (uint32_t)(e.x ^ bx) | (uint32_t)(e.y ^ by) | (uint32_t)(e.z ^ bz);
}

// (wx, wy, wz): the sample position in voxel units relative to block (bx, by, bz), i.e.
// (p - b * block_length) / voxel_length as computed by the caller (tracer.cu:193-195);
// `data`: that block's pool slot.
//
// tracer.cu:197-199: the low corner is voxel floor(w - 0.5) of the block, in [-1, 7];
// index -1 / 8 along an axis means the neighbouring block (GetVoxel, :114-188). Seen from
// the block that HOLDS the low corner (one block down along every axis whose index is
// -1), the low corner has index l0 = i0 & 7 and the high corner (l0 + 1) & 7, and the
// high corner sits in the next block up exactly when l0 == 7: corner (dx, dy, dz) is in
// neighbour (dx & cx, dy & cy, dz & cz) of that base block.
template <bool COUNT, bool POOL32>
__device__ __forceinline__ Corners<POOL32> resolve_corners(const PointParams& P, int4* dir, int bx, int by, int bz,
    int data, float wx, float wy, float wz)
{
  const int ix = f2i(floorf(wx - 0.5f));
  const int iy = f2i(floorf(wy - 0.5f));
  const int iz = f2i(floorf(wz - 0.5f));

  // base block and in-block indices (>> is an arithmetic shift: -1 >> 3 == -1)
  const int base_x = bx + (ix >> 3), base_y = by + (iy >> 3), base_z = bz + (iz >> 3);
  const int l0x = ix & 7, l0y = iy & 7, l0z = iz & 7;
  const int l1x = (ix + 1) & 7, l1y = (iy + 1) & 7, l1z = (iz + 1) & 7;
  const bool cx = l0x == 7, cy = l0y == 7, cz = l0z == 7;
  const bool moved = ((ix | iy | iz) >> 3) != 0;     // the base block is not (bx, by, bz)

  // Pool slots of the eight blocks the corners live in. Corner (dx, dy, dz) lives in block base + (dx & cx, dy & cy,
  // dz & cz): with the HIGH coordinate of an axis defined as base + (crossing ? 1 : 0), block m = (nx[m & 1], ny[m >> 1
  // & 1], nz[m >> 2]) IS corner m's block, whatever crosses — a lane that crosses nothing asks for its base block eight
  // times (one LDS broadcast each) and no selection network is needed afterwards.
  //
  // Round 5, from the ISA (profiles/r05_raycast_trip_isa.txt): until round 4 the lookups were guarded one by one
  // (`if (__any(need))`), and the compiler had turned `e.x == nx && e.y == ny && e.z == nz` into control flow — read {x,
  // slot}, wait, compare, branch, read {y, z}, wait, compare: SIXTEEN dependent LDS round trips and ~175 instructions per
  // sample, most of a lone wave's 2 us per sampled trip. Now: eight 16-byte reads leave together (lds_read16x8: the
  // compiler can neither split them nor put a wait between them), ONE wait, four instructions per block for the tag
  // test (three xor, one or3), and the rare miss — a block this wave has not met — is found by OR-ing the eight.
  const int nx[2] = {base_x, base_x + (cx ? 1 : 0)}, ny[2] = {base_y, base_y + (cy ? 1 : 0)}, nz[2] = {base_z, base_z + (cz ? 1 : 0)};
  const uint32_t dx[2] = {(uint32_t)(nx[0] & 3) << 4, (uint32_t)(nx[1] & 3) << 4};          // byte offsets into the int4 directory
  const uint32_t dy[2] = {(uint32_t)(ny[0] & 3) << 6, (uint32_t)(ny[1] & 3) << 6};
  const uint32_t dz[2] = {(uint32_t)(nz[0] & 3) << 8, (uint32_t)(nz[1] & 3) << 8};
  const uint32_t dir_lds = lds_address(dir);

  v4i e0, e1, e2, e3, e4, e5, e6, e7;
  lds_read16x8(e0, e1, e2, e3, e4, e5, e6, e7, dir_lds + (dx[0] | dy[0] | dz[0]), dx[1] - dx[0], dy[1] - dy[0], dz[1] - dz[0]);

  int n[8] = {e0.w, e1.w, e2.w, e3.w, e4.w, e5.w, e6.w, e7.w};
  // tag test: zero where the entry holds exactly this block
  uint32_t t[8];
  t[0] = tag_difference(e0, nx[0], ny[0], nz[0]);
  t[1] = tag_difference(e1, nx[1], ny[0], nz[0]);
  t[2] = tag_difference(e2, nx[0], ny[1], nz[0]);
  t[3] = tag_difference(e3, nx[1], ny[1], nz[0]);
  t[4] = tag_difference(e4, nx[0], ny[0], nz[1]);
  t[5] = tag_difference(e5, nx[1], ny[0], nz[1]);
  t[6] = tag_difference(e6, nx[0], ny[1], nz[1]);
  t[7] = tag_difference(e7, nx[1], ny[1], nz[1]);
  // the block the march stands in is known without the directory (its entry may have gone to another lane's block since)
  n[0] = moved ? n[0] : data;
  t[0] = moved ? t[0] : 0u;

  if (__any((t[0] | t[1] | t[2] | t[3] | t[4] | t[5] | t[6] | t[7]) != 0u))
  {
    // A block the wave has not met yet (rare once the march is under way): the DISTINCT missing blocks of a lane —
    // neighbour m is a block of its own only along the axes that cross — are resolved against the global table, all
    // lanes in parallel, filed, and handed on to the corners that share them.
    uint32_t missing = 0;
#pragma unroll
    for (int m = 0; m < 8; ++m)
    {
      const bool own = ((m & 1) ? cx : true) && ((m & 2) ? cy : true) && ((m & 4) ? cz : true);
      missing |= (own && t[m] != 0u) ? (1u << m) : 0u;
    }
    while (__any(missing != 0))
    {
      const bool mine = missing != 0;
      const int m = __ffs((int)missing) - 1;           // this lane's next missing neighbour
      const int qx = base_x + (m & 1), qy = base_y + ((m >> 1) & 1), qz = base_z + ((m >> 2) & 1);
      int found = -1;
      if (mine) found = probe_table(P, qx, qy, qz);
      file_blocks(dir, mine, qx, qy, qz, found);
#pragma unroll
      for (int k = 0; k < 8; ++k) n[k] = (mine && m == k) ? found : n[k];
      missing &= missing - 1;
    }
    // corner c is in block (c & crossing mask)
    const int a1 = cx ? n[1] : n[0];
    const int a2 = cy ? n[2] : n[0];
    const int n13 = cy ? n[3] : n[1];
    const int a3 = cx ? n13 : a2;
    const int z0 = cz ? n[4] : n[0], z1 = cz ? n[5] : n[1], z2 = cz ? n[6] : n[2], z3 = cz ? n[7] : n[3];
    const int a5 = cx ? z1 : z0;
    const int a6 = cy ? z2 : z0;
    const int z13 = cy ? z3 : z1;
    const int a7 = cx ? z13 : a6;
    n[1] = a1; n[2] = a2; n[3] = a3; n[4] = z0; n[5] = a5; n[6] = a6; n[7] = a7;
  }
  const int (&slot)[8] = n;

  Corners<POOL32> C;
  const global_floats pool = (global_floats)reinterpret_cast<const float*>(P.voxels);
  const int ox[2] = {l0x * 5, l1x * 5};
  const int oy[2] = {l0y * 40, l1y * 40};
  const int oz[2] = {l0z * 320, l1z * 320};
  C.absent = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c)
  {
    const int voxel = oz[c >> 2] + oy[(c >> 1) & 1] + ox[c & 1];        // in floats
    const bool absent = slot[c] < 0;
    const uint32_t s = absent ? 0u : (uint32_t)slot[c];                 // an absent block reads slot 0 and is overridden
    C.absent |= absent ? (1u << c) : 0u;
    if (POOL32) C.offset[c] = __umul24(s, (uint32_t)(VK_BLOCK_VOXELS * 20)) + (uint32_t)(4 * voxel);
    else C.voxel[c] = pool + (size_t)s * (VK_BLOCK_VOXELS * 5) + voxel;  // a pool may hold tens of millions of blocks
    if (COUNT) { if (!absent) P.touched[s] = 1; }
  }

  C.fx = wx - (ix + 0.5f);
  C.fy = wy - (iy + 0.5f);
  C.fz = wz - (iz + 0.5f);
  return C;
}

// tracer.cu:266-280,312: trilinear distance (an absent block's voxels are Voxel::Empty(): 1)
template <bool POOL32>
__device__ __forceinline__ float corner_distance(const PointParams& P, const Corners<POOL32>& C)
{
  float d[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) d[c] = C.word(P, c, 0);
#pragma unroll
  for (int c = 0; c < 8; ++c) d[c] = ((C.absent >> c) & 1u) ? 1.0f : d[c];
  const float w1x = C.fx, w1y = C.fy, w1z = C.fz;
  const float w0x = 1.0f - w1x, w0y = 1.0f - w1y, w0z = 1.0f - w1z;
  const float n00 = d[0] * w0x + d[1] * w1x;
  const float n01 = d[2] * w0x + d[3] * w1x;
  const float n10 = d[4] * w0x + d[5] * w1x;
  const float n11 = d[6] * w0x + d[7] * w1x;
  const float n0 = n00 * w0y + n01 * w1y;
  const float n1 = n10 * w0y + n11 * w1y;
  return n0 * w0z + n1 * w1z;
}

// tracer.cu:282-310: `a*b*c*(cw>0) ? 1 : 0` == `(a*b*c*(cw>0)) ? 1 : 0`, i.e. the colour
// is the plain mean of the corners that carry colour. A voxel whose colour weight is 0
// has never had its colour written (every colour update increments the weight), so its
// colour is the initial (0,0,0): the 12 colour bytes are only USED when the weight is positive.
template <bool POOL32>
__device__ __forceinline__ f3 corner_color(const PointParams& P, const Corners<POOL32>& C)
{
  // (one load per corner: until round 4 the weights were read first and the 12 colour bytes only where the weight was
  // positive — two scattered loads per corner where colour exists, and a sampled trip is bound by the number of those)
  vf4u tail[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) tail[c] = C.tail(P, c);
  int cw[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) cw[c] = (int)(int16_t)(__float_as_uint(tail[c].w) >> 16);
#pragma unroll
  for (int c = 0; c < 8; ++c) cw[c] = ((C.absent >> c) & 1u) ? 0 : cw[c];

  f3 acc = make3(0.0f, 0.0f, 0.0f);
  const float w1x = C.fx, w1y = C.fy, w1z = C.fz;
  const float w0x = 1.0f - w1x, w0y = 1.0f - w1y, w0z = 1.0f - w1z;
  float total = 0.0f;
  float cwt[8];
  vf3 rgb[8];
#pragma unroll
  for (int c = 0; c < 8; ++c)
  {
    rgb[c] = vf3{0.0f, 0.0f, 0.0f};
    if (cw[c] > 0) rgb[c] = vf3{tail[c].x, tail[c].y, tail[c].z};
  }
#pragma unroll
  for (int c = 0; c < 8; ++c)
  {
    const float fz = ((c >> 2) & 1) ? w1z : w0z;
    const float fy = ((c >> 1) & 1) ? w1y : w0y;
    const float fx = (c & 1) ? w1x : w0x;
    const float prod = fz * fy * fx * (float)(cw[c] > 0 ? 1 : 0);
    cwt[c] = (prod != 0.0f) ? 1.0f : 0.0f;   // NaN counts as true, as in C
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) total += cwt[c];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc = add3(acc, scale3(make3(rgb[c].x, rgb[c].y, rgb[c].z), cwt[c]));
  if (total > 0) acc = div3(acc, total);
  return acc;
}

// ref: tracer.cu:317-451 for the pixel (x, y) of this lane; `bound` is its cell's
// (near, far). Writes depth and colour.
template <bool COUNT, bool POOL32>
__device__ __forceinline__ void march_ray_nested(const PointParams& P, int4* bdir, int x, int y, float2 bound, int log_wave = -1)
{
  float final_depth = 0;
  f3 color = make3(0, 0, 0);

  // where the last trilinear sample was taken (its colour is the pixel's colour)
  bool sampled = false;
  int sbx = 0, sby = 0, sbz = 0, sdata = -1;
  float swx = 0, swy = 0, swz = 0;
  bool capped = false;
  int trips = 0;              // COUNT only
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

    // The reference's loop body (tracer.cu:358-444) contains a second, nested lookup +
    // interpolation for the step that follows the first sample behind the surface
    // (:395-423). Here that step is one more trip through the same loop body with
    // `refine` set, so the lanes of a wave share ONE lookup site and ONE sampling site
    // whatever phase each ray is in.
    bool refine = false;
    bool in_band = false;       // COUNT only (trip_log): the last trip took a sample
    int log_pass = 0;           // COUNT only
    int absent_run = 0;         // trips in a row through blocks that are not there (trip_log; VK_MARCH_AHEAD)
#if VK_MARCH_AHEAD > 0
    int pass = 0;
#endif

    for (;;)
    {
      const int bx = f2i(floorf(div_uniform(p.x, P.inv_block_length)));
      const int by = f2i(floorf(div_uniform(p.y, P.inv_block_length)));
      const int bz = f2i(floorf(div_uniform(p.z, P.inv_block_length)));
      if (COUNT)
      {
        if (P.trip_log && log_wave >= 0)
        {
          const unsigned long long marching = __ballot(1), running = __ballot(absent_run >= 1), sampling = __ballot(in_band);
          if (log_pass < P.trip_log_passes && lane_id() == __ffsll((long long)marching) - 1)
            P.trip_log[(size_t)log_wave * P.trip_log_passes + log_pass] = ((unsigned long long)(uint32_t)wall_clock64() << 32) |
                ((unsigned long long)__popcll(marching) << 24) | ((unsigned long long)__popcll(running) << 16) |
                ((unsigned long long)__popcll(sampling) << 8);
          ++log_pass;
        }
      }
#if VK_MARCH_AHEAD > 0
      // Round 6, an EXPERIMENT behind -DVK_MARCH_AHEAD=4 [-DVK_MARCH_AHEAD_AFTER=n]: round 4's run-ahead
      // through absent blocks — a ray on a run through empty space reads the table entries of its next VK_MARCH_AHEAD
      // positions TOGETHER (they do not depend on what is read: p advances by dir * block_length per absent block) and takes
      // the trips one after the other as far as the blocks are indeed not there: the same positions, the same checks, in
      // the same order, so the same bits — but GATED: only a wave that is past its VK_MARCH_AHEAD_AFTER-th pass through
      // this loop (a wave-uniform count; the mean wave is done in 6 passes, the slowest of the tracking scene take 46)
      // executes it. Ungated it lost because every wave paid its instructions while the device was full (r04: 91.8 us).
      ++pass;
      if (__builtin_amdgcn_readfirstlane(pass) > VK_MARCH_AHEAD_AFTER && absent_run >= 2)
      {
        constexpr int kAhead = VK_MARCH_AHEAD;
        int4 ahead[kAhead];
        int qx[kAhead], qy[kAhead], qz[kAhead];
        {
          f3 q = p;
#pragma unroll
          for (int k = 0; k < kAhead; ++k)
          {
            qx[k] = f2i(floorf(div_uniform(q.x, P.inv_block_length)));
            qy[k] = f2i(floorf(div_uniform(q.y, P.inv_block_length)));
            qz[k] = f2i(floorf(div_uniform(q.z, P.inv_block_length)));
            ahead[k] = reinterpret_cast<const int4*>(P.entries)[block_hash(qx[k], qy[k], qz[k], P.K)];
            q = add3(q, scale3(dir, P.block_length));
          }
          // (all sixteen words are wanted HERE: left alone, the compiler reads an entry's y word only after its x word has
          // matched — a second round trip per entry, one behind the other)
#pragma unroll
          for (int k = 0; k < kAhead; ++k)
            asm volatile("" : "+v"(ahead[k].x), "+v"(ahead[k].y), "+v"(ahead[k].z), "+v"(ahead[k].w));
        }
        bool ended = false, stopped = false;
#pragma unroll
        for (int k = 0; k < kAhead; ++k)
        {
          if (!stopped && !ended)
          {
            Entry entry;
            entry.ox = (int16_t)(ahead[k].x & 0xffff);
            entry.oy = (int16_t)((uint32_t)ahead[k].x >> 16);
            entry.oz = (int16_t)(ahead[k].y & 0xffff);
            entry.pad = 0;
            entry.data = ahead[k].z;
            entry.next = ahead[k].w;
            for (int guard = 0; !entry_is(entry, qx[k], qy[k], qz[k]) && entry.next != -1 && guard < kMaxChain; ++guard)
              entry = load_whole_entry(P.entries, (uint32_t)entry.next);
            const bool is = entry_is(entry, qx[k], qy[k], qz[k]);
            if (!is || entry.data == -1)
            {
              // the block is not there: the trip of the loop's last branch and its checks
              p = add3(p, scale3(dir, P.block_length));
              if (COUNT) ++trips;
              const float depth = xform_point(P.Tcw, p).z;
              if (++iters >= 500) { capped = true; ended = true; }
              else if (!(depth < bound.y)) ended = true;
            }
            else
            {
              stopped = true;
              cache.bx = qx[k]; cache.by = qy[k]; cache.bz = qz[k]; cache.data = entry.data; cache.valid = true;
            }
          }
        }
        if (ended) break;
        if (stopped) absent_run = 0;
        continue;
      }
#endif
      const int data = find_block(P, cache, bdir, bx, by, bz);
      absent_run = (data < 0 && !refine) ? absent_run + 1 : 0;
      if (COUNT) ++trips;
      bool done = false;

      if (data >= 0)
      {
        float sdf;
        bool sample = refine;

        // position in voxel units inside the block: tracer.cu:373-375 for the nearest-voxel
        // read and, with the same expression, :193-195 for the sample
        const float wx = div_uniform(p.x - bx * P.block_length, P.inv_voxel_length);
        const float wy = div_uniform(p.y - by * P.block_length, P.inv_voxel_length);
        const float wz = div_uniform(p.z - bz * P.block_length, P.inv_voxel_length);

        float nearest = 0.0f;
        // int(w) can reach 8 on a block face: clamped, see DESIGN.md
        const int vx = vmini(f2i(wx), 7);
        const int vy = vmini(f2i(wy), 7);
        const int vz = vmini(f2i(wz), 7);
        if (!refine)
        {
          const global_floats pool = (global_floats)reinterpret_cast<const float*>(P.voxels);
          if (POOL32)
            nearest = *(global_floats)((const char __attribute__((address_space(1)))*)pool +
                                       (__umul24((uint32_t)data, (uint32_t)(VK_BLOCK_VOXELS * 20)) + (uint32_t)(20 * (vz * 64 + vy * 8 + vx))));
          else
            nearest = (pool + ((size_t)(uint32_t)data * VK_BLOCK_VOXELS + (size_t)(vz * 64 + vy * 8 + vx)) * 5)[0];
          if (COUNT) P.touched[data] = 1;
        }
        // (Measured and rejected, r02: resolving the block one block-length further along the
        // ray while the nearest-voxel load is in flight — 31.8 us against 31.0.)
        // (Measured and rejected, r03, for rays that cross dozens of absent blocks behind an object's
        // silhouette — the room scene of the tracking workload, where 1 % of the waves live 75-105 us
        // and the launch with them, profiles/r03_e_wave_times_room.txt: reading the table entries of
        // the next four positions together and taking the trips through the absent ones in one go.
        // Inside this loop the fusion benchmark's raycast went from 31.9 to 34.0 us; as a second
        // instantiation chosen per wave by the depth of its bounds cell neither scene changed (31.8 /
        // 89.7 us against 31.4 / 89.6); as a phase before the loop nothing either — the long empty
        // runs come AFTER the ray has grazed the object's band. The same blocks prefetched into the
        // directory instead: room 90 -> 104 us. The slow waves are slow because their lanes are out
        // of phase — some sample while others still march — not because of the table reads: with the
        // reads batched a ray waits for at most 10 of them on 48 trips, and lives as long.)
        // (Measured and rejected, r03, on that theory: lanes that want a sample WAIT — state untouched,
        // they come back to the same decision — while they are the minority of the lanes still
        // marching, so that a tile's samples are taken together. Bit-exact, and slower: tracking scene
        // 78.4 -> 106 us, fusion benchmark 30.6 -> 31.8 us. A held lane's trips are added to the other
        // lanes' trips instead of riding along in the same iterations.)
        if (!refine) sample = (nearest <= 0.1f && nearest >= -0.5f);
        sdf = nearest;
        if (sample)
        {
          const Corners<POOL32> C = resolve_corners<COUNT, POOL32>(P, bdir, bx, by, bz, data, wx, wy, wz);
          sdf = corner_distance(P, C);
        }
        in_band = sample;

        if (sample)
        {
          sampled = true;
          sbx = bx; sby = by; sbz = bz; sdata = data;
          swx = wx; swy = wy; swz = wz;
        }

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
        capped = true;                                          // :437-442
        break;
      }

      if (!(depth < bound.y)) break;
    }
  }

  // the colour of the last sample (the reference computes one at every sample and keeps
  // the last, tracer.cu:392-393,412-413)
  if (__any(sampled))
  {
    if (sampled)
    {
      const Corners<POOL32> C = resolve_corners<COUNT, POOL32>(P, bdir, sbx, sby, sbz, sdata, swx, swy, swz);
      color = corner_color(P, C);
    }
  }
  if (capped) color = make3(1, 0, 0);

  const int pixel = y * P.image_width + x;
  if (COUNT) { if (P.march_steps) P.march_steps[pixel] = trips; }
  // (with readers in the same launch, on other XCDs: written through to where they all see it)
  if (P.rows_done) __hip_atomic_store(&P.depths[pixel], final_depth, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else P.depths[pixel] = final_depth;
  P.colors[3 * pixel + 0] = color.x;
  P.colors[3 * pixel + 1] = color.y;
  P.colors[3 * pixel + 2] = color.z;
}

template <bool COUNT, bool POOL32>
__device__ __forceinline__ void march_ray(const PointParams& P, int4* bdir, int x, int y, float2 bound, int log_wave = -1)
{
  march_ray_nested<COUNT, POOL32>(P, bdir, x, y, bound, log_wave);
}

}  // namespace vk
