// vk_requests.hpp — the request pass of Volume::SetView (ref: src/volume.cu:87-301 CreateAllocationRequestsKernel,
// :497-518): device code shared by vk_volume.hip (the pass as a launch of its own, and the later rounds, which probe the
// same way) and vk_trace.hip (the pass of the NEXT frame as trailing workgroups of a raycast launch,
// vk_trace_ahead_requests). Moved out of vk_volume.hip in round 4, unchanged.
#pragma once

#include "vk_common.hpp"

#include <string.h>

namespace vk
{

__device__ __forceinline__ unsigned long long request_key(int type, int bx, int by, int bz)
{
  return ((unsigned long long)(uint16_t)type << 48) | ((unsigned long long)(uint16_t)(int16_t)bz << 32) |
         ((unsigned long long)(uint16_t)(int16_t)by << 16) | (unsigned long long)(uint16_t)(int16_t)bx;
}

// SetView normally starts with a pass that turns every TRUE of the previous frame
// into UNKNOWN (volume.cu:465-471). The fused vk_volume_set_view skips that pass:
// whatever marks an entry visible during the frame sets bit 2 of the byte on top
// of its old value instead, and update_visibility_kernel — which reads every byte
// anyway — decodes (bit 2 ? TRUE : the reset of the old value) and stores the plain
// 0/1/2 the reference would hold. Bit 2 never survives the call.
constexpr uint8_t kTouched = 4;
// Bits 3..6 of a touched byte: 1 + the image row band (of VK_BANDS) of the depth pixel whose ray set the bit, 0 when
// whatever touched the entry had no pixel (the handle pass). The visibility pass bins the visible entries by it
// (banded_lists below) on its way to storing the plain 0/1/2; like bit 2 the tag never survives the call. Any of the
// rays that cross a block may win the byte: a block projects to a few rows, every answer is as good.
__device__ __forceinline__ uint32_t band_tag(int y, int height)
{
  int band = (int)(((long long)y * VK_BANDS) / (height > 0 ? height : 1));
  band = band < 0 ? 0 : (band >= VK_BANDS ? VK_BANDS - 1 : band);
  return (uint32_t)(band + 1) << 3;
}

// how a request pass marks an entry visible: plainly (the staged entry points), with the touched
// bit (the fused SetView's first round, decoded by its visibility pass), or plainly AND, when the
// entry was not visible yet, entered in the visible list (the fused SetView's later rounds, which
// run after the visibility pass)
enum { MARK_PLAIN = 0, MARK_DEFER = 1, MARK_APPEND = 2 };

// `known` >= 0: the entry's byte as the caller has already read it (with the probe's table entry, so
// that the two reads travel together instead of one behind the other)
template <int MARK>
__device__ __forceinline__ void mark_visible(const vk_volume& v, uint32_t index, int known = -1, uint32_t tag = 0)
{
  // the reference stores unconditionally (volume.cu:190); reading first keeps
  // hundreds of rays that cross the same block from all storing the same byte
  uint8_t* vis = v.block_visibility;
  const uint8_t old = known >= 0 ? (uint8_t)known : vis[index];
  if (MARK == MARK_DEFER) { if (!(old & kTouched)) vis[index] = (uint8_t)((old & 3u) | kTouched | tag); }
  else if (MARK == MARK_PLAIN) { if (old != VK_VISIBILITY_TRUE) vis[index] = VK_VISIBILITY_TRUE; }
  else if (old != VK_VISIBILITY_TRUE)
  {
    // exactly one of the lanes that find the entry not visible lists it: the byte is swapped in
    // its word (the buffer is 4-byte aligned, check_volume; rare path)
    uint32_t* word = reinterpret_cast<uint32_t*>(vis) + (index >> 2);
    const int shift = 8 * (int)(index & 3u);
    uint32_t seen = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;)
    {
      if (((seen >> shift) & 0xffu) == (uint32_t)VK_VISIBILITY_TRUE) return;
      const uint32_t want = (seen & ~(0xffu << shift)) | ((uint32_t)VK_VISIBILITY_TRUE << shift);
      const uint32_t got = atomicCAS(word, seen, want);
      if (got == seen) break;
      seen = got;
    }
    v.visible_blocks[atomicAdd(&v.counters[VK_CTR_VISIBLE], 1)] = (int)index;
  }
}

// What the fused SetView (vk_volume_set_view_rounds) keeps of a request pass so that its later
// rounds need not walk the rays again. A bucket takes one request per round; when two DIFFERENT
// blocks ask for the same bucket, one of them loses and has to ask again in another round
// (SetView is called three times per frame upstream, apps/vulcan/vulcan.cu:316-318, for exactly
// this). Every key ever posted to a slot either finds a different key there or is later
// replaced by one whose poster finds it, and whoever sees the two keys files the SMALLER one —
// the loser: after the pass the list holds every block that lost, once, and `contended` is set
// if and only if there is one. `posted` (LDS, optional): the
// buckets that received their first request of this pass.
// behind the VK_CTR_PUBLIC counters: two key sets, then their two slot lists (vk.h)
__host__ __device__ inline unsigned long long* retry_table(int32_t* counters, int which)
{
  return reinterpret_cast<unsigned long long*>(counters + VK_CTR_PUBLIC) + (size_t)which * VK_RETRY_SLOTS;
}
__host__ __device__ inline int* retry_slots(int32_t* counters, int which)
{
  return counters + VK_CTR_PUBLIC + 2 * 2 * VK_RETRY_SLOTS + which * VK_RETRY_KEYS;
}

// ... and behind those the buckets that received their first request of the current request pass
// (bit 31: the request is an EXCESS one — every ray that asks for a bucket sees the same main entry,
// so the first poster's type is the bucket's type)
__host__ __device__ inline int* posted_list(int32_t* counters)
{
  return counters + VK_CTR_PUBLIC + 2 * 2 * VK_RETRY_SLOTS + 2 * VK_RETRY_KEYS;
}
constexpr uint32_t kPostedExcess = 0x80000000u;
constexpr uint32_t kPostedBucket = 0x7fffffffu;

struct Retry
{
  int* contended;               // nullptr: nothing is recorded (the staged entry points)
  int* origin_seen;             // set when block (0,0,0) is "found" in an unallocated main entry (see probe_block)
  int* count;                   // distinct keys filed so far (may exceed capacity: then *overflow is set)
  int* overflow;
  unsigned long long* table;    // open-addressing set of the keys (VK_RETRY_SLOTS slots, 0 = free)
  int* slots;                   // where in the table each distinct key sits, in order of arrival
  int capacity;                 // entries `slots` holds
  int* posted;
  int* posted_tail;             // optional, next to `posted`: the last entry of the bucket's chain (an EXCESS request links there)
  int* posted_count;
  int posted_capacity;
};

// files a lost key once: hundreds of rays cross the same block and each of them finds out that
// it lost, so the list is fronted by a set (linear probing from a multiplicative hash).
// (Measured, r03: as a function of its own — the code leaves the request pass, but its call sites
// spill through scratch memory: 18.6 us against 15.7 inlined; recording contests at all costs the
// request pass ~1.5 us, profiles/r03_c_* against r03_h_*.)
__device__ __forceinline__ void file_loser(const Retry& retry, unsigned long long key)
{
  uint32_t at = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 40) & (uint32_t)(VK_RETRY_SLOTS - 1);
  for (int tries = 0; tries < 64; ++tries)
  {
    const unsigned long long old = atomicCAS(retry.table + at, 0ull, key);
    if (old == key) return;
    if (old == 0ull)
    {
      const int n = atomicAdd(retry.count, 1);
      if (n < retry.capacity) retry.slots[n] = (int)at;
      else __hip_atomic_store(retry.overflow, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    at = (at + 1) & (uint32_t)(VK_RETRY_SLOTS - 1);
  }
  __hip_atomic_store(retry.overflow, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void post_request(const vk_volume& v, uint32_t h, int type, int bx, int by, int bz,
    const Retry& retry, uint32_t tail)
{
  unsigned long long* slot = reinterpret_cast<unsigned long long*>(v.allocation_blocks) + h;
  const unsigned long long key = request_key(type, bx, by, bz);
  // monotonic max: skip the atomic when the slot already holds a key >= ours
  unsigned long long seen = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (seen < key) seen = atomicMax(slot, key);
  if (retry.contended)
  {
    if (seen != 0ull && seen != key)
    {
      file_loser(retry, seen < key ? seen : key);
      if (__hip_atomic_load(retry.contended, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
        __hip_atomic_store(retry.contended, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (seen == 0ull && retry.posted)
    {
      const int at = atomicAdd(retry.posted_count, 1);
      if (at < retry.posted_capacity)
      {
        retry.posted[at] = (int)(h | (type == VK_ALLOC_EXCESS ? kPostedExcess : 0u));
        if (retry.posted_tail) retry.posted_tail[at] = (int)tail;
      }
    }
  }
  if (v.allocation_types[h] != (uint8_t)type) v.allocation_types[h] = (uint8_t)type;
}

// volume.cu:183-239: what one ray does with one crossed block once the bucket's
// main entry is known
template <int MARK>
__device__ __forceinline__ void probe_block(const vk_volume& v, uint32_t hash_code, Entry entry,
    int bx, int by, int bz, const Retry& retry, int main_byte = -1, uint32_t tag = 0)
{
  if (entry_is(entry, bx, by, bz))
  {
    mark_visible<MARK>(v, hash_code, main_byte, tag);
    // An unallocated main entry holds block (0,0,0) and compares equal to it (volume.cu:186-191):
    // the origin block counts as present without ever having been requested — until another
    // block takes that entry, from when on its rays do request it. It is the one block a later
    // SetView round can request that did not lose in the round before; the rounds need to know.
    if (entry.data == -1 && retry.contended &&
        __hip_atomic_load(retry.origin_seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
      __hip_atomic_store(retry.origin_seen, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  else if (entry.data == -1)
  {
    mark_visible<MARK>(v, hash_code, main_byte, tag);
    post_request(v, hash_code, VK_ALLOC_MAIN, bx, by, bz, retry, hash_code);
  }
  else
  {
    bool found = false;
    uint32_t index = hash_code;

    // bounded by the excess region: a corrupt table (a cycle) must not hang the GPU
    for (int guard = 0; entry.next != -1 && guard < v.excess_block_count; ++guard)
    {
      index = (uint32_t)entry.next;
      entry = load_entry(v.hash_entries, index);

      if (entry_is(entry, bx, by, bz))
      {
        mark_visible<MARK>(v, index, -1, tag);
        found = true;
        break;
      }
    }

    // (`index` is now the chain's last entry: the table does not change during the pass, so every ray
    // that asks for this bucket ends its walk there)
    if (!found) post_request(v, hash_code, VK_ALLOC_EXCESS, bx, by, bz, retry, index);
  }
}

// value held by lane - 1 (lane 0 gets its own): one DPP move (wave_shr:1)
// instead of the ds_bpermute that __shfl_up costs
__device__ __forceinline__ int left_lane(int v)
{
  return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false);
}

#ifndef VK_REQUEST_PROBES
#define VK_REQUEST_PROBES 6
#endif

struct RequestParams
{
  vk_volume v;
  const float* depth;
  int width, height;
  Projection k;            // with 1 / fx, 1 / fy (vk_common.hpp)
  Rt Twd;
  // PREP only: LightIntegrator's per-pixel preparation rides along (vk_volume_set_view_prepare)
  const float* colors;
  const float* normals;
  Rt Tcd;
  float depth_threshold;
  float* mask;
  float4* records;
  float* normals_out;   // PREP == 2: the frame's normals are computed here (Frame::ComputeNormals) and written out
};

// ref: volume.cu:87-301, the walk of one depth pixel (x, y). Called by WHOLE waves whose 64
// lanes hold 64 consecutive pixels of one row (lanes past the image stay in: their neighbours
// read their registers), so that the depth read is one coalesced 256-byte load.
template <int MARK>
__device__ __forceinline__ void request_walk(const RequestParams& P, int x, int y, const Retry& retry)
{
  const vk_volume& v = P.v;
  const uint32_t K = (uint32_t)v.main_block_count;
  const float block_length = VK_BLOCK_RESOLUTION * v.voxel_length;
  const float truncation_length = v.truncation_length;
  const uint32_t tag = MARK == MARK_DEFER ? band_tag(y, P.height) : 0u;   // the same for the wave's 64 pixels of a row

  f3 direction = unproject(P.k, x + 0.5f, y + 0.5f);
  direction = xform_dir(P.Twd, direction);
  const f3 origin = make3(P.Twd.r[3], P.Twd.r[7], P.Twd.r[11]);

  // lanes without a usable depth stay in the wave (their neighbours read their
  // registers below) but walk nothing
  float depth = 0.0f;
  if (x < P.width) depth = P.depth[y * P.width + x];
  const bool usable = x < P.width && !(depth < v.min_depth || depth > v.max_depth);

  const f3 Xwp = add3(origin, scale3(direction, depth));
  direction = normalized3(direction);
  const f3 begin = sub3(Xwp, scale3(direction, truncation_length));
  const f3 end = add3(Xwp, scale3(direction, truncation_length));

  const int step_x = (direction.x < 0) ? -1 : 1;
  const int step_y = (direction.y < 0) ? -1 : 1;
  const int step_z = (direction.z < 0) ? -1 : 1;

  const float inv_block_length = 1.0f / block_length;
  int bx = f2i(floorf(begin.x * inv_block_length));
  int by = f2i(floorf(begin.y * inv_block_length));
  int bz = f2i(floorf(begin.z * inv_block_length));
  const int ex = f2i(floorf(end.x * inv_block_length));
  const int ey = f2i(floorf(end.y * inv_block_length));
  const int ez = f2i(floorf(end.z * inv_block_length));

  const float ox = block_length * (bx + vmaxi(0, step_x)) - begin.x;
  const float oy = block_length * (by + vmaxi(0, step_y)) - begin.y;
  const float oz = block_length * (bz + vmaxi(0, step_z)) - begin.z;

  float tmax_x = ox / direction.x;
  float tmax_y = oy / direction.y;
  float tmax_z = oz / direction.z;
  if (direction.x == 0) tmax_x = (float)1E20;
  if (direction.y == 0) tmax_y = (float)1E20;
  if (direction.z == 0) tmax_z = (float)1E20;

  const float tdelta_x = (step_x * block_length) / direction.x;
  const float tdelta_y = (step_y * block_length) / direction.y;
  const float tdelta_z = (step_z * block_length) / direction.z;

  // The walk itself never depends on what the hash table holds, so it is run
  // first and its probes are issued together: the reference's loop (one dependent
  // table read per crossed block, :174-299) becomes kProbe independent reads in
  // flight. A 2*trunc segment crosses 3-4 blocks; walks longer than kProbe fall
  // back to the step-by-step loop below.
  constexpr int kProbe = VK_REQUEST_PROBES;
  int sbx[kProbe], sby[kProbe], sbz[kProbe];
  uint32_t shash[kProbe];
  bool walking = usable;

  // the three products of the hash (volume.cu:168-180) follow the walk by addition:
  // a step changes one coordinate by +-1, i.e. its product by +-prime (mod 2^32), which
  // replaces three quarter-rate 32-bit multiplies per crossed block by one add
  uint32_t hx = (uint32_t)bx * 73856093u, hy = (uint32_t)by * 19349669u, hz = (uint32_t)bz * 83492791u;
  const uint32_t dhx = step_x < 0 ? 0u - 73856093u : 73856093u;
  const uint32_t dhy = step_y < 0 ? 0u - 19349669u : 19349669u;
  const uint32_t dhz = step_z < 0 ? 0u - 83492791u : 83492791u;

#pragma unroll
  for (int sidx = 0; sidx < kProbe; ++sidx)
  {
    sbx[sidx] = bx; sby[sidx] = by; sbz[sidx] = bz;
    shash[sidx] = walking ? (hx ^ hy ^ hz) % K : 0xffffffffu;

    if (walking)
    {
      // :242-295 advance to the next block; `walking` drops when the end block is passed
      if (tmax_x < tmax_y)
      {
        if (tmax_x < tmax_z) { bx += step_x; hx += dhx; if (bx == ex + step_x) walking = false; else tmax_x += tdelta_x; }
        else                 { bz += step_z; hz += dhz; if (bz == ez + step_z) walking = false; else tmax_z += tdelta_z; }
      }
      else
      {
        if (tmax_y < tmax_z) { by += step_y; hy += dhy; if (by == ey + step_y) walking = false; else tmax_y += tdelta_y; }
        else                 { bz += step_z; hz += dhz; if (bz == ez + step_z) walking = false; else tmax_z += tdelta_z; }
      }
    }
  }

  // Neighbouring pixels of a row cross the same blocks (a block is ~11 px wide at
  // 2 m), and everything a probe does — marking the entry visible, posting the
  // max-key request — is idempotent. A lane therefore skips a probe when the lane
  // to its left makes the identical one in the same slot; what remains is about
  // one probe per distinct block per wave instead of one per ray.
  const int lane = lane_id();
#pragma unroll
  for (int sidx = 0; sidx < kProbe; ++sidx)
  {
    const uint32_t left_hash = (uint32_t)left_lane((int)shash[sidx]);
    const int left_x = left_lane(sbx[sidx]), left_y = left_lane(sby[sidx]), left_z = left_lane(sbz[sidx]);
    const bool same = lane > 0 && left_hash == shash[sidx] && left_x == sbx[sidx] && left_y == sby[sidx] && left_z == sbz[sidx];
    if (same) shash[sidx] = 0xffffffffu;
  }

  // (the visibility byte of each bucket's main entry is asked for with the entry: most blocks are
  // found there, and marking them visible would otherwise start with a read of its own)
  Entry sent[kProbe];
  uint8_t sbyte[kProbe];
#pragma unroll
  for (int sidx = 0; sidx < kProbe; ++sidx)
  {
    const uint32_t at = shash[sidx] == 0xffffffffu ? 0u : shash[sidx];
    sent[sidx] = load_entry(v.hash_entries, at);
    sbyte[sidx] = v.block_visibility[at];
  }

#pragma unroll
  for (int sidx = 0; sidx < kProbe; ++sidx)
  {
    if (shash[sidx] == 0xffffffffu) continue;
    probe_block<MARK>(v, shash[sidx], sent[sidx], sbx[sidx], sby[sidx], sbz[sidx], retry, (int)sbyte[sidx], tag);
  }

  // A segment of 2*trunc crosses a bounded number of blocks; the cap only
  // guarantees that every wave exits on NaN / degenerate input.
  for (int guard = 0; walking && guard < 4096; ++guard)
  {
    const uint32_t hash_code = block_hash(bx, by, bz, K);
    probe_block<MARK>(v, hash_code, load_entry(v.hash_entries, hash_code), bx, by, bz, retry, -1, tag);

    if (tmax_x < tmax_y)
    {
      if (tmax_x < tmax_z)
      {
        bx += step_x;
        if (bx == ex + step_x) break;
        tmax_x += tdelta_x;
      }
      else
      {
        bz += step_z;
        if (bz == ez + step_z) break;
        tmax_z += tdelta_z;
      }
    }
    else
    {
      if (tmax_y < tmax_z)
      {
        by += step_y;
        if (by == ey + step_y) break;
        tmax_y += tdelta_y;
      }
      else
      {
        bz += step_z;
        if (bz == ez + step_z) break;
        tmax_z += tdelta_z;
      }
    }
  }
}

// ref: volume.cu:87-301. One lane per depth pixel; the lanes of a wave cover a
// 64x1 run of a row so the depth read is one coalesced 256-byte load.
//
// PREP: the same pass also leaves LightIntegrator's frame mask and per-pixel records
// (light_integrator.cu:16-94,215-225; frame_mask_kernel in vk_integrate.hip is the launch of
// its own) — both walk the depth image one lane per pixel, and as a launch of its own the
// mask pass costs ~6 us of which ~4.5 are the launch. The workgroup's 64x4 pixels need the
// depth window [x-1, x+5] x [y-1, y+5]: a 70x10 tile in LDS.
// PREP == 2: the frame's normals are not there yet — Frame::ComputeNormals (frame.cu:9-122) is a
// 5-tap stencil on the same depth image, two pixels to each side, so the tile grows by one column
// and one row (71x11 from (x-2, y-2)), the pass computes the normal it needs for the record itself
// and writes it to the frame's normal image: one launch (~3.4 us of the frame) and a 3.7 MB read less.
// The work of ONE workgroup of the request pass: the 64 x 4 pixels whose first is (64 group_x, 4 group_y). A function,
// not the kernel, so that the pass can run as a launch of its own (create_requests_kernel, vk_volume.hip) and as
// trailing workgroups of the previous frame's raycast launch (trace_and_request_kernel, vk_trace.hip).
template <bool DEFER, int PREP>
__device__ __forceinline__ void requests_group(const RequestParams P, const Retry retry, const int group_x, const int group_y)
{
  const int x = group_x * 64 + (int)(threadIdx.x & 63);
  const int y = group_y * 4 + (int)(threadIdx.x >> 6);
  // the banded visible lists start empty: the visibility pass of this SetView fills them
  if (DEFER && group_x == 0 && group_y == 0 && threadIdx.x < VK_BANDS) band_counts(P.v.counters)[threadIdx.x] = 0;

  // PREP: the loads of the depth tile and of the pixel's colour and normal are issued first and
  // consumed after the request walk, which hides their latency
  constexpr int HALO = (PREP == 2) ? 2 : 1;                 // columns / rows in front of the workgroup's pixels
  constexpr int TW = 69 + HALO, TH = 9 + HALO, TS = 72;
  constexpr int LOADS = (TW * TH + 255) / 256;
  __shared__ float tile[PREP ? TH * TS : 1];
  float staged[LOADS] = {};
  vf3 prep_rgb = {0.0f, 0.0f, 0.0f}, prep_n = {0.0f, 0.0f, 0.0f};
  if (PREP)
  {
    const int x0 = group_x * 64 - HALO, y0 = group_y * 4 - HALO;
#pragma unroll
    for (int t = 0; t < LOADS; ++t)
    {
      const int i = (int)threadIdx.x + 256 * t;
      const int r = i / TW, c = i - r * TW;
      const int vx = x0 + c, vy = y0 + r;
      if (i < TW * TH && vx >= 0 && vx < P.width && vy >= 0 && vy < P.height) staged[t] = P.depth[vy * P.width + vx];
    }
    if (x < P.width && y < P.height)
    {
      const int index = y * P.width + x;
      prep_rgb = *reinterpret_cast<const vf3*>(P.colors + 3 * (size_t)index);
      if (PREP == 1) prep_n = *reinterpret_cast<const vf3*>(P.normals + 3 * (size_t)index);
    }
  }

  if (y < P.height) request_walk<DEFER ? MARK_DEFER : MARK_PLAIN>(P, x, y, retry);   // whole wave

  if (PREP)
  {
#pragma unroll
    for (int t = 0; t < LOADS; ++t)
    {
      const int i = (int)threadIdx.x + 256 * t;
      if (i < TW * TH) tile[(i / TW) * TS + (i % TW)] = staged[t];
    }
    __syncthreads();
    // The frame mask's 7 x 7 window (light_window_mask, vk_common.hpp: smallest and largest depth, NaNs passed over) in two
    // steps: the seven taps of a row once for every (row, centre column) the workgroup needs — 10 rows x 64 columns, 2.5 per
    // thread — then seven of those per pixel: 31 LDS reads and 49 min / max per thread instead of 49 and 98. The same set of
    // values goes through fminf / fmaxf, which do not care about the order.
    __shared__ float row_min[10 * 64], row_max[10 * 64];
#pragma unroll
    for (int t = 0; t < 3; ++t)
    {
      const int e = (int)threadIdx.x + 256 * t;
      if (e < 10 * 64)
      {
        const float* at = tile + (HALO - 1 + (e >> 6)) * TS + (HALO + 2 + (e & 63));     // the centre of the row's seven taps
        float lo = +FLT_MAX, hi = -FLT_MAX;
#pragma unroll
        for (int j = -3; j <= 3; ++j)
        {
          lo = fminf(at[j], lo);
          hi = fmaxf(at[j], hi);
        }
        row_min[e] = lo;
        row_max[e] = hi;
      }
    }
    __syncthreads();
    if (x < P.width && y < P.height)
    {
      const int index = y * P.width + x;
      const int lx = (int)(threadIdx.x & 63) + HALO, ly = (int)(threadIdx.x >> 6) + HALO;   // this pixel in the tile
      f3 normal = make3(prep_n.x, prep_n.y, prep_n.z);
      if (PREP == 2)
      {
        normal = normal_from_taps(P.k, x, y, tile[ly * TS + lx], tile[ly * TS + lx - 2], tile[ly * TS + lx + 2],
            tile[(ly - 2) * TS + lx], tile[(ly + 2) * TS + lx]);
        P.normals_out[3 * (size_t)index + 0] = normal.x;
        P.normals_out[3 * (size_t)index + 1] = normal.y;
        P.normals_out[3 * (size_t)index + 2] = normal.z;
      }
      const f3 Xcn = xform_dir(P.Tcd, normal);      // light_integrator.cu:223
      float m = 0.0f;
      if (light_color_usable(prep_rgb.x, prep_rgb.y, prep_rgb.z))
      {
        // the window's centre is (x + 2, y + 2): rows y - 1 .. y + 5 are the workgroup's rows (threadIdx.x >> 6) .. + 6
        const int first = (int)(threadIdx.x >> 6) * 64 + (int)(threadIdx.x & 63);
        float dmin = +FLT_MAX, dmax = -FLT_MAX;
#pragma unroll
        for (int i = 0; i < 7; ++i)
        {
          dmin = fminf(row_min[first + 64 * i], dmin);
          dmax = fmaxf(row_max[first + 64 * i], dmax);
        }
        m = (dmax - dmin <= P.depth_threshold) ? 1.0f : 0.0f;
      }
      P.mask[index] = m;
      P.records[index] = make_float4(Xcn.x, Xcn.y, Xcn.z, m);
    }
  }
}

// ---- host side: the launch parameters of the pass ------------------------------------------------------------------

constexpr int kArrivalMaxWorkgroups = 65535;   // the handle + visibility launch counts its arrivals in 16 bits (vk_volume.hip)

// buckets the posted list holds (vk_test_hooks.posted_capacity: a small list sends the handle pass to the flags)
inline int posted_capacity()
{
  const int n = vk_hook(VK_HOOK_POSTED_CAPACITY);
  return (n >= 0 && n < VK_POSTED_SLOTS) ? n : VK_POSTED_SLOTS;
}

// distinct keys per retry list (vk_test_hooks.retry_capacity: a small list overflows on purpose)
inline int retry_capacity()
{
  const int n = vk_hook(VK_HOOK_RETRY_CAPACITY);
  return (n > 0 && n < VK_RETRY_KEYS) ? n : VK_RETRY_KEYS;
}

// vk_test_hooks.set_view_unfused: the fused SetView as three launches (requests, handle + later rounds,
// visibility) instead of two — kept for comparison, and as the reference for the two-launch form
// (also the form for a table too large for the arrival count of the two-launch form: > 67 M entries)
inline bool set_view_unfused(const vk_volume* v)
{
  const bool unfused = vk_hook(VK_HOOK_SET_VIEW_UNFUSED) == 1;
  return unfused || ((long long)v->main_block_count + v->excess_block_count) / 1024 + 16 > kArrivalMaxWorkgroups;
}

// whether LightIntegrator's preparation can ride in the request pass of `frame`: the frame has what the integrator needs,
// in the depth image's size (light_integrator.cu:277-293 walks the colour image with it), and names its content
inline bool prep_rides(const vk_light_prep* prep, const vk_frame* frame)
{
  return prep && prep->mask && prep->records && frame->color && frame->normals && frame->content_id != 0 &&
      (long long)frame->width * frame->height <= (long long)prep->capacity &&
      (reinterpret_cast<uintptr_t>(prep->records) & 15) == 0 &&
      (frame->color_width <= 0 || frame->color_width == frame->width) &&
      (frame->color_height <= 0 || frame->color_height == frame->height);
}

// what vk_volume_set_view* notes in the preparation once its request pass has made it for `frame`
inline void prep_note_made(vk_light_prep* prep, const vk_frame* frame)
{
  prep->depth = frame->depth;
  prep->color = frame->color;
  prep->normals = frame->normals;
  prep->width = frame->width;
  prep->height = frame->height;
  prep->depth_to_color = frame->depth_to_color;
  prep->content_id = frame->content_id;
  prep->prepared_threshold = prep->depth_threshold;
  prep->valid = 1;
  prep->normals_out = nullptr;   // done
}

// RequestParams / Retry of the pass for (v, depth image, view); `fused`: the pass of vk_volume_set_view* (files the requests
// that lose, lists the buckets it posts to); prep_frame + prep: LightIntegrator's preparation rides along. Returns the
// PREP template argument the launch needs: 0, 1, or 2 (the frame's normals are computed on the way).
inline int build_request_pass(RequestParams& P, Retry& retry, const vk_volume* v, const float* depth, int width, int height,
    const vk_projection* projection, const vk_transform* Twd, const vk_frame* prep_frame, const vk_light_prep* prep, bool fused)
{
  P.v = *v;
  P.depth = depth;
  P.width = width;
  P.height = height;
  P.k = make_projection(*projection);
  P.Twd = make_rt(Twd->m);
  P.colors = P.normals = nullptr;
  P.Tcd = P.Twd;
  P.depth_threshold = 0.0f;
  P.mask = nullptr;
  P.records = nullptr;
  P.normals_out = nullptr;
  // the fused SetView files the requests that lose a bucket contest (post_request) for its later rounds
  memset(&retry, 0, sizeof(retry));
  if (fused)
  {
    retry.contended = v->counters + VK_CTR_CONTENDED;
    retry.origin_seen = v->counters + VK_CTR_ORIGIN_SEEN;
    retry.count = v->counters + VK_CTR_RETRY_COUNT;
    retry.overflow = v->counters + VK_CTR_RETRY_OVERFLOW;
    retry.table = retry_table(v->counters, 0);
    retry.slots = retry_slots(v->counters, 0);
    retry.capacity = retry_capacity();
    if (!set_view_unfused(v))
    {
      retry.posted = posted_list(v->counters);
      retry.posted_tail = posted_list(v->counters) + VK_POSTED_SLOTS;
      retry.posted_count = v->counters + VK_CTR_POSTED;
      retry.posted_capacity = posted_capacity();
    }
  }
  if (!(prep && prep_frame)) return 0;
  P.colors = prep_frame->color;
  P.normals = prep_frame->normals;
  P.Tcd = make_rt(prep_frame->depth_to_color.m);
  P.depth_threshold = prep->depth_threshold;
  P.mask = prep->mask;
  P.records = reinterpret_cast<float4*>(prep->records);
  P.normals_out = prep->normals_out;
  return prep->normals_out ? 2 : 1;
}

}  // namespace vk
