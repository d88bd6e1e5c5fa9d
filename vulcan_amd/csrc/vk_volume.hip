// vk_volume.hip — hashed voxel volume: initialisation, block allocation and
// visible-block compaction for gfx950 (ref: src/volume.cu).
//
// Per SetView the reference issues thrust::replace + 3 kernels and reads the
// visible count back to the host (volume.cu:494). Here the count stays in
// v.counters[VK_CTR_VISIBLE]; every consumer reads it on the device.
#include "vk_common.hpp"

#include <stdlib.h>
#include <string.h>

using namespace vk;

namespace
{

// ------------------------------------------------------------------ init ----

// Voxel::Empty() = {1.0f, 0, 0, 0, (short)0, (short)0}: a 5-dword pattern.
__global__ __launch_bounds__(256) void fill_voxels_kernel(float4* __restrict__ voxels4, size_t n4)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n4; j += stride)
  {
    const size_t d = j * 4;  // first dword index of this float4
    float4 v;
    v.x = ((d + 0) % 5 == 0) ? 1.0f : 0.0f;
    v.y = ((d + 1) % 5 == 0) ? 1.0f : 0.0f;
    v.z = ((d + 2) % 5 == 0) ? 1.0f : 0.0f;
    v.w = ((d + 3) % 5 == 0) ? 1.0f : 0.0f;
    voxels4[j] = v;
  }
}

__global__ __launch_bounds__(256) void init_tables_kernel(vk_volume v)
{
  const int max_count = v.main_block_count + v.excess_block_count;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;

  if (i < max_count)
  {
    reinterpret_cast<int4*>(v.hash_entries)[i] = make_int4(0, 0, -1, -1);  // HashEntry()
    v.free_voxel_blocks[i] = i;
    v.block_visibility[i] = VK_VISIBILITY_FALSE;
  }

  if (i < v.main_block_count)
  {
    v.allocation_types[i] = VK_ALLOC_NONE;
    reinterpret_cast<unsigned long long*>(v.allocation_blocks)[i] = 0ull;
  }

  if (i < VK_CTR_COUNT)
  {
    int value = 0;
    if (i == VK_CTR_EXCESS_PTR) value = v.main_block_count;  // volume.cu:581-586
    if (i == VK_CTR_VOXEL_PTR) value = max_count - 1;        // volume.cu:596-601
    v.counters[i] = value;
  }
}

// ------------------------------------------------------- reset visibility ----

// TRUE (2) -> UNKNOWN (0), FALSE (1) stays (volume.cu:469); 4 entries per lane.
__global__ __launch_bounds__(256) void reset_visibility_kernel(uint8_t* __restrict__ vis, int count)
{
  const int words = count >> 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;

  if (i < words)
  {
    uint32_t w = reinterpret_cast<uint32_t*>(vis)[i];
    // a byte is 2 exactly when bit 1 is set (values are 0, 1, 2)
    const uint32_t is_true = (w >> 1) & 0x01010101u;
    w &= ~(is_true * 0x3u);
    reinterpret_cast<uint32_t*>(vis)[i] = w;
  }

  if (i < (count & 3))
  {
    const int j = (words << 2) + i;
    if (vis[j] == VK_VISIBILITY_TRUE) vis[j] = VK_VISIBILITY_UNKNOWN;
  }
}

// ------------------------------------------------------ allocation requests ----

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

template <bool DEFER>
__device__ __forceinline__ void mark_visible(uint8_t* vis, uint32_t index)
{
  // the reference stores unconditionally (volume.cu:190); reading first keeps
  // hundreds of rays that cross the same block from all storing the same byte
  const uint8_t old = vis[index];
  if (DEFER) { if (!(old & kTouched)) vis[index] = old | kTouched; }
  else if (old != VK_VISIBILITY_TRUE) vis[index] = VK_VISIBILITY_TRUE;
}

// `contended` (optional): set when two DIFFERENT blocks ask for the same bucket in one round —
// the loser has to ask again in another round (SetView is called three times per frame upstream,
// apps/vulcan/vulcan.cu:316-318, for exactly this). Every key ever posted to a slot either
// finds a different key there or is later replaced by one whose poster finds it, so the flag
// is exact: it is set if and only if some request of this round is lost.
__device__ __forceinline__ void post_request(const vk_volume& v, uint32_t h, int type, int bx, int by, int bz,
    int* contended)
{
  unsigned long long* slot = reinterpret_cast<unsigned long long*>(v.allocation_blocks) + h;
  const unsigned long long key = request_key(type, bx, by, bz);
  // monotonic max: skip the atomic when the slot already holds a key >= ours
  unsigned long long seen = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (seen < key) seen = atomicMax(slot, key);
  if (contended && seen != 0ull && seen != key &&
      __hip_atomic_load(contended, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
    __hip_atomic_store(contended, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (v.allocation_types[h] != (uint8_t)type) v.allocation_types[h] = (uint8_t)type;
}

// volume.cu:183-239: what one ray does with one crossed block once the bucket's
// main entry is known
template <bool DEFER>
__device__ __forceinline__ void probe_block(const vk_volume& v, uint32_t hash_code, Entry entry,
    int bx, int by, int bz, int* contended)
{
  if (entry_is(entry, bx, by, bz))
  {
    mark_visible<DEFER>(v.block_visibility, hash_code);
  }
  else if (entry.data == -1)
  {
    mark_visible<DEFER>(v.block_visibility, hash_code);
    post_request(v, hash_code, VK_ALLOC_MAIN, bx, by, bz, contended);
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
        mark_visible<DEFER>(v.block_visibility, index);
        found = true;
        break;
      }
    }

    if (!found) post_request(v, hash_code, VK_ALLOC_EXCESS, bx, by, bz, contended);
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
  vk_projection k;
  Rt Twd;
  // PREP only: LightIntegrator's per-pixel preparation rides along (vk_volume_set_view_prepare)
  const float* colors;
  const float* normals;
  Rt Tcd;
  float depth_threshold;
  float* mask;
  float4* records;
};

// ref: volume.cu:87-301, the walk of one depth pixel (x, y). Called by WHOLE waves whose 64
// lanes hold 64 consecutive pixels of one row (lanes past the image stay in: their neighbours
// read their registers), so that the depth read is one coalesced 256-byte load.
template <bool DEFER>
__device__ __forceinline__ void request_walk(const RequestParams& P, int x, int y, int* contended)
{
  const vk_volume& v = P.v;
  const uint32_t K = (uint32_t)v.main_block_count;
  const float block_length = VK_BLOCK_RESOLUTION * v.voxel_length;
  const float truncation_length = v.truncation_length;

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

  Entry sent[kProbe];
#pragma unroll
  for (int sidx = 0; sidx < kProbe; ++sidx)
    sent[sidx] = load_entry(v.hash_entries, shash[sidx] == 0xffffffffu ? 0u : shash[sidx]);

#pragma unroll
  for (int sidx = 0; sidx < kProbe; ++sidx)
  {
    if (shash[sidx] == 0xffffffffu) continue;
    probe_block<DEFER>(v, shash[sidx], sent[sidx], sbx[sidx], sby[sidx], sbz[sidx], contended);
  }

  // A segment of 2*trunc crosses a bounded number of blocks; the cap only
  // guarantees that every wave exits on NaN / degenerate input.
  for (int guard = 0; walking && guard < 4096; ++guard)
  {
    const uint32_t hash_code = block_hash(bx, by, bz, K);
    probe_block<DEFER>(v, hash_code, load_entry(v.hash_entries, hash_code), bx, by, bz, contended);

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
template <bool DEFER, bool PREP>
__global__ __launch_bounds__(256) void create_requests_kernel(RequestParams P, int* contended)
{
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);

  // PREP: the loads of the depth tile and of the pixel's colour and normal are issued first and
  // consumed after the request walk, which hides their latency
  constexpr int TW = 70, TH = 10, TS = 72;
  __shared__ float tile[PREP ? TH * TS : 1];
  float staged[3] = {0.0f, 0.0f, 0.0f};
  vf3 prep_rgb = {0.0f, 0.0f, 0.0f}, prep_n = {0.0f, 0.0f, 0.0f};
  if (PREP)
  {
    const int x0 = (int)blockIdx.x * 64 - 1, y0 = (int)blockIdx.y * 4 - 1;
#pragma unroll
    for (int t = 0; t < 3; ++t)
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
      prep_n = *reinterpret_cast<const vf3*>(P.normals + 3 * (size_t)index);
    }
  }

  // the fused SetView: later rounds (settle_visibility_kernel) are gated and counted through
  // these words, which nothing else touches while this kernel runs
  if (contended && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
  {
    P.v.counters[VK_CTR_GATE + 0] = 0;
    P.v.counters[VK_CTR_GATE + 1] = 0;
    P.v.counters[VK_CTR_BARRIER] = 0;
  }

  if (y < P.height) request_walk<DEFER>(P, x, y, contended);   // whole wave

  if (PREP)
  {
#pragma unroll
    for (int t = 0; t < 3; ++t)
    {
      const int i = (int)threadIdx.x + 256 * t;
      if (i < TW * TH) tile[(i / TW) * TS + (i % TW)] = staged[t];
    }
    __syncthreads();
    if (x < P.width && y < P.height)
    {
      const int index = y * P.width + x;
      const f3 Xcn = xform_dir(P.Tcd, make3(prep_n.x, prep_n.y, prep_n.z));      // light_integrator.cu:223
      float m = 0.0f;
      if (light_color_usable(prep_rgb.x, prep_rgb.y, prep_rgb.z))
        m = light_window_mask(tile, TS, (int)(threadIdx.x & 63) + 3, (int)(threadIdx.x >> 6) + 3, P.depth_threshold);
      P.mask[index] = m;
      P.records[index] = make_float4(Xcn.x, Xcn.y, Xcn.z, m);
    }
  }
}

// --------------------------------------------------------- handle requests ----

constexpr int kHandleThreads = 256;
constexpr int kHandlePerGroup = 4 * kHandleThreads;   // 1024 buckets per workgroup, one dword of flags per lane

// sums two ints over the workgroup; every lane gets the totals
__device__ __forceinline__ void block_sum2(int& a, int& b, int* lds)
{
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1)
  {
    a += __shfl_xor(a, d);
    b += __shfl_xor(b, d);
  }
  const int wave = threadIdx.x >> 6;
  if (lane_id() == 0) { lds[2 * wave] = a; lds[2 * wave + 1] = b; }
  __syncthreads();
  a = 0;
  b = 0;
#pragma unroll
  for (int w = 0; w < kHandleThreads / 64; ++w) { a += lds[2 * w]; b += lds[2 * w + 1]; }
  __syncthreads();
}

// byte != 0 <=> bit0 | bit1 ; byte == EXCESS(2) <=> bit1  (values are 0, 1, 2)
__device__ __forceinline__ void count_flags(uint32_t w, int& n_all, int& n_excess)
{
  n_all += __popc((w | (w >> 1)) & 0x01010101u);
  n_excess += __popc((w >> 1) & 0x01010101u);
}

// ref: volume.cu:304-368. Pool slots and excess indices must come out as if the
// reference's threads ran in ascending bucket order (reproducible allocation):
// request r (in bucket order) takes free slot voxel_pointer - r, EXCESS request
// e takes entry excess_pointer + e. Each 1024-bucket workgroup obtains its base
// ranks by counting the request flags of ALL buckets before it (a few KB of
// bytes, L2-resident) — redundant work instead of an inter-workgroup scan, so
// there is no cross-workgroup hand-off at all. Neither the counters nor the
// request flags are modified here (other workgroups still read them): the last
// workgroup publishes the totals in VK_CTR_PENDING_*, and finish_handle() —
// run by the kernel that follows — folds them in and clears the flags
// (volume.cu:365).
//
// `gate` (-1: none): the counter that tells the fused SetView whether another round is needed —
// set when this round lost a request to a bucket contest (VK_CTR_CONTENDED, left by the request
// pass and cleared here) or dropped one (upstream asks again, and drops again, on every call).
// One call handles the 1024 buckets of `group` with a 256-lane workgroup (uniform call).
__device__ __forceinline__ void handle_group(const vk_volume& v, int group, int groups, int zero_visible,
    int deferred_reset, int gate)
{
  __shared__ int red[2 * (kHandleThreads / 64)];
  __shared__ int wave_a[kHandleThreads / 64], wave_b[kHandleThreads / 64];

  const int count = v.main_block_count;
  const int max_count = v.main_block_count + v.excess_block_count;
  const int voxel_ptr0 = v.counters[VK_CTR_VOXEL_PTR];
  const int excess_ptr0 = v.counters[VK_CTR_EXCESS_PTR];
  const int first = group * kHandlePerGroup;

  // (1) requests in buckets [0, first): 16 bytes per load, first is a multiple of 1024
  int base_all = 0, base_excess = 0;
  const uint4* flags16 = reinterpret_cast<const uint4*>(v.allocation_types);
  // up to 16 independent 16-byte loads per lane: issued eight at a time so that
  // their latencies overlap (the loop is otherwise one L2 round trip per trip)
  const int chunks = first / 16;
  int j = threadIdx.x;
  for (; j + 7 * kHandleThreads < chunks; j += 8 * kHandleThreads)
  {
    uint4 q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) q[u] = flags16[j + u * kHandleThreads];
#pragma unroll
    for (int u = 0; u < 8; ++u)
    {
      count_flags(q[u].x, base_all, base_excess);
      count_flags(q[u].y, base_all, base_excess);
      count_flags(q[u].z, base_all, base_excess);
      count_flags(q[u].w, base_all, base_excess);
    }
  }
  for (; j < chunks; j += kHandleThreads)
  {
    const uint4 q = flags16[j];
    count_flags(q.x, base_all, base_excess);
    count_flags(q.y, base_all, base_excess);
    count_flags(q.z, base_all, base_excess);
    count_flags(q.w, base_all, base_excess);
  }
  block_sum2(base_all, base_excess, red);

  // (2) this lane's four buckets
  const int bucket0 = first + 4 * threadIdx.x;
  uint32_t w = 0;
  if (bucket0 + 4 <= count)
    w = reinterpret_cast<const uint32_t*>(v.allocation_types)[bucket0 >> 2];
  else
    for (int b = 0; b < count - bucket0; ++b) w |= (uint32_t)v.allocation_types[bucket0 + b] << (8 * b);

  int n_all = 0, n_excess = 0;
  count_flags(w, n_all, n_excess);

  // exclusive scan over the workgroup (wave shuffle scan + 4 wave totals)
  int incl_a = n_all, incl_b = n_excess;
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1)
  {
    const int ta = __shfl_up(incl_a, d);
    const int tb = __shfl_up(incl_b, d);
    if (lane >= d) { incl_a += ta; incl_b += tb; }
  }
  if (lane == 63) { wave_a[wave] = incl_a; wave_b[wave] = incl_b; }
  __syncthreads();
  int group_all = 0, group_excess = 0;
  int rank_all = base_all + incl_a - n_all, rank_excess = base_excess + incl_b - n_excess;
#pragma unroll
  for (int k = 0; k < kHandleThreads / 64; ++k)
  {
    if (k < wave) { rank_all += wave_a[k]; rank_excess += wave_b[k]; }
    group_all += wave_a[k];
    group_excess += wave_b[k];
  }

  // (3) commit this lane's requests in bucket order
  int dropped = 0;
  if (n_all > 0)
  {
    for (int k = 0; k < 4; ++k)
    {
      const int type = (w >> (8 * k)) & 0xff;
      if (type == VK_ALLOC_NONE) continue;

      const int index = bucket0 + k;
      const unsigned long long packed = reinterpret_cast<const unsigned long long*>(v.allocation_blocks)[index];
      int entry_index = index;

      if (type == VK_ALLOC_EXCESS)
      {
        int other_index = index;
        int next = v.hash_entries[other_index].next;
        for (int guard = 0; next != -1 && guard < max_count; ++guard)
        {
          other_index = next;
          next = v.hash_entries[other_index].next;
        }

        entry_index = excess_ptr0 + rank_excess;
        ++rank_excess;

        if (entry_index < max_count)
        {
          v.hash_entries[other_index].next = entry_index;
          v.block_visibility[entry_index] = deferred_reset ? (uint8_t)(VK_VISIBILITY_TRUE | kTouched) : (uint8_t)VK_VISIBILITY_TRUE;
        }
      }

      const int voxel_index = voxel_ptr0 - rank_all;
      ++rank_all;

      if (entry_index < max_count && voxel_index >= 0)
      {
        // entry = {block (pad cleared), data = free slot, next = -1}
        const int lo = (int)(packed & 0xffffffffull);
        const int hi = (int)((packed >> 32) & 0xffffull);
        reinterpret_cast<int4*>(v.hash_entries)[entry_index] =
            make_int4(lo, hi, v.free_voxel_blocks[voxel_index], -1);
      }
      else
      {
        ++dropped;
      }
      // the request flag is cleared by finish_handle(), not here: later
      // workgroups are still counting the flags of these buckets
    }
  }

  if (dropped)
  {
    atomicAdd(&v.counters[VK_CTR_DROPPED], dropped);
    if (gate >= 0) __hip_atomic_store(&v.counters[gate], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

  if (group == groups - 1 && threadIdx.x == 0)
  {
    v.counters[VK_CTR_PENDING_ALL] = base_all + group_all;
    v.counters[VK_CTR_PENDING_EXCESS] = base_excess + group_excess;
    v.counters[VK_CTR_REQUESTS] = base_all + group_all;
    // volume.cu:488 ResetBufferSize for the visibility pass that follows in SetView
    if (zero_visible) v.counters[VK_CTR_VISIBLE] = 0;
    if (gate >= 0)
    {
      if (__hip_atomic_load(&v.counters[VK_CTR_CONTENDED], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
      {
        __hip_atomic_store(&v.counters[gate], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&v.counters[VK_CTR_CONTENDED], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  __syncthreads();   // the LDS words above are reused by the next call of this workgroup
}

__global__ __launch_bounds__(kHandleThreads) void handle_requests_kernel(vk_volume v, int zero_visible, int deferred_reset,
    int gate)
{
  handle_group(v, (int)blockIdx.x, (int)gridDim.x, zero_visible, deferred_reset, gate);
}

// Second half of the handle pass, one lane per main bucket: clear the request
// (volume.cu:365) and, once, apply the pointer updates the reference does with
// atomics (volume.cu:337,352).
__device__ __forceinline__ void finish_handle(const vk_volume& v, int index)
{
  if (index < v.main_block_count && v.allocation_types[index] != VK_ALLOC_NONE)
  {
    v.allocation_types[index] = VK_ALLOC_NONE;
    reinterpret_cast<unsigned long long*>(v.allocation_blocks)[index] = 0ull;
  }

  if (index == 0)
  {
    const int all = v.counters[VK_CTR_PENDING_ALL];
    const int excess = v.counters[VK_CTR_PENDING_EXCESS];
    if (all | excess)
    {
      v.counters[VK_CTR_VOXEL_PTR] -= all;
      v.counters[VK_CTR_EXCESS_PTR] += excess;
      v.counters[VK_CTR_PENDING_ALL] = 0;
      v.counters[VK_CTR_PENDING_EXCESS] = 0;
    }
  }
}

__global__ __launch_bounds__(256) void finish_handle_kernel(vk_volume v)
{
  finish_handle(v, blockIdx.x * blockDim.x + threadIdx.x);
}

// -------------------------------------------------------- update visibility ----

struct VisibilityParams
{
  vk_volume v;
  int finish_handle;   // SetView: this kernel also completes the handle pass before it
  int deferred_reset;  // SetView: the reset pass was skipped, bytes may carry kTouched
  int width, height;
  vk_projection k;
  Rt Tdw;
};

constexpr int kVisThreads = 1024;

// ref: volume.cu:25-84. Frustum test of UNKNOWN entries + compaction of the
// visible ones. Compaction: wave ballot -> per-wave count in LDS -> ONE atomic
// per workgroup-sized chunk (72 atomics for 73 216 entries with 1024 lanes, instead of the
// reference's 10-step LDS scan + atomic per 512). One call takes the THREADS entries
// that start at `first` (uniform call).
template <int THREADS>
__device__ __forceinline__ void visibility_chunk(const VisibilityParams& P, int first, int finish, int deferred_reset)
{
  __shared__ int wave_count[THREADS / 64];
  __shared__ int block_base;

  const vk_volume& v = P.v;
  const int count = v.main_block_count + v.excess_block_count;
  const int index = first + (int)threadIdx.x;
  const float block_length = VK_BLOCK_RESOLUTION * v.voxel_length;
  bool visible = false;

  if (finish) finish_handle(v, index);

  if (index < count)
  {
    const int stored = v.block_visibility[index];
    int visibility = stored;
    if (deferred_reset)
    {
      // touched this frame -> TRUE; otherwise what the reset pass would have left
      const int before = stored & 3;
      visibility = (stored & kTouched) ? VK_VISIBILITY_TRUE
                 : (before == VK_VISIBILITY_TRUE ? VK_VISIBILITY_UNKNOWN : before);
    }
    visible = (visibility == VK_VISIBILITY_TRUE);
    int result = visibility;

    if (visibility == VK_VISIBILITY_UNKNOWN)
    {
      const Entry e = load_entry(v.hash_entries, index);

      for (int i = 0; i < 8; ++i)
      {
        f3 Xwp;
        Xwp.x = block_length * (e.ox + ((i & 1) >> 0));
        Xwp.y = block_length * (e.oy + ((i & 2) >> 1));
        Xwp.z = block_length * (e.oz + ((i & 4) >> 2));
        const f3 Xdp = xform_point(P.Tdw, Xwp);
        if (Xdp.z < 0) continue;

        float u, w;
        project(P.k, Xdp, u, w);

        if (u >= 0 && u <= P.width && w >= 0 && w <= P.height)
        {
          visible = true;
          break;
        }
      }

      if (!visible) result = VK_VISIBILITY_FALSE;
    }

    if (result != stored) v.block_visibility[index] = (uint8_t)result;
  }

  const unsigned long long mask = __ballot(visible);
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  if (lane == 0) wave_count[wave] = __popcll(mask);
  __syncthreads();

  if (threadIdx.x == 0)
  {
    int total = 0;
    for (int w = 0; w < THREADS / 64; ++w)
    {
      const int c = wave_count[w];
      wave_count[w] = total;
      total += c;
    }
    block_base = (total > 0) ? atomicAdd(&v.counters[VK_CTR_VISIBLE], total) : 0;
  }
  __syncthreads();

  if (visible)
  {
    const int offset = block_base + wave_count[wave] + __popcll(mask & ((1ull << lane) - 1ull));
    v.visible_blocks[offset] = index;
  }
  __syncthreads();   // the LDS words are reused by the next call of this workgroup
}

__global__ __launch_bounds__(kVisThreads) void update_visibility_kernel(VisibilityParams P)
{
  visibility_chunk<kVisThreads>(P, (int)blockIdx.x * kVisThreads, P.finish_handle, P.deferred_reset);
}

// ------------------------------------------------ SetView, several rounds in one call ----

// The reference's frame loop calls SetView three times per frame (apps/vulcan/vulcan.cu:316-318)
// because a bucket takes one request per call: a block that loses the contest for its bucket
// has to ask again. That is rare — a handful of new blocks per frame into 65 024 buckets — so
// the later calls almost always find nothing to do, and as launches of their own they would
// cost three launch floors each (~3 us per launch that does nothing on this part).
//
// vk_volume_set_view_rounds(.., max_rounds) gives the state of `max_rounds` consecutive SetView
// calls with the same frame, exactly (every buffer and counter), in three launches: the last
// kernel of the first round — this one: finish the handle pass, visibility test, compaction —
// looks at the round's gate word (handle_group) and ends unless a request was lost or dropped.
// Only then does it run whole further SetViews inside the launch (reset, requests, handle,
// visibility), its workgroups separated by grid-wide barriers. The grid is sized so that all
// of its workgroups are resident (settle_grid); a barrier nevertheless gives up after 2 s.
struct SettleParams
{
  VisibilityParams vis;
  RequestParams req;     // PREP fields unused
  int max_rounds;
};

constexpr int kSettleThreads = 256;
constexpr unsigned long long kBarrierTimeout = 200000000ull;   // wall_clock64 ticks (100 MHz): 2 s

// All workgroups of the launch meet here for the `nth` time (1, 2, ...; VK_CTR_BARRIER was
// zeroed by the request kernel of the same SetView). Agent-scope fences on both sides: what
// other workgroups — on other XCDs, behind other L2s — wrote before is visible after.
__device__ __forceinline__ bool grid_barrier(int* counter, int nth)
{
  __shared__ int ok;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0)
  {
    const int target = nth * (int)gridDim.x;
    __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long deadline = (unsigned long long)wall_clock64() + kBarrierTimeout;
    int good = 1;
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target)
    {
      if ((unsigned long long)wall_clock64() > deadline) { good = 0; break; }
      __builtin_amdgcn_s_sleep(8);
    }
    ok = good;
  }
  __syncthreads();
  const bool result = ok != 0;
  __threadfence();
  __syncthreads();
  return result;
}

__global__ __launch_bounds__(kSettleThreads) void settle_visibility_kernel(SettleParams P)
{
  __shared__ int gate_lds;
  const vk_volume& v = P.vis.v;
  const int max_count = v.main_block_count + v.excess_block_count;
  const int chunks = (max_count + kSettleThreads - 1) / kSettleThreads;
  const int groups = (v.main_block_count + kHandlePerGroup - 1) / kHandlePerGroup;
  const int lane = lane_id();
  int barriers = 0;

  for (int round = 1;; ++round)
  {
    // the last stage of SetView number `round`: the first one ran its other stages as launches
    // of their own (with the reset pass folded in, kTouched), the later ones ran them below
    for (int c = (int)blockIdx.x; c < chunks; c += (int)gridDim.x)
      visibility_chunk<kSettleThreads>(P.vis, c * kSettleThreads, 1, round == 1 ? P.vis.deferred_reset : 0);

    if (threadIdx.x == 0)
      gate_lds = round < P.max_rounds
          ? __hip_atomic_load(&v.counters[VK_CTR_GATE + (round & 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    __syncthreads();
    const int again = gate_lds;
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0 && !again)
    {
      v.counters[VK_CTR_ROUNDS] += round;
      // a round that is not run would have seen no request (volume.cu:520-535 on an empty list)
      if (round < P.max_rounds) v.counters[VK_CTR_REQUESTS] = 0;
      // what the caller may want to know: did the last round leave a request unanswered?
      v.counters[VK_CTR_UNSETTLED] =
          __hip_atomic_load(&v.counters[VK_CTR_GATE + (round & 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!again) return;

    // ---- another whole SetView (volume.cu:430-437), inside this launch
    if (!grid_barrier(&v.counters[VK_CTR_BARRIER], ++barriers)) break;

    // ResetBlockVisibility (volume.cu:465-471) + ResetBufferSize (:488)
    {
      const int words = max_count >> 2;
      uint32_t* vis32 = reinterpret_cast<uint32_t*>(v.block_visibility);
      for (int i = (int)(blockIdx.x * kSettleThreads + threadIdx.x); i < words; i += (int)(gridDim.x * kSettleThreads))
      {
        const uint32_t w = vis32[i];
        const uint32_t is_true = (w >> 1) & 0x01010101u;
        if (is_true) vis32[i] = w & ~(is_true * 0x3u);
      }
      if (blockIdx.x == 0 && (int)threadIdx.x < (max_count & 3))
      {
        const int j = (words << 2) + (int)threadIdx.x;
        if (v.block_visibility[j] == VK_VISIBILITY_TRUE) v.block_visibility[j] = VK_VISIBILITY_UNKNOWN;
      }
      if (blockIdx.x == 0 && threadIdx.x == 0)
      {
        v.counters[VK_CTR_VISIBLE] = 0;
        __hip_atomic_store(&v.counters[VK_CTR_GATE + ((round + 1) & 1)], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (!grid_barrier(&v.counters[VK_CTR_BARRIER], ++barriers)) break;

    // CreateAllocationRequests (volume.cu:497-518): a wave per 64-pixel run of a row
    {
      const int runs_x = (P.req.width + 63) / 64;
      const int runs = runs_x * P.req.height;
      const int waves = (int)gridDim.x * (kSettleThreads / 64);
      for (int run = (int)blockIdx.x * (kSettleThreads / 64) + (int)(threadIdx.x >> 6); run < runs; run += waves)
      {
        const int y = run / runs_x;
        const int x = (run - y * runs_x) * 64 + lane;
        request_walk<false>(P.req, x, y, &v.counters[VK_CTR_CONTENDED]);
      }
    }
    if (!grid_barrier(&v.counters[VK_CTR_BARRIER], ++barriers)) break;

    // HandleAllocationRequests (volume.cu:520-535)
    for (int g = (int)blockIdx.x; g < groups; g += (int)gridDim.x)
      handle_group(v, g, groups, 0, 0, VK_CTR_GATE + ((round + 1) & 1));
    if (!grid_barrier(&v.counters[VK_CTR_BARRIER], ++barriers)) break;
  }

  // only reached from a barrier that timed out (never seen): say so instead of spinning for ever
  if (threadIdx.x == 0) v.counters[VK_CTR_UNSETTLED] = -1;
}

// workgroups of settle_visibility_kernel that are resident at the same time on this device
int settle_grid(int chunks)
{
  static int capacity[16] = {0};
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= 16) return 0;
  int cap = __atomic_load_n(&capacity[device], __ATOMIC_ACQUIRE);
  if (cap == 0)
  {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, settle_visibility_kernel, kSettleThreads, 0) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess)
      return 0;
    cap = per_cu * cus;
    if (cap <= 0) return 0;
    __atomic_store_n(&capacity[device], cap, __ATOMIC_RELEASE);
  }
  // test aid: a small grid walks the same code with many chunks / groups / runs per workgroup
  if (const char* env = getenv("VK_SETTLE_GRID_CAP"))
  {
    const int n = atoi(env);
    if (n > 0 && n < cap) cap = n;
  }
  // half of what fits: a barrier must not depend on the last free slot of a CU
  if (cap > 2) cap /= 2;
  return chunks < cap ? chunks : cap;
}

int check_volume(const vk_volume* v)
{
  if (!v) return VK_ERR_ARGUMENT;
  if (!v->voxels || !v->hash_entries || !v->free_voxel_blocks || !v->allocation_types ||
      !v->allocation_blocks || !v->block_visibility || !v->visible_blocks || !v->counters)
    return VK_ERR_ARGUMENT;
  if (v->main_block_count <= 0 || v->excess_block_count < 0) return VK_ERR_ARGUMENT;
  if (!(v->voxel_length > 0) || !(v->truncation_length > 0)) return VK_ERR_ARGUMENT;
  if ((reinterpret_cast<uintptr_t>(v->allocation_blocks) & 7) || (reinterpret_cast<uintptr_t>(v->hash_entries) & 15) ||
      (reinterpret_cast<uintptr_t>(v->voxels) & 15) || (reinterpret_cast<uintptr_t>(v->block_visibility) & 3) ||
      (reinterpret_cast<uintptr_t>(v->allocation_types) & 15))
    return VK_ERR_ARGUMENT;
  return VK_OK;
}

int launch_create_requests(const vk_volume* v, const float* depth, int width, int height,
    const vk_projection* projection, const vk_transform* Twd, bool deferred_reset, hipStream_t s,
    const vk_frame* prep_frame = nullptr, const vk_light_prep* prep = nullptr, bool fused = false,
    RequestParams* params_out = nullptr)
{
  RequestParams P;
  P.v = *v;
  P.depth = depth;
  P.width = width;
  P.height = height;
  P.k = *projection;
  P.Twd = make_rt(Twd->m);
  P.colors = P.normals = nullptr;
  P.Tcd = P.Twd;
  P.depth_threshold = 0.0f;
  P.mask = nullptr;
  P.records = nullptr;
  const dim3 grid((width + 63) / 64, (height + 3) / 4);
  // the fused SetView records bucket contests (post_request) for its later rounds
  int* contended = fused ? v->counters + VK_CTR_CONTENDED : nullptr;
  if (params_out) *params_out = P;
  if (prep && prep_frame)
  {
    P.colors = prep_frame->color;
    P.normals = prep_frame->normals;
    P.Tcd = make_rt(prep_frame->depth_to_color.m);
    P.depth_threshold = prep->depth_threshold;
    P.mask = prep->mask;
    P.records = reinterpret_cast<float4*>(prep->records);
    hipLaunchKernelGGL((create_requests_kernel<true, true>), grid, dim3(256), 0, s, P, contended);
  }
  else if (deferred_reset) hipLaunchKernelGGL((create_requests_kernel<true, false>), grid, dim3(256), 0, s, P, contended);
  else hipLaunchKernelGGL((create_requests_kernel<false, false>), grid, dim3(256), 0, s, P, contended);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

int launch_update_visibility(const vk_volume* v, int width, int height,
    const vk_projection* projection, const float* Tdw_m, bool finish, bool deferred_reset, hipStream_t s)
{
  VisibilityParams P;
  P.v = *v;
  P.finish_handle = finish ? 1 : 0;
  P.deferred_reset = deferred_reset ? 1 : 0;
  P.width = width;
  P.height = height;
  P.k = *projection;
  P.Tdw = make_rt(Tdw_m);
  const int count = v->main_block_count + v->excess_block_count;
  hipLaunchKernelGGL(update_visibility_kernel, dim3((count + kVisThreads - 1) / kVisThreads),
      dim3(kVisThreads), 0, s, P);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

int launch_reset_visibility(const vk_volume* v, hipStream_t s)
{
  const int count = v->main_block_count + v->excess_block_count;
  const int threads = (count >> 2) > 4 ? (count >> 2) : 4;
  hipLaunchKernelGGL(reset_visibility_kernel, dim3((threads + 255) / 256), dim3(256), 0, s,
      v->block_visibility, count);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

}  // namespace

extern "C" {

int vk_volume_initialize(const vk_volume* v, void* stream)
{
  const int rc = check_volume(v);
  if (rc != VK_OK) return rc;
  const int max_count = v->main_block_count + v->excess_block_count;
  const size_t n4 = (size_t)max_count * VK_BLOCK_VOXELS * sizeof(vk_voxel) / 16;
  const int blocks = (int)((n4 + 255) / 256 < (size_t)(kCUs * 16) ? (n4 + 255) / 256 : (size_t)(kCUs * 16));
  hipLaunchKernelGGL(fill_voxels_kernel, dim3(blocks), dim3(256), 0, vk_s(stream),
      reinterpret_cast<float4*>(v->voxels), n4);
  VK_LAUNCH_CHECK();
  hipLaunchKernelGGL(init_tables_kernel, dim3((max_count + 255) / 256), dim3(256), 0, vk_s(stream), *v);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

int vk_volume_reset_block_visibility(const vk_volume* v, void* stream)
{
  const int rc = check_volume(v);
  if (rc != VK_OK) return rc;
  return launch_reset_visibility(v, vk_s(stream));
}

int vk_volume_create_allocation_requests(const vk_volume* v, const float* depth, int width,
    int height, const vk_projection* projection, const vk_transform* Twd, void* stream)
{
  const int rc = check_volume(v);
  if (rc != VK_OK) return rc;
  VK_REQUIRE(depth && projection && Twd && width > 0 && height > 0);
  return launch_create_requests(v, depth, width, height, projection, Twd, false, vk_s(stream));
}

int vk_volume_handle_allocation_requests(const vk_volume* v, void* stream)
{
  const int rc = check_volume(v);
  if (rc != VK_OK) return rc;
  hipLaunchKernelGGL(handle_requests_kernel, dim3((v->main_block_count + kHandlePerGroup - 1) / kHandlePerGroup),
      dim3(kHandleThreads), 0, vk_s(stream), *v, 0, 0, -1);
  VK_LAUNCH_CHECK();
  hipLaunchKernelGGL(finish_handle_kernel, dim3((v->main_block_count + 255) / 256), dim3(256), 0, vk_s(stream), *v);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

int vk_volume_update_block_visibility(const vk_volume* v, int width, int height,
    const vk_projection* projection, const vk_transform* Tdw, void* stream)
{
  const int rc = check_volume(v);
  if (rc != VK_OK) return rc;
  VK_REQUIRE(projection && Tdw && width > 0 && height > 0);
  // volume.cu:488 ResetBufferSize
  VK_CHECK(hipMemsetAsync(v->counters + VK_CTR_VISIBLE, 0, sizeof(int32_t), vk_s(stream)));
  return launch_update_visibility(v, width, height, projection, Tdw->m, false, false, vk_s(stream));
}

// does `prep` hold the preparation of exactly this frame (same images, size, threshold, Tcd)?
static bool prep_is_for(const vk_light_prep* prep, const vk_frame* frame)
{
  return prep && prep->valid && frame->content_id != 0 && prep->content_id == frame->content_id &&
      prep->depth == frame->depth && prep->color == frame->color &&
      prep->normals == frame->normals && prep->width == frame->width && prep->height == frame->height &&
      memcmp(&prep->depth_to_color, &frame->depth_to_color, sizeof(vk_transform)) == 0;
}

int vk_light_prepared(const vk_light_prep* prep, const vk_frame* frame, float depth_threshold)
{
  return (frame && prep_is_for(prep, frame) && prep->prepared_threshold == depth_threshold) ? 1 : 0;
}

static int set_view(const vk_volume* v, const vk_frame* frame, vk_light_prep* prep, int max_rounds, void* stream)
{
  const int rc = check_volume(v);
  if (rc != VK_OK) return rc;
  VK_REQUIRE(frame && frame->depth && frame->width > 0 && frame->height > 0 && max_rounds >= 1);
  hipStream_t s = vk_s(stream);
  // the preparation rides along when the frame has what LightIntegrator needs, in the
  // depth image's size (light_integrator.cu:277-293 walks the colour image with it)
  const bool ride = prep && prep->mask && prep->records && frame->color && frame->normals && frame->content_id != 0 &&
      (long long)frame->width * frame->height <= (long long)prep->capacity &&
      (reinterpret_cast<uintptr_t>(prep->records) & 15) == 0 &&
      (frame->color_width <= 0 || frame->color_width == frame->width) &&
      (frame->color_height <= 0 || frame->color_height == frame->height);
  if (prep) prep->valid = 0;
  const int max_count = v->main_block_count + v->excess_block_count;
  const int grid = settle_grid((max_count + kSettleThreads - 1) / kSettleThreads);
  VK_REQUIRE(grid > 0);
  // three launches: the reset pass is folded into the other three (see kTouched), and so are
  // all rounds after the first (settle_visibility_kernel)
  int r;
  SettleParams S;
  if ((r = launch_create_requests(v, frame->depth, frame->width, frame->height,
           &frame->depth_projection, &frame->depth_to_world, true, s, ride ? frame : nullptr, ride ? prep : nullptr,
           true, &S.req)) != VK_OK) return r;
  if (ride)
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
  }
  hipLaunchKernelGGL(handle_requests_kernel, dim3((v->main_block_count + kHandlePerGroup - 1) / kHandlePerGroup),
      dim3(kHandleThreads), 0, s, *v, 1, 1, VK_CTR_GATE + 1);
  VK_LAUNCH_CHECK();  // also zeroes counters[VK_CTR_VISIBLE]; pointers are folded in by the next kernel
  S.vis.v = *v;
  S.vis.finish_handle = 1;
  S.vis.deferred_reset = 1;
  S.vis.width = frame->width;
  S.vis.height = frame->height;
  S.vis.k = frame->depth_projection;
  S.vis.Tdw = make_rt(frame->depth_to_world.inv);
  S.max_rounds = max_rounds;
  hipLaunchKernelGGL(settle_visibility_kernel, dim3(grid), dim3(kSettleThreads), 0, s, S);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

int vk_volume_set_view(const vk_volume* v, const vk_frame* frame, void* stream)
{
  return set_view(v, frame, nullptr, 1, stream);
}

int vk_volume_set_view_prepare(const vk_volume* v, const vk_frame* frame, vk_light_prep* prep, void* stream)
{
  return set_view(v, frame, prep, 1, stream);
}

int vk_volume_set_view_rounds(const vk_volume* v, const vk_frame* frame, vk_light_prep* prep, int max_rounds,
    void* stream)
{
  return set_view(v, frame, prep, max_rounds, stream);
}

int vk_volume_read_counters_sync(const vk_volume* v, int32_t* host_out, void* stream)
{
  VK_REQUIRE(v && v->counters && host_out);
  VK_CHECK(hipMemcpyAsync(host_out, v->counters, sizeof(int32_t) * VK_CTR_COUNT,
      hipMemcpyDeviceToHost, vk_s(stream)));
  VK_CHECK(hipStreamSynchronize(vk_s(stream)));
  return VK_OK;
}

}  // extern "C"
