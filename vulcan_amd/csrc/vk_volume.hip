// vk_volume.hip — hashed voxel volume: initialisation, block allocation and
// visible-block compaction for gfx950 (ref: src/volume.cu).
//
// Per SetView the reference issues thrust::replace + 3 kernels and reads the
// visible count back to the host (volume.cu:494). Here the count stays in
// v.counters[VK_CTR_VISIBLE]; every consumer reads it on the device.
#include "vk_common.hpp"
#include "vk_requests.hpp"

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

  if (i < VK_CTR_PUBLIC)
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
// (the pass's device code: vk_requests.hpp)

// (five waves per SIMD: the 4 800 waves of a 640 x 480 frame are 4.7 per SIMD and must be resident in one generation; the
// allocator lands on 96 or 97 VGPRs for the variant that also computes the normals, and 97 would cost the fifth wave)
template <bool DEFER, int PREP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void create_requests_kernel(RequestParams P, Retry retry)
{
  requests_group<DEFER, PREP>(P, retry, (int)blockIdx.x, (int)blockIdx.y);
}

// The same pass at a pose that is still ON THE DEVICE when the launch is enqueued (vk_volume_requests_at_device_pose, round 6):
// the tracked pose a Gauss-Newton loop launch in front of it in the stream leaves in *pose. Twelve uniform loads; the rest
// is create_requests_kernel<true, PREP> to the instruction (make_rt's transposition of the column-major matrix).
template <int PREP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void create_requests_at_kernel(RequestParams P, Retry retry,
    const vk_transform* __restrict__ pose)
{
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) P.Twd.r[r * 4 + c] = pose->m[c * 4 + r];
  requests_group<true, PREP>(P, retry, (int)blockIdx.x, (int)blockIdx.y);
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
// `dropped_now` (optional): set when a request of this pass is dropped (pool or excess list
// exhausted). One call handles the 1024 buckets of `group` with a 256-lane workgroup (uniform call).
__device__ __forceinline__ void handle_group(const vk_volume& v, int group, int groups, int zero_visible,
    int deferred_reset, int* dropped_now)
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
          // (2: the fused handle + visibility launch marks the round's new excess entries by their
          // index range, visibility_chunk; a byte written here could land after that pass's)
          if (deferred_reset != 2)
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
    if (dropped_now)
    {
      __hip_atomic_store(dropped_now, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence();   // rare; read by the workgroup that finishes the round in the same launch
    }
  }

  // (2: the workgroup that finishes the round counts the requests itself, handle_visibility_kernel)
  if (group == groups - 1 && threadIdx.x == 0 && deferred_reset != 2)
  {
    v.counters[VK_CTR_PENDING_ALL] = base_all + group_all;
    v.counters[VK_CTR_PENDING_EXCESS] = base_excess + group_excess;
    v.counters[VK_CTR_REQUESTS] = base_all + group_all;
    // volume.cu:488 ResetBufferSize for the visibility pass that follows in SetView
    if (zero_visible) v.counters[VK_CTR_VISIBLE] = 0;
  }
  __syncthreads();   // the LDS words above are reused by the next call of this workgroup
}

__global__ __launch_bounds__(kHandleThreads) void handle_requests_kernel(vk_volume v, int zero_visible, int deferred_reset)
{
  handle_group(v, (int)blockIdx.x, (int)gridDim.x, zero_visible, deferred_reset, nullptr);
}

// the origin block would be requested by another SetView round: its rays met it in an unallocated
// main entry (probe_block) and that entry has been given to another block since
__device__ __forceinline__ int origin_block_pending(const vk_volume& v)
{
  if (__hip_atomic_load(&v.counters[VK_CTR_ORIGIN_SEEN], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return 0;
  const int data = __hip_atomic_load(&v.hash_entries[block_hash(0, 0, 0, (uint32_t)v.main_block_count)].data,
      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return data != -1 ? 1 : 0;
}

// Second half of the handle pass, one lane per main bucket: clear the request
// (volume.cu:365) and, once, apply the pointer updates the reference does with
// atomics (volume.cu:337,352).
// `max_rounds` > 0 (the fused SetView only): also the end-of-call bookkeeping of
// vk_volume_set_view_rounds — VK_CTR_ROUNDS, VK_CTR_UNSETTLED, VK_CTR_REQUESTS of rounds that were
// not needed — and the reset of the words the next call's request pass records into.
__device__ __forceinline__ void finish_handle(const vk_volume& v, int index, int max_rounds = 0)
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
    if (max_rounds > 0)
    {
      // what is still unanswered: a request lost to a bucket contest (after the last round that
      // ran) or dropped, or more losers than the retry list holds
      const int unsettled = (v.counters[VK_CTR_CONTENDED] | v.counters[VK_CTR_DROPPED_NOW] | v.counters[VK_CTR_RETRY_OVERFLOW] |
          origin_block_pending(v)) != 0;
      v.counters[VK_CTR_UNSETTLED] = unsettled;
      v.counters[VK_CTR_ROUNDS] += 1;            // the first round; later_rounds() has added the others
      // a round that is not run because nothing was pending would have seen no request
      if (max_rounds > 1 && !unsettled && v.counters[VK_CTR_TICKET] == 0) v.counters[VK_CTR_REQUESTS] = 0;
      v.counters[VK_CTR_CONTENDED] = 0;
      v.counters[VK_CTR_DROPPED_NOW] = 0;
      v.counters[VK_CTR_RETRY_OVERFLOW] = 0;
      v.counters[VK_CTR_ORIGIN_SEEN] = 0;
      v.counters[VK_CTR_TICKET] = 0;
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
  int finish_handle;   // SetView: this kernel also completes the handle pass before it; the fused
                       // SetView passes its max_rounds here (>= 1)
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
// `new_lo`, `new_hi` (the fused handle + visibility launch): entries [new_lo, new_hi) are the excess
// entries this round's handle pass creates; they are visible (volume.cu:344) whatever their byte holds.
template <int THREADS>
__device__ __forceinline__ void visibility_chunk(const VisibilityParams& P, int first, int finish, int deferred_reset,
    int new_lo = 0, int new_hi = 0)
{
  __shared__ int wave_count[THREADS / 64];
  __shared__ int block_base;

  const vk_volume& v = P.v;
  const int count = v.main_block_count + v.excess_block_count;
  const int index = first + (int)threadIdx.x;
  const float block_length = VK_BLOCK_RESOLUTION * v.voxel_length;
  bool visible = false;

  if (finish) finish_handle(v, index, finish);

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
    if (index >= new_lo && index < new_hi) visibility = VK_VISIBILITY_TRUE;
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
  if (blockIdx.x == 0 && threadIdx.x == 0) P.v.counters[VK_CTR_BANDED] = -1;   // this pass writes the plain list only
  if (P.finish_handle && blockIdx.x == 0)
  {
    // the fused SetView: losers that no later round consumed (max_rounds reached, or the rounds
    // were cut short) leave the sets, so that the next call's request pass starts from empty ones
    for (int which = 0; which < 2; ++which)
    {
      int n = P.v.counters[VK_CTR_RETRY_COUNT + which];
      if (n > VK_RETRY_KEYS) n = VK_RETRY_KEYS;
      unsigned long long* table = retry_table(P.v.counters, which);
      const int* slots = retry_slots(P.v.counters, which);
      for (int i = (int)threadIdx.x; i < n; i += kVisThreads) table[slots[i]] = 0ull;
    }
    __syncthreads();
    if (threadIdx.x < 2) P.v.counters[VK_CTR_RETRY_COUNT + threadIdx.x] = 0;
  }
}

// ------------------------------------------------ SetView, several rounds in one call ----

// The reference's frame loop calls SetView three times per frame (apps/vulcan/vulcan.cu:316-318)
// because a bucket takes one request per call: a block that loses the contest for its bucket
// has to ask again. As launches of their own the later calls would cost three launch floors
// each (~3 us per launch that does nothing on this part) and walk all the rays again although
// only the losers have anything to ask.
//
// vk_volume_set_view_rounds(.., max_rounds) leaves the state of `max_rounds` consecutive SetView
// calls with the same frame in the launches of one. What another SetView call with the same
// frame changes is exactly this: every block that lost asks again (its rays walk the same blocks
// as before, find everything else in the table, and mark visible what is already marked), the
// winners are committed, and the entries so created become visible. So the request pass files the
// losers in a list (Retry), and when there are any the workgroup of the handle pass that finishes
// LAST plays the later rounds from that list by itself, before the visibility pass runs: probe
// (volume.cu:183-239), commit in bucket order (:304-368), until a round loses nothing or
// max_rounds is reached. Without losers — almost every frame — none of this runs, and no
// workgroup pays for the arrival count either.
//
// The rounds end early (VK_CTR_UNSETTLED says so) with a round that drops a request (pool or
// excess list exhausted: upstream's own state is inconsistent from there on, its later calls
// link entries it never writes) and when more blocks lose in one round than the list holds.
constexpr int kMaxPosted = 4096;   // buckets that can receive a request in one later round


// volume.cu:365 for every bucket at once, by one 256-lane workgroup: the requests are cleared, and
// (optionally) counted — all of them, and the EXCESS ones. Sixteen flags per load (the array is
// 16-byte aligned, check_volume), the tail byte by byte. Ends with a barrier.
__device__ __forceinline__ void clear_requests(const vk_volume& v, int* total_all, int* total_excess)
{
  __shared__ int red[2 * (kHandleThreads / 64)];
  const int count = v.main_block_count;
  const uint4* flags16 = reinterpret_cast<const uint4*>(v.allocation_types);
  const int chunks = count / 16;
  int n_all = 0, n_excess = 0;
  // eight 16-byte loads in flight per lane (one workgroup walks the whole array: a load per trip
  // would be one L2 round trip per trip)
  for (int c0 = (int)threadIdx.x; c0 < chunks; c0 += 8 * kHandleThreads)
  {
    uint4 q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
    {
      const int c = c0 + u * kHandleThreads;
      q[u] = c < chunks ? flags16[c] : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
    {
      if ((q[u].x | q[u].y | q[u].z | q[u].w) == 0u) continue;
      const int c = c0 + u * kHandleThreads;
      count_flags(q[u].x, n_all, n_excess);
      count_flags(q[u].y, n_all, n_excess);
      count_flags(q[u].z, n_all, n_excess);
      count_flags(q[u].w, n_all, n_excess);
      const uint32_t words[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
#pragma unroll
      for (int w = 0; w < 4; ++w)
      {
        if (words[w] == 0u) continue;
        for (int b = 0; b < 4; ++b)
          if ((words[w] >> (8 * b)) & 0xffu)
            reinterpret_cast<unsigned long long*>(v.allocation_blocks)[c * 16 + w * 4 + b] = 0ull;
      }
      const_cast<uint4*>(flags16)[c] = make_uint4(0u, 0u, 0u, 0u);
    }
  }
  for (int i = chunks * 16 + (int)threadIdx.x; i < count; i += kHandleThreads)
  {
    const int type = v.allocation_types[i];
    if (type != VK_ALLOC_NONE)
    {
      n_all += 1;
      n_excess += type == VK_ALLOC_EXCESS ? 1 : 0;
      v.allocation_types[i] = VK_ALLOC_NONE;
      reinterpret_cast<unsigned long long*>(v.allocation_blocks)[i] = 0ull;
    }
  }
  block_sum2(n_all, n_excess, red);
  if (total_all) *total_all = n_all;
  if (total_excess) *total_excess = n_excess;
}

// FUSED: called from the handle + visibility launch, after the first round is complete in every
// respect (requests cleared, pointers moved, visible list built): entries are marked visible plainly
// and entered in the visible list as they appear.
// Returns (in lane 0) whether requests are still unanswered after the last round that ran.
template <bool FUSED>
__device__ __forceinline__ int later_rounds(const vk_volume& v, int max_rounds, int retry_capacity)
{
  __shared__ int posted[kMaxPosted];
  __shared__ uint8_t posted_type[kMaxPosted];
  __shared__ int posted_count, dropped_total, stop, origin;

  const int count = v.main_block_count;
  const int max_count = v.main_block_count + v.excess_block_count;

  // the first round's second half (finish_handle): requests cleared, pointers moved
  if (!FUSED) clear_requests(v, nullptr, nullptr);
  if (threadIdx.x == 0)
  {
    if (!FUSED)
    {
      v.counters[VK_CTR_VOXEL_PTR] -= v.counters[VK_CTR_PENDING_ALL];
      v.counters[VK_CTR_EXCESS_PTR] += v.counters[VK_CTR_PENDING_EXCESS];
      v.counters[VK_CTR_PENDING_ALL] = 0;
      v.counters[VK_CTR_PENDING_EXCESS] = 0;
    }
    stop = (v.counters[VK_CTR_DROPPED_NOW] | v.counters[VK_CTR_RETRY_OVERFLOW]) != 0;
  }
  __syncthreads();

  int current = 0, rounds_run = 1, last_posted = 0;
  while (rounds_run < max_rounds && !stop)
  {
    __threadfence();     // rare path: what the other lanes wrote in the phase before is read from L2
    __syncthreads();
    int listed = __hip_atomic_load(&v.counters[VK_CTR_RETRY_COUNT + current], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (listed > retry_capacity) listed = retry_capacity;   // the surplus raised VK_CTR_RETRY_OVERFLOW
    // the origin block (probe_block): its rays request it in this round if they met it in the
    // round before and its bucket's main entry has been taken since
    if (threadIdx.x == 0) origin = origin_block_pending(v);
    __syncthreads();
    const bool origin_asks = origin != 0;
    if (listed == 0 && !origin_asks) break;
    __syncthreads();
    if (threadIdx.x == 0)
    {
      posted_count = 0;
      dropped_total = 0;
      if (origin_asks) __hip_atomic_store(&v.counters[VK_CTR_ORIGIN_SEEN], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // atomic stores + fence: the lanes' atomic increments that follow must not be overtaken
      __hip_atomic_store(&v.counters[VK_CTR_RETRY_COUNT + (current ^ 1)], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&v.counters[VK_CTR_CONTENDED], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __threadfence();
    __syncthreads();

    // CreateAllocationRequests for the blocks that lost (volume.cu:183-239)
    Retry next;
    next.contended = v.counters + VK_CTR_CONTENDED;
    next.origin_seen = v.counters + VK_CTR_ORIGIN_SEEN;
    next.count = v.counters + VK_CTR_RETRY_COUNT + (current ^ 1);
    next.overflow = v.counters + VK_CTR_RETRY_OVERFLOW;
    next.table = retry_table(v.counters, current ^ 1);
    next.slots = retry_slots(v.counters, current ^ 1);
    next.capacity = retry_capacity;
    next.posted = posted;
    next.posted_tail = nullptr;
    next.posted_count = &posted_count;
    next.posted_capacity = kMaxPosted;
    unsigned long long* table = retry_table(v.counters, current);
    const int* slots = retry_slots(v.counters, current);
    for (int i = (int)threadIdx.x; i < listed; i += kHandleThreads)
    {
      const unsigned long long key = table[slots[i]];
      table[slots[i]] = 0ull;                       // the set is left empty for the call after this one
      const int bx = (int16_t)(key & 0xffffull), by = (int16_t)((key >> 16) & 0xffffull), bz = (int16_t)((key >> 32) & 0xffffull);
      const uint32_t h = block_hash(bx, by, bz, (uint32_t)count);
      probe_block<FUSED ? MARK_APPEND : MARK_DEFER>(v, h, load_entry(v.hash_entries, h), bx, by, bz, next);
    }
    if (origin_asks && threadIdx.x == 0)
    {
      const uint32_t h = block_hash(0, 0, 0, (uint32_t)count);
      probe_block<FUSED ? MARK_APPEND : MARK_DEFER>(v, h, load_entry(v.hash_entries, h), 0, 0, 0, next);
    }
    __threadfence();
    __syncthreads();
    const int m = posted_count;
    if (m > kMaxPosted || __hip_atomic_load(&v.counters[VK_CTR_RETRY_OVERFLOW], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    {
      // more than this workgroup can order: the requests stay posted for the next SetView call
      // (which has to find them in the flags: they are on no list)
      if (threadIdx.x == 0)
      {
        v.counters[VK_CTR_RETRY_OVERFLOW] = 1;
        stop = 1;
        if (FUSED) v.counters[VK_CTR_POSTED] = VK_POSTED_SLOTS + 1;
      }
      __syncthreads();
      break;
    }

    // HandleAllocationRequests (volume.cu:304-368) in ascending bucket order
    for (int i = (int)threadIdx.x; i < m; i += kHandleThreads)
    {
      const int bucket = (int)((uint32_t)posted[i] & kPostedBucket);
      posted[i] = bucket;
      posted_type[i] = v.allocation_types[bucket];
    }
    __syncthreads();
    const int voxel_ptr0 = v.counters[VK_CTR_VOXEL_PTR];
    const int excess_ptr0 = v.counters[VK_CTR_EXCESS_PTR];
    int m_excess = 0;
    for (int j = 0; j < m; ++j) m_excess += posted_type[j] == VK_ALLOC_EXCESS ? 1 : 0;
    int dropped = 0;
    for (int i = (int)threadIdx.x; i < m; i += kHandleThreads)
    {
      const int index = posted[i];
      const int type = posted_type[i];
      int rank_all = 0, rank_excess = 0;
      for (int j = 0; j < m; ++j)
      {
        const bool before = posted[j] < index;
        rank_all += before ? 1 : 0;
        rank_excess += (before && posted_type[j] == VK_ALLOC_EXCESS) ? 1 : 0;
      }
      const unsigned long long packed = reinterpret_cast<const unsigned long long*>(v.allocation_blocks)[index];
      int entry_index = index;
      if (type == VK_ALLOC_EXCESS)
      {
        int other_index = index;
        int next_index = v.hash_entries[other_index].next;
        for (int guard = 0; next_index != -1 && guard < max_count; ++guard)
        {
          other_index = next_index;
          next_index = v.hash_entries[other_index].next;
        }
        entry_index = excess_ptr0 + rank_excess;
        if (entry_index < max_count)
        {
          v.hash_entries[other_index].next = entry_index;
          if (FUSED)
          {
            // the visibility pass has run: the entry is visible (volume.cu:344) and listed from now on
            v.block_visibility[entry_index] = (uint8_t)VK_VISIBILITY_TRUE;
            v.visible_blocks[atomicAdd(&v.counters[VK_CTR_VISIBLE], 1)] = entry_index;
          }
          else v.block_visibility[entry_index] = (uint8_t)(VK_VISIBILITY_TRUE | kTouched);   // decoded by the visibility pass
        }
      }
      const int voxel_index = voxel_ptr0 - rank_all;
      if (entry_index < max_count && voxel_index >= 0)
      {
        const int lo = (int)(packed & 0xffffffffull);
        const int hi = (int)((packed >> 32) & 0xffffull);
        reinterpret_cast<int4*>(v.hash_entries)[entry_index] = make_int4(lo, hi, v.free_voxel_blocks[voxel_index], -1);
      }
      else
      {
        ++dropped;
      }
      v.allocation_types[index] = VK_ALLOC_NONE;                                          // volume.cu:365
      reinterpret_cast<unsigned long long*>(v.allocation_blocks)[index] = 0ull;
    }
    if (dropped) atomicAdd(&dropped_total, dropped);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0)
    {
      v.counters[VK_CTR_VOXEL_PTR] = voxel_ptr0 - m;
      v.counters[VK_CTR_EXCESS_PTR] = excess_ptr0 + m_excess;
      if (dropped_total)
      {
        v.counters[VK_CTR_DROPPED] += dropped_total;
        v.counters[VK_CTR_DROPPED_NOW] = 1;
        stop = 1;
      }
    }
    __syncthreads();
    if (threadIdx.x == 0)   // consumed
      __hip_atomic_store(&v.counters[VK_CTR_RETRY_COUNT + current], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_posted = m;
    ++rounds_run;
    current ^= 1;
  }

  int still_pending = 0;
  if (threadIdx.x == 0)
  {
    const bool pending = stop || origin_block_pending(v) ||
        __hip_atomic_load(&v.counters[VK_CTR_RETRY_COUNT + current], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    v.counters[VK_CTR_ROUNDS] += rounds_run - 1;
    v.counters[VK_CTR_CONTENDED] = pending ? 1 : 0;
    // a round that did not run because nothing was pending would have seen no request
    v.counters[VK_CTR_REQUESTS] = (rounds_run < max_rounds && !pending) ? 0 : (rounds_run > 1 ? last_posted : v.counters[VK_CTR_REQUESTS]);
    still_pending = pending ? 1 : 0;
  }
  return still_pending;
}

// The handle pass of the fused SetView. With losers on file (VK_CTR_CONTENDED, left by the request
// pass: the same answer in every workgroup) each workgroup announces its end — release, count,
// acquire: the only fences of the whole SetView, paid in the rare case alone — and the one that
// arrives last runs the later rounds.
__global__ __launch_bounds__(kHandleThreads) void handle_rounds_kernel(vk_volume v, int max_rounds, int retry_capacity)
{
  __shared__ int last;
  // (the origin block: met in an unallocated main entry that this round gives to another block)
  const bool losers = max_rounds > 1 && (v.counters[VK_CTR_CONTENDED] != 0 ||
      (v.counters[VK_CTR_ORIGIN_SEEN] != 0 && v.allocation_types[block_hash(0, 0, 0, (uint32_t)v.main_block_count)] != VK_ALLOC_NONE));
  handle_group(v, (int)blockIdx.x, (int)gridDim.x, 1, 1, v.counters + VK_CTR_DROPPED_NOW);
  if (!losers) return;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0)
    last = __hip_atomic_fetch_add(&v.counters[VK_CTR_TICKET], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
  __syncthreads();
  if (!last) return;
  __threadfence();
  later_rounds<false>(v, max_rounds, retry_capacity);
}

// ------------------------------------------ handle + visibility in one launch ----

// The handle pass and the visibility pass of a SetView do not depend on each other, with two
// exceptions: the excess entries the handle pass creates must come out visible, and the second
// half of the handle pass (requests cleared, pointers moved) has to wait for every reader of the
// request flags and pointers. Both have an answer that needs no hand-off between workgroups: the
// new entries are exactly the index range [excess pointer, excess pointer + number of EXCESS
// requests), which a visibility workgroup that overlaps the excess region counts for itself; and
// the workgroup that arrives LAST at a ticket (a relaxed atomic: nothing it does depends on what
// the others WROTE) clears the requests and moves the pointers. One launch (~5 us at this size)
// less per SetView.
//
// The handle pass itself works from the list of buckets the request pass posted to
// (VK_CTR_POSTED, posted_list): a frame of a moving camera asks for tens of blocks, and ranking
// those among themselves (request r in bucket order takes free slot voxel_pointer - r, handle_group)
// is nothing next to counting the flags of 65 024 buckets in every workgroup. A request pass that
// posts to more than VK_POSTED_SLOTS buckets (the first frame) is handled from the flags as before.
//
// With losers on file the last workgroup also replays the later rounds (later_rounds<true>); only
// then do the workgroups fence around the ticket.
struct FusedParams
{
  VisibilityParams vis;     // finish_handle / deferred_reset unused
  int handle_wgs;           // workgroups [0, handle_wgs): the handle pass; the rest: visibility chunks
  int max_rounds;
  int retry_capacity;
  int posted_capacity;      // entries the posted list holds (VK_POSTED_SLOTS; less as a test aid)
};

constexpr int kVisPerGroup = 4 * kHandleThreads;   // entries per visibility workgroup: four per lane

// The handle pass from the posted list: listed entries wg * 256 + lane, + wgs * 256, ... Each
// request's flags are cleared with it (volume.cu:365) — no other workgroup looks at them in this
// form — except those of `keep_bucket` (the origin block's bucket, read by every workgroup on
// arrival in the kernel). Returns (uniform) whether a request was dropped; *excess_total (workgroup
// 0 only): the EXCESS requests of the whole list.
__device__ __forceinline__ int handle_listed(const vk_volume& v, int m, int wg, int wgs, uint32_t keep_bucket,
    int voxel_ptr0, int excess_ptr0, int* excess_total)
{
  __shared__ __attribute__((aligned(16))) uint32_t listed[VK_POSTED_SLOTS];
  __shared__ int free_slot[VK_POSTED_SLOTS];
  __shared__ int red[2 * (kHandleThreads / 64)];
  *excess_total = 0;
  if (wg * kHandleThreads >= m) return 0;   // (uniform)
  const uint32_t* list = reinterpret_cast<const uint32_t*>(posted_list(v.counters));
  const int* tails = posted_list(v.counters) + VK_POSTED_SLOTS;
  const int padded = (m + 3) & ~3;
  int n_excess = 0, unused = 0;
  for (int i = (int)threadIdx.x; i < padded; i += kHandleThreads)
  {
    const uint32_t e = i < m ? list[i] : kPostedBucket;   // sentinel: before nothing
    listed[i] = e;
    n_excess += (int)(e >> 31);
    // the request of rank i takes free slot voxel_pointer - i: all of them are fetched now, together
    // with the list, instead of one by one behind the ranks
    if (i < m && voxel_ptr0 - i >= 0) free_slot[i] = v.free_voxel_blocks[voxel_ptr0 - i];
  }
  if (wg == 0)
  {
    block_sum2(unused, n_excess, red);
    *excess_total = n_excess;
  }
  else __syncthreads();

  const int max_count = v.main_block_count + v.excess_block_count;
  int dropped = 0;
  for (int i = wg * kHandleThreads + (int)threadIdx.x; i < m; i += wgs * kHandleThreads)
  {
    const uint32_t mine = listed[i];
    const uint32_t bucket = mine & kPostedBucket;
    // (loaded before the ranks are counted: the counting hides the latency)
    const unsigned long long packed = reinterpret_cast<const unsigned long long*>(v.allocation_blocks)[bucket];
    const int tail = tails[i];      // of the bucket's chain, noted by the request pass (post_request)
    // ranks in bucket order among this pass's requests
    int rank_all = 0, rank_excess = 0;
    for (int j = 0; j < padded; j += 4)
    {
      const uint4 q = *reinterpret_cast<const uint4*>(&listed[j]);
      const uint32_t e[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int u = 0; u < 4; ++u)
      {
        const int before = (e[u] & kPostedBucket) < bucket ? 1 : 0;
        rank_all += before;
        rank_excess += before & (int)(e[u] >> 31);
      }
    }

    // volume.cu:316-363 for this request (as handle_group (3))
    int entry_index = (int)bucket;
    if (mine & kPostedExcess)
    {
      entry_index = excess_ptr0 + rank_excess;
      if (entry_index < max_count) v.hash_entries[tail].next = entry_index;   // visible by its index (visibility_quads)
    }
    const int voxel_index = voxel_ptr0 - rank_all;
    if (entry_index < max_count && voxel_index >= 0)
    {
      const int lo = (int)(packed & 0xffffffffull);
      const int hi = (int)((packed >> 32) & 0xffffull);
      reinterpret_cast<int4*>(v.hash_entries)[entry_index] = make_int4(lo, hi, free_slot[rank_all], -1);
    }
    else ++dropped;
    if (bucket != keep_bucket)
    {
      v.allocation_types[bucket] = VK_ALLOC_NONE;
      reinterpret_cast<unsigned long long*>(v.allocation_blocks)[bucket] = 0ull;
    }
  }
  if (dropped) atomicAdd(&v.counters[VK_CTR_DROPPED], dropped);
  return __syncthreads_or(dropped);
}

// How the workgroups of handle_visibility_kernel arrive: ONE 64-bit atomic add that carries
// everything the finishing workgroup needs to know of the others — so that it has nothing to load:
//   bits  0-31  visible entries so far (a visibility workgroup's old value is the base of its slots)
//   bits 32-47  workgroups that have arrived
//   bits 48-59  EXCESS requests of the round (from the handle workgroup that counted the list)
//   bits 60-63  handle workgroups that dropped a request
__device__ __forceinline__ unsigned long long arrival(int visible, int excess, int dropped)
{
  return (unsigned long long)(uint32_t)visible | (1ull << 32) | ((unsigned long long)(uint32_t)excess << 48) |
         ((unsigned long long)(dropped ? 1u : 0u) << 60);
}
// Memory order. With `fenced` the arrival is acquire-release at agent scope and the workgroups fence around it (a
// write-back and an invalidation of the XCD's L2: the price of a hand-off between workgroups on this part, which is why
// it is paid only when the last workgroup has to READ what the others wrote). Without, the arrival is RELAXED, and what
// the last workgroup then does is store over things the others have read (the origin bucket's request, the counters,
// the arrival word itself). That the others' reads come first is not the memory model's promise but this hardware's:
// a workgroup's arrival is issued by lane 0 after a workgroup barrier, every value a lane of it read from those
// locations has been consumed by then (the arrival's addend and the slots it reserves are computed from them, so the
// loads have returned), and a CU issues a wave's memory operations in order. A port to a part that lets a load
// complete after a younger atomic of another lane of its workgroup would have to arrive with release semantics.
#ifndef VK_HANDLE_ALWAYS_FENCED
#define VK_HANDLE_ALWAYS_FENCED 0
#endif
__device__ __forceinline__ unsigned long long arrive(const vk_volume& v, unsigned long long add, bool fenced)
{
  unsigned long long* word = reinterpret_cast<unsigned long long*>(v.counters + VK_CTR_ARRIVALS);
  return fenced ? __hip_atomic_fetch_add(word, add, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT)
                : __hip_atomic_fetch_add(word, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// volume.cu:25-84 as visibility_chunk (deferred reset), four consecutive entries per lane: the
// 1024 entries from `first` (a multiple of 4) by a 256-lane workgroup, one atomic for all of them.
// Entries [new_lo, new_hi) are the excess entries this round's handle pass creates; they are
// visible (volume.cu:344) whatever their byte holds.
// The workgroup's arrival (see arrival()) is the atomic that reserves its slots of the visible list;
// returns the arrival word as it was before.
// the four visibility bytes of this lane's entries (entries past the end read as 0), and how many there are
__device__ __forceinline__ uint32_t visibility_quad_load(const vk_volume& v, int first, int& n)
{
  const int count = v.main_block_count + v.excess_block_count;
  const int index0 = first + 4 * (int)threadIdx.x;
  n = count - index0;
  n = n < 0 ? 0 : (n > 4 ? 4 : n);
  uint32_t stored = 0;
  if (n == 4) stored = *reinterpret_cast<const uint32_t*>(v.block_visibility + index0);
  else for (int k = 0; k < n; ++k) stored |= (uint32_t)v.block_visibility[index0 + k] << (8 * k);
  return stored;
}

// `stored`, `n`: visibility_quad_load, issued by the caller ahead of everything else. *my_arrival:
// what this workgroup added to the arrival word.
__device__ __forceinline__ unsigned long long visibility_quads(const VisibilityParams& P, int first, uint32_t stored, int n,
    int new_lo, int new_hi, bool fenced, unsigned long long* my_arrival)
{
  __shared__ int wave_total[kHandleThreads / 64];
  __shared__ unsigned long long arrived, added;
  // the visible entries binned by image row band (vk.h VK_BANDS): a workgroup's entries take their places in its
  // share of each band's list by LDS atomics, the shares come from one 8-lane atomic per workgroup
  __shared__ int band_fill[VK_BANDS], band_base[VK_BANDS];

  const vk_volume& v = P.v;
  const int index0 = first + 4 * (int)threadIdx.x;
  const float block_length = VK_BLOCK_RESOLUTION * v.voxel_length;
  if (threadIdx.x < VK_BANDS) band_fill[threadIdx.x] = 0;
  __syncthreads();

  uint32_t result = stored, seen = 0, bands = 0;   // bands: three bits per entry
#pragma unroll 1
  for (int k = 0; k < n; ++k)
  {
    const int index = index0 + k;
    const int byte = (int)((stored >> (8 * k)) & 0xffu);
    // touched this frame -> TRUE; otherwise what the reset pass would have left
    const int before = byte & 3;
    int visibility = (byte & kTouched) ? VK_VISIBILITY_TRUE : (before == VK_VISIBILITY_TRUE ? VK_VISIBILITY_UNKNOWN : before);
    if (index >= new_lo && index < new_hi) visibility = VK_VISIBILITY_TRUE;
    bool visible = (visibility == VK_VISIBILITY_TRUE);
    int out = visibility;
    // the band a ray left with the touched bit; entries nobody's ray touched (the handle pass's new ones, the
    // excess range) are spread evenly
    const int tagged = (byte & kTouched) ? ((byte >> 3) & 15) : 0;
    int band = tagged ? tagged - 1 : (index & (VK_BANDS - 1));
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
          band = vclampi(f2i(w * (float)VK_BANDS / (float)P.height), 0, VK_BANDS - 1);   // the row of that corner
          break;
        }
      }
      if (!visible) out = VK_VISIBILITY_FALSE;
    }
    result = (result & ~(0xffu << (8 * k))) | ((uint32_t)out << (8 * k));
    seen |= visible ? (1u << k) : 0u;
    bands |= (uint32_t)band << (3 * k);
  }
  // this lane's places in the workgroup's share of each band's list
  uint32_t places = 0, places_hi = 0;   // sixteen bits per entry (a workgroup holds 1024 entries)
#pragma unroll
  for (int k = 0; k < 4; ++k)
  {
    if (!(seen & (1u << k))) continue;
    const uint32_t at = (uint32_t)atomicAdd(&band_fill[(bands >> (3 * k)) & 7u], 1);   // < 1024
    if (k < 2) places |= at << (16 * k); else places_hi |= at << (16 * (k - 2));
  }
  if (result != stored)
  {
    if (n == 4) *reinterpret_cast<uint32_t*>(v.block_visibility + index0) = result;
    else for (int k = 0; k < n; ++k) v.block_visibility[index0 + k] = (uint8_t)(result >> (8 * k));
  }

  // compaction: lane counts -> wave scan -> one atomic per workgroup
  const int mine = __popc(seen);
  int incl = mine;
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1)
  {
    const int t = __shfl_up(incl, d);
    if (lane >= d) incl += t;
  }
  if (lane == 63) wave_total[wave] = incl;
  if (fenced) __threadfence();
  __syncthreads();
  if (threadIdx.x == 0)
  {
    int total = 0;
    for (int w = 0; w < kHandleThreads / 64; ++w)
    {
      const int c = wave_total[w];
      wave_total[w] = total;
      total += c;
    }
    added = arrival(total, 0, 0);
    arrived = arrive(v, added, fenced);
  }
  // (another wave, beside lane 0's arrival: the workgroup's share of each band's list)
  if (threadIdx.x >= 64 && threadIdx.x < 64 + VK_BANDS)
  {
    const int b = (int)threadIdx.x - 64;
    const int n_band = band_fill[b];
    band_base[b] = n_band > 0 ? atomicAdd(&band_counts(v.counters)[b], n_band) : 0;
  }
  __syncthreads();
  const unsigned long long before_us = arrived;
  *my_arrival = added;
  int offset = (int)(uint32_t)before_us + wave_total[wave] + incl - mine;
  int32_t* lists = band_lists(v.counters);
  for (int k = 0; k < 4; ++k)
    if (seen & (1u << k))
    {
      v.visible_blocks[offset++] = index0 + k;
      const int b = (int)((bands >> (3 * k)) & 7u);
      const int at = band_base[b] + (int)(((k < 2 ? places : places_hi) >> (16 * (k & 1))) & 0xffffu);
      if (at < VK_BAND_SLOTS) lists[b * VK_BAND_SLOTS + at] = index0 + k;   // (a band past its slots: the count says so)
    }
  __syncthreads();   // the LDS words are reused
  return before_us;
}

__device__ __forceinline__ void handle_visibility_body(const FusedParams& P)
{
  __shared__ unsigned long long arrived, added;
  __shared__ int range[2];
  __shared__ int red[2 * (kHandleThreads / 64)];
  const vk_volume& v = P.vis.v;
  // a visibility workgroup's bytes: asked for before anything else, they travel with the counters
  const bool vis_group = (int)blockIdx.x >= P.handle_wgs;
  const int vis_first = ((int)blockIdx.x - P.handle_wgs) * kVisPerGroup;
  int vis_n = 0;
  uint32_t vis_stored = 0;
  if (vis_group) vis_stored = visibility_quad_load(v, vis_first, vis_n);
  const int main_count = v.main_block_count;
  const int max_count = v.main_block_count + v.excess_block_count;
  // what the previous launches left (one cache line; nothing in this launch writes it before the end)
  const int contended = v.counters[VK_CTR_CONTENDED];
  const int origin_seen = v.counters[VK_CTR_ORIGIN_SEEN];
  const int posted = v.counters[VK_CTR_POSTED];
  const int voxel_ptr0 = v.counters[VK_CTR_VOXEL_PTR];
  const int excess_ptr0 = v.counters[VK_CTR_EXCESS_PTR];
  const int rounds0 = v.counters[VK_CTR_ROUNDS];
  const int overflow0 = v.counters[VK_CTR_RETRY_OVERFLOW];
  const int filed0 = v.counters[VK_CTR_RETRY_COUNT + 0], filed1 = v.counters[VK_CTR_RETRY_COUNT + 1];
  // Something the finishing workgroup has to READ of what the others write in this launch: only with
  // losers on file, or when the origin block was met in an unallocated main entry that this round
  // gives to another block. Both are known before the launch: the same answer in every workgroup.
  const uint32_t origin_bucket = block_hash(0, 0, 0, (uint32_t)main_count);
  // the requests of this round: listed, or — too many for the list — to be found in the flags
  const bool listed = posted <= P.posted_capacity;
  const bool contest = contended != 0 || (origin_seen != 0 && v.allocation_types[origin_bucket] != VK_ALLOC_NONE);
  // (the flag-scan form also arrives with release / acquire: its last workgroup reads VK_CTR_DROPPED_NOW, which the
  // others wrote, and clears the flags all of them read; it is the rare form, the fences cost it nothing that matters)
  // (-DVK_HANDLE_ALWAYS_FENCED=1: every arrival ordered, for the A/B ADVICE r3 / VERDICT r4 asked for; the numbers are
  // beside arrive())
  const bool fenced = VK_HANDLE_ALWAYS_FENCED || contest || !listed;
  const bool losers = P.max_rounds > 1 && contest;
  const uint32_t* list = reinterpret_cast<const uint32_t*>(posted_list(v.counters));

  unsigned long long before_us, my_arrival;
  if (!vis_group)
  {
    int excess_total = 0, dropped = 0;
    // (too many requests for the list: the flags are scanned by the visibility workgroups, which are
    // as many as there are groups of 1024 buckets, or more; these few have nothing to do then)
    if (listed)
      dropped = handle_listed(v, posted, (int)blockIdx.x, P.handle_wgs, origin_bucket, voxel_ptr0, excess_ptr0, &excess_total);
    if (fenced) __threadfence();
    __syncthreads();
    if (threadIdx.x == 0)
    {
      added = arrival(0, excess_total, dropped);
      arrived = arrive(v, added, fenced);
    }
    __syncthreads();
    before_us = arrived;
    my_arrival = added;
  }
  else
  {
    const int first = vis_first;
    int new_lo = 0, new_hi = 0;
    if (first + kVisPerGroup > excess_ptr0 && posted > 0)
    {
      // this chunk may hold entries the handle pass is creating right now: how many are there?
      int n_all = 0, n_excess = 0;
      if (listed)
      {
        for (int i = (int)threadIdx.x; i < posted; i += kHandleThreads) n_excess += (int)(list[i] >> 31);
      }
      else
      {
        const uint4* flags16 = reinterpret_cast<const uint4*>(v.allocation_types);
        const int chunks = main_count / 16;
        for (int c0 = (int)threadIdx.x; c0 < chunks; c0 += 8 * kHandleThreads)   // eight loads in flight
        {
          uint4 q[8];
#pragma unroll
          for (int u = 0; u < 8; ++u)
          {
            const int c = c0 + u * kHandleThreads;
            q[u] = c < chunks ? flags16[c] : make_uint4(0u, 0u, 0u, 0u);
          }
#pragma unroll
          for (int u = 0; u < 8; ++u)
          {
            count_flags(q[u].x, n_all, n_excess);
            count_flags(q[u].y, n_all, n_excess);
            count_flags(q[u].z, n_all, n_excess);
            count_flags(q[u].w, n_all, n_excess);
          }
        }
        for (int i = chunks * 16 + (int)threadIdx.x; i < main_count; i += kHandleThreads)
          n_excess += v.allocation_types[i] == VK_ALLOC_EXCESS ? 1 : 0;
      }
      block_sum2(n_all, n_excess, red);
      new_lo = excess_ptr0;
      new_hi = excess_ptr0 + n_excess < max_count ? excess_ptr0 + n_excess : max_count;
    }
    if (!listed)
    {
      // the handle pass from the flags (handle_group), one group of 1024 buckets per visibility
      // workgroup — before this workgroup arrives: the flags are cleared by the last arrival
      const int groups = (main_count + kHandlePerGroup - 1) / kHandlePerGroup;
      const int vis_groups = (int)gridDim.x - P.handle_wgs;
      for (int group = (int)blockIdx.x - P.handle_wgs; group < groups; group += vis_groups)
        handle_group(v, group, groups, 0, 2, v.counters + VK_CTR_DROPPED_NOW);
    }
    before_us = visibility_quads(P.vis, first, vis_stored, vis_n, new_lo, new_hi, fenced, &my_arrival);
  }

  // the last one to arrive finishes the round
  if ((int)((before_us >> 32) & 0xffffull) != (int)gridDim.x - 1) return;
  if (fenced) __threadfence();
  const unsigned long long all = before_us + my_arrival;     // the word after the last arrival: ours
  const int visible_total = (int)(uint32_t)all;

  if (listed && !losers)
  {
    // the common case: everything is known, nothing is loaded (volume.cu:365, :352, :337 and the
    // end-of-call bookkeeping of finish_handle)
    if (threadIdx.x == 0)
    {
      const int excess_total = (int)((all >> 48) & 0xfffull);
      const int dropped = (int)(all >> 60);
      int pending = (contended | dropped | overflow0) != 0;
      if (origin_seen) pending |= origin_block_pending(v);
      if (posted > 0)
      {
        v.allocation_types[origin_bucket] = VK_ALLOC_NONE;
        reinterpret_cast<unsigned long long*>(v.allocation_blocks)[origin_bucket] = 0ull;
      }
      v.counters[VK_CTR_VISIBLE] = visible_total;
      v.counters[VK_CTR_BANDED] = visible_total;     // every visible entry is in a banded list as well
      v.counters[VK_CTR_VOXEL_PTR] = voxel_ptr0 - posted;
      v.counters[VK_CTR_EXCESS_PTR] = excess_ptr0 + excess_total;
      // a round that is not run because nothing was pending would have seen no request
      v.counters[VK_CTR_REQUESTS] = (P.max_rounds > 1 && !pending) ? 0 : posted;
      v.counters[VK_CTR_POSTED] = 0;
      v.counters[VK_CTR_UNSETTLED] = pending;
      v.counters[VK_CTR_ROUNDS] = rounds0 + 1;
      v.counters[VK_CTR_CONTENDED] = 0;
      v.counters[VK_CTR_DROPPED_NOW] = 0;
      v.counters[VK_CTR_RETRY_OVERFLOW] = 0;
      v.counters[VK_CTR_ORIGIN_SEEN] = 0;
      v.counters[VK_CTR_ARRIVALS] = 0;
      v.counters[VK_CTR_ARRIVALS + 1] = 0;
    }
    if (filed0 | filed1)   // (max_rounds == 1 with losers: they leave the set)
    {
      for (int which = 0; which < 2; ++which)
      {
        int n = which ? filed1 : filed0;
        if (n > VK_RETRY_KEYS) n = VK_RETRY_KEYS;
        unsigned long long* table = retry_table(v.counters, which);
        const int* slots = retry_slots(v.counters, which);
        for (int i = (int)threadIdx.x; i < n; i += kHandleThreads) table[slots[i]] = 0ull;
      }
      if (threadIdx.x < 2) v.counters[VK_CTR_RETRY_COUNT + threadIdx.x] = 0;
    }
    return;
  }

  // ---- the rare cases: too many requests for the list, or losers on file
  int total_all = posted, total_excess = (int)((all >> 48) & 0xfffull);
  if (!listed) clear_requests(v, &total_all, &total_excess);
  if (threadIdx.x == 0)
  {
    if (listed && posted > 0)
    {
      v.allocation_types[origin_bucket] = VK_ALLOC_NONE;
      reinterpret_cast<unsigned long long*>(v.allocation_blocks)[origin_bucket] = 0ull;
    }
    if (listed && (all >> 60)) v.counters[VK_CTR_DROPPED_NOW] = 1;
    v.counters[VK_CTR_VISIBLE] = visible_total;
    // later rounds append to the plain list only: the banded lists are complete if none runs
    v.counters[VK_CTR_BANDED] = losers ? -1 : visible_total;
    v.counters[VK_CTR_VOXEL_PTR] = voxel_ptr0 - total_all;
    v.counters[VK_CTR_EXCESS_PTR] = excess_ptr0 + total_excess;
    v.counters[VK_CTR_REQUESTS] = total_all;
    v.counters[VK_CTR_POSTED] = 0;
  }
  __syncthreads();
  int pending = 0;
  if (losers) pending = later_rounds<true>(v, P.max_rounds, P.retry_capacity);   // (lane 0's value)
  __syncthreads();

  // end-of-call bookkeeping (as finish_handle with max_rounds), and the sets left empty
  if (threadIdx.x == 0)
  {
    if (!losers)
    {
      // what is still unanswered: a request lost to a bucket contest or dropped, or more losers
      // than the retry list holds
      const int dropped_now = __hip_atomic_load(&v.counters[VK_CTR_DROPPED_NOW], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pending = (contended | dropped_now | overflow0 | origin_block_pending(v)) != 0;
    }
    v.counters[VK_CTR_UNSETTLED] = pending;
    v.counters[VK_CTR_ROUNDS] += 1;              // the first round; later_rounds() has added the others
    // a round that is not run because nothing was pending would have seen no request
    if (P.max_rounds > 1 && !pending && !losers) v.counters[VK_CTR_REQUESTS] = 0;
    range[0] = __hip_atomic_load(&v.counters[VK_CTR_RETRY_COUNT + 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    range[1] = __hip_atomic_load(&v.counters[VK_CTR_RETRY_COUNT + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  for (int which = 0; which < 2; ++which)
  {
    int n = range[which];
    if (n > VK_RETRY_KEYS) n = VK_RETRY_KEYS;
    unsigned long long* table = retry_table(v.counters, which);
    const int* slots = retry_slots(v.counters, which);
    for (int i = (int)threadIdx.x; i < n; i += kHandleThreads) table[slots[i]] = 0ull;
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    v.counters[VK_CTR_CONTENDED] = 0;
    v.counters[VK_CTR_DROPPED_NOW] = 0;
    v.counters[VK_CTR_RETRY_OVERFLOW] = 0;
    v.counters[VK_CTR_ORIGIN_SEEN] = 0;
    v.counters[VK_CTR_RETRY_COUNT + 0] = 0;
    v.counters[VK_CTR_RETRY_COUNT + 1] = 0;
    v.counters[VK_CTR_ARRIVALS] = 0;
    v.counters[VK_CTR_ARRIVALS + 1] = 0;
  }
}

__global__ __launch_bounds__(kHandleThreads) void handle_visibility_kernel(FusedParams P)
{
  handle_visibility_body(P);
}

// The same launch with the view's pose taken from the DEVICE (vk_volume_set_view_at_device_pose, round 6): the frustum test's
// world -> depth transform is the inverse a tracker's launch in front of this one left in *pose. Twelve uniform loads; the
// rest is handle_visibility_kernel to the instruction (make_rt's transposition of the column-major matrix).
__global__ __launch_bounds__(kHandleThreads) void handle_visibility_at_kernel(FusedParams P, const vk_transform* __restrict__ pose)
{
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) P.vis.Tdw.r[r * 4 + c] = pose->inv[c * 4 + r];
  handle_visibility_body(P);
}

int check_volume(const vk_volume* v)
{
  if (!v) return VK_ERR_ARGUMENT;
  if (!v->voxels || !v->hash_entries || !v->free_voxel_blocks || !v->allocation_types ||
      !v->allocation_blocks || !v->block_visibility || !v->visible_blocks || !v->counters)
    return VK_ERR_ARGUMENT;
  if (v->main_block_count <= 0 || v->excess_block_count < 0) return VK_ERR_ARGUMENT;
  if (!(v->voxel_length > 0) || !(v->truncation_length > 0)) return VK_ERR_ARGUMENT;
  if ((reinterpret_cast<uintptr_t>(v->counters) & 7) || (reinterpret_cast<uintptr_t>(v->allocation_blocks) & 7) || (reinterpret_cast<uintptr_t>(v->hash_entries) & 15) ||
      (reinterpret_cast<uintptr_t>(v->voxels) & 15) || (reinterpret_cast<uintptr_t>(v->block_visibility) & 3) ||
      (reinterpret_cast<uintptr_t>(v->allocation_types) & 15))
    return VK_ERR_ARGUMENT;
  return VK_OK;
}

int launch_create_requests(const vk_volume* v, const float* depth, int width, int height,
    const vk_projection* projection, const vk_transform* Twd, bool deferred_reset, hipStream_t s,
    const vk_frame* prep_frame = nullptr, const vk_light_prep* prep = nullptr, bool fused = false,
    const vk_transform* pose_dev = nullptr)
{
  RequestParams P;
  Retry retry;
  const int with_prep = build_request_pass(P, retry, v, depth, width, height, projection, Twd, prep_frame, prep, fused);
  const dim3 grid((width + 63) / 64, (height + 3) / 4);
  if (pose_dev)
  {
    if (with_prep == 2 || !deferred_reset) return VK_ERR_UNSUPPORTED;
    if (with_prep == 1) hipLaunchKernelGGL((create_requests_at_kernel<1>), grid, dim3(256), 0, s, P, retry, pose_dev);
    else hipLaunchKernelGGL((create_requests_at_kernel<0>), grid, dim3(256), 0, s, P, retry, pose_dev);
    VK_LAUNCH_CHECK();
    return VK_OK;
  }
  if (with_prep == 2) hipLaunchKernelGGL((create_requests_kernel<true, 2>), grid, dim3(256), 0, s, P, retry);
  else if (with_prep == 1) hipLaunchKernelGGL((create_requests_kernel<true, 1>), grid, dim3(256), 0, s, P, retry);
  else if (deferred_reset) hipLaunchKernelGGL((create_requests_kernel<true, 0>), grid, dim3(256), 0, s, P, retry);
  else hipLaunchKernelGGL((create_requests_kernel<false, 0>), grid, dim3(256), 0, s, P, retry);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

int launch_update_visibility(const vk_volume* v, int width, int height,
    const vk_projection* projection, const float* Tdw_m, int finish, bool deferred_reset, hipStream_t s)
{
  VisibilityParams P;
  P.v = *v;
  P.finish_handle = finish;      // 0: the staged pass; n >= 1: the fused SetView of n rounds
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
  // the retry sets and slot lists behind the counters start empty
  VK_CHECK(hipMemsetAsync(v->counters + VK_CTR_PUBLIC, 0, sizeof(int32_t) * (VK_CTR_COUNT - VK_CTR_PUBLIC), vk_s(stream)));
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
      dim3(kHandleThreads), 0, vk_s(stream), *v, 0, 0);
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
  return launch_update_visibility(v, width, height, projection, Tdw->m, 0, false, vk_s(stream));
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

// SetView's second launch — the handle pass, the visibility pass and the later rounds (volume.cu:520-535, :473-495): what
// follows a request pass, whoever made it and when. `Tdw`: the 16 floats of depth_to_world's cached inverse.
static int launch_handle_visibility(const vk_volume* v, int width, int height, const vk_projection& k, const float* Tdw,
    int max_rounds, hipStream_t s, const vk_transform* pose_dev = nullptr)
{
  const int handle_groups = (v->main_block_count + kHandlePerGroup - 1) / kHandlePerGroup;
  const int vis_groups = (v->main_block_count + v->excess_block_count + kVisPerGroup - 1) / kVisPerGroup;
  if (!set_view_unfused(v))
  {
    // two launches: requests, then handle + visibility (+ the later rounds, if a request lost)
    FusedParams F;
    F.vis.v = *v;
    F.vis.finish_handle = 0;
    F.vis.deferred_reset = 1;
    F.vis.width = width;
    F.vis.height = height;
    F.vis.k = k;
    F.vis.Tdw = make_rt(Tdw);
    F.handle_wgs = handle_groups < VK_POSTED_SLOTS / kHandleThreads ? handle_groups : VK_POSTED_SLOTS / kHandleThreads;
    F.max_rounds = max_rounds;
    F.retry_capacity = retry_capacity();
    F.posted_capacity = posted_capacity();
    if (pose_dev) hipLaunchKernelGGL(handle_visibility_at_kernel, dim3(F.handle_wgs + vis_groups), dim3(kHandleThreads), 0, s, F, pose_dev);
    else hipLaunchKernelGGL(handle_visibility_kernel, dim3(F.handle_wgs + vis_groups), dim3(kHandleThreads), 0, s, F);
    VK_LAUNCH_CHECK();
    return VK_OK;
  }
  if (pose_dev) return VK_ERR_UNSUPPORTED;
  // the three-launch form (vk_test_hooks.set_view_unfused: kept for comparison and as the reference for the fused one)
  hipLaunchKernelGGL(handle_rounds_kernel, dim3(handle_groups), dim3(kHandleThreads), 0, s, *v, max_rounds, retry_capacity());
  VK_LAUNCH_CHECK();  // also zeroes counters[VK_CTR_VISIBLE]; pointers are folded in by the next kernel
  return launch_update_visibility(v, width, height, &k,
      Tdw, max_rounds, true, s);
}

// `request_stream` / `ordering` (vk_volume_set_view_rounds_split): the request pass — and the normals launch, when the
// call has to make it — go to request_stream; `ordering` is recorded behind them there and `stream` waits for it before
// the handle + visibility pass.
// `requests` (vk_volume_set_view_rounds_ahead): the request pass of exactly this call may have run already, behind the
// previous frame's raycast (vk_trace_ahead_requests); then only the handle + visibility pass is launched.
static bool requests_made_for(const vk_requests_ahead* r, const vk_volume* v, const vk_frame* frame, const vk_light_prep* prep, bool ride)
{
  return r && r->valid == 1 && r->counters == v->counters && r->depth == frame->depth && r->width == frame->width &&
      r->height == frame->height && frame->content_id != 0 && r->content_id == frame->content_id &&
      memcmp(&r->depth_projection, &frame->depth_projection, sizeof(vk_projection)) == 0 &&
      (r->pose_on_device == 1 || memcmp(&r->depth_to_world, &frame->depth_to_world, sizeof(vk_transform)) == 0) &&
      r->prep == (ride ? (const void*)prep : nullptr);
}

static int set_view(const vk_volume* v, const vk_frame* frame, vk_light_prep* prep, int max_rounds, void* stream,
    void* request_stream = nullptr, void* ordering = nullptr, vk_requests_ahead* requests = nullptr)
{
  const int rc = check_volume(v);
  if (rc != VK_OK) return rc;
  VK_REQUIRE(frame && frame->depth && frame->width > 0 && frame->height > 0 && max_rounds >= 1);
  hipStream_t s = vk_s(stream);
  const bool split = ordering != nullptr;
  void* const first_stream = split ? request_stream : stream;
  // the preparation rides along when the frame has what LightIntegrator needs (prep_rides, vk_requests.hpp)
  const bool ride = prep_rides(prep, frame);
  // (a pass made ahead has also made the preparation: it is still valid for this frame, and its normals are written)
  const bool made_ahead = requests_made_for(requests, v, frame, prep, ride) &&
      (!ride || (prep->valid == 1 && prep->content_id == frame->content_id &&
                 (prep->normals_out == nullptr || prep->normals_out == frame->normals)));
  // A pass made ahead for ANOTHER frame has left that frame's requests in the volume: handled together with this frame's
  // they would allocate in an order no sequence of upstream calls gives. Refused, with nothing launched and the record kept.
  if (requests && requests->valid == 1 && !made_ahead) return VK_ERR_ARGUMENT;
  if (requests) requests->valid = 0;      // used once
  if (made_ahead && prep) prep->normals_out = nullptr;   // (asked for again by the caller: they were written with the pass)
  if (!made_ahead)
  {
    if (prep) prep->valid = 0;
    if (prep && prep->normals_out)
    {
      // the frame's normals are still to be computed (vk_light_prep.normals_out): on the way when the
      // preparation rides along, else by the launch the caller left out
      VK_REQUIRE(prep->normals_out == frame->normals);
      if (!ride)
      {
        int rn = vk_frame_compute_normals(frame->depth, &frame->depth_projection, prep->normals_out, frame->width, frame->height, first_stream);
        prep->normals_out = nullptr;
        if (rn != VK_OK) return rn;
      }
    }
    // two launches (requests; handle + visibility): the reset pass is folded into them (see kTouched),
    // and so are all rounds after the first (later_rounds)
    int r;
    if ((r = launch_create_requests(v, frame->depth, frame->width, frame->height,
             &frame->depth_projection, &frame->depth_to_world, true, vk_s(first_stream), ride ? frame : nullptr, ride ? prep : nullptr,
             true)) != VK_OK) return r;
    if (ride) prep_note_made(prep, frame);
  }
  if (split)
  {
    VK_CHECK(hipEventRecord(reinterpret_cast<hipEvent_t>(ordering), vk_s(first_stream)));
    VK_CHECK(hipStreamWaitEvent(s, reinterpret_cast<hipEvent_t>(ordering), 0));
  }
  return launch_handle_visibility(v, frame->width, frame->height, frame->depth_projection, frame->depth_to_world.inv, max_rounds, s);
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

int vk_volume_set_view_rounds_split(const vk_volume* v, const vk_frame* frame, vk_light_prep* prep, int max_rounds,
    void* request_stream, void* ordering_event, void* stream)
{
  VK_REQUIRE(ordering_event);
  return set_view(v, frame, prep, max_rounds, stream, request_stream, ordering_event);
}

int vk_volume_set_view_rounds_ahead(const vk_volume* v, const vk_frame* frame, vk_light_prep* prep, int max_rounds,
    vk_requests_ahead* requests, void* stream)
{
  return set_view(v, frame, prep, max_rounds, stream, nullptr, nullptr, requests);
}

int vk_volume_requests_at_device_pose(const vk_volume* v, const vk_frame* frame, const vk_transform* pose_dev, vk_light_prep* prep,
    vk_requests_ahead* requests, void* stream)
{
  const int rc = check_volume(v);
  if (rc != VK_OK) return rc;
  VK_REQUIRE(frame && frame->depth && frame->width > 0 && frame->height > 0 && pose_dev && requests && frame->content_id != 0);
  // a record that still announces a frame: its requests are in the volume (vk_trace_ahead_requests refuses the same way)
  if (requests->valid == 1) return VK_ERR_ARGUMENT;
  // the frame's normals must exist: computing them on the way (normals_out) is SetView's own pass
  if (prep && prep->normals_out) return VK_ERR_UNSUPPORTED;
  const bool ride = prep_rides(prep, frame);
  if (prep) prep->valid = 0;
  const int rl = launch_create_requests(v, frame->depth, frame->width, frame->height, &frame->depth_projection, &frame->depth_to_world,
      true, vk_s(stream), ride ? frame : nullptr, ride ? prep : nullptr, true, pose_dev);
  if (rl != VK_OK) return rl;
  if (ride) prep_note_made(prep, frame);
  requests->counters = v->counters;
  requests->depth = frame->depth;
  requests->prep = ride ? prep : nullptr;
  requests->width = frame->width;
  requests->height = frame->height;
  requests->depth_projection = frame->depth_projection;
  requests->depth_to_world = frame->depth_to_world;      // the caller's best knowledge (the Track's start pose): what a cancel uses
  requests->content_id = frame->content_id;
  requests->normals_made = 0;
  requests->pose_on_device = 1;
  requests->valid = 1;
  return VK_OK;
}

int vk_volume_set_view_at_device_pose(const vk_volume* v, const vk_frame* frame, const vk_transform* pose_dev, vk_light_prep* prep,
    int max_rounds, void* stream)
{
  const int rc = check_volume(v);
  if (rc != VK_OK) return rc;
  VK_REQUIRE(frame && frame->depth && frame->width > 0 && frame->height > 0 && pose_dev && max_rounds >= 1);
  if ((prep && prep->normals_out) || set_view_unfused(v)) return VK_ERR_UNSUPPORTED;
  const bool ride = prep_rides(prep, frame);
  if (prep) prep->valid = 0;
  int rl = launch_create_requests(v, frame->depth, frame->width, frame->height, &frame->depth_projection, &frame->depth_to_world,
      true, vk_s(stream), ride ? frame : nullptr, ride ? prep : nullptr, true, pose_dev);
  if (rl != VK_OK) return rl;
  if (ride) prep_note_made(prep, frame);
  return launch_handle_visibility(v, frame->width, frame->height, frame->depth_projection, frame->depth_to_world.inv, max_rounds,
      vk_s(stream), pose_dev);
}

int vk_requests_ahead_cancel(const vk_volume* v, vk_requests_ahead* requests, int max_rounds, void* stream)
{
  const int rc = check_volume(v);
  if (rc != VK_OK) return rc;
  VK_REQUIRE(requests && max_rounds >= 1);
  if (requests->valid != 1) return VK_OK;                       // nothing announced, nothing pending
  VK_REQUIRE(requests->counters == v->counters && requests->width > 0 && requests->height > 0);
  requests->valid = 0;
  // the announced frame's SetView, completed: its requests are handled, its visible list made — a state upstream reaches
  return launch_handle_visibility(v, requests->width, requests->height, requests->depth_projection,
      requests->depth_to_world.inv, max_rounds, vk_s(stream));
}

int vk_volume_read_counters_sync(const vk_volume* v, int32_t* host_out, void* stream)
{
  VK_REQUIRE(v && v->counters && host_out);
  VK_CHECK(hipMemcpyAsync(host_out, v->counters, sizeof(int32_t) * VK_CTR_PUBLIC,
      hipMemcpyDeviceToHost, vk_s(stream)));
  VK_CHECK(hipStreamSynchronize(vk_s(stream)));
  return VK_OK;
}

}  // extern "C"
