// vk_detect.hip — box detector over a point cloud (include/vk.h "detector";
// ref: src/detector.cu).
//
// The reference is three kernels glued by host round trips: a compaction whose
// output order depends on which thread block wins an atomic, three cublasSasum +
// one cublasSdot, a blocking count readback after each compaction. Here the whole
// of Detect() is eight launches with no host involvement:
//
//   count<box> -> scatter<box> -> |x| partials -> d^2 partials
//              -> count<sigma> -> scatter<sigma> -> |x| partials -> position
//
//  * compaction is stable (input order): a workgroup owns a chunk of 4096 points,
//    16 rows of 256; ranks inside a row come from wave ballots, row/wave bases
//    from a 64-entry LDS scan, the chunk base from the per-chunk counts of the
//    first pass;
//  * sums use one fixed tree (per chunk: 256 strided partials, binary tree;
//    chunks added in order by whoever needs the total), so centre, limit and
//    position are bit-reproducible (the CPU restatement used by the tests follows the same tree);
//  * counts stay on the device: grids are sized by the host's upper bound and
//    chunks beyond the live count retire at once.
#include "vk_common.hpp"

using namespace vk;

namespace
{

constexpr int kChunk = 4096;
constexpr int kLanes = 256;
constexpr int kRows = kChunk / kLanes;

struct DetectParams
{
  vk_detector d;
  const float* in;
  float* out;
  vk_detect_state* state;
  int32_t* chunk_counts;   // [chunks]
  float* partials;         // [chunks][4]
  int count;               // host upper bound = number of input points
  int chunks;
};

__device__ __forceinline__ f3 load_point(const float* points, int i)
{
  return make3(points[3 * i + 0], points[3 * i + 1], points[3 * i + 2]);
}

__device__ __forceinline__ float norm3(f3 a) { return sqrtf(sqnorm3(a)); }

// STAGE 0: detector.cu:22-29 (radius, then the three intervals)
// STAGE 1: detector.cu:48 (within `limit` of the centroid)
template <int STAGE>
__device__ __forceinline__ bool keep(const DetectParams& P, f3 centre, float limit, f3 p)
{
  if (STAGE == 0)
  {
    const f3 origin = make3(P.d.origin[0], P.d.origin[1], P.d.origin[2]);
    bool valid = (P.d.radius <= 0 || norm3(sub3(p, origin)) < P.d.radius);
    const float own[3] = {p.x, p.y, p.z};
#pragma unroll
    for (int a = 0; a < 3; ++a)
    {
      const float v = P.d.bounds_use_own_axis ? own[a] : p.x;
      const float lo = P.d.bounds[a][0], hi = P.d.bounds[a][1];
      valid = valid && (lo > hi || (v >= lo && v <= hi));
    }
    return valid;
  }
  return norm3(sub3(p, centre)) <= limit;
}

template <int STAGE>
__device__ __forceinline__ int live_count(const DetectParams& P)
{
  return STAGE == 0 ? P.count : P.state->filtered_count;
}

// sum of the first `chunks` per-chunk partials of `column`, in chunk order
__device__ __forceinline__ float ordered_total(const float* partials, int live, int column)
{
  float total = 0.0f;
  for (int c = 0; c * kChunk < live; ++c) total += partials[4 * c + column];
  return total;
}

// The sigma stage needs centre and limit; every workgroup derives them from the
// partial sums in the same order (workgroup 0 also records them in the state).
struct Spread
{
  f3 centre;
  float limit;
};

__device__ __forceinline__ Spread load_spread(const DetectParams& P, bool record)
{
  __shared__ float shared[4];
  const int n = P.state->filtered_count;
  if (threadIdx.x == 0)
  {
    const float squared_error = ordered_total(P.partials, n, 3);
    const float stdev = sqrtf(squared_error / (float)n);     // detector.cu:178
    shared[3] = 1.5f * stdev;                                 // :179
    if (record)
    {
      P.state->squared_error = squared_error;
      P.state->limit = shared[3];
    }
  }
  if (threadIdx.x < 3) shared[threadIdx.x] = P.state->center[threadIdx.x];
  __syncthreads();
  Spread s;
  s.centre = make3(shared[0], shared[1], shared[2]);
  s.limit = shared[3];
  return s;
}

template <int STAGE>
__global__ __launch_bounds__(kLanes) void detect_count_kernel(DetectParams P)
{
  __shared__ int wave_totals[4];
  const int chunk = blockIdx.x;
  const int n = live_count<STAGE>(P);
  Spread s;
  s.centre = make3(0, 0, 0);
  s.limit = 0;
  if (STAGE == 1 && n > 0) s = load_spread(P, chunk == 0);

  int kept = 0;   // wave-uniform
  if (chunk * kChunk < n)
  {
#pragma unroll 4
    for (int row = 0; row < kRows; ++row)
    {
      const int i = chunk * kChunk + row * kLanes + (int)threadIdx.x;
      const bool flag = (i < n) && keep<STAGE>(P, s.centre, s.limit, load_point(P.in, i));
      kept += __popcll(__ballot(flag));
    }
  }
  if (lane_id() == 0) wave_totals[threadIdx.x >> 6] = kept;
  __syncthreads();
  if (threadIdx.x == 0) P.chunk_counts[chunk] = wave_totals[0] + wave_totals[1] + wave_totals[2] + wave_totals[3];
}

template <int STAGE>
__global__ __launch_bounds__(kLanes) void detect_scatter_kernel(DetectParams P)
{
  __shared__ int cells[kRows * 4];   // kept points per (row, wave), then their exclusive scan
  __shared__ int chunk_base;
  __shared__ int lane_sums[64];

  const int chunk = blockIdx.x;
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const int n = live_count<STAGE>(P);
  Spread s;
  s.centre = make3(0, 0, 0);
  s.limit = 0;
  if (STAGE == 1 && n > 0) s = load_spread(P, false);

  // points kept in the chunks before this one
  if (wave == 0)
  {
    int before = 0;
    for (int c = lane; c < chunk; c += 64) before += P.chunk_counts[c];
    lane_sums[lane] = before;
  }

  uint32_t flags = 0;
  if (chunk * kChunk < n)
  {
#pragma unroll 4
    for (int row = 0; row < kRows; ++row)
    {
      const int i = chunk * kChunk + row * kLanes + (int)threadIdx.x;
      const bool flag = (i < n) && keep<STAGE>(P, s.centre, s.limit, load_point(P.in, i));
      const unsigned long long ballot = __ballot(flag);
      if (lane == 0) cells[row * 4 + wave] = __popcll(ballot);
      flags |= (flag ? 1u : 0u) << row;
    }
  }
  else
  {
    if (lane < kRows) cells[lane * 4 + wave] = 0;
  }
  __syncthreads();

  if (threadIdx.x == 0)
  {
    int base = 0;
    for (int l = 0; l < 64; ++l) base += lane_sums[l];
    chunk_base = base;
    int running = 0;
    for (int c = 0; c < kRows * 4; ++c)
    {
      const int here = cells[c];
      cells[c] = running;
      running += here;
    }
    if (chunk == (int)gridDim.x - 1)
    {
      if (STAGE == 0) P.state->filtered_count = base + running;
      else P.state->inlier_count = base + running;
    }
  }
  __syncthreads();

  if (chunk * kChunk >= n) return;
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll 4
  for (int row = 0; row < kRows; ++row)
  {
    const bool flag = (flags >> row) & 1u;
    const unsigned long long ballot = __ballot(flag);
    if (flag)
    {
      const int i = chunk * kChunk + row * kLanes + (int)threadIdx.x;
      const int o = chunk_base + cells[row * 4 + wave] + __popcll(ballot & below);
      const f3 p = load_point(P.in, i);
      P.out[3 * o + 0] = p.x;
      P.out[3 * o + 1] = p.y;
      P.out[3 * o + 2] = p.z;
    }
  }
}

// 256 strided partial sums folded by a binary tree: strides 128 and 64 through
// LDS, 32 ... 1 inside wave 0 — the same pairing as the oracle's array loop.
__device__ __forceinline__ float chunk_tree(float value, float* scratch)
{
  scratch[threadIdx.x] = value;
  __syncthreads();
  if (threadIdx.x < 128) scratch[threadIdx.x] += scratch[threadIdx.x + 128];
  __syncthreads();
  float v = 0.0f;
  if (threadIdx.x < 64)
  {
    v = scratch[threadIdx.x] + scratch[threadIdx.x + 64];
#pragma unroll
    for (int stride = 32; stride >= 1; stride >>= 1) v += __shfl_down(v, stride, 64);
  }
  __syncthreads();
  return v;   // valid in thread 0
}

// MODE 0: per-axis sum of |x| (cublasSasum, detector.cu:137-139) -> partials[.][0..2]
// MODE 1: sum of Norm(p - centre)^2 (DistanceKernel + Sdot, :54-64,177) -> partials[.][3];
//         the centre comes from the MODE 0 partials of the same points
template <int MODE>
__global__ __launch_bounds__(kLanes) void detect_partials_kernel(DetectParams P, const int32_t* live)
{
  __shared__ float scratch[kLanes];
  __shared__ float centre_s[3];
  const int chunk = blockIdx.x;
  const int n = *live;
  if (chunk * kChunk >= n) return;

  f3 centre = make3(0, 0, 0);
  if (MODE == 1)
  {
    if (threadIdx.x < 3)
    {
      const float inv = 1.0f / (float)n;                               // matrix.h:290-295
      const float c = ordered_total(P.partials, n, (int)threadIdx.x) * inv;
      centre_s[threadIdx.x] = c;
      if (chunk == 0) P.state->center[threadIdx.x] = c;
    }
    __syncthreads();
    centre = make3(centre_s[0], centre_s[1], centre_s[2]);
  }

  float acc[3] = {0.0f, 0.0f, 0.0f};
  for (int i = chunk * kChunk + (int)threadIdx.x; i < n && i < (chunk + 1) * kChunk; i += kLanes)
  {
    const f3 p = load_point(P.in, i);
    if (MODE == 0)
    {
      acc[0] += fabsf(p.x);
      acc[1] += fabsf(p.y);
      acc[2] += fabsf(p.z);
    }
    else
    {
      const float d = norm3(sub3(p, centre));
      acc[0] += d * d;
    }
  }

  if (MODE == 0)
  {
#pragma unroll
    for (int a = 0; a < 3; ++a)
    {
      const float total = chunk_tree(acc[a], scratch);
      if (threadIdx.x == 0) P.partials[4 * chunk + a] = total;
    }
  }
  else
  {
    const float total = chunk_tree(acc[0], scratch);
    if (threadIdx.x == 0) P.partials[4 * chunk + 3] = total;
  }
}

// detector.cu:120-147: the survivors' sum|x| / n, or NaN when too few survive
__global__ void detect_position_kernel(DetectParams P)
{
  if (threadIdx.x >= 3) return;
  const int n = P.state->inlier_count;
  const bool detected = n >= P.d.min_inlier_count;
  float value = __builtin_nanf("");
  if (detected)
  {
    const float inv = 1.0f / (float)n;
    value = ordered_total(P.partials, n, (int)threadIdx.x) * inv;
  }
  P.state->position[threadIdx.x] = value;
  if (threadIdx.x == 0) P.state->detected = detected ? 1 : 0;
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

int chunks_for(int count) { return count > 0 ? (count + kChunk - 1) / kChunk : 1; }

int run_filter(const vk_detector* detector, const float* points, int32_t count, float* inliers,
    vk_detect_state* state_dev, void* workspace, hipStream_t s, DetectParams& P)
{
  VK_REQUIRE(detector && state_dev && workspace && count >= 0);
  VK_REQUIRE(count == 0 || (points && inliers));
  VK_REQUIRE(((uintptr_t)workspace & 15) == 0 && ((uintptr_t)state_dev & 3) == 0);

  const int chunks = chunks_for(count);
  char* ws = static_cast<char*>(workspace);
  P.d = *detector;
  P.state = state_dev;
  P.chunk_counts = reinterpret_cast<int32_t*>(ws);
  P.partials = reinterpret_cast<float*>(ws + align256(sizeof(int32_t) * chunks));
  float* filtered = reinterpret_cast<float*>(ws + align256(sizeof(int32_t) * chunks) + align256(sizeof(float) * 4 * chunks));
  P.count = count;
  P.chunks = chunks;

  VK_CHECK(hipMemsetAsync(state_dev, 0, sizeof(vk_detect_state), s));
  if (count == 0) return VK_OK;

  const dim3 grid(chunks), block(kLanes);
  P.in = points;
  P.out = filtered;
  hipLaunchKernelGGL(detect_count_kernel<0>, grid, block, 0, s, P);
  hipLaunchKernelGGL(detect_scatter_kernel<0>, grid, block, 0, s, P);
  P.in = filtered;
  P.out = inliers;
  hipLaunchKernelGGL(detect_partials_kernel<0>, grid, block, 0, s, P, &state_dev->filtered_count);
  hipLaunchKernelGGL(detect_partials_kernel<1>, grid, block, 0, s, P, &state_dev->filtered_count);
  hipLaunchKernelGGL(detect_count_kernel<1>, grid, block, 0, s, P);
  hipLaunchKernelGGL(detect_scatter_kernel<1>, grid, block, 0, s, P);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

}  // namespace

extern "C" {

VK_API size_t vk_detect_workspace_bytes(int32_t count)
{
  if (count < 0) return 0;
  const int chunks = chunks_for(count);
  return align256(sizeof(int32_t) * chunks) + align256(sizeof(float) * 4 * chunks) +
         align256(sizeof(float) * 3 * (size_t)(count > 0 ? count : 1));
}

VK_API int vk_detect_filter(const vk_detector* detector, const float* points, int32_t count,
    float* inliers, vk_detect_state* state_dev, void* workspace, void* stream)
{
  DetectParams P;
  return run_filter(detector, points, count, inliers, state_dev, workspace, vk_s(stream), P);
}

VK_API int vk_detect(const vk_detector* detector, const float* points, int32_t count,
    float* inliers, vk_detect_state* state_dev, void* workspace, void* stream)
{
  DetectParams P;
  const int rc = run_filter(detector, points, count, inliers, state_dev, workspace, vk_s(stream), P);
  if (rc != VK_OK) return rc;
  const hipStream_t s = vk_s(stream);
  if (count > 0)
  {
    P.in = inliers;
    hipLaunchKernelGGL(detect_partials_kernel<0>, dim3(P.chunks), dim3(kLanes), 0, s, P,
        &state_dev->inlier_count);
  }
  else
  {
    // no kernel above ran: the position kernel only needs the zeroed state
    P.partials = nullptr;
  }
  hipLaunchKernelGGL(detect_position_kernel, dim3(1), dim3(64), 0, s, P);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

}  // extern "C"
