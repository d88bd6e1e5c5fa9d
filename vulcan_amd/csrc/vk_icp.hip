// vk_icp.hip — projective point-to-plane ICP normal system, Gauss-Newton update
// and the image pyramid for gfx950 (ref: src/depth_tracker.cu,
// src/tracker.cpp:124-163, src/depth_tracker.cpp:22-86, src/image.cu:101-165).
//
// The reference reduces 27 sums with 9 rounds of three 256-wide LDS trees per
// workgroup and 27 float atomicAdds per workgroup (order-nondeterministic), then
// copies 42 floats to the host every Gauss-Newton iteration for an Eigen LDLT.
// Here: registers -> wave64 butterfly -> one LDS hop -> per-group partials, summed in a
// fixed order (bit-reproducible, independent of the device), and the 6x6 solve + SE(3)
// update run on the device: Track() is one launch for the whole loop (track_loop_kernel).
#include "vk_gauss_newton.hpp"

#include <time.h>
#include "vk_rig_protocol.h"

#include <string.h>

using namespace vk;

namespace
{

struct View
{
  const float* depths;
  const float* normals;
  int width, height;
  vk_projection k;
};

struct IcpParams
{
  View key, frm;
  Rt Twm, Tmw, Twc;
  const vk_transform* Twc_dev;  // optional device override of Twc
  const int32_t* state;         // optional {iterations, converged}: a converged solve skips the pass
  int group_pixels;             // frame pixels per partial sum (group_pixels_for)
};

__device__ __forceinline__ Rt rt_from_colmajor(const float* m)
{
  Rt t;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) t.r[r * 4 + c] = m[c * 4 + r];
  return t;
}

// ref: depth_tracker.cu:18-94 Evaluate<translation_enabled>; returns false when
// the pixel contributes nothing (residual 0, Jacobian 0).
// what a frame pixel contributes that does not depend on the pose being solved for: its
// depth and normal (loaded before the pose is known, in the one-launch-per-step kernel)
struct FramePixel
{
  float depth;
  f3 normal;
};

__device__ __forceinline__ FramePixel load_frame_pixel(const View& frm, int frame_x, int frame_y)
{
  FramePixel px;
  px.depth = 0.0f;
  px.normal = make3(0, 0, 0);
  if (frame_x < frm.width && frame_y < frm.height)
  {
    const int frame_index = frame_y * frm.width + frame_x;
    px.depth = frm.depths[frame_index];
    const vf3 n = *reinterpret_cast<const vf3*>(frm.normals + 3 * frame_index);
    px.normal = make3(n.x, n.y, n.z);
  }
  return px;
}

template <bool TRANSLATION>
__device__ __forceinline__ bool evaluate(const IcpParams& P, const Rt& Twc, int frame_x, int frame_y,
    const FramePixel& px, float& residual, float J[6])
{
  residual = 0.0f;
#pragma unroll
  for (int i = 0; i < 6; ++i) J[i] = 0.0f;

  const View& frm = P.frm;
  const View& key = P.key;
  if (!(frame_x < frm.width && frame_y < frm.height)) return false;

  const float frame_depth = px.depth;
  if (!(frame_depth > 0)) return false;

  const f3 Xcp = unproject_d(frm.k, frame_x + 0.5f, frame_y + 0.5f, frame_depth);
  const f3 Xwp = xform_point(Twc, Xcp);
  const f3 Xmp = xform_point(P.Tmw, Xwp);
  float ku, kv;
  project(key.k, Xmp, ku, kv);
  if (!(ku >= 0 && ku < key.width && kv >= 0 && kv < key.height)) return false;

  const int keyframe_index = (int)kv * key.width + (int)ku;
  const float keyframe_depth = key.depths[keyframe_index];
  if (!(keyframe_depth > 0)) return false;

  const f3 frame_normal = xform_dir(Twc, px.normal);
  const vf3 kn = *reinterpret_cast<const vf3*>(key.normals + 3 * keyframe_index);   // one 12-byte load
  f3 keyframe_normal = make3(kn.x, kn.y, kn.z);
  keyframe_normal = xform_dir(P.Twm, keyframe_normal);

  if (!(sqnorm3(keyframe_normal) > 0.0f && dot3(frame_normal, keyframe_normal) > 0.5f)) return false;

  const f3 Ymp = unproject_d(key.k, floorf(ku) + 0.5f, floorf(kv) + 0.5f, keyframe_depth);
  const f3 Ywp = xform_point(P.Twm, Ymp);
  const f3 delta = sub3(Xwp, Ywp);
  if (!(sqnorm3(delta) < 0.05f)) return false;

  residual = dot3(delta, keyframe_normal);
  J[0] = keyframe_normal.z * Xwp.y - keyframe_normal.y * Xwp.z;
  J[1] = keyframe_normal.x * Xwp.z - keyframe_normal.z * Xwp.x;
  J[2] = keyframe_normal.y * Xwp.x - keyframe_normal.x * Xwp.y;
  if (TRANSLATION)
  {
    J[3] = keyframe_normal.x;
    J[4] = keyframe_normal.y;
    J[5] = keyframe_normal.z;
  }
  return true;
}

template <bool TRANSLATION>
__device__ __forceinline__ bool evaluate(const IcpParams& P, const Rt& Twc, int frame_x, int frame_y,
    float& residual, float J[6])
{
  return evaluate<TRANSLATION>(P, Twc, frame_x, frame_y, load_frame_pixel(P.frm, frame_x, frame_y), residual, J);
}

// ref: depth_tracker.cu:97-118
__global__ __launch_bounds__(256) void residuals_kernel(IcpParams P, float* __restrict__ residuals)
{
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= P.frm.width || y >= P.frm.height) return;
  float r, J[6];
  evaluate<false>(P, P.Twc, x, y, r, J);
  residuals[y * P.frm.width + x] = r;
}

// ref: depth_tracker.cu:120-141
template <bool TRANSLATION>
__global__ __launch_bounds__(256) void jacobian_kernel(IcpParams P, float* __restrict__ jacobian)
{
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= P.frm.width || y >= P.frm.height) return;
  float r, J[6];
  evaluate<TRANSLATION>(P, P.Twc, x, y, r, J);
  float* out = jacobian + 6 * (size_t)(y * P.frm.width + x);
#pragma unroll
  for (int i = 0; i < 6; ++i) out[i] = J[i];
}

__device__ void solve_update(const float* hessian, const float* gradient, int translation_enabled,
    vk_transform* Twc, int32_t* state, float* update_out, Mirror mirror);

// ref: depth_tracker.cu:144-268. Slot layout of a partial: [0,6) J^T r,
// [6,27) packed lower triangle of J^T J in (r, c<=r) row-major order.
//
// Measured and rejected: letting the workgroup that finishes last (atomic ticket)
// also run the second stage, for one launch per iteration instead of two. On this
// multi-XCD part the agent-scope release/acquire fences that publish the partials
// write back and invalidate whole L2s, once per workgroup: the fused iteration
// took 2.5x as long as the two launches.
// pixel `trip` of this lane within a group (kIcpThreads apart, so that every load is
// coalesced); -1 beyond the group or the image
__device__ __forceinline__ int lane_pixel(const IcpParams& P, int group, int trip)
{
  const int within = trip * kIcpThreads + (int)threadIdx.x;
  const int pixel = group * P.group_pixels + within;
  return (within < P.group_pixels && pixel < P.frm.width * P.frm.height) ? pixel : -1;
}

// evaluate() in two halves, so that a lane with several pixels can have all its keyframe
// loads in flight at once: everything up to the keyframe pixel's address ...
struct Candidate
{
  f3 Xwp, frame_normal;
  float ku, kv;
  int keyframe_index;   // 0 when the pixel has no candidate (the load is issued all the same)
  bool ok;
};

__device__ __forceinline__ Candidate prepare_pixel(const IcpParams& P, const Rt& Twc, int frame_x, int frame_y,
    bool inside, const FramePixel& px)
{
  Candidate c;
  c.ok = false;
  c.keyframe_index = 0;
  c.ku = c.kv = 0.0f;
  c.Xwp = c.frame_normal = make3(0, 0, 0);
  if (!inside || !(px.depth > 0)) return c;
  const f3 Xcp = unproject_d(P.frm.k, frame_x + 0.5f, frame_y + 0.5f, px.depth);
  c.Xwp = xform_point(Twc, Xcp);
  const f3 Xmp = xform_point(P.Tmw, c.Xwp);
  project(P.key.k, Xmp, c.ku, c.kv);
  if (!(c.ku >= 0 && c.ku < P.key.width && c.kv >= 0 && c.kv < P.key.height)) return c;
  c.keyframe_index = (int)c.kv * P.key.width + (int)c.ku;
  c.frame_normal = xform_dir(Twc, px.normal);
  c.ok = true;
  return c;
}

// ... and everything after the keyframe's depth and normal have arrived (same operations
// in the same order as evaluate())
template <bool TRANSLATION>
__device__ __forceinline__ bool finish_pixel(const IcpParams& P, const Candidate& c, float keyframe_depth, const vf3& kn,
    float& residual, float J[6])
{
  if (!c.ok || !(keyframe_depth > 0)) return false;
  f3 keyframe_normal = make3(kn.x, kn.y, kn.z);
  keyframe_normal = xform_dir(P.Twm, keyframe_normal);
  if (!(sqnorm3(keyframe_normal) > 0.0f && dot3(c.frame_normal, keyframe_normal) > 0.5f)) return false;
  const f3 Ymp = unproject_d(P.key.k, floorf(c.ku) + 0.5f, floorf(c.kv) + 0.5f, keyframe_depth);
  const f3 Ywp = xform_point(P.Twm, Ymp);
  const f3 delta = sub3(c.Xwp, Ywp);
  if (!(sqnorm3(delta) < 0.05f)) return false;
  residual = dot3(delta, keyframe_normal);
  J[0] = keyframe_normal.z * c.Xwp.y - keyframe_normal.y * c.Xwp.z;
  J[1] = keyframe_normal.x * c.Xwp.z - keyframe_normal.z * c.Xwp.x;
  J[2] = keyframe_normal.y * c.Xwp.x - keyframe_normal.x * c.Xwp.y;
  J[3] = J[4] = J[5] = 0.0f;
  if (TRANSLATION)
  {
    J[3] = keyframe_normal.x;
    J[4] = keyframe_normal.y;
    J[5] = keyframe_normal.z;
  }
  return true;
}

// the 27 products of kIcpPixels of the lane's pixels of one group (trips first ...), added
// pixel by pixel onto acc
template <bool TRANSLATION>
__device__ __forceinline__ void accumulate_pixels(const IcpParams& P, const Rt& Twc, int group, int first,
    const FramePixel (&px)[kIcpPixels], float (&acc)[27])
{
  Candidate cand[kIcpPixels];
  float key_depth[kIcpPixels];
  vf3 key_normal[kIcpPixels];
#pragma unroll
  for (int k = 0; k < kIcpPixels; ++k)
  {
    const int pixel = lane_pixel(P, group, first + k);
    const int safe = pixel < 0 ? 0 : pixel;
    cand[k] = prepare_pixel(P, Twc, safe % P.frm.width, safe / P.frm.width, pixel >= 0, px[k]);
  }
#pragma unroll
  for (int k = 0; k < kIcpPixels; ++k)
  {
    key_depth[k] = P.key.depths[cand[k].keyframe_index];
    key_normal[k] = *reinterpret_cast<const vf3*>(P.key.normals + 3 * cand[k].keyframe_index);
  }
#pragma unroll
  for (int k = 0; k < kIcpPixels; ++k)
  {
    float r, J[6], one[27];
    if (finish_pixel<TRANSLATION>(P, cand[k], key_depth[k], key_normal[k], r, J))
    {
      outer_products(J, r, one);
#pragma unroll
      for (int i = 0; i < 27; ++i) acc[i] += one[i];
    }
  }
}

__device__ __forceinline__ void load_pixels(const IcpParams& P, int group, int first, FramePixel (&px)[kIcpPixels])
{
#pragma unroll
  for (int k = 0; k < kIcpPixels; ++k)
  {
    const int pixel = lane_pixel(P, group, first + k);
    px[k] = load_frame_pixel(P.frm, pixel < 0 ? 0 : pixel % P.frm.width, pixel < 0 ? P.frm.height : pixel / P.frm.width);
  }
}

// a group's trips, kIcpPixels at a time
template <bool TRANSLATION>
__device__ __forceinline__ void accumulate_group(const IcpParams& P, const Rt& Twc, int group, float (&acc)[27])
{
  const int trips = (P.group_pixels + kIcpThreads - 1) / kIcpThreads;
  for (int first = 0; first < trips; first += kIcpPixels)
  {
    FramePixel px[kIcpPixels];
    load_pixels(P, group, first, px);
    accumulate_pixels<TRANSLATION>(P, Twc, group, first, px, acc);
  }
}

template <bool TRANSLATION>
__global__ __launch_bounds__(kIcpThreads) void system_partial_kernel(IcpParams P, float* __restrict__ workspace)
{
  __shared__ float lds[kIcpThreads / 64][kSysStride];

  // tracker.cpp:162: once the update norm fell below 1e-6 the reference leaves its
  // loop; here the remaining (already enqueued) iterations turn into empty launches
  if (P.state && P.state[1]) return;

  const Rt Twc = P.Twc_dev ? rt_from_colmajor(P.Twc_dev->m) : P.Twc;
  float acc[27];
#pragma unroll
  for (int i = 0; i < 27; ++i) acc[i] = 0.0f;
  accumulate_group<TRANSLATION>(P, Twc, blockIdx.x, acc);
  store_partial<kIcpThreads / 64>(acc, lds, workspace);
}

// Second stage: one workgroup. With `Twc` non-null it also solves and updates the
// pose, so a Gauss-Newton iteration is two launches.
__global__ __launch_bounds__(256) void system_final_kernel(const float* __restrict__ workspace,
    int partials, int translation_enabled, float* __restrict__ hessian, float* __restrict__ gradient,
    vk_transform* Twc, int32_t* state, float* update_out, Mirror mirror)
{
  __shared__ float slices[kSysSlices][kSysStride];
  __shared__ float sums[48];   // hessian[36] | gradient[6]: the solve reads them from LDS
  if (state && state[1]) return;   // converged: the system was not recomputed, keep the last one
  sum_partials(workspace, partials, translation_enabled, hessian, gradient, slices, sums);
  if (Twc && threadIdx.x == 0) solve_update(sums, sums + 36, translation_enabled, Twc, state, update_out, mirror);
}

// ---- pose update on the device ------------------------------------------------

// ref: tracker.cpp:124-163 + depth_tracker.cpp:22-86. One lane; 6x6 is too
// small to spread.
// the new pose (out_m, out_i) from the system and the old pose matrix
// M = Tinc(update) * old pose matrix; the new pose is rigid_from(M)
template <int N>
__device__ __forceinline__ void pose_matrix(const float* hessian, const float* gradient, const float (&old_m)[16],
    float (&M)[16], float (&update)[6])
{
  solve_step<N>(hessian, gradient, update);

  // depth_tracker.cpp:33-53, including Tinc(1,2) = +update[0]
  float Tinc[16];
  Tinc[0] = 1.0f;        Tinc[4] = -update[2]; Tinc[8] = +update[1];  Tinc[12] = +update[3];
  Tinc[1] = +update[2];  Tinc[5] = 1.0f;       Tinc[9] = +update[0];  Tinc[13] = +update[4];
  Tinc[2] = -update[1];  Tinc[6] = +update[0]; Tinc[10] = 1.0f;       Tinc[14] = +update[5];
  Tinc[3] = 0.0f;        Tinc[7] = 0.0f;       Tinc[11] = 0.0f;       Tinc[15] = 1.0f;

  matmul4(Tinc, old_m, M);
}

template <int N>
__device__ __forceinline__ void pose_step(const float* hessian, const float* gradient, const float (&old_m)[16],
    float (&out_m)[16], float (&out_i)[16], float (&update)[6])
{
  float M[16];
  pose_matrix<N>(hessian, gradient, old_m, M, update);
  rigid_from(M, out_m, out_i);
}

template <int N>
__device__ __forceinline__ void solve_update_n(const float* hessian, const float* gradient,
    vk_transform* Twc, int32_t* state, float* update_out, Mirror mirror)
{
  float update[6], old_m[16], out_m[16], out_i[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) old_m[i] = Twc->m[i];
  pose_step<N>(hessian, gradient, old_m, out_m, out_i, update);
#pragma unroll
  for (int i = 0; i < 16; ++i) { Twc->m[i] = out_m[i]; Twc->inv[i] = out_i[i]; }
  finish_step<N>(update, state, update_out, mirror);
}

__device__ void solve_update(const float* hessian, const float* gradient,
    int translation_enabled, vk_transform* Twc, int32_t* state, float* update_out, Mirror mirror)
{
  if (state && state[1]) return;  // converged earlier: tracker.cpp:162
  if (translation_enabled) solve_update_n<6>(hessian, gradient, Twc, state, update_out, mirror);
  else solve_update_n<3>(hessian, gradient, Twc, state, update_out, mirror);
}

__global__ void solve_update_kernel(const float* __restrict__ hessian, const float* __restrict__ gradient,
    int translation_enabled, vk_transform* Twc, int32_t* state, float* update_out, Mirror mirror)
{
  if (threadIdx.x == 0 && blockIdx.x == 0)
    solve_update(hessian, gradient, translation_enabled, Twc, state, update_out, mirror);
}

// ---- the whole Gauss-Newton loop in one launch ------------------------------------
//
// History: two launches per step (r01: partials, then sum + solve), one launch per step
// (r02: every workgroup finishes the previous step itself), and now one launch per LOOP:
// the workgroups exchange their sums inside the launch (vk_gauss_newton.hpp, "partials
// exchanged inside a launch"), every workgroup adds all of them in the fixed order and
// solves the 6x6 system itself — same instructions, same inputs, bit-identical poses
// everywhere — and goes on to the next step at the new pose. What that removes per step:
// the launch (~4.5 us on this part), the cold start of the caches (the images now stay
// in L2 for the whole loop; the frame pixels of a lane stay in its registers), the
// empty launches after convergence and the host's polling for it (tracker.cpp:162 is a
// `break` again). Workgroup 0 alone publishes pose, system and state, once, at the end.
struct LoopParams
{
  Exchange exchange;             // {tag, value} words of the launch's workgroups
  VK_LOOP_TIMING_FIELD
  vk_transform* pose;            // in: the pose to start from; out: the pose after the loop
  int groups;                    // 1024-pixel groups of the frame (gridDim.x <= groups)
  int iterations;
  int fresh_state;               // 1: the loop starts at {0 steps, not converged} whatever `state` holds;
                                 // 2: the same, unless an earlier level of this Track was aborted
  int force_abort;               // test aid, vk_forced_loop_abort()
  vk_rig_exchange rig;           // world > 0: the sums are added over the ranks of a rig after every step
  int last_launch;               // 1: this launch ends the Track (it leaves the pose for vk_track_wait)
  float* hessian;
  float* gradient;
  int32_t* state;
  float* update_out;
  Mirror mirror;
};

template <bool TRANSLATION>
__global__ __launch_bounds__(kIcpThreads) void track_loop_kernel(IcpParams P, LoopParams L)
{
  constexpr int N = TRANSLATION ? 6 : 3;
  __shared__ float lds[kIcpThreads / 64][kSysStride];
  __shared__ float slices[kSysSlices][kSysStride];
  __shared__ float sums[48];
  __shared__ float pose_m[16];
  __shared__ float result[16 + 6];        // workgroup 0: M (see below) and update of the last step
  __shared__ float solve_scratch[64];     // wave_solve_step / wave_rigid_from
  __shared__ int stop, failed;

  // tracker.cpp:162 / Tracker::CreateState: a state that already says "converged" ends the call
  const int steps_before = L.fresh_state ? 0 : L.state[0];
  if (!L.fresh_state && L.state[1])           // uniform over the grid: nobody waits for anybody
  {
    if (blockIdx.x == 0 && L.last_launch && L.state[1] != VK_TRACK_ABORTED) publish_host_pose(L.mirror, L.pose);
    return;
  }
  // an aborted level ends the whole Track: the next level must not start from the pose it left
  // behind and report success (the host then runs the Track again, launch per stage)
  if (L.fresh_state == 2 && L.state[1] == VK_TRACK_ABORTED) return;
  if (L.force_abort)
  {
    if (blockIdx.x == 0 && threadIdx.x == 0) L.state[1] = VK_TRACK_ABORTED;
    return;
  }

  // with one group of at most kIcpPixels trips per workgroup the lane's frame pixels never
  // change: loaded once, kept in registers over all steps
  const bool resident = (int)gridDim.x >= L.groups && P.group_pixels <= kIcpPixels * kIcpThreads;
  FramePixel px[kIcpPixels];
  if (resident) load_pixels(P, blockIdx.x, 0, px);

  if (threadIdx.x < 16) pose_m[threadIdx.x] = L.pose->m[threadIdx.x];
  if (threadIdx.x == 0) { stop = 0; failed = 0; }
  __syncthreads();

  const bool publisher = blockIdx.x == 0;
  int steps = 0;
  for (int it = 0; it < L.iterations; ++it)
  {
    VK_STAMP(0);
    const Rt Twc = rt_from_colmajor(pose_m);
    for (int group = blockIdx.x; group < L.groups; group += gridDim.x)
    {
      if (group != (int)blockIdx.x) __syncthreads();   // the previous group's sums have left the LDS
      float acc[27];
#pragma unroll
      for (int i = 0; i < 27; ++i) acc[i] = 0.0f;
      if (resident) accumulate_pixels<TRANSLATION>(P, Twc, group, 0, px, acc);
      else accumulate_group<TRANSLATION>(P, Twc, group, acc);
      VK_STAMP(1);
#ifdef VK_LOOP_ATOMIC_EXCHANGE
      publish_atomic<kIcpThreads / 64>(acc, lds, L.exchange, it);
#else
      publish_partial<kIcpThreads / 64>(acc, lds, L.exchange, it, group);
#endif
    }
    VK_STAMP(2);
    VK_STAMP(3);
#ifdef VK_LOOP_ATOMIC_EXCHANGE
    if (!gather_atomic(L.exchange, it, TRANSLATION, publisher ? L.hessian : nullptr, publisher ? L.gradient : nullptr, sums, &failed))
      break;
#else
    if (!gather_partials<kIcpThreads>(L.exchange, it, TRANSLATION, publisher ? L.hessian : nullptr,
            publisher ? L.gradient : nullptr, slices, sums, &failed))
      break;
#endif
    if (L.rig.world > 0)
    {
      // a rigid rig: this view's sums go to every rank, every rank's come back (vk_rig_protocol.h);
      // `sums` holds the packed hessian at [0, 21) and the gradient at [36, 42)
      if (publisher && threadIdx.x < VK_RIG_VALUES)
        rig_publish(L.rig.areas, L.rig.rank, L.rig.world, L.rig.sequence, it, (int)threadIdx.x,
            threadIdx.x < 6 ? sums[36 + threadIdx.x] : sums[threadIdx.x - 6],
            [](unsigned long long* at, unsigned long long word) { __hip_atomic_store(at, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); });
      if (threadIdx.x < VK_RIG_VALUES)
      {
        const unsigned long long deadline = (unsigned long long)wall_clock64() + kExchangeTimeout;
        float total = 0.0f;
        const bool arrived = rig_gather(L.rig.areas[L.rig.rank], L.rig.world, L.rig.sequence, it, (int)threadIdx.x, total,
            [](const unsigned long long* at) { return __hip_atomic_load(at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); },
            [deadline] { __builtin_amdgcn_s_sleep(VK_POLL_GAP); return (unsigned long long)wall_clock64() > deadline; });
        if (!arrived) failed = 1;
        else
        {
          sums[threadIdx.x < 6 ? 36 + threadIdx.x : threadIdx.x - 6] = total;
          if (publisher)
          {
            if (threadIdx.x < 6) { if (L.gradient) L.gradient[threadIdx.x] = total; }
            else if (L.hessian) L.hessian[threadIdx.x - 6] = total;
          }
        }
      }
      __syncthreads();
      if (failed) break;
    }
    steps = it + 1;
    VK_STAMP(4);

#ifdef VK_SCALAR_SOLVE
    if (threadIdx.x == 0)
    {
      // the pixels only ever need the pose's matrix; its inverse (a second 4x4 product per
      // step) is made once, after the loop, from the last step's M
      float update[6], old_m[16], M[16], out_m[16], unused_i[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) old_m[i] = pose_m[i];
      pose_matrix<N>(sums, sums + 36, old_m, M, update);
      rigid_from(M, out_m, unused_i);
      float sq = 0.0f;
#pragma unroll
      for (int i = 0; i < N; ++i) sq += update[i] * update[i];
      stop = (sqrtf(sq) < 1E-6f) ? 1 : 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) pose_m[i] = out_m[i];
      if (publisher)
      {
#pragma unroll
        for (int i = 0; i < 16; ++i) result[i] = M[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) result[16 + i] = update[i];
      }
    }
#else
    if (threadIdx.x < 64)
    {
      // solve + pose update across the lanes of the first wave (wave_solve_step): the bits of
      // pose_matrix<N> + rigid_from on one lane. The pixels only ever need the pose's matrix; its
      // inverse (a second 4x4 product per step) is made once, after the loop, from the last M.
      float update[6];
      wave_solve_step<N>(sums, solve_scratch, update);
      // depth_tracker.cpp:33-53, including Tinc(1,2) = +update[0]; element l = c * 4 + r
      const int l = (int)threadIdx.x & 15;
      float tinc = (l % 5 == 0) ? 1.0f : 0.0f;
      tinc = (l == 4) ? -update[2] : tinc;  tinc = (l == 8) ? +update[1] : tinc;  tinc = (l == 12) ? +update[3] : tinc;
      tinc = (l == 1) ? +update[2] : tinc;  tinc = (l == 9) ? +update[0] : tinc;  tinc = (l == 13) ? +update[4] : tinc;
      tinc = (l == 2) ? -update[1] : tinc;  tinc = (l == 6) ? +update[0] : tinc;  tinc = (l == 14) ? +update[5] : tinc;
      if (threadIdx.x < 16) solve_scratch[threadIdx.x] = tinc;
      wave_lds_fence();
      const float M_lane = matmul4_lane(solve_scratch, pose_m, (int)threadIdx.x);     // Tinc * old pose
      wave_lds_fence();
      const float out = wave_rigid_from(M_lane, solve_scratch);
      float sq = 0.0f;
#pragma unroll
      for (int i = 0; i < N; ++i) sq += update[i] * update[i];
      if (threadIdx.x < 16)
      {
        pose_m[threadIdx.x] = out;
        if (publisher) result[threadIdx.x] = M_lane;
      }
      if (threadIdx.x == 0)
      {
        stop = (sqrtf(sq) < 1E-6f) ? 1 : 0;
        if (publisher)
        {
#pragma unroll
          for (int i = 0; i < 6; ++i) result[16 + i] = update[i];
        }
      }
    }
#endif
    __syncthreads();
    VK_STAMP(5);
    if (stop) break;             // tracker.cpp:162
  }

  if (failed)
  {
    // some workgroup's sums never came (see kExchangeTimeout): every workgroup ends up here
    if (threadIdx.x == 0) L.state[1] = VK_TRACK_ABORTED;
    return;
  }
  if (!publisher) return;
  if (steps > 0)
  {
    if (threadIdx.x == 0)
    {
      float M[16], out_m[16], out_i[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) M[i] = result[i];
      rigid_from(M, out_m, out_i);
#pragma unroll
      for (int i = 0; i < 16; ++i) { L.pose->m[i] = out_m[i]; L.pose->inv[i] = out_i[i]; }
    }
    if (threadIdx.x < 6 && L.update_out) L.update_out[threadIdx.x] = result[16 + threadIdx.x];
    if (threadIdx.x == 0)
    {
      const int iterations = steps_before + steps;
      L.state[0] = iterations;
      L.state[1] = stop;
      if (L.mirror.word)
        __hip_atomic_store(L.mirror.word, ((unsigned long long)(L.mirror.epoch & 0xffffu) << 48) |
            ((unsigned long long)(uint32_t)(stop & 1) << 32) | (uint32_t)iterations,
            __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (L.last_launch)
  {
    __syncthreads();   // lane 0's pose stores are visible to the 32 lanes that copy them out
    publish_host_pose(L.mirror, L.pose);
  }
}

// ------------------------------------------------------------------ pyramid ----

// ref: image.cu:101-131
__global__ __launch_bounds__(256) void downsample_kernel(int src_w, int dst_w, int dst_h,
    const float* __restrict__ src, float* __restrict__ dst, int nearest)
{
  const int dst_x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int dst_y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (dst_x >= dst_w || dst_y >= dst_h) return;

  const int src_x = 2 * dst_x, src_y = 2 * dst_y;
  float sample = 0;

  if (nearest)
  {
    sample = src[src_y * src_w + src_x];
  }
  else
  {
    sample += src[(src_y + 0) * src_w + (src_x + 1)];
    sample += src[(src_y + 0) * src_w + (src_x + 0)];
    sample += src[(src_y + 1) * src_w + (src_x + 1)];
    sample += src[(src_y + 1) * src_w + (src_x + 0)];
    sample *= 0.25f;
  }

  dst[dst_y * dst_w + dst_x] = sample;
}

// ref: image.cu:133-165
__global__ __launch_bounds__(256) void downsample3_kernel(int src_w, int dst_w, int dst_h,
    const float* __restrict__ src, float* __restrict__ dst, int nearest)
{
  const int dst_x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int dst_y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (dst_x >= dst_w || dst_y >= dst_h) return;

  const int src_x = 2 * dst_x, src_y = 2 * dst_y;

#pragma unroll
  for (int c = 0; c < 3; ++c)
  {
    float sample = 0;
    if (nearest)
    {
      sample = src[3 * (src_y * src_w + src_x) + c];
    }
    else
    {
      sample += src[3 * ((src_y + 0) * src_w + (src_x + 1)) + c];
      sample += src[3 * ((src_y + 0) * src_w + (src_x + 0)) + c];
      sample += src[3 * ((src_y + 1) * src_w + (src_x + 1)) + c];
      sample += src[3 * ((src_y + 1) * src_w + (src_x + 0)) + c];
      sample *= 0.25f;
    }
    dst[3 * (dst_y * dst_w + dst_x) + c] = sample;
  }
}

// Frame::Downsample (frame.cpp:38-51) as one launch: blockIdx.z = 0 depth (nearest), 1 colour
// (2x2 box), 2 normals (nearest); the same expressions as downsample_kernel / downsample3_kernel
struct FrameLevel
{
  const float* src[3];
  float* dst[3];
  int src_w[3], dst_w[3], dst_h[3];
};

__global__ __launch_bounds__(256) void frame_downsample_kernel(FrameLevel L)
{
  const int job = blockIdx.z;
  if (!L.dst[job]) return;
  const int dst_x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int dst_y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (dst_x >= L.dst_w[job] || dst_y >= L.dst_h[job]) return;
  const int src_w = L.src_w[job], src_x = 2 * dst_x, src_y = 2 * dst_y;
  const float* src = L.src[job];
  float* dst = L.dst[job];
  if (job == 0)
  {
    dst[dst_y * L.dst_w[job] + dst_x] = src[src_y * src_w + src_x];
  }
  else if (job == 2)
  {
    const vf3 n = *reinterpret_cast<const vf3*>(src + 3 * (src_y * src_w + src_x));
    *reinterpret_cast<vf3*>(dst + 3 * (dst_y * L.dst_w[job] + dst_x)) = n;
  }
  else
  {
#pragma unroll
    for (int c = 0; c < 3; ++c)
    {
      float sample = 0;
      sample += src[3 * ((src_y + 0) * src_w + (src_x + 1)) + c];
      sample += src[3 * ((src_y + 0) * src_w + (src_x + 0)) + c];
      sample += src[3 * ((src_y + 1) * src_w + (src_x + 1)) + c];
      sample += src[3 * ((src_y + 1) * src_w + (src_x + 0)) + c];
      sample *= 0.25f;
      dst[3 * (dst_y * L.dst_w[job] + dst_x) + c] = sample;
    }
  }
}

// One pyramid level of BOTH sides of a depth-tracking problem in one launch (blockIdx.z:
// keyframe / frame): nearest depth and nearest normals, exactly Image::Downsample(nearest)
// and ColorImage::Downsample(nearest) of Frame::Downsample (frame.cpp:49-51). The colour
// image Frame::Downsample also halves is not an input of DepthTracker and is not touched.
struct LevelParams
{
  const float* src_depth[2];
  const float* src_normals[2];
  float* dst_depth[2];
  float* dst_normals[2];
  int src_w[2], dst_w[2], dst_h[2];
  // vk_icp_pyramid_track_frame: the FRAME's normal image is still to be computed (Frame::ComputeNormals, frame.cu:9-122) —
  // blockIdx.z == 2 writes it, and the frame's half-resolution normals are computed at the pixels they are sampled from
  // instead of copied (nearest sampling: the same pixel's normal, bit for bit) — and the pose the Track starts from is
  // stored by workgroup 0 (vk_transform_upload's launch)
  // (round 5) the same for the KEYFRAME — the raycast's normal image, which Tracer::Trace otherwise computes with a launch
  // of its own right behind the raycast (tracer.cpp:97-100) and which nobody reads before this Track: side 0
  float* normals_out[2];           // [0] keyframe, [1] frame; nullptr: that side's normals exist
  int src_h[2];
  vk_projection k[2];
  int due_side[2];                 // blockIdx.z == 2 + i computes the whole normal image of side due_side[i]
  vk_transform* pose_out;          // nullptr: the pose is on the device already
  vk_transform pose_start;
};

__device__ __forceinline__ float level_depth_at(const float* depths, int w, int h, int x, int y)
{
  return (x >= 0 && x < w && y >= 0 && y < h) ? depths[y * w + x] : 0.0f;
}

// the normal of pixel (x, y) of a full-resolution depth image: compute_normals_kernel's expressions (vk_trace.hip)
__device__ __forceinline__ f3 level_normal(const float* depths, const vk_projection& k, int w, int h, int x, int y)
{
  const int pad = 2;
  const float depth = depths[y * w + x];
  f3 normal = make3(0, 0, 0);
  if (depth > 0)
    normal = normal_from_taps(k, x, y, depth, level_depth_at(depths, w, h, x - pad, y), level_depth_at(depths, w, h, x + pad, y),
        level_depth_at(depths, w, h, x, y - pad), level_depth_at(depths, w, h, x, y + pad));
  return normal;
}

__global__ __launch_bounds__(256) void pyramid_level_kernel(LevelParams L)
{
  if (L.pose_out && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x < 32)
  {
    if (threadIdx.x < 16) L.pose_out->m[threadIdx.x] = L.pose_start.m[threadIdx.x];
    else L.pose_out->inv[threadIdx.x - 16] = L.pose_start.inv[threadIdx.x - 16];
  }
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (blockIdx.z >= 2)
  {
    // a side's own normal image, every pixel
    const int side = L.due_side[blockIdx.z - 2];
    if (x >= L.src_w[side] || y >= L.src_h[side]) return;
    const f3 n = level_normal(L.src_depth[side], L.k[side], L.src_w[side], L.src_h[side], x, y);
    float* out = L.normals_out[side] + 3 * ((size_t)y * L.src_w[side] + x);
    out[0] = n.x;  out[1] = n.y;  out[2] = n.z;
    return;
  }
  const int side = blockIdx.z;
  const int dst_x = x, dst_y = y;
  if (dst_x >= L.dst_w[side] || dst_y >= L.dst_h[side]) return;
  const int src = (2 * dst_y) * L.src_w[side] + 2 * dst_x;
  const int dst = dst_y * L.dst_w[side] + dst_x;
  L.dst_depth[side][dst] = L.src_depth[side][src];
  vf3 n;
  if (L.normals_out[side])
  {
    const f3 computed = level_normal(L.src_depth[side], L.k[side], L.src_w[side], L.src_h[side], 2 * dst_x, 2 * dst_y);
    n = vf3{computed.x, computed.y, computed.z};
  }
  else n = *reinterpret_cast<const vf3*>(L.src_normals[side] + 3 * src);
  *reinterpret_cast<vf3*>(L.dst_normals[side] + 3 * dst) = n;
}

int fill_icp(IcpParams& P, const vk_icp_view* keyframe, const vk_transform* Twm,
    const vk_icp_view* frame, const vk_transform* Twc)
{
  if (!keyframe || !Twm || !frame || !Twc) return VK_ERR_ARGUMENT;
  if (!keyframe->depths || !keyframe->normals || !frame->depths || !frame->normals) return VK_ERR_ARGUMENT;
  if (keyframe->width <= 0 || keyframe->height <= 0 || frame->width <= 0 || frame->height <= 0)
    return VK_ERR_ARGUMENT;
  P.key.depths = keyframe->depths;
  P.key.normals = keyframe->normals;
  P.key.width = keyframe->width;
  P.key.height = keyframe->height;
  P.key.k = keyframe->projection;
  P.frm.depths = frame->depths;
  P.frm.normals = frame->normals;
  P.frm.width = frame->width;
  P.frm.height = frame->height;
  P.frm.k = frame->projection;
  P.Twm = make_rt(Twm->m);
  P.Tmw = make_rt(Twm->inv);
  P.Twc = make_rt(Twc->m);
  P.Twc_dev = nullptr;
  P.state = nullptr;
  P.group_pixels = group_pixels_for(frame->width * frame->height);
  return VK_OK;
}

void launch_partials(const IcpParams& P, int translation_enabled, int partials, float* workspace, hipStream_t s)
{
  vk_loop_area_written(workspace);      // float partials over the loop kernels' tagged words: the next loop launch clears them
  if (translation_enabled)
    hipLaunchKernelGGL(system_partial_kernel<true>, dim3(partials), dim3(kIcpThreads), 0, s, P, workspace);
  else
    hipLaunchKernelGGL(system_partial_kernel<false>, dim3(partials), dim3(kIcpThreads), 0, s, P, workspace);
}

// the rig's launch-per-stage loop ends with this when the caller wants the pose in host memory
__global__ void publish_pose_kernel(Mirror mirror, const vk_transform* pose)
{
  publish_host_pose(mirror, pose);
}

// the non-rig loop: one launch (track_loop_kernel)
int launch_loop(const IcpParams& P, vk_transform* Twc_dev, int iterations, int translation_enabled, int groups,
    float* workspace, float* hessian, float* gradient, int32_t* state_dev, float* update_dev, Mirror mirror,
    int fresh_state, bool ends_track, hipStream_t s, const vk_rig_exchange* rig = nullptr)
{
  VK_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 7u) == 0);   // the exchange holds 64-bit words
  const int capacity = translation_enabled ? resident_workgroups(track_loop_kernel<true>, kIcpThreads)
                                           : resident_workgroups(track_loop_kernel<false>, kIcpThreads);
  if (capacity <= 0) return VK_ERR_ARGUMENT;
  const int grid = groups < capacity ? groups : capacity;
  LoopParams L;
  L.pose = Twc_dev;
  L.groups = groups;
  L.hessian = hessian;
  L.gradient = gradient;
  L.state = state_dev;
  L.update_out = update_dev;
  L.mirror = mirror;
  memset(&L.rig, 0, sizeof(L.rig));
  if (rig) L.rig = *rig;
  // ten tag bits name the step: a longer loop continues in another launch (which returns at
  // once if the state says the loop has converged)
  for (int done = 0; done < iterations; done += kExchangeSteps)
  {
    L.exchange.words = reinterpret_cast<unsigned long long*>(workspace);
    L.exchange.count = groups;
    { const int rc = vk_loop_epoch_begin(workspace, exchange_floats(groups) * sizeof(float), s, &L.exchange.epoch);  if (rc != VK_OK) return rc; }
    VK_LOOP_TIMING_ATTACH(L, s);
    L.iterations = iterations - done < kExchangeSteps ? iterations - done : kExchangeSteps;
    L.fresh_state = (fresh_state && done == 0) ? fresh_state : 0;
    L.force_abort = vk_forced_loop_abort();
    L.last_launch = (ends_track && done + kExchangeSteps >= iterations) ? 1 : 0;
    IcpParams Pk = P;
#ifdef VK_LOOP_ATOMIC_EXCHANGE
    // (the experiment's counter and accumulators start from zero; the product's tagged words need no such launch)
    VK_CHECK(hipMemsetAsync(workspace, 0, (16 + 3 * kSysStride) * sizeof(float), s));
#endif
    vk_loop_launch_begin(s);
    const hipError_t le = translation_enabled ? launch_loop_kernel(track_loop_kernel<true>, grid, kIcpThreads, s, Pk, L)
                                              : launch_loop_kernel(track_loop_kernel<false>, grid, kIcpThreads, s, Pk, L);
    vk_loop_launch_end(s);
    VK_CHECK(le);
    VK_LAUNCH_CHECK();
  }
  return VK_OK;
}

}  // namespace

extern "C" {

int vk_icp_compute_residuals(const vk_icp_view* keyframe, const vk_transform* Twm,
    const vk_icp_view* frame, const vk_transform* Twc, float* residuals, void* stream)
{
  IcpParams P;
  const int rc = fill_icp(P, keyframe, Twm, frame, Twc);
  if (rc != VK_OK) return rc;
  VK_REQUIRE(residuals);
  const dim3 grid((frame->width + 63) / 64, (frame->height + 3) / 4);
  hipLaunchKernelGGL(residuals_kernel, grid, dim3(256), 0, vk_s(stream), P, residuals);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

int vk_icp_compute_jacobian(const vk_icp_view* keyframe, const vk_transform* Twm,
    const vk_icp_view* frame, const vk_transform* Twc, int translation_enabled, float* jacobian,
    void* stream)
{
  IcpParams P;
  const int rc = fill_icp(P, keyframe, Twm, frame, Twc);
  if (rc != VK_OK) return rc;
  VK_REQUIRE(jacobian);
  const dim3 grid((frame->width + 63) / 64, (frame->height + 3) / 4);
  if (translation_enabled)
    hipLaunchKernelGGL(jacobian_kernel<true>, grid, dim3(256), 0, vk_s(stream), P, jacobian);
  else
    hipLaunchKernelGGL(jacobian_kernel<false>, grid, dim3(256), 0, vk_s(stream), P, jacobian);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

size_t vk_icp_workspace_floats(int width, int height)
{
  if (width <= 0 || height <= 0) return 0;
  // the in-launch exchange of the loop kernels (two parities of one slot per pixel group; the
  // rig's launch-per-stage loops use the first `groups` * kSysStride floats), sized for the
  // smallest groups any tracker forms: 256 pixels
  return exchange_floats((width * height + 255) / 256) + sizeof(vk_transform) / sizeof(float);
}

int vk_icp_compute_system(const vk_icp_view* keyframe, const vk_transform* Twm,
    const vk_icp_view* frame, const vk_transform* Twc, const vk_transform* Twc_dev,
    int translation_enabled, float* workspace, float* hessian, float* gradient, void* stream)
{
  IcpParams P;
  vk_transform identity;
  if (!Twc && Twc_dev)
  {
    for (int i = 0; i < 16; ++i) identity.m[i] = identity.inv[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    Twc = &identity;
  }
  const int rc = fill_icp(P, keyframe, Twm, frame, Twc);
  if (rc != VK_OK) return rc;
  VK_REQUIRE(workspace && hessian && gradient);
  P.Twc_dev = Twc_dev;
  const int partials = group_count_for(frame->width * frame->height, P.group_pixels);
  launch_partials(P, translation_enabled, partials, workspace, vk_s(stream));
  VK_LAUNCH_CHECK();
  hipLaunchKernelGGL(system_final_kernel, dim3(1), dim3(256), 0, vk_s(stream), workspace, partials,
      translation_enabled, hessian, gradient, (vk_transform*)nullptr, (int32_t*)nullptr, (float*)nullptr,
      Mirror{nullptr, 0, nullptr});
  VK_LAUNCH_CHECK();
  return VK_OK;
}

int vk_icp_track(const vk_icp_view* keyframe, const vk_transform* Twm, const vk_icp_view* frame,
    vk_transform* Twc_dev, int iterations, int translation_enabled, float* workspace, float* system,
    int32_t* state_dev, float* update_dev, vk_icp_reduce_fn reduce, void* reduce_user,
    const vk_track_poll* poll, void* stream)
{
  IcpParams P;
  vk_transform identity;
  for (int i = 0; i < 16; ++i) identity.m[i] = identity.inv[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  const int rc = fill_icp(P, keyframe, Twm, frame, &identity);
  if (rc != VK_OK) return rc;
  VK_REQUIRE(Twc_dev && workspace && system && state_dev && iterations > 0);
  P.Twc_dev = Twc_dev;
  P.state = state_dev;
  float* hessian = system;
  float* gradient = system + 36;
  const int partials = group_count_for(frame->width * frame->height, P.group_pixels);
  hipStream_t s = vk_s(stream);
  const bool chunked = polling(poll);
  const Mirror mirror = begin_mirror(poll);

  if (!reduce)
    return launch_loop(P, Twc_dev, iterations, translation_enabled, partials, workspace, hessian, gradient,
        state_dev, update_dev, mirror, /*fresh_state*/ 0, /*ends_track*/ true, s);

  for (int it = 0; it < iterations; ++it)
  {
    launch_partials(P, translation_enabled, partials, workspace, s);

    // multi-GPU rig: sum the packed system over ranks before every rank solves it
    hipLaunchKernelGGL(system_final_kernel, dim3(1), dim3(256), 0, s, workspace, partials, translation_enabled,
        hessian, gradient, (vk_transform*)nullptr, (int32_t*)nullptr, (float*)nullptr, Mirror{nullptr, 0, nullptr});
    VK_LAUNCH_CHECK();
    const int rr = reduce(system, 48, reduce_user, stream);
    if (rr != 0) return rr;
    hipLaunchKernelGGL(solve_update_kernel, dim3(1), dim3(64), 0, s, hessian, gradient, translation_enabled,
        Twc_dev, state_dev, update_dev, mirror);
    VK_LAUNCH_CHECK();

    // tracker.cpp:162: the reference leaves its loop once |update| < 1e-6. Steps enqueued
    // after that point are no-ops, but each still costs its launches; so the host looks
    // at the mirror every `chunk` steps and stops enqueuing when the loop has converged.
    if (chunked && (it + 1) % poll->chunk == 0 && it + 1 >= 2 * poll->chunk && it + 1 < iterations &&
        wait_for_steps(mirror, it + 1 - poll->chunk, s)) break;
  }
  if (mirror.host_pose)
  {
    hipLaunchKernelGGL(publish_pose_kernel, dim3(1), dim3(64), 0, s, mirror, Twc_dev);
    VK_LAUNCH_CHECK();
  }
  return VK_OK;
}

// the pose travels in the dispatch packet: no staging copy, no host synchronisation
__global__ void store_transform_kernel(vk_transform* dst, vk_transform value)
{
  if (threadIdx.x < 32)
  {
    const float v = threadIdx.x < 16 ? value.m[threadIdx.x] : value.inv[threadIdx.x - 16];
    if (threadIdx.x < 16) dst->m[threadIdx.x] = v; else dst->inv[threadIdx.x - 16] = v;
  }
}

int vk_transform_upload(vk_transform* dst_dev, const vk_transform* src_host, void* stream)
{
  VK_REQUIRE(dst_dev && src_host);
  hipLaunchKernelGGL(store_transform_kernel, dim3(1), dim3(64), 0, vk_s(stream), dst_dev, *src_host);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

size_t vk_icp_pyramid_floats(int key_width, int key_height, int frame_width, int frame_height)
{
  if (key_width <= 0 || key_height <= 0 || frame_width <= 0 || frame_height <= 0) return 0;
  if ((key_width | key_height | frame_width | frame_height) & 1) return 0;
  return 4 * ((size_t)(key_width / 2) * (key_height / 2) + (size_t)(frame_width / 2) * (frame_height / 2));
}

// `level_built`: the half-resolution level and both normal images were made behind the previous raycast
// (vk_trace_ahead_pyramid): no pyramid launch
static int pyramid_track(const vk_icp_view* keyframe, const vk_transform* Twm, const vk_icp_view* frame,
    vk_transform* Twc_dev, const vk_transform* Twc_start, int frame_normals_due, float* pyramid, float* workspace, float* system,
    int32_t* state_dev, float* update_dev, vk_icp_reduce_fn reduce, void* reduce_user, const vk_track_poll* poll, void* stream,
    bool level_built = false, bool frame_side_only = false)
{
  VK_REQUIRE(keyframe && Twm && frame && Twc_dev && pyramid && workspace && system && state_dev);
  VK_REQUIRE(keyframe->depths && keyframe->normals && frame->depths && frame->normals);
  VK_REQUIRE(vk_icp_pyramid_floats(keyframe->width, keyframe->height, frame->width, frame->height) > 0);
  hipStream_t s = vk_s(stream);

  // pyramid_tracker.cpp:58-62: half-resolution frame and keyframe (the quarter level is built
  // upstream but never tracked: :64-77 are commented out)
  vk_icp_view half[2] = {*keyframe, *frame};
  LevelParams L;
  float* at = pyramid;
  const vk_icp_view* full[2] = {keyframe, frame};
  for (int side = 0; side < 2; ++side)
  {
    const int w = full[side]->width / 2, h = full[side]->height / 2;
    L.src_depth[side] = full[side]->depths;
    L.src_normals[side] = full[side]->normals;
    L.dst_depth[side] = at;
    L.dst_normals[side] = at + (size_t)w * h;
    at += 4 * (size_t)w * h;
    L.src_w[side] = full[side]->width;
    L.dst_w[side] = w;
    L.dst_h[side] = h;
    half[side].depths = L.dst_depth[side];
    half[side].normals = L.dst_normals[side];
    half[side].width = w;
    half[side].height = h;
    // frame.cpp:53-56: focal length and centre / 2 (Vector2f / 2 multiplies by 1 / 2, matrix.h:279-295)
    half[side].projection.fx = full[side]->projection.fx * 0.5f;
    half[side].projection.fy = full[side]->projection.fy * 0.5f;
    half[side].projection.cx = full[side]->projection.cx * 0.5f;
    half[side].projection.cy = full[side]->projection.cy * 0.5f;
  }
  int gw = half[0].width > half[1].width ? half[0].width : half[1].width;
  int gh = half[0].height > half[1].height ? half[0].height : half[1].height;
  // frame_normals_due: bit 0 the frame's normal image, bit 1 the keyframe's (vk.h)
  VK_REQUIRE(frame_normals_due >= 0 && frame_normals_due <= 3);
  int due = 0;
  L.due_side[0] = L.due_side[1] = 0;
  for (int side = 0; side < 2; ++side)
  {
    const bool wanted = (frame_normals_due >> (1 - side)) & 1;          // side 0 = keyframe = bit 1, side 1 = frame = bit 0
    L.normals_out[side] = wanted ? const_cast<float*>(full[side]->normals) : nullptr;
    L.src_h[side] = full[side]->height;
    L.k[side] = full[side]->projection;
    if (wanted)
    {
      L.due_side[due++] = side;
      gw = gw > full[side]->width ? gw : full[side]->width;
      gh = gh > full[side]->height ? gh : full[side]->height;
    }
  }
  L.pose_out = Twc_start ? Twc_dev : nullptr;
  if (Twc_start) L.pose_start = *Twc_start;
  if (!level_built || frame_side_only)
  {
    // (frame_side_only — an experiment, -DVK_TP_KEY_SIDE=0 — : the frame's level exists, its workgroups leave at once)
    if (level_built) L.dst_w[1] = L.dst_h[1] = 0;
    hipLaunchKernelGGL(pyramid_level_kernel, dim3((gw + 63) / 64, (gh + 3) / 4, 2 + due), dim3(256), 0, s, L);
    VK_LAUNCH_CHECK();
  }
  else if (Twc_start)
  {
    hipLaunchKernelGGL(store_transform_kernel, dim3(1), dim3(64), 0, s, Twc_dev, *Twc_start);
    VK_LAUNCH_CHECK();
  }

  // :79-83 half level, 15 steps; :85-89 full level, 20 steps, from the pose the half level left.
  // Tracker::CreateState (tracker.cpp:107-110) starts every Track at iteration 0.
  if (!reduce)
  {
    // one launch per level; the loop kernel starts from a fresh state by itself
    const vk_icp_view* views[2][2] = {{&half[0], &half[1]}, {keyframe, frame}};
    const int steps[2] = {15, 20};
    const Mirror mirror = begin_mirror(poll);
    vk_transform identity;
    for (int i = 0; i < 16; ++i) identity.m[i] = identity.inv[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    for (int level = 0; level < 2; ++level)
    {
      IcpParams P;
      const int rc = fill_icp(P, views[level][0], Twm, views[level][1], &identity);
      if (rc != VK_OK) return rc;
      const int rl = launch_loop(P, Twc_dev, steps[level], 1, group_count_for(views[level][1]->width * views[level][1]->height, P.group_pixels),
          workspace, system, system + 36, state_dev, update_dev, mirror, /*fresh_state*/ level == 0 ? 1 : 2, /*ends_track*/ level == 1, s);
      if (rl != VK_OK) return rl;
    }
    return VK_OK;
  }
  VK_CHECK(hipMemsetAsync(state_dev, 0, 2 * sizeof(int32_t), s));
  int rc = vk_icp_track(&half[0], Twm, &half[1], Twc_dev, 15, 1, workspace, system, state_dev, update_dev,
      reduce, reduce_user, poll, stream);
  if (rc != VK_OK) return rc;
  VK_CHECK(hipMemsetAsync(state_dev, 0, 2 * sizeof(int32_t), s));
  return vk_icp_track(keyframe, Twm, frame, Twc_dev, 20, 1, workspace, system, state_dev, update_dev,
      reduce, reduce_user, poll, stream);
}

int vk_icp_pyramid_track(const vk_icp_view* keyframe, const vk_transform* Twm, const vk_icp_view* frame,
    vk_transform* Twc_dev, float* pyramid, float* workspace, float* system, int32_t* state_dev,
    float* update_dev, vk_icp_reduce_fn reduce, void* reduce_user, const vk_track_poll* poll, void* stream)
{
  return pyramid_track(keyframe, Twm, frame, Twc_dev, nullptr, 0, pyramid, workspace, system, state_dev, update_dev, reduce,
      reduce_user, poll, stream);
}

int vk_icp_pyramid_track_frame(const vk_icp_view* keyframe, const vk_transform* Twm, const vk_icp_view* frame,
    vk_transform* Twc_dev, const vk_transform* Twc_start, int frame_normals_due, float* pyramid, float* workspace, float* system,
    int32_t* state_dev, float* update_dev, vk_icp_reduce_fn reduce, void* reduce_user, const vk_track_poll* poll, void* stream)
{
  return pyramid_track(keyframe, Twm, frame, Twc_dev, Twc_start, frame_normals_due, pyramid, workspace, system, state_dev,
      update_dev, reduce, reduce_user, poll, stream);
}

int vk_icp_pyramid_track_built(const vk_icp_view* keyframe, const vk_transform* Twm, const vk_icp_view* frame,
    vk_transform* Twc_dev, const vk_transform* Twc_start, int frame_normals_due, vk_pyramid_ahead* built, float* pyramid,
    float* workspace, float* system, int32_t* state_dev, float* update_dev, vk_icp_reduce_fn reduce, void* reduce_user,
    const vk_track_poll* poll, void* stream)
{
  VK_REQUIRE(keyframe && frame);
  // the record names the images and the buffer the level was built from / into, and serves once
  const bool level_built = built && built->valid == 1 && built->pyramid == pyramid &&
      built->key_depths == keyframe->depths && built->key_normals == keyframe->normals &&
      built->frame_depths == frame->depths && built->frame_normals == frame->normals &&
      built->key_width == keyframe->width && built->key_height == keyframe->height &&
      built->frame_width == frame->width && built->frame_height == frame->height;
  const bool frame_side_only = level_built && built->pad_ == 1;
  if (built) built->valid = 0;
  return pyramid_track(keyframe, Twm, frame, Twc_dev, Twc_start, level_built ? (frame_side_only ? 2 : 0) : frame_normals_due, pyramid,
      workspace, system, state_dev, update_dev, reduce, reduce_user, poll, stream, level_built, frame_side_only);
}

int vk_reduce_nothing(float*, int, void*, void*) { return 0; }

size_t vk_rig_area_bytes(void) { return rig_area_words() * sizeof(unsigned long long); }

int vk_icp_track_rig(const vk_icp_view* keyframe, const vk_transform* Twm, const vk_icp_view* frame,
    vk_transform* Twc_dev, int iterations, int translation_enabled, float* workspace, float* system,
    int32_t* state_dev, float* update_dev, const vk_rig_exchange* rig, const vk_track_poll* poll, void* stream)
{
  IcpParams P;
  vk_transform identity;
  for (int i = 0; i < 16; ++i) identity.m[i] = identity.inv[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  const int rc = fill_icp(P, keyframe, Twm, frame, &identity);
  if (rc != VK_OK) return rc;
  VK_REQUIRE(Twc_dev && workspace && system && state_dev && iterations > 0 && iterations <= 1000);
  VK_REQUIRE(rig && rig->world >= 1 && rig->world <= VK_RIG_MAX_RANKS && rig->rank >= 0 && rig->rank < rig->world);
  VK_REQUIRE(rig->sequence != 0 && rig->sequence < (1u << 22));
  for (int r = 0; r < rig->world; ++r) VK_REQUIRE(rig->areas[r] && (reinterpret_cast<uintptr_t>(rig->areas[r]) & 7u) == 0);
  P.Twc_dev = Twc_dev;
  P.state = state_dev;
  const int partials = group_count_for(frame->width * frame->height, P.group_pixels);
  return launch_loop(P, Twc_dev, iterations, translation_enabled, partials, workspace, system, system + 36,
      state_dev, update_dev, begin_mirror(poll), /*fresh_state*/ 0, /*ends_track*/ true, vk_s(stream), rig);
}

int vk_track_wait(const vk_track_poll* poll, void* stream)
{
  VK_REQUIRE(poll && poll->host_state && poll->host_pose);
  const volatile int32_t* host = poll->host_state;
  const int32_t tag = host[2];
  if (tag == 0) return VK_ERR_UNSUPPORTED;      // no Track has been issued with this block (tags are never 0)
  hipStream_t s = vk_s(stream);
  // The word is watched without a pause; the stream is asked whether it has drained (the safety net for a Track that left no
  // pose) only every quarter of a millisecond. Until round 6 it was asked every 1 024 looks — a hipStreamQuery takes
  // microseconds, so the caller spent half its wait inside it and saw the pose up to a query late: 20 us of the tracked
  // frame's host round trip (profiles/r06_tracked_frame_timeline.txt).
  timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  long long next_query_ns = 250000;
  for (unsigned spin = 0;; ++spin)
  {
    if (host[3] == tag) return VK_OK;
    if ((spin & 255u) == 255u)
    {
      timespec t;
      clock_gettime(CLOCK_MONOTONIC, &t);
      const long long waited = (long long)(t.tv_sec - t0.tv_sec) * 1000000000ll + (t.tv_nsec - t0.tv_nsec);
      if (waited >= next_query_ns)
      {
        if (hipStreamQuery(s) != hipErrorNotReady) return host[3] == tag ? VK_OK : VK_ERR_UNSUPPORTED;   // drained: final
        next_query_ns = waited + 250000;
      }
    }
  }
}

int vk_icp_solve_update(const float* hessian, const float* gradient, int translation_enabled,
    vk_transform* Twc_dev, int32_t* state_dev, float* update_dev, void* stream)
{
  VK_REQUIRE(hessian && gradient && Twc_dev);
  hipLaunchKernelGGL(solve_update_kernel, dim3(1), dim3(64), 0, vk_s(stream), hessian, gradient,
      translation_enabled, Twc_dev, state_dev, update_dev, Mirror{nullptr, 0, nullptr});
  VK_LAUNCH_CHECK();
  return VK_OK;
}

int vk_frame_downsample(const vk_frame* frame, float* depth_out, float* color_out, float* normals_out, void* stream)
{
  VK_REQUIRE(frame && frame->width > 0 && frame->height > 0 && (frame->width % 2) == 0 && (frame->height % 2) == 0);
  VK_REQUIRE((!depth_out || frame->depth) && (!color_out || frame->color) && (!normals_out || frame->normals));
  const int cw = frame->color_width > 0 ? frame->color_width : frame->width;
  const int ch = frame->color_height > 0 ? frame->color_height : frame->height;
  VK_REQUIRE(!color_out || ((cw % 2) == 0 && (ch % 2) == 0));
  FrameLevel L;
  const float* src[3] = {frame->depth, frame->color, frame->normals};
  float* dst[3] = {depth_out, color_out, normals_out};
  int gw = 0, gh = 0;
  for (int job = 0; job < 3; ++job)
  {
    const int w = job == 1 ? cw : frame->width, h = job == 1 ? ch : frame->height;
    L.src[job] = src[job];
    L.dst[job] = dst[job];
    L.src_w[job] = w;
    L.dst_w[job] = w / 2;
    L.dst_h[job] = h / 2;
    if (dst[job]) { gw = gw > w / 2 ? gw : w / 2; gh = gh > h / 2 ? gh : h / 2; }
  }
  if (gw == 0) return VK_OK;
  hipLaunchKernelGGL(frame_downsample_kernel, dim3((gw + 63) / 64, (gh + 3) / 4, 3), dim3(256), 0, vk_s(stream), L);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

int vk_image_downsample(int src_w, int src_h, const float* src, float* dst, int nearest, void* stream)
{
  VK_REQUIRE(src && dst && src_w > 0 && src_h > 0 && (src_w % 2) == 0 && (src_h % 2) == 0);
  const int dst_w = src_w / 2, dst_h = src_h / 2;
  const dim3 grid((dst_w + 63) / 64, (dst_h + 3) / 4);
  hipLaunchKernelGGL(downsample_kernel, grid, dim3(256), 0, vk_s(stream), src_w, dst_w, dst_h, src, dst, nearest);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

int vk_color_image_downsample(int src_w, int src_h, const float* src, float* dst, int nearest, void* stream)
{
  VK_REQUIRE(src && dst && src_w > 0 && src_h > 0 && (src_w % 2) == 0 && (src_h % 2) == 0);
  const int dst_w = src_w / 2, dst_h = src_h / 2;
  const dim3 grid((dst_w + 63) / 64, (dst_h + 3) / 4);
  hipLaunchKernelGGL(downsample3_kernel, grid, dim3(256), 0, vk_s(stream), src_w, dst_w, dst_h, src, dst, nearest);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

}  // extern "C"
