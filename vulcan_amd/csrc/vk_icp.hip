// vk_icp.hip — projective point-to-plane ICP normal system, Gauss-Newton update
// and the image pyramid for gfx950 (ref: src/depth_tracker.cu,
// src/tracker.cpp:124-163, src/depth_tracker.cpp:22-86, src/image.cu:101-165).
//
// The reference reduces 27 sums with 9 rounds of three 256-wide LDS trees per
// workgroup and 27 float atomicAdds per workgroup (order-nondeterministic), then
// copies 42 floats to the host every Gauss-Newton iteration for an Eigen LDLT.
// Here: registers -> wave64 butterfly -> one LDS hop -> per-workgroup partials,
// summed by a second kernel in a fixed order (bit-reproducible), and the 6x6
// solve + SE(3) update can run on the device so an iteration needs no readback.
#include "vk_gauss_newton.hpp"

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
template <bool TRANSLATION>
__device__ __forceinline__ bool evaluate(const IcpParams& P, const Rt& Twc, int frame_x, int frame_y,
    float& residual, float J[6])
{
  residual = 0.0f;
#pragma unroll
  for (int i = 0; i < 6; ++i) J[i] = 0.0f;

  const View& frm = P.frm;
  const View& key = P.key;
  if (!(frame_x < frm.width && frame_y < frm.height)) return false;

  const int frame_index = frame_y * frm.width + frame_x;
  const float frame_depth = frm.depths[frame_index];
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

  f3 frame_normal = make3(frm.normals[3 * frame_index + 0], frm.normals[3 * frame_index + 1],
      frm.normals[3 * frame_index + 2]);
  frame_normal = xform_dir(Twc, frame_normal);
  f3 keyframe_normal = make3(key.normals[3 * keyframe_index + 0], key.normals[3 * keyframe_index + 1],
      key.normals[3 * keyframe_index + 2]);
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
    vk_transform* Twc, int32_t* state, float* update_out, unsigned long long* mirror);

// ref: depth_tracker.cu:144-268. Slot layout of a partial: [0,6) J^T r,
// [6,27) packed lower triangle of J^T J in (r, c<=r) row-major order.
//
// Measured and rejected: letting the workgroup that finishes last (atomic ticket)
// also run the second stage, for one launch per iteration instead of two. On this
// multi-XCD part the agent-scope release/acquire fences that publish the partials
// write back and invalidate whole L2s, once per workgroup: the fused iteration
// took 2.5x as long as the two launches.
template <bool TRANSLATION>
__global__ __launch_bounds__(kSysThreads) void system_partial_kernel(IcpParams P, float* __restrict__ workspace)
{
  __shared__ float lds[kSysWaves][kSysStride];

  // tracker.cpp:162: once the update norm fell below 1e-6 the reference leaves its
  // loop; here the remaining (already enqueued) iterations turn into empty launches
  if (P.state && P.state[1]) return;

  const Rt Twc = P.Twc_dev ? rt_from_colmajor(P.Twc_dev->m) : P.Twc;
  const int total = P.frm.width * P.frm.height;
  const int pixel = blockIdx.x * kSysThreads + (int)threadIdx.x;

  float acc[27];
#pragma unroll
  for (int i = 0; i < 27; ++i) acc[i] = 0.0f;

  float r, J[6];
  if (pixel < total && evaluate<TRANSLATION>(P, Twc, pixel % P.frm.width, pixel / P.frm.width, r, J))
    outer_products(J, r, acc);

  store_partial(acc, lds, workspace);
}

// Second stage: one workgroup. With `Twc` non-null it also solves and updates the
// pose, so a Gauss-Newton iteration is two launches.
__global__ __launch_bounds__(256) void system_final_kernel(const float* __restrict__ workspace,
    int partials, int translation_enabled, float* __restrict__ hessian, float* __restrict__ gradient,
    vk_transform* Twc, int32_t* state, float* update_out, unsigned long long* mirror)
{
  __shared__ float slices[8][kSysStride];
  __shared__ float sums[48];   // hessian[36] | gradient[6]: the solve reads them from LDS
  if (state && state[1]) return;   // converged: the system was not recomputed, keep the last one
  sum_partials(workspace, partials, translation_enabled, hessian, gradient, slices, sums);
  if (Twc && threadIdx.x == 0) solve_update(sums, sums + 36, translation_enabled, Twc, state, update_out, mirror);
}

// ---- pose update on the device ------------------------------------------------

// ref: tracker.cpp:124-163 + depth_tracker.cpp:22-86. One lane; 6x6 is too
// small to spread.
template <int N>
__device__ __forceinline__ void solve_update_n(const float* hessian, const float* gradient,
    vk_transform* Twc, int32_t* state, float* update_out, unsigned long long* mirror)
{
  float update[6];
  solve_step<N>(hessian, gradient, update);

  // depth_tracker.cpp:33-53, including Tinc(1,2) = +update[0]
  float Tinc[16];
  Tinc[0] = 1.0f;        Tinc[4] = -update[2]; Tinc[8] = +update[1];  Tinc[12] = +update[3];
  Tinc[1] = +update[2];  Tinc[5] = 1.0f;       Tinc[9] = +update[0];  Tinc[13] = +update[4];
  Tinc[2] = -update[1];  Tinc[6] = +update[0]; Tinc[10] = 1.0f;       Tinc[14] = +update[5];
  Tinc[3] = 0.0f;        Tinc[7] = 0.0f;       Tinc[11] = 0.0f;       Tinc[15] = 1.0f;

  float old_m[16], M[16], out_m[16], out_i[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) old_m[i] = Twc->m[i];
  matmul4(Tinc, old_m, M);
  rigid_from(M, out_m, out_i);
#pragma unroll
  for (int i = 0; i < 16; ++i) { Twc->m[i] = out_m[i]; Twc->inv[i] = out_i[i]; }

  finish_step<N>(update, state, update_out, mirror);
}

__device__ void solve_update(const float* hessian, const float* gradient,
    int translation_enabled, vk_transform* Twc, int32_t* state, float* update_out, unsigned long long* mirror)
{
  if (state && state[1]) return;  // converged earlier: tracker.cpp:162
  if (translation_enabled) solve_update_n<6>(hessian, gradient, Twc, state, update_out, mirror);
  else solve_update_n<3>(hessian, gradient, Twc, state, update_out, mirror);
}

__global__ void solve_update_kernel(const float* __restrict__ hessian, const float* __restrict__ gradient,
    int translation_enabled, vk_transform* Twc, int32_t* state, float* update_out, unsigned long long* mirror)
{
  if (threadIdx.x == 0 && blockIdx.x == 0)
    solve_update(hessian, gradient, translation_enabled, Twc, state, update_out, mirror);
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
  return VK_OK;
}

void launch_partials(const IcpParams& P, int translation_enabled, int partials, float* workspace, hipStream_t s)
{
  if (translation_enabled)
    hipLaunchKernelGGL(system_partial_kernel<true>, dim3(partials), dim3(kSysThreads), 0, s, P, workspace);
  else
    hipLaunchKernelGGL(system_partial_kernel<false>, dim3(partials), dim3(kSysThreads), 0, s, P, workspace);
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
  return (size_t)partial_count(width, height) * kSysStride;
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
  const int partials = partial_count(frame->width, frame->height);
  launch_partials(P, translation_enabled, partials, workspace, vk_s(stream));
  VK_LAUNCH_CHECK();
  hipLaunchKernelGGL(system_final_kernel, dim3(1), dim3(256), 0, vk_s(stream), workspace, partials,
      translation_enabled, hessian, gradient, (vk_transform*)nullptr, (int32_t*)nullptr, (float*)nullptr,
      (unsigned long long*)nullptr);
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
  const int partials = partial_count(frame->width, frame->height);
  hipStream_t s = vk_s(stream);
  const bool chunked = polling(poll);
  unsigned long long* mirror = chunked ? reinterpret_cast<unsigned long long*>(poll->host_state) : nullptr;
  if (chunked) *reinterpret_cast<volatile unsigned long long*>(mirror) = 0;   // the caller zeroed state_dev too

  for (int it = 0; it < iterations; ++it)
  {
    launch_partials(P, translation_enabled, partials, workspace, s);

    if (reduce)
    {
      // multi-GPU rig: sum the packed system over ranks before every rank solves it
      hipLaunchKernelGGL(system_final_kernel, dim3(1), dim3(256), 0, s, workspace, partials, translation_enabled,
          hessian, gradient, (vk_transform*)nullptr, (int32_t*)nullptr, (float*)nullptr, (unsigned long long*)nullptr);
      VK_LAUNCH_CHECK();
      const int rr = reduce(system, 48, reduce_user, stream);
      if (rr != 0) return rr;
      hipLaunchKernelGGL(solve_update_kernel, dim3(1), dim3(64), 0, s, hessian, gradient, translation_enabled,
          Twc_dev, state_dev, update_dev, mirror);
    }
    else
    {
      hipLaunchKernelGGL(system_final_kernel, dim3(1), dim3(256), 0, s, workspace, partials, translation_enabled,
          hessian, gradient, Twc_dev, state_dev, update_dev, mirror);
    }
    VK_LAUNCH_CHECK();

    // tracker.cpp:162: the reference leaves its loop once |update| < 1e-6. Steps enqueued
    // after that point are no-ops, but each still costs two launches; so the host looks
    // at the mirror every `chunk` steps and stops enqueuing when the loop has converged.
    if (chunked && (it + 1) % poll->chunk == 0 && it + 1 < iterations && wait_for_steps(poll, it + 1, s)) break;
  }
  return VK_OK;
}

int vk_icp_solve_update(const float* hessian, const float* gradient, int translation_enabled,
    vk_transform* Twc_dev, int32_t* state_dev, float* update_dev, void* stream)
{
  VK_REQUIRE(hessian && gradient && Twc_dev);
  hipLaunchKernelGGL(solve_update_kernel, dim3(1), dim3(64), 0, vk_s(stream), hessian, gradient,
      translation_enabled, Twc_dev, state_dev, update_dev, (unsigned long long*)nullptr);
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
