// vk_color_tracker.hip — photometric frame-to-keyframe tracking for gfx950
// (ref: src/color_tracker.cu, src/color_tracker.cpp:34-96, src/image.cu:10-99).
//
// Same shape as the depth tracker (vk_icp.hip): keyframe pixels in groups defined by the
// image, the 27 sums reduced by DPP row operations and a fixed-order second stage, the 6x6
// solve and the pose update done by one lane out of registers. Track() is ONE launch for
// the whole Gauss-Newton loop (color_loop_kernel; three launches per step only with the
// rig's reduce hook) after one launch that prepares the images (color_begin_kernel).
// The reference zero-fills, accumulates with 27 float atomics per thread block and
// copies 42 floats to the host for Eigen every iteration.
#include "vk_gauss_newton.hpp"

using namespace vk;

namespace
{

// ---------------------------------------------------------------- images ----

// image.cu:10-19, one pixel
__device__ __forceinline__ float intensity_of(const float* rgb, int index)
{
  return (rgb[3 * index + 0] + rgb[3 * index + 1] + rgb[3 * index + 2]) / 3.0f;
}

// ref: image.cu:10-19
__global__ __launch_bounds__(256) void convert_kernel(int total, const float* __restrict__ src,
    float* __restrict__ dst)
{
  const int index = blockIdx.x * 256 + threadIdx.x;
  if (index < total) dst[index] = intensity_of(src, index);
}

__device__ __forceinline__ float padded(const float* v, int w, int h, int x, int y)
{
  return (x >= 0 && x < w && y >= 0 && y < h) ? v[y * w + x] : 0.0f;
}

// the same tap taken from the colour image: the intensity is recomputed (same expression,
// same bits as the converted image holds)
__device__ __forceinline__ float padded_intensity(const float* rgb, int w, int h, int x, int y)
{
  return (x >= 0 && x < w && y >= 0 && y < h) ? intensity_of(rgb, y * w + x) : 0.0f;
}

// ref: image.cu:21-99. The eight taps come through L1/L2 (each pixel is read by
// its eight neighbours within the same few waves) instead of an 18x18 LDS patch.
__global__ __launch_bounds__(256) void gradients_kernel(int width, int height, const float* __restrict__ values,
    float* __restrict__ gx_out, float* __restrict__ gy_out)
{
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= width || y >= height) return;

  const float i00 = 0.125f * padded(values, width, height, x - 1, y - 1);
  const float i01 = 0.250f * padded(values, width, height, x + 0, y - 1);
  const float i02 = 0.125f * padded(values, width, height, x + 1, y - 1);
  const float i10 = 0.250f * padded(values, width, height, x - 1, y + 0);
  const float i12 = 0.250f * padded(values, width, height, x + 1, y + 0);
  const float i20 = 0.125f * padded(values, width, height, x - 1, y + 1);
  const float i21 = 0.250f * padded(values, width, height, x + 0, y + 1);
  const float i22 = 0.125f * padded(values, width, height, x + 1, y + 1);

  gx_out[y * width + x] = (i02 + i12 + i22) - (i00 + i10 + i20);
  gy_out[y * width + x] = (i20 + i21 + i22) - (i00 + i01 + i02);
}


// ---- everything a Track needs before its first step, in one launch ----------------
//
// Tracker::BeginSolve + ColorTracker / LightTracker::BeginSolve (tracker.cpp:65-76,
// color_tracker.cpp:19-25, light_tracker.cpp:34-41) are, upstream and in the staged entry
// points here, a pose upload, a state reset and four image passes: keyframe intensities, frame
// intensities, frame gradients, frame mask — six launches of ~4.5 us of which a fraction is
// work. None depends on another once the gradients take their taps from the colour image
// (recomputing the intensity of a tap gives the bits the converted image holds), so they run
// side by side in one launch: blockIdx.z picks the job.
struct BeginParams
{
  const float* key_color;  float* key_intensities;  int key_total;
  const float* frm_color;  float* frm_intensities;  float* gradient_x;  float* gradient_y;
  int width, height;                      // of the frame's colour image
  // light tracker only (mask == nullptr otherwise)
  const float* frm_depth;  float depth_threshold;  float* mask;
  // Tracker::BeginSolve
  vk_transform pose;  vk_color_pose* pose_dev;  int32_t* state_dev;
};

__global__ __launch_bounds__(256) void color_begin_kernel(BeginParams B)
{
  __shared__ float buffer[22 * 22];
  const int job = blockIdx.z;
  const int linear = (blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
  if (job == 0)
  {
    if (linear < B.key_total) B.key_intensities[linear] = intensity_of(B.key_color, linear);
    if (linear < 32 && B.pose_dev)
    {
      const float v = linear < 16 ? B.pose.m[linear] : B.pose.inv[linear - 16];
      if (linear < 16) B.pose_dev->depth_to_world.m[linear] = v; else B.pose_dev->depth_to_world.inv[linear - 16] = v;
    }
    // a Track's first level (it uploads the pose) starts from a clean state; a later level keeps
    // an abort of the level before it, so that the Track fails as a whole
    if (linear < 2 && B.state_dev && (B.pose_dev || B.state_dev[1] != VK_TRACK_ABORTED)) B.state_dev[linear] = 0;
  }
  else if (job == 1)
  {
    if (linear < B.width * B.height) B.frm_intensities[linear] = intensity_of(B.frm_color, linear);
  }
  else if (job == 2)
  {
    // gradients_kernel with the taps converted on the fly
    const int x = linear % B.width, y = linear / B.width;
    if (y >= B.height) return;
    const float* c = B.frm_color;
    const int w = B.width, h = B.height;
    const float i00 = 0.125f * padded_intensity(c, w, h, x - 1, y - 1);
    const float i01 = 0.250f * padded_intensity(c, w, h, x + 0, y - 1);
    const float i02 = 0.125f * padded_intensity(c, w, h, x + 1, y - 1);
    const float i10 = 0.250f * padded_intensity(c, w, h, x - 1, y + 0);
    const float i12 = 0.250f * padded_intensity(c, w, h, x + 1, y + 0);
    const float i20 = 0.125f * padded_intensity(c, w, h, x - 1, y + 1);
    const float i21 = 0.250f * padded_intensity(c, w, h, x + 0, y + 1);
    const float i22 = 0.125f * padded_intensity(c, w, h, x + 1, y + 1);
    B.gradient_x[linear] = (i02 + i12 + i22) - (i00 + i10 + i20);
    B.gradient_y[linear] = (i20 + i21 + i22) - (i00 + i01 + i02);
  }
  else
  {
    // frame_mask_kernel (vk_integrate.hip): 16x16 pixels per workgroup, 22x22 depth tile
    if (!B.mask) return;
    const int tiles_x = (B.width + 15) / 16, tiles_y = (B.height + 15) / 16;
    const int tile = blockIdx.y * gridDim.x + blockIdx.x;
    if (tile >= tiles_x * tiles_y) return;             // whole workgroup
    const int bx = tile % tiles_x, by = tile / tiles_x;
    for (int sindex = threadIdx.x; sindex < 22 * 22; sindex += 256)
    {
      float depth = 0;
      const int vx = (bx * 16 - 1) + (sindex % 22);
      const int vy = (by * 16 - 1) + (sindex / 22);
      if (vx >= 0 && vx < B.width && vy >= 0 && vy < B.height) depth = B.frm_depth[vy * B.width + vx];
      buffer[sindex] = depth;
    }
    __syncthreads();
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int x = bx * 16 + tx, y = by * 16 + ty;
    if (x < B.width && y < B.height)
    {
      const int index = y * B.width + x;
      const float c0 = B.frm_color[3 * index + 0], c1 = B.frm_color[3 * index + 1], c2 = B.frm_color[3 * index + 2];
      B.mask[index] = light_color_usable(c0, c1, c2) ? light_window_mask(buffer, 22, tx + 3, ty + 3, B.depth_threshold) : 0.0f;
    }
  }
}

// -------------------------------------------------------------- evaluate ----

struct ColorParams
{
  vk_color_view key, frm;
  Rt Tcm;
  const vk_transform* Tcm_dev;  // optional device override of Tcm
  const int32_t* state;         // optional {iterations, converged}: a converged solve skips the pass
  int group_pixels;             // keyframe pixels per partial sum (group_pixels_for)
  // light tracker only
  const float* mask;
  vk_light light;
  Rt Tcd;
};

// ref: color_tracker.cu:17-41
__device__ __forceinline__ float sample(int w, const float* values, float u, float v)
{
  const int x = f2i(floorf(u - 0.5f));
  const int y = f2i(floorf(v - 0.5f));

  const float v00 = values[(y + 0) * w + (x + 0)];
  const float v01 = values[(y + 0) * w + (x + 1)];
  const float v10 = values[(y + 1) * w + (x + 0)];
  const float v11 = values[(y + 1) * w + (x + 1)];

  const float u1 = u - (x + 0.5f);
  const float v1 = v - (y + 0.5f);
  const float u0 = 1.0f - u1;
  const float v0 = 1.0f - v1;

  const float w00 = v0 * u0;
  const float w01 = v0 * u1;
  const float w10 = v1 * u0;
  const float w11 = v1 * u1;

  return (w00 * v00) + (w01 * v01) + (w10 * v10) + (w11 * v11);
}

// ref: color_tracker.cu:43-138 Evaluate<translation_enabled>; returns false when
// the pixel contributes nothing (residual 0, Jacobian 0).
template <bool TRANSLATION, bool JACOBIAN>
__device__ __forceinline__ bool evaluate(const ColorParams& P, const Rt& Tcm, int keyframe_x, int keyframe_y,
    float& residual, float (&J)[6])
{
  residual = 0.0f;
#pragma unroll
  for (int i = 0; i < 6; ++i) J[i] = 0.0f;

  const vk_color_view& key = P.key;
  const vk_color_view& frm = P.frm;

  const int keyframe_index = keyframe_y * key.width + keyframe_x;
  const float keyframe_depth = key.depths[keyframe_index];
  if (!(keyframe_depth > 0)) return false;

  const f3 Xmp = unproject_d(key.projection, keyframe_x + 0.5f, keyframe_y + 0.5f, keyframe_depth);
  const f3 Xcp = xform_point(Tcm, Xmp);
  float fu, fv;
  project(frm.projection, Xcp, fu, fv);

  if (!(fu >= 0.5f && fu < frm.width - 0.5f && fv >= 0.5f && fv < frm.height - 0.5f)) return false;

  const int frame_index = f2i(fv) * frm.width + f2i(fu);
  const float frame_depth = frm.depths[frame_index];
  if (!(fabsf(frame_depth - Xcp.z) < 0.1f)) return false;

  const f3 frame_normal = make3(frm.normals[3 * frame_index + 0], frm.normals[3 * frame_index + 1],
      frm.normals[3 * frame_index + 2]);
  f3 keyframe_normal = make3(key.normals[3 * keyframe_index + 0], key.normals[3 * keyframe_index + 1],
      key.normals[3 * keyframe_index + 2]);
  keyframe_normal = xform_dir(Tcm, keyframe_normal);

  if (!(sqnorm3(keyframe_normal) > 0.5f && dot3(frame_normal, keyframe_normal) > 0.5f)) return false;

  const float Im = key.intensities[keyframe_index];
  const float Ic = sample(frm.width, frm.intensities, fu, fv);
  residual = Ic - Im;

  if (JACOBIAN)
  {
    const float px = Xcp.x, py = Xcp.y, pz = Xcp.z;
    const float inv_pz = 1.0f / pz;
    const float cu = fu, cv = fv;
    const float fx = frm.projection.fx, fy = frm.projection.fy;
    const float cx = frm.projection.cx, cy = frm.projection.cy;
    const float gx = sample(frm.width, frm.gradient_x, fu, fv);
    const float gy = sample(frm.width, frm.gradient_y, fu, fv);

    J[0] = gy * ((py * cy - pz * fy) * inv_pz - py * cv * inv_pz) - gx * (py * cu * inv_pz - cx * py * inv_pz);
    J[1] = gy * (px * cv * inv_pz - cy * px * inv_pz) - gx * ((px * cx - pz * fx) * inv_pz - px * cu * inv_pz);
    J[2] = (gy * fy * px - gx * fx * py) * inv_pz;

    if (TRANSLATION)
    {
      J[3] = gx * fx * inv_pz;
      J[4] = gy * fy * inv_pz;
      J[5] = (gx * (cx - cu) + gy * (cy - cv)) * inv_pz;
    }
  }
  return true;
}

// ref: light_tracker.cu:133-330 Evaluate<translation_enabled>. Residual and the
// point-to-plane fallback (:283-322) line by line. The photometric Jacobian is the
// reference's derivative in factored form instead of its six machine-generated
// powf / sqrt expressions (:233-242): with r = Ic - aa * S, S = ii * (n . d) / |d|^3,
// d = light - p,
//   grad_p r = grad_p Ic - aa * ii * (-n / |d|^3 + 3 (n . d) d / |d|^5)
//   grad_n r =           - aa * ii * d / |d|^3
//   J[3..5] = grad_p r,   J[0..2] = p x grad_p r + n x grad_n r
// (grad_p Ic and p x grad_p Ic are ColorTracker's expressions).
template <bool TRANSLATION, bool JACOBIAN>
__device__ __forceinline__ bool evaluate_light(const ColorParams& P, const Rt& Tcm, int keyframe_x, int keyframe_y,
    float& residual, float (&J)[6])
{
  residual = 0.0f;
#pragma unroll
  for (int i = 0; i < 6; ++i) J[i] = 0.0f;

  const vk_color_view& key = P.key;
  const vk_color_view& frm = P.frm;

  const int keyframe_index = keyframe_y * key.width + keyframe_x;
  const float keyframe_depth = key.depths[keyframe_index];
  if (!(keyframe_depth > 0)) return false;

  const f3 Xmp = unproject_d(key.projection, keyframe_x + 0.5f, keyframe_y + 0.5f, keyframe_depth);
  const f3 Xcp = xform_point(Tcm, Xmp);
  float fu, fv;
  project(frm.projection, Xcp, fu, fv);

  if (!(fu >= 0.5f && fu < frm.width - 0.5f && fv >= 0.5f && fv < frm.height - 0.5f)) return false;

  const int frame_x = f2i(fu), frame_y = f2i(fv);
  const int frame_index = frame_y * frm.width + frame_x;
  const float frame_depth = frm.depths[frame_index];
  if (!(fabsf(frame_depth - Xcp.z) < 0.2f)) return false;

  const f3 frame_Xdn = make3(frm.normals[3 * frame_index + 0], frm.normals[3 * frame_index + 1],
      frm.normals[3 * frame_index + 2]);
  const f3 frame_normal = xform_dir(P.Tcd, frame_Xdn);
  f3 n = make3(key.normals[3 * keyframe_index + 0], key.normals[3 * keyframe_index + 1],
      key.normals[3 * keyframe_index + 2]);
  n = xform_dir(Tcm, n);

  if (!(sqnorm3(n) > 0.5f && dot3(frame_normal, n) > 0.8f)) return false;

  const float px = Xcp.x, py = Xcp.y, pz = Xcp.z;

  if (P.mask[frame_index] > 0.5f)
  {
    const float aa = key.intensities[keyframe_index];
    if (!(aa > 0)) return false;

    // light.h:53-60 GetShading
    const f3 d = sub3(make3(P.light.position[0], P.light.position[1], P.light.position[2]), Xcp);
    const float d2 = sqnorm3(d);
    const float dn = sqrtf(d2);
    const float inv_dn = 1.0f / dn;
    const f3 direction = scale3(d, inv_dn);
    const float cos_theta = dot3(n, direction);
    const float ii = P.light.intensity;
    const float shading = ii * cos_theta / d2;
    const float Im = shading * aa;
    const float Ic = sample(frm.width, frm.intensities, fu, fv);
    residual = Ic - Im;

    if (JACOBIAN)
    {
      const float inv_pz = 1.0f / pz;
      const float cu = fu, cv = fv;
      const float fx = frm.projection.fx, fy = frm.projection.fy;
      const float cx = frm.projection.cx, cy = frm.projection.cy;
      const float gx = sample(frm.width, frm.gradient_x, fu, fv);
      const float gy = sample(frm.width, frm.gradient_y, fu, fv);

      float I[6];
      I[0] = gy * ((py * cy - pz * fy) * inv_pz - py * cv * inv_pz) - gx * (py * cu * inv_pz - cx * py * inv_pz);
      I[1] = gy * (px * cv * inv_pz - cy * px * inv_pz) - gx * ((px * cx - pz * fx) * inv_pz - px * cu * inv_pz);
      I[2] = (gy * fy * px - gx * fx * py) * inv_pz;
      I[3] = gx * fx * inv_pz;
      I[4] = gy * fy * inv_pz;
      I[5] = (gx * (cx - cu) + gy * (cy - cv)) * inv_pz;

      const float nd = dot3(n, d);
      const float inv_d3 = 1.0f / (d2 * dn);
      const float inv_d5 = inv_d3 / d2;
      const float k3 = 3.0f * nd * inv_d5;
      const float s = aa * ii;
      const f3 a = make3(s * (k3 * d.x - n.x * inv_d3), s * (k3 * d.y - n.y * inv_d3), s * (k3 * d.z - n.z * inv_d3));
      const f3 b = make3(s * d.x * inv_d3, s * d.y * inv_d3, s * d.z * inv_d3);
      const f3 pa = cross3(Xcp, a);
      const f3 nb = cross3(n, b);

      J[0] = I[0] - (pa.x + nb.x);
      J[1] = I[1] - (pa.y + nb.y);
      J[2] = I[2] - (pa.z + nb.z);
      if (TRANSLATION)
      {
        J[3] = I[3] - a.x;
        J[4] = I[4] - a.y;
        J[5] = I[5] - a.z;
      }
    }
  }
  else
  {
    // :283-322 default to standard depth tracking
    const f3 Xcq = unproject_d(frm.projection, frame_x + 0.5f, frame_y + 0.5f, frame_depth);
    const f3 delta = sub3(Xcp, Xcq);
    residual = dot3(delta, n);

    if (JACOBIAN)
    {
      J[0] = delta.z * n.y - delta.y * n.z - n.y * pz + n.z * py;
      J[1] = delta.x * n.z - delta.z * n.x + n.x * pz - n.z * px;
      J[2] = delta.y * n.x - delta.x * n.y - n.x * py + n.y * px;
      if (TRANSLATION)
      {
        J[3] = n.x;
        J[4] = n.y;
        J[5] = n.z;
      }
    }
  }
  return true;
}

// one entry for both trackers
template <bool LIGHT, bool TRANSLATION, bool JACOBIAN>
__device__ __forceinline__ bool evaluate_any(const ColorParams& P, const Rt& Tcm, int x, int y, float& residual,
    float (&J)[6])
{
  if (LIGHT) return evaluate_light<TRANSLATION, JACOBIAN>(P, Tcm, x, y, residual, J);
  return evaluate<TRANSLATION, JACOBIAN>(P, Tcm, x, y, residual, J);
}

__device__ __forceinline__ Rt rt_of(const float* m)   // column-major 4x4 -> rows 0..2
{
  Rt t;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) t.r[r * 4 + c] = m[c * 4 + r];
  return t;
}

// ref: color_tracker.cu:140-163
template <bool LIGHT>
__global__ __launch_bounds__(256) void color_residuals_kernel(ColorParams P, float* __restrict__ residuals)
{
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= P.key.width || y >= P.key.height) return;
  float r, J[6];
  evaluate_any<LIGHT, false, false>(P, P.Tcm, x, y, r, J);
  residuals[y * P.key.width + x] = r;
}

// ref: color_tracker.cu:165-204
template <bool LIGHT, bool TRANSLATION>
__global__ __launch_bounds__(256) void color_jacobian_kernel(ColorParams P, float* __restrict__ jacobian)
{
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= P.key.width || y >= P.key.height) return;
  float r, J[6];
  evaluate_any<LIGHT, TRANSLATION, true>(P, P.Tcm, x, y, r, J);
  float* out = jacobian + 6 * (size_t)(y * P.key.width + x);
#pragma unroll
  for (int i = 0; i < 6; ++i) out[i] = J[i];
}

// One workgroup of kColorThreads lanes per group of keyframe pixels (group_pixels_for,
// vk_gauss_newton.hpp).
#ifndef VK_COLOR_THREADS
#define VK_COLOR_THREADS 512
#endif
constexpr int kColorThreads = VK_COLOR_THREADS;

// the 27 products of this lane's pixels of one group, added pixel by pixel onto acc
template <bool LIGHT, bool TRANSLATION>
__device__ __forceinline__ void accumulate_group(const ColorParams& P, const Rt& Tcm, int group, float (&acc)[27])
{
  const int total = P.key.width * P.key.height;
  const int first = group * P.group_pixels;
  for (int p = (int)threadIdx.x; p < P.group_pixels; p += kColorThreads)
  {
    const int pixel = first + p;
    float r, J[6], one[27];
    if (pixel < total && evaluate_any<LIGHT, TRANSLATION, true>(P, Tcm, pixel % P.key.width, pixel / P.key.width, r, J))
    {
      outer_products(J, r, one);
#pragma unroll
      for (int i = 0; i < 27; ++i) acc[i] += one[i];
    }
  }
}

// ref: color_tracker.cu:206-343, first stage (see vk_gauss_newton.hpp)
template <bool LIGHT, bool TRANSLATION>
__global__ __launch_bounds__(kColorThreads) void color_partial_kernel(ColorParams P, float* __restrict__ workspace)
{
  __shared__ float lds[kColorThreads / 64][kSysStride];

  if (P.state && P.state[1]) return;   // tracker.cpp:162, see system_partial_kernel

  const Rt Tcm = P.Tcm_dev ? rt_of(P.Tcm_dev->m) : P.Tcm;
  float acc[27];
#pragma unroll
  for (int i = 0; i < 27; ++i) acc[i] = 0.0f;
  accumulate_group<LIGHT, TRANSLATION>(P, Tcm, blockIdx.x, acc);
  store_partial<kColorThreads / 64>(acc, lds, workspace);
}

// ------------------------------------------------------------ pose update ----

// color_tracker.cu:312-320: Tcm = (frame_Tcd * frame_Twd^-1) * keyframe_Tcw^-1 with
// (A * B).m = A.m * B.m, (A * B).inv = B.inv * A.inv (transform.h:146-159)
__device__ __forceinline__ void derive_tcm(const vk_transform& frame_Tcd, const vk_transform& key_Twc,
    vk_color_pose* pose)
{
  float tcd_m[16], tcd_i[16], twd_m[16], twd_i[16], twc_m[16], twc_i[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
  {
    tcd_m[i] = frame_Tcd.m[i];  tcd_i[i] = frame_Tcd.inv[i];
    twd_m[i] = pose->depth_to_world.m[i];  twd_i[i] = pose->depth_to_world.inv[i];
    twc_m[i] = key_Twc.m[i];  twc_i[i] = key_Twc.inv[i];
  }
  float tcw_m[16], tcw_i[16], out_m[16], out_i[16];
  matmul4(tcd_m, twd_i, tcw_m);    // frame_Tcw.m   = Tcd.m * (Twd^-1).m
  matmul4(twd_m, tcd_i, tcw_i);    // frame_Tcw.inv = (Twd^-1).inv * Tcd.inv
  matmul4(tcw_m, twc_m, out_m);
  matmul4(twc_i, tcw_i, out_i);
#pragma unroll
  for (int i = 0; i < 16; ++i) { pose->Tcm.m[i] = out_m[i]; pose->Tcm.inv[i] = out_i[i]; }
}

// color_tracker.cu:312-320 on arrays: Tcm from the frame's and the keyframe's poses
__device__ __forceinline__ void derive_tcm_arrays(const vk_transform& frame_Tcd, const vk_transform& key_Twc,
    const float (&twd_m)[16], const float (&twd_i)[16], float (&out_m)[16], float (&out_i)[16])
{
  float tcd_m[16], tcd_i[16], twc_m[16], twc_i[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
  {
    tcd_m[i] = frame_Tcd.m[i];  tcd_i[i] = frame_Tcd.inv[i];
    twc_m[i] = key_Twc.m[i];  twc_i[i] = key_Twc.inv[i];
  }
  float tcw_m[16], tcw_i[16];
  matmul4(tcd_m, twd_i, tcw_m);    // frame_Tcw.m   = Tcd.m * (Twd^-1).m
  matmul4(twd_m, tcd_i, tcw_i);    // frame_Tcw.inv = (Twd^-1).inv * Tcd.inv
  matmul4(tcw_m, twc_m, out_m);
  matmul4(twc_i, tcw_i, out_i);
}

// ref: tracker.cpp:124-163 + color_tracker.cpp:34-96 on arrays: the update from the system,
// the new depth_to_world (m, inv) from the old inverse
template <int N>
__device__ __forceinline__ void color_pose_matrix(const float* hessian, const float* gradient, const float (&old_i)[16],
    float (&M)[16], float (&update)[6])
{
  solve_step<N>(hessian, gradient, update);

  // color_tracker.cpp:45-65: a proper skew matrix (DepthTracker's has Tinc(1,2) = +u0)
  float Tinc[16];
  Tinc[0] = 1.0f;        Tinc[4] = -update[2]; Tinc[8] = +update[1];  Tinc[12] = +update[3];
  Tinc[1] = +update[2];  Tinc[5] = 1.0f;       Tinc[9] = -update[0];  Tinc[13] = +update[4];
  Tinc[2] = -update[1];  Tinc[6] = +update[0]; Tinc[10] = 1.0f;       Tinc[14] = +update[5];
  Tinc[3] = 0.0f;        Tinc[7] = 0.0f;       Tinc[11] = 0.0f;       Tinc[15] = 1.0f;

  matmul4(Tinc, old_i, M);             // :67  M = Tinc * Twd^-1
}

template <int N>
__device__ __forceinline__ void color_pose_step(const float* hessian, const float* gradient, const float (&old_i)[16],
    float (&twd_m)[16], float (&twd_i)[16], float (&update)[6])
{
  float M[16];
  color_pose_matrix<N>(hessian, gradient, old_i, M, update);
  rigid_from(M, twd_i, twd_m);         // :69-95 world -> depth, re-orthonormalised; .Inverse() swaps the two
}

// one lane
template <int N>
__device__ __forceinline__ void color_solve_update_n(const float* hessian, const float* gradient,
    const vk_transform& frame_Tcd, const vk_transform& key_Twc, vk_color_pose* pose, int32_t* state,
    float* update_out, Mirror mirror)
{
  float update[6], old_i[16], twd_m[16], twd_i[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) old_i[i] = pose->depth_to_world.inv[i];
  color_pose_step<N>(hessian, gradient, old_i, twd_m, twd_i, update);
#pragma unroll
  for (int i = 0; i < 16; ++i) { pose->depth_to_world.m[i] = twd_m[i]; pose->depth_to_world.inv[i] = twd_i[i]; }

  derive_tcm(frame_Tcd, key_Twc, pose);
  finish_step<N>(update, state, update_out, mirror);
}

__device__ void color_solve_update(const float* hessian, const float* gradient, int translation_enabled,
    const vk_transform& frame_Tcd, const vk_transform& key_Twc, vk_color_pose* pose, int32_t* state,
    float* update_out, Mirror mirror)
{
  if (state && state[1]) return;  // converged earlier: tracker.cpp:162
  if (translation_enabled) color_solve_update_n<6>(hessian, gradient, frame_Tcd, key_Twc, pose, state, update_out, mirror);
  else color_solve_update_n<3>(hessian, gradient, frame_Tcd, key_Twc, pose, state, update_out, mirror);
}

struct PoseArgs
{
  vk_transform frame_Tcd, key_Twc;
  vk_color_pose* pose;      // null: sums only
  int32_t* state;
  float* update_out;
  Mirror mirror;   // pinned host {iterations, converged}, or null (vk_track_poll)
};

// second stage; with a pose it also solves and updates (one workgroup)
__global__ __launch_bounds__(256) void color_final_kernel(const float* __restrict__ workspace, int partials,
    int translation_enabled, float* __restrict__ hessian, float* __restrict__ gradient, PoseArgs A)
{
  __shared__ float slices[kSysSlices][kSysStride];
  __shared__ float sums[48];
  if (A.state && A.state[1]) return;
  sum_partials(workspace, partials, translation_enabled, hessian, gradient, slices, sums);
  if (A.pose && threadIdx.x == 0)
    color_solve_update(sums, sums + 36, translation_enabled, A.frame_Tcd, A.key_Twc, A.pose, A.state, A.update_out, A.mirror);
}

__global__ void color_solve_kernel(const float* __restrict__ hessian, const float* __restrict__ gradient,
    int translation_enabled, PoseArgs A)
{
  if (threadIdx.x == 0 && blockIdx.x == 0)
    color_solve_update(hessian, gradient, translation_enabled, A.frame_Tcd, A.key_Twc, A.pose, A.state, A.update_out, A.mirror);
}

__global__ void color_publish_pose_kernel(Mirror mirror, const vk_color_pose* pose)
{
  publish_host_pose(mirror, &pose->depth_to_world);
}

__global__ void color_prepare_kernel(PoseArgs A)
{
  if (threadIdx.x == 0 && blockIdx.x == 0) derive_tcm(A.frame_Tcd, A.key_Twc, A.pose);
}

// ---- the whole Gauss-Newton loop of the photometric trackers in one launch ----------
//
// As track_loop_kernel of the depth tracker (vk_icp.hip; the exchange is described in
// vk_gauss_newton.hpp): every workgroup evaluates its keyframe pixels, the workgroups
// exchange their 27 sums inside the launch, every workgroup adds all of them, solves and
// moves depth_to_world and Tcm itself, in LDS; workgroup 0 publishes once, at the end.
struct ColorLoopParams
{
  Exchange exchange;
  vk_transform frame_Tcd, key_Twc;
  vk_color_pose* pose;     // in: depth_to_world; out: depth_to_world and Tcm after the loop
  int groups;
  int iterations;
  int fresh_state;
  int force_abort;         // test aid, vk_forced_loop_abort()
  int last_launch;         // 1: this launch ends the Track (it leaves the pose for vk_track_wait)
  float* hessian;
  float* gradient;
  int32_t* state;
  float* update_out;
  Mirror mirror;
  VK_LOOP_TIMING_FIELD
};

template <bool LIGHT, bool TRANSLATION>
__global__ __launch_bounds__(kColorThreads, 1024 / kColorThreads) void color_loop_kernel(ColorParams P, ColorLoopParams L)
{
  constexpr int N = TRANSLATION ? 6 : 3;
  __shared__ float lds[kColorThreads / 64][kSysStride];
  __shared__ float slices[kSysSlices][kSysStride];
  __shared__ float sums[48];
  __shared__ float twd[32];      // depth_to_world: matrix, inverse
  __shared__ float tcm[32];      // Tcm: matrix, inverse
  __shared__ float last_update[6];
  __shared__ float last_M[16];   // workgroup 0: Tinc * Twd^-1 of the last step
  __shared__ float solve_scratch[64];   // wave_solve_step / wave_rigid_from
  __shared__ float fixed_m[32];         // frame_Tcd.m, key_Twc.m: indexed per lane by the wave-wide products
  __shared__ int stop, failed;

  const int steps_before = L.fresh_state ? 0 : L.state[0];
  if (!L.fresh_state && L.state[1])           // uniform over the grid
  {
    // converged earlier: the pose stands. Aborted earlier (a level of a coarse-to-fine Track):
    // no pose is published, the host sees the Track fail and runs it again, launch per stage
    if (blockIdx.x == 0 && L.last_launch && L.state[1] != VK_TRACK_ABORTED) publish_host_pose(L.mirror, &L.pose->depth_to_world);
    return;
  }
  if (L.force_abort)
  {
    if (blockIdx.x == 0 && threadIdx.x == 0) L.state[1] = VK_TRACK_ABORTED;
    return;
  }

  if (threadIdx.x < 32)
  {
    twd[threadIdx.x] = threadIdx.x < 16 ? L.pose->depth_to_world.m[threadIdx.x] : L.pose->depth_to_world.inv[threadIdx.x - 16];
    fixed_m[threadIdx.x] = threadIdx.x < 16 ? L.frame_Tcd.m[threadIdx.x] : L.key_Twc.m[threadIdx.x - 16];
  }
  if (threadIdx.x == 0) { stop = 0; failed = 0; }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    // Tracker::BeginSolve: Tcm of the pose the loop starts from
    float m[16], i[16], out_m[16], out_i[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { m[k] = twd[k]; i[k] = twd[16 + k]; }
    derive_tcm_arrays(L.frame_Tcd, L.key_Twc, m, i, out_m, out_i);
#pragma unroll
    for (int k = 0; k < 16; ++k) { tcm[k] = out_m[k]; tcm[16 + k] = out_i[k]; }
  }
  __syncthreads();

  const bool publisher = blockIdx.x == 0;
  int steps = 0;
  for (int it = 0; it < L.iterations; ++it)
  {
    VK_STAMP(0);
    const Rt Tcm = rt_of(tcm);
    for (int group = blockIdx.x; group < L.groups; group += gridDim.x)
    {
      if (group != (int)blockIdx.x) __syncthreads();   // the previous group's sums have left the LDS
      float acc[27];
#pragma unroll
      for (int i = 0; i < 27; ++i) acc[i] = 0.0f;
      accumulate_group<LIGHT, TRANSLATION>(P, Tcm, group, acc);
      VK_STAMP(1);
      publish_partial<kColorThreads / 64>(acc, lds, L.exchange, it, group);
    }
    VK_STAMP(2);
    VK_STAMP(3);
    if (!gather_partials<kColorThreads>(L.exchange, it, TRANSLATION, publisher ? L.hessian : nullptr,
            publisher ? L.gradient : nullptr, slices, sums, &failed))
      break;
    steps = it + 1;
    VK_STAMP(4);

#ifdef VK_SCALAR_SOLVE
    if (threadIdx.x == 0)
    {
      // A step needs depth_to_world^-1 (the next update multiplies it) and Tcm's matrix (the
      // pixels); depth_to_world itself and Tcm^-1 — three more 4x4 products per step — are made
      // once, after the loop, from the last step's M. (The unused halves are dead code here.)
      float update[6], old_i[16], M[16], m[16], i[16], out_m[16], out_i[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) old_i[k] = twd[16 + k];
      color_pose_matrix<N>(sums, sums + 36, old_i, M, update);
      rigid_from(M, i, m);
      derive_tcm_arrays(L.frame_Tcd, L.key_Twc, m, i, out_m, out_i);
      float sq = 0.0f;
#pragma unroll
      for (int k = 0; k < N; ++k) sq += update[k] * update[k];
      stop = (sqrtf(sq) < 1E-6f) ? 1 : 0;
#pragma unroll
      for (int k = 0; k < 16; ++k) { twd[16 + k] = i[k]; tcm[k] = out_m[k]; }
      if (publisher)
      {
#pragma unroll
        for (int k = 0; k < 16; ++k) last_M[k] = M[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) last_update[k] = update[k];
      }
    }
#else
    if (threadIdx.x < 64)
    {
      // The same step across the lanes of the first wave (vk_gauss_newton.hpp wave_solve_step: the
      // bits of the one-lane code). A step needs depth_to_world^-1 (the next update multiplies it)
      // and Tcm's matrix (the pixels); depth_to_world itself and Tcm^-1 are made once, after the
      // loop, from the last step's M.
      float update[6];
      wave_solve_step<N>(sums, solve_scratch, update);
      // color_tracker.cpp:45-65: a proper skew matrix (DepthTracker's has Tinc(1,2) = +u0); element l = c * 4 + r
      const int l = (int)threadIdx.x & 15;
      float tinc = (l % 5 == 0) ? 1.0f : 0.0f;
      tinc = (l == 4) ? -update[2] : tinc;  tinc = (l == 8) ? +update[1] : tinc;  tinc = (l == 12) ? +update[3] : tinc;
      tinc = (l == 1) ? +update[2] : tinc;  tinc = (l == 9) ? -update[0] : tinc;  tinc = (l == 13) ? +update[4] : tinc;
      tinc = (l == 2) ? -update[1] : tinc;  tinc = (l == 6) ? +update[0] : tinc;  tinc = (l == 14) ? +update[5] : tinc;
      if (threadIdx.x < 16) solve_scratch[threadIdx.x] = tinc;
      wave_lds_fence();
      const float M_lane = matmul4_lane(solve_scratch, twd + 16, (int)threadIdx.x);       // :67  M = Tinc * Twd^-1
      wave_lds_fence();
      const float inv_lane = wave_rigid_from(M_lane, solve_scratch);                       // :69-95: the new Twd^-1
      if (threadIdx.x < 16) twd[16 + threadIdx.x] = inv_lane;
      wave_lds_fence();
      const float tcw_lane = matmul4_lane(fixed_m, twd + 16, (int)threadIdx.x);            // frame_Tcw.m = Tcd.m * (Twd^-1).m
      if (threadIdx.x < 16) solve_scratch[32 + threadIdx.x] = tcw_lane;
      wave_lds_fence();
      const float tcm_lane = matmul4_lane(solve_scratch + 32, fixed_m + 16, (int)threadIdx.x);   // Tcm.m = frame_Tcw.m * key_Twc.m
      wave_lds_fence();
      float sq = 0.0f;
#pragma unroll
      for (int k = 0; k < N; ++k) sq += update[k] * update[k];
      if (threadIdx.x < 16)
      {
        tcm[threadIdx.x] = tcm_lane;
        if (publisher) last_M[threadIdx.x] = M_lane;
      }
      if (threadIdx.x == 0)
      {
        stop = (sqrtf(sq) < 1E-6f) ? 1 : 0;
        if (publisher)
        {
#pragma unroll
          for (int k = 0; k < 6; ++k) last_update[k] = update[k];
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
    if (threadIdx.x == 0) L.state[1] = VK_TRACK_ABORTED;
    return;
  }
  if (!publisher) return;
  if (steps > 0)
  {
    if (threadIdx.x == 0)
    {
      float M[16], m[16], i[16], out_m[16], out_i[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) M[k] = last_M[k];
      rigid_from(M, i, m);
      derive_tcm_arrays(L.frame_Tcd, L.key_Twc, m, i, out_m, out_i);
#pragma unroll
      for (int k = 0; k < 16; ++k) { twd[k] = m[k]; twd[16 + k] = i[k]; tcm[k] = out_m[k]; tcm[16 + k] = out_i[k]; }
    }
    __syncthreads();
  }
  // the derived Tcm is part of the pose even when no step ran (color_prepare_kernel's job)
  if (threadIdx.x < 32)
  {
    const float v = tcm[threadIdx.x];
    if (threadIdx.x < 16) L.pose->Tcm.m[threadIdx.x] = v; else L.pose->Tcm.inv[threadIdx.x - 16] = v;
  }
  if (steps > 0)
  {
    if (threadIdx.x < 32)
    {
      const float v = twd[threadIdx.x];
      if (threadIdx.x < 16) L.pose->depth_to_world.m[threadIdx.x] = v; else L.pose->depth_to_world.inv[threadIdx.x - 16] = v;
    }
    if (threadIdx.x < 6 && L.update_out) L.update_out[threadIdx.x] = last_update[threadIdx.x];
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
    __syncthreads();
    publish_host_pose(L.mirror, &L.pose->depth_to_world);
  }
}

template <bool LIGHT, bool TRANSLATION>
int launch_color_loop_of(const ColorParams& P, ColorLoopParams& L, int iterations, float* workspace, hipStream_t s)
{
  const int capacity = resident_workgroups(color_loop_kernel<LIGHT, TRANSLATION>, kColorThreads);
  if (capacity <= 0) return VK_ERR_ARGUMENT;
  const int grid = L.groups < capacity ? L.groups : capacity;
  for (int done = 0; done < iterations; done += kExchangeSteps)
  {
    L.exchange.words = reinterpret_cast<unsigned long long*>(workspace);
    L.exchange.count = L.groups;
    { const int rc = vk_loop_epoch_begin(workspace, exchange_floats(L.groups) * sizeof(float), s, &L.exchange.epoch);  if (rc != VK_OK) return rc; }
    VK_LOOP_TIMING_ATTACH(L, s);
    L.iterations = iterations - done < kExchangeSteps ? iterations - done : kExchangeSteps;
    L.last_launch = done + kExchangeSteps >= iterations ? 1 : 0;
    L.force_abort = vk_forced_loop_abort();
    ColorParams Pk = P;
    vk_loop_launch_begin(s);
    const hipError_t le = launch_loop_kernel(color_loop_kernel<LIGHT, TRANSLATION>, grid, kColorThreads, s, Pk, L);
    vk_loop_launch_end(s);
    VK_CHECK(le);
    VK_LAUNCH_CHECK();
    L.fresh_state = 0;
  }
  return VK_OK;
}

// -------------------------------------------------------------- host side ----

int fill_color(ColorParams& P, const vk_color_view* keyframe, const vk_color_view* frame, const vk_transform* Tcm,
    bool need_gradients)
{
  if (!keyframe || !frame || !Tcm) return VK_ERR_ARGUMENT;
  if (!keyframe->depths || !keyframe->normals || !keyframe->intensities) return VK_ERR_ARGUMENT;
  if (!frame->depths || !frame->normals || !frame->intensities) return VK_ERR_ARGUMENT;
  if (need_gradients && (!frame->gradient_x || !frame->gradient_y)) return VK_ERR_ARGUMENT;
  // the bilinear taps need a 2x2 neighbourhood inside the frame (color_tracker.cu:22)
  if (keyframe->width <= 0 || keyframe->height <= 0 || frame->width < 2 || frame->height < 2)
    return VK_ERR_ARGUMENT;
  P.key = *keyframe;
  P.frm = *frame;
  P.Tcm = make_rt(Tcm->m);
  P.Tcm_dev = nullptr;
  P.state = nullptr;
  P.group_pixels = group_pixels_for(keyframe->width * keyframe->height);
  P.mask = nullptr;
  P.light.intensity = 1.0f;
  P.light.position[0] = P.light.position[1] = P.light.position[2] = 0.0f;
  P.Tcd = P.Tcm;
  return VK_OK;
}

int fill_light(ColorParams& P, const vk_light_terms* terms)
{
  if (!terms || !terms->frame_mask) return VK_ERR_ARGUMENT;
  P.mask = terms->frame_mask;
  P.light = terms->light;
  P.Tcd = make_rt(terms->frame_Tcd.m);
  return VK_OK;
}

template <bool LIGHT>
void launch_partials_of(const ColorParams& P, int translation_enabled, int partials, float* workspace, hipStream_t s)
{
  if (translation_enabled)
    hipLaunchKernelGGL((color_partial_kernel<LIGHT, true>), dim3(partials), dim3(kColorThreads), 0, s, P, workspace);
  else
    hipLaunchKernelGGL((color_partial_kernel<LIGHT, false>), dim3(partials), dim3(kColorThreads), 0, s, P, workspace);
}

void launch_color_partials(const ColorParams& P, int translation_enabled, int partials, float* workspace, hipStream_t s)
{
  vk_loop_area_written(workspace);      // float partials over the loop kernels' tagged words: the next loop launch clears them
  if (P.mask) launch_partials_of<true>(P, translation_enabled, partials, workspace, s);
  else launch_partials_of<false>(P, translation_enabled, partials, workspace, s);
}

vk_transform identity_transform()
{
  vk_transform t;
  for (int i = 0; i < 16; ++i) t.m[i] = t.inv[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  return t;
}

}  // namespace

extern "C" {

VK_API int vk_color_image_convert(int total, const float* src, float* dst, void* stream)
{
  VK_REQUIRE(total >= 0 && (total == 0 || (src && dst)));
  if (total == 0) return VK_OK;
  hipLaunchKernelGGL(convert_kernel, dim3((total + 255) / 256), dim3(256), 0, vk_s(stream), total, src, dst);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

VK_API int vk_color_tracker_begin(const vk_frame* keyframe, const vk_frame* frame, float* keyframe_intensities,
    float* frame_intensities, float* gradient_x, float* gradient_y, float depth_threshold, float* frame_mask,
    const vk_transform* pose_host, vk_color_pose* pose_dev, int32_t* state_dev, void* stream)
{
  VK_REQUIRE(keyframe && frame && keyframe->color && frame->color && keyframe_intensities && frame_intensities);
  VK_REQUIRE(gradient_x && gradient_y && keyframe->width > 0 && keyframe->height > 0 && frame->width > 0 && frame->height > 0);
  VK_REQUIRE(!frame_mask || frame->depth);
  VK_REQUIRE((pose_host != nullptr) == (pose_dev != nullptr));
  // the trackers read the colour image with the depth image's size (color_tracker.cu:296-343)
  VK_REQUIRE((frame->color_width <= 0 || frame->color_width == frame->width) &&
             (frame->color_height <= 0 || frame->color_height == frame->height));
  VK_REQUIRE((keyframe->color_width <= 0 || keyframe->color_width == keyframe->width) &&
             (keyframe->color_height <= 0 || keyframe->color_height == keyframe->height));
  BeginParams B;
  B.key_color = keyframe->color;
  B.key_intensities = keyframe_intensities;
  B.key_total = keyframe->width * keyframe->height;
  B.frm_color = frame->color;
  B.frm_intensities = frame_intensities;
  B.gradient_x = gradient_x;
  B.gradient_y = gradient_y;
  B.width = frame->width;
  B.height = frame->height;
  B.frm_depth = frame->depth;
  B.depth_threshold = depth_threshold;
  B.mask = frame_mask;
  if (pose_host) B.pose = *pose_host; else B.pose = identity_transform();
  B.pose_dev = pose_dev;
  B.state_dev = state_dev;
  const int total = B.key_total > B.width * B.height ? B.key_total : B.width * B.height;
  const int tiles = ((B.width + 15) / 16) * ((B.height + 15) / 16);      // the mask job: one workgroup per 16x16 tile
  const int groups = (total + 255) / 256 > tiles ? (total + 255) / 256 : tiles;
  const int gx = groups < 1024 ? groups : 1024;
  hipLaunchKernelGGL(color_begin_kernel, dim3(gx, (groups + gx - 1) / gx, frame_mask ? 4 : 3), dim3(256), 0, vk_s(stream), B);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

VK_API int vk_image_gradients(int width, int height, const float* src, float* gradient_x, float* gradient_y,
    void* stream)
{
  VK_REQUIRE(src && gradient_x && gradient_y && width > 0 && height > 0);
  const dim3 grid((width + 63) / 64, (height + 3) / 4);
  hipLaunchKernelGGL(gradients_kernel, grid, dim3(256), 0, vk_s(stream), width, height, src, gradient_x, gradient_y);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

static int residuals_impl(const vk_color_view* keyframe, const vk_color_view* frame, const vk_light_terms* terms,
    bool light, const vk_transform* Tcm, float* residuals, void* stream)
{
  ColorParams P;
  int rc = fill_color(P, keyframe, frame, Tcm, false);
  if (rc != VK_OK) return rc;
  if (light && (rc = fill_light(P, terms)) != VK_OK) return rc;
  VK_REQUIRE(residuals);
  const dim3 grid((keyframe->width + 63) / 64, (keyframe->height + 3) / 4);
  if (light) hipLaunchKernelGGL(color_residuals_kernel<true>, grid, dim3(256), 0, vk_s(stream), P, residuals);
  else hipLaunchKernelGGL(color_residuals_kernel<false>, grid, dim3(256), 0, vk_s(stream), P, residuals);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

VK_API int vk_color_tracker_compute_residuals(const vk_color_view* keyframe, const vk_color_view* frame,
    const vk_transform* Tcm, float* residuals, void* stream)
{
  return residuals_impl(keyframe, frame, nullptr, false, Tcm, residuals, stream);
}

VK_API int vk_light_tracker_compute_residuals(const vk_color_view* keyframe, const vk_color_view* frame,
    const vk_light_terms* terms, const vk_transform* Tcm, float* residuals, void* stream)
{
  return residuals_impl(keyframe, frame, terms, true, Tcm, residuals, stream);
}

static int jacobian_impl(const vk_color_view* keyframe, const vk_color_view* frame, const vk_light_terms* terms,
    bool light, const vk_transform* Tcm, int translation_enabled, float* jacobian, void* stream)
{
  ColorParams P;
  int rc = fill_color(P, keyframe, frame, Tcm, true);
  if (rc != VK_OK) return rc;
  if (light && (rc = fill_light(P, terms)) != VK_OK) return rc;
  VK_REQUIRE(jacobian);
  const dim3 grid((keyframe->width + 63) / 64, (keyframe->height + 3) / 4);
  hipStream_t s = vk_s(stream);
  if (light)
  {
    if (translation_enabled) hipLaunchKernelGGL((color_jacobian_kernel<true, true>), grid, dim3(256), 0, s, P, jacobian);
    else hipLaunchKernelGGL((color_jacobian_kernel<true, false>), grid, dim3(256), 0, s, P, jacobian);
  }
  else
  {
    if (translation_enabled) hipLaunchKernelGGL((color_jacobian_kernel<false, true>), grid, dim3(256), 0, s, P, jacobian);
    else hipLaunchKernelGGL((color_jacobian_kernel<false, false>), grid, dim3(256), 0, s, P, jacobian);
  }
  VK_LAUNCH_CHECK();
  return VK_OK;
}

VK_API int vk_color_tracker_compute_jacobian(const vk_color_view* keyframe, const vk_color_view* frame,
    const vk_transform* Tcm, int translation_enabled, float* jacobian, void* stream)
{
  return jacobian_impl(keyframe, frame, nullptr, false, Tcm, translation_enabled, jacobian, stream);
}

VK_API int vk_light_tracker_compute_jacobian(const vk_color_view* keyframe, const vk_color_view* frame,
    const vk_light_terms* terms, const vk_transform* Tcm, int translation_enabled, float* jacobian, void* stream)
{
  return jacobian_impl(keyframe, frame, terms, true, Tcm, translation_enabled, jacobian, stream);
}

static int system_impl(const vk_color_view* keyframe, const vk_color_view* frame, const vk_light_terms* terms,
    bool light, const vk_transform* Tcm, const vk_transform* Tcm_dev, int translation_enabled, float* workspace,
    float* hessian, float* gradient, void* stream)
{
  ColorParams P;
  const vk_transform identity = identity_transform();
  if (!Tcm && Tcm_dev) Tcm = &identity;
  int rc = fill_color(P, keyframe, frame, Tcm, true);
  if (rc != VK_OK) return rc;
  if (light && (rc = fill_light(P, terms)) != VK_OK) return rc;
  VK_REQUIRE(workspace && hessian && gradient);
  P.Tcm_dev = Tcm_dev;
  const int partials = group_count_for(keyframe->width * keyframe->height, P.group_pixels);
  launch_color_partials(P, translation_enabled, partials, workspace, vk_s(stream));
  VK_LAUNCH_CHECK();
  PoseArgs A;
  A.frame_Tcd = identity;
  A.key_Twc = identity;
  A.pose = nullptr;
  A.state = nullptr;
  A.update_out = nullptr;
  A.mirror = Mirror{nullptr, 0, nullptr};
  hipLaunchKernelGGL(color_final_kernel, dim3(1), dim3(256), 0, vk_s(stream), workspace, partials,
      translation_enabled, hessian, gradient, A);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

VK_API int vk_color_tracker_compute_system(const vk_color_view* keyframe, const vk_color_view* frame,
    const vk_transform* Tcm, const vk_transform* Tcm_dev, int translation_enabled, float* workspace,
    float* hessian, float* gradient, void* stream)
{
  return system_impl(keyframe, frame, nullptr, false, Tcm, Tcm_dev, translation_enabled, workspace, hessian,
      gradient, stream);
}

VK_API int vk_light_tracker_compute_system(const vk_color_view* keyframe, const vk_color_view* frame,
    const vk_light_terms* terms, const vk_transform* Tcm, const vk_transform* Tcm_dev, int translation_enabled,
    float* workspace, float* hessian, float* gradient, void* stream)
{
  return system_impl(keyframe, frame, terms, true, Tcm, Tcm_dev, translation_enabled, workspace, hessian,
      gradient, stream);
}

VK_API int vk_color_tracker_solve_update(const float* hessian, const float* gradient, int translation_enabled,
    const vk_transform* frame_Tcd, const vk_transform* keyframe_Twc, vk_color_pose* pose_dev,
    int32_t* state_dev, float* update_dev, void* stream)
{
  VK_REQUIRE(hessian && gradient && frame_Tcd && keyframe_Twc && pose_dev);
  PoseArgs A;
  A.frame_Tcd = *frame_Tcd;
  A.key_Twc = *keyframe_Twc;
  A.pose = pose_dev;
  A.state = state_dev;
  A.update_out = update_dev;
  A.mirror = Mirror{nullptr, 0, nullptr};
  hipLaunchKernelGGL(color_solve_kernel, dim3(1), dim3(64), 0, vk_s(stream), hessian, gradient,
      translation_enabled, A);
  VK_LAUNCH_CHECK();
  return VK_OK;
}

static int track_impl(const vk_color_view* keyframe, const vk_color_view* frame, const vk_light_terms* terms,
    bool light, const vk_transform* frame_Tcd, const vk_transform* keyframe_Twc, vk_color_pose* pose_dev, int iterations,
    int translation_enabled, float* workspace, float* system, int32_t* state_dev, float* update_dev,
    vk_icp_reduce_fn reduce, void* reduce_user, const vk_track_poll* poll, void* stream)
{
  ColorParams P;
  const vk_transform identity = identity_transform();
  int rc = fill_color(P, keyframe, frame, &identity, true);
  if (rc != VK_OK) return rc;
  if (light && (rc = fill_light(P, terms)) != VK_OK) return rc;
  VK_REQUIRE(frame_Tcd && keyframe_Twc && pose_dev && workspace && system && state_dev && iterations > 0);
  P.Tcm_dev = &pose_dev->Tcm;
  P.state = state_dev;
  float* hessian = system;
  float* gradient = system + 36;
  const int partials = group_count_for(keyframe->width * keyframe->height, P.group_pixels);
  hipStream_t s = vk_s(stream);

  PoseArgs A;
  A.frame_Tcd = *frame_Tcd;
  A.key_Twc = *keyframe_Twc;
  A.pose = pose_dev;
  A.state = state_dev;
  A.update_out = update_dev;
  const bool chunked = polling(poll);
  A.mirror = begin_mirror(poll);
  PoseArgs sums_only = A;
  sums_only.mirror = Mirror{nullptr, 0, nullptr};
  sums_only.pose = nullptr;
  sums_only.state = nullptr;
  sums_only.update_out = nullptr;

  if (!reduce)
  {
    // the whole loop in one launch (color_loop_kernel)
    VK_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 7u) == 0);   // the exchange holds 64-bit words
    ColorLoopParams L;
    L.frame_Tcd = *frame_Tcd;
    L.key_Twc = *keyframe_Twc;
    L.pose = pose_dev;
    L.groups = partials;
    L.fresh_state = 0;
    L.hessian = hessian;
    L.gradient = gradient;
    L.state = state_dev;
    L.update_out = update_dev;
    L.mirror = A.mirror;
    if (light)
      return translation_enabled ? launch_color_loop_of<true, true>(P, L, iterations, workspace, s)
                                 : launch_color_loop_of<true, false>(P, L, iterations, workspace, s);
    return translation_enabled ? launch_color_loop_of<false, true>(P, L, iterations, workspace, s)
                               : launch_color_loop_of<false, false>(P, L, iterations, workspace, s);
  }

  hipLaunchKernelGGL(color_prepare_kernel, dim3(1), dim3(64), 0, s, A);
  VK_LAUNCH_CHECK();

  for (int it = 0; it < iterations; ++it)
  {
    launch_color_partials(P, translation_enabled, partials, workspace, s);

    if (reduce)
    {
      // multi-GPU rig: sum the packed system over ranks before every rank solves it
      hipLaunchKernelGGL(color_final_kernel, dim3(1), dim3(256), 0, s, workspace, partials, translation_enabled,
          hessian, gradient, sums_only);
      VK_LAUNCH_CHECK();
      const int rr = reduce(system, 48, reduce_user, stream);
      if (rr != 0) return rr;
      hipLaunchKernelGGL(color_solve_kernel, dim3(1), dim3(64), 0, s, hessian, gradient, translation_enabled, A);
    }
    else
    {
      hipLaunchKernelGGL(color_final_kernel, dim3(1), dim3(256), 0, s, workspace, partials, translation_enabled,
          hessian, gradient, A);
    }
    VK_LAUNCH_CHECK();
    // stop enqueuing once the loop has converged (tracker.cpp:162), see vk_icp_track: the look is
    // at the state one chunk back, so a chunk of launches is always queued behind it
    if (chunked && (it + 1) % poll->chunk == 0 && it + 1 >= 2 * poll->chunk && it + 1 < iterations &&
        wait_for_steps(A.mirror, it + 1 - poll->chunk, s)) break;
  }
  if (A.mirror.host_pose)
  {
    hipLaunchKernelGGL(color_publish_pose_kernel, dim3(1), dim3(64), 0, s, A.mirror, pose_dev);
    VK_LAUNCH_CHECK();
  }
  return VK_OK;
}

VK_API int vk_color_tracker_track(const vk_color_view* keyframe, const vk_color_view* frame,
    const vk_transform* frame_Tcd, const vk_transform* keyframe_Twc, vk_color_pose* pose_dev, int iterations,
    int translation_enabled, float* workspace, float* system, int32_t* state_dev, float* update_dev,
    vk_icp_reduce_fn reduce, void* reduce_user, const vk_track_poll* poll, void* stream)
{
  return track_impl(keyframe, frame, nullptr, false, frame_Tcd, keyframe_Twc, pose_dev, iterations,
      translation_enabled, workspace, system, state_dev, update_dev, reduce, reduce_user, poll, stream);
}

VK_API int vk_light_tracker_track(const vk_color_view* keyframe, const vk_color_view* frame,
    const vk_light_terms* terms, const vk_transform* keyframe_Twc, vk_color_pose* pose_dev, int iterations,
    int translation_enabled, float* workspace, float* system, int32_t* state_dev, float* update_dev,
    vk_icp_reduce_fn reduce, void* reduce_user, const vk_track_poll* poll, void* stream)
{
  VK_REQUIRE(terms);
  return track_impl(keyframe, frame, terms, true, &terms->frame_Tcd, keyframe_Twc, pose_dev, iterations,
      translation_enabled, workspace, system, state_dev, update_dev, reduce, reduce_user, poll, stream);
}

}  // extern "C"
