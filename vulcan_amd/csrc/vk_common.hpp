// vk_common.hpp — shared host/device helpers for libvk_hip.so (gfx950 only).
//
// Device arithmetic follows the reference's operation order exactly and the
// library is built with -ffp-contract=off, so results do not depend on where
// the compiler would have fused a multiply-add.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>

#include "../../include/vk.h"

// A runtime call that fails is reported to OUR caller as its code — and taken out of the runtime's per-thread "last error",
// which is sticky: left there it would be picked up by the next VK_LAUNCH_CHECK of an unrelated, good launch, or by the
// caller's own framework (round 6: vk_event_elapsed_ms on a pair that was never recorded made torch's next call raise).
#define VK_CHECK(expr)                                  \
  do {                                                  \
    const hipError_t vk_e__ = (expr);                   \
    if (vk_e__ != hipSuccess)                           \
    {                                                   \
      (void)hipGetLastError();                          \
      return (int)vk_e__;                               \
    }                                                   \
  } while (0)

#define VK_REQUIRE(cond)                                \
  do {                                                  \
    if (!(cond)) return VK_ERR_ARGUMENT;                \
  } while (0)

// Launch errors are picked up with hipGetLastError (no device sync). The debug build
// (-DVK_DEBUG_SYNC: libvk_hip_debug.so, `make debug`) also waits for the device after every
// launch and reports what the kernel itself raised, at the launch that raised it — the
// reference's CUDA_LAUNCH does this whenever NDEBUG is not defined (device.h:48-52).
#ifdef VK_DEBUG_SYNC
#define VK_LAUNCH_CHECK()                               \
  do {                                                  \
    VK_CHECK(hipGetLastError());                        \
    VK_CHECK(hipDeviceSynchronize());                   \
    VK_CHECK(hipGetLastError());                        \
  } while (0)
#else
#define VK_LAUNCH_CHECK() VK_CHECK(hipGetLastError())
#endif

static inline hipStream_t vk_s(void* stream) { return reinterpret_cast<hipStream_t>(stream); }

// a field of vk_test_hooks (vk.h), by position; defined in vk_runtime.hip
enum { VK_HOOK_POSTED_CAPACITY = 0, VK_HOOK_RETRY_CAPACITY, VK_HOOK_SET_VIEW_UNFUSED, VK_HOOK_FORCE_LOOP_ABORT,
       VK_HOOK_LOOP_GRID_CAP, VK_HOOK_LOOP_COOPERATIVE, VK_HOOK_FORCE_NORMALS_EXPIRY };
int vk_hook(int field);
int vk_hook_take(int field);   // the value, and 0 from then on

namespace vk
{

constexpr int kWave = 64;       // CDNA4 wavefront
constexpr int kCUs = 256;       // MI355X compute units
constexpr int kXCDs = 8;        // accelerator dies; workgroup g of a launch runs on XCD g % kXCDs, each with its own L2

// Behind the counters, the retry sets and the posted list (vk.h): the banded visible lists — VK_BANDS counts, then
// VK_BANDS lists of VK_BAND_SLOTS entry indices (written by vk_volume.hip's visibility pass, read by vk_integrate.hip)
__host__ __device__ inline int32_t* band_counts(int32_t* counters)
{
  return counters + VK_CTR_PUBLIC + 2 * 2 * VK_RETRY_SLOTS + 2 * VK_RETRY_KEYS + 2 * VK_POSTED_SLOTS;
}
__host__ __device__ inline const int32_t* band_counts(const int32_t* counters)
{
  return counters + VK_CTR_PUBLIC + 2 * 2 * VK_RETRY_SLOTS + 2 * VK_RETRY_KEYS + 2 * VK_POSTED_SLOTS;
}
__host__ __device__ inline int32_t* band_lists(int32_t* counters) { return band_counts(counters) + VK_BANDS; }
__host__ __device__ inline const int32_t* band_lists(const int32_t* counters) { return band_counts(counters) + VK_BANDS; }
static_assert(VK_BANDS == kXCDs, "one band of image rows per XCD");

// 8- and 12-byte loads of packed floats at 4-byte aligned addresses (gfx950 global
// loads only need dword alignment): a 12-byte Vector3f is one dwordx3 load
typedef float vf2 __attribute__((ext_vector_type(2), aligned(4)));
typedef float vf3 __attribute__((ext_vector_type(3), aligned(4)));

// ---- float3 in the reference's operation order (matrix.h) ------------------

struct f3 { float x, y, z; };

__device__ __forceinline__ f3 make3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 add3(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 sub3(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 scale3(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
// matrix.h:157-169 Dot: starts from 0, adds in index order
__device__ __forceinline__ float dot3(f3 a, f3 b)
{
  float r = 0.0f;
  r += a.x * b.x;
  r += a.y * b.y;
  r += a.z * b.z;
  return r;
}
__device__ __forceinline__ float sqnorm3(f3 a) { return dot3(a, a); }
// matrix.h:131-154 Normalize(d): multiply by 1/sqrt(dot)
__device__ __forceinline__ f3 normalized3(f3 a)
{
  const float inv = 1.0f / sqrtf(sqnorm3(a));
  return scale3(a, inv);
}
// matrix.h:279-295 operator/ multiplies by the reciprocal
__device__ __forceinline__ f3 div3(f3 a, float s) { return scale3(a, 1.0f / s); }
__device__ __forceinline__ f3 cross3(f3 a, f3 b)
{
  return f3{(a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x)};
}

// math.h:9-32 (operand order of the comparisons matters for NaN)
__device__ __forceinline__ float vmin(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float vmax(float a, float b) { return (b > a) ? b : a; }
__device__ __forceinline__ int vmini(int a, int b) { return (b < a) ? b : a; }
__device__ __forceinline__ int vmaxi(int a, int b) { return (b > a) ? b : a; }
__device__ __forceinline__ float vclamp(float v, float lo, float hi) { return vmin(hi, vmax(lo, v)); }
__device__ __forceinline__ int vclampi(int v, int lo, int hi) { return vmini(hi, vmaxi(lo, v)); }

// float -> int / short: saturating, NaN -> 0 (what the reference gets from the
// GPU's cvt instruction where C leaves the conversion undefined). v_cvt_i32_f32
// has exactly these semantics; inline asm keeps the out-of-range case out of
// the optimiser's hands.
__device__ __forceinline__ int f2i(float x)
{
  int r;
  asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}
__device__ __forceinline__ int f2s(float x)
{
  if (x != x) return 0;
  if (x >= 32767.0f) return 32767;
  if (x <= -32768.0f) return -32768;
  return (int)x;
}

// Rigid transform rows as kernel arguments: 12 floats, row-major 3x4.
struct Rt
{
  float r[12];
};

// column-major 4x4 -> row-major 3x4
static inline Rt make_rt(const float* m)
{
  Rt t;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) t.r[r * 4 + c] = m[c * 4 + r];
  return t;
}

// transform.h:43-60 Transform::operator*(Vector4f) with w = 1 / w = 0
__device__ __forceinline__ f3 xform_point(const Rt& t, f3 p)
{
  return f3{t.r[0] * p.x + t.r[1] * p.y + t.r[2] * p.z + t.r[3] * 1.0f,
            t.r[4] * p.x + t.r[5] * p.y + t.r[6] * p.z + t.r[7] * 1.0f,
            t.r[8] * p.x + t.r[9] * p.y + t.r[10] * p.z + t.r[11] * 1.0f};
}
__device__ __forceinline__ f3 xform_dir(const Rt& t, f3 p)
{
  return f3{t.r[0] * p.x + t.r[1] * p.y + t.r[2] * p.z + t.r[3] * 0.0f,
            t.r[4] * p.x + t.r[5] * p.y + t.r[6] * p.z + t.r[7] * 0.0f,
            t.r[8] * p.x + t.r[9] * p.y + t.r[10] * p.z + t.r[11] * 0.0f};
}

// projection.h:63-70
__device__ __forceinline__ void project(const vk_projection& k, f3 X, float& u, float& v)
{
  const float inv_w = 1.0f / X.z;
  u = inv_w * k.fx * X.x + k.cx;
  v = inv_w * k.fy * X.y + k.cy;
}
// projection.h:78-88
// A projection that carries 1 / fx and 1 / fy (projection.h:79-86 divides per call): the two quotients are launch constants,
// but a GPU lane has no cheaper way to a correctly rounded uniform quotient than the ten-instruction division, twice per
// thread of every kernel that unprojects; the host's float division gives the same bits (make_projection).
struct Projection : vk_projection
{
  float ifx, ify;
};
inline Projection make_projection(const vk_projection& k)
{
  Projection p;
  p.fx = k.fx;  p.fy = k.fy;  p.cx = k.cx;  p.cy = k.cy;
  p.ifx = 1.0f / k.fx;
  p.ify = 1.0f / k.fy;
  return p;
}
__device__ __forceinline__ float inverse_fx(const vk_projection& k) { return 1.0f / k.fx; }
__device__ __forceinline__ float inverse_fy(const vk_projection& k) { return 1.0f / k.fy; }
__device__ __forceinline__ float inverse_fx(const Projection& k) { return k.ifx; }
__device__ __forceinline__ float inverse_fy(const Projection& k) { return k.ify; }

template <typename K>
__device__ __forceinline__ f3 unproject(const K& k, float u, float v)
{
  const float ifx = inverse_fx(k);
  const float ify = inverse_fy(k);
  return f3{ifx * u - k.cx * ifx, ify * v - k.cy * ify, 1.0f};
}
// projection.h:96-100: d * Unproject(uv)
template <typename K>
__device__ __forceinline__ f3 unproject_d(const K& k, float u, float v, float d)
{
  return scale3(unproject(k, u, v), d);
}

// volume.cu:168-180, tracer.cu:151-155
__device__ __forceinline__ uint32_t block_hash(int bx, int by, int bz, uint32_t K)
{
  return (((uint32_t)bx * 73856093u) ^ ((uint32_t)by * 19349669u) ^ ((uint32_t)bz * 83492791u)) % K;
}

// 16-byte hash entry as one vector load
struct Entry
{
  int16_t ox, oy, oz, pad;
  int32_t data;
  int32_t next;
};
static_assert(sizeof(Entry) == 16, "HashEntry layout");

__device__ __forceinline__ Entry load_entry(const vk_hash_entry* entries, uint32_t index)
{
  const int4 raw = reinterpret_cast<const int4*>(entries)[index];
  Entry e;
  e.ox = (int16_t)(raw.x & 0xffff);
  e.oy = (int16_t)((uint32_t)raw.x >> 16);
  e.oz = (int16_t)(raw.y & 0xffff);
  e.pad = (int16_t)((uint32_t)raw.y >> 16);
  e.data = raw.z;
  e.next = raw.w;
  return e;
}

// Block(bx,by,bz) truncates to short (block.h:26) before the comparison
__device__ __forceinline__ bool entry_is(const Entry& e, int bx, int by, int bz)
{
  return e.ox == (int16_t)bx && e.oy == (int16_t)by && e.oz == (int16_t)bz;
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// Lanes of ONE wave exchanging data through LDS need no hardware barrier (a
// wave's LDS operations execute in order), but the compiler must be told that
// other lanes wrote what this lane is about to read: without it the optimiser may
// forward this lane's own earlier store to the load. Emits no instruction.
__device__ __forceinline__ void wave_lds_fence()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


// ---- Frame::ComputeNormals for one pixel (ref: frame.cu:9-122): `depth` at (x, y) and the four taps
// two pixels away (0 = no measurement, or outside the image). Shared by compute_normals_kernel and
// by the request pass of vk_volume_set_view_prepare when it computes the frame's normals on the way.
template <typename K>
__device__ __forceinline__ f3 normal_from_taps(const K& k, int x, int y, float depth,
    float left, float right, float up, float down)
{
  const int pad = 2;
  f3 normal = make3(0, 0, 0);
  if (depth > 0)
  {
    const f3 z0 = unproject_d(k, (x + 0) + 0.5f, (y + 0) + 0.5f, depth);
    const f3 x0 = (left == 0) ? z0 : scale3(unproject(k, (x - pad) + 0.5f, (y + 0) + 0.5f), left);
    const f3 x1 = (right == 0) ? z0 : scale3(unproject(k, (x + pad) + 0.5f, (y + 0) + 0.5f), right);
    const f3 y0 = (up == 0) ? z0 : scale3(unproject(k, (x + 0) + 0.5f, (y - pad) + 0.5f), up);
    const f3 y1 = (down == 0) ? z0 : scale3(unproject(k, (x + 0) + 0.5f, (y + pad) + 0.5f), down);
    const f3 dx = sub3(x0, x1);
    const f3 dy = sub3(y0, y1);
    if (sqnorm3(dx) > 0 && sqnorm3(dy) > 0) normal = normalized3(cross3(dy, dx));
  }
  return normal;
}

// ---- LightIntegrator's per-pixel preparation (ref: light_integrator.cu:16-94 frame mask,
// :215-225 the per-pixel half of the colour kernel) — shared by frame_mask_kernel and by the
// request pass of vk_volume_set_view_prepare. `window`: a depth tile in LDS, zero outside the
// image, `stride` floats per row, in which pixel (x, y) sits at window[(cy - 2) * stride + cx - 2]
// — upstream stages its tile with a -1 halo offset and reads it with a +3 centre, so the
// 7x7 window of pixel (x, y) is [x-1, x+5] x [y-1, y+5] (kept as is) and (cx, cy)
// is the window's CENTRE (x + 2, y + 2) in tile coordinates.
__device__ __forceinline__ float light_window_mask(const float* window, int stride, int cx, int cy, float depth_threshold)
{
  float dmin = +FLT_MAX;
  float dmax = -FLT_MAX;
  for (int i = -3; i <= 3; ++i)
    for (int j = -3; j <= 3; ++j)
    {
      const float depth = window[(cy + i) * stride + (cx + j)];
      dmin = fminf(depth, dmin);
      dmax = fmaxf(depth, dmax);
    }
  return (dmax - dmin <= depth_threshold) ? 1.0f : 0.0f;
}

__device__ __forceinline__ bool light_color_usable(float c0, float c1, float c2)
{
  return !(c0 < 0.02f || c0 > 0.98f || c1 < 0.02f || c1 > 0.98f || c2 < 0.02f || c2 > 0.98f);
}

}  // namespace vk
