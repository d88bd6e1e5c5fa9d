/*
 * oracle_math.h — scalar restatement of the reference's header math with the
 * same operation order (TEST INFRASTRUCTURE, see oracle.h).
 *
 *   Matrix<T,M,N>   include/vulcan/matrix.h   (column-major, :320-333)
 *   Transform       include/vulcan/transform.h
 *   Projection      include/vulcan/projection.h
 *   min/max/clamp   include/vulcan/math.h:9-32
 */
#ifndef ORACLE_MATH_H_
#define ORACLE_MATH_H_

#include <math.h>
#include <float.h>
#include <stdint.h>
#include "../include/vk.h"

typedef struct { float v[3]; } of3;
typedef struct { float v[4]; } of4;

/* math.h:9-20 — note the operand order of the comparisons */
static inline float o_min(float a, float b) { return (b < a) ? b : a; }
static inline float o_max(float a, float b) { return (b > a) ? b : a; }
static inline int   o_mini(int a, int b) { return (b < a) ? b : a; }
static inline int   o_maxi(int a, int b) { return (b > a) ? b : a; }
static inline float o_clamp(float v, float lo, float hi) { return o_min(hi, o_max(lo, v)); }
static inline int   o_clampi(int v, int lo, int hi) { return o_mini(hi, o_maxi(lo, v)); }

/* float -> integer conversions. The reference relies on the GPU's saturating
 * cvt for out-of-range values (UB in C); defined here as saturating, NaN -> 0,
 * which is what cvt.rzi.s32.f32 / cvt.rzi.s16.f32 produce. */
static inline int o_f2i(float x)
{
  if (x != x) return 0;
  if (x >= 2147483648.0f) return INT32_MAX;
  if (x <= -2147483648.0f) return INT32_MIN;
  return (int)x;
}
static inline int16_t o_f2s(float x)
{
  if (x != x) return 0;
  if (x >= 32767.0f) return 32767;
  if (x <= -32768.0f) return -32768;
  return (int16_t)x;
}

static inline of3 o3(float x, float y, float z) { of3 r = {{x, y, z}}; return r; }

/* matrix.h:157-169 Dot: result = 0; result += a[i]*b[i] in index order */
static inline float o_dot3(of3 a, of3 b)
{
  float r = 0;
  r += a.v[0] * b.v[0];
  r += a.v[1] * b.v[1];
  r += a.v[2] * b.v[2];
  return r;
}
static inline float o_sqnorm3(of3 a) { return o_dot3(a, a); }
/* matrix.h:118-123 Norm */
static inline float o_norm3(of3 a) { return sqrtf(o_sqnorm3(a)); }
/* matrix.h:131-154 Normalize/Normalized: multiply by 1/Norm */
static inline of3 o_normalized3(of3 a)
{
  const float inv = 1.0f / o_norm3(a);
  return o3(a.v[0] * inv, a.v[1] * inv, a.v[2] * inv);
}
/* matrix.h:171-181 Cross */
static inline of3 o_cross3(of3 a, of3 b)
{
  return o3((a.v[1] * b.v[2]) - (a.v[2] * b.v[1]),
            (a.v[2] * b.v[0]) - (a.v[0] * b.v[2]),
            (a.v[0] * b.v[1]) - (a.v[1] * b.v[0]));
}
static inline of3 o_add3(of3 a, of3 b) { return o3(a.v[0] + b.v[0], a.v[1] + b.v[1], a.v[2] + b.v[2]); }
static inline of3 o_sub3(of3 a, of3 b) { return o3(a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2]); }
/* matrix.h:257-277 operator*(scalar): every element *= s */
static inline of3 o_scale3(of3 a, float s) { return o3(a.v[0] * s, a.v[1] * s, a.v[2] * s); }
/* matrix.h:279-295 operator/(scalar): multiply by inv = 1.0f / s */
static inline of3 o_div3(of3 a, float s) { const float inv = 1.0f / s; return o_scale3(a, inv); }

/* transform.h:43-60 Transform::operator*(Vector4f), M(r,c) = m[c*4+r] */
static inline of4 o_xform(const float* m, float x, float y, float z, float w)
{
  of4 r;
  r.v[0] = m[0] * x + m[4] * y + m[8]  * z + m[12] * w;
  r.v[1] = m[1] * x + m[5] * y + m[9]  * z + m[13] * w;
  r.v[2] = m[2] * x + m[6] * y + m[10] * z + m[14] * w;
  r.v[3] = w;
  return r;
}
static inline of3 o_xform_point(const float* m, of3 p)
{
  const of4 r = o_xform(m, p.v[0], p.v[1], p.v[2], 1.0f);
  return o3(r.v[0], r.v[1], r.v[2]);
}
static inline of3 o_xform_dir(const float* m, of3 p)
{
  const of4 r = o_xform(m, p.v[0], p.v[1], p.v[2], 0.0f);
  return o3(r.v[0], r.v[1], r.v[2]);
}

/* matrix.h:297-318 4x4 product, result(m,p) = 0; += A(m,n)*B(n,p) over n */
static inline void o_matmul4(const float* A, const float* B, float* C)
{
  for (int p = 0; p < 4; ++p)
    for (int m = 0; m < 4; ++m)
    {
      float r = 0;
      for (int n = 0; n < 4; ++n) r += A[n * 4 + m] * B[p * 4 + n];
      C[p * 4 + m] = r;
    }
}
/* transform.h:62-66 Transform::operator*(Transform) */
static inline vk_transform o_transform_mul(const vk_transform* a, const vk_transform* b)
{
  vk_transform r;
  o_matmul4(a->m, b->m, r.m);
  o_matmul4(b->inv, a->inv, r.inv);
  return r;
}
/* transform.h:68-72 Inverse: swap */
static inline vk_transform o_transform_inverse(const vk_transform* a)
{
  vk_transform r;
  for (int i = 0; i < 16; ++i) { r.m[i] = a->inv[i]; r.inv[i] = a->m[i]; }
  return r;
}

/* projection.h:63-70 Project */
static inline void o_project(const vk_projection* k, of3 X, float* u, float* v)
{
  const float inv_w = 1.0f / X.v[2];
  *u = inv_w * k->fx * X.v[0] + k->cx;
  *v = inv_w * k->fy * X.v[1] + k->cy;
}
/* projection.h:78-88 Unproject(uv) */
static inline of3 o_unproject(const vk_projection* k, float u, float v)
{
  const float ifx = 1.0f / k->fx;
  const float ify = 1.0f / k->fy;
  return o3(ifx * u - k->cx * ifx, ify * v - k->cy * ify, 1.0f);
}
/* projection.h:96-100 Unproject(uv, d) = d * Unproject(uv) */
static inline of3 o_unproject_d(const vk_projection* k, float u, float v, float d)
{
  return o_scale3(o_unproject(k, u, v), d);
}

/* the hash used in volume.cu:168-180, tracer.cu:151-155,344-363 */
static inline uint32_t o_hash(int bx, int by, int bz, uint32_t K)
{
  const uint32_t P1 = 73856093u, P2 = 19349669u, P3 = 83492791u;
  return (((uint32_t)bx * P1) ^ ((uint32_t)by * P2) ^ ((uint32_t)bz * P3)) % K;
}

static inline int o_block_eq(const vk_block* b, int bx, int by, int bz)
{
  /* Block(bx,by,bz) truncates int -> short (block.h:26) */
  return b->origin[0] == (int16_t)bx && b->origin[1] == (int16_t)by &&
         b->origin[2] == (int16_t)bz;
}

/* shared by the trackers (defined in oracle_icp.c) */
void orc_ldlt_solve(int n, const float* A, const float* b, float* x);
void orc_solve_step(const float* hessian_packed, const float* gradient, int translation_enabled, float* update);
vk_transform orc_rigid_from(const float* M);

#endif
