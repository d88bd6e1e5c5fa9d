/*
 * oracle_icp.c — CPU restatement of src/depth_tracker.cu (Evaluate and the
 * three kernels built on it), src/tracker.cpp:124-163 (ComputeUpdate),
 * src/depth_tracker.cpp:22-86 (ApplyUpdate) and src/image.cu:101-165
 * (Downsample). TEST INFRASTRUCTURE, see oracle.h.
 */
#include <float.h>
#include <string.h>
#include "oracle.h"
#include "oracle_math.h"

/* ref: depth_tracker.cu:18-94 Evaluate<translation_enabled> */
static void evaluate(int translation_enabled, int frame_x, int frame_y,
    const vk_transform* Twm, const vk_transform* Twc, const vk_icp_view* key,
    const vk_icp_view* frm, float* residual, float* jacobian)
{
  if (residual) *residual = 0;
  if (jacobian) for (int i = 0; i < 6; ++i) jacobian[i] = 0;

  if (!(frame_x < frm->width && frame_y < frm->height)) return;

  const int frame_index = frame_y * frm->width + frame_x;
  const float frame_depth = frm->depths[frame_index];
  if (!(frame_depth > 0)) return;

  const float frame_u = frame_x + 0.5f;
  const float frame_v = frame_y + 0.5f;
  const of3 Xcp = o_unproject_d(&frm->projection, frame_u, frame_v, frame_depth);
  const of3 Xwp = o_xform_point(Twc->m, Xcp);
  const of3 Xmp = o_xform_point(Twm->inv, Xwp);
  float ku, kv;
  o_project(&key->projection, Xmp, &ku, &kv);

  if (!(ku >= 0 && ku < key->width && kv >= 0 && kv < key->height)) return;

  const int keyframe_x = (int)ku;
  const int keyframe_y = (int)kv;
  const int keyframe_index = keyframe_y * key->width + keyframe_x;
  const float keyframe_depth = key->depths[keyframe_index];
  if (!(keyframe_depth > 0)) return;

  of3 frame_normal = o3(frm->normals[3 * frame_index + 0],
      frm->normals[3 * frame_index + 1], frm->normals[3 * frame_index + 2]);
  frame_normal = o_xform_dir(Twc->m, frame_normal);

  of3 keyframe_normal = o3(key->normals[3 * keyframe_index + 0],
      key->normals[3 * keyframe_index + 1], key->normals[3 * keyframe_index + 2]);
  keyframe_normal = o_xform_dir(Twm->m, keyframe_normal);

  if (!(o_sqnorm3(keyframe_normal) > 0.0f && o_dot3(frame_normal, keyframe_normal) > 0.5f)) return;

  const float fu = floorf(ku) + 0.5f;
  const float fv = floorf(kv) + 0.5f;
  const of3 Ymp = o_unproject_d(&key->projection, fu, fv, keyframe_depth);
  const of3 Ywp = o_xform_point(Twm->m, Ymp);
  const of3 delta = o_sub3(Xwp, Ywp);

  if (!(o_sqnorm3(delta) < 0.05f)) return;

  if (residual) *residual = o_dot3(delta, keyframe_normal);

  if (jacobian)
  {
    const float* n = keyframe_normal.v;
    const float* X = Xwp.v;
    jacobian[0] = n[2] * X[1] - n[1] * X[2];
    jacobian[1] = n[0] * X[2] - n[2] * X[0];
    jacobian[2] = n[1] * X[0] - n[0] * X[1];

    if (translation_enabled)
    {
      jacobian[3] = n[0];
      jacobian[4] = n[1];
      jacobian[5] = n[2];
    }
  }
}

/* ref: depth_tracker.cu:97-118 */
void orc_icp_compute_residuals(const vk_icp_view* keyframe, const vk_transform* Twm,
    const vk_icp_view* frame, const vk_transform* Twc, float* residuals)
{
#pragma omp parallel for num_threads(orc_get_threads())
  for (int y = 0; y < frame->height; ++y)
    for (int x = 0; x < frame->width; ++x)
    {
      float r;
      evaluate(0, x, y, Twm, Twc, keyframe, frame, &r, NULL);
      residuals[y * frame->width + x] = r;
    }
}

/* ref: depth_tracker.cu:120-141 */
void orc_icp_compute_jacobian(const vk_icp_view* keyframe, const vk_transform* Twm,
    const vk_icp_view* frame, const vk_transform* Twc, int translation_enabled,
    float* jacobian)
{
#pragma omp parallel for num_threads(orc_get_threads())
  for (int y = 0; y < frame->height; ++y)
    for (int x = 0; x < frame->width; ++x)
      evaluate(translation_enabled, x, y, Twm, Twc, keyframe, frame, NULL,
          &jacobian[6 * (y * frame->width + x)]);
}

/* ref: depth_tracker.cu:144-268 ComputeSystemKernel. Per-pixel products are
 * float32 as in the kernel (:163-165, :208-214); the sums run in double. */
void orc_icp_compute_system(const vk_icp_view* keyframe, const vk_transform* Twm,
    const vk_icp_view* frame, const vk_transform* Twc, int translation_enabled,
    double* hessian, double* gradient)
{
  const int parameter_count = translation_enabled ? 6 : 3;
  for (int i = 0; i < 21; ++i) hessian[i] = 0;
  for (int i = 0; i < 6; ++i) gradient[i] = 0;

  for (int y = 0; y < frame->height; ++y)
    for (int x = 0; x < frame->width; ++x)
    {
      float r, J[6];
      evaluate(translation_enabled, x, y, Twm, Twc, keyframe, frame, &r, J);

      for (int i = 0; i < parameter_count; ++i) gradient[i] += (double)(J[i] * r);

      int counter = 0;
      for (int rr = 0; rr < parameter_count; ++rr)
        for (int c = 0; c <= rr; ++c, ++counter)
          hessian[counter] += (double)(J[rr] * J[c]);
    }
}

/* LDL^T without pivoting, float32. The reference calls Eigen::LDLT (pivoted,
 * tracker.cpp:127,153-159; Eigen is not vendored and its version is unpinned,
 * so the solve is "parity unpinned": agreement is to rounding, not bit-exact).
 * Zero pivots follow Eigen's published behaviour (LDLT.h: a column under an
 * invalid pivot is not divided; solve zeroes y[i] where |D[i]| <= FLT_MIN), so an
 * empty system solves to x = 0 rather than NaN. */
void orc_ldlt_solve(int n, const float* A /* n*n row-major, symmetric */,
    const float* b, float* x)
{
  float L[36], D[6], y[6];
  memset(L, 0, sizeof(L));

  for (int j = 0; j < n; ++j)
  {
    float d = A[j * n + j];
    for (int k = 0; k < j; ++k) d -= L[j * n + k] * L[j * n + k] * D[k];
    D[j] = d;
    L[j * n + j] = 1.0f;

    for (int i = j + 1; i < n; ++i)
    {
      float s = A[i * n + j];
      for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k] * D[k];
      L[i * n + j] = (fabsf(d) > 0.0f) ? s / d : 0.0f;
    }
  }

  for (int i = 0; i < n; ++i)
  {
    float s = b[i];
    for (int k = 0; k < i; ++k) s -= L[i * n + k] * y[k];
    y[i] = s;
  }

  for (int i = 0; i < n; ++i) y[i] = (fabsf(D[i]) > FLT_MIN) ? y[i] / D[i] : 0.0f;

  for (int i = n - 1; i >= 0; --i)
  {
    float s = y[i];
    for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * x[k];
    x[i] = s;
  }
}

/* The same solve the way Eigen does it — a second statement of tracker.cpp:127,153-159, kept
 * beside the unpivoted one to bound what pivoting can change (tests/test_oracle_icp.py).
 * Eigen::LDLT is a PIVOTED factorisation P A P^T = L D L^T: at step k the largest remaining
 * |diagonal| entry is brought to position k by a symmetric row / column exchange before the
 * column is eliminated (published algorithm: Eigen 3.3 Cholesky/LDLT.h, ldlt_inplace<Lower>::
 * unblocked; solve: LDLT::_solve_impl — x = P^T L^-T D^-1 L^-1 P b with components under a
 * pivot of magnitude <= 1 / FLT_MAX set to 0). Eigen is not vendored and its version is not
 * pinned (CMakeLists.txt:18), and its vectorised inner products do not promise an order of
 * additions: this is the algorithm in plain loops, float32 — PARITY UNPINNED to Eigen's bits,
 * like orc_ldlt_solve. Test infrastructure only. */
void orc_ldlt_solve_pivoted(int n, const float* A /* n*n row-major, symmetric */,
    const float* b, float* x)
{
  float M[36], temp[6], y[6];
  int transpositions[6];
  memcpy(M, A, sizeof(float) * (size_t)(n * n));   /* only the lower triangle M[i*n+j], i >= j, is used */

  for (int k = 0; k < n; ++k)
  {
    int biggest = k;
    float big = fabsf(M[k * n + k]);
    for (int i = k + 1; i < n; ++i)
      if (fabsf(M[i * n + i]) > big) { big = fabsf(M[i * n + i]); biggest = i; }
    transpositions[k] = biggest;

    if (biggest != k)
    {
      float t;
      for (int j = 0; j < k; ++j) { t = M[k * n + j]; M[k * n + j] = M[biggest * n + j]; M[biggest * n + j] = t; }
      for (int i = biggest + 1; i < n; ++i) { t = M[i * n + k]; M[i * n + k] = M[i * n + biggest]; M[i * n + biggest] = t; }
      t = M[k * n + k]; M[k * n + k] = M[biggest * n + biggest]; M[biggest * n + biggest] = t;
      for (int i = k + 1; i < biggest; ++i) { t = M[i * n + k]; M[i * n + k] = M[biggest * n + i]; M[biggest * n + i] = t; }
    }

    if (k > 0)
    {
      float acc = 0.0f;
      for (int j = 0; j < k; ++j) temp[j] = M[j * n + j] * M[k * n + j];
      for (int j = 0; j < k; ++j) acc += M[k * n + j] * temp[j];
      M[k * n + k] -= acc;
      for (int i = k + 1; i < n; ++i)
      {
        acc = 0.0f;
        for (int j = 0; j < k; ++j) acc += M[i * n + j] * temp[j];
        M[i * n + k] -= acc;
      }
    }

    const float akk = M[k * n + k];
    const int valid = fabsf(akk) > 0.0f;
    if (k == 0 && !valid)
    {
      /* the matrix is zero: every transposition is the identity, D = 0 */
      for (int j = 0; j < n; ++j) transpositions[j] = j;
      break;
    }
    if (valid)
      for (int i = k + 1; i < n; ++i) M[i * n + k] /= akk;
  }

  for (int i = 0; i < n; ++i) y[i] = b[i];
  for (int k = 0; k < n; ++k) { const float t = y[k]; y[k] = y[transpositions[k]]; y[transpositions[k]] = t; }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < i; ++j) y[i] -= M[i * n + j] * y[j];
  for (int i = 0; i < n; ++i) y[i] = (fabsf(M[i * n + i]) > 1.0f / FLT_MAX) ? y[i] / M[i * n + i] : 0.0f;
  for (int i = n - 1; i >= 0; --i)
    for (int j = i + 1; j < n; ++j) y[i] -= M[j * n + i] * y[j];
  for (int k = n - 1; k >= 0; --k) { const float t = y[k]; y[k] = y[transpositions[k]]; y[transpositions[k]] = t; }
  for (int i = 0; i < n; ++i) x[i] = y[i];
}

/* ref: tracker.cpp:142-159: unpack the packed lower triangle (mirrored), solve,
 * update = -x; entries beyond the parameter count stay 0 */
void orc_solve_step(const float* hessian_packed, const float* gradient, int translation_enabled,
    float* update)
{
  const int n = translation_enabled ? 6 : 3;
  float H[36], x[6];

  int index = 0;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j)
    {
      H[i * n + j] = hessian_packed[index];
      H[j * n + i] = hessian_packed[index];
      ++index;
    }

  orc_ldlt_solve(n, H, gradient, x);
  for (int i = 0; i < 6; ++i) update[i] = 0;
  for (int i = 0; i < n; ++i) update[i] = -x[i];   /* tracker.cpp:159 */
}

/* depth_tracker.cpp:57-84 / color_tracker.cpp:67-95: re-orthonormalised
 * Translate(t) * Rotate(R) of the 4x4 M (transform.h:62-66,74-99,146-159) */
vk_transform orc_rigid_from(const float* M)
{
  of3 x_axis = o3(M[0], M[1], M[2]);
  of3 y_axis = o3(M[4], M[5], M[6]);
  of3 z_axis;
  x_axis = o_normalized3(x_axis);
  y_axis = o_normalized3(y_axis);
  z_axis = o_cross3(x_axis, y_axis);
  y_axis = o_cross3(z_axis, x_axis);

  vk_transform T, R;
  memset(&T, 0, sizeof(T));
  memset(&R, 0, sizeof(R));
  for (int i = 0; i < 4; ++i) { T.m[5 * i] = 1.0f; T.inv[5 * i] = 1.0f; R.m[5 * i] = 1.0f; }
  T.m[12] = M[12];    T.m[13] = M[13];    T.m[14] = M[14];
  T.inv[12] = -M[12]; T.inv[13] = -M[13]; T.inv[14] = -M[14];
  for (int r = 0; r < 3; ++r)
  {
    R.m[0 * 4 + r] = x_axis.v[r];
    R.m[1 * 4 + r] = y_axis.v[r];
    R.m[2 * 4 + r] = z_axis.v[r];
  }
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c)
      R.inv[c * 4 + r] = R.m[r * 4 + c];  /* matrix.Transpose() */

  return o_transform_mul(&T, &R);
}

/* ref: tracker.cpp:124-163 ComputeUpdate + depth_tracker.cpp:22-86 ApplyUpdate */
float orc_icp_solve_update(const float* hessian_packed, const float* gradient,
    int translation_enabled, vk_transform* Twc, float* update_out)
{
  const int n = translation_enabled ? 6 : 3;
  float update[6];
  orc_solve_step(hessian_packed, gradient, translation_enabled, update);

  /* depth_tracker.cpp:33-53; note Tinc(1,2) = +update[0] */
  float Tinc[16];
  Tinc[0 + 4 * 0] = 1.0f;       Tinc[0 + 4 * 1] = -update[2]; Tinc[0 + 4 * 2] = +update[1]; Tinc[0 + 4 * 3] = +update[3];
  Tinc[1 + 4 * 0] = +update[2]; Tinc[1 + 4 * 1] = 1.0f;       Tinc[1 + 4 * 2] = +update[0]; Tinc[1 + 4 * 3] = +update[4];
  Tinc[2 + 4 * 0] = -update[1]; Tinc[2 + 4 * 1] = +update[0]; Tinc[2 + 4 * 2] = 1.0f;       Tinc[2 + 4 * 3] = +update[5];
  Tinc[3 + 4 * 0] = 0.0f;       Tinc[3 + 4 * 1] = 0.0f;       Tinc[3 + 4 * 2] = 0.0f;       Tinc[3 + 4 * 3] = 1.0f;

  float M[16];
  o_matmul4(Tinc, Twc->m, M);  /* :55 */
  *Twc = orc_rigid_from(M);    /* :57-84 */

  float sq = 0;
  for (int i = 0; i < n; ++i) sq += update[i] * update[i];
  if (update_out) for (int i = 0; i < 6; ++i) update_out[i] = update[i];
  return sqrtf(sq);
}

/* ref: image.cu:101-131 DownsampleKernel<nearest>(float) */
void orc_image_downsample(int src_w, int src_h, const float* src, float* dst, int nearest)
{
  const int dst_w = src_w / 2, dst_h = src_h / 2;

  for (int dst_y = 0; dst_y < dst_h; ++dst_y)
    for (int dst_x = 0; dst_x < dst_w; ++dst_x)
    {
      float sample = 0;
      const int src_x = 2 * dst_x;
      const int src_y = 2 * dst_y;

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
}

/* ref: image.cu:133-165 DownsampleKernel<nearest>(Vector3f) */
void orc_color_image_downsample(int src_w, int src_h, const float* src, float* dst, int nearest)
{
  const int dst_w = src_w / 2, dst_h = src_h / 2;

  for (int dst_y = 0; dst_y < dst_h; ++dst_y)
    for (int dst_x = 0; dst_x < dst_w; ++dst_x)
      for (int c = 0; c < 3; ++c)
      {
        float sample = 0;
        const int src_x = 2 * dst_x;
        const int src_y = 2 * dst_y;

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
