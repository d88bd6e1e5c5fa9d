/*
 * oracle_color_tracker.c — CPU restatement of the reference's photometric
 * tracker: src/color_tracker.cu (Sample, Evaluate and the three kernels built on
 * it), src/color_tracker.cpp:34-96 (ApplyUpdate) and the two image operators it
 * needs, src/image.cu:10-19 (ConvertKernel) and :21-99 (GetGradientsKernel).
 * TEST INFRASTRUCTURE, see oracle.h. Pinned by tests/test_oracle_color_tracker.py
 * (the cases of tests/color_tracker_test.cu).
 */
#include <math.h>
#include <string.h>
#include "oracle.h"
#include "oracle_math.h"

/* ref: image.cu:10-19 */
void orc_color_image_convert(int total, const float* src, float* dst)
{
  for (int i = 0; i < total; ++i)
    dst[i] = (src[3 * i + 0] + src[3 * i + 1] + src[3 * i + 2]) / 3.0f;
}

static float padded(int width, int height, const float* v, int x, int y)
{
  return (x >= 0 && x < width && y >= 0 && y < height) ? v[y * width + x] : 0.0f;   /* image.cu:43-56 */
}

/* ref: image.cu:21-99. The shared-memory patch of the kernel is the image with a
 * one-pixel border of zeros; the arithmetic below is :78-93 verbatim. */
void orc_image_gradients(int width, int height, const float* src, float* gx_out, float* gy_out)
{
#pragma omp parallel for
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x)
    {
      const float i00 = 0.125f * padded(width, height, src, x - 1, y - 1);
      const float i01 = 0.250f * padded(width, height, src, x + 0, y - 1);
      const float i02 = 0.125f * padded(width, height, src, x + 1, y - 1);
      const float i10 = 0.250f * padded(width, height, src, x - 1, y + 0);
      const float i12 = 0.250f * padded(width, height, src, x + 1, y + 0);
      const float i20 = 0.125f * padded(width, height, src, x - 1, y + 1);
      const float i21 = 0.250f * padded(width, height, src, x + 0, y + 1);
      const float i22 = 0.125f * padded(width, height, src, x + 1, y + 1);
      gx_out[y * width + x] = (i02 + i12 + i22) - (i00 + i10 + i20);
      gy_out[y * width + x] = (i20 + i21 + i22) - (i00 + i01 + i02);
    }
}

/* ref: color_tracker.cu:17-41 */
static float sample(int w, const float* values, float u, float v)
{
  const int x = (int)floorf(u - 0.5f);
  const int y = (int)floorf(v - 0.5f);

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

/* ref: color_tracker.cu:43-138 Evaluate<translation_enabled> */
static void evaluate(int translation_enabled, int keyframe_x, int keyframe_y,
    const vk_transform* Tcm, const vk_color_view* key, const vk_color_view* frm,
    float* residual, float* jacobian)
{
  if (residual) *residual = 0;
  if (jacobian) for (int i = 0; i < 6; ++i) jacobian[i] = 0;

  const int keyframe_index = keyframe_y * key->width + keyframe_x;
  const float keyframe_depth = key->depths[keyframe_index];
  if (!(keyframe_depth > 0)) return;

  const float keyframe_u = keyframe_x + 0.5f;
  const float keyframe_v = keyframe_y + 0.5f;
  const of3 Xmp = o_unproject_d(&key->projection, keyframe_u, keyframe_v, keyframe_depth);
  const of3 Xcp = o_xform_point(Tcm->m, Xmp);
  float fu, fv;
  o_project(&frm->projection, Xcp, &fu, &fv);

  if (!(fu >= 0.5f && fu < frm->width - 0.5f && fv >= 0.5f && fv < frm->height - 0.5f)) return;

  const int frame_x = (int)fu;
  const int frame_y = (int)fv;
  const int frame_index = frame_y * frm->width + frame_x;
  const float frame_depth = frm->depths[frame_index];
  if (!(fabsf(frame_depth - Xcp.v[2]) < 0.1f)) return;

  const of3 frame_normal = o3(frm->normals[3 * frame_index + 0], frm->normals[3 * frame_index + 1],
      frm->normals[3 * frame_index + 2]);
  of3 keyframe_normal = o3(key->normals[3 * keyframe_index + 0], key->normals[3 * keyframe_index + 1],
      key->normals[3 * keyframe_index + 2]);
  keyframe_normal = o_xform_dir(Tcm->m, keyframe_normal);

  if (!(o_sqnorm3(keyframe_normal) > 0.5f && o_dot3(frame_normal, keyframe_normal) > 0.5f)) return;

  const float Im = key->intensities[keyframe_index];
  const float Ic = sample(frm->width, frm->intensities, fu, fv);
  if (residual) *residual = Ic - Im;

  if (jacobian)
  {
    const float px = Xcp.v[0];
    const float py = Xcp.v[1];
    const float pz = Xcp.v[2];
    const float inv_pz = 1.0f / pz;

    const float cu = fu;
    const float cv = fv;

    const float fx = frm->projection.fx;
    const float fy = frm->projection.fy;
    const float cx = frm->projection.cx;
    const float cy = frm->projection.cy;

    const float gx = sample(frm->width, frm->gradient_x, fu, fv);
    const float gy = sample(frm->width, frm->gradient_y, fu, fv);

    jacobian[0] = gy * ((py * cy - pz * fy) * inv_pz - py * cv * inv_pz) -
                  gx * (py * cu * inv_pz - cx * py * inv_pz);

    jacobian[1] = gy * (px * cv * inv_pz - cy * px * inv_pz) -
                  gx * ((px * cx - pz * fx) * inv_pz - px * cu * inv_pz);

    jacobian[2] = (gy * fy * px - gx * fx * py) * inv_pz;

    if (translation_enabled)
    {
      jacobian[3] = gx * fx * inv_pz;
      jacobian[4] = gy * fy * inv_pz;
      jacobian[5] = (gx * (cx - cu) + gy * (cy - cv)) * inv_pz;
    }
  }
}

/* ref: color_tracker.cu:140-163 */
void orc_color_tracker_compute_residuals(const vk_color_view* keyframe, const vk_color_view* frame,
    const vk_transform* Tcm, float* residuals)
{
#pragma omp parallel for
  for (int y = 0; y < keyframe->height; ++y)
    for (int x = 0; x < keyframe->width; ++x)
      evaluate(0, x, y, Tcm, keyframe, frame, &residuals[y * keyframe->width + x], NULL);
}

/* ref: color_tracker.cu:165-204 */
void orc_color_tracker_compute_jacobian(const vk_color_view* keyframe, const vk_color_view* frame,
    const vk_transform* Tcm, int translation_enabled, float* jacobian)
{
#pragma omp parallel for
  for (int y = 0; y < keyframe->height; ++y)
    for (int x = 0; x < keyframe->width; ++x)
      evaluate(translation_enabled, x, y, Tcm, keyframe, frame, NULL,
          &jacobian[6 * (y * keyframe->width + x)]);
}

/* ref: color_tracker.cu:206-343. Per-pixel products are float32 as in the kernel
 * (:240-242, :283-289); the sums run in double. */
void orc_color_tracker_compute_system(const vk_color_view* keyframe, const vk_color_view* frame,
    const vk_transform* Tcm, int translation_enabled, double* hessian, double* gradient)
{
  const int parameter_count = translation_enabled ? 6 : 3;
  for (int i = 0; i < 21; ++i) hessian[i] = 0;
  for (int i = 0; i < 6; ++i) gradient[i] = 0;

  for (int y = 0; y < keyframe->height; ++y)
    for (int x = 0; x < keyframe->width; ++x)
    {
      float r, J[6];
      evaluate(translation_enabled, x, y, Tcm, keyframe, frame, &r, J);

      for (int i = 0; i < parameter_count; ++i) gradient[i] += (double)(J[i] * r);

      int counter = 0;
      for (int rr = 0; rr < parameter_count; ++rr)
        for (int c = 0; c <= rr; ++c, ++counter)
          hessian[counter] += (double)(J[rr] * J[c]);
    }
}

/* color_tracker.cu:312-320: Tcm = (frame_Tcd * frame_Twd^-1) * keyframe_Tcw^-1,
 * products in the reference's order (transform.h:146-159) */
void orc_color_tracker_tcm(const vk_transform* frame_Tcd, const vk_transform* keyframe_Twc,
    vk_color_pose* pose)
{
  const vk_transform frame_Tdw = o_transform_inverse(&pose->depth_to_world);
  const vk_transform frame_Tcw = o_transform_mul(frame_Tcd, &frame_Tdw);
  pose->Tcm = o_transform_mul(&frame_Tcw, keyframe_Twc);
}

/* ref: tracker.cpp:124-163 ComputeUpdate + color_tracker.cpp:34-96 ApplyUpdate */
float orc_color_tracker_solve_update(const float* hessian_packed, const float* gradient,
    int translation_enabled, const vk_transform* frame_Tcd, const vk_transform* keyframe_Twc,
    vk_color_pose* pose, float* update_out)
{
  const int n = translation_enabled ? 6 : 3;
  float update[6];
  orc_solve_step(hessian_packed, gradient, translation_enabled, update);

  float Tinc[16];   /* color_tracker.cpp:45-65: a proper skew matrix, unlike DepthTracker's */
  Tinc[0 + 4 * 0] = 1.0f;       Tinc[0 + 4 * 1] = -update[2]; Tinc[0 + 4 * 2] = +update[1]; Tinc[0 + 4 * 3] = +update[3];
  Tinc[1 + 4 * 0] = +update[2]; Tinc[1 + 4 * 1] = 1.0f;       Tinc[1 + 4 * 2] = -update[0]; Tinc[1 + 4 * 3] = +update[4];
  Tinc[2 + 4 * 0] = -update[1]; Tinc[2 + 4 * 1] = +update[0]; Tinc[2 + 4 * 2] = 1.0f;       Tinc[2 + 4 * 3] = +update[5];
  Tinc[3 + 4 * 0] = 0.0f;       Tinc[3 + 4 * 1] = 0.0f;       Tinc[3 + 4 * 2] = 0.0f;       Tinc[3 + 4 * 3] = 1.0f;

  float M[16];
  o_matmul4(Tinc, pose->depth_to_world.inv, M);            /* :67 */
  const vk_transform world_to_depth = orc_rigid_from(M);   /* :69-95 */
  pose->depth_to_world = o_transform_inverse(&world_to_depth);
  orc_color_tracker_tcm(frame_Tcd, keyframe_Twc, pose);

  float sq = 0;
  for (int i = 0; i < n; ++i) sq += update[i] * update[i];
  if (update_out) for (int i = 0; i < 6; ++i) update_out[i] = update[i];
  return sqrtf(sq);
}

/* ------------------------------------------------------------ light tracker ---- */

/* ref: light_tracker.cu:133-330 Evaluate<translation_enabled>. The residual and
 * the point-to-plane fallback (:283-322) follow the reference line by line. The
 * photometric Jacobian does not: the reference holds six machine-generated
 * expressions built from powf and sqrt (:233-242); this is the same derivative in
 * factored form, with r = Ic - aa * S, S = ii * (n . d) / |d|^3, d = light - p:
 *   grad_p r = grad_p Ic - aa * ii * (-n / |d|^3 + 3 (n . d) d / |d|^5)
 *   grad_n r =           - aa * ii * d / |d|^3
 *   J[3..5] = grad_p r,   J[0..2] = p x grad_p r + n x grad_n r
 * (grad_p Ic and p x grad_p Ic are ColorTracker's expressions, to which the
 * reference's image terms reduce). Pinned by light_tracker_test.cu:454-528. */
static void evaluate_light(int translation_enabled, int keyframe_x, int keyframe_y,
    const vk_transform* Tcm, const vk_light_terms* terms, const vk_color_view* key,
    const vk_color_view* frm, float* residual, float* jacobian)
{
  if (residual) *residual = 0;
  if (jacobian) for (int i = 0; i < 6; ++i) jacobian[i] = 0;

  const int keyframe_index = keyframe_y * key->width + keyframe_x;
  const float keyframe_depth = key->depths[keyframe_index];
  if (!(keyframe_depth > 0)) return;

  const of3 Xmp = o_unproject_d(&key->projection, keyframe_x + 0.5f, keyframe_y + 0.5f, keyframe_depth);
  const of3 Xcp = o_xform_point(Tcm->m, Xmp);
  float fu, fv;
  o_project(&frm->projection, Xcp, &fu, &fv);

  if (!(fu >= 0.5f && fu < frm->width - 0.5f && fv >= 0.5f && fv < frm->height - 0.5f)) return;

  const int frame_x = (int)fu;
  const int frame_y = (int)fv;
  const int frame_index = frame_y * frm->width + frame_x;
  const float frame_depth = frm->depths[frame_index];
  if (!(fabsf(frame_depth - Xcp.v[2]) < 0.2f)) return;                                   /* :174 */

  const of3 frame_Xdn = o3(frm->normals[3 * frame_index + 0], frm->normals[3 * frame_index + 1],
      frm->normals[3 * frame_index + 2]);
  const of3 frame_normal = o_xform_dir(terms->frame_Tcd.m, frame_Xdn);                    /* :178 */
  of3 n = o3(key->normals[3 * keyframe_index + 0], key->normals[3 * keyframe_index + 1],
      key->normals[3 * keyframe_index + 2]);
  n = o_xform_dir(Tcm->m, n);

  if (!(o_sqnorm3(n) > 0.5f && o_dot3(frame_normal, n) > 0.8f)) return;                   /* :183-184 */

  const float px = Xcp.v[0], py = Xcp.v[1], pz = Xcp.v[2];

  if (terms->frame_mask[frame_index] > 0.5f)
  {
    const float aa = key->intensities[keyframe_index];
    if (!(aa > 0)) return;

    /* light.h:53-60 GetShading */
    const of3 d = o_sub3(o3(terms->light.position[0], terms->light.position[1], terms->light.position[2]), Xcp);
    const float d2 = o_sqnorm3(d);
    const float dn = sqrtf(d2);
    const float inv_dn = 1.0f / dn;
    const of3 direction = o_scale3(d, inv_dn);
    const float cos_theta = o_dot3(n, direction);
    const float ii = terms->light.intensity;
    const float shading = ii * cos_theta / d2;
    const float Im = shading * aa;
    const float Ic = sample(frm->width, frm->intensities, fu, fv);
    if (residual) *residual = Ic - Im;

    if (jacobian)
    {
      const float inv_pz = 1.0f / pz;
      const float cu = fu, cv = fv;
      const float fx = frm->projection.fx, fy = frm->projection.fy;
      const float cx = frm->projection.cx, cy = frm->projection.cy;
      const float gx = sample(frm->width, frm->gradient_x, fu, fv);
      const float gy = sample(frm->width, frm->gradient_y, fu, fv);

      /* image terms (ColorTracker's) */
      float J[6];
      J[0] = gy * ((py * cy - pz * fy) * inv_pz - py * cv * inv_pz) - gx * (py * cu * inv_pz - cx * py * inv_pz);
      J[1] = gy * (px * cv * inv_pz - cy * px * inv_pz) - gx * ((px * cx - pz * fx) * inv_pz - px * cu * inv_pz);
      J[2] = (gy * fy * px - gx * fx * py) * inv_pz;
      J[3] = gx * fx * inv_pz;
      J[4] = gy * fy * inv_pz;
      J[5] = (gx * (cx - cu) + gy * (cy - cv)) * inv_pz;

      /* shading terms */
      const float nd = o_dot3(n, d);
      const float inv_d3 = 1.0f / (d2 * dn);
      const float inv_d5 = inv_d3 / d2;
      const float k3 = 3.0f * nd * inv_d5;
      const float s = aa * ii;
      /* a = aa * dS/dp, b = aa * dS/dn */
      const of3 a = o3(s * (k3 * d.v[0] - n.v[0] * inv_d3), s * (k3 * d.v[1] - n.v[1] * inv_d3), s * (k3 * d.v[2] - n.v[2] * inv_d3));
      const of3 b = o3(s * d.v[0] * inv_d3, s * d.v[1] * inv_d3, s * d.v[2] * inv_d3);
      const of3 pa = o_cross3(Xcp, a);
      const of3 nb = o_cross3(n, b);

      jacobian[0] = J[0] - (pa.v[0] + nb.v[0]);
      jacobian[1] = J[1] - (pa.v[1] + nb.v[1]);
      jacobian[2] = J[2] - (pa.v[2] + nb.v[2]);
      if (translation_enabled)
      {
        jacobian[3] = J[3] - a.v[0];
        jacobian[4] = J[4] - a.v[1];
        jacobian[5] = J[5] - a.v[2];
      }
    }
  }
  else
  {
    /* :283-322 default to standard depth tracking */
    const of3 Xcq = o_unproject_d(&frm->projection, frame_x + 0.5f, frame_y + 0.5f, frame_depth);
    const of3 delta = o_sub3(Xcp, Xcq);
    if (residual) *residual = o_dot3(delta, n);

    if (jacobian)
    {
      const float dx = delta.v[0], dy = delta.v[1], dz = delta.v[2];
      const float nx = n.v[0], ny = n.v[1], nz = n.v[2];
      jacobian[0] = dz * ny - dy * nz - ny * pz + nz * py;
      jacobian[1] = dx * nz - dz * nx + nx * pz - nz * px;
      jacobian[2] = dy * nx - dx * ny - nx * py + ny * px;
      if (translation_enabled)
      {
        jacobian[3] = nx;
        jacobian[4] = ny;
        jacobian[5] = nz;
      }
    }
  }
}

void orc_light_tracker_compute_residuals(const vk_color_view* keyframe, const vk_color_view* frame,
    const vk_light_terms* terms, const vk_transform* Tcm, float* residuals)
{
#pragma omp parallel for
  for (int y = 0; y < keyframe->height; ++y)
    for (int x = 0; x < keyframe->width; ++x)
      evaluate_light(0, x, y, Tcm, terms, keyframe, frame, &residuals[y * keyframe->width + x], NULL);
}

void orc_light_tracker_compute_jacobian(const vk_color_view* keyframe, const vk_color_view* frame,
    const vk_light_terms* terms, const vk_transform* Tcm, int translation_enabled, float* jacobian)
{
#pragma omp parallel for
  for (int y = 0; y < keyframe->height; ++y)
    for (int x = 0; x < keyframe->width; ++x)
      evaluate_light(translation_enabled, x, y, Tcm, terms, keyframe, frame, NULL,
          &jacobian[6 * (y * keyframe->width + x)]);
}

/* ref: light_tracker.cu:373-531; float32 products, double sums (as the other trackers) */
void orc_light_tracker_compute_system(const vk_color_view* keyframe, const vk_color_view* frame,
    const vk_light_terms* terms, const vk_transform* Tcm, int translation_enabled, double* hessian,
    double* gradient)
{
  const int parameter_count = translation_enabled ? 6 : 3;
  for (int i = 0; i < 21; ++i) hessian[i] = 0;
  for (int i = 0; i < 6; ++i) gradient[i] = 0;

  for (int y = 0; y < keyframe->height; ++y)
    for (int x = 0; x < keyframe->width; ++x)
    {
      float r, J[6];
      evaluate_light(translation_enabled, x, y, Tcm, terms, keyframe, frame, &r, J);
      for (int i = 0; i < parameter_count; ++i) gradient[i] += (double)(J[i] * r);
      int counter = 0;
      for (int rr = 0; rr < parameter_count; ++rr)
        for (int c = 0; c <= rr; ++c, ++counter)
          hessian[counter] += (double)(J[rr] * J[c]);
    }
}
