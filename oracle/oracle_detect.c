/* oracle_detect.c — CPU restatement of the reference's Detector
 * (src/detector.cu). TEST INFRASTRUCTURE ONLY, see oracle.h.
 *
 * PARITY UNPINNED for this file: nothing in the reference exercises Detector
 * (tests/detector_test.cu is a 0-byte placeholder; no app or source
 * instantiates the class), so there is no vector to pin against. What is
 * restated, line by line:
 *   FilterKernel   detector.cu:14-35   radius + interval predicate (incl. the
 *                                      upstream quirk that all three intervals
 *                                      test point[0], :26-28)
 *   GetValidPosition :131-141         sum|x| / n per axis (cublasSasum)
 *   DistanceKernel + Sdot :54-64,171-177   sum of Norm(p - c)^2
 *   RemovalKernel  :37-52             keep Norm(p - c) <= 1.5 * stdev
 *   Detect         :120-124           position, or NaN below min_inlier_count
 *
 * Two things the reference leaves open are fixed here (and identically in
 * vk_detect.hip) so that results are reproducible:
 *   - compaction keeps input order (upstream: per-thread-block atomic, any order)
 *   - float sums use ONE fixed tree: chunks of 4096 points; inside a chunk 256
 *     strided partial sums (element t, t+256, ...), folded by a binary tree
 *     (stride 128 ... 1); chunk sums added in chunk order. cuBLAS does not
 *     document its order, so any order is "a" reference result.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "oracle_math.h"

enum { CHUNK = 4096, LANES = 256 };

typedef float (*term_fn)(const float* point, int axis, const float* center);

static float term_abs(const float* point, int axis, const float* center)
{
  (void)center;
  return fabsf(point[axis]);
}

static float term_dist2(const float* point, int axis, const float* center)
{
  (void)axis;
  const of3 d = o_sub3(o3(point[0], point[1], point[2]), o3(center[0], center[1], center[2]));
  const float n = o_norm3(d);      /* DistanceKernel stores the norm ... */
  return n * n;                    /* ... and Sdot squares it again       */
}

static float tree_sum(const float* points, int count, int axis, const float* center, term_fn term)
{
  float total = 0.0f;
  for (int base = 0; base < count; base += CHUNK)
  {
    float part[LANES];
    for (int t = 0; t < LANES; ++t)
    {
      float acc = 0.0f;
      for (int i = base + t; i < count && i < base + CHUNK; i += LANES) acc += term(points + 3 * i, axis, center);
      part[t] = acc;
    }
    for (int stride = LANES / 2; stride >= 1; stride /= 2)
      for (int t = 0; t < stride; ++t) part[t] += part[t + stride];
    total += part[0];
  }
  return total;
}

static void centroid(const float* points, int count, float* out)
{
  const float inv = 1.0f / (float)count;            /* Matrix::operator/=, matrix.h:290-295 */
  for (int a = 0; a < 3; ++a) out[a] = tree_sum(points, count, a, NULL, term_abs) * inv;
}

static int inside(const vk_detector* d, const float* p)
{
  const of3 rel = o_sub3(o3(p[0], p[1], p[2]), o3(d->origin[0], d->origin[1], d->origin[2]));
  if (!(d->radius <= 0 || o_norm3(rel) < d->radius)) return 0;
  for (int a = 0; a < 3; ++a)
  {
    const float v = d->bounds_use_own_axis ? p[a] : p[0];
    const float lo = d->bounds[a][0], hi = d->bounds[a][1];
    if (!(lo > hi || (v >= lo && v <= hi))) return 0;
  }
  return 1;
}

void orc_detect(const vk_detector* detector, const float* points, int count, float* inliers,
    vk_detect_state* state)
{
  const float nan = NAN;
  memset(state, 0, sizeof(*state));
  float* filtered = (float*)malloc(sizeof(float) * 3 * (size_t)(count > 0 ? count : 1));

  int n = 0;
  for (int i = 0; i < count; ++i)
    if (inside(detector, points + 3 * i)) { memcpy(filtered + 3 * n, points + 3 * i, 12); ++n; }
  state->filtered_count = n;

  int m = 0;
  if (n > 0)                                           /* detector.cu:163 */
  {
    centroid(filtered, n, state->center);
    state->squared_error = tree_sum(filtered, n, 0, state->center, term_dist2);
    const float stdev = sqrtf(state->squared_error / (float)n);
    state->limit = 1.5f * stdev;
    for (int i = 0; i < n; ++i)
    {
      const float* p = filtered + 3 * i;
      const of3 d = o_sub3(o3(p[0], p[1], p[2]), o3(state->center[0], state->center[1], state->center[2]));
      if (o_norm3(d) <= state->limit) { memcpy(inliers + 3 * m, p, 12); ++m; }
    }
  }
  state->inlier_count = m;
  state->detected = (m >= detector->min_inlier_count) ? 1 : 0;
  if (state->detected) centroid(inliers, m, state->position);   /* m == 0 gives 0 * inf = NaN, as upstream */
  else state->position[0] = state->position[1] = state->position[2] = nan;
  free(filtered);
}
