/*
 * oracle_trace.c — CPU restatement of src/tracer.cu and src/frame.cu (TEST
 * INFRASTRUCTURE, see oracle.h). One loop iteration per reference thread.
 */
#include <string.h>
#include "oracle.h"
#include "oracle_math.h"

/* ref: tracer.cu:13-87 ComputePatchesKernel. Threads run in ascending index
 * order, so patch offsets are the serial prefix sum (util.cuh:52-95). A block
 * whose count is 0 writes nothing (the reference stores to patches[-1 + ..]
 * there). */
void orc_trace_compute_patches(const int32_t* indices, const vk_hash_entry* entries,
    const vk_transform* Tcw, const vk_projection* projection, float block_length,
    float min_depth, float max_depth, int block_count, int image_width,
    int image_height, int bounds_width, int bounds_height, vk_patch* patches,
    int patch_capacity, int32_t* patch_count)
{
  const int patch_size = VK_PATCH_MAX_SIZE;
  int total = *patch_count;

  for (int index = 0; index < block_count; ++index)
  {
    int16_t bmax[2] = { -1, -1 };
    int16_t bmin[2] = { (int16_t)bounds_width, (int16_t)bounds_height };
    float depth_bounds[2] = { +FLT_MAX, -FLT_MAX };

    const vk_hash_entry entry = entries[indices[index]];
    const int16_t* origin = entry.block.origin;

    for (int z = 0; z <= 1; ++z)
    {
      const float wz = block_length * (z + origin[2]);

      for (int y = 0; y <= 1; ++y)
      {
        const float wy = block_length * (y + origin[1]);

        for (int x = 0; x <= 1; ++x)
        {
          const float wx = block_length * (x + origin[0]);
          const of4 X = o_xform(Tcw->m, wx, wy, wz, 1.0f);
          const of3 Xcp = o3(X.v[0], X.v[1], X.v[2]);
          float u, v;
          o_project(projection, Xcp, &u, &v);

          u = bounds_width * u / image_width;     /* :51 */
          v = bounds_height * v / image_height;   /* :52 */

          /* :54-58, min/max/clamp on short */
          bmin[0] = (int16_t)o_clampi(o_mini(o_f2s(floorf(u)), bmin[0]), 0, bounds_width - 1);
          bmin[1] = (int16_t)o_clampi(o_mini(o_f2s(floorf(v)), bmin[1]), 0, bounds_height - 1);
          bmax[0] = (int16_t)o_clampi(o_maxi(o_f2s(ceilf(u)), bmax[0]), 0, bounds_width - 1);
          bmax[1] = (int16_t)o_clampi(o_maxi(o_f2s(ceilf(v)), bmax[1]), 0, bounds_height - 1);

          /* :60-61 */
          depth_bounds[0] = o_clamp(o_min(Xcp.v[2], depth_bounds[0]), min_depth, max_depth);
          depth_bounds[1] = o_clamp(o_max(Xcp.v[2], depth_bounds[1]), min_depth, max_depth);
        }
      }
    }

    /* :67-71 */
    const int rx = bmax[0] - bmin[0];
    const int ry = bmax[1] - bmin[1];
    const int gx = (rx + patch_size - 1) / patch_size;
    const int gy = (ry + patch_size - 1) / patch_size;
    const int count = (depth_bounds[1] > depth_bounds[0]) ? gx * gy : 0;
    if (count <= 0) continue;

    const int offset = total;
    total += count;

    /* :74-86 */
    for (int i = 0; i < gy; ++i)
      for (int j = 0; j < gx; ++j)
      {
        vk_patch patch;
        const int output = offset + i * gx + j;
        patch.origin[0] = (int16_t)(bmin[0] + patch_size * j);
        patch.origin[1] = (int16_t)(bmin[1] + patch_size * i);
        patch.size[0] = (int16_t)o_mini(patch_size, bmax[0] - patch.origin[0] + 1);
        patch.size[1] = (int16_t)o_mini(patch_size, bmax[1] - patch.origin[1] + 1);
        patch.bounds[0] = depth_bounds[0];
        patch.bounds[1] = depth_bounds[1];
        if (output < patch_capacity) patches[output] = patch;
      }
  }

  *patch_count = total;
}

/* ref: tracer.cu:494-500 */
void orc_trace_reset_bounds(float* bounds, int count)
{
  for (int i = 0; i < count; ++i)
  {
    bounds[2 * i + 0] = +FLT_MAX;
    bounds[2 * i + 1] = -FLT_MAX;
  }
}

/* ref: tracer.cu:89-112 ComputeBoundsKernel; atomicMin/Max util.cuh:20-50 */
void orc_trace_compute_bounds(const vk_patch* patches, float* bounds,
    int bounds_width, int patch_count)
{
  for (int index = 0; index < patch_count; ++index)
  {
    const vk_patch patch = patches[index];

    for (int i = 0; i < patch.size[1]; ++i)
    {
      const int y = patch.origin[1] + i;

      for (int j = 0; j < patch.size[0]; ++j)
      {
        const int x = patch.origin[0] + j;
        const int pixel = y * bounds_width + x;
        bounds[2 * pixel + 0] = o_min(patch.bounds[0], bounds[2 * pixel + 0]);
        bounds[2 * pixel + 1] = o_max(patch.bounds[1], bounds[2 * pixel + 1]);
      }
    }
  }
}

static inline vk_voxel voxel_empty(void)  /* voxel.h:31-39 */
{
  vk_voxel e;
  e.distance = 1;
  e.color[0] = e.color[1] = e.color[2] = 0;
  e.distance_weight = 0;
  e.color_weight = 0;
  return e;
}

/* ref: tracer.cu:114-188 GetVoxel */
static vk_voxel get_voxel(uint32_t K, const vk_hash_entry* entries,
    const vk_voxel* voxels, int bx, int by, int bz, int vx, int vy, int vz)
{
  const int r = VK_BLOCK_RESOLUTION;

  if (vx < 0) { --bx; vx = r + vx; } else if (vx >= r) { ++bx; vx = vx - r; }
  if (vy < 0) { --by; vy = r + vy; } else if (vy >= r) { ++by; vy = vy - r; }
  if (vz < 0) { --bz; vz = r + vz; } else if (vz >= r) { ++bz; vz = vz - r; }

  vk_hash_entry entry = entries[o_hash(bx, by, bz, K)];
  int found = 0;

  for (;;)
  {
    if (o_block_eq(&entry.block, bx, by, bz)) { found = 1; break; }
    else if (entry.next == -1) break;
    entry = entries[entry.next];
  }

  if (found && entry.data != -1)
    return voxels[VK_BLOCK_VOXELS * entry.data + vz * 64 + vy * 8 + vx];

  return voxel_empty();
}

/* ref: tracer.cu:190-315 GetInterpolatedDistance; out = (sdf, r, g, b) */
static void get_interpolated_distance(const vk_hash_entry* entries,
    const vk_voxel* voxels, uint32_t K, float block_length, float voxel_length,
    int bx, int by, int bz, const vk_hash_entry* entry, of3 p, float out[4])
{
  const float wx = (p.v[0] - bx * block_length) / voxel_length;
  const float wy = (p.v[1] - by * block_length) / voxel_length;
  const float wz = (p.v[2] - bz * block_length) / voxel_length;

  const int block_offset = VK_BLOCK_VOXELS * entry->data;

  const int i0x = o_f2i(floorf(wx - 0.5f));
  const int i0y = o_f2i(floorf(wy - 0.5f));
  const int i0z = o_f2i(floorf(wz - 0.5f));

  vk_voxel vv[8];  /* index = dz*4 + dy*2 + dx : v000 v001 v010 v011 v100 ... */

  if (i0x >= 0 && i0y >= 0 && i0z >= 0 && i0x < 7 && i0y < 7 && i0z < 7)
  {
    for (int k = 0; k < 8; ++k)
    {
      const int dx = k & 1, dy = (k >> 1) & 1, dz = (k >> 2) & 1;
      vv[k] = voxels[block_offset + (i0z + dz) * 64 + (i0y + dy) * 8 + (i0x + dx)];
    }
  }
  else
  {
    for (int k = 0; k < 8; ++k)
    {
      const int dx = k & 1, dy = (k >> 1) & 1, dz = (k >> 2) & 1;
      vv[k] = get_voxel(K, entries, voxels, bx, by, bz, i0x + dx, i0y + dy, i0z + dz);
    }
  }

  float w1[3], w0[3];
  w1[0] = wx - (i0x + 0.5f);
  w1[1] = wy - (i0y + 0.5f);
  w1[2] = wz - (i0z + 0.5f);
  w0[0] = 1.0f - w1[0];
  w0[1] = 1.0f - w1[1];
  w0[2] = 1.0f - w1[2];

  /* :274-280 */
  const float n00 = vv[0].distance * w0[0] + vv[1].distance * w1[0];
  const float n01 = vv[2].distance * w0[0] + vv[3].distance * w1[0];
  const float n10 = vv[4].distance * w0[0] + vv[5].distance * w1[0];
  const float n11 = vv[6].distance * w0[0] + vv[7].distance * w1[0];
  const float n0 = n00 * w0[1] + n01 * w1[1];
  const float n1 = n10 * w0[1] + n11 * w1[1];

  /* :282-289 — `a*b*c*(cw>0) ? 1 : 0` parses as `(a*b*c*(cw>0)) ? 1 : 0`,
   * so each contributing corner has weight 1. */
  float cwt[8];
  for (int k = 0; k < 8; ++k)
  {
    const int dx = k & 1, dy = (k >> 1) & 1, dz = (k >> 2) & 1;
    const float wz_ = dz ? w1[2] : w0[2];
    const float wy_ = dy ? w1[1] : w0[1];
    const float wx_ = dx ? w1[0] : w0[0];
    const float prod = wz_ * wy_ * wx_ * (vv[k].color_weight > 0);
    cwt[k] = prod ? 1.0f : 0.0f;
  }

  float total = 0;
  for (int k = 0; k < 8; ++k) total += cwt[k];

  of3 color = o3(0, 0, 0);
  for (int k = 0; k < 8; ++k)
    color = o_add3(color, o_scale3(o3(vv[k].color[0], vv[k].color[1], vv[k].color[2]), cwt[k]));
  if (total > 0) color = o_div3(color, total);

  out[0] = n0 * w0[2] + n1 * w1[2];
  out[1] = color.v[0];
  out[2] = color.v[1];
  out[3] = color.v[2];
}

/* walk a bucket chain until the block matches or the chain ends, tracer.cu:364-369 */
static inline vk_hash_entry find_entry(const vk_hash_entry* entries, uint32_t K,
    int bx, int by, int bz)
{
  vk_hash_entry entry = entries[o_hash(bx, by, bz, K)];
  while (!o_block_eq(&entry.block, bx, by, bz) && entry.next != -1)
    entry = entries[entry.next];
  return entry;
}

/* ref: tracer.cu:317-451 ComputePointsKernel, one call per pixel */
static void compute_point(const vk_hash_entry* entries, const vk_voxel* voxels,
    const float* bounds, uint32_t K, float block_length, float voxel_length,
    float trunc_length, const vk_transform* Twc, const vk_projection* projection,
    float* depths, float* colors, int image_width, int image_height,
    int bounds_width, int bounds_height, int x, int y, int32_t* steps)
{
  const int px = bounds_width * x / image_width;
  const int py = bounds_height * y / image_height;
  const float bound0 = bounds[2 * (py * bounds_width + px) + 0];
  const float bound1 = bounds[2 * (py * bounds_width + px) + 1];

  float depth = 0;
  float final_depth = 0;
  of3 color = o3(0, 0, 0);
  int iters = 0;

  if (bound0 < bound1)
  {
    const of3 Xcp = o_unproject_d(projection, x + 0.5f, y + 0.5f, bound0);
    const of3 Xwp = o_xform_point(Twc->m, Xcp);
    const of3 dir = o_normalized3(o_xform_dir(Twc->m, Xcp));
    const float* Tcw = Twc->inv;  /* :350 */

    of3 p = Xwp;
    depth = bound0;
    color = o3(0, 0, 0);

    do
    {
      const int bx = o_f2i(floorf(p.v[0] / block_length));
      const int by = o_f2i(floorf(p.v[1] / block_length));
      const int bz = o_f2i(floorf(p.v[2] / block_length));
      const vk_hash_entry entry = find_entry(entries, K, bx, by, bz);

      if (o_block_eq(&entry.block, bx, by, bz) && entry.data != -1)
      {
        const float wx = (p.v[0] - bx * block_length) / voxel_length;
        const float wy = (p.v[1] - by * block_length) / voxel_length;
        const float wz = (p.v[2] - bz * block_length) / voxel_length;

        /* :377-379 int(w) can reach 8 at a block face; the
         * reference then reads a voxel of the next row / next pool slot.
         * Clamped to 7 here and in the HIP path (deliberate, memory safety). */
        const int vx = o_mini(o_f2i(wx), 7);
        const int vy = o_mini(o_f2i(wy), 7);
        const int vz = o_mini(o_f2i(wz), 7);

        const vk_voxel voxel = voxels[VK_BLOCK_VOXELS * entry.data + vz * 64 + vy * 8 + vx];
        float sdf = voxel.distance;

        if (sdf <= 0.1f && sdf >= -0.5f)
        {
          float v4[4];
          get_interpolated_distance(entries, voxels, K, block_length, voxel_length,
              bx, by, bz, &entry, p, v4);
          sdf = v4[0];
          color = o3(v4[1], v4[2], v4[3]);
        }

        if (sdf <= 0.0f)
        {
          p = o_add3(p, o_scale3(dir, trunc_length * sdf));

          const int bx2 = o_f2i(floorf(p.v[0] / block_length));
          const int by2 = o_f2i(floorf(p.v[1] / block_length));
          const int bz2 = o_f2i(floorf(p.v[2] / block_length));
          const vk_hash_entry entry2 = find_entry(entries, K, bx2, by2, bz2);

          if (o_block_eq(&entry2.block, bx2, by2, bz2) && entry2.data != -1)
          {
            float v4[4];
            get_interpolated_distance(entries, voxels, K, block_length, voxel_length,
                bx2, by2, bz2, &entry2, p, v4);
            sdf = v4[0];
            color = o3(v4[1], v4[2], v4[3]);
            p = o_add3(p, o_scale3(dir, trunc_length * sdf));
          }

          final_depth = o_xform_point(Tcw, p).v[2];
          break;
        }
        else
        {
          p = o_add3(p, o_scale3(dir, o_max(voxel_length, trunc_length * sdf)));
        }
      }
      else
      {
        p = o_add3(p, o_scale3(dir, block_length));
      }

      depth = o_xform_point(Tcw, p).v[2];

      if (++iters >= 500)
      {
        color = o3(1, 0, 0);
        break;
      }
    }
    while (depth < bound1);
  }

  const int pixel = y * image_width + x;
  depths[pixel] = final_depth;
  colors[3 * pixel + 0] = color.v[0];
  colors[3 * pixel + 1] = color.v[1];
  colors[3 * pixel + 2] = color.v[2];
  if (steps) steps[pixel] = iters;
  (void)image_height;
}

/* ref: tracer.cu:478-492 */
void orc_trace_compute_points(const vk_hash_entry* entries, const vk_voxel* voxels,
    const float* bounds, int block_count, float block_length, float voxel_length,
    float trunc_length, const vk_transform* Twc, const vk_projection* projection,
    float* depths, float* colors, int image_width, int image_height,
    int bounds_width, int bounds_height, int32_t* steps)
{
#pragma omp parallel for schedule(dynamic, 4) num_threads(orc_get_threads())
  for (int y = 0; y < image_height; ++y)
    for (int x = 0; x < image_width; ++x)
      compute_point(entries, voxels, bounds, (uint32_t)block_count, block_length,
          voxel_length, trunc_length, Twc, projection, depths, colors, image_width,
          image_height, bounds_width, bounds_height, x, y, steps);
}

/* ref: frame.cu:9-122 ComputeNormalsKernel<16>; the LDS tile holds 0 outside
 * the image (:24-33) */
static inline float depth_at(const float* depths, int w, int h, int x, int y)
{
  return (x >= 0 && x < w && y >= 0 && y < h) ? depths[y * w + x] : 0.0f;
}

void orc_frame_compute_normals(const float* depths, const vk_projection* projection,
    float* normals, int image_width, int image_height)
{
  const int pad = 2;

#pragma omp parallel for num_threads(orc_get_threads())
  for (int y = 0; y < image_height; ++y)
    for (int x = 0; x < image_width; ++x)
    {
      const float depth = depths[y * image_width + x];
      of3 normal = o3(0, 0, 0);

      if (depth > 0)
      {
        float d;
        const of3 z0 = o_unproject_d(projection, (x + 0) + 0.5f, (y + 0) + 0.5f, depth);

        of3 x0, x1, y0, y1;
        d = depth_at(depths, image_width, image_height, x - pad, y);
        x0 = (d == 0) ? z0 : o_scale3(o_unproject(projection, (x - pad) + 0.5f, (y + 0) + 0.5f), d);
        d = depth_at(depths, image_width, image_height, x + pad, y);
        x1 = (d == 0) ? z0 : o_scale3(o_unproject(projection, (x + pad) + 0.5f, (y + 0) + 0.5f), d);
        d = depth_at(depths, image_width, image_height, x, y - pad);
        y0 = (d == 0) ? z0 : o_scale3(o_unproject(projection, (x + 0) + 0.5f, (y - pad) + 0.5f), d);
        d = depth_at(depths, image_width, image_height, x, y + pad);
        y1 = (d == 0) ? z0 : o_scale3(o_unproject(projection, (x + 0) + 0.5f, (y + pad) + 0.5f), d);

        const of3 dx = o_sub3(x0, x1);
        const of3 dy = o_sub3(y0, y1);

        if (o_sqnorm3(dx) > 0 && o_sqnorm3(dy) > 0)
          normal = o_normalized3(o_cross3(dy, dx));
      }

      const int output = y * image_width + x;
      normals[3 * output + 0] = normal.v[0];
      normals[3 * output + 1] = normal.v[1];
      normals[3 * output + 2] = normal.v[2];
    }
}

/* ref: frame.cu:126-181 FilterDepthsKernel<16> */
void orc_frame_filter_depths(int image_width, int image_height, const float* src,
    float* dst)
{
  const int pad = 3;

#pragma omp parallel for num_threads(orc_get_threads())
  for (int y = 0; y < image_height; ++y)
    for (int x = 0; x < image_width; ++x)
    {
      const float d0 = src[y * image_width + x];
      float dn = 0;
      float w = 0;

      for (int i = -pad; i <= pad; ++i)
        for (int j = -pad; j <= pad; ++j)
        {
          const float dk = depth_at(src, image_width, image_height, x + j, y + i);
          const float delta = d0 - dk;
          float sq = 0;  /* Vector2f(i, j).SquaredNorm() */
          sq += (float)i * (float)i;
          sq += (float)j * (float)j;
          const float wr = expf(-sq / (pad * pad));
          const float ws = expf(-(delta * delta) / 0.0004f);
          const float ww = wr * ws;
          dn += ww * dk;
          w += ww;
        }

      dst[y * image_width + x] = dn / w;
    }
}
