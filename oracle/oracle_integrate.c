/*
 * oracle_integrate.c — CPU restatement of src/depth_integrator.cu,
 * src/color_integrator.cu and src/light_integrator.cu (TEST INFRASTRUCTURE,
 * see oracle.h). One loop iteration per reference thread: the outer loop is
 * blockIdx.x over the visible list, the inner loops are threadIdx z,y,x.
 */
#include "oracle.h"
#include "oracle_math.h"

/* voxel position in the world frame, depth_integrator.cu:35-39 */
static inline of3 voxel_world(const vk_block* block, float block_length,
    float voxel_length, int x, int y, int z)
{
  const of3 block_offset = o_scale3(o3(block->origin[0], block->origin[1], block->origin[2]), block_length);
  const of3 voxel_offset = o_scale3(o3(x + 0.5f, y + 0.5f, z + 0.5f), voxel_length);
  return o_add3(block_offset, voxel_offset);
}

/* ref: depth_integrator.cu:17-80 IntegrateKernel (== color_integrator.cu:18-80,
 * light_integrator.cu:106-168) */
static void integrate_depth_block(const vk_volume* v, const vk_integrator* p,
    const vk_frame* f, const float* Tdw, int entry_index)
{
  const vk_hash_entry entry = v->hash_entries[entry_index];
  const float voxel_length = v->voxel_length;
  const float block_length = VK_BLOCK_RESOLUTION * voxel_length;
  const float truncation_length = v->truncation_length;

  /* The reference marks the never-allocated origin block visible with
   * data == -1 and then indexes voxels[-512..]: skipped. */
  if (entry.data < 0) return;

  for (int z = 0; z < 8; ++z)
    for (int y = 0; y < 8; ++y)
      for (int x = 0; x < 8; ++x)
      {
        const of3 Xwp = voxel_world(&entry.block, block_length, voxel_length, x, y, z);
        const of3 Xdp = o_xform_point(Tdw, Xwp);
        float u, w;
        o_project(&f->depth_projection, Xdp, &u, &w);

        if (u >= 0 && u < f->width && w >= 0 && w < f->height)
        {
          const int image_index = (int)w * f->width + (int)u;
          const float depth = f->depth[image_index];
          if (depth < p->min_depth || depth > p->max_depth) continue;

          const float distance = depth - Xdp.v[2];

          if (distance > -truncation_length)
          {
            const int voxel_index = entry.data * VK_BLOCK_VOXELS + z * 64 + y * 8 + x;
            vk_voxel voxel = v->voxels[voxel_index];

            const float prev_dist = voxel.distance_weight * voxel.distance;
            const float curr_dist = o_min(1.0f, distance / truncation_length);
            const float dist_weight = voxel.distance_weight + 1;
            voxel.distance_weight = (int16_t)o_min(p->max_distance_weight, dist_weight);
            voxel.distance = (prev_dist + curr_dist) / dist_weight;

            v->voxels[voxel_index] = voxel;
          }
        }
      }
}

/* ref: depth_integrator.cu:89-115 */
void orc_integrate_depth(const vk_volume* v, const vk_integrator* p, const vk_frame* f)
{
  const int count = v->counters[VK_CTR_VISIBLE];
  const float* Tdw = f->depth_to_world.inv;  /* Twd.Inverse(), :104 */

#pragma omp parallel for schedule(dynamic, 16) num_threads(orc_get_threads())
  for (int i = 0; i < count; ++i)
    integrate_depth_block(v, p, f, Tdw, v->visible_blocks[i]);
}

/* ref: color_integrator.cu:83-135 IntegrateColorKernel */
static void integrate_color_block(const vk_volume* v, const vk_integrator* p,
    const vk_frame* f, const float* Tcw, int entry_index)
{
  const vk_hash_entry entry = v->hash_entries[entry_index];
  const float voxel_length = v->voxel_length;
  const float block_length = VK_BLOCK_RESOLUTION * voxel_length;

  /* color_integrator.cu:183-184: the COLOUR image's size */
  const int image_width = f->color_width > 0 ? f->color_width : f->width;
  const int image_height = f->color_height > 0 ? f->color_height : f->height;

  if (entry.data < 0) return;

  for (int z = 0; z < 8; ++z)
    for (int y = 0; y < 8; ++y)
      for (int x = 0; x < 8; ++x)
      {
        const of3 Xwp = voxel_world(&entry.block, block_length, voxel_length, x, y, z);
        const of3 Xcp = o_xform_point(Tcw, Xwp);
        float u, w;
        o_project(&f->color_projection, Xcp, &u, &w);

        if (u >= 0 && u < image_width && w >= 0 && w < image_height)
        {
          const int voxel_index = entry.data * VK_BLOCK_VOXELS + z * 64 + y * 8 + x;
          vk_voxel voxel = v->voxels[voxel_index];

          if (fabsf(voxel.distance) < 1.0)
          {
            const int image_index = (int)w * image_width + (int)u;
            const float cw = voxel.color_weight;
            const of3 prev_color = o_scale3(o3(voxel.color[0], voxel.color[1], voxel.color[2]), cw);
            const of3 curr_color = o3(f->color[3 * image_index + 0],
                f->color[3 * image_index + 1], f->color[3 * image_index + 2]);
            const float color_weight = voxel.color_weight + 1;
            voxel.color_weight = (int16_t)o_min(p->max_color_weight, color_weight);
            const of3 c = o_div3(o_add3(prev_color, curr_color), color_weight);
            voxel.color[0] = c.v[0];
            voxel.color[1] = c.v[1];
            voxel.color[2] = c.v[2];
            v->voxels[voxel_index] = voxel;
          }
        }
      }
}

/* ref: color_integrator.cu:178-204 */
void orc_integrate_color(const vk_volume* v, const vk_integrator* p, const vk_frame* f)
{
  const int count = v->counters[VK_CTR_VISIBLE];
  const vk_transform Tdw = o_transform_inverse(&f->depth_to_world);
  const vk_transform Tcw = o_transform_mul(&f->depth_to_color, &Tdw);  /* :192 */

#pragma omp parallel for schedule(dynamic, 16) num_threads(orc_get_threads())
  for (int i = 0; i < count; ++i)
    integrate_color_block(v, p, f, Tcw.m, v->visible_blocks[i]);
}

/* ref: light_integrator.cu:17-103 ComputeFrameMaskKernel<16,3>. The tile is
 * loaded with a -1 halo offset but read with a +3 centre (:40-41 vs :83-84), so
 * pixel (x,y) inspects [x-1,x+5] x [y-1,y+5]; out-of-image samples are 0. */
void orc_light_compute_frame_mask(const vk_frame* f, float depth_threshold, float* mask)
{
  const int width = f->width, height = f->height;

#pragma omp parallel for num_threads(orc_get_threads())
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x)
    {
      const int index = y * width + x;
      const float* value = &f->color[3 * index];

      if (value[0] < 0.02f || value[0] > 0.98f ||
          value[1] < 0.02f || value[1] > 0.98f ||
          value[2] < 0.02f || value[2] > 0.98f)
      {
        mask[index] = 0.0f;
        continue;
      }

      float dmin = +FLT_MAX;
      float dmax = -FLT_MAX;

      for (int i = -3; i <= 3; ++i)
        for (int j = -3; j <= 3; ++j)
        {
          const int vx = x + 2 + j;
          const int vy = y + 2 + i;
          float depth = 0;
          if (vx >= 0 && vx < width && vy >= 0 && vy < height) depth = f->depth[vy * width + vx];
          dmin = fminf(depth, dmin);
          dmax = fmaxf(depth, dmax);
        }

      mask[index] = (dmax - dmin <= depth_threshold) ? 1.0f : 0.0f;
    }
}

/* ref: light.h:53-60 Light::GetShading */
static inline float light_shading(const vk_light* l, of3 point, of3 normal)
{
  const of3 delta = o_sub3(o3(l->position[0], l->position[1], l->position[2]), point);
  const of3 direction = o_normalized3(delta);
  const float distance_squared = o_sqnorm3(delta);
  const float cos_theta = o_dot3(normal, direction);
  return l->intensity * cos_theta / distance_squared;
}

/* ref: light_integrator.cu:170-250 IntegrateColorKernel */
static void integrate_light_block(const vk_volume* v, const vk_integrator* p,
    const vk_light* light, const float* mask, const vk_frame* f, const float* Tdw,
    const float* Tcw, const float* Tcd, int entry_index)
{
  const vk_hash_entry entry = v->hash_entries[entry_index];
  const float voxel_length = v->voxel_length;
  const float block_length = VK_BLOCK_RESOLUTION * voxel_length;
  const int image_width = f->width, image_height = f->height;

  if (entry.data < 0) return;

  for (int z = 0; z < 8; ++z)
    for (int y = 0; y < 8; ++y)
      for (int x = 0; x < 8; ++x)
      {
        const of3 Xwp = voxel_world(&entry.block, block_length, voxel_length, x, y, z);
        const of3 Xdp = o_xform_point(Tdw, Xwp);
        const of3 Xcp = o_xform_point(Tcw, Xwp);
        float du, dv, cu, cv;
        o_project(&f->depth_projection, Xdp, &du, &dv);
        o_project(&f->color_projection, Xcp, &cu, &cv);

        if (du >= 0 && du < image_width && dv >= 0 && dv < image_height &&
            cu >= 0 && cu < image_width && cv >= 0 && cv < image_height)
        {
          const int depth_index = (int)dv * image_width + (int)du;
          const int color_index = (int)cv * image_width + (int)cu;

          if (mask[depth_index] > 0.5f)
          {
            const int voxel_index = entry.data * VK_BLOCK_VOXELS + z * 64 + y * 8 + x;
            vk_voxel voxel = v->voxels[voxel_index];

            if (fabsf(voxel.distance) < 1.0)
            {
              of3 curr_color = o3(f->color[3 * color_index + 0],
                  f->color[3 * color_index + 1], f->color[3 * color_index + 2]);
              const of3 Xdn = o3(f->normals[3 * depth_index + 0],
                  f->normals[3 * depth_index + 1], f->normals[3 * depth_index + 2]);
              const of3 Xcn = o_xform_dir(Tcd, Xdn);
              const float shading = light_shading(light, Xcp, Xcn);

              if (shading > 0.05f)
              {
                curr_color = o_div3(curr_color, shading);
                const float color_weight = voxel.color_weight + 1;
                const float cw = voxel.color_weight;
                const of3 prev_color = o_scale3(o3(voxel.color[0], voxel.color[1], voxel.color[2]), cw);
                const of3 c = o_div3(o_add3(prev_color, curr_color), color_weight);
                voxel.color[0] = c.v[0];
                voxel.color[1] = c.v[1];
                voxel.color[2] = c.v[2];
                voxel.color_weight = (int16_t)o_min(p->max_color_weight, color_weight);
                v->voxels[voxel_index] = voxel;
              }
            }
          }
        }
      }
}

/* ref: light_integrator.cu:323-354 */
void orc_integrate_light_color(const vk_volume* v, const vk_integrator* p,
    const vk_light* light, const float* mask, const vk_frame* f)
{
  const int count = v->counters[VK_CTR_VISIBLE];
  const vk_transform Tdw = o_transform_inverse(&f->depth_to_world);
  const vk_transform Tcw = o_transform_mul(&f->depth_to_color, &Tdw);  /* :337-338 */

#pragma omp parallel for schedule(dynamic, 16) num_threads(orc_get_threads())
  for (int i = 0; i < count; ++i)
    integrate_light_block(v, p, light, mask, f, Tdw.m, Tcw.m, f->depth_to_color.m,
        v->visible_blocks[i]);
}
