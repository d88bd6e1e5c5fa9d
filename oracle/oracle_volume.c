/*
 * oracle_volume.c — CPU restatement of src/volume.cu (TEST INFRASTRUCTURE,
 * see oracle.h). Each function runs the reference kernel's threads one after
 * another in ascending thread order.
 */
#include <string.h>
#include <stddef.h>
#include "oracle.h"
#include "oracle_math.h"

static int g_threads = 1;
int  orc_version(void) { return 100; }
void orc_set_threads(int n) { g_threads = n > 0 ? n : 1; }
int  orc_get_threads(void) { return g_threads; }

/* ref: volume.cu:552-627 Volume::Initialize */
void orc_volume_initialize(const vk_volume* v)
{
  const int max_count = v->main_block_count + v->excess_block_count;
  const size_t voxels = (size_t)max_count * VK_BLOCK_VOXELS;

  for (size_t i = 0; i < voxels; ++i)  /* volume.cu:565-571, voxel.h:31-39 */
  {
    vk_voxel e;
    e.distance = 1;
    e.color[0] = e.color[1] = e.color[2] = 0;
    e.distance_weight = 0;
    e.color_weight = 0;
    v->voxels[i] = e;
  }

  for (int i = 0; i < max_count; ++i)
  {
    vk_hash_entry e;  /* hash.h:18-22, block.h:19-22 */
    e.block.origin[0] = e.block.origin[1] = e.block.origin[2] = 0;
    e.block.pad = 0;
    e.data = -1;
    e.next = -1;
    v->hash_entries[i] = e;                        /* volume.cu:573-579 */
    v->free_voxel_blocks[i] = i;                   /* volume.cu:588-594 */
    v->block_visibility[i] = VK_VISIBILITY_FALSE;  /* volume.cu:616-622 */
  }

  for (int i = 0; i < v->main_block_count; ++i)
  {
    v->allocation_types[i] = VK_ALLOC_NONE;        /* volume.cu:603-609 */
    memset(&v->allocation_blocks[i], 0, sizeof(vk_block));
  }

  memset(v->counters, 0, sizeof(int32_t) * VK_CTR_COUNT);
  v->counters[VK_CTR_EXCESS_PTR] = v->main_block_count;  /* volume.cu:581-586 */
  v->counters[VK_CTR_VOXEL_PTR] = max_count - 1;         /* volume.cu:596-601 */
}

/* ref: volume.cu:465-471 */
void orc_volume_reset_block_visibility(const vk_volume* v)
{
  const int count = v->main_block_count + v->excess_block_count;
  for (int i = 0; i < count; ++i)
    if (v->block_visibility[i] == VK_VISIBILITY_TRUE)
      v->block_visibility[i] = VK_VISIBILITY_UNKNOWN;
}

static inline uint64_t request_key(int type, int bx, int by, int bz)
{
  return ((uint64_t)(uint16_t)type << 48) | ((uint64_t)(uint16_t)(int16_t)bz << 32) |
         ((uint64_t)(uint16_t)(int16_t)by << 16) | (uint64_t)(uint16_t)(int16_t)bx;
}

static inline void post_request(const vk_volume* v, uint32_t h, int type, int bx,
    int by, int bz, int policy)
{
  vk_block b;
  b.origin[0] = (int16_t)bx;
  b.origin[1] = (int16_t)by;
  b.origin[2] = (int16_t)bz;
  b.pad = (int16_t)type;

  if (policy == ORC_POLICY_MAXKEY)
  {
    uint64_t cur;
    memcpy(&cur, &v->allocation_blocks[h], 8);
    if (request_key(type, bx, by, bz) <= cur) return;
  }

  v->allocation_types[h] = (uint8_t)type;
  v->allocation_blocks[h] = b;
}

/* ref: volume.cu:87-301 CreateAllocationRequestsKernel, one call per pixel */
static void create_requests_pixel(const vk_volume* v, const float* depths,
    int image_width, int x, int y, const vk_projection* projection,
    const vk_transform* Twd, int policy)
{
  const uint32_t K = (uint32_t)v->main_block_count;
  const float block_length = VK_BLOCK_RESOLUTION * v->voxel_length;  /* volume.cu:508 */
  const float truncation_length = v->truncation_length;

  /* :102-111 */
  of3 direction = o_unproject(projection, x + 0.5f, y + 0.5f);
  direction = o_xform_dir(Twd->m, direction);
  const of3 origin = o3(Twd->m[12], Twd->m[13], Twd->m[14]);

  /* :114-117 */
  const float depth = depths[y * image_width + x];
  if (depth < v->min_depth || depth > v->max_depth) return;

  /* :122-129 */
  const of3 Xwp = o_add3(origin, o_scale3(direction, depth));
  direction = o_normalized3(direction);
  const of3 begin = o_sub3(Xwp, o_scale3(direction, truncation_length));
  const of3 end = o_add3(Xwp, o_scale3(direction, truncation_length));

  /* :132-134 */
  const int step_x = (direction.v[0] < 0) ? -1 : 1;
  const int step_y = (direction.v[1] < 0) ? -1 : 1;
  const int step_z = (direction.v[2] < 0) ? -1 : 1;

  /* :137-145 */
  const float inv_block_length = 1.0f / block_length;
  int bx = o_f2i(floorf(begin.v[0] * inv_block_length));
  int by = o_f2i(floorf(begin.v[1] * inv_block_length));
  int bz = o_f2i(floorf(begin.v[2] * inv_block_length));
  const int ex = o_f2i(floorf(end.v[0] * inv_block_length));
  const int ey = o_f2i(floorf(end.v[1] * inv_block_length));
  const int ez = o_f2i(floorf(end.v[2] * inv_block_length));

  /* :148-150 */
  const float ox = block_length * (bx + o_maxi(0, step_x)) - begin.v[0];
  const float oy = block_length * (by + o_maxi(0, step_y)) - begin.v[1];
  const float oz = block_length * (bz + o_maxi(0, step_z)) - begin.v[2];

  /* :153-160 */
  float tmax_x = ox / direction.v[0];
  float tmax_y = oy / direction.v[1];
  float tmax_z = oz / direction.v[2];
  if (direction.v[0] == 0) tmax_x = (float)1E20;
  if (direction.v[1] == 0) tmax_y = (float)1E20;
  if (direction.v[2] == 0) tmax_z = (float)1E20;

  /* :163-165 */
  const float tdelta_x = (step_x * block_length) / direction.v[0];
  const float tdelta_y = (step_y * block_length) / direction.v[1];
  const float tdelta_z = (step_z * block_length) / direction.v[2];

  /* :174-299 */
  for (;;)
  {
    const uint32_t hash_code = o_hash(bx, by, bz, K);
    vk_hash_entry entry = v->hash_entries[hash_code];

    if (o_block_eq(&entry.block, bx, by, bz))             /* :186 */
    {
      v->block_visibility[hash_code] = VK_VISIBILITY_TRUE;
    }
    else if (entry.data == -1)                             /* :193 */
    {
      v->block_visibility[hash_code] = VK_VISIBILITY_TRUE;
      post_request(v, hash_code, VK_ALLOC_MAIN, bx, by, bz, policy);
    }
    else                                                   /* :203 */
    {
      int found = 0;
      uint32_t index = hash_code;

      while (entry.next != -1)
      {
        index = (uint32_t)entry.next;
        entry = v->hash_entries[index];

        if (o_block_eq(&entry.block, bx, by, bz))
        {
          v->block_visibility[index] = VK_VISIBILITY_TRUE;
          found = 1;
          break;
        }
      }

      if (!found) post_request(v, hash_code, VK_ALLOC_EXCESS, bx, by, bz, policy);
    }

    if (tmax_x < tmax_y)                                   /* :242 */
    {
      if (tmax_x < tmax_z)
      {
        bx += step_x;
        if (bx == ex + step_x) break;
        tmax_x += tdelta_x;
      }
      else
      {
        bz += step_z;
        if (bz == ez + step_z) break;
        tmax_z += tdelta_z;
      }
    }
    else
    {
      if (tmax_y < tmax_z)
      {
        by += step_y;
        if (by == ey + step_y) break;
        tmax_y += tdelta_y;
      }
      else
      {
        bz += step_z;
        if (bz == ez + step_z) break;
        tmax_z += tdelta_z;
      }
    }
  }
}

/* ref: volume.cu:497-518 */
void orc_volume_create_allocation_requests(const vk_volume* v, const float* depth,
    int width, int height, const vk_projection* projection,
    const vk_transform* Twd, int policy)
{
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x)
      create_requests_pixel(v, depth, width, x, y, projection, Twd, policy);
}

/* ref: volume.cu:304-368,520-535 — threads in ascending index order */
void orc_volume_handle_allocation_requests(const vk_volume* v)
{
  const int count = v->main_block_count;
  const int max_count = v->main_block_count + v->excess_block_count;
  int committed = 0, dropped = 0;

  for (int index = 0; index < count; ++index)
  {
    const int type = v->allocation_types[index];
    if (type == VK_ALLOC_NONE) continue;

    vk_hash_entry entry;
    int entry_index = index;
    entry.block = v->allocation_blocks[index];
    entry.block.pad = 0;
    entry.data = -1;
    entry.next = -1;

    if (type == VK_ALLOC_EXCESS)
    {
      int other_index = index;
      vk_hash_entry other = v->hash_entries[other_index];

      while (other.next != -1)
      {
        other_index = other.next;
        other = v->hash_entries[other_index];
      }

      entry_index = v->counters[VK_CTR_EXCESS_PTR]++;       /* :337 atomicAdd */

      if (entry_index < max_count)
      {
        v->hash_entries[other_index].next = entry_index;    /* :344 */
        v->block_visibility[entry_index] = VK_VISIBILITY_TRUE;
      }
    }

    const int voxel_index = v->counters[VK_CTR_VOXEL_PTR]--;  /* :352 atomicSub */

    if (entry_index < max_count && voxel_index >= 0)           /* :356 */
    {
      entry.data = v->free_voxel_blocks[voxel_index];
      v->hash_entries[entry_index] = entry;
      ++committed;
    }
    else
    {
      ++dropped;
    }

    v->allocation_types[index] = VK_ALLOC_NONE;               /* :365 */
    memset(&v->allocation_blocks[index], 0, sizeof(vk_block));
  }

  v->counters[VK_CTR_REQUESTS] = committed + dropped;
  v->counters[VK_CTR_DROPPED] += dropped;
}

/* ref: volume.cu:25-84,473-495 — threads in ascending index order, so the
 * compacted list comes out sorted (the reference's order is unspecified) */
void orc_volume_update_block_visibility(const vk_volume* v, int image_width,
    int image_height, const vk_projection* projection, const vk_transform* Tdw)
{
  const int count = v->main_block_count + v->excess_block_count;
  const float block_length = VK_BLOCK_RESOLUTION * v->voxel_length;
  int out = 0;                                                /* :488 ResetBufferSize */

  for (int index = 0; index < count; ++index)
  {
    const int visibility = v->block_visibility[index];
    int visible = (visibility == VK_VISIBILITY_TRUE);

    if (visibility == VK_VISIBILITY_UNKNOWN)
    {
      const int16_t* origin = v->hash_entries[index].block.origin;

      for (int i = 0; i < 8; ++i)
      {
        const float wx = block_length * (origin[0] + ((i & 1) >> 0));
        const float wy = block_length * (origin[1] + ((i & 2) >> 1));
        const float wz = block_length * (origin[2] + ((i & 4) >> 2));
        const of4 Xdp = o_xform(Tdw->m, wx, wy, wz, 1.0f);

        if (Xdp.v[2] < 0) continue;

        float u, w;
        o_project(projection, o3(Xdp.v[0], Xdp.v[1], Xdp.v[2]), &u, &w);

        if (u >= 0 && u <= image_width && w >= 0 && w <= image_height)
        {
          visible = 1;
          break;
        }
      }

      if (!visible) v->block_visibility[index] = VK_VISIBILITY_FALSE;
    }

    if (visible) v->visible_blocks[out++] = index;
  }

  v->counters[VK_CTR_VISIBLE] = out;
}

/* ref: volume.cu:430-437 */
void orc_volume_set_view(const vk_volume* v, const vk_frame* frame, int policy)
{
  const vk_transform Tdw = o_transform_inverse(&frame->depth_to_world);
  orc_volume_reset_block_visibility(v);
  orc_volume_create_allocation_requests(v, frame->depth, frame->width, frame->height,
      &frame->depth_projection, &frame->depth_to_world, policy);
  orc_volume_handle_allocation_requests(v);
  orc_volume_update_block_visibility(v, frame->width, frame->height,
      &frame->depth_projection, &Tdw);
}

/* ---- known-answer hooks used by tests/test_oracle_kats.py ---------------- */
uint32_t orc_kat_hash(int bx, int by, int bz, uint32_t K) { return o_hash(bx, by, bz, K); }
void orc_kat_project(const vk_projection* k, float x, float y, float z, float* uv)
{
  o_project(k, o3(x, y, z), &uv[0], &uv[1]);
}
void orc_kat_unproject(const vk_projection* k, float u, float v, float d, float* xyz)
{
  const of3 r = o_unproject_d(k, u, v, d);
  xyz[0] = r.v[0]; xyz[1] = r.v[1]; xyz[2] = r.v[2];
}
void orc_kat_sizes(int* out)
{
  out[0] = (int)sizeof(vk_voxel);      out[1] = (int)sizeof(vk_block);
  out[2] = (int)sizeof(vk_hash_entry); out[3] = (int)sizeof(vk_patch);
  out[4] = (int)sizeof(vk_projection); out[5] = (int)sizeof(vk_transform);
  out[6] = (int)sizeof(vk_light);      out[7] = (int)offsetof(vk_voxel, distance_weight);
}
