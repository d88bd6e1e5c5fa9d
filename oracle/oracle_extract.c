/*
 * oracle_extract.c — CPU statement of the mesh extraction (TEST INFRASTRUCTURE, see oracle.h).
 *
 * The reference's extractor is unfinished: src/extractor.cu:129-389 classify the cubes of
 * one block, list the cut edges among the three that leave each cube's minimum corner and
 * place a vertex at every such edge's midpoint; ExtractVertexIndicesKernel and
 * ExtractFacesKernel (:392-430) are empty, vertices of cubes on the block's far faces are
 * dropped (:246-296), positions use a block stride of resolution - 1 voxels (:319-321) and
 * the edge-offset table is indexed past its end for edges 3 and 8 (:323-325). There is no
 * upstream test (tests/extractor_test.cpp is empty). PARITY UNPINNED: what is stated here
 * is the finished algorithm the device implements, with the reference's conventions kept
 * where it has any:
 *   - corner c of a cube is voxel (x + (c & 1), y + (c >> 1 & 1), z + (c >> 2)); bit c of the
 *     state is set when that voxel's distance is > 0 (:185-200); a cube with a corner whose
 *     distance weight is 0 — or whose block is not allocated — is empty (:202-210);
 *   - a cube owns the vertices on its edges 0 (+x), 3 (+y), 8 (+z) (:16-31, :61-118);
 *   - blocks come from the visible list, as upstream (":TODO: replace with all allocated
 *     blocks", :455-457), or from the whole table.
 * Finished here: corners beyond a block's far faces are read from the neighbouring blocks
 * through the hash table; a voxel sits at voxel_length * (8 * block + xyz + 0.5), where the
 * integrators put it (depth_integrator.cu:35-38); the vertex divides its edge where the
 * linearly interpolated distance is 0 (the weights of the reference's commented-out
 * :345-357 are the wrong way round; `interpolate == 0` gives its active midpoint rule,
 * :361); faces come from the table of tools/gen_mc_table.py.
 * Order: blocks in list order, cubes by z*64 + y*8 + x, vertices by axis x, y, z, faces in
 * table order.
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "oracle_math.h"
#include "oracle_mc_table.inc"

static vk_hash_entry find_block(const vk_hash_entry* entries, uint32_t K, int bx, int by, int bz)
{
  vk_hash_entry entry = entries[o_hash(bx, by, bz, K)];
  while (!o_block_eq(&entry.block, bx, by, bz) && entry.next != -1) entry = entries[entry.next];
  if (!o_block_eq(&entry.block, bx, by, bz)) entry.data = -1;
  return entry;
}

/* voxel (vx, vy, vz) in [0, 16) relative to block (bx, by, bz): returns 0 when unknown */
static int get_corner(const vk_volume* v, const int* slots /* 8 neighbour slots */, int vx, int vy, int vz,
    float* distance)
{
  const int m = (vx >> 3) | ((vy >> 3) << 1) | ((vz >> 3) << 2);
  const int slot = slots[m];
  if (slot < 0) return 0;
  const vk_voxel* voxel = &v->voxels[(size_t)slot * VK_BLOCK_VOXELS + (vz & 7) * 64 + (vy & 7) * 8 + (vx & 7)];
  if (voxel->distance_weight == 0) return 0;      /* extractor.cu:202-205 */
  *distance = voxel->distance;
  return 1;
}

/* owner cube (offset from this cube) and axis of each of the 12 edges */
static const signed char kEdgeOwner[12][4] = {
  {0, 0, 0, 0}, {1, 0, 0, 1}, {0, 1, 0, 0}, {0, 0, 0, 1},
  {0, 0, 1, 0}, {1, 0, 1, 1}, {0, 1, 1, 0}, {0, 0, 1, 1},
  {0, 0, 0, 2}, {1, 0, 0, 2}, {1, 1, 0, 2}, {0, 1, 0, 2},
};

/* points: [3 * point_capacity], faces: [3 * face_capacity]; counts[0..2] = points, faces, cubes skipped
 * because a vertex they need belongs to a block that is not in the list. Returns 0, or -1 when a
 * capacity is too small (counts are still the full totals). */
int orc_extract_mesh(const vk_volume* v, int all_allocated, int interpolate, float* points, int point_capacity,
    int32_t* faces, int face_capacity, int32_t* counts)
{
  const int total = v->main_block_count + v->excess_block_count;
  const uint32_t K = (uint32_t)v->main_block_count;
  int* list = (int*)malloc(sizeof(int) * (size_t)total);
  int n = 0;
  if (all_allocated)
  {
    for (int i = 0; i < total; ++i) if (v->hash_entries[i].data >= 0) list[n++] = i;
  }
  else
  {
    const int count = v->counters[VK_CTR_VISIBLE];
    for (int i = 0; i < count; ++i) if (v->hash_entries[v->visible_blocks[i]].data >= 0) list[n++] = v->visible_blocks[i];
  }

  /* position of every pool slot in the list (-1: not listed) and the first vertex index of
   * each (listed block, cube, axis), filled in the first sweep */
  int* listed = (int*)malloc(sizeof(int) * (size_t)total);
  for (int i = 0; i < total; ++i) listed[i] = -1;
  for (int i = 0; i < n; ++i) listed[v->hash_entries[list[i]].data] = i;
  int32_t* vertex_of = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1) * 512 * 3);
  unsigned char* state_of = (unsigned char*)malloc((size_t)(n > 0 ? n : 1) * 512);
  unsigned char* valid_of = (unsigned char*)malloc((size_t)(n > 0 ? n : 1) * 512);

  int np = 0, nf = 0, skipped = 0;
  const float L = v->voxel_length;

  for (int i = 0; i < n; ++i)      /* sweep 1: cube states and vertices */
  {
    const vk_hash_entry entry = v->hash_entries[list[i]];
    const int bx = entry.block.origin[0], by = entry.block.origin[1], bz = entry.block.origin[2];
    int slots[8];
    for (int m = 0; m < 8; ++m)
      slots[m] = (m == 0) ? entry.data : find_block(v->hash_entries, K, bx + (m & 1), by + ((m >> 1) & 1), bz + (m >> 2)).data;

    for (int z = 0; z < 8; ++z)
      for (int y = 0; y < 8; ++y)
        for (int x = 0; x < 8; ++x)
        {
          const int cube = z * 64 + y * 8 + x;
          float d[8];
          int known[8], all_known = 1;
          unsigned state = 0;
          for (int c = 0; c < 8; ++c)
          {
            known[c] = get_corner(v, slots, x + (c & 1), y + ((c >> 1) & 1), z + (c >> 2), &d[c]);
            if (!known[c]) all_known = 0;
            else if (d[c] > 0) state |= 1u << c;
          }
          state_of[(size_t)i * 512 + cube] = (unsigned char)state;
          valid_of[(size_t)i * 512 + cube] = (unsigned char)all_known;

          /* the three owned edges: corner 0 -> corner 1 (+x), 2 (+y), 4 (+z) */
          for (int axis = 0; axis < 3; ++axis)
          {
            const int c = 1 << axis;
            int32_t index = -1;
            if (known[0] && known[c] && ((d[0] > 0) != (d[c] > 0)))
            {
              index = np;
              if (np < point_capacity)
              {
                /* depth_integrator.cu:35-38: voxel centre = block_length * origin + voxel_length * (xyz + 0.5) */
                float p[3];
                p[0] = (VK_BLOCK_RESOLUTION * L) * (float)bx + L * ((float)x + 0.5f);
                p[1] = (VK_BLOCK_RESOLUTION * L) * (float)by + L * ((float)y + 0.5f);
                p[2] = (VK_BLOCK_RESOLUTION * L) * (float)bz + L * ((float)z + 0.5f);
                const float t = interpolate ? d[0] / (d[0] - d[c]) : 0.5f;
                p[axis] = p[axis] + t * L;
                points[3 * np + 0] = p[0];
                points[3 * np + 1] = p[1];
                points[3 * np + 2] = p[2];
              }
              ++np;
            }
            vertex_of[((size_t)i * 512 + cube) * 3 + axis] = index;
          }
        }
  }

  for (int i = 0; i < n; ++i)      /* sweep 2: faces */
  {
    const vk_hash_entry entry = v->hash_entries[list[i]];
    const int bx = entry.block.origin[0], by = entry.block.origin[1], bz = entry.block.origin[2];
    int owner_list[8];
    for (int m = 0; m < 8; ++m)
    {
      const int slot = (m == 0) ? entry.data : find_block(v->hash_entries, K, bx + (m & 1), by + ((m >> 1) & 1), bz + (m >> 2)).data;
      owner_list[m] = slot >= 0 ? listed[slot] : -1;
    }
    for (int cube = 0; cube < 512; ++cube)
    {
      const unsigned state = state_of[(size_t)i * 512 + cube];
      if (!valid_of[(size_t)i * 512 + cube] || state == 0 || state == 255) continue;
      const int x = cube & 7, y = (cube >> 3) & 7, z = cube >> 6;
      int32_t edge_vertex[12];
      int complete = 1;
      for (int e = 0; e < 12; ++e)
      {
        const int ox = x + kEdgeOwner[e][0], oy = y + kEdgeOwner[e][1], oz = z + kEdgeOwner[e][2];
        const int m = (ox >> 3) | ((oy >> 3) << 1) | ((oz >> 3) << 2);
        const int li = owner_list[m];
        edge_vertex[e] = (li < 0) ? -2 : vertex_of[((size_t)li * 512 + (oz & 7) * 64 + (oy & 7) * 8 + (ox & 7)) * 3 + kEdgeOwner[e][3]];
      }
      for (int t = 0; t < orc_mc_count[state]; ++t)
        for (int k = 0; k < 3; ++k)
          if (edge_vertex[orc_mc_edges[state][3 * t + k]] < 0) complete = 0;
      if (!complete) { ++skipped; continue; }
      for (int t = 0; t < orc_mc_count[state]; ++t)
      {
        if (nf < face_capacity)
          for (int k = 0; k < 3; ++k) faces[3 * nf + k] = edge_vertex[orc_mc_edges[state][3 * t + k]];
        ++nf;
      }
    }
  }

  counts[0] = np;
  counts[1] = nf;
  counts[2] = skipped;
  free(list); free(listed); free(vertex_of); free(state_of); free(valid_of);
  return (np <= point_capacity && nf <= face_capacity) ? 0 : -1;
}

/* the tables themselves, for tests/test_oracle_extract.py */
int orc_mc_triangles(int state, signed char* edges15)
{
  memcpy(edges15, orc_mc_edges[state & 255], 3 * ORC_MC_MAX_TRIANGLES);
  return orc_mc_count[state & 255];
}
