// tracer.cuh — the raycaster's stage functions on raw device pointers
// (ref: include/vulcan/tracer.cuh:14-32), each one call into the C ABI.
#pragma once

#include <vulcan/matrix.h>

namespace vulcan
{

class HashEntry;
struct Patch;
class Projection;
class Transform;
class Voxel;

void ComputePatches(const int* indices, const HashEntry* entries,
    const Transform& Tcw, const Projection& projection, float block_length,
    float min_depth, float max_depth, int block_count, int image_width,
    int image_height, int bounds_width, int bounds_height, Patch* patches,
    int* patch_count);

void ComputeBounds(const Patch* patches, Vector2f* bounds, int bounds_width,
    int patch_count);

void ComputePoints(const HashEntry* entries, const Voxel* voxels,
    const Vector2f* bounds, int block_count, float block_length,
    float voxel_length, float trunc_length, const Transform& Twc,
    const Projection& projection, float* depths, Vector3f* colors,
    int image_width, int image_height, int bounds_width, int bounds_height);

void ComputeNormals(const float* depths, const Projection& projection,
    Vector3f* normals, int image_width, int image_height);

void ResetBoundsBuffer(Vector2f* bounds, int count);

} // namespace vulcan
