// voxel.h — 20-byte TSDF sample, binary-compatible with vk_voxel and with the
// reference's Voxel (ref: include/vulcan/voxel.h:43-49).
#pragma once

#include <vk.h>
#include <vulcan/matrix.h>

namespace vulcan
{

class Voxel
{
  public:

    Voxel() {}

    const Vector3f& GetColor() const { return color; }

    void SetColor(const Vector3f& c) { color = c; }

    // never-observed voxel: far in front of any surface, no weight
    static Voxel Empty()
    {
      Voxel v;
      v.distance = 1;
      v.color = Vector3f::Zeros();
      v.distance_weight = 0;
      v.color_weight = 0;
      return v;
    }

  public:

    float distance;

    Vector3f color;

    short distance_weight;

    short color_weight;
};

static_assert(sizeof(Voxel) == sizeof(vk_voxel), "Voxel must match vk_voxel");

} // namespace vulcan
