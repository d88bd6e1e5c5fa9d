// projection.h — pinhole camera (ref: include/vulcan/projection.h). Defaults
// f = (500, 500), c = (320, 240); Project/Unproject keep the reference's
// expression order so host-side expectations match the device arithmetic.
#pragma once

#include <vk.h>
#include <vulcan/matrix.h>

namespace vulcan
{

class Projection
{
  public:

    Projection() : focal_length_(500, 500), center_point_(320, 240) {}

    const Vector2f& GetFocalLength() const { return focal_length_; }

    void SetFocalLength(const Vector2f& length)
    {
      VULCAN_DEBUG(length[0] > 0 && length[1] > 0);
      VULCAN_DEBUG(!isnan(length[0]) && !isnan(length[1]));
      focal_length_ = length;
    }

    void SetFocalLength(float w, float h) { SetFocalLength(Vector2f(w, h)); }

    const Vector2f& GetCenterPoint() const { return center_point_; }

    void SetCenterPoint(const Vector2f& point)
    {
      VULCAN_DEBUG(!isnan(point[0]) && !isnan(point[1]));
      center_point_ = point;
    }

    void SetCenterPoint(float w, float h) { SetCenterPoint(Vector2f(w, h)); }

    Vector2f Project(const Vector3f& Xcp) const
    {
      const float inv_w = 1.0f / Xcp[2];
      return Vector2f(inv_w * focal_length_[0] * Xcp[0] + center_point_[0],
                      inv_w * focal_length_[1] * Xcp[1] + center_point_[1]);
    }

    Vector2f Project(float x, float y, float z) const { return Project(Vector3f(x, y, z)); }

    Vector3f Unproject(const Vector2f& uv) const
    {
      const float ifx = 1.0f / focal_length_[0];
      const float ify = 1.0f / focal_length_[1];
      return Vector3f(ifx * uv[0] - center_point_[0] * ifx, ify * uv[1] - center_point_[1] * ify, 1);
    }

    Vector3f Unproject(float u, float v) const { return Unproject(Vector2f(u, v)); }

    Vector3f Unproject(const Vector2f& uv, float d) const { return d * Unproject(uv); }

    Vector3f Unproject(float u, float v, float d) const { return Unproject(Vector2f(u, v), d); }

    vk_projection ToVk() const
    {
      vk_projection k;
      k.fx = focal_length_[0];
      k.fy = focal_length_[1];
      k.cx = center_point_[0];
      k.cy = center_point_[1];
      return k;
    }

  protected:

    Vector2f focal_length_;

    Vector2f center_point_;
};

} // namespace vulcan
