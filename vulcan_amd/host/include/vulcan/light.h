// light.h — point light used by LightIntegrator (ref: include/vulcan/light.h):
// shading = intensity * cos(theta) / distance^2.
#pragma once

#include <vk.h>
#include <vulcan/matrix.h>

namespace vulcan
{

class Light
{
  public:

    Light() : intensity_(1.0f), position_(0, 0, 0) {}

    float GetIntensity() const { return intensity_; }

    void SetIntensity(float intensity)
    {
      VULCAN_ASSERT(intensity >= 0);
      intensity_ = intensity;
    }

    const Vector3f& GetPosition() const { return position_; }

    void SetPosition(const Vector3f& position) { position_ = position; }

    void SetPosition(float x, float y, float z) { SetPosition(Vector3f(x, y, z)); }

    float GetShading(const Vector3f& point, const Vector3f& normal) const
    {
      const Vector3f delta = position_ - point;
      const Vector3f direction = delta.Normalized();
      return intensity_ * normal.Dot(direction) / delta.SquaredNorm();
    }

    vk_light ToVk() const
    {
      vk_light l;
      l.intensity = intensity_;
      for (int i = 0; i < 3; ++i) l.position[i] = position_[i];
      return l;
    }

  protected:

    float intensity_;

    Vector3f position_;
};

} // namespace vulcan
