// fusion.h — the three TSDF integrators of the hot path, declared together
// because they share one device kernel family (vk_integrate_*) and one
// parameter block (vk_integrator).
//
// API parity: class names, public methods and protected stage methods follow
// the reference's integrator.h, depth_integrator.h, color_integrator.h and
// light_integrator.h; those four header names remain as forwarders to this file.
#pragma once

#include <memory>
#include <vk.h>
#include <vulcan/image.h>
#include <vulcan/light.h>
#include <vulcan/matrix.h>

namespace vulcan
{

struct Frame;
class Volume;

// Common state: the volume being fused into, the accepted depth interval and
// the two running-average caps (distance weight, colour weight).
class Integrator
{
  public:
    explicit Integrator(std::shared_ptr<Volume> volume);
    virtual ~Integrator() {}

    std::shared_ptr<Volume> GetVolume() const;

    // accepted depth interval [min, max] in metres
    const Vector2f& GetDepthRange() const;
    void SetDepthRange(const Vector2f& range);
    void SetDepthRange(float min, float max);

    // running-average caps
    float GetMaxDistanceWeight() const;
    void SetMaxDistanceWeight(float weight);
    float GetMaxColorWeight() const;
    void SetMaxColorWeight(float weight);

    // fuse one frame into the blocks Volume::SetView marked visible
    virtual void Integrate(const Frame& frame) = 0;

  protected:
    vk_integrator ToVk() const;   // parameter block handed to the C ABI

    std::shared_ptr<Volume> volume_;
    Vector2f depth_range_;
    float max_distance_weight_;
    float max_color_weight_;
};

// Distance only (vk_integrate_depth).
class DepthIntegrator : public Integrator
{
  public:
    explicit DepthIntegrator(std::shared_ptr<Volume> volume);
    void Integrate(const Frame& frame) override;
};

// Distance + colour. Integrate() is ONE pass over the visible voxels
// (vk_integrate_depth_color); the two protected stages remain for callers and
// tests that want the reference's two-pass form.
class ColorIntegrator : public Integrator
{
  public:
    explicit ColorIntegrator(std::shared_ptr<Volume> volume);
    void Integrate(const Frame& frame) override;

  protected:
    void IntegrateDepth(const Frame& frame);
    void IntegrateColor(const Frame& frame);
};

// Distance + albedo: colour divided by the shading of a point light at the
// camera, gated by a per-pixel mask (saturation and depth discontinuities).
class LightIntegrator : public Integrator
{
  public:
    explicit LightIntegrator(std::shared_ptr<Volume> volume);

    ~LightIntegrator();   // takes its buffers off the volume's light preparation record

    const Light& GetLight() const;
    void SetLight(const Light& light);

    void Integrate(const Frame& frame) override;

  protected:
    void ComputeFrameMask(const Frame& frame);
    void IntegrateDepth(const Frame& frame);
    void IntegrateColor(const Frame& frame);

    Light light_;
    Image frame_mask_;
    Image pixel_records_;           // {Tcd * normal, mask} per pixel as a 4w x h image (vk_light_prepare); not upstream
    float depth_threshold_;
};

} // namespace vulcan
