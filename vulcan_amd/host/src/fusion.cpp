// fusion.cpp — Integrator and its three subclasses over vk_integrate_*
// (ref: src/integrator.cu, src/depth_integrator.cu:83-115,
//  src/color_integrator.cu:139-204, src/light_integrator.cu:254-354).
#include <vulcan/fusion.h>
#include <vulcan/exception.h>
#include <vulcan/observation.h>
#include <vulcan/tsdf_volume.h>

namespace vulcan
{

Integrator::Integrator(std::shared_ptr<Volume> volume) :
  volume_(volume),
  depth_range_(0.1f, 5.0f),
  max_distance_weight_(16),
  max_color_weight_(16)
{
}

std::shared_ptr<Volume> Integrator::GetVolume() const { return volume_; }

const Vector2f& Integrator::GetDepthRange() const { return depth_range_; }

void Integrator::SetDepthRange(const Vector2f& range)
{
  VULCAN_DEBUG(range[0] > 0 && range[0] < range[1]);
  depth_range_ = range;
}

void Integrator::SetDepthRange(float min, float max) { SetDepthRange(Vector2f(min, max)); }

float Integrator::GetMaxDistanceWeight() const { return max_distance_weight_; }

void Integrator::SetMaxDistanceWeight(float weight)
{
  VULCAN_DEBUG(weight > 0);
  max_distance_weight_ = weight;
}

float Integrator::GetMaxColorWeight() const { return max_color_weight_; }

void Integrator::SetMaxColorWeight(float weight)
{
  VULCAN_DEBUG(weight > 0);
  max_color_weight_ = weight;
}

vk_integrator Integrator::ToVk() const
{
  vk_integrator p;
  p.min_depth = depth_range_[0];
  p.max_depth = depth_range_[1];
  p.max_distance_weight = max_distance_weight_;
  p.max_color_weight = max_color_weight_;
  return p;
}

// ---- depth -------------------------------------------------------------------

DepthIntegrator::DepthIntegrator(std::shared_ptr<Volume> volume) : Integrator(volume) {}

// The Integrate() of each subclass goes through vk_integrate_ahead: when a Tracer is
// attached to the volume, the same launch prepares the raycast bounds of this view.
void DepthIntegrator::Integrate(const Frame& frame)
{
  const vk_volume v = volume_->ToVk();
  const vk_integrator p = ToVk();
  const vk_frame f = frame.ToVk();
  VK_ASSERT(vk_integrate_ahead(&v, &p, &f, 0, nullptr, nullptr, nullptr, volume_->GetViewBounds(), Device::GetStream()));
  volume_->NoteIntegrated();
}

// ---- colour ------------------------------------------------------------------

ColorIntegrator::ColorIntegrator(std::shared_ptr<Volume> volume) : Integrator(volume) {}

void ColorIntegrator::Integrate(const Frame& frame)
{
  const vk_volume v = volume_->ToVk();
  const vk_integrator p = ToVk();
  const vk_frame f = frame.ToVk();
  VK_ASSERT(vk_integrate_ahead(&v, &p, &f, 1, nullptr, nullptr, nullptr, volume_->GetViewBounds(), Device::GetStream()));
  volume_->NoteIntegrated();
}

void ColorIntegrator::IntegrateDepth(const Frame& frame)
{
  const vk_volume v = volume_->ToVk();
  const vk_integrator p = ToVk();
  const vk_frame f = frame.ToVk();
  VK_ASSERT(vk_integrate_depth(&v, &p, &f, Device::GetStream()));
  volume_->NoteIntegrated();
}

void ColorIntegrator::IntegrateColor(const Frame& frame)
{
  const vk_volume v = volume_->ToVk();
  const vk_integrator p = ToVk();
  const vk_frame f = frame.ToVk();
  VK_ASSERT(vk_integrate_color(&v, &p, &f, Device::GetStream()));
  volume_->NoteIntegrated();
}

// ---- light -------------------------------------------------------------------

LightIntegrator::LightIntegrator(std::shared_ptr<Volume> volume) :
  Integrator(volume),
  depth_threshold_(0.2f)
{
}

LightIntegrator::~LightIntegrator() { volume_->DetachLightPreparation(frame_mask_.GetData()); }

const Light& LightIntegrator::GetLight() const { return light_; }

void LightIntegrator::SetLight(const Light& light) { light_ = light; }

void LightIntegrator::Integrate(const Frame& frame)
{
  // ComputeFrameMask (light_integrator.cu:277-293) plus the per-pixel half of IntegrateColor
  // (:215-225): mask and Tcd * normal as one record per pixel for the fused voxel pass
  VULCAN_ASSERT_MSG(frame.depth_image && frame.color_image && frame.normal_image, "missing depth, color or normal image");
  const int w = frame.depth_image->GetWidth(), h = frame.depth_image->GetHeight();
  frame_mask_.Resize(w, h);
  pixel_records_.Resize(4 * w, h);
  const vk_volume v = volume_->ToVk();
  const vk_integrator p = ToVk();
  const vk_light l = light_.ToVk();
  const vk_frame f = frame.ToVk();
  // Volume::SetView prepares these buffers in its own request pass once they are registered
  // (vk_light_prep); whenever it has not done so for this very frame the pass runs here
  vk_light_prep* prep = volume_->GetLightPreparation();
  if (!prep || prep->mask != frame_mask_.GetData() || prep->records != pixel_records_.GetData() ||
      prep->capacity != w * h || prep->depth_threshold != depth_threshold_)
  {
    volume_->AttachLightPreparation(frame_mask_.GetData(), pixel_records_.GetData(), w * h, depth_threshold_);
    prep = volume_->GetLightPreparation();
  }
  if (vk_light_prepared(prep, &f, depth_threshold_)) prep->valid = 0;   // used once
  else VK_ASSERT(vk_light_prepare(&f, depth_threshold_, frame_mask_.GetData(), pixel_records_.GetData(), Device::GetStream()));
  VK_ASSERT(vk_integrate_ahead(&v, &p, &f, 2, &l, frame_mask_.GetData(), pixel_records_.GetData(),
      volume_->GetViewBounds(), Device::GetStream()));
  volume_->NoteIntegrated();
}

void LightIntegrator::ComputeFrameMask(const Frame& frame)
{
  VULCAN_ASSERT_MSG(frame.depth_image && frame.color_image, "missing depth or color image");
  frame_mask_.Resize(frame.depth_image->GetWidth(), frame.depth_image->GetHeight());
  const vk_frame f = frame.ToVk();
  VK_ASSERT(vk_light_compute_frame_mask(&f, depth_threshold_, frame_mask_.GetData(), Device::GetStream()));
}

void LightIntegrator::IntegrateDepth(const Frame& frame)
{
  const vk_volume v = volume_->ToVk();
  const vk_integrator p = ToVk();
  const vk_frame f = frame.ToVk();
  VK_ASSERT(vk_integrate_depth(&v, &p, &f, Device::GetStream()));
  volume_->NoteIntegrated();
}

void LightIntegrator::IntegrateColor(const Frame& frame)
{
  const vk_volume v = volume_->ToVk();
  const vk_integrator p = ToVk();
  const vk_light l = light_.ToVk();
  const vk_frame f = frame.ToVk();
  VK_ASSERT(vk_integrate_light_color(&v, &p, &l, frame_mask_.GetData(), &f, Device::GetStream()));
  volume_->NoteIntegrated();
}

} // namespace vulcan
