// detection.cpp — Detector host class over vk_detect* (ref: src/detector.cu:66-222).
#include <vulcan/detection.h>
#include <cstring>
#include <limits>
#include <vulcan/exception.h>

namespace vulcan
{

Detector::Detector() : origin_(0, 0, 0)
{
  std::memset(&params_, 0, sizeof(params_));
  std::memset(&result_, 0, sizeof(result_));
  params_.radius = 2.0f;                                  // detector.cu:66-72
  params_.min_inlier_count = 100;
  for (int axis = 0; axis < 3; ++axis)                    // :214-221: open intervals
  {
    bounds_[axis] = Vector2f(1, -1);
    params_.bounds[axis][0] = 1;
    params_.bounds[axis][1] = -1;
  }
  state_.Resize(1);
}

Detector::~Detector() {}

float Detector::GetRadius() const { return params_.radius; }

void Detector::SetRadius(float radius) { params_.radius = radius; }

const Vector3f& Detector::GetOrigin() const { return origin_; }

void Detector::SetOrigin(const Vector3f& origin)
{
  origin_ = origin;
  for (int i = 0; i < 3; ++i) params_.origin[i] = origin[i];
}

const Vector2f& Detector::GetBounds(int axis) const
{
  VULCAN_DEBUG(axis >= 0 && axis < 3);
  return bounds_[axis];
}

void Detector::SetBounds(int axis, const Vector2f& bounds)
{
  VULCAN_DEBUG(axis >= 0 && axis < 3);
  bounds_[axis] = bounds;
  params_.bounds[axis][0] = bounds[0];
  params_.bounds[axis][1] = bounds[1];
}

bool Detector::GetBoundsUseOwnAxis() const { return params_.bounds_use_own_axis != 0; }

void Detector::SetBoundsUseOwnAxis(bool enabled) { params_.bounds_use_own_axis = enabled ? 1 : 0; }

int Detector::GetMinInlierCount() const { return params_.min_inlier_count; }

void Detector::SetMinInlierCount(int count)
{
  VULCAN_DEBUG(count > 0);
  params_.min_inlier_count = count;
}

void Detector::Prepare(const Buffer<Vector3f>& points)
{
  const int count = int(points.GetSize());
  points_.Resize(count > 0 ? count : 1);
  workspace_.Resize(vk_detect_workspace_bytes(count));
}

void Detector::ReadState()
{
  state_.CopyToHost(&result_);                            // blocking
  points_.Resize(result_.inlier_count);
}

Vector3f Detector::Detect(const Buffer<Vector3f>& points)
{
  Prepare(points);
  VK_ASSERT(vk_detect(&params_, reinterpret_cast<const float*>(points.GetData()), int(points.GetSize()),
      reinterpret_cast<float*>(points_.GetData()), state_.GetData(), workspace_.GetData(),
      Device::GetStream()));
  ReadState();
  return BoxDetected() ? GetValidPosition() : GetInvalidPosition();
}

void Detector::Filter(const Buffer<Vector3f>& points)
{
  Prepare(points);
  VK_ASSERT(vk_detect_filter(&params_, reinterpret_cast<const float*>(points.GetData()),
      int(points.GetSize()), reinterpret_cast<float*>(points_.GetData()), state_.GetData(),
      workspace_.GetData(), Device::GetStream()));
  ReadState();
}

const vk_detect_state& Detector::GetState() const { return result_; }

const Buffer<Vector3f>& Detector::GetInliers() const { return points_; }

bool Detector::BoxDetected() const { return int(points_.GetSize()) >= params_.min_inlier_count; }

// valid after Detect(); after a bare Filter() the position has not been computed
Vector3f Detector::GetValidPosition() const
{
  return Vector3f(result_.position[0], result_.position[1], result_.position[2]);
}

Vector3f Detector::GetInvalidPosition() const
{
  const float nan = std::numeric_limits<float>::quiet_NaN();
  return Vector3f(nan, nan, nan);
}

int Detector::GetBufferSize() const { return result_.inlier_count; }

} // namespace vulcan
