// ref: include/vulcan/depth_tracker.h — projective point-to-plane ICP.
#pragma once

#include <vulcan/matrix.h>
#include <vulcan/tracker.h>

namespace vulcan
{

class DepthTracker : public Tracker
{
  public:

    DepthTracker();

    virtual ~DepthTracker();

    void ComputeResiduals(const Frame& frame, Buffer<float>& residuals) const;

    void ComputeJacobian(const Frame& frame, Buffer<Vector6f>& jacobian) const;

    // host form of the pose update (depth_tracker.cpp:22-86), for tests
    void ApplyUpdate(Frame& frame, const Vector6f& x) const override;

  protected:

    int GetResidualCount(const Frame& frame) const override;

    void ComputeSystem(const Frame& frame) override;

    void TrackOnDevice(Frame& frame) override;
};

} // namespace vulcan
