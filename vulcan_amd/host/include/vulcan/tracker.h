// tracker.h — Gauss-Newton frame-to-keyframe tracker base
// (ref: include/vulcan/tracker.h, src/tracker.cpp). The reference solves the 6x6
// system on the host with Eigen::LDLT after a 42-float readback every iteration;
// here the solve and the SE(3) update run on the device (vk_icp_solve_update),
// so Track() enqueues all iterations and reads the pose back once. ApplyUpdate
// therefore takes a Vector6f instead of an Eigen::VectorXf.
#pragma once

#include <memory>
#include <vk.h>
#include <vulcan/buffer.h>
#include <vulcan/matrix.h>

namespace vulcan
{

struct Frame;

class Tracker
{
  public:

    Tracker();

    virtual ~Tracker();

    std::shared_ptr<const Frame> GetKeyframe() const;

    void SetKeyframe(std::shared_ptr<const Frame> keyframe);

    bool GetTranslationEnabled() const;

    void SetTranslationEnabled(bool enabled);

    int GetMaxIterations() const;

    void SetMaxIterations(int iterations);

    void Track(Frame& frame);

    // Called between ComputeSystem and the solve with the packed device system
    // (48 floats: hessian[36], gradient[6], pad): a multi-GPU rig all-reduces it
    // here.
    typedef void (*ReduceHook)(float* system_device, int count, void* user);

    void SetReduceHook(ReduceHook hook, void* user);

  protected:

    bool IsSolving() const;

    virtual void TrackOnDevice(Frame& frame);

    virtual void BeginSolve(const Frame& frame);

    void ValidateKeyframe() const;

    void ValidateFrame(const Frame& frame) const;

    virtual void ComputeSystem(const Frame& frame) = 0;

    virtual void ApplyUpdate(Frame& frame, const Vector6f& x) const = 0;

    void EndSolve(Frame& frame);

    void ResizeBuffers(const Frame& frame);

    virtual int GetResidualCount(const Frame& frame) const = 0;

    int GetParameterCount() const;

  protected:

    bool translation_enabled_;

    int iteration_;

    int max_iterations_;

    std::shared_ptr<const Frame> keyframe_;

    Buffer<float> system_;     // hessian_ = system_[0..36), gradient_ = system_[36..42)

    Buffer<float> workspace_;

    Buffer<vk_transform> pose_;

    Buffer<int> state_;        // {iterations run, converged}

    Buffer<float> update_;

    ReduceHook reduce_hook_;

    void* reduce_user_;
};

} // namespace vulcan
