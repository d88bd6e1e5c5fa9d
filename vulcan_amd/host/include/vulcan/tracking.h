// tracking.h — frame-to-keyframe pose tracking: the Gauss-Newton base, the
// projective point-to-plane ICP and the coarse-to-fine wrapper.
//
// API parity: Tracker, DepthTracker and PyramidTracker<T> keep the names and
// methods of the reference's tracker.h, depth_tracker.h and pyramid_tracker.h
// (which remain as forwarders to this file). What differs underneath: the
// reference reads 42 floats back and solves with Eigen::LDLT on the host every
// iteration (src/tracker.cpp:124-163); here the 6x6 solve and the SE(3) update
// run on the device (vk_icp_solve_update / vk_icp_track), Track() enqueues every
// iteration and reads the pose back once, and ApplyUpdate therefore takes a
// Vector6f rather than an Eigen::VectorXf.
#pragma once

#include <memory>
#include <vk.h>
#include <vulcan/buffer.h>
#include <vulcan/image.h>
#include <vulcan/light.h>
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

    // refine frame.depth_to_world_transform against the keyframe
    void Track(Frame& frame);

    // Multi-GPU rigs: called between ComputeSystem and the solve with the packed
    // device system (48 floats: hessian[36], gradient[6], pad[6]); the hook
    // all-reduces it in place on the compute stream.
    typedef void (*ReduceHook)(float* system_device, int count, void* user);
    void SetReduceHook(ReduceHook hook, void* user);

  protected:
    // -- the reference's solve loop, stage by stage --
    bool IsSolving() const;
    virtual void BeginSolve(const Frame& frame);
    virtual void ComputeSystem(const Frame& frame) = 0;
    virtual void ApplyUpdate(Frame& frame, const Vector6f& x) const = 0;
    void EndSolve(Frame& frame);
    bool FinishSolve(Frame& frame, bool must_succeed);   // EndSolve that reports VK_TRACK_ABORTED as false
    vk_icp_reduce_fn DeviceHook() const;                  // the user's hook, or the no-op one while falling back

    // whole loop on the device; subclasses with a fused ABI entry override it
    virtual void TrackOnDevice(Frame& frame);

    void ValidateKeyframe() const;
    void ValidateFrame(const Frame& frame) const;
    void ResizeBuffers(const Frame& frame);
    virtual int GetResidualCount(const Frame& frame) const = 0;
    int GetParameterCount() const;

    bool translation_enabled_;
    int iteration_;
    int max_iterations_;
    std::shared_ptr<const Frame> keyframe_;

    Buffer<float> system_;        // hessian = [0,36), gradient = [36,42)
    Buffer<float> workspace_;     // per-workgroup partial sums
    Buffer<vk_transform> pose_;   // device copy of the pose being refined

    // where BeginSolve puts the pose and EndSolve reads it back from (a tracker whose device loop
    // keeps a larger pose record points this at the record's depth_to_world)
    virtual vk_transform* DevicePose() { return pose_.GetData(); }
    Buffer<int> state_;           // {iterations run, converged}
    Buffer<float> update_;        // last 6-vector

    ReduceHook reduce_hook_;
    void* reduce_user_;
    bool staged_only_ = false;    // falling back from an aborted one-launch loop: one launch per stage

    // early exit of the device loop (vk_track_poll): pinned {iterations, converged} mirror
    // and the number of steps enqueued between two looks at it (0 = enqueue all steps)
    vk_track_poll poll_;

  public:
    // Gauss-Newton steps the last Track ran at its last level (from the pinned mirror the
    // device loop writes; not upstream — its loop counter iteration_ has no getter)
    int GetIterationsRun() const { return poll_.host_state ? poll_.host_state[0] : 0; }
    int GetPollChunk() const { return poll_.chunk; }
    void SetPollChunk(int steps) { poll_.chunk = steps; }
};

// Projective point-to-plane ICP on depth + normals.
class DepthTracker : public Tracker
{
  public:
    DepthTracker();
    virtual ~DepthTracker();

    // per-pixel residuals / Jacobian rows, exposed for tests as upstream
    void ComputeResiduals(const Frame& frame, Buffer<float>& residuals) const;
    void ComputeJacobian(const Frame& frame, Buffer<Vector6f>& jacobian) const;

    // host form of the pose update (ref: depth_tracker.cpp:22-86)
    void ApplyUpdate(Frame& frame, const Vector6f& x) const override;

    // PyramidTracker<DepthTracker>::Track as one call into the C ABI (vk_icp_pyramid_track)
    // `normals_due`: frame.normal_image is still to be computed (Frame::ComputeNormals) and is, by the pyramid's launch
    // `keyframe_normals_due`: the KEYFRAME's normal image is still to be computed too (Tracer::TraceWithoutNormals left it
    // out) and is, by the same launch
    // `set_view_of` (round 6): Volume::SetView(frame, set_view_rounds) is enqueued BEHIND the Track at the pose the loop leaves
    // on the device, before this call waits for that pose (Volume::SetViewAtDevicePose); *set_view_done says whether it was
    // (false: the caller calls SetView itself — also after a Track that aborted and was repeated stage by stage)
    void TrackPyramid(std::shared_ptr<const Frame> keyframe, Frame& frame, bool normals_due = false, bool keyframe_normals_due = false,
        class Volume* set_view_of = nullptr, int set_view_rounds = 1, bool* set_view_done = nullptr);

  protected:
    int GetResidualCount(const Frame& frame) const override;
    void ComputeSystem(const Frame& frame) override;
    void TrackOnDevice(Frame& frame) override;

    Buffer<float> pyramid_;       // half-resolution depth + normals of keyframe and frame
};

// Photometric tracking: one intensity residual per keyframe pixel, sampled
// bilinearly in the frame (ref: color_tracker.h, color_tracker.cu,
// color_tracker.cpp). Uses the COLOUR intrinsics and depth_to_color_transform of
// both frames, as upstream.
class ColorTracker : public Tracker
{
  public:
    ColorTracker();
    virtual ~ColorTracker();

    void ComputeResiduals(const Frame& frame, Buffer<float>& residuals);
    void ComputeJacobian(const Frame& frame, Buffer<Vector6f>& jacobian);

    // host form of the pose update (ref: color_tracker.cpp:34-96)
    void ApplyUpdate(Frame& frame, const Vector6f& x) const override;

    // PyramidTracker<ColorTracker / LightTracker>::Track (pyramid_tracker.cpp:79-89) with both
    // levels enqueued back to back: the full-resolution loop starts from the pose the half-
    // resolution loop left ON THE DEVICE, the host waits once, for the final pose (not upstream)
    void TrackCoarseToFine(std::shared_ptr<const Frame> half_keyframe, Frame& half_frame,
        std::shared_ptr<const Frame> keyframe, Frame& frame);

  protected:
    void BeginSolve(const Frame& frame) override;
    int GetResidualCount(const Frame& frame) const override;
    void ComputeSystem(const Frame& frame) override;
    void TrackOnDevice(Frame& frame) override;

    void ComputeKeyframeIntensities();
    void ComputeFrameIntensities(const Frame& frame);
    void ComputeFrameGradients(const Frame& frame);

    // BeginSolve of this class and of LightTracker as one launch (vk_color_tracker_begin); not upstream
    void BeginOnDevice(const Frame& frame, Image* mask, bool upload_pose = true);
    virtual float MaskThreshold() const { return 0.0f; }
    virtual Image* MaskImage() { return nullptr; }      // the light tracker's frame mask

    vk_color_view KeyframeView() const;
    vk_color_view FrameView(const Frame& frame) const;
    vk_transform GetTcm(const Frame& frame) const;        // color_tracker.cu:312-320
    vk_transform GetKeyframeTwc() const;

    Image keyframe_intensities_;
    Image frame_intensities_;
    Image frame_gradient_x_;
    Image frame_gradient_y_;
    Buffer<vk_color_pose> color_pose_;   // depth_to_world + derived Tcm, on the device

    vk_transform* DevicePose() override { return &color_pose_.GetData()->depth_to_world; }
};

// ColorTracker with a shading model: where the frame mask is set the residual is
// Ic - albedo * light.GetShading(Xcp, n), elsewhere the point-to-plane distance
// (ref: light_tracker.h, light_tracker.cu, light_tracker.cpp). The keyframe's
// colour image is read as albedo. Upstream's WriteDataFiles / WriteImage /
// TraceImage debugging output (light_tracker.cpp:116-178) is not provided.
class LightTracker : public ColorTracker
{
  public:
    LightTracker();
    virtual ~LightTracker();

    const Light& GetLight() const;
    void SetLight(const Light& light);

    void ComputeResiduals(const Frame& frame, Buffer<float>& residuals);
    void ComputeJacobian(const Frame& frame, Buffer<Vector6f>& jacobian);

  protected:
    void BeginSolve(const Frame& frame) override;
    void ComputeSystem(const Frame& frame) override;
    void TrackOnDevice(Frame& frame) override;
    void ComputeFrameMask(const Frame& frame);
    vk_light_terms GetTerms(const Frame& frame) const;

    Light light_;
    float MaskThreshold() const override;
    Image* MaskImage() override { return &frame_mask_; }

    Image frame_mask_;
    float depth_threshold_;
};

// Coarse-to-fine: half resolution first (15 iterations), then full (20).
template <typename Tracker>
class PyramidTracker
{
  public:
    PyramidTracker();
    explicit PyramidTracker(std::shared_ptr<Tracker> tracker);
    virtual ~PyramidTracker();

    std::shared_ptr<const Tracker> GetTracker() const;
    std::shared_ptr<const Frame> GetKeyframe() const;
    void SetKeyframe(std::shared_ptr<const Frame> keyframe);

    void Track(Frame& frame);
    // frame.ComputeNormals(); Track(frame); (vulcan.cu:297-311) in one call — not upstream: with a DepthTracker the normal
    // image is computed by the launch that builds the pyramid (vk_icp_pyramid_track_frame), same bits
    // `keyframe_normals_due` (DepthTracker): the keyframe came from Tracer::TraceWithoutNormals — its normal image is computed by
    // the same launch as well (other trackers: keyframe normals are computed first, by a launch of their own)
    void ComputeNormalsAndTrack(Frame& frame, bool keyframe_normals_due = false);
    // frame.ComputeNormals(); Track(frame); volume.SetView(frame, rounds) — the head of the reference's frame loop
    // (vulcan.cu:297-318) — in one call, not upstream (round 6): with a DepthTracker SetView is enqueued BEHIND the Track at the
    // pose the loop leaves on the device, before the host waits for that pose, so the device does not idle for the host's
    // round trip (Volume::SetViewAtDevicePose; ~5 us of a 280 us frame, profiles/r06_tracked_frame_timeline.txt). The same
    // state as the three calls, bit for bit. Other trackers, and a volume that cannot take the call: the three calls.
    void ComputeNormalsTrackAndSetView(Frame& frame, class Volume& volume, int rounds = 1, bool keyframe_normals_due = false);

  protected:
    void TrackLevels(Frame& frame);   // the generic two-level loop (pyramid_tracker.cpp:52-90)

    std::shared_ptr<Tracker> tracker_;
    std::shared_ptr<const Frame> keyframe_;
    std::shared_ptr<Frame> half_keyframe_;
    std::shared_ptr<Frame> half_frame_;      // upstream: a local of Track (pyramid_tracker.cpp:58); kept so its images are reused
    std::shared_ptr<Frame> quarter_keyframe_;
    int iter_;
};

} // namespace vulcan
