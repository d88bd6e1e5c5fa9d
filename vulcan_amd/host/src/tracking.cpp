// tracking.cpp — Tracker, DepthTracker, ColorTracker, LightTracker and PyramidTracker<T>
// (ref: src/tracker.cpp, src/depth_tracker.cpp, src/depth_tracker.cu:272-378,
//  src/color_tracker.cpp, src/color_tracker.cu:296-470, src/pyramid_tracker.cpp).
#include <vulcan/tracking.h>
#include <vulcan/exception.h>
#include <vulcan/observation.h>
#include <vulcan/tsdf_volume.h>

namespace vulcan
{

namespace
{

vk_icp_view ViewOf(const Frame& frame)
{
  vk_icp_view v;
  v.depths = frame.depth_image->GetData();
  v.normals = reinterpret_cast<const float*>(frame.normal_image->GetData());
  v.width = frame.depth_image->GetWidth();
  v.height = frame.depth_image->GetHeight();
  v.projection = frame.depth_projection.ToVk();
  return v;
}

} // namespace

// ---- Tracker -------------------------------------------------------------------

namespace
{

struct HookAdapter { Tracker::ReduceHook hook; void* user; };

int CallReduceHook(float* system_dev, int count, void* user, void*)
{
  const HookAdapter* a = static_cast<const HookAdapter*>(user);
  a->hook(system_dev, count, a->user);
  return 0;
}

} // namespace

Tracker::Tracker() :
  translation_enabled_(true),
  iteration_(0),
  max_iterations_(20),
  reduce_hook_(nullptr),
  reduce_user_(nullptr)
{
  system_.Resize(48);
  pose_.Resize(1);
  state_.Resize(2);
  update_.Resize(6);
  poll_.host_state = nullptr;
  poll_.chunk = 4;
  void* pinned = nullptr;
  VK_ASSERT(vk_malloc_host(&pinned, 4 * sizeof(int32_t)));
  poll_.host_state = static_cast<int32_t*>(pinned);
  for (int i = 0; i < 4; ++i) poll_.host_state[i] = 0;
  poll_.host_pose = nullptr;
  VK_ASSERT(vk_malloc_host(&pinned, sizeof(vk_transform)));
  poll_.host_pose = static_cast<vk_transform*>(pinned);
}

Tracker::~Tracker()
{
  // a loop kernel that is still running writes both pinned blocks: drain the stream first
  vk_stream_synchronize(Device::GetStream());
  vk_free_host(poll_.host_pose);
  vk_free_host(poll_.host_state);
}

std::shared_ptr<const Frame> Tracker::GetKeyframe() const { return keyframe_; }

void Tracker::SetKeyframe(std::shared_ptr<const Frame> keyframe) { keyframe_ = keyframe; }

bool Tracker::GetTranslationEnabled() const { return translation_enabled_; }

void Tracker::SetTranslationEnabled(bool enabled) { translation_enabled_ = enabled; }

int Tracker::GetMaxIterations() const { return max_iterations_; }

void Tracker::SetMaxIterations(int iterations)
{
  VULCAN_DEBUG(iterations > 0);
  max_iterations_ = iterations;
}

void Tracker::SetReduceHook(ReduceHook hook, void* user)
{
  reduce_hook_ = hook;
  reduce_user_ = user;
}

// ref: tracker.cpp:53-63. Every iteration is enqueued without a host round
// trip: system -> (optional all-reduce) -> device solve + pose update. Once the
// update norm drops below 1e-6 (tracker.cpp:162) the device-side state marks
// the solve converged and the remaining updates are no-ops.
void Tracker::Track(Frame& frame)
{
  const Transform start = frame.depth_to_world_transform;
  for (int attempt = 0; attempt < 2; ++attempt)
  {
    BeginSolve(frame);
    TrackOnDevice(frame);
    iteration_ = max_iterations_;
    if (FinishSolve(frame, attempt == 1)) break;
    // the one-launch loop could not get its workgroups onto the device together
    // (VK_TRACK_ABORTED): once more from the start pose, one launch per stage
    frame.depth_to_world_transform = start;
    staged_only_ = true;
  }
  staged_only_ = false;
}

// default: iterate ComputeSystem + solve; DepthTracker enqueues the whole loop with one C call
void Tracker::TrackOnDevice(Frame& frame)
{
  while (IsSolving())
  {
    ComputeSystem(frame);
    if (reduce_hook_) reduce_hook_(system_.GetData(), 48, reduce_user_);
    VK_ASSERT(vk_icp_solve_update(system_.GetData(), system_.GetData() + 36, translation_enabled_ ? 1 : 0,
        pose_.GetData(), state_.GetData(), update_.GetData(), Device::GetStream()));
    ++iteration_;
  }
}

bool Tracker::IsSolving() const { return iteration_ < max_iterations_; }

void Tracker::BeginSolve(const Frame& frame)
{
  ValidateKeyframe();
  ValidateFrame(frame);
  ResizeBuffers(frame);
  iteration_ = 0;
  const vk_transform pose = frame.depth_to_world_transform.ToVk();
  poll_.host_state[3] = 0;   // no pose of this solve has arrived yet (tags are never 0)
  VK_ASSERT(vk_transform_upload(DevicePose(), &pose, Device::GetStream()));
  VK_ASSERT(vk_memset(state_.GetData(), 0, 2 * sizeof(int), Device::GetStream()));
}

void Tracker::EndSolve(Frame& frame) { FinishSolve(frame, true); }

// Tracker::EndSolve (tracker.cpp:78-82) that can say "aborted" instead of throwing: false when
// the device loop ended with VK_TRACK_ABORTED and `must_succeed` is not set (the frame's pose is
// then left alone)
bool Tracker::FinishSolve(Frame& frame, bool must_succeed)
{
  // the device loops leave the pose in pinned memory (vk_track_wait); a loop that does not
  // (the step-by-step default of this class) is read back with a copy
  vk_transform pose;
  if (vk_track_wait(&poll_, Device::GetStream()) == VK_OK) pose = *poll_.host_pose;
  else
  {
    VK_ASSERT(vk_memcpy_d2h(&pose, DevicePose(), sizeof(pose), Device::GetStream()));
    int32_t state[2] = {0, 0};
    VK_ASSERT(vk_memcpy_d2h(state, state_.GetData(), sizeof(state), Device::GetStream()));
    if (state[1] == VK_TRACK_ABORTED)
    {
      VULCAN_ASSERT_MSG(!must_succeed,
          "the tracking kernel could not get all of its workgroups onto the device (is another process using it?)");
      return false;
    }
  }
  frame.depth_to_world_transform = Transform::FromVk(pose);
  return true;
}

// the hook handed to the C ABI: the user's, or — while falling back from an aborted one-launch
// loop — one that changes nothing and thereby selects the launch-per-stage loop
vk_icp_reduce_fn Tracker::DeviceHook() const
{
  return reduce_hook_ ? CallReduceHook : (staged_only_ ? vk_reduce_nothing : nullptr);
}

void Tracker::ValidateKeyframe() const
{
  VULCAN_DEBUG_MSG(keyframe_, "keyframe has not been assigned");
  VULCAN_DEBUG_MSG(keyframe_->depth_image, "keyframe missing depth image");
  VULCAN_DEBUG_MSG(keyframe_->normal_image, "keyframe missing normal image");
  VULCAN_DEBUG_MSG(keyframe_->depth_image->GetTotal() > 0, "invalid keyframe depth image size");
  VULCAN_DEBUG_MSG(keyframe_->depth_image->GetSize() == keyframe_->normal_image->GetSize(),
      "keyframe image size mismatch");
}

void Tracker::ValidateFrame(const Frame& frame) const
{
  VULCAN_DEBUG_MSG(frame.depth_image, "frame missing depth image");
  VULCAN_DEBUG_MSG(frame.normal_image, "frame missing normal image");
  VULCAN_DEBUG_MSG(frame.depth_image->GetTotal() > 0, "invalid frame depth image size");
  VULCAN_DEBUG_MSG(frame.depth_image->GetSize() == frame.normal_image->GetSize(),
      "frame image size mismatch");
}

void Tracker::ResizeBuffers(const Frame& frame)
{
  const size_t floats = vk_icp_workspace_floats(frame.depth_image->GetWidth(), frame.depth_image->GetHeight());
  if (floats > workspace_.GetSize())
  {
    // zeroed: memory that comes back from an allocator may hold anything, and the library — which clears a workspace only
    // when it is new to it, old, grown or overwritten by a staged loop (vk.h) — cannot see that
    workspace_.Resize(floats);
    VK_ASSERT(vk_memset(workspace_.GetData(), 0, floats * sizeof(float), Device::GetStream()));
  }
}

int Tracker::GetParameterCount() const { return translation_enabled_ ? 6 : 3; }

// ---- DepthTracker ----------------------------------------------------------------

DepthTracker::DepthTracker() {}

DepthTracker::~DepthTracker() {}

int DepthTracker::GetResidualCount(const Frame& frame) const
{
  return frame.depth_image->GetWidth() * frame.depth_image->GetHeight();
}

void DepthTracker::ComputeResiduals(const Frame& frame, Buffer<float>& residuals) const
{
  residuals.Resize(GetResidualCount(frame));
  const vk_icp_view key = ViewOf(*keyframe_), frm = ViewOf(frame);
  const vk_transform Twm = keyframe_->depth_to_world_transform.ToVk();
  const vk_transform Twc = frame.depth_to_world_transform.ToVk();
  VK_ASSERT(vk_icp_compute_residuals(&key, &Twm, &frm, &Twc, residuals.GetData(), Device::GetStream()));
}

void DepthTracker::ComputeJacobian(const Frame& frame, Buffer<Vector6f>& jacobian) const
{
  jacobian.Resize(GetResidualCount(frame));
  const vk_icp_view key = ViewOf(*keyframe_), frm = ViewOf(frame);
  const vk_transform Twm = keyframe_->depth_to_world_transform.ToVk();
  const vk_transform Twc = frame.depth_to_world_transform.ToVk();
  VK_ASSERT(vk_icp_compute_jacobian(&key, &Twm, &frm, &Twc, translation_enabled_ ? 1 : 0,
      reinterpret_cast<float*>(jacobian.GetData()), Device::GetStream()));
}


void DepthTracker::TrackOnDevice(Frame& frame)
{
  const vk_icp_view key = ViewOf(*keyframe_), frm = ViewOf(frame);
  const vk_transform Twm = keyframe_->depth_to_world_transform.ToVk();
  HookAdapter adapter = { reduce_hook_, reduce_user_ };
  VK_ASSERT(vk_icp_track(&key, &Twm, &frm, pose_.GetData(), max_iterations_, translation_enabled_ ? 1 : 0,
      workspace_.GetData(), system_.GetData(), state_.GetData(), update_.GetData(),
      DeviceHook(), &adapter, &poll_, Device::GetStream()));
}

void DepthTracker::TrackPyramid(std::shared_ptr<const Frame> keyframe, Frame& frame, bool normals_due, bool keyframe_normals_due,
    Volume* set_view_of, int set_view_rounds, bool* set_view_done)
{
  if (set_view_done) *set_view_done = false;
  if (normals_due)
  {
    VULCAN_ASSERT_MSG(frame.depth_image, "missing depth image");
    if (!frame.normal_image) frame.normal_image = std::make_shared<ColorImage>();
    frame.normal_image->Resize(frame.depth_image->GetWidth(), frame.depth_image->GetHeight());
    frame.normal_image->GetData();                 // (stamps the image: new content)
  }
  keyframe_ = keyframe;
  max_iterations_ = 20;          // what pyramid_tracker.cpp:85-86 leaves behind
  translation_enabled_ = true;
  ValidateKeyframe();
  ValidateFrame(frame);
  ResizeBuffers(frame);
  const vk_icp_view key = ViewOf(*keyframe_), frm = ViewOf(frame);
  const size_t floats = vk_icp_pyramid_floats(key.width, key.height, frm.width, frm.height);
  VULCAN_ASSERT_MSG(floats > 0, "pyramid tracking needs even image sizes");
  if (floats > pyramid_.GetSize()) pyramid_.Resize(floats);
  const vk_transform Twm = keyframe_->depth_to_world_transform.ToVk();
  const Transform start = frame.depth_to_world_transform;
  for (int attempt = 0; attempt < 2; ++attempt)
  {
    const vk_transform pose = frame.depth_to_world_transform.ToVk();
    poll_.host_state[3] = 0;
    HookAdapter adapter = { reduce_hook_, reduce_user_ };
    // the start pose and (first attempt) the frame's normals travel with the pyramid's launch
    // (bit 0: the frame's normal image, bit 1: the keyframe's — vk.h)
    const int due = attempt == 0 ? ((normals_due ? 1 : 0) | (keyframe_normals_due ? 2 : 0)) : 0;
    VK_ASSERT(vk_icp_pyramid_track_frame(&key, &Twm, &frm, pose_.GetData(), &pose, due,
        pyramid_.GetData(), workspace_.GetData(), system_.GetData(), state_.GetData(), update_.GetData(), DeviceHook(), &adapter,
        &poll_, Device::GetStream()));
    iteration_ = max_iterations_;
    // Volume::SetView at the pose this Track leaves on the device, enqueued before the host waits for it (first attempt only;
    // a reduce hook's loop is enqueued in chunks and is not covered)
    bool early = false;
    if (attempt == 0 && set_view_of && !reduce_hook_) early = set_view_of->SetViewAtDevicePose(frame, pose_.GetData(), set_view_rounds);
    if (FinishSolve(frame, attempt == 1))
    {
      if (set_view_done) *set_view_done = early;
      break;
    }
    // aborted: again, one launch per stage (Tracker::Track). An early SetView has then run at the START pose (the aborted
    // loop leaves it on the device): a state upstream reaches; the caller's own SetView follows the repeated Track.
    frame.depth_to_world_transform = start;
    staged_only_ = true;
  }
  staged_only_ = false;
}

void DepthTracker::ComputeSystem(const Frame& frame)
{
  const vk_icp_view key = ViewOf(*keyframe_), frm = ViewOf(frame);
  const vk_transform Twm = keyframe_->depth_to_world_transform.ToVk();
  const vk_transform Twc = frame.depth_to_world_transform.ToVk();
  // the pose being refined lives on the device (pose_), Twc is only the fallback
  VK_ASSERT(vk_icp_compute_system(&key, &Twm, &frm, &Twc, pose_.GetData(), translation_enabled_ ? 1 : 0,
      workspace_.GetData(), system_.GetData(), system_.GetData() + 36, Device::GetStream()));
}

// ref: depth_tracker.cpp:22-86 (host form; Track() uses the device form)
void DepthTracker::ApplyUpdate(Frame& frame, const Vector6f& update) const
{
  Matrix4f Tinc = Matrix4f::Identity();
  Tinc(0, 1) = -update[2]; Tinc(0, 2) = +update[1]; Tinc(0, 3) = +update[3];
  Tinc(1, 0) = +update[2]; Tinc(1, 2) = +update[0]; Tinc(1, 3) = +update[4];   // (1,2) sign as upstream
  Tinc(2, 0) = -update[1]; Tinc(2, 1) = +update[0]; Tinc(2, 3) = +update[5];

  const Matrix4f M = Tinc * frame.depth_to_world_transform.GetMatrix();

  Vector3f x_axis(M(0, 0), M(1, 0), M(2, 0));
  Vector3f y_axis(M(0, 1), M(1, 1), M(2, 1));
  x_axis.Normalize();
  y_axis.Normalize();
  const Vector3f z_axis = x_axis.Cross(y_axis);
  y_axis = z_axis.Cross(x_axis);

  Matrix3f R;
  for (int r = 0; r < 3; ++r)
  {
    R(r, 0) = x_axis[r];
    R(r, 1) = y_axis[r];
    R(r, 2) = z_axis[r];
  }

  const Vector3f t(M(0, 3), M(1, 3), M(2, 3));
  frame.depth_to_world_transform = Transform::Translate(t) * Transform::Rotate(R);
}

// ---- ColorTracker ----------------------------------------------------------------

ColorTracker::ColorTracker() { color_pose_.Resize(1); }

ColorTracker::~ColorTracker() {}

void ColorTracker::ComputeKeyframeIntensities() { keyframe_->color_image->ConvertTo(keyframe_intensities_); }

void ColorTracker::ComputeFrameIntensities(const Frame& frame) { frame.color_image->ConvertTo(frame_intensities_); }

void ColorTracker::ComputeFrameGradients(const Frame& frame)
{
  frame_intensities_.GetGradients(frame_gradient_x_, frame_gradient_y_);
}

// ref: color_tracker.cpp:19-25
void ColorTracker::BeginSolve(const Frame& frame)
{
  BeginOnDevice(frame, nullptr);
}

// Tracker::BeginSolve + the image passes of ColorTracker / LightTracker::BeginSolve as one launch
// (vk_color_tracker_begin); `mask`: the light tracker's frame mask, or null
void ColorTracker::BeginOnDevice(const Frame& frame, Image* mask, bool upload_pose)
{
  ValidateKeyframe();
  ValidateFrame(frame);
  ResizeBuffers(frame);
  iteration_ = 0;
  poll_.host_state[3] = 0;   // no pose of this solve has arrived yet
  VULCAN_ASSERT_MSG(keyframe_->color_image && frame.color_image, "missing color image");
  keyframe_intensities_.Resize(keyframe_->color_image->GetSize());
  frame_intensities_.Resize(frame.color_image->GetSize());
  frame_gradient_x_.Resize(frame.color_image->GetSize());
  frame_gradient_y_.Resize(frame.color_image->GetSize());
  if (mask) mask->Resize(frame.depth_image->GetWidth(), frame.depth_image->GetHeight());
  const vk_frame key = keyframe_->ToVk(), frm = frame.ToVk();
  const vk_transform pose = frame.depth_to_world_transform.ToVk();
  VK_ASSERT(vk_color_tracker_begin(&key, &frm, keyframe_intensities_.GetData(), frame_intensities_.GetData(),
      frame_gradient_x_.GetData(), frame_gradient_y_.GetData(), MaskThreshold(), mask ? mask->GetData() : nullptr,
      upload_pose ? &pose : nullptr, upload_pose ? color_pose_.GetData() : nullptr, state_.GetData(), Device::GetStream()));
  // the residuals are per KEYFRAME pixel (color_tracker.cpp:27-32)
  const size_t floats = vk_icp_workspace_floats(keyframe_->depth_image->GetWidth(), keyframe_->depth_image->GetHeight());
  if (floats > workspace_.GetSize())
  {
    // zeroed: memory that comes back from an allocator may hold anything, and the library — which clears a workspace only
    // when it is new to it, old, grown or overwritten by a staged loop (vk.h) — cannot see that
    workspace_.Resize(floats);
    VK_ASSERT(vk_memset(workspace_.GetData(), 0, floats * sizeof(float), Device::GetStream()));
  }
}

void ColorTracker::TrackCoarseToFine(std::shared_ptr<const Frame> half_keyframe, Frame& half_frame,
    std::shared_ptr<const Frame> keyframe, Frame& frame)
{
  const Transform start = half_frame.depth_to_world_transform;
  for (int attempt = 0; attempt < 2; ++attempt)
  {
    // :79-83 half level, 15 steps, from the frame's pose
    SetMaxIterations(15);
    SetTranslationEnabled(true);
    SetKeyframe(half_keyframe);
    BeginOnDevice(half_frame, MaskImage(), true);
    TrackOnDevice(half_frame);
    // :85-89 full level, 20 steps, from the pose the half level left in color_pose_ (upstream
    // carries it through half_frame.depth_to_world_transform: the same bits)
    SetMaxIterations(20);
    SetKeyframe(keyframe);
    BeginOnDevice(frame, MaskImage(), false);
    TrackOnDevice(frame);
    iteration_ = max_iterations_;
    if (FinishSolve(frame, attempt == 1)) break;
    half_frame.depth_to_world_transform = start;   // aborted: again, one launch per stage (Tracker::Track)
    staged_only_ = true;
  }
  staged_only_ = false;
  half_frame.depth_to_world_transform = frame.depth_to_world_transform;   // not the half level's own result: see above
}

int ColorTracker::GetResidualCount(const Frame&) const
{
  return keyframe_->depth_image->GetWidth() * keyframe_->depth_image->GetHeight();
}

vk_color_view ColorTracker::KeyframeView() const
{
  vk_color_view v;
  v.depths = keyframe_->depth_image->GetData();
  v.normals = reinterpret_cast<const float*>(keyframe_->normal_image->GetData());
  v.intensities = keyframe_intensities_.GetData();
  v.gradient_x = nullptr;
  v.gradient_y = nullptr;
  v.width = keyframe_->depth_image->GetWidth();
  v.height = keyframe_->depth_image->GetHeight();
  v.projection = keyframe_->color_projection.ToVk();
  return v;
}

vk_color_view ColorTracker::FrameView(const Frame& frame) const
{
  vk_color_view v;
  v.depths = frame.depth_image->GetData();
  v.normals = reinterpret_cast<const float*>(frame.normal_image->GetData());
  v.intensities = frame_intensities_.GetData();
  v.gradient_x = frame_gradient_x_.GetData();
  v.gradient_y = frame_gradient_y_.GetData();
  v.width = frame.depth_image->GetWidth();
  v.height = frame.depth_image->GetHeight();
  v.projection = frame.color_projection.ToVk();
  return v;
}

vk_transform ColorTracker::GetKeyframeTwc() const
{
  const Transform Tcw = keyframe_->depth_to_color_transform * keyframe_->depth_to_world_transform.Inverse();
  return Tcw.Inverse().ToVk();
}

vk_transform ColorTracker::GetTcm(const Frame& frame) const
{
  const Transform keyframe_Tcw = keyframe_->depth_to_color_transform * keyframe_->depth_to_world_transform.Inverse();
  const Transform frame_Tcw = frame.depth_to_color_transform * frame.depth_to_world_transform.Inverse();
  return (frame_Tcw * keyframe_Tcw.Inverse()).ToVk();
}

void ColorTracker::ComputeResiduals(const Frame& frame, Buffer<float>& residuals)
{
  ComputeKeyframeIntensities();
  ComputeFrameIntensities(frame);
  residuals.Resize(GetResidualCount(frame));
  const vk_color_view key = KeyframeView(), frm = FrameView(frame);
  const vk_transform Tcm = GetTcm(frame);
  VK_ASSERT(vk_color_tracker_compute_residuals(&key, &frm, &Tcm, residuals.GetData(), Device::GetStream()));
}

void ColorTracker::ComputeJacobian(const Frame& frame, Buffer<Vector6f>& jacobian)
{
  ComputeKeyframeIntensities();
  ComputeFrameIntensities(frame);
  ComputeFrameGradients(frame);
  jacobian.Resize(GetResidualCount(frame));
  const vk_color_view key = KeyframeView(), frm = FrameView(frame);
  const vk_transform Tcm = GetTcm(frame);
  VK_ASSERT(vk_color_tracker_compute_jacobian(&key, &frm, &Tcm, translation_enabled_ ? 1 : 0,
      reinterpret_cast<float*>(jacobian.GetData()), Device::GetStream()));
}

void ColorTracker::ComputeSystem(const Frame& frame)
{
  const vk_color_view key = KeyframeView(), frm = FrameView(frame);
  const vk_transform Tcm = GetTcm(frame);
  VK_ASSERT(vk_color_tracker_compute_system(&key, &frm, &Tcm, nullptr, translation_enabled_ ? 1 : 0,
      workspace_.GetData(), system_.GetData(), system_.GetData() + 36, Device::GetStream()));
}

void ColorTracker::TrackOnDevice(Frame& frame)
{
  const vk_color_view key = KeyframeView(), frm = FrameView(frame);
  const vk_transform frame_Tcd = frame.depth_to_color_transform.ToVk();
  const vk_transform key_Twc = GetKeyframeTwc();
  // BeginSolve put the pose into color_pose_->depth_to_world (DevicePose)
  HookAdapter adapter = { reduce_hook_, reduce_user_ };
  VK_ASSERT(vk_color_tracker_track(&key, &frm, &frame_Tcd, &key_Twc, color_pose_.GetData(), max_iterations_,
      translation_enabled_ ? 1 : 0, workspace_.GetData(), system_.GetData(), state_.GetData(), update_.GetData(),
      DeviceHook(), &adapter, &poll_, Device::GetStream()));
}

// ref: color_tracker.cpp:34-96 (host form; Track() uses the device form)
void ColorTracker::ApplyUpdate(Frame& frame, const Vector6f& update) const
{
  Matrix4f Tinc = Matrix4f::Identity();
  Tinc(0, 1) = -update[2]; Tinc(0, 2) = +update[1]; Tinc(0, 3) = +update[3];
  Tinc(1, 0) = +update[2]; Tinc(1, 2) = -update[0]; Tinc(1, 3) = +update[4];
  Tinc(2, 0) = -update[1]; Tinc(2, 1) = +update[0]; Tinc(2, 3) = +update[5];

  const Matrix4f M = Tinc * frame.depth_to_world_transform.GetInverseMatrix();

  Vector3f x_axis(M(0, 0), M(1, 0), M(2, 0));
  Vector3f y_axis(M(0, 1), M(1, 1), M(2, 1));
  x_axis.Normalize();
  y_axis.Normalize();
  const Vector3f z_axis = x_axis.Cross(y_axis);
  y_axis = z_axis.Cross(x_axis);

  Matrix3f R;
  for (int r = 0; r < 3; ++r)
  {
    R(r, 0) = x_axis[r];
    R(r, 1) = y_axis[r];
    R(r, 2) = z_axis[r];
  }

  const Vector3f t(M(0, 3), M(1, 3), M(2, 3));
  frame.depth_to_world_transform = (Transform::Translate(t) * Transform::Rotate(R)).Inverse();
}

// ---- LightTracker ----------------------------------------------------------------

LightTracker::LightTracker() : depth_threshold_(0.2f) {}   // light_tracker.cpp:13-14

LightTracker::~LightTracker() {}

const Light& LightTracker::GetLight() const { return light_; }

void LightTracker::SetLight(const Light& light) { light_ = light; }

// ref: light_tracker.cu:548-564 (the kernel is the light integrator's)
void LightTracker::ComputeFrameMask(const Frame& frame)
{
  frame_mask_.Resize(frame.depth_image->GetWidth(), frame.depth_image->GetHeight());
  const vk_frame f = frame.ToVk();
  VK_ASSERT(vk_light_compute_frame_mask(&f, depth_threshold_, frame_mask_.GetData(), Device::GetStream()));
}

vk_light_terms LightTracker::GetTerms(const Frame& frame) const
{
  vk_light_terms t;
  t.frame_mask = frame_mask_.GetData();
  t.light = light_.ToVk();
  t.frame_Tcd = frame.depth_to_color_transform.ToVk();
  return t;
}

// ref: light_tracker.cpp:34-41
void LightTracker::BeginSolve(const Frame& frame)
{
  BeginOnDevice(frame, &frame_mask_);
}

float LightTracker::MaskThreshold() const { return depth_threshold_; }

// Upstream's ComputeResiduals / ComputeJacobian read frame_mask_ without computing
// it (light_tracker.cu:569-577,613-623); here they compute it first.
void LightTracker::ComputeResiduals(const Frame& frame, Buffer<float>& residuals)
{
  ComputeKeyframeIntensities();
  ComputeFrameIntensities(frame);
  ComputeFrameMask(frame);
  residuals.Resize(GetResidualCount(frame));
  const vk_color_view key = KeyframeView(), frm = FrameView(frame);
  const vk_light_terms terms = GetTerms(frame);
  const vk_transform Tcm = GetTcm(frame);
  VK_ASSERT(vk_light_tracker_compute_residuals(&key, &frm, &terms, &Tcm, residuals.GetData(), Device::GetStream()));
}

void LightTracker::ComputeJacobian(const Frame& frame, Buffer<Vector6f>& jacobian)
{
  ComputeKeyframeIntensities();
  ComputeFrameIntensities(frame);
  ComputeFrameGradients(frame);
  ComputeFrameMask(frame);
  jacobian.Resize(GetResidualCount(frame));
  const vk_color_view key = KeyframeView(), frm = FrameView(frame);
  const vk_light_terms terms = GetTerms(frame);
  const vk_transform Tcm = GetTcm(frame);
  VK_ASSERT(vk_light_tracker_compute_jacobian(&key, &frm, &terms, &Tcm, translation_enabled_ ? 1 : 0,
      reinterpret_cast<float*>(jacobian.GetData()), Device::GetStream()));
}

void LightTracker::ComputeSystem(const Frame& frame)
{
  const vk_color_view key = KeyframeView(), frm = FrameView(frame);
  const vk_light_terms terms = GetTerms(frame);
  const vk_transform Tcm = GetTcm(frame);
  VK_ASSERT(vk_light_tracker_compute_system(&key, &frm, &terms, &Tcm, nullptr, translation_enabled_ ? 1 : 0,
      workspace_.GetData(), system_.GetData(), system_.GetData() + 36, Device::GetStream()));
}

void LightTracker::TrackOnDevice(Frame& frame)
{
  const vk_color_view key = KeyframeView(), frm = FrameView(frame);
  const vk_light_terms terms = GetTerms(frame);
  const vk_transform key_Twc = GetKeyframeTwc();
  HookAdapter adapter = { reduce_hook_, reduce_user_ };
  VK_ASSERT(vk_light_tracker_track(&key, &frm, &terms, &key_Twc, color_pose_.GetData(), max_iterations_,
      translation_enabled_ ? 1 : 0, workspace_.GetData(), system_.GetData(), state_.GetData(), update_.GetData(),
      DeviceHook(), &adapter, &poll_, Device::GetStream()));
}

// ---- PyramidTracker ---------------------------------------------------------------

template <typename Tracker>
PyramidTracker<Tracker>::PyramidTracker() :
  tracker_(std::shared_ptr<Tracker>(new Tracker())),
  half_keyframe_(std::make_shared<Frame>()),
  half_frame_(std::make_shared<Frame>()),
  quarter_keyframe_(std::make_shared<Frame>()),
  iter_(0)
{
}

template <typename Tracker>
PyramidTracker<Tracker>::PyramidTracker(std::shared_ptr<Tracker> tracker) :
  tracker_(tracker),
  half_keyframe_(std::make_shared<Frame>()),
  half_frame_(std::make_shared<Frame>()),
  quarter_keyframe_(std::make_shared<Frame>()),
  iter_(0)
{
}

template <typename Tracker>
PyramidTracker<Tracker>::~PyramidTracker() {}

template <typename Tracker>
std::shared_ptr<const Tracker> PyramidTracker<Tracker>::GetTracker() const { return tracker_; }

template <typename Tracker>
std::shared_ptr<const Frame> PyramidTracker<Tracker>::GetKeyframe() const { return keyframe_; }

template <typename Tracker>
void PyramidTracker<Tracker>::SetKeyframe(std::shared_ptr<const Frame> keyframe) { keyframe_ = keyframe; }

// ref: pyramid_tracker.cpp:52-90. The quarter level is built but not tracked
// upstream (:69-77 commented out); it is not built here.
template <typename Tracker>
void PyramidTracker<Tracker>::Track(Frame& frame)
{
  TrackLevels(frame);
}

// the depth tracker's two levels are one call into the C ABI: nothing but the final pose
// crosses the host boundary
template <>
void PyramidTracker<DepthTracker>::Track(Frame& frame)
{
  VULCAN_DEBUG(keyframe_);
  tracker_->TrackPyramid(keyframe_, frame);
  ++iter_;
}

template <typename Tracker>
void PyramidTracker<Tracker>::ComputeNormalsAndTrack(Frame& frame, bool keyframe_normals_due)
{
  // (the key frame is shared as const: its normal IMAGE is what is written, through the image's own pointer)
  if (keyframe_normals_due && keyframe_ && keyframe_->normal_image && keyframe_->depth_image)
  {
    Frame key = *keyframe_;
    key.ComputeNormals();
  }
  frame.ComputeNormals();
  Track(frame);
}

template <>
void PyramidTracker<DepthTracker>::ComputeNormalsAndTrack(Frame& frame, bool keyframe_normals_due)
{
  VULCAN_DEBUG(keyframe_);
  tracker_->TrackPyramid(keyframe_, frame, true, keyframe_normals_due);
  ++iter_;
}

template <typename Tracker>
void PyramidTracker<Tracker>::ComputeNormalsTrackAndSetView(Frame& frame, Volume& volume, int rounds, bool keyframe_normals_due)
{
  ComputeNormalsAndTrack(frame, keyframe_normals_due);
  volume.SetView(frame, rounds);
}

template <>
void PyramidTracker<DepthTracker>::ComputeNormalsTrackAndSetView(Frame& frame, Volume& volume, int rounds, bool keyframe_normals_due)
{
  VULCAN_DEBUG(keyframe_);
  bool done = false;
  tracker_->TrackPyramid(keyframe_, frame, true, keyframe_normals_due, &volume, rounds, &done);
  ++iter_;
  if (!done) volume.SetView(frame, rounds);
}

// the photometric trackers run both levels without a host round trip in between
template <typename Tracker>
static void TrackPhotometricLevels(Tracker& tracker, std::shared_ptr<const Frame> keyframe,
    std::shared_ptr<Frame> half_keyframe, Frame& half_frame, Frame& frame)
{
  frame.Downsample(half_frame);
  keyframe->Downsample(*half_keyframe);
  tracker.TrackCoarseToFine(half_keyframe, half_frame, keyframe, frame);
}

template <>
void PyramidTracker<ColorTracker>::Track(Frame& frame)
{
  VULCAN_DEBUG(keyframe_);
  TrackPhotometricLevels(*tracker_, keyframe_, half_keyframe_, *half_frame_, frame);
  ++iter_;
}

template <>
void PyramidTracker<LightTracker>::Track(Frame& frame)
{
  VULCAN_DEBUG(keyframe_);
  TrackPhotometricLevels(*tracker_, keyframe_, half_keyframe_, *half_frame_, frame);
  ++iter_;
}

template <typename Tracker>
void PyramidTracker<Tracker>::TrackLevels(Frame& frame)
{
  VULCAN_DEBUG(keyframe_);

  Frame& half_frame = *half_frame_;
  frame.Downsample(half_frame);
  keyframe_->Downsample(*half_keyframe_);

  tracker_->SetMaxIterations(15);
  tracker_->SetTranslationEnabled(true);
  tracker_->SetKeyframe(half_keyframe_);
  tracker_->Track(half_frame);

  tracker_->SetMaxIterations(20);
  tracker_->SetTranslationEnabled(true);
  frame.depth_to_world_transform = half_frame.depth_to_world_transform;
  tracker_->SetKeyframe(keyframe_);
  tracker_->Track(frame);
  ++iter_;
}

template class PyramidTracker<DepthTracker>;
template class PyramidTracker<ColorTracker>;
template class PyramidTracker<LightTracker>;

} // namespace vulcan
