// tsdf_volume.cpp — Volume host class over vk_volume_* (ref: src/volume.cu:370-627).
#include <vulcan/tsdf_volume.h>
#include <cstdio>
#include <cstring>
#include <vulcan/block.h>
#include <vulcan/exception.h>
#include <vulcan/observation.h>
#include <vulcan/hash.h>
#include <vulcan/voxel.h>

namespace vulcan
{

Volume::Volume(int main_block_count, int excess_block_count) :
  depth_range_(0.1f, 5.0f),
  max_block_count_(main_block_count + excess_block_count),
  main_block_count_(main_block_count),
  excess_block_count_(excess_block_count),
  truncation_length_(0.04f),
  voxel_length_(0.008f),
  empty_(true),
  visible_count_stale_(false),
  request_stream_(nullptr),
  requested_(nullptr),
  integrated_(nullptr),
  integrated_recorded_(false),
  normals_late_(nullptr)
{
  std::memset(&view_bounds_, 0, sizeof(view_bounds_));
  std::memset(&light_prep_, 0, sizeof(light_prep_));
  std::memset(&requests_ahead_, 0, sizeof(requests_ahead_));
  Initialize();
}

Volume::~Volume()
{
#ifndef NDEBUG
  // a pool that ran dry is said once before the volume goes (debug builds; a blocking readback is harmless here)
  if (!empty_ && !pool_exhaustion_noted_ && counters_.GetData())
  {
    int32_t counters[VK_CTR_PUBLIC];
    const vk_volume v = ToVk();
    if (vk_volume_read_counters_sync(&v, counters, Device::GetStream()) == VK_OK) NotePoolExhaustion(counters[VK_CTR_DROPPED]);
  }
#endif
  if (normals_late_) (void)vk_free_host(normals_late_);
  if (request_stream_)
  {
    (void)vk_stream_synchronize(request_stream_);
    (void)vk_event_destroy(requested_);
    (void)vk_event_destroy(integrated_);
    (void)vk_stream_destroy(request_stream_);
  }
}

int Volume::GetMainBlockCount() const { return main_block_count_; }

int Volume::GetExcessBlockCount() const { return excess_block_count_; }

const Vector2f& Volume::GetDepthRange() const { return depth_range_; }

void Volume::SetDepthRange(const Vector2f& range)
{
  VULCAN_DEBUG(range[0] > 0 && range[0] < range[1]);
  depth_range_ = range;
}

void Volume::SetDepthRange(float min, float max) { SetDepthRange(Vector2f(min, max)); }

float Volume::GetVoxelLength() const { return voxel_length_; }

void Volume::SetVoxelLength(float length)
{
  VULCAN_DEBUG(length > 0);
  voxel_length_ = length;
}

float Volume::GetTruncationLength() const { return truncation_length_; }

void Volume::SetTruncationLength(float length)
{
  VULCAN_DEBUG(length > 0);
  truncation_length_ = length;
}

vk_volume Volume::ToVk() const
{
  vk_volume v;
  v.voxels = reinterpret_cast<vk_voxel*>(const_cast<Voxel*>(voxels_.GetData()));
  v.hash_entries = reinterpret_cast<vk_hash_entry*>(const_cast<HashEntry*>(hash_entries_.GetData()));
  v.free_voxel_blocks = const_cast<int*>(free_voxel_blocks_.GetData());
  v.allocation_types = reinterpret_cast<uint8_t*>(const_cast<AllocationType*>(allocation_types_.GetData()));
  v.allocation_blocks = reinterpret_cast<vk_block*>(const_cast<Block*>(allocation_blocks_.GetData()));
  v.block_visibility = reinterpret_cast<uint8_t*>(const_cast<Visibility*>(block_visibility_.GetData()));
  v.visible_blocks = const_cast<int*>(visible_blocks_.GetData());
  v.counters = const_cast<int*>(counters_.GetData());
  v.main_block_count = main_block_count_;
  v.excess_block_count = excess_block_count_;
  v.voxel_length = voxel_length_;
  v.truncation_length = truncation_length_;
  v.min_depth = depth_range_[0];
  v.max_depth = depth_range_[1];
  return v;
}

vk_view_bounds* Volume::GetViewBounds() const { return view_bounds_.scratch ? &view_bounds_ : nullptr; }

void Volume::AttachViewBounds(float* scratch, int bounds_width, int bounds_height, const Vector2f& depth_range) const
{
  std::memset(&view_bounds_, 0, sizeof(view_bounds_));
  // the pinned word a normals workgroup of Tracer::Trace(keyframe, next_frame) sets when its bounded wait expires
  // (vk.h vk_view_bounds.late_host): the next Trace, or Tracer::SettleNormals, then throws instead of leaving wrong normals
  if (!normals_late_)
  {
    void* word = nullptr;
    VK_ASSERT(vk_malloc_host(&word, sizeof(int32_t)));
    normals_late_ = static_cast<int32_t*>(word);
  }
  *normals_late_ = 0;
  view_bounds_.late_host = normals_late_;
  view_bounds_.scratch = scratch;
  view_bounds_.bounds_width = bounds_width;
  view_bounds_.bounds_height = bounds_height;
  view_bounds_.min_depth = depth_range[0];
  view_bounds_.max_depth = depth_range[1];
}

void Volume::DetachViewBounds(const float* scratch) const
{
  if (view_bounds_.scratch == scratch) std::memset(&view_bounds_, 0, sizeof(view_bounds_));
}

void Volume::CancelRequestsAhead(int rounds)
{
  if (requests_ahead_.valid != 1) return;
  view_bounds_.valid = 0;
  const vk_volume v = ToVk();
  VK_ASSERT(vk_requests_ahead_cancel(&v, &requests_ahead_, rounds, Device::GetStream()));
  visible_count_stale_ = true;
  empty_ = false;
}

vk_light_prep* Volume::GetLightPreparation() const { return light_prep_.mask ? &light_prep_ : nullptr; }

void Volume::AttachLightPreparation(float* mask, float* records, int capacity_pixels, float depth_threshold) const
{
  std::memset(&light_prep_, 0, sizeof(light_prep_));
  light_prep_.mask = mask;
  light_prep_.records = records;
  light_prep_.capacity = capacity_pixels;
  light_prep_.depth_threshold = depth_threshold;
}

void Volume::DetachLightPreparation(const float* mask) const
{
  if (light_prep_.mask == mask) std::memset(&light_prep_, 0, sizeof(light_prep_));
}

void Volume::SetView(const Frame& frame) { SetView(frame, 1); }

void Volume::SetView(const Frame& frame, int rounds)
{
  view_bounds_.valid = 0;   // the visible list is about to change
  VULCAN_ASSERT_MSG(frame.depth_image, "missing depth image");
  VULCAN_ASSERT_MSG(rounds >= 1, "SetView needs at least one round");
  const vk_volume v = ToVk();
  const vk_frame f = frame.ToVk();
  if (request_stream_)
  {
    VULCAN_ASSERT_MSG(requests_ahead_.valid != 1, "a request pass made ahead and a request stream exclude each other");
    if (integrated_recorded_) VK_ASSERT(vk_stream_wait_event(request_stream_, integrated_));
    VK_ASSERT(vk_volume_set_view_rounds_split(&v, &f, GetLightPreparation(), rounds, request_stream_, requested_, Device::GetStream()));
  }
  else
  {
    // (with a record that is not valid this is vk_volume_set_view_rounds)
    const int code = vk_volume_set_view_rounds_ahead(&v, &f, GetLightPreparation(), rounds, &requests_ahead_, Device::GetStream());
    VULCAN_ASSERT_MSG(!(code == VK_ERR_ARGUMENT && requests_ahead_.valid == 1),
        "SetView of another frame than the one Tracer::Trace(keyframe, next_frame) announced");
    VK_ASSERT(code);
  }
  visible_count_stale_ = true;
  empty_ = false;
}

bool Volume::SetViewAtDevicePose(const Frame& frame, const vk_transform* pose_device, int rounds)
{
  VULCAN_ASSERT_MSG(frame.depth_image, "missing depth image");
  VULCAN_ASSERT_MSG(rounds >= 1 && pose_device, "SetViewAtDevicePose needs a device pose and at least one round");
  if (request_stream_ || requests_ahead_.valid == 1) return false;
  const vk_volume v = ToVk();
  const vk_frame f = frame.ToVk();
  vk_light_prep* prep = GetLightPreparation();
  if (prep && prep->normals_out) return false;
  const int code = vk_volume_set_view_at_device_pose(&v, &f, pose_device, prep, rounds, Device::GetStream());
  if (code == VK_ERR_UNSUPPORTED) return false;
  VK_ASSERT(code);
  view_bounds_.valid = 0;   // the visible list has changed
  visible_count_stale_ = true;
  empty_ = false;
  return true;
}

void Volume::EnableRequestStream()
{
  if (request_stream_) return;
  VK_ASSERT(vk_stream_create(&request_stream_));
  VK_ASSERT(vk_event_create_ordering(&requested_, 1));    // behind a pass that WRITES what the waiting stream reads (vk.h)
  VK_ASSERT(vk_event_create_ordering(&integrated_, 0));
  integrated_recorded_ = false;
}

void Volume::NoteIntegrated() const
{
  if (!request_stream_) return;
  VK_ASSERT(vk_event_record(integrated_, Device::GetStream()));
  integrated_recorded_ = true;
}

void Volume::ComputeNormalsAndSetView(Frame& frame, int rounds)
{
  if (requests_ahead_.valid == 1 && requests_ahead_.normals_made == 1)
  {
    // Tracer::Trace(keyframe, frame, true) announced this frame and its normals came with the pass: nothing is due.
    // (Stamping the normal image again would give the frame a content id the record does not name; another frame
    // than the announced one is refused by SetView.)
    SetView(frame, rounds);
    return;
  }
  if (requests_ahead_.valid == 1)
  {
    // Announced WITHOUT its normals (Tracer::Trace(keyframe, frame) — next_needs_normals defaults to false): they are
    // still due (this used to be skipped on the record's validity alone, and LightIntegrator then shaded from
    // a stale or uninitialised normal image, silently). The record names the frame by its images' stamps, so the normal
    // image is written in place without a new stamp (a kernel launch, as Frame::ComputeNormals; ref: src/frame.cpp:21-36),
    // the announced SetView runs, and whatever the pass prepared from the OLD normals is void: LightIntegrator::Integrate
    // then prepares again (vk_light_prepare) from the new ones.
    VULCAN_ASSERT_MSG(frame.depth_image, "missing depth image");
    const int w = frame.depth_image->GetWidth(), h = frame.depth_image->GetHeight();
    const bool sized = frame.normal_image && frame.normal_image->GetWidth() == w && frame.normal_image->GetHeight() == h;
    // (a frame announced without a normal image of its size cannot have had a preparation riding: prep_rides needs one,
    // and a new image is new content — SetView below then refuses the frame, as for any frame that was not announced)
    if (!sized) { frame.ComputeNormals(); SetView(frame, rounds); return; }
    const ColorImage& normals = *frame.normal_image;                  // const view: no new stamp
    const vk_projection k = frame.depth_projection.ToVk();
    const Image& depths = *frame.depth_image;
    VK_ASSERT(vk_frame_compute_normals(depths.GetData(), &k,
        const_cast<float*>(reinterpret_cast<const float*>(normals.GetData())), w, h, Device::GetStream()));
    vk_light_prep* made = GetLightPreparation();
    SetView(frame, rounds);
    if (made) made->valid = 0;
    return;
  }
  vk_light_prep* prep = GetLightPreparation();
  if (!prep)
  {
    frame.ComputeNormals();
    SetView(frame, rounds);
    return;
  }
  VULCAN_ASSERT_MSG(frame.depth_image, "missing depth image");
  if (!frame.normal_image) frame.normal_image = std::make_shared<ColorImage>();
  frame.normal_image->Resize(frame.depth_image->GetWidth(), frame.depth_image->GetHeight());
  prep->normals_out = reinterpret_cast<float*>(frame.normal_image->GetData());   // (GetData stamps the image: new content)
  SetView(frame, rounds);
}

const Buffer<HashEntry>& Volume::GetHashEntries() const { return hash_entries_; }

const Buffer<int>& Volume::GetAllocatedBlocks() const
{
  VULCAN_THROW("not implemented");  // as in the reference (volume.cu:444-448)
}

const Buffer<int>& Volume::GetVisibleBlocks() const
{
  if (visible_count_stale_)
  {
    visible_blocks_.Resize(GetBufferSize());  // volume.cu:494, deferred to first use
    visible_count_stale_ = false;
  }
  return visible_blocks_;
}

const Buffer<Voxel>& Volume::GetVoxels() const { return voxels_; }

Buffer<Voxel>& Volume::GetVoxels() { return voxels_; }

void Volume::GetCounters(int32_t* counters) const
{
  const vk_volume v = ToVk();
  VK_ASSERT(vk_volume_read_counters_sync(&v, counters, Device::GetStream()));
  NotePoolExhaustion(counters[VK_CTR_DROPPED]);
}

// Upstream asserts on the device, in debug builds, when the pool runs dry (VULCAN_DEBUG_MSG "voxel memory exhausted" /
// "excess memory exhausted", src/volume.cu:338,353) and drops the request silently otherwise. Here the dropped requests are
// counted on the device (VK_CTR_DROPPED) and never leave it inside the frame loop, so the note is made where the class
// already reads the counters: once per volume, on stderr, in builds without NDEBUG — upstream's condition for its message.
void Volume::NotePoolExhaustion(int32_t dropped) const
{
#ifndef NDEBUG
  if (dropped > 0 && !pool_exhaustion_noted_)
  {
    pool_exhaustion_noted_ = true;
    std::fprintf(stderr, "vulcan::Volume: voxel / excess memory exhausted: %d allocation requests dropped so far "
        "(pool of %d + %d blocks; ref: src/volume.cu:338,353)\n", dropped, main_block_count_, excess_block_count_);
  }
#else
  (void)dropped;
#endif
}

int Volume::GetAllocatedBlockCount() const
{
  int32_t counters[VK_CTR_PUBLIC];
  GetCounters(counters);
  // the free-slot pointer keeps falling below -1 once the pool is empty, as upstream's does (src/volume.cu:352-356)
  const int taken = max_block_count_ - 1 - counters[VK_CTR_VOXEL_PTR];
  return taken < max_block_count_ ? taken : max_block_count_;
}

void Volume::ResetBlockVisibility()
{
  VULCAN_ASSERT_MSG(requests_ahead_.valid != 1, "a frame announced by Tracer::Trace(keyframe, next_frame) has its requests in the volume: SetView(that frame) or CancelRequestsAhead() first");
  const vk_volume v = ToVk();
  VK_ASSERT(vk_volume_reset_block_visibility(&v, Device::GetStream()));
}

void Volume::UpdateBlockVisibility(const Frame& frame)
{
  view_bounds_.valid = 0;
  const vk_volume v = ToVk();
  const vk_projection k = frame.depth_projection.ToVk();
  const vk_transform Tdw = frame.depth_to_world_transform.Inverse().ToVk();
  VK_ASSERT(vk_volume_update_block_visibility(&v, frame.depth_image->GetWidth(),
      frame.depth_image->GetHeight(), &k, &Tdw, Device::GetStream()));
  visible_blocks_.Resize(GetBufferSize());
  visible_count_stale_ = false;
}

void Volume::CreateAllocationRequests(const Frame& frame)
{
  VULCAN_ASSERT_MSG(requests_ahead_.valid != 1, "a frame announced by Tracer::Trace(keyframe, next_frame) has its requests in the volume: SetView(that frame) or CancelRequestsAhead() first");
  const vk_volume v = ToVk();
  const vk_projection k = frame.depth_projection.ToVk();
  const vk_transform Twd = frame.depth_to_world_transform.ToVk();
  VK_ASSERT(vk_volume_create_allocation_requests(&v, frame.depth_image->GetData(),
      frame.depth_image->GetWidth(), frame.depth_image->GetHeight(), &k, &Twd, Device::GetStream()));
}

void Volume::HandleAllocationRequests()
{
  VULCAN_ASSERT_MSG(requests_ahead_.valid != 1, "a frame announced by Tracer::Trace(keyframe, next_frame) has its requests in the volume: SetView(that frame) or CancelRequestsAhead() first");
  const vk_volume v = ToVk();
  VK_ASSERT(vk_volume_handle_allocation_requests(&v, Device::GetStream()));
}

int Volume::GetBufferSize() const
{
  int32_t counters[VK_CTR_PUBLIC];
  GetCounters(counters);
  return counters[VK_CTR_VISIBLE];
}

void Volume::ResetBufferSize() const
{
  VK_ASSERT(vk_memset(const_cast<int*>(counters_.GetData()) + VK_CTR_VISIBLE, 0, sizeof(int),
      Device::GetStream()));
}

void Volume::Initialize()
{
  // this translation unit and libvk_hip.so must agree on the structs and on the size of the counters block (vk.h)
  VK_ASSERT(vk_abi_check(VK_ABI_VERSION, sizeof(vk_volume), sizeof(vk_frame), VK_CTR_COUNT));
  // sizes as in volume.cu:565-627
  voxels_.Resize(size_t(max_block_count_) * Block::voxel_count);
  hash_entries_.Resize(max_block_count_);
  free_voxel_blocks_.Resize(max_block_count_);
  allocation_types_.Resize(main_block_count_);
  allocation_blocks_.Resize(main_block_count_);
  block_visibility_.Resize(max_block_count_);
  visible_blocks_.Reserve(max_block_count_);
  counters_.Resize(VK_CTR_COUNT);
  const vk_volume v = ToVk();
  VK_ASSERT(vk_volume_initialize(&v, Device::GetStream()));
}

} // namespace vulcan
