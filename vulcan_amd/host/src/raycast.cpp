// raycast.cpp — Tracer host class and the raycast stage functions
// (ref: src/tracer.cpp, src/tracer.cu:453-500).
#include <vulcan/raycast.h>
#include <vulcan/block.h>
#include <vulcan/device.h>
#include <vulcan/exception.h>
#include <vulcan/observation.h>
#include <vulcan/hash.h>
#include <vulcan/tsdf_volume.h>
#include <vulcan/voxel.h>

namespace vulcan
{

static_assert(sizeof(Patch) == sizeof(vk_patch), "Patch must match vk_patch");

// ---- stage functions (raw device pointers) -----------------------------------

void ComputePatches(const int* indices, const HashEntry* entries,
    const Transform& Tcw, const Projection& projection, float block_length,
    float min_depth, float max_depth, int block_count, int image_width,
    int image_height, int bounds_width, int bounds_height, Patch* patches,
    int* patch_count)
{
  const vk_transform T = Tcw.ToVk();
  const vk_projection k = projection.ToVk();
  // the reference takes no capacity; Tracer reserves 262144 patches (tracer.cpp:134)
  VK_ASSERT(vk_trace_compute_patches(indices, reinterpret_cast<const vk_hash_entry*>(entries), &T, &k,
      block_length, min_depth, max_depth, block_count, nullptr, image_width, image_height,
      bounds_width, bounds_height, reinterpret_cast<vk_patch*>(patches), 262144, patch_count,
      Device::GetStream()));
}

void ComputeBounds(const Patch* patches, Vector2f* bounds, int bounds_width, int patch_count)
{
  VK_ASSERT(vk_trace_compute_bounds(reinterpret_cast<const vk_patch*>(patches),
      reinterpret_cast<float*>(bounds), bounds_width, patch_count, nullptr, Device::GetStream()));
}

void ComputePoints(const HashEntry* entries, const Voxel* voxels,
    const Vector2f* bounds, int block_count, float block_length,
    float voxel_length, float trunc_length, const Transform& Twc,
    const Projection& projection, float* depths, Vector3f* colors,
    int image_width, int image_height, int bounds_width, int bounds_height)
{
  const vk_transform T = Twc.ToVk();
  const vk_projection k = projection.ToVk();
  VK_ASSERT(vk_trace_compute_points(reinterpret_cast<const vk_hash_entry*>(entries),
      reinterpret_cast<const vk_voxel*>(voxels), reinterpret_cast<const float*>(bounds), block_count,
      block_length, voxel_length, trunc_length, &T, &k, depths, reinterpret_cast<float*>(colors),
      image_width, image_height, bounds_width, bounds_height, Device::GetStream()));
}

void ResetBoundsBuffer(Vector2f* bounds, int count)
{
  VK_ASSERT(vk_trace_reset_bounds(reinterpret_cast<float*>(bounds), count, Device::GetStream()));
}

// ---- Tracer --------------------------------------------------------------------

Tracer::Tracer(std::shared_ptr<const Volume> volume) :
  volume_(volume),
  depth_range_(0.1f, 5.0f),
  bounds_width_(80),    // tracer.cpp:56-57 "TODO: expose variable"
  bounds_height_(60)
{
  Initialize();
}

Tracer::~Tracer()
{
  // the volume must not keep a pointer into a buffer that is about to be freed
  if (volume_) volume_->DetachViewBounds(reinterpret_cast<const float*>(bounds_.GetData()));
}

std::shared_ptr<const Volume> Tracer::GetVolume() const { return volume_; }

const Vector2f& Tracer::GetDepthRange() const { return depth_range_; }

void Tracer::SetDepthRange(const Vector2f& range)
{
  VULCAN_DEBUG(range[0] > 0 && range[0] < range[1]);
  depth_range_ = range;
}

void Tracer::SetDepthRange(float min, float max) { SetDepthRange(Vector2f(min, max)); }

void Tracer::Trace(Frame& frame)
{
  TraceWith(frame, nullptr, false);
}

void Tracer::TraceWithoutNormals(Frame& frame)
{
  TraceWith(frame, nullptr, false, false);
}

void Tracer::SettleNormals()
{
  vk_view_bounds* ahead = volume_->GetViewBounds();
  if (!ahead) return;
  // VK_ERR_TIMEOUT -> vulcan::Exception: upstream's contract is that a failed device step always throws (device.h:14-17)
  VK_ASSERT(vk_trace_normals_settle(ahead, Device::GetStream()));
}

void Tracer::Trace(Frame& frame, Frame& next_frame, bool next_needs_normals)
{
  TraceWith(frame, &next_frame, next_needs_normals);
}

void Tracer::TraceWith(Frame& frame, Frame* next, bool next_needs_normals, bool with_normals)
{
  VULCAN_ASSERT_MSG(frame.depth_image, "missing depth image");
  const int w = frame.depth_image->GetWidth();
  const int h = frame.depth_image->GetHeight();
  if (!frame.color_image) frame.color_image = std::make_shared<ColorImage>();
  if (!frame.normal_image) frame.normal_image = std::make_shared<ColorImage>();
  frame.color_image->Resize(w, h);
  frame.normal_image->Resize(w, h);

  const vk_volume v = volume_->ToVk();
  const vk_frame f = frame.ToVk();
  // the volume's record, if it is (still) this tracer's; the bounds pass is skipped
  // when an integrator has already prepared the bounds of this very view
  vk_view_bounds* ahead = volume_->GetViewBounds();
  float* scratch = reinterpret_cast<float*>(bounds_.GetData());
  if (!ahead || ahead->scratch != scratch || ahead->min_depth != depth_range_[0] || ahead->max_depth != depth_range_[1])
  {
    volume_->AttachViewBounds(scratch, bounds_width_, bounds_height_, depth_range_);
    ahead = volume_->GetViewBounds();
  }
  float* depth_out = frame.depth_image->GetData();
  float* color_out = reinterpret_cast<float*>(frame.color_image->GetData());
  float* normals_out = reinterpret_cast<float*>(frame.normal_image->GetData());
  if (!next || volume_->GetRequestStream())
  {
    VK_ASSERT(vk_trace_ahead(&v, &f, ahead, depth_out, color_out, with_normals ? normals_out : nullptr, Device::GetStream()));
    if (next && next_needs_normals) next->ComputeNormals();
    return;
  }
  VULCAN_ASSERT_MSG(next->depth_image, "missing depth image");
  vk_light_prep* prep = volume_->GetLightPreparation();
  if (next_needs_normals)
  {
    if (!prep) next->ComputeNormals();
    else
    {
      if (!next->normal_image) next->normal_image = std::make_shared<ColorImage>();
      next->normal_image->Resize(next->depth_image->GetWidth(), next->depth_image->GetHeight());
      prep->normals_out = reinterpret_cast<float*>(next->normal_image->GetData());   // (GetData stamps the image: new content)
    }
  }
  const vk_frame n = next->ToVk();
  VK_ASSERT(vk_trace_ahead_requests(&v, &f, ahead, depth_out, color_out, normals_out, &n, prep, volume_->GetRequestsAhead(),
      Device::GetStream()));
  // the record says whether the announced frame's normals are made (the library knows of the riding ones only)
  if (next_needs_normals && !prep && volume_->GetRequestsAhead()->valid == 1) volume_->GetRequestsAhead()->normals_made = 1;
  if (prep && prep->normals_out && volume_->GetRequestsAhead()->valid != 1)
  {
    // the pass could not be made ahead (vk.h): the normals are then the caller's launch, as in Frame::ComputeNormals
    prep->normals_out = nullptr;
    next->ComputeNormals();
  }
}

void Tracer::ComputePatches(const Frame& frame)
{
  const Buffer<int>& visible = volume_->GetVisibleBlocks();
  const Buffer<HashEntry>& entries = volume_->GetHashEntries();
  const float block_length = Block::resolution * volume_->GetVoxelLength();

  ResetBufferSize();

  vulcan::ComputePatches(visible.GetData(), entries.GetData(),
      frame.depth_to_world_transform.Inverse(), frame.depth_projection, block_length,
      depth_range_[0], depth_range_[1], visible.GetSize(), frame.depth_image->GetWidth(),
      frame.depth_image->GetHeight(), bounds_width_, bounds_height_, patches_.GetData(),
      buffer_size_.GetData());

  patches_.Resize(min<size_t>(GetBufferSize(), patches_.GetCapacity()));
}

void Tracer::ComputeBounds(const Frame&)
{
  ResetBoundsBuffer();
  vulcan::ComputeBounds(patches_.GetData(), bounds_.GetData(), bounds_width_, patches_.GetSize());
}

void Tracer::ComputePoints(Frame& frame)
{
  const int w = frame.depth_image->GetWidth();
  const int h = frame.depth_image->GetHeight();
  const float voxel_length = volume_->GetVoxelLength();
  if (!frame.color_image) frame.color_image = std::make_shared<ColorImage>();
  frame.color_image->Resize(w, h);

  vulcan::ComputePoints(volume_->GetHashEntries().GetData(), volume_->GetVoxels().GetData(),
      bounds_.GetData(), volume_->GetMainBlockCount(), Block::resolution * voxel_length, voxel_length,
      volume_->GetTruncationLength(), frame.depth_to_world_transform, frame.depth_projection,
      frame.depth_image->GetData(), frame.color_image->GetData(), w, h, bounds_width_, bounds_height_);
}

void Tracer::ComputeNormals(Frame& frame)
{
  frame.ComputeNormals();
}

void Tracer::ResetBoundsBuffer()
{
  vulcan::ResetBoundsBuffer(bounds_.GetData(), bounds_width_ * bounds_height_);
}

void Tracer::ResetBufferSize()
{
  VK_ASSERT(vk_memset(buffer_size_.GetData(), 0, sizeof(int), Device::GetStream()));
}

int Tracer::GetBufferSize()
{
  int value = 0;
  VK_ASSERT(vk_memcpy_d2h(&value, buffer_size_.GetData(), sizeof(int), Device::GetStream()));
  return value;
}

void Tracer::Initialize()
{
  patches_.Reserve(262144);  // tracer.cpp:134
  // tracer.cpp:140 sizes this 80*60; the fused bounds pass keeps its per-workgroup
  // copies behind the grid (vk_trace_bounds_floats)
  bounds_.Resize(vk_trace_bounds_floats(bounds_width_, bounds_height_) / 2);
  buffer_size_.Resize(1);
  // from now on the volume's integrators prepare this tracer's bounds
  volume_->AttachViewBounds(reinterpret_cast<float*>(bounds_.GetData()), bounds_width_, bounds_height_, depth_range_);
}

} // namespace vulcan
