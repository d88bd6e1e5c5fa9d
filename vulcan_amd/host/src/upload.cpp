// upload.cpp — FrameUploader (vulcan/upload.h): double-buffered asynchronous upload of camera frames
// (ref: include/vulcan/image.h:100-123, apps/vulcan/vulcan.cu:220,232 — blocking there).
#include <vulcan/upload.h>

namespace vulcan
{

FrameUploader::FrameUploader(int width, int height, bool with_color) : width_(width), height_(height),
  with_color_(with_color), copy_stream_(nullptr), submitted_(0), acquired_(0), released_(0)
{
  VULCAN_ASSERT(width > 0 && height > 0);
  VK_ASSERT(vk_stream_create(&copy_stream_));
  const size_t pixels = size_t(width) * height;
  for (int s = 0; s < slot_count; ++s)
  {
    void* p = nullptr;
    VK_ASSERT(vk_malloc_host(&p, pixels * sizeof(float)));
    staging_depth_[s] = static_cast<float*>(p);
    staging_color_[s] = nullptr;
    if (with_color)
    {
      VK_ASSERT(vk_malloc_host(&p, pixels * sizeof(Vector3f)));
      staging_color_[s] = static_cast<Vector3f*>(p);
    }
    depth_[s] = std::make_shared<Image>(width, height);
    if (with_color) color_[s] = std::make_shared<ColorImage>(width, height);
    VK_ASSERT(vk_event_create_ordering(&uploaded_[s], 1));   // the copy wrote what the frame's kernels read
    VK_ASSERT(vk_event_create_ordering(&consumed_[s], 0));   // the kernels only read what the next copy overwrites
    consumed_recorded_[s] = uploaded_recorded_[s] = false;
  }
}

FrameUploader::~FrameUploader()
{
  // nothing may still be copying out of the staging buffers or into the images
  (void)vk_stream_synchronize(copy_stream_);
  (void)vk_stream_synchronize(Device::GetStream());
  for (int s = 0; s < slot_count; ++s)
  {
    (void)vk_event_destroy(uploaded_[s]);
    (void)vk_event_destroy(consumed_[s]);
    (void)vk_free_host(staging_depth_[s]);
    (void)vk_free_host(staging_color_[s]);
  }
  (void)vk_stream_destroy(copy_stream_);
}

float* FrameUploader::StagingDepth()
{
  const int s = submitted_ % slot_count;
  if (uploaded_recorded_[s]) VK_ASSERT(vk_event_synchronize(uploaded_[s]));   // the copy has left the buffer
  return staging_depth_[s];
}

Vector3f* FrameUploader::StagingColor()
{
  const int s = submitted_ % slot_count;
  if (uploaded_recorded_[s]) VK_ASSERT(vk_event_synchronize(uploaded_[s]));
  return staging_color_[s];
}

void FrameUploader::Submit()
{
  VULCAN_ASSERT(submitted_ - released_ < slot_count);
  const int s = submitted_ % slot_count;
  // the frame that used the slot's images last (slot_count frames back) must have read them: the HOST waits (upload.h)
  if (consumed_recorded_[s]) VK_ASSERT(vk_event_synchronize(consumed_[s]));
  const size_t pixels = size_t(width_) * height_;
  VK_ASSERT(vk_memcpy_h2d_async(depth_[s]->GetData(), staging_depth_[s], pixels * sizeof(float), copy_stream_));
  if (with_color_)
    VK_ASSERT(vk_memcpy_h2d_async(color_[s]->GetData(), staging_color_[s], pixels * sizeof(Vector3f), copy_stream_));
  VK_ASSERT(vk_event_record(uploaded_[s], copy_stream_));
  uploaded_recorded_[s] = true;
  ++submitted_;
}

void FrameUploader::Acquire(Frame& frame, void* second_stream)
{
  VULCAN_ASSERT(acquired_ < submitted_ && acquired_ == released_);
  const int s = acquired_ % slot_count;
  VK_ASSERT(vk_stream_wait_event(Device::GetStream(), uploaded_[s]));
  if (second_stream) VK_ASSERT(vk_stream_wait_event(second_stream, uploaded_[s]));
  depth_[s]->Touch();                       // new content (vk_frame.content_id): nothing prepared for the old one applies
  frame.depth_image = depth_[s];
  if (with_color_)
  {
    color_[s]->Touch();
    frame.color_image = color_[s];
  }
  ++acquired_;
}

void FrameUploader::Release()
{
  VULCAN_ASSERT(released_ < acquired_);
  const int s = released_ % slot_count;
  VK_ASSERT(vk_event_record(consumed_[s], Device::GetStream()));
  consumed_recorded_[s] = true;
  ++released_;
}

} // namespace vulcan
