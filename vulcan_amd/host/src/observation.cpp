// observation.cpp — Frame / Image host methods over the C ABI
// (ref: src/frame.cpp, src/image.cu:183-262).
#include <vulcan/observation.h>
#include <vulcan/exception.h>

namespace vulcan
{

void Image::Downsample(Image& image, bool nearest) const
{
  VULCAN_DEBUG_MSG(size_[0] % 2 == 0 && size_[1] % 2 == 0, "even image dimensions required");
  image.Resize(size_ / 2);
  VK_ASSERT(vk_image_downsample(size_[0], size_[1], data_, image.GetData(), nearest ? 1 : 0,
      Device::GetStream()));
}

void ColorImage::Downsample(ColorImage& image, bool nearest) const
{
  VULCAN_DEBUG_MSG(size_[0] % 2 == 0 && size_[1] % 2 == 0, "even image dimensions required");
  image.Resize(size_ / 2);
  VK_ASSERT(vk_color_image_downsample(size_[0], size_[1], reinterpret_cast<const float*>(data_),
      reinterpret_cast<float*>(image.GetData()), nearest ? 1 : 0, Device::GetStream()));
}

void Image::GetGradients(Image& gx, Image& gy) const
{
  gx.Resize(size_);
  gy.Resize(size_);
  VK_ASSERT(vk_image_gradients(size_[0], size_[1], data_, gx.GetData(), gy.GetData(), Device::GetStream()));
}

void ColorImage::ConvertTo(Image& image) const
{
  image.Resize(size_);
  VK_ASSERT(vk_color_image_convert(GetTotal(), reinterpret_cast<const float*>(data_), image.GetData(),
      Device::GetStream()));
}

void ComputeNormals(const float* depths, const Projection& projection,
    Vector3f* normals, int image_width, int image_height)
{
  const vk_projection k = projection.ToVk();
  VK_ASSERT(vk_frame_compute_normals(depths, &k, reinterpret_cast<float*>(normals), image_width,
      image_height, Device::GetStream()));
}

void FilterDepths(int image_width, int image_height, const float* src, float* dst)
{
  VK_ASSERT(vk_frame_filter_depths(image_width, image_height, src, dst, Device::GetStream()));
}

void Frame::FilterDepths()
{
  VULCAN_ASSERT_MSG(depth_image, "missing depth image");
  const int w = depth_image->GetWidth();
  const int h = depth_image->GetHeight();
  Image filtered(w, h);
  vulcan::FilterDepths(w, h, depth_image->GetData(), filtered.GetData());
  VK_ASSERT(vk_memcpy_d2d(depth_image->GetData(), filtered.GetData(), filtered.GetBytes(),
      Device::GetStream()));
  Device::Synchronize();  // `filtered` is freed on return
}

void Frame::ComputeNormals()
{
  VULCAN_ASSERT_MSG(depth_image, "missing depth image");
  if (!normal_image) normal_image = std::make_shared<ColorImage>();
  const int w = depth_image->GetWidth();
  const int h = depth_image->GetHeight();
  normal_image->Resize(w, h);
  vulcan::ComputeNormals(depth_image->GetData(), depth_projection, normal_image->GetData(), w, h);
}

void Frame::Downsample(Frame& frame) const
{
  VULCAN_DEBUG(depth_image);
  VULCAN_DEBUG(color_image);
  VULCAN_DEBUG(normal_image);

  if (!frame.depth_image) frame.depth_image = std::make_shared<Image>();
  if (!frame.color_image) frame.color_image = std::make_shared<ColorImage>();
  if (!frame.normal_image) frame.normal_image = std::make_shared<ColorImage>();

  // depth and normals nearest, colour 2x2 box (frame.cpp:49-51): one launch for the three
  VULCAN_DEBUG_MSG(depth_image->GetWidth() % 2 == 0 && depth_image->GetHeight() % 2 == 0, "even image dimensions required");
  VULCAN_DEBUG_MSG(color_image->GetWidth() % 2 == 0 && color_image->GetHeight() % 2 == 0, "even image dimensions required");
  frame.depth_image->Resize(depth_image->GetSize() / 2);
  frame.color_image->Resize(color_image->GetSize() / 2);
  frame.normal_image->Resize(normal_image->GetSize() / 2);
  const vk_frame f = ToVk();
  VK_ASSERT(vk_frame_downsample(&f, frame.depth_image->GetData(), reinterpret_cast<float*>(frame.color_image->GetData()),
      reinterpret_cast<float*>(frame.normal_image->GetData()), Device::GetStream()));

  frame.depth_projection.SetFocalLength(depth_projection.GetFocalLength() / 2);
  frame.depth_projection.SetCenterPoint(depth_projection.GetCenterPoint() / 2);
  frame.color_projection.SetFocalLength(color_projection.GetFocalLength() / 2);
  frame.color_projection.SetCenterPoint(color_projection.GetCenterPoint() / 2);
  frame.depth_to_world_transform = depth_to_world_transform;
  frame.depth_to_color_transform = depth_to_color_transform;
}

vk_frame Frame::ToVk() const
{
  // read-only views: a non-const GetData() would stamp the images as modified
  const Image* depth = depth_image.get();
  const ColorImage* color = color_image.get();
  const ColorImage* normals = normal_image.get();
  vk_frame f;
  f.depth = depth ? depth->GetData() : nullptr;
  f.color = color ? reinterpret_cast<const float*>(color->GetData()) : nullptr;
  f.normals = normals ? reinterpret_cast<const float*>(normals->GetData()) : nullptr;
  f.width = depth ? depth->GetWidth() : (color ? color->GetWidth() : 0);
  f.height = depth ? depth->GetHeight() : (color ? color->GetHeight() : 0);
  f.color_width = color ? color->GetWidth() : 0;     // color_integrator.cu:183-184
  f.color_height = color ? color->GetHeight() : 0;
  f.depth_projection = depth_projection.ToVk();
  f.color_projection = color_projection.ToVk();
  f.depth_to_world = depth_to_world_transform.ToVk();
  f.depth_to_color = depth_to_color_transform.ToVk();
  // which content this is: the three images' stamps, mixed (0 = unknown is never produced here)
  const uint64_t a = depth ? depth->GetContentStamp() : 0, b = color ? color->GetContentStamp() : 0,
      c = normals ? normals->GetContentStamp() : 0;
  f.content_id = ((a * 0x9E3779B97F4A7C15ull) ^ (b * 0xC2B2AE3D27D4EB4Full) ^ (c * 0x165667B19E3779F9ull)) | 1ull;
  return f;
}

} // namespace vulcan
