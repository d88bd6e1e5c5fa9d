// observation.h — one RGB-D observation (struct Frame) and the two image
// operators that act on it.
//
// API parity: Frame keeps the member names of the reference's frame.h:11-32 so
// callers that fill a Frame by hand compile unchanged; ComputeNormals and
// FilterDepths keep the signatures of frame.cuh:10-14. Both header names remain
// as forwarders to this file.
#pragma once

#include <memory>
#include <vk.h>
#include <vulcan/image.h>
#include <vulcan/matrix.h>
#include <vulcan/projection.h>
#include <vulcan/transform.h>

namespace vulcan
{

struct Frame
{
  // intrinsics of the two cameras
  Projection depth_projection;
  Projection color_projection;

  // depth camera -> world, depth camera -> colour camera
  Transform depth_to_world_transform;
  Transform depth_to_color_transform;

  // device images; shared so pyramids and keyframes can alias them
  std::shared_ptr<Image> depth_image;
  std::shared_ptr<ColorImage> color_image;
  std::shared_ptr<ColorImage> normal_image;

  void FilterDepths();                   // in-place bilateral-style depth filter
  void ComputeNormals();                 // normal_image from depth_image
  void Downsample(Frame& frame) const;   // half-resolution copy, intrinsics scaled

  vk_frame ToVk() const;                 // C-ABI view; pointers stay owned by the images

  // ---- conveniences that are not upstream ----

  // size of the depth image (0 x 0 while it is unset)
  int GetWidth() const { return depth_image ? depth_image->GetWidth() : 0; }
  int GetHeight() const { return depth_image ? depth_image->GetHeight() : 0; }

  bool HasColor() const { return color_image && color_image->GetTotal() > 0; }
  bool HasNormals() const { return normal_image && normal_image->GetTotal() > 0; }

  // create (or resize) the images a Tracer::Trace target needs
  void Allocate(int width, int height, bool with_color = true, bool with_normals = true)
  {
    if (!depth_image) depth_image = std::make_shared<Image>();
    depth_image->Resize(width, height);
    if (with_color)
    {
      if (!color_image) color_image = std::make_shared<ColorImage>();
      color_image->Resize(width, height);
    }
    if (with_normals)
    {
      if (!normal_image) normal_image = std::make_shared<ColorImage>();
      normal_image->Resize(width, height);
    }
  }
};

void ComputeNormals(const float* depths, const Projection& projection,
    Vector3f* normals, int image_width, int image_height);

void FilterDepths(int image_width, int image_height, const float* src, float* dst);

} // namespace vulcan
