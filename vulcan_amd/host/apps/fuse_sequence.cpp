// fuse_sequence.cpp — the frame loop of the reference's demo app
// (ref: apps/vulcan/vulcan.cu:181-367) on synthetic input: the HAL camera,
// OpenCV conversion and PNG dumps are replaced by a closed-form depth image
// (camera at the centre of a sphere, yawing), everything else is the same
// sequence of class calls:  [Track] -> SetView -> Integrate -> Trace.
//
//   fuse_sequence [frames=200] [mode=0|1|2]
//     0  SetView + DepthIntegrator + Tracer                       (BASELINE configs[1])
//     1  PyramidTracker<DepthTracker> in front of mode 0          (configs[2] tracking)
//     2  PyramidTracker<LightTracker> + LightIntegrator + Tracer: what the shipped
//        app is configured to run (vulcan.cu:87-111), on a textured, lit sphere
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <vulcan/vulcan.h>

using namespace vulcan;

int main(int argc, char** argv)
{
  const int frames = argc > 1 ? std::atoi(argv[1]) : 200;
  const int mode = argc > 2 ? std::atoi(argv[2]) : 0;
  const bool track = mode != 0;
  const int w = 640, h = 480;
  const float radius = 2.0f;

  int devices = 0;
  VK_ASSERT(vk_device_count(&devices));
  if (devices == 0) { std::fprintf(stderr, "no HIP device\n"); return 2; }

  // app defaults: vulcan.cu:12-13 (65024 + 8192 blocks), :283-287 intrinsics, 5 mm voxels
  auto volume = std::make_shared<Volume>(65024, 8192);
  volume->SetVoxelLength(0.005f);
  volume->SetTruncationLength(0.04f);
  DepthIntegrator depth_integrator(volume);
  LightIntegrator light_integrator(volume);
  Tracer tracer(volume);
  PyramidTracker<DepthTracker> depth_tracker;
  PyramidTracker<LightTracker> light_tracker;
  Light light;
  light.SetIntensity(2.0f);                       // vulcan.cu:87-88
  light.SetPosition(0.025f, 0.08f, 0.0f);
  light_integrator.SetLight(light);
  std::const_pointer_cast<LightTracker>(light_tracker.GetTracker())->SetLight(light);

  Frame frame;
  frame.depth_projection.SetFocalLength(544.162f, 544.3847f);
  frame.depth_projection.SetCenterPoint(311.2701f, 234.7798f);
  frame.color_projection = frame.depth_projection;
  std::vector<float> depth(size_t(w) * h);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x)
    {
      const Vector3f ray = frame.depth_projection.Unproject(x + 0.5f, y + 0.5f);
      depth[size_t(y) * w + x] = radius / ray.Norm();
    }
  frame.depth_image = std::make_shared<Image>(w, h);
  frame.depth_image->CopyFromHost(depth.data());
  frame.color_image = std::make_shared<ColorImage>(w, h);
  if (mode == 2)
  {
    // a smooth texture on the sphere, shaded by the light next to the camera
    std::vector<Vector3f> colors(size_t(w) * h);
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x)
      {
        const Vector3f ray = frame.depth_projection.Unproject(x + 0.5f, y + 0.5f);
        const Vector3f point = depth[size_t(y) * w + x] * ray;
        const Vector3f normal = Vector3f(-point[0], -point[1], -point[2]) / radius;   // inward
        const float albedo = 0.5f + 0.2f * std::cos(9.0f * point[0]) + 0.2f * std::cos(7.0f * point[1]);
        const float c = albedo * light.GetShading(point, normal);
        colors[size_t(y) * w + x] = Vector3f(c, c, c);
      }
    frame.color_image->CopyFromHost(colors.data());
  }
  frame.ComputeNormals();

  auto keyframe = std::make_shared<Frame>();
  keyframe->depth_projection = keyframe->color_projection = frame.depth_projection;
  keyframe->depth_image = std::make_shared<Image>(w, h);
  keyframe->color_image = std::make_shared<ColorImage>(w, h);
  keyframe->normal_image = std::make_shared<ColorImage>(w, h);

  Device::Synchronize();
  const auto t0 = std::chrono::steady_clock::now();

  for (int i = 0; i < frames; ++i)
  {
    const float half = 0.5f * (0.5f * i) * float(M_PI) / 180.0f;   // 0.5 degree of yaw per frame
    const Transform truth = Transform::Rotate(std::cos(half), 0.0f, std::sin(half), 0.0f);

    if (track && i > 0)
    {
      // start from the previous pose, refine against the raycast keyframe (vulcan.cu:300-311)
      if (mode == 2) { light_tracker.SetKeyframe(keyframe); light_tracker.Track(frame); }
      else { depth_tracker.SetKeyframe(keyframe); depth_tracker.Track(frame); }
    }
    else
    {
      frame.depth_to_world_transform = truth;
    }

    volume->SetView(frame);            // vulcan.cu:316-318
    if (mode == 2) light_integrator.Integrate(frame); else depth_integrator.Integrate(frame);   // :321
    keyframe->depth_to_world_transform = frame.depth_to_world_transform;
    tracer.Trace(*keyframe);           // :325
  }

  Device::Synchronize();
  const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  int32_t counters[VK_CTR_COUNT];
  volume->GetCounters(counters);
  std::printf("frames %d  time %.3f s  fps %.1f  visible %d  allocated %d  dropped %d  tracking %s\n", frames,
      seconds, frames / seconds, counters[VK_CTR_VISIBLE], 65024 + 8192 - 1 - counters[VK_CTR_VOXEL_PTR],
      counters[VK_CTR_DROPPED], mode == 0 ? "off" : (mode == 1 ? "depth" : "light"));
  const Matrix4f M = frame.depth_to_world_transform.GetMatrix();
  std::printf("final pose row0: %.5f %.5f %.5f %.5f\n", M(0, 0), M(0, 1), M(0, 2), M(0, 3));
  return 0;
}
