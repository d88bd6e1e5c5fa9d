// fuse_sequence.cpp — the frame loop of the reference's demo app
// (ref: apps/vulcan/vulcan.cu:181-367) on synthetic input: the HAL camera,
// OpenCV conversion and PNG dumps are replaced by closed-form images that are
// resident on the device before the clock starts; everything else is the same
// sequence of class calls per frame:
//   frame.ComputeNormals() -> [tracker.Track(frame)] -> volume.SetView(frame) x3 ->
//   integrator.Integrate(frame) -> tracer.Trace(keyframe)       (vulcan.cu:297-325)
// The three SetView calls are one SetView(frame, 3) (same state, tsdf_volume.h).
//
//   fuse_sequence [frames=200] [mode=0|1|2|3] [stream=0|1] [split=0|1] [ahead=0|1]
//     ahead = 1 (mode 0): every raycast also makes the NEXT frame's request pass, in its own launch
//     (Tracer::Trace(keyframe, next_frame)); SetView is then left with its handle + visibility launch
//     split = 1 (mode 0): SetView's request pass on a stream of its own, beside the previous frame's raycast
//     (Volume::EnableRequestStream; the poses are known in advance in mode 0)
//     stream = 1 (mode 0): every frame's depth image is UPLOADED from pinned host memory while the frame before it is
//     fused (FrameUploader, vulcan/upload.h) instead of waiting in device memory — upstream uploads each frame with a
//     blocking copy (image.h:100-123, vulcan.cu:220,232)
//     0  DepthIntegrator + Tracer, no tracking                    (BASELINE configs[1]):
//        camera at the centre of a 2 m sphere, yawing 0.5 degree per frame
//     1  PyramidTracker<DepthTracker> in front of mode 0          (configs[2] tracking)
//     2  PyramidTracker<LightTracker> (15 + 20 Gauss-Newton steps) + LightIntegrator (default weight
//        caps 16 / 16) + Tracer: the line upstream keeps commented out (vulcan.cu:109)
//     3  the shipped app's configuration, exactly (vulcan.cu:89-111): a plain LightTracker with
//        SetMaxIterations(1), LightIntegrator with SetMaxDistanceWeight(100) / SetMaxColorWeight(16)
//   Modes 1 to 3 run CLOSED LOOP in the room scene (room_scene.h): every frame is tracked
//   from the previous tracked pose against the previous raycast, then fused and raycast at
//   the tracked pose; the true poses only score the result.
// Prints one human-readable line and one JSON line.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include <vulcan/vulcan.h>

#include "room_scene.h"

using namespace vulcan;

namespace
{

// translation error (m) and rotation error (degrees) of `got` against `truth`
void PoseError(const Transform& got, const Transform& truth, double& translation, double& rotation)
{
  const Matrix4f D = got.GetMatrix() * truth.GetInverseMatrix();
  const double trace = double(D(0, 0)) + D(1, 1) + D(2, 2);
  rotation = std::acos(std::min(1.0, std::max(-1.0, (trace - 1.0) / 2.0))) * 180.0 / M_PI;
  const Vector3f a = got.GetTranslation(), b = truth.GetTranslation();
  translation = std::sqrt(double(a[0] - b[0]) * (a[0] - b[0]) + double(a[1] - b[1]) * (a[1] - b[1]) +
      double(a[2] - b[2]) * (a[2] - b[2]));
}

}  // namespace

int main(int argc, char** argv)
{
  const int frames = argc > 1 ? std::atoi(argv[1]) : 200;
  const int mode = argc > 2 ? std::atoi(argv[2]) : 0;
  const bool track = mode != 0;
  const bool stream_input = argc > 3 && std::atoi(argv[3]) == 1 && mode == 0;
  const bool split_streams = argc > 4 && std::atoi(argv[4]) == 1 && mode == 0;
  const bool requests_ahead = argc > 5 && std::atoi(argv[5]) == 1 && mode == 0 && !split_streams && !stream_input;
  const int w = 640, h = 480;
  const float radius = 2.0f;

  int devices = 0;
  VK_ASSERT(vk_device_count(&devices));
  if (devices == 0) { std::fprintf(stderr, "no HIP device\n"); return 2; }

  // The classes submit to Device::GetStream(), which starts as the legacy default stream like upstream's stream 0
  // (device.h:40-52). On some boxes of this pool consecutive launches on the legacy stream follow one another about 1 us
  // later than on a created one (bench.py: 85.6-86.8 against 82.5-83.4 us per frame in alternating runs), so the app creates
  // its stream; VK_APP_LEGACY_STREAM=1 keeps upstream's.
  void* app_stream = nullptr;
  const char* legacy = std::getenv("VK_APP_LEGACY_STREAM");
  if (!(legacy && legacy[0] == '1'))
  {
    VK_ASSERT(vk_stream_create(&app_stream));
    Device::SetStream(app_stream);
  }

  // app defaults: vulcan.cu:12-13 (65024 + 8192 blocks), :283-287 intrinsics, 5 mm voxels
  // apps/vulcan/vulcan.cu:12-13: Volume(65024, 8192). A 7th argument gives the EXCESS block count instead (round 6,
  // profiles/r06_soak.json finding 1: the app's 8 192 chained entries are what runs out first — at 26 to 33 k allocated
  // blocks — and upstream's allocator then drains the pool; 65 536 entries cost 590 MB of the GPU's 288 GB)
  const int excess_blocks = argc > 6 && std::atoi(argv[6]) > 0 ? std::atoi(argv[6]) : 8192;
  auto volume = std::make_shared<Volume>(65024, excess_blocks);
  volume->SetVoxelLength(0.005f);
  volume->SetTruncationLength(0.04f);
  if (split_streams) volume->EnableRequestStream();
  DepthIntegrator depth_integrator(volume);
  LightIntegrator light_integrator(volume);
  Tracer tracer(volume);
  PyramidTracker<DepthTracker> depth_tracker;
  PyramidTracker<LightTracker> light_tracker;
  LightTracker app_tracker;                       // vulcan.cu:106-111: no pyramid, one step per frame
  Light light;
  light.SetIntensity(2.0f);                       // vulcan.cu:87-88
  light.SetPosition(0.025f, 0.08f, 0.0f);
  light_integrator.SetLight(light);
  std::const_pointer_cast<LightTracker>(light_tracker.GetTracker())->SetLight(light);
  app_tracker.SetLight(light);
  app_tracker.SetMaxIterations(1);                // vulcan.cu:111
  const bool photometric = mode == 2 || mode == 3;
  if (mode == 3)
  {
    light_integrator.SetMaxDistanceWeight(100);   // vulcan.cu:92-93
    light_integrator.SetMaxColorWeight(16);
  }

  Frame frame;
  frame.depth_projection.SetFocalLength(544.162f, 544.3847f);
  frame.depth_projection.SetCenterPoint(311.2701f, 234.7798f);
  frame.color_projection = frame.depth_projection;

  // ---- the camera's frames, resident on the device before the clock starts ----
  std::vector<std::shared_ptr<Image>> depth_images;
  std::vector<std::shared_ptr<ColorImage>> color_images;
  std::vector<Transform> truth(frames);
  std::vector<float> depth;
  std::vector<Vector3f> colors;
  if (!track)
  {
    depth.resize(size_t(w) * h);
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x)
        depth[size_t(y) * w + x] = radius / frame.depth_projection.Unproject(x + 0.5f, y + 0.5f).Norm();
    depth_images.push_back(std::make_shared<Image>(w, h));
    depth_images[0]->CopyFromHost(depth.data());
    color_images.push_back(std::make_shared<ColorImage>(w, h));
    for (int i = 0; i < frames; ++i)
    {
      const float half = 0.5f * (0.5f * i) * float(M_PI) / 180.0f;   // 0.5 degree of yaw per frame
      truth[i] = Transform::Rotate(std::cos(half), 0.0f, std::sin(half), 0.0f);
    }
  }
  else
  {
    const double lamp[3] = {0.025, 0.08, 0.0};
    for (int i = 0; i < frames; ++i)
    {
      truth[i] = room::Pose(i);
      room::Render(frame.depth_projection, truth[i], w, h, 2.0, lamp, depth, colors);
      depth_images.push_back(std::make_shared<Image>(w, h));
      depth_images[i]->CopyFromHost(depth.data());
      color_images.push_back(std::make_shared<ColorImage>(w, h));
      color_images[i]->CopyFromHost(colors.data());
    }
  }
  frame.depth_image = depth_images[0];
  frame.color_image = color_images[0];
  frame.normal_image = std::make_shared<ColorImage>(w, h);

  auto keyframe = std::make_shared<Frame>();
  keyframe->depth_projection = keyframe->color_projection = frame.depth_projection;
  keyframe->depth_image = std::make_shared<Image>(w, h);
  keyframe->color_image = std::make_shared<ColorImage>(w, h);
  keyframe->normal_image = std::make_shared<ColorImage>(w, h);

  std::unique_ptr<FrameUploader> uploader;
  if (stream_input)
  {
    uploader.reset(new FrameUploader(w, h, false));
    std::copy(depth.begin(), depth.end(), uploader->StagingDepth());    // frame 0 on its way before the clock starts
    uploader->Submit();
  }

  std::vector<int> steps_histogram(64, 0);
  std::vector<int> steps_run;
  double worst_translation = 0, worst_rotation = 0, last_translation = 0, last_rotation = 0;
  frame.depth_to_world_transform = truth[0];      // the first frame defines the map

  Device::Synchronize();
  const auto t0 = std::chrono::steady_clock::now();

  // the same loop without its cold start (frame 0 allocates the whole first view: a 200 us request pass, the table's first
  // fill): an event behind frame `warm - 1` and one behind the last frame, on the classes' stream
  // mode 1: SetView enqueued behind the Track at the pose on the device (round 6); the environment switch is read once, here
  const bool set_view_with_track = mode == 1 && !(std::getenv("VK_APP_SET_VIEW_AFTER_TRACK") && std::getenv("VK_APP_SET_VIEW_AFTER_TRACK")[0] == '1');
  bool set_view_done = false;
  const int warm = 20;
  void* steady_from = nullptr;
  void* steady_to = nullptr;
  VK_ASSERT(vk_event_create(&steady_from));
  VK_ASSERT(vk_event_create(&steady_to));

  for (int i = 0; i < frames; ++i)
  {
    if (i == warm) VK_ASSERT(vk_event_record(steady_from, Device::GetStream()));
    if (track)
    {
      frame.depth_image = depth_images[i];
      frame.color_image = color_images[i];
    }
    if (stream_input)
    {
      // frame i + 1 crosses the bus while frame i is fused. The staging buffers were filled before the clock started
      // (a camera driver writes them by DMA; copying 1.2 MB with the CPU here would time the CPU): both hold the frame
      if (i + 1 < frames)
      {
        if (i + 1 < FrameUploader::slot_count) std::copy(depth.begin(), depth.end(), uploader->StagingDepth());
        else uploader->StagingDepth();             // waits until the buffer's last copy has left it, as a writer would
        uploader->Submit();
      }
      uploader->Acquire(frame, volume->GetRequestStream());
    }
    // vulcan.cu:297 (DepthIntegrator alone needs none; with the pyramid depth tracker they are computed by its first launch)
    if (mode != 0 && !(mode == 1 && i > 0)) frame.ComputeNormals();

    if (track && i > 0)
    {
      // start from the previous pose, refine against the raycast keyframe (vulcan.cu:300-311)
      int run = 0;
      if (mode == 3) { app_tracker.SetKeyframe(keyframe); app_tracker.Track(frame); run = app_tracker.GetIterationsRun(); }
      else if (mode == 2) { light_tracker.SetKeyframe(keyframe); light_tracker.Track(frame); run = light_tracker.GetTracker()->GetIterationsRun(); }
      // (mode 1: the raycast below leaves its normal image to this call — Tracer::TraceWithoutNormals — one launch less per frame)
      // (round 6: and the SetView below is enqueued behind the Track, at the pose on the device — ComputeNormalsTrackAndSetView;
      // VK_APP_SET_VIEW_AFTER_TRACK=1: the three calls, as until round 5)
      else if (set_view_with_track) { depth_tracker.SetKeyframe(keyframe); depth_tracker.ComputeNormalsTrackAndSetView(frame, *volume, 3, true); run = depth_tracker.GetTracker()->GetIterationsRun(); set_view_done = true; }
      else { depth_tracker.SetKeyframe(keyframe); depth_tracker.ComputeNormalsAndTrack(frame, true); run = depth_tracker.GetTracker()->GetIterationsRun(); }
      steps_run.push_back(run);
      if (run >= 0 && run < 64) ++steps_histogram[run];
      PoseError(frame.depth_to_world_transform, truth[i], last_translation, last_rotation);
      worst_translation = std::max(worst_translation, last_translation);
      worst_rotation = std::max(worst_rotation, last_rotation);
    }
    else if (!track)
    {
      frame.depth_to_world_transform = truth[i];
    }

    if (!set_view_done) volume->SetView(frame, 3);         // vulcan.cu:316-318: three SetView calls
    set_view_done = false;
    if (photometric) light_integrator.Integrate(frame); else depth_integrator.Integrate(frame);   // :321
    if (stream_input) uploader->Release();         // the frame's images have no reader after Integrate
    keyframe->depth_to_world_transform = frame.depth_to_world_transform;
    if (requests_ahead && i + 1 < frames)
    {
      Frame next = frame;              // the same resident images, the next pose
      next.depth_to_world_transform = truth[i + 1];
      tracer.Trace(*keyframe, next);   // :325, + the request pass of the next SetView
    }
    else if (track && mode == 1 && i + 1 < frames) tracer.TraceWithoutNormals(*keyframe);   // :325; the normals: the next Track's launch
    else tracer.Trace(*keyframe);      // :325
  }

  // how long the host needed to ENQUEUE the frames (mode 0 never waits for the device inside the loop): when this is
  // close to the total, the loop is bound by the host's launch calls, not by the kernels
  const double enqueue_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (frames > warm) VK_ASSERT(vk_event_record(steady_to, Device::GetStream()));
  Device::Synchronize();
  const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  int32_t counters[VK_CTR_PUBLIC];
  volume->GetCounters(counters);
  // blocks of the pool in use: the free-slot pointer keeps falling below -1 once the pool is empty (as upstream's,
  // src/volume.cu:352-356), so capacity - 1 - pointer overshoots the capacity by the dropped requests; VERDICT r5 weak #9
  const int capacity = 65024 + excess_blocks;
  const int allocated = std::min(capacity, capacity - 1 - counters[VK_CTR_VOXEL_PTR]);
  const bool pool_exhausted = counters[VK_CTR_DROPPED] > 0;
  if (pool_exhausted)
    std::printf("POOL EXHAUSTED: %d allocation requests were dropped — the map is missing those blocks and the figures below are "
        "not those of the configured workload\n", counters[VK_CTR_DROPPED]);
  std::printf("frames %d  time %.3f s  fps %.1f  visible %d  allocated %d  dropped %d  input %s  tracking %s\n", frames,
      seconds, frames / seconds, counters[VK_CTR_VISIBLE], allocated,
      counters[VK_CTR_DROPPED], stream_input ? (split_streams ? "uploaded per frame, requests on their own stream" : "uploaded per frame") :
          (split_streams ? "resident, requests on their own stream" : (requests_ahead ? "resident, requests made ahead" : "resident")), mode == 0 ? "off" : (mode == 1 ? "depth" : (mode == 2 ? "light (pyramid)" : "light (app)")));
  const Matrix4f M = frame.depth_to_world_transform.GetMatrix();
  std::printf("host enqueue %.1f us per frame of %.1f\n", 1e6 * enqueue_seconds / frames, 1e6 * seconds / frames);
  if (frames > warm)
  {
    float steady_ms = 0;
    VK_ASSERT(vk_event_elapsed_ms(steady_from, steady_to, &steady_ms));
    std::printf("steady %.1f us per frame = %.1f frames/s (frames %d..%d between two events on the stream: without the cold start)\n",
        1e3 * steady_ms / (frames - warm), (frames - warm) / (1e-3 * steady_ms), warm, frames - 1);
  }
  (void)vk_event_destroy(steady_from);
  (void)vk_event_destroy(steady_to);
  std::printf("final pose row0: %.5f %.5f %.5f %.5f\n", M(0, 0), M(0, 1), M(0, 2), M(0, 3));

  double motion_translation = 0, motion_rotation = 0;
  PoseError(truth[frames - 1], truth[0], motion_translation, motion_rotation);
  int median = 0;
  if (!steps_run.empty())
  {
    std::sort(steps_run.begin(), steps_run.end());
    median = steps_run[steps_run.size() / 2];
  }
  std::printf("{\"app\": \"fuse_sequence\", \"mode\": %d, \"tracker\": \"%s\", \"integrator\": \"%s\", \"frames\": %d, "
      "\"frames_per_s\": %.1f, \"us_per_frame\": %.1f, \"set_view_rounds_run_per_frame\": %.4f, \"visible_blocks_last\": %d, "
      "\"allocated_blocks\": %d, \"dropped_requests\": %d, \"pool_exhausted\": %s, \"tracked_pose_drives_fusion\": %s, "
      "\"pose_error_max\": {\"translation_m\": %.6f, \"rotation_deg\": %.5f}, "
      "\"pose_error_last_frame\": {\"translation_m\": %.6f, \"rotation_deg\": %.5f}, "
      "\"camera_motion_over_run\": {\"translation_m\": %.4f, \"rotation_deg\": %.3f}, \"gn_steps_median\": %d, "
      "\"gn_steps_histogram_full_resolution_level\": {",
      mode, mode == 0 ? "none" : (mode == 1 ? "PyramidTracker<DepthTracker>" : (mode == 2 ? "PyramidTracker<LightTracker>" :
          "LightTracker, SetMaxIterations(1) (apps/vulcan/vulcan.cu:106-111)")),
      mode == 3 ? "LightIntegrator, weight caps 100 / 16 (vulcan.cu:92-93)" : (mode == 2 ? "LightIntegrator" : "DepthIntegrator"), frames, frames / seconds, 1e6 * seconds / frames,
      double(counters[VK_CTR_ROUNDS]) / frames, counters[VK_CTR_VISIBLE], allocated,
      counters[VK_CTR_DROPPED], pool_exhausted ? "true" : "false", track ? "true" : "false", worst_translation, worst_rotation, last_translation,
      last_rotation, motion_translation, motion_rotation, median);
  bool first = true;
  for (int n = 0; n < 64; ++n)
    if (steps_histogram[n]) { std::printf("%s\"%d\": %d", first ? "" : ", ", n, steps_histogram[n]); first = false; }
  std::printf("}}\n");
  return 0;
}
