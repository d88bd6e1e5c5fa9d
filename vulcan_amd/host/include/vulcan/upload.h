// upload.h — the input side of a frame: camera images uploaded while the frame before them is being fused.
//
// Upstream uploads every frame with a blocking copy (ref: include/vulcan/image.h:100-123 — Image::Load converts with
// OpenCV and ends in cudaMemcpy — called at apps/vulcan/vulcan.cu:220,232): at 640x480 that is 1.2 MB of depth and
// 3.7 MB of colour per frame, about the 100 us the whole fusion + raycast step takes here. FrameUploader overlaps
// the two: pinned staging buffers the camera driver (or a decoder) writes into, a copy stream of its own, and four
// slots of device images, so that frame i+1 crosses the bus while frame i is fused.
//
//   FrameUploader uploader(640, 480, true);
//   fill(uploader.StagingDepth(), uploader.StagingColor());  uploader.Submit();       // frame 0 on its way
//   for (...) {
//     fill(uploader.StagingDepth(), uploader.StagingColor());  uploader.Submit();     // frame i+1 on its way
//     uploader.Acquire(frame);           // frame i's images; the compute stream waits for their upload
//     frame.ComputeNormals(); volume->SetView(frame, 3); integrator.Integrate(frame);
//     uploader.Release();                // everything that reads frame i's images has been enqueued
//     tracer.Trace(*keyframe);
//   }
//
// No reference counterpart for the overlap itself; the images and the Frame are upstream's.
#pragma once

#include <memory>
#include <vulcan/device.h>
#include <vulcan/image.h>
#include <vulcan/observation.h>

namespace vulcan
{

class FrameUploader
{
  public:

    // The copy stream never waits for the compute stream ON THE DEVICE (with such a wait in front of a copy the runtime
    // serialised it with the frame's kernels: 357 us per frame instead of 114). The HOST waits instead, in Submit(), for
    // the readers of the slot it is about to overwrite — slot_count frames back, so it can run slot_count - 1 frames ahead.
    static const int slot_count = 4;

    FrameUploader(int width, int height, bool with_color);

    ~FrameUploader();

    // pinned host memory of the slot the next Submit() sends: [width * height] floats / Vector3f. The call blocks
    // until the copy that last read this buffer (two Submits ago) has left it.
    float* StagingDepth();

    Vector3f* StagingColor();

    // enqueue the staged frame's upload on the copy stream. Returns at once unless the slot's last readers (slot_count
    // frames back) are still running. At most slot_count frames may be submitted and not yet released.
    void Submit();

    // the oldest submitted frame: `frame.depth_image` / `color_image` become that slot's device images (stamped as
    // new content) and Device::GetStream() waits for their upload — and `second_stream` too, when given (a Volume's
    // request stream, Volume::GetRequestStream(), reads the depth image as well)
    void Acquire(Frame& frame, void* second_stream = nullptr);

    // the acquired frame's readers are all enqueued on Device::GetStream(): its slot may be overwritten once they ran
    void Release();

    int GetSubmitted() const { return submitted_; }

  private:

    FrameUploader(const FrameUploader&);

    FrameUploader& operator=(const FrameUploader&);

    int width_, height_;

    bool with_color_;

    void* copy_stream_;

    float* staging_depth_[slot_count];

    Vector3f* staging_color_[slot_count];

    std::shared_ptr<Image> depth_[slot_count];

    std::shared_ptr<ColorImage> color_[slot_count];

    void* uploaded_[slot_count];   // recorded on the copy stream behind a slot's copies
    void* consumed_[slot_count];   // recorded on the compute stream behind a slot's readers

    bool consumed_recorded_[slot_count], uploaded_recorded_[slot_count];

    int submitted_, acquired_, released_;   // running counts; slot = count % slot_count
};

} // namespace vulcan
