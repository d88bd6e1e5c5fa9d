// raycast.h — renders the fused volume back into a depth / colour / normal
// frame. One header for the Tracer class, its Patch record and the stage
// functions on raw device pointers that the white-box tests call.
//
// API parity: names follow the reference's tracer.h (class Tracer, struct
// Patch) and tracer.cuh (the five free functions); both header names remain as
// forwarders. Trace() itself goes through vk_trace, whose bounds stage works
// per block instead of per patch; the protected four-stage form is kept.
#pragma once

#include <memory>
#include <vulcan/buffer.h>
#include <vulcan/matrix.h>

namespace vulcan
{

struct Frame;
class HashEntry;
class Projection;
class Transform;
class Volume;
class Voxel;

// Screen-space footprint of (part of) a visible block: at most 16x16 cells of
// the 8x-downsampled bounds grid, with the block's depth interval.
struct Patch
{
  static const int max_size = 16;
  Vector2s origin;   // first bounds cell
  Vector2s size;     // cells covered
  Vector2f bounds;   // (near, far) depth of the block's corners
};

class Tracer
{
  public:
    explicit Tracer(std::shared_ptr<const Volume> volume);
    virtual ~Tracer();

    std::shared_ptr<const Volume> GetVolume() const;

    const Vector2f& GetDepthRange() const;
    void SetDepthRange(const Vector2f& range);
    void SetDepthRange(float min, float max);

    // fills frame.depth_image, color_image and normal_image from the pose and
    // intrinsics already set on the frame
    void Trace(Frame& frame);

    // Not upstream: Trace(frame), and in the same launch the request pass of the volume's NEXT SetView — for a caller that
    // knows the next frame (images and pose) while it raycasts this one: fusion at given poses, not a tracking loop, whose
    // next pose comes out of this raycast. `next_frame` must then be the very next frame passed to Volume::SetView,
    // unchanged (any other SetView throws). With `next_needs_normals` the pass also computes next_frame's normal image
    // (Frame::ComputeNormals), which needs a LightIntegrator attached to the volume (its preparation rides along, as in
    // Volume::ComputeNormalsAndSetView); without one the normals are computed by a launch of their own first.
    // The results are those of Trace(frame); ...; SetView(next_frame), bit for bit (vk_trace_ahead_requests).
    void Trace(Frame& frame, Frame& next_frame, bool next_needs_normals = false);

    // Not upstream: Trace(frame) without its last stage (tracer.cpp:97-100 ComputeNormals): depth and colour images only.
    // For the tracking loop, where the raycast's normal image is read next by the tracker of the following frame:
    // PyramidTracker<DepthTracker>::ComputeNormalsAndTrack(frame, /*keyframe_normals_due*/ true) computes it — the same
    // bits — in the launch that builds its pyramid, one launch less per frame. The normal image is allocated; whoever
    // reads it before that Track calls frame.ComputeNormals().
    void TraceWithoutNormals(Frame& frame);

    // Not upstream: the normals of Trace(frame, next_frame) are computed by workgroups of the raycast's own launch that
    // WAIT for the depths they need — a bounded wait. SettleNormals synchronises the stream and throws (upstream: a failed
    // device step always throws, device.h:14-17) if a wait expired; the normal image has then been recomputed by a launch
    // of its own. The next Trace makes the same check without synchronising (vk_trace_normals_settle).
    void SettleNormals();

  protected:
    void ComputePatches(const Frame& frame);
    void ComputeBounds(const Frame& frame);
    void ComputePoints(Frame& frame);
    void ComputeNormals(Frame& frame);
    void ResetBoundsBuffer();
    void ResetBufferSize();
    int GetBufferSize();

    Buffer<Patch> patches_;
    Buffer<Vector2f> bounds_;       // bounds grid, then the fused pass's scratch
    Buffer<int> buffer_size_;       // device patch counter
    std::shared_ptr<const Volume> volume_;
    Vector2f depth_range_;
    int bounds_width_;
    int bounds_height_;

  private:
    void Initialize();
    void TraceWith(Frame& frame, Frame* next, bool next_needs_normals, bool with_normals = true);
};

// ---- stage functions (device pointers in, device pointers out) -----------------

// visible blocks -> patches; *patch_count is a device counter
void ComputePatches(const int* indices, const HashEntry* entries,
    const Transform& Tcw, const Projection& projection, float block_length,
    float min_depth, float max_depth, int block_count, int image_width,
    int image_height, int bounds_width, int bounds_height, Patch* patches,
    int* patch_count);

// patches -> per-cell (min near, max far)
void ComputeBounds(const Patch* patches, Vector2f* bounds, int bounds_width,
    int patch_count);

// (+inf, -inf) in every cell
void ResetBoundsBuffer(Vector2f* bounds, int count);

// per-pixel march between the cell's bounds; writes depth and colour
void ComputePoints(const HashEntry* entries, const Voxel* voxels,
    const Vector2f* bounds, int block_count, float block_length,
    float voxel_length, float trunc_length, const Transform& Twc,
    const Projection& projection, float* depths, Vector3f* colors,
    int image_width, int image_height, int bounds_width, int bounds_height);

// depth -> normals by central differences (also declared in observation.h)
void ComputeNormals(const float* depths, const Projection& projection,
    Vector3f* normals, int image_width, int image_height);

} // namespace vulcan
