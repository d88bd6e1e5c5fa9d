// tsdf_volume.h — the voxel-hashed TSDF volume and the per-entry flag enums.
//
// API parity: class Volume keeps the public AND protected surface of the
// reference's volume.h:16-117 (its tests subclass Volume to reach the four
// SetView stages, tests/volume_test.cpp:23-31); Visibility / AllocationType
// keep the values of types.h. Both header names remain as forwarders.
// What differs behind the names:
//   * the device counters (visible count, free-slot pointer, excess pointer)
//     live in a per-volume buffer instead of process-wide __device__ symbols
//     (volume.cu:17-21), so one process can hold several volumes;
//   * SetView never blocks; the visible count is read back lazily by the first
//     GetVisibleBlocks() after it (volume.cu:494 synchronised every frame).
#pragma once

#include <cstdint>
#include <vk.h>
#include <vulcan/buffer.h>
#include <vulcan/matrix.h>

namespace vulcan
{

// per hash entry: was the block seen by the current view
enum Visibility : uint8_t
{
  VISIBILITY_UNKNOWN = 0,
  VISIBILITY_FALSE   = 1,
  VISIBILITY_TRUE    = 2,
};

// per hash entry: what kind of allocation the current view requested there
enum AllocationType : uint8_t
{
  ALLOC_TYPE_NONE   = 0,
  ALLOC_TYPE_MAIN   = 1,
  ALLOC_TYPE_EXCESS = 2,
};

class Block;
struct Frame;
class HashEntry;
class Voxel;

class Volume
{
  public:
    Volume(int main_block_count, int excess_block_count);
    virtual ~Volume();

    // ---- geometry ----
    int GetMainBlockCount() const;
    int GetExcessBlockCount() const;
    float GetVoxelLength() const;
    void SetVoxelLength(float length);
    float GetTruncationLength() const;
    void SetTruncationLength(float length);
    const Vector2f& GetDepthRange() const;
    void SetDepthRange(const Vector2f& range);
    void SetDepthRange(float min, float max);

    // ---- per frame: allocate what the depth image touches, list what is visible ----
    void SetView(const Frame& frame);
    // `rounds` consecutive SetView(frame) calls (the app makes three per frame, vulcan.cu:316-318)
    // in one: the later rounds run on the device, and only when the round before lost a request
    void SetView(const Frame& frame, int rounds);
    // frame.ComputeNormals(); SetView(frame, rounds); in one call (not upstream): with a LightIntegrator
    // attached the normals are computed inside SetView's request pass (vk_light_prep.normals_out)
    void ComputeNormalsAndSetView(Frame& frame, int rounds = 1);

    // ---- storage ----
    const Buffer<HashEntry>& GetHashEntries() const;
    const Buffer<int>& GetAllocatedBlocks() const;
    const Buffer<int>& GetVisibleBlocks() const;   // first call after SetView syncs
    const Buffer<Voxel>& GetVoxels() const;
    Buffer<Voxel>& GetVoxels();

    vk_volume ToVk() const;                        // device view for the C ABI
    void GetCounters(int32_t* counters) const;     // blocking readback of the VK_CTR_PUBLIC counters
    // Not upstream: blocks of the pool in use, min(capacity, capacity - 1 - free-slot pointer) (blocking readback). The
    // pointer itself keeps falling once the pool is empty (upstream's does too, src/volume.cu:352-356): VK_CTR_DROPPED
    // counts the requests that found it empty.
    int GetAllocatedBlockCount() const;
    // Not upstream (round 6): SetView(frame, rounds) at a pose that is still ON THE DEVICE — the vk_transform a tracker's
    // launches in front of this call leave there — so that a tracking loop can enqueue SetView behind its Track before it waits
    // for the pose (vk_volume_set_view_at_device_pose; PyramidTracker<DepthTracker>::ComputeNormalsTrackAndSetView uses it).
    // frame.depth_to_world_transform is ignored. false: not possible in this state (a frame announced by Tracer::Trace, a
    // request stream, the three-launch test form) — nothing was launched, the caller calls SetView once it has the pose.
    bool SetViewAtDevicePose(const Frame& frame, const vk_transform* pose_device, int rounds = 1);

    // Raycast bounds prepared ahead of time (vk_view_bounds, not upstream): a Tracer
    // registers its scratch buffer and settings here, the integrators then compute
    // the bounds of the view they integrate inside their own launch and
    // Tracer::Trace skips that pass when it raycasts the same view. SetView (a new
    // visible list) invalidates the record. nullptr while no Tracer is attached.
    vk_view_bounds* GetViewBounds() const;
    void AttachViewBounds(float* scratch, int bounds_width, int bounds_height, const Vector2f& depth_range) const;
    void DetachViewBounds(const float* scratch) const;   // no-op unless `scratch` is the attached one

    // LightIntegrator's per-pixel preparation, ahead of time (vk_light_prep, not upstream): a
    // LightIntegrator registers its mask / record buffers here and SetView fills them in its own
    // request pass, for the Integrate of the same frame that follows (the frame must not be
    // modified in between). nullptr while nothing is attached.
    vk_light_prep* GetLightPreparation() const;
    void AttachLightPreparation(float* mask, float* records, int capacity_pixels, float depth_threshold) const;
    void DetachLightPreparation(const float* mask) const;   // no-op unless `mask` is the attached one

    // Not upstream (which runs everything on stream 0): SetView's request pass on a stream of its own
    // (vk_volume_set_view_rounds_split), so that it runs beside the PREVIOUS frame's Tracer::Trace instead of behind it. For
    // callers that know a frame's pose before the previous frame's raycast has finished — fusion at given poses; a tracking
    // loop gains nothing, its pose comes out of that raycast. The request pass then waits only for the previous Integrate
    // (the integrators call NoteIntegrated): whatever it reads — the frame's depth image, and colour / normal images when a
    // LightIntegrator's preparation rides along — must be complete without work enqueued on Device::GetStream() after that
    // Integrate (ComputeNormalsAndSetView computes the normals inside the pass; FrameUploader::Acquire takes the stream to wait on).
    void EnableRequestStream();
    void* GetRequestStream() const { return request_stream_; }
    void NoteIntegrated() const;

    // Not upstream: the record of a request pass made AHEAD of its SetView (vk_requests_ahead) — by
    // Tracer::Trace(keyframe, next_frame), inside the raycast's own launch. SetView(next_frame) then launches only its
    // handle + visibility pass; any other SetView while the record is valid throws (the announced frame's requests
    // are in the volume and have to be handled first).
    vk_requests_ahead* GetRequestsAhead() const { return &requests_ahead_; }
    // The way out of an announced frame that will not be fused as announced (vk_requests_ahead_cancel): its SetView is
    // completed from the record — `rounds` as in SetView — and any frame may follow. No-op without a valid record.
    void CancelRequestsAhead(int rounds = 1);

  protected:
    // the four stages of SetView, in call order
    void ResetBlockVisibility();
    void CreateAllocationRequests(const Frame& frame);
    void HandleAllocationRequests();
    void UpdateBlockVisibility(const Frame& frame);

    int GetBufferSize() const;
    void ResetBufferSize() const;

    Buffer<Voxel> voxels_;
    Buffer<HashEntry> hash_entries_;
    Buffer<int> free_voxel_blocks_;
    Buffer<AllocationType> allocation_types_;
    Buffer<Block> allocation_blocks_;
    Buffer<Visibility> block_visibility_;
    mutable Buffer<int> visible_blocks_;
    Buffer<int> counters_;

    Vector2f depth_range_;
    int max_block_count_;
    int main_block_count_;
    int excess_block_count_;
    float truncation_length_;
    float voxel_length_;
    bool empty_;
    mutable bool visible_count_stale_;
    mutable vk_view_bounds view_bounds_;
    mutable vk_light_prep light_prep_;
    mutable vk_requests_ahead requests_ahead_;
    void* request_stream_;              // EnableRequestStream(): nullptr = everything on Device::GetStream()
    void* requested_;                   // event behind the request pass
    void* integrated_;                  // event behind the last Integrate
    mutable bool integrated_recorded_;
    mutable bool pool_exhaustion_noted_ = false;
    void NotePoolExhaustion(int32_t dropped) const;
    mutable int32_t* normals_late_;     // pinned: vk_view_bounds.late_host of the attached tracer's record

  private:
    void Initialize();
    Volume(const Volume&);              // not copyable
    Volume& operator=(const Volume&);
};

} // namespace vulcan
