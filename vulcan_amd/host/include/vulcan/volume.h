// volume.h — voxel-hashed TSDF volume (ref: include/vulcan/volume.h:16-117).
// Same public and protected surface; the protected members are part of the
// de-facto API because the reference's tests subclass Volume to reach them
// (tests/volume_test.cpp:23-31). Differences, all behind the same names:
//   * the device counters (visible count, free-slot pointer, excess pointer)
//     are per volume (counters_) instead of process-wide __device__ symbols
//     (volume.cu:17-21), so several volumes can live on one device;
//   * SetView does not read the visible count back; GetVisibleBlocks() syncs
//     lazily on first use after a SetView (volume.cu:494 synced every call).
#pragma once

#include <cstdint>
#include <vk.h>
#include <vulcan/buffer.h>
#include <vulcan/matrix.h>
#include <vulcan/types.h>

namespace vulcan
{

class Block;
struct Frame;
class HashEntry;
class Voxel;

class Volume
{
  public:

    Volume(int main_block_count, int excess_block_count);

    virtual ~Volume();

    int GetMainBlockCount() const;

    int GetExcessBlockCount() const;

    const Vector2f& GetDepthRange() const;

    void SetDepthRange(const Vector2f& range);

    void SetDepthRange(float min, float max);

    float GetVoxelLength() const;

    void SetVoxelLength(float length);

    float GetTruncationLength() const;

    void SetTruncationLength(float length);

    void SetView(const Frame& frame);

    const Buffer<HashEntry>& GetHashEntries() const;

    const Buffer<int>& GetAllocatedBlocks() const;

    const Buffer<int>& GetVisibleBlocks() const;

    const Buffer<Voxel>& GetVoxels() const;

    Buffer<Voxel>& GetVoxels();

    // device view for the C ABI; the visible count is read on the device
    vk_volume ToVk() const;

    // blocking readback of the VK_CTR_* counters
    void GetCounters(int32_t* counters) const;

  protected:

    void ResetBlockVisibility();

    void UpdateBlockVisibility(const Frame& frame);

    void CreateAllocationRequests(const Frame& frame);

    void HandleAllocationRequests();

    int GetBufferSize() const;

    void ResetBufferSize() const;

  private:

    void Initialize();

    Volume(const Volume&);

    Volume& operator=(const Volume&);

  protected:

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
};

} // namespace vulcan
