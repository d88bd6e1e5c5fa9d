// block.h — integer origin of an 8x8x8 voxel block, 8 bytes
// (ref: include/vulcan/block.h; the trailing short is padding).
#pragma once

#include <vk.h>
#include <vulcan/matrix.h>

namespace vulcan
{

class Block
{
  public:

    static const int resolution = VK_BLOCK_RESOLUTION;

    static const int voxel_count = VK_BLOCK_VOXELS;

  public:

    Block() : origin_(0, 0, 0), pad_(0) {}

    Block(short x, short y, short z) : origin_(x, y, z), pad_(0) {}

    Block(const Vector3s& origin) : origin_(origin), pad_(0) {}

    const Vector3s& GetOrigin() const { return origin_; }

    bool operator==(const Block& block) const { return origin_ == block.origin_; }

    bool operator!=(const Block& block) const { return !(origin_ == block.origin_); }

    const short& operator[](int index) const
    {
      VULCAN_DEBUG(index < 3);
      return origin_[index];
    }

    short& operator[](int index)
    {
      VULCAN_DEBUG(index < 3);
      return origin_[index];
    }

  protected:

    Vector3s origin_;

  private:

    short pad_;
};

static_assert(sizeof(Block) == sizeof(vk_block), "Block must match vk_block");

} // namespace vulcan
