// hash.h — one slot of the voxel-hash table, 16 bytes (ref: include/vulcan/hash.h):
// block origin, pool slot (`data`, -1 = unallocated), chain link (`next`, -1 = end).
#pragma once

#include <climits>
#include <vk.h>
#include <vulcan/block.h>

namespace vulcan
{

class HashEntry
{
  public:

    static const int invalid = -1;

  public:

    HashEntry() : data(invalid), next(invalid) {}

    bool IsAllocated() const { return data != invalid; }

    void InvalidateData() { data = invalid; }

    bool HasNext() const { return next != invalid; }

    void InvalidateNext() { next = invalid; }

  public:

    Block block;

    int data;

    int next;
};

static_assert(sizeof(HashEntry) == sizeof(vk_hash_entry), "HashEntry must match vk_hash_entry");

} // namespace vulcan
