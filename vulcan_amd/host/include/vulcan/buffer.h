// buffer.h — owning, non-copyable device array (ref: include/vulcan/buffer.h).
// Growth discards the old contents, exactly like the reference (free then
// allocate, buffer.h:64-73). Memory comes from vk_malloc / vk_free.
#pragma once

#include <cstddef>
#include <vulcan/device.h>

namespace vulcan
{

template <typename T>
class Buffer
{
  public:

    Buffer() : data_(nullptr), capacity_(0), size_(0) {}

    Buffer(size_t size) : data_(nullptr), capacity_(0), size_(0) { Resize(size); }

    ~Buffer() { vk_free(data_); }

    const T* GetData() const { return data_; }

    T* GetData() { return data_; }

    size_t GetSize() const { return size_; }

    size_t GetCapacity() const { return capacity_; }

    size_t IsEmpty() const { return size_ == 0; }

    void Resize(size_t size)
    {
      if (size > capacity_) Reserve(size);
      size_ = size;
    }

    void Reserve(size_t capacity)
    {
      if (capacity <= capacity_) return;
      VK_ASSERT(vk_free(data_));
      data_ = nullptr;
      void* ptr = nullptr;
      VK_ASSERT(vk_malloc(&ptr, sizeof(T) * capacity));
      data_ = static_cast<T*>(ptr);
      capacity_ = capacity;
    }

    void CopyFromDevice(T* buffer)
    {
      VK_ASSERT(vk_memcpy_d2d(data_, buffer, sizeof(T) * size_, Device::GetStream()));
    }

    void CopyToDevice(T* buffer) const
    {
      VK_ASSERT(vk_memcpy_d2d(buffer, data_, sizeof(T) * size_, Device::GetStream()));
    }

    void CopyFromHost(const T* buffer)
    {
      VK_ASSERT(vk_memcpy_h2d(data_, buffer, sizeof(T) * size_, Device::GetStream()));
    }

    void CopyToHost(T* buffer) const
    {
      VK_ASSERT(vk_memcpy_d2h(buffer, data_, sizeof(T) * size_, Device::GetStream()));
    }

  private:

    Buffer(const Buffer& buffer);

    Buffer& operator=(const Buffer& buffer);

  protected:

    T* data_;

    size_t capacity_;

    size_t size_;
};

} // namespace vulcan
