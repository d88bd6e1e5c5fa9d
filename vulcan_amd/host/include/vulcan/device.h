// device.h — the seam between the host classes and the GPU. In the reference
// this header wraps the CUDA runtime (CUDA_ASSERT / CUDA_LAUNCH, device.h:14-54);
// here every device action is a call into the C ABI of include/vk.h and
// VK_ASSERT turns its integer status into the same vulcan::Exception.
#pragma once

#include <cstddef>
#include <string>
#include <vk.h>
#include <vulcan/exception.h>

#define VK_ASSERT(cmd) do {                                                     \
  const int vk_code__ = (cmd);                                                  \
  if (vk_code__ != VK_OK)                                                       \
    VULCAN_THROW(::vulcan::GetDeviceErrorString(vk_code__));                    \
} while (0)

namespace vulcan
{

// ref: device.h:106-110 GetCudaErrorString — "<text> [<kind> error <code>]"
inline std::string GetDeviceErrorString(int code)
{
  const std::string text = vk_error_string(code);
  if (code < 0) return text;
  return text + " [hip error " + std::to_string(code) + "]";
}

// ref: device.h:114-117
inline size_t GetKernelBlocks(size_t total, size_t threads)
{
  return (total + threads - 1) / threads;
}

// The stream every class in this process submits to (the reference uses the
// default stream 0 everywhere, device.h:40-52). One host thread per GPU.
class Device
{
  public:

    static void* GetStream() { return Current(); }

    static void SetStream(void* stream) { Current() = stream; }

    static void Synchronize() { VK_ASSERT(vk_stream_synchronize(Current())); }

  private:

    static void*& Current()
    {
      static thread_local void* stream = nullptr;
      return stream;
    }
};

} // namespace vulcan
