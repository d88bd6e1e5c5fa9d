// vk_runtime.hip — device, stream, memory and event entry points of vk.h.
// These exist so that hosts built without HIP headers (the C++ class layer in
// vulcan_amd/host, ctypes) can own device memory; ref: buffer.h:64-103,
// image.h:85-97, device.h:14-54.
#include "vk_common.hpp"

#include <string.h>
#include <time.h>

#include <atomic>
#include <mutex>
#include <unordered_map>

// Launch tags of the loop kernels (vk_gauss_newton.hpp, "partials exchanged inside a launch").
//
// One 64-bit count of loop launches for the whole library image, so that two trackers that are handed the same memory
// one after the other never use the same tag. The 22-bit epoch of launch n is 1 + n mod (2^22 - 1): never 0, and two
// launches share an epoch only if their counts differ by a multiple of 2^22 - 1 (at the tracked loop's ~7 300 loop
// launches per second: every ~9.6 minutes). A reader accepts a word whose tag is {its launch's epoch, its step}; a
// word that an EARLIER launch with the same epoch left in the same slot would be taken for the step's — a silently
// wrong pose. What excludes it:
//
//   every exchange area is cleared (hipMemsetAsync on the launch's stream, in front of the launch) when it is first
//   seen, when a launch needs more of it than was cleared, when something that is not a loop launch wrote into it
//   (the launch-per-stage loops keep float partials there: vk_loop_area_written), and whenever the count has moved
//   2^21 or more since the area's last clear.
//
// So a tagged word found in an area was written by a loop launch m with cleared_at <= m < n and n - m < 2^21
// < 2^22 - 1: its epoch differs from launch n's. (Tag 0 — cleared memory — names no launch: epochs start at 1 and the
// step field at 1.) The registry is keyed by the area's address; an address that comes back after a free holds either
// the words it held (same entry, same argument) or whatever its interim owner wrote (the caller's contract, vk.h:
// nothing but the library writes a workspace between two calls that use it). The cost is one table lookup per loop
// launch and one 0.6 MB memset per area every 2^21 launches. The count starts at a value taken from the clock and the
// library's load address: a second copy of the library in the same process (the debug build next to the release
// build) then counts from somewhere else. `VK_LOOP_EPOCH_UNGUARDED` (a build for the test's own proof that it bites:
// tests/test_gpu_epoch_wrap.py, profiles/r06_epoch_wrap.txt) leaves the clears out.
namespace
{
constexpr uint64_t kEpochPeriod = (1ull << 22) - 1;
constexpr uint64_t kAreaMaxAge = 1ull << 21;
struct LoopArea { uint64_t cleared_at; size_t cleared_bytes; bool foreign; };
struct LoopAreas
{
  std::mutex lock;
  uint64_t count;
  std::unordered_map<const void*, LoopArea> areas;
  uint64_t clears = 0;
  LoopAreas()
  {
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    uint64_t x = (uint64_t)t.tv_nsec ^ ((uint64_t)t.tv_sec << 20) ^ (uint64_t)reinterpret_cast<uintptr_t>(&vk_hook);
    x ^= x >> 33;  x *= 0xff51afd7ed558ccdull;  x ^= x >> 33;
    count = x >> 8;          // 56 bits: never reaches 2^64 in a process's life
  }
};
LoopAreas& loop_areas() { static LoopAreas a; return a; }
}

void vk_loop_area_written(const void* area);

int vk_loop_epoch_begin(void* area, size_t bytes, hipStream_t s, uint32_t* epoch)
{
  LoopAreas& A = loop_areas();
  bool clear = false;
  {
    std::lock_guard<std::mutex> hold(A.lock);
    const uint64_t n = A.count++;
    *epoch = (uint32_t)(1 + n % kEpochPeriod);
    if (A.areas.size() > 8192) A.areas.clear();        // forgotten areas are cleared at their next launch
    auto it = A.areas.find(area);
    if (it == A.areas.end() || it->second.foreign || it->second.cleared_bytes < bytes || n - it->second.cleared_at >= kAreaMaxAge)
    {
      clear = true;
      A.areas[area] = LoopArea{n, bytes, false};
      ++A.clears;
    }
  }
#ifndef VK_LOOP_EPOCH_UNGUARDED
  if (clear)
  {
    const hipError_t e = hipMemsetAsync(area, 0, bytes, s);
    if (e != hipSuccess)
    {
      // the clear was not enqueued: the area is not what the registry says — it is cleared at its next launch
      (void)hipGetLastError();
      vk_loop_area_written(area);
      return (int)e;
    }
  }
#endif
  return VK_OK;
}

void vk_loop_area_written(const void* area)
{
  LoopAreas& A = loop_areas();
  std::lock_guard<std::mutex> hold(A.lock);
  auto it = A.areas.find(area);
  if (it != A.areas.end()) it->second.foreign = true;
}

extern "C" int vk_test_hooks_loop_count(const uint64_t* set_to, uint64_t* count_now, uint64_t* clears)
{
  LoopAreas& A = loop_areas();
  std::lock_guard<std::mutex> hold(A.lock);
  if (set_to) A.count = *set_to;
  if (count_now) *count_now = A.count;
  if (clears) *clears = A.clears;
  return VK_OK;
}


// vk_test_hooks (vk.h): one copy per library image, every field an atomic of its own
namespace
{
std::atomic<int32_t> g_hooks[7] = {{-1}, {0}, {0}, {0}, {0}, {0}, {0}};
}
int vk_hook(int field) { return g_hooks[field].load(std::memory_order_relaxed); }
// a hook that fires once: its value, and 0 from then on
int vk_hook_take(int field) { return g_hooks[field].exchange(0, std::memory_order_relaxed); }

extern "C" {
int vk_test_hooks_set(const vk_test_hooks* h)
{
  const vk_test_hooks defaults = {-1, 0, 0, 0, 0, 0, 0};
  if (!h) h = &defaults;
  const int32_t v[7] = {h->posted_capacity, h->retry_capacity, h->set_view_unfused, h->force_loop_abort,
                        h->loop_grid_cap, h->loop_cooperative, h->force_normals_expiry};
  for (int i = 0; i < 7; ++i) g_hooks[i].store(v[i], std::memory_order_relaxed);
  return VK_OK;
}
int vk_test_hooks_get(vk_test_hooks* out)
{
  VK_REQUIRE(out);
  out->posted_capacity = vk_hook(0);
  out->retry_capacity = vk_hook(1);
  out->set_view_unfused = vk_hook(2);
  out->force_loop_abort = vk_hook(3);
  out->loop_grid_cap = vk_hook(4);
  out->loop_cooperative = vk_hook(5);
  out->force_normals_expiry = vk_hook(6);
  return VK_OK;
}
}

// Loop kernels (one launch per Gauss-Newton loop) need all their workgroups on the device at
// the same time, and one of them fills it. Two of them started at the same moment on two
// streams could each get a part of the device and wait for the rest — for two seconds, until
// both give up (VK_TRACK_ABORTED). Loop kernels of one process are therefore chained across
// streams: once a second stream is seen on a device, every loop launch waits for the
// previous one's event. A process that tracks on one stream per device never creates an
// event. (Two PROCESSES sharing a device are not covered: give each its own GPU.)
namespace
{
struct LoopChain
{
  std::mutex lock;
  hipStream_t last_stream = nullptr;
  bool seen = false, chained = false;
  hipEvent_t event = nullptr;
};
LoopChain g_loop_chain[16];
}

void vk_loop_launch_begin(hipStream_t s)
{
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= 16) return;
  LoopChain& c = g_loop_chain[device];
  c.lock.lock();
  if (c.seen && !c.chained && c.last_stream != s)
  {
    // a second stream: from now on launches are chained; the launches so far had no event
    c.chained = true;
    (void)hipStreamSynchronize(c.last_stream);
    (void)hipEventCreateWithFlags(&c.event, hipEventDisableTiming | hipEventDisableSystemFence);
  }
  else if (c.chained && c.event && c.last_stream != s)
  {
    (void)hipStreamWaitEvent(s, c.event, 0);
  }
}

void vk_loop_launch_end(hipStream_t s)
{
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= 16) return;
  LoopChain& c = g_loop_chain[device];
  if (c.chained && c.event) (void)hipEventRecord(c.event, s);
  c.last_stream = s;
  c.seen = true;
  c.lock.unlock();
}

extern "C" {

const char* vk_error_string(int code)
{
  switch (code)
  {
    case VK_OK: return "success";
    case VK_ERR_ARGUMENT: return "invalid argument [vk error -1]";
    case VK_ERR_UNSUPPORTED: return "unsupported [vk error -2]";
    case VK_ERR_NO_DEVICE: return "no HIP device [vk error -3]";
    case VK_ERR_TIMEOUT: return "a bounded wait inside a launch expired [vk error -6]";
    default: break;
  }
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "unknown error";
}

int vk_version(void) { return 100; }

int vk_abi_version(void) { return VK_ABI_VERSION; }

int vk_abi_check(int header_abi_version, size_t sizeof_vk_volume, size_t sizeof_vk_frame, int ctr_count)
{
  const bool same = header_abi_version == VK_ABI_VERSION && sizeof_vk_volume == sizeof(vk_volume) &&
      sizeof_vk_frame == sizeof(vk_frame) && ctr_count == VK_CTR_COUNT;
  return same ? VK_OK : VK_ERR_UNSUPPORTED;
}

int vk_device_count(int* count)
{
  VK_REQUIRE(count);
  *count = 0;
  const hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) { *count = 0; return (int)e; }
  return VK_OK;
}

int vk_set_device(int device)
{
  VK_CHECK(hipSetDevice(device));
  return VK_OK;
}

int vk_device_name(char* out, size_t bytes)
{
  VK_REQUIRE(out && bytes > 0);
  int dev = 0;
  VK_CHECK(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  VK_CHECK(hipGetDeviceProperties(&prop, dev));
  strncpy(out, prop.gcnArchName, bytes - 1);
  out[bytes - 1] = 0;
  return VK_OK;
}

int vk_stream_create(void** stream)
{
  VK_REQUIRE(stream);
  hipStream_t s;
  VK_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = s;
  return VK_OK;
}

int vk_stream_destroy(void* stream)
{
  VK_CHECK(hipStreamDestroy(vk_s(stream)));
  return VK_OK;
}

int vk_stream_synchronize(void* stream)
{
  VK_CHECK(hipStreamSynchronize(vk_s(stream)));
  return VK_OK;
}

int vk_malloc(void** ptr, size_t bytes)
{
  VK_REQUIRE(ptr);
  *ptr = nullptr;
  if (bytes == 0) return VK_OK;
  VK_CHECK(hipMalloc(ptr, bytes));
  return VK_OK;
}

int vk_free(void* ptr)
{
  if (!ptr) return VK_OK;
  VK_CHECK(hipFree(ptr));
  return VK_OK;
}

int vk_malloc_host(void** ptr, size_t bytes)
{
  VK_REQUIRE(ptr);
  *ptr = nullptr;
  if (bytes == 0) return VK_OK;
  VK_CHECK(hipHostMalloc(ptr, bytes, hipHostMallocMapped | hipHostMallocCoherent));
  return VK_OK;
}

int vk_free_host(void* ptr)
{
  if (!ptr) return VK_OK;
  VK_CHECK(hipHostFree(ptr));
  return VK_OK;
}

int vk_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream)
{
  if (bytes == 0) return VK_OK;
  VK_REQUIRE(dst && src);
  VK_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, vk_s(stream)));
  VK_CHECK(hipStreamSynchronize(vk_s(stream)));
  return VK_OK;
}

int vk_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream)
{
  if (bytes == 0) return VK_OK;
  VK_REQUIRE(dst && src);
  VK_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, vk_s(stream)));
  VK_CHECK(hipStreamSynchronize(vk_s(stream)));
  return VK_OK;
}

int vk_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream)
{
  if (bytes == 0) return VK_OK;
  VK_REQUIRE(dst && src);
  VK_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, vk_s(stream)));
  return VK_OK;
}

int vk_memset(void* dst, int value, size_t bytes, void* stream)
{
  if (bytes == 0) return VK_OK;
  VK_REQUIRE(dst);
  VK_CHECK(hipMemsetAsync(dst, value, bytes, vk_s(stream)));
  return VK_OK;
}

int vk_event_create(void** event)
{
  VK_REQUIRE(event);
  hipEvent_t e;
  // timing events for kernels of one stream: no system-scope release (cache flush) at the record
  VK_CHECK(hipEventCreateWithFlags(&e, hipEventDisableSystemFence));
  *event = e;
  return VK_OK;
}

int vk_event_destroy(void* event)
{
  VK_CHECK(hipEventDestroy(reinterpret_cast<hipEvent_t>(event)));
  return VK_OK;
}

int vk_event_record(void* event, void* stream)
{
  VK_CHECK(hipEventRecord(reinterpret_cast<hipEvent_t>(event), vk_s(stream)));
  return VK_OK;
}

int vk_memcpy_h2d_async(void* dst, const void* src_pinned, size_t bytes, void* stream)
{
  if (bytes == 0) return VK_OK;
  VK_REQUIRE(dst && src_pinned);
  VK_CHECK(hipMemcpyAsync(dst, src_pinned, bytes, hipMemcpyHostToDevice, vk_s(stream)));
  return VK_OK;
}

int vk_event_create_ordering(void** event, int publishes)
{
  VK_REQUIRE(event);
  hipEvent_t e;
  VK_CHECK(hipEventCreateWithFlags(&e, publishes ? hipEventDisableTiming : (hipEventDisableTiming | hipEventDisableSystemFence)));
  *event = e;
  return VK_OK;
}

int vk_stream_wait_event(void* stream, void* event)
{
  VK_REQUIRE(event);
  VK_CHECK(hipStreamWaitEvent(vk_s(stream), reinterpret_cast<hipEvent_t>(event), 0));
  return VK_OK;
}

int vk_event_synchronize(void* event)
{
  VK_REQUIRE(event);
  VK_CHECK(hipEventSynchronize(reinterpret_cast<hipEvent_t>(event)));
  return VK_OK;
}

int vk_event_elapsed_ms(void* start, void* stop, float* ms)
{
  VK_REQUIRE(ms);
  VK_CHECK(hipEventSynchronize(reinterpret_cast<hipEvent_t>(stop)));
  VK_CHECK(hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop)));
  return VK_OK;
}

}  // extern "C"
