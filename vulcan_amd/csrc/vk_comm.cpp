// vk_comm.cpp — libvk_comm.so: the rig's all-reduce over RCCL (include/vk_comm.h).
// Plain C++ (no device code). RCCL is bound with dlopen/dlsym so that the library
// has no link-time dependency on it.
#include "../../include/vk_comm.h"

#include <dlfcn.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

namespace
{

enum { VK_OK_ = 0, VK_ERR_ARGUMENT_ = -1 };

// the part of rccl.h this file needs (rccl.h:40-43, 187, 220, 260, 339 and ncclAllReduce)
struct UniqueId { char internal[VK_COMM_ID_BYTES]; };
typedef void* Comm;
typedef int Result;               // ncclResult_t, ncclSuccess == 0
enum { kFloat32 = 7, kSum = 0, kMin = 3 };  // ncclFloat32, ncclSum, ncclMin

struct Rccl
{
  void* handle = nullptr;
  Result (*GetUniqueId)(UniqueId*) = nullptr;
  Result (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  Result (*AllReduce)(const void*, void*, size_t, int, int, Comm, void*) = nullptr;
  Result (*AllGather)(const void*, void*, size_t, int, Comm, void*) = nullptr;
  Result (*CommDestroy)(Comm) = nullptr;
  Result (*CommCount)(Comm, int*) = nullptr;
  const char* (*GetErrorString)(Result) = nullptr;
  bool ok = false;
};

Rccl g_rccl;
std::once_flag g_once;

void load_rccl()
{
  const char* names[] = { getenv("VK_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
  for (const char* name : names)
  {
    if (!name || !*name) continue;
    g_rccl.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (g_rccl.handle) break;
  }
  if (!g_rccl.handle) return;
  g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(g_rccl.handle, "ncclGetUniqueId"));
  g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(g_rccl.handle, "ncclCommInitRank"));
  g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(dlsym(g_rccl.handle, "ncclAllReduce"));
  g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(dlsym(g_rccl.handle, "ncclAllGather"));
  g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(g_rccl.handle, "ncclCommDestroy"));
  g_rccl.CommCount = reinterpret_cast<decltype(g_rccl.CommCount)>(dlsym(g_rccl.handle, "ncclCommCount"));
  g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(g_rccl.handle, "ncclGetErrorString"));
  g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.AllReduce && g_rccl.CommDestroy;
}

const Rccl* rccl()
{
  std::call_once(g_once, load_rccl);
  return g_rccl.ok ? &g_rccl : nullptr;
}

struct Communicator
{
  Comm comm;     // null for the single-rank loopback
  int rank, world;
  // Device buffers of vk_comm_exchange_attach's collectives (handle all-gather + verdict all-reduce), allocated by
  // vk_comm_init: a rank that ran out of memory INSIDE the attach could not take part in the collective that tells
  // its peers so — here the attach has nothing left to allocate but the area itself, whose failure travels
  // in the verdict. Null for the loopback.
  void* staging;
};

// ---- the part of the HIP runtime the rig exchange needs (hip_runtime_api.h), bound like RCCL ----
enum { kIpcHandleBytes = 64, kMallocFinegrained = 0x1, kIpcLazyPeerAccess = 0x1, kCopyDefault = 4, kInt8 = 0 /* ncclInt8 */ };
struct IpcHandle { char reserved[kIpcHandleBytes]; };

struct Hip
{
  void* handle = nullptr;
  int (*ExtMallocWithFlags)(void**, size_t, unsigned) = nullptr;
  int (*Malloc)(void**, size_t) = nullptr;
  int (*Free)(void*) = nullptr;
  int (*Memset)(void*, int, size_t) = nullptr;
  int (*Memcpy)(void*, const void*, size_t, int) = nullptr;
  int (*DeviceSynchronize)() = nullptr;
  int (*IpcGetMemHandle)(IpcHandle*, void*) = nullptr;
  int (*IpcOpenMemHandle)(void**, IpcHandle, unsigned) = nullptr;
  int (*IpcCloseMemHandle)(void*) = nullptr;
  bool ok = false;
};

Hip g_hip;
std::once_flag g_hip_once;

void load_hip()
{
  const char* names[] = { getenv("VK_HIP_RUNTIME_LIBRARY"), "libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6", "/opt/rocm/lib/libamdhip64.so" };
  for (const char* name : names)
  {
    if (!name || !*name) continue;
    g_hip.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (g_hip.handle) break;
  }
  if (!g_hip.handle) return;
#define VK_BIND(field, symbol) g_hip.field = reinterpret_cast<decltype(g_hip.field)>(dlsym(g_hip.handle, symbol))
  VK_BIND(ExtMallocWithFlags, "hipExtMallocWithFlags");
  VK_BIND(Malloc, "hipMalloc");
  VK_BIND(Free, "hipFree");
  VK_BIND(Memset, "hipMemset");
  VK_BIND(Memcpy, "hipMemcpy");
  VK_BIND(DeviceSynchronize, "hipDeviceSynchronize");
  VK_BIND(IpcGetMemHandle, "hipIpcGetMemHandle");
  VK_BIND(IpcOpenMemHandle, "hipIpcOpenMemHandle");
  VK_BIND(IpcCloseMemHandle, "hipIpcCloseMemHandle");
#undef VK_BIND
  g_hip.ok = g_hip.ExtMallocWithFlags && g_hip.Malloc && g_hip.Free && g_hip.Memset && g_hip.Memcpy &&
      g_hip.DeviceSynchronize && g_hip.IpcGetMemHandle && g_hip.IpcOpenMemHandle && g_hip.IpcCloseMemHandle;
}

const Hip* hip()
{
  std::call_once(g_hip_once, load_hip);
  return g_hip.ok ? &g_hip : nullptr;
}

// include/vk.h vk_rig_exchange and vulcan_amd/csrc/vk_rig_protocol.h rig_area_words(), restated so that this
// file needs neither header (static_asserts in vk_icp.hip would not see this file; tests/test_abi.py compares)
enum { kRigMaxRanks = 8, kRigAreaBytes = 4 * 8 * 32 * 8 };
struct RigExchange
{
  unsigned long long* areas[kRigMaxRanks];
  int32_t rank, world;
  uint32_t sequence;
};
enum { kRigLastSequence = (1u << 22) - 2u };      // vk_rig_protocol.h VK_RIG_LAST_SEQUENCE

// this rank's area: fine-grained device memory, zeroed (a zero word carries no tag)
int make_area(const Hip* h, RigExchange* x, int rank, int world)
{
  memset(x, 0, sizeof(*x));
  x->rank = rank;
  x->world = world;
  x->sequence = 1;
  void* own = nullptr;
  int rc = h->ExtMallocWithFlags(&own, kRigAreaBytes, kMallocFinegrained);
  if (rc != 0) return rc;
  rc = h->Memset(own, 0, kRigAreaBytes);
  if (rc == 0) rc = h->DeviceSynchronize();
  if (rc != 0) { h->Free(own); return rc; }
  x->areas[rank] = static_cast<unsigned long long*>(own);
  return 0;
}

int map_peers(const Hip* h, RigExchange* x, const IpcHandle* all)
{
  for (int r = 0; r < x->world; ++r)
  {
    if (r == x->rank) continue;
    void* peer = nullptr;
    const int rc = h->IpcOpenMemHandle(&peer, all[r], kIpcLazyPeerAccess);
    if (rc != 0) return rc;
    x->areas[r] = static_cast<unsigned long long*>(peer);
  }
  return 0;
}

inline int from_nccl(Result r) { return r == 0 ? VK_OK_ : 1000 + r; }

// send handle | world received handles | {my verdict, everybody's}
constexpr size_t kStagingBytes = (size_t)kIpcHandleBytes * (size_t)(kRigMaxRanks + 1) + 2 * sizeof(float);

}  // namespace

extern "C" {

int vk_comm_unique_id(void* id)
{
  if (!id) return VK_ERR_ARGUMENT_;
  const Rccl* r = rccl();
  if (!r) return VK_COMM_ERR_NO_RCCL;
  UniqueId u;
  const Result rc = r->GetUniqueId(&u);
  if (rc != 0) return from_nccl(rc);
  memcpy(id, u.internal, VK_COMM_ID_BYTES);
  return VK_OK_;
}

int vk_comm_init(void** comm, const void* id, int rank, int world)
{
  if (!comm) return VK_ERR_ARGUMENT_;
  *comm = nullptr;
  if (world < 1 || rank < 0 || rank >= world) return VK_ERR_ARGUMENT_;
  Communicator* c = new Communicator{nullptr, rank, world, nullptr};
  if (world > 1 && !id) { delete c; return VK_ERR_ARGUMENT_; }
  if (id)   // world == 1 with an id: a real one-rank RCCL communicator; without: a loopback
  {
    const Rccl* r = rccl();
    if (!r) { delete c; return VK_COMM_ERR_NO_RCCL; }
    UniqueId u;
    memcpy(u.internal, id, VK_COMM_ID_BYTES);
    const Result rc = r->CommInitRank(&c->comm, world, u, rank);
    if (rc != 0) { delete c; return from_nccl(rc); }
    // the attach's device buffers, now (a rank-local failure: the caller agrees on the outcome of vk_comm_init over its
    // own channel before anybody enters a collective of this communicator, bench.py vk_comm_rig). Without a HIP runtime
    // the communicator still sums; vk_comm_exchange_attach then returns VK_COMM_ERR_NO_RCCL on every rank alike.
    if (const Hip* h = hip())
    {
      const int mrc = h->Malloc(&c->staging, kStagingBytes);
      if (mrc != 0) { (void)r->CommDestroy(c->comm); delete c; return mrc; }
    }
  }
  *comm = c;
  return VK_OK_;
}

int vk_comm_rank(const void* comm, int* rank, int* world)
{
  if (!comm) return VK_ERR_ARGUMENT_;
  const Communicator* c = static_cast<const Communicator*>(comm);
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  return VK_OK_;
}

int vk_comm_count(const void* comm, int* ranks)
{
  if (!comm || !ranks) return VK_ERR_ARGUMENT_;
  const Communicator* c = static_cast<const Communicator*>(comm);
  *ranks = 1;
  if (!c->comm) return VK_OK_;
  if (!g_rccl.CommCount) return VK_COMM_ERR_NO_RCCL;
  return from_nccl(g_rccl.CommCount(c->comm, ranks));
}

int vk_comm_allreduce_system(void* comm, float* system_dev, int count, void* stream)
{
  if (!comm || !system_dev || count <= 0) return VK_ERR_ARGUMENT_;
  Communicator* c = static_cast<Communicator*>(comm);
  if (!c->comm) return VK_OK_;                // loopback: the sum over one rank
  return from_nccl(g_rccl.AllReduce(system_dev, system_dev, (size_t)count, kFloat32, kSum, c->comm, stream));
}

int vk_comm_reduce_hook(float* system_dev, int count, void* comm, void* stream)
{
  return vk_comm_allreduce_system(comm, system_dev, count, stream);
}

int vk_comm_exchange_attach(void* comm, void* exchange)
{
  if (!comm || !exchange) return VK_ERR_ARGUMENT_;
  Communicator* c = static_cast<Communicator*>(comm);
  RigExchange* x = static_cast<RigExchange*>(exchange);
  if (c->world > kRigMaxRanks) return VK_ERR_ARGUMENT_;
  const Hip* h = hip();
  if (!h) return VK_COMM_ERR_NO_RCCL;
  int rc = make_area(h, x, c->rank, c->world);
  if (c->world == 1 || !c->comm) return rc;          // a loopback: nobody to agree with
  if (!g_rccl.AllGather) { vk_comm_exchange_detach(comm, exchange); return VK_COMM_ERR_NO_RCCL; }

  // From here on every rank makes the SAME collective calls whatever happened to it locally (a rank that failed
  // sends a zeroed handle and maps nothing), and the last one combines the outcomes: min over ranks of "I am
  // complete". A rank that returned early would leave its peers inside a collective for ever.
  // Every rank's handle to every rank: one all-gather of 64 bytes per rank, through device memory.
  IpcHandle mine;
  memset(&mine, 0, sizeof(mine));
  if (rc == 0) rc = h->IpcGetMemHandle(&mine, x->areas[c->rank]);
  // the collectives' buffers were allocated by vk_comm_init: nothing on this path can keep a rank out of them
  void* staging = c->staging;
  if (!staging) { vk_comm_exchange_detach(comm, exchange); return VK_ERR_ARGUMENT_; }   // (a communicator not made by vk_comm_init)
  char* send = static_cast<char*>(staging);
  char* recv = send + kIpcHandleBytes;
  float* verdict = reinterpret_cast<float*>(recv + (size_t)kIpcHandleBytes * (size_t)c->world);
  int step = h->Memcpy(send, &mine, kIpcHandleBytes, kCopyDefault);
  if (rc == 0) rc = step;
  Result nr = g_rccl.AllGather(send, recv, kIpcHandleBytes, kInt8, c->comm, nullptr);
  step = nr == 0 ? h->DeviceSynchronize() : 1000 + nr;
  if (rc == 0) rc = step;
  IpcHandle all[kRigMaxRanks];
  step = h->Memcpy(all, recv, (size_t)kIpcHandleBytes * (size_t)c->world, kCopyDefault);
  if (rc == 0) rc = step;
  if (rc == 0) rc = map_peers(h, x, all);

  const float complete = rc == 0 ? 1.0f : 0.0f;
  float everyone = 0.0f;
  step = h->Memcpy(verdict, &complete, sizeof(float), kCopyDefault);
  if (step == 0)
  {
    nr = g_rccl.AllReduce(verdict, verdict + 1, 1, kFloat32, kMin, c->comm, nullptr);
    step = nr == 0 ? h->DeviceSynchronize() : 1000 + nr;
  }
  if (step == 0) step = h->Memcpy(&everyone, verdict + 1, sizeof(float), kCopyDefault);
  if (rc == 0 && step != 0) rc = step;
  if (rc == 0 && everyone != 1.0f) rc = VK_COMM_ERR_PEER;
  if (rc != 0) vk_comm_exchange_detach(comm, exchange);
  return rc;
}

int vk_comm_exchange_create(void* exchange, int rank, int world, void* handle_out)
{
  if (!exchange || !handle_out || world < 1 || world > kRigMaxRanks || rank < 0 || rank >= world) return VK_ERR_ARGUMENT_;
  const Hip* h = hip();
  if (!h) return VK_COMM_ERR_NO_RCCL;
  RigExchange* x = static_cast<RigExchange*>(exchange);
  int rc = make_area(h, x, rank, world);
  IpcHandle mine;
  memset(&mine, 0, sizeof(mine));
  if (rc == 0) rc = h->IpcGetMemHandle(&mine, x->areas[rank]);
  if (rc != 0) { vk_comm_exchange_detach(nullptr, exchange); return rc; }
  memcpy(handle_out, &mine, kIpcHandleBytes);
  return VK_OK_;
}

int vk_comm_exchange_attach_handles(void* exchange, const void* handles)
{
  if (!exchange || !handles) return VK_ERR_ARGUMENT_;
  RigExchange* x = static_cast<RigExchange*>(exchange);
  if (x->world < 1 || x->world > kRigMaxRanks || x->rank < 0 || x->rank >= x->world || !x->areas[x->rank]) return VK_ERR_ARGUMENT_;
  const Hip* h = hip();
  if (!h) return VK_COMM_ERR_NO_RCCL;
  IpcHandle all[kRigMaxRanks];
  memcpy(all, handles, (size_t)kIpcHandleBytes * (size_t)x->world);
  const int rc = map_peers(h, x, all);
  if (rc != 0) vk_comm_exchange_detach(nullptr, exchange);
  return rc;
}

unsigned vk_comm_exchange_next_sequence(unsigned sequence)
{
  return sequence >= (unsigned)kRigLastSequence ? 1u : sequence + 1u;
}

int vk_comm_exchange_detach(void* comm, void* exchange)
{
  if (!exchange) return VK_ERR_ARGUMENT_;
  RigExchange* x = static_cast<RigExchange*>(exchange);
  const int rank = comm ? static_cast<Communicator*>(comm)->rank : x->rank;
  const Hip* h = hip();
  if (!h) return VK_COMM_ERR_NO_RCCL;
  (void)h->DeviceSynchronize();          // no loop kernel may still be writing into a peer
  for (int r = 0; r < kRigMaxRanks; ++r)
  {
    if (!x->areas[r]) continue;
    if (r == rank) (void)h->Free(x->areas[r]); else (void)h->IpcCloseMemHandle(x->areas[r]);
    x->areas[r] = nullptr;
  }
  return VK_OK_;
}

int vk_comm_destroy(void* comm)
{
  if (!comm) return VK_OK_;
  Communicator* c = static_cast<Communicator*>(comm);
  int rc = VK_OK_;
  if (c->comm) rc = from_nccl(g_rccl.CommDestroy(c->comm));
  if (c->staging) { if (const Hip* h = hip()) (void)h->Free(c->staging); }
  delete c;
  return rc;
}

const char* vk_comm_error_string(int code)
{
  if (code == VK_OK_) return "success";
  if (code == VK_ERR_ARGUMENT_) return "invalid argument [vk error -1]";
  if (code == VK_COMM_ERR_NO_RCCL) return "librccl could not be loaded (set VK_RCCL_LIBRARY) [vk error -4]";
  if (code == VK_COMM_ERR_PEER) return "another rank failed in the same collective call [vk error -5]";
  if (code >= 1000 && g_rccl.GetErrorString) return g_rccl.GetErrorString(code - 1000);
  return "unknown error";
}

}  // extern "C"
