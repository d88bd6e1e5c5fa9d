// vk_comm.cpp — libvk_comm.so: the rig's all-reduce over RCCL (include/vk_comm.h).
// Plain C++ (no device code). RCCL is bound with dlopen/dlsym so that the library
// has no link-time dependency on it.
#include "../../include/vk_comm.h"

#include <dlfcn.h>
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
enum { kFloat32 = 7, kSum = 0 };  // ncclFloat32, ncclSum

struct Rccl
{
  void* handle = nullptr;
  Result (*GetUniqueId)(UniqueId*) = nullptr;
  Result (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  Result (*AllReduce)(const void*, void*, size_t, int, int, Comm, void*) = nullptr;
  Result (*CommDestroy)(Comm) = nullptr;
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
  g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(g_rccl.handle, "ncclCommDestroy"));
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
};

inline int from_nccl(Result r) { return r == 0 ? VK_OK_ : 1000 + r; }

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
  Communicator* c = new Communicator{nullptr, rank, world};
  if (world > 1 && !id) { delete c; return VK_ERR_ARGUMENT_; }
  if (id)   // world == 1 with an id: a real one-rank RCCL communicator; without: a loopback
  {
    const Rccl* r = rccl();
    if (!r) { delete c; return VK_COMM_ERR_NO_RCCL; }
    UniqueId u;
    memcpy(u.internal, id, VK_COMM_ID_BYTES);
    const Result rc = r->CommInitRank(&c->comm, world, u, rank);
    if (rc != 0) { delete c; return from_nccl(rc); }
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

int vk_comm_destroy(void* comm)
{
  if (!comm) return VK_OK_;
  Communicator* c = static_cast<Communicator*>(comm);
  int rc = VK_OK_;
  if (c->comm) rc = from_nccl(g_rccl.CommDestroy(c->comm));
  delete c;
  return rc;
}

const char* vk_comm_error_string(int code)
{
  if (code == VK_OK_) return "success";
  if (code == VK_ERR_ARGUMENT_) return "invalid argument [vk error -1]";
  if (code == VK_COMM_ERR_NO_RCCL) return "librccl could not be loaded (set VK_RCCL_LIBRARY) [vk error -4]";
  if (code >= 1000 && g_rccl.GetErrorString) return g_rccl.GetErrorString(code - 1000);
  return "unknown error";
}

}  // extern "C"
