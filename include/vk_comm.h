/*
 * vk_comm.h — C ABI of the one exchange step the hot path has (libvk_comm.so).
 *
 * The fusion + raycast path shards by camera: every GPU owns a replica volume
 * and no voxel ever crosses xGMI. A rigid multi-camera rig
 * (BASELINE configs[4]) adds exactly one collective: after each rank has built the
 * ICP normal system of its own view (ref: src/depth_tracker.cu:144-268, the 36+6
 * floats Tracker::ComputeUpdate copies to the host at src/tracker.cpp:136-137),
 * the packed system is summed over ranks, and every rank runs the same solve
 * (src/tracker.cpp:153-162). The reference has no multi-GPU code; this is the
 * slot it would occupy. Message: 48 floats, latency bound — one ncclAllReduce on
 * the compute stream, straight from device memory.
 *
 * RCCL is resolved at run time (dlopen of $VK_RCCL_LIBRARY, else librccl.so.1):
 * the library loads, and its argument checks work, on a machine without RCCL
 * or without a GPU, and a process that already carries an RCCL (PyTorch) shares it.
 *
 * Conventions as in vk.h: plain C, 0 on success, negative VK_ERR_* on bad
 * arguments, positive = 1000 + ncclResult_t on an RCCL failure (vk_comm_error_string).
 */
#ifndef VK_COMM_H_
#define VK_COMM_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef VK_API
#define VK_API __attribute__((visibility("default")))
#endif

#define VK_COMM_ID_BYTES 128        /* sizeof(ncclUniqueId) */
#define VK_COMM_ERR_NO_RCCL (-4)    /* librccl could not be loaded */
#define VK_COMM_SYSTEM_FLOATS 48    /* hessian[36] | gradient[6] | pad[6] */

/* Rank 0 creates the id and hands it to the other ranks by whatever the host
 * application uses (a file, MPI, a socket, torch.distributed). ncclGetUniqueId. */
VK_API int vk_comm_unique_id(void* id /* VK_COMM_ID_BYTES */);

/* Collective over all `world` ranks, each on its own current HIP device.
 * world == 1 with id == NULL needs no RCCL: the communicator is a loopback (the
 * sum over one rank). ncclCommInitRank. */
VK_API int vk_comm_init(void** comm, const void* id, int rank, int world);

VK_API int vk_comm_rank(const void* comm, int* rank, int* world);

/* How many ranks RCCL itself says the communicator has (ncclCommCount; 1 for the loopback): a
 * launcher's WORLD_SIZE is checked against this, not only against its own environment. */
VK_API int vk_comm_count(const void* comm, int* ranks);

/* In-place sum of `count` floats (the packed ICP system) over all ranks on
 * `stream`; asynchronous. ref: the slot is src/tracker.cpp:136-153. */
VK_API int vk_comm_allreduce_system(void* comm, float* system_dev, int count, void* stream);

/* The same as a vk_icp_reduce_fn (vk.h): pass it as `reduce` and the
 * communicator as `reduce_user` to vk_icp_track / vk_color_tracker_track /
 * vk_light_tracker_track, or wrap it in Tracker::SetReduceHook. */
VK_API int vk_comm_reduce_hook(float* system_dev, int count, void* comm, void* stream);

/* The rig's exchange INSIDE the one-launch Gauss-Newton loop (vk.h vk_rig_exchange, vk_icp_track_rig;
 * protocol: vulcan_amd/csrc/vk_rig_protocol.h). Collective over the communicator's ranks: every rank
 * allocates its area (fine-grained device memory, zeroed), the ranks exchange the areas' IPC handles
 * (hipIpcGetMemHandle, one ncclAllGather) and map each other's (hipIpcOpenMemHandle with lazy peer
 * access, i.e. direct stores over xGMI). Fills *exchange (a vk_rig_exchange: areas[0 .. world), rank,
 * world, sequence = 1); the caller adds 1 to `sequence` after every Track, on every rank. With a
 * loopback communicator (world == 1) only the own area is made. The all-reduce hook above remains
 * the fallback (a rank that cannot map a peer returns the HIP error, and the caller keeps the hook).
 * HIP is resolved at run time like RCCL ($VK_HIP_RUNTIME_LIBRARY, else libamdhip64.so).
 * NOT exercised with more than one rank on hardware: this pool hands out single-GPU boxes. */
VK_API int vk_comm_exchange_attach(void* comm, void* exchange /* vk_rig_exchange* */);

/* unmaps the peers' areas and frees the own one (call on every rank, before vk_comm_destroy) */
VK_API int vk_comm_exchange_detach(void* comm, void* exchange /* vk_rig_exchange* */);

VK_API int vk_comm_destroy(void* comm);

VK_API const char* vk_comm_error_string(int code);

#ifdef __cplusplus
}
#endif
#endif  /* VK_COMM_H_ */
