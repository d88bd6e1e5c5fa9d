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
 *
 * Two PROCESSES on one GPU (a shared node; `bench.py --gpus 2` on a one-GPU box; round 6: stated here, it used to live
 * in a round-2 note). One rank per GPU is the supported layout. What a shared device gets:
 *   - RCCL itself refuses two ranks on one device (ncclCommInitRank fails): vk_comm_init returns 1000 + ncclResult_t on
 *     every rank; the exchange can still be set up with vk_comm_exchange_create / _attach_handles over the host's own
 *     channel (tests/test_gpu_rig_two_ranks.py does exactly that).
 *   - The one-launch Gauss-Newton loops (vk_icp_track & co., vk_icp_track_rig) need ALL their workgroups on the device
 *     at once, and one such launch fills it. The library chains the loop launches of ONE process across its streams
 *     (vk_runtime.hip vk_loop_launch_begin / _end); it cannot chain two processes. When two processes' loop launches
 *     meet, each may hold a part of the device and wait for the rest: the waits inside the launch are bounded (two
 *     seconds), both launches then end with state[1] = VK_TRACK_ABORTED, no pose is published (vk_track_wait returns
 *     VK_ERR_UNSUPPORTED), and the hosts' documented way out runs: the same Track again from the start pose as the
 *     launch-per-stage loop (`reduce` = vk_reduce_nothing, or the all-reduce hook on a rig), which waits for nobody.
 *     The class layer and vulcan_amd/api.py do this by themselves (Tracker::Track, _with_fallback); on a rig
 *     Communicator.track_rig makes the ranks agree on the abort first. The result is the same pose (the staged loop is
 *     held to the same oracle, tests/test_gpu_loop_abort.py); the cost is up to two seconds per collision. Nothing is
 *     wrong silently; nothing hangs.
 *   - The raycast's riding normals (vk_trace_ahead_requests) wait for workgroups of their OWN launch only; a second
 *     process slows them down but cannot starve them; should the bounded wait still expire, VK_ERR_TIMEOUT and
 *     vk_trace_normals_settle (vk.h) apply.
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
#define VK_COMM_ERR_PEER (-5)       /* a collective step failed on ANOTHER rank: every rank returns an error, the same way */
#define VK_COMM_IPC_HANDLE_BYTES 64 /* sizeof(hipIpcMemHandle_t) */

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
 * world, sequence = 1); after every Track it entered — aborted ones included — every rank moves `sequence`
 * on with vk_comm_exchange_next_sequence. With a loopback communicator (world == 1) only the own area is made.
 * The all-reduce hook above remains the fallback for a rig whose attach failed.
 * HIP is resolved at run time like RCCL ($VK_HIP_RUNTIME_LIBRARY, else libamdhip64.so).
 * The RESULT is collective too: the ranks' outcomes are combined (one ncclAllReduce, min) before anyone returns, so
 * either every rank holds a complete exchange and VK_OK, or every rank has detached and returns an error — its own,
 * or VK_COMM_ERR_PEER when it was another rank's mapping that failed. Nobody is left on the all-reduce hook alone.
 * NOT exercised with more than one GPU on hardware: this pool hands out single-GPU boxes (two processes on one GPU
 * run the path below, tests/test_gpu_rig_two_ranks.py). */
VK_API int vk_comm_exchange_attach(void* comm, void* exchange /* vk_rig_exchange* */);

/* The same exchange for a host that moves the handles itself (MPI, a socket, torch.distributed over gloo — and
 * then also for ranks that share a device, which RCCL refuses): no communicator involved.
 *   1. vk_comm_exchange_create: allocates and zeroes this rank's area, fills *exchange (rank, world, sequence = 1,
 *      areas[rank]) and writes the area's IPC handle (VK_COMM_IPC_HANDLE_BYTES) to handle_out;
 *   2. the host gathers all ranks' handles, in rank order;
 *   3. vk_comm_exchange_attach_handles: maps the peers' areas (handles[rank] is skipped). On failure the exchange is
 *      detached. The ranks must agree on the outcome through their own channel before any of them starts a Track.
 * vk_comm_exchange_detach(NULL, exchange) undoes both. */
VK_API int vk_comm_exchange_create(void* exchange /* vk_rig_exchange* */, int rank, int world, void* handle_out);
VK_API int vk_comm_exchange_attach_handles(void* exchange /* vk_rig_exchange* */, const void* handles /* world x 64 bytes */);

/* The Track number that follows `sequence` (vk_rig_exchange.sequence): 1, 2, ... 2^22 - 2, 1, ... — every rank advances
 * after every Track it entered, aborted ones included (vulcan_amd/csrc/vk_rig_protocol.h rig_next_sequence). */
VK_API unsigned vk_comm_exchange_next_sequence(unsigned sequence);

/* unmaps the peers' areas and frees the own one (call on every rank, before vk_comm_destroy); comm may be NULL */
VK_API int vk_comm_exchange_detach(void* comm, void* exchange /* vk_rig_exchange* */);

VK_API int vk_comm_destroy(void* comm);

VK_API const char* vk_comm_error_string(int code);

#ifdef __cplusplus
}
#endif
#endif  /* VK_COMM_H_ */
