/*
 * vk_probe.h — measurement aids (libvk_probe.so). NOT part of the product ABI
 * (include/vk.h) and not linked into libvk_hip.so: bench.py and tools/ load it to
 * put a kernel's GB/s next to what the same GPU sustains on a plain copy and on
 * the integrate kernel's own access pattern with the arithmetic removed
 * (SURVEY.md §8d "Peak to divide by: measured"). No reference counterpart.
 */
#ifndef VK_PROBE_H_
#define VK_PROBE_H_

#include "../../include/vk.h"

#ifdef __cplusplus
extern "C" {
#endif

/* float4 device copy, `bytes` a multiple of 16. shape:
 *   0  one float4 per lane, one-shot grid (bytes / 4096 workgroups of 256)
 *   1  four float4 per lane (4 KiB per wave-instruction group), one-shot grid
 *   2  persistent grid (8 workgroups per CU), 4 loads in flight per lane
 *   3  as 1 with non-temporal loads and stores
 */
VK_API int vk_probe_stream_copy(void* dst, const void* src, size_t bytes, int shape, void* stream);
/* float4 read-only sweep (sum folded into one store per workgroup): the read half alone */
VK_API int vk_probe_stream_read(const void* src, size_t bytes, float* sink, void* stream);
/* every visible block read and written back unchanged (10 240 B each way) */
VK_API int vk_probe_block_rmw(const vk_volume* v, int mode, void* stream);   /* 0 plain, 1 nt stores, 2 nt loads+stores */

/* Nhit of SURVEY.md section 8d: the raycast kernel of the product (vk_raycast.hpp, the very same
 * code, instantiated with its counting hook) run over the whole image; `touched` (device,
 * one byte per pool slot, zeroed by the caller) receives 1 for every voxel block a ray read.
 * Arguments as vk_trace_compute_points (vk.h); depths / colors are written as usual.
 * `touched` NULL: the plain kernel (32-bit pool offsets) is run instead. `wave_clocks` (optional,
 * device, 2 x 4 x workgroups u64): start and end of every wave on the 100 MHz wall clock. */
VK_API int vk_probe_trace_touched(const vk_hash_entry* entries, const vk_voxel* voxels, const float* bounds,
    int block_count, float block_length, float voxel_length, float trunc_length, const vk_transform* Twc,
    const vk_projection* projection, float* depths, float* colors, int image_width, int image_height,
    int bounds_width, int bounds_height, uint8_t* touched, unsigned long long* wave_clocks, void* stream);

/* as vk_probe_trace_touched; `march_steps` (optional, device int[h*w], with `touched` only): trips through the
 * march loop per pixel */
VK_API int vk_probe_trace_steps(const vk_hash_entry* entries, const vk_voxel* voxels, const float* bounds,
    int block_count, float block_length, float voxel_length, float trunc_length, const vk_transform* Twc,
    const vk_projection* projection, float* depths, float* colors, int image_width, int image_height,
    int bounds_width, int bounds_height, uint8_t* touched, unsigned long long* wave_clocks, int* march_steps, void* stream);

/* as vk_probe_trace_steps; `trip_log` (optional, device u64[waves x trip_log_passes], zeroed by the caller, with `touched`
 * only): one word per wave and pass through the march loop, see PointParams::trip_log (vk_raycast.hpp) */
VK_API int vk_probe_trace_log(const vk_hash_entry* entries, const vk_voxel* voxels, const float* bounds,
    int block_count, float block_length, float voxel_length, float trunc_length, const vk_transform* Twc,
    const vk_projection* projection, float* depths, float* colors, int image_width, int image_height,
    int bounds_width, int bounds_height, uint8_t* touched, unsigned long long* wave_clocks, int* march_steps,
    unsigned long long* trip_log, int trip_log_passes, void* stream);

/* launch floor: `replays` x `launches` kernels of `workgroups` x 256 lanes that read counters[0] (device, 0)
 * and leave, in stream order */
VK_API int vk_probe_launch_floor(const int32_t* counters, float* sink, int workgroups, int launches, int replays,
    void* stream);

/* the same chain captured once into a hipGraph on a stream of its own (the legacy default stream cannot be
 * captured) and replayed: *us_per_launch = time per kernel over `replays` graph launches. Returns the HIP
 * error code of the first call that fails (positive), 0 on success. */
/* vk_probe.hip's rcp_rn_mid / sqrt_rn_mid (an experiment of round 5, not in the product) against the compiler's 1.0f / x and sqrtf(x) for EVERY float in [2^-60, 2^60] (both
 * signs for the reciprocal): out[0..5] (device) = tested, differing (reciprocal); tested, differing (square root); first
 * differing input of each as bit pattern + 1 (0: none). */
VK_API int vk_probe_rounding(unsigned long long* out_dev6, void* stream);

/* FETCH_SIZE calibration for the raycast's access shape (round 6): one wave per 10 240-byte voxel block of
 * `voxels` (block b of `blocks`, every block touched exactly once, so no line is read twice), 4- and 12-byte loads at the
 * 20-byte voxel stride the raycast issues. The set of lines a launch touches is known on the host (tools/fetch_calibration.py
 * enumerates the same addresses):
 *   mode 0  dense:   every lane reads the 4-byte distance of 8 voxels — all 512 voxels, every line of the block
 *   mode 1  sparse:  lanes 0..63 read the distance of voxel 8 * lane — one 4-byte word every 160 bytes
 *   mode 2  corners: every lane reads distance (4 B) and rgb (12 B) of the 2 x 2 x 2 voxels at a position derived from
 *                    (block, lane) by a fixed hash — the raycast's trilinear sample
 *   mode 3  float4:  every lane reads ten aligned 16-byte words, the whole block as ten wave-wide 1 KiB loads (the guide's
 *                    calibrated shape, x2: the control)
 * The loaded values are folded into sink[wave] so that no load is dead. */
VK_API int vk_probe_gather(const void* voxels, int blocks, int mode, float* sink, void* stream);

VK_API int vk_probe_launch_floor_graph(const int32_t* counters, float* sink, int workgroups, int launches, int replays,
    float* us_per_launch);

#ifdef __cplusplus
}
#endif
#endif
