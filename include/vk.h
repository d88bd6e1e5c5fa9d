/*
 * vk.h — C ABI of the MI355X-native fusion + raycast hot path (libvk_hip.so).
 *
 * This is the drop-in boundary. The reference (mkaspr/Vulcan) has no FFI: its
 * boundary is the C++ class surface Volume / Integrator(+3) / Tracer / Frame /
 * DepthTracker plus the free-function layer tracer.cuh:14-32 / frame.cuh.  Each
 * entry point below replaces the body of one of those methods (cited as
 * `ref: file:line`, paths relative to the reference tree); the C++ classes in
 * vulcan_amd/host/ keep the reference's names and forward here.
 *
 * Conventions
 *  - plain C: PODs, raw DEVICE pointers, ints, floats. No C++/torch types.
 *  - every function returns 0 on success, a hipError_t value (>0) on a HIP
 *    failure, or a negative VK_ERR_* code on bad arguments. Nothing throws.
 *    vk_error_string() maps a code to text.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream). All
 *    work is stream-ordered and asynchronous unless the function name ends in
 *    `_sync` or the comment says "blocking".
 *  - matrices are 4x4 column-major float[16], exactly the reference's
 *    Matrix4f (matrix.h:320-333); vk_transform carries matrix + cached inverse
 *    like Transform (transform.h:168-170). The inverse is never recomputed.
 *  - counts that the reference reads back to the host between kernels
 *    (visible-block count volume.cu:494, patch count tracer.cpp:67) stay in
 *    device memory (vk_volume.counters); consumers read them on the device.
 */
#ifndef VK_H_
#define VK_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VK_API __attribute__((visibility("default")))

/* ---------------------------------------------------------------- PODs -- */

/* ref: include/vulcan/voxel.h:43-49 — 20 bytes, align 4 */
typedef struct vk_voxel {
  float   distance;
  float   color[3];
  int16_t distance_weight;
  int16_t color_weight;
} vk_voxel;

/* ref: include/vulcan/block.h:81-85 — 8 bytes. `pad` is uninitialised in the
 * reference; here it is always written as 0 in hash entries. */
typedef struct vk_block {
  int16_t origin[3];
  int16_t pad;
} vk_block;

/* ref: include/vulcan/hash.h:51-55 — 16 bytes */
typedef struct vk_hash_entry {
  vk_block block;
  int32_t  data;  /* voxel pool slot, -1 = unallocated */
  int32_t  next;  /* entry index of next in chain, -1 = end */
} vk_hash_entry;

/* ref: include/vulcan/tracer.h:13-22 — 16 bytes */
typedef struct vk_patch {
  int16_t origin[2];
  int16_t size[2];
  float   bounds[2];
} vk_patch;

/* ref: include/vulcan/projection.h:111-113 — focal length then centre */
typedef struct vk_projection {
  float fx, fy, cx, cy;
} vk_projection;

/* ref: include/vulcan/transform.h:168-170 */
typedef struct vk_transform {
  float m[16];    /* matrix_, column-major */
  float inv[16];  /* inv_matrix_ */
} vk_transform;

/* ref: include/vulcan/light.h:62-66 */
typedef struct vk_light {
  float intensity;
  float position[3];
} vk_light;

/* ref: include/vulcan/types.h:6-18 */
enum { VK_VISIBILITY_UNKNOWN = 0, VK_VISIBILITY_FALSE = 1, VK_VISIBILITY_TRUE = 2 };
enum { VK_ALLOC_NONE = 0, VK_ALLOC_MAIN = 1, VK_ALLOC_EXCESS = 2 };

enum {
  VK_BLOCK_RESOLUTION = 8,    /* block.h:13 */
  VK_BLOCK_VOXELS     = 512,  /* block.h:15 */
  VK_PATCH_MAX_SIZE   = 16    /* tracer.h:15 */
};

/* Per-volume device counters. The reference keeps these as file-scope
 * __device__ symbols shared by every Volume (volume.cu:17-21); here each
 * volume owns an int[VK_CTR_COUNT] in device memory, 8-byte aligned. */
enum {
  VK_CTR_VISIBLE    = 0,  /* buffer_size: number of entries in visible_blocks */
  VK_CTR_VOXEL_PTR  = 1,  /* voxel_pointer: top of the free-slot stack */
  VK_CTR_EXCESS_PTR = 2,  /* excess_pointer: next free excess entry */
  VK_CTR_PATCHES    = 3,  /* tracer patch count (tracer.h buffer_size_) */
  VK_CTR_REQUESTS   = 4,  /* requests seen by the last handle pass */
  VK_CTR_DROPPED    = 5,  /* requests dropped so far: pool or excess list exhausted */
  VK_CTR_PENDING_ALL    = 6,  /* internal: pointer updates of a handle pass, not yet folded in */
  VK_CTR_PENDING_EXCESS = 7,
  VK_CTR_ROUNDS     = 8,  /* SetView rounds run so far by vk_volume_set_view* (cumulative) */
  VK_CTR_UNSETTLED  = 9,  /* the last vk_volume_set_view* left a request unanswered: lost to a bucket
                             contest in its last round, or dropped (pool / excess list exhausted), or
                             more losers than the retry list holds — another SetView call has work */
  VK_CTR_CONTENDED  = 10, /* internal: the request pass lost a request to a bucket contest */
  VK_CTR_TICKET     = 11, /* internal: handle workgroups that have finished the first round */
  VK_CTR_RETRY_COUNT = 12, /* internal [2]: keys filed in the two retry lists */
  VK_CTR_RETRY_OVERFLOW = 14, /* internal: a retry list (or the ordering buffer of a later round) was too small */
  VK_CTR_DROPPED_NOW = 15, /* internal: the current call dropped a request */
  VK_CTR_ORIGIN_SEEN = 16, /* internal: a ray met block (0,0,0) while the main entry of its bucket was
                              unallocated — such an entry compares equal to that block (volume.cu:186-191),
                              so the block counts as present until the entry is taken by another block */
  VK_CTR_POSTED     = 17, /* internal: buckets that received their first request in the current request pass
                             (may exceed VK_POSTED_SLOTS: then the handle pass scans the request flags instead) */
  VK_CTR_ARRIVALS   = 18, /* internal [2], one 64-bit word: the arrivals of the workgroups of the handle + visibility
                             launch of vk_volume_set_view* (zero between calls) */
  VK_CTR_BANDED     = 20, /* internal: the banded visible lists (below) hold exactly VK_CTR_VISIBLE entries — set to that
                             count by the handle + visibility launch of vk_volume_set_view* when it has listed every
                             visible entry in them, to -1 by everything else that writes the visible list */
  VK_CTR_PUBLIC     = 24, /* what vk_volume_read_counters_sync copies */
  /* behind the counters, for vk_volume_set_view_rounds: the blocks whose request lost its bucket in
   * the round before — two open-addressing sets of VK_RETRY_SLOTS 64-bit request keys (current
   * round / next round) and, for each, the list of the slots in use (VK_RETRY_KEYS ints) */
  VK_RETRY_SLOTS    = 65536,
  VK_RETRY_KEYS     = 8192,
  /* then the buckets with a request of the current request pass, in order of arrival (bit 31: an
   * EXCESS request), and for each the last entry of its chain: what the handle pass of
   * vk_volume_set_view* works from when there are few */
  VK_POSTED_SLOTS   = 2048,
  /* then the visible entries once more, binned by the image row band of a depth pixel whose ray touched the block
   * (VK_BANDS bands of height / VK_BANDS rows): VK_BANDS counts, then VK_BANDS lists of VK_BAND_SLOTS entry indices.
   * The integrate kernels deal the bands to the XCDs, so that each XCD's L2 holds the image rows its blocks project
   * to (round 4). A band with more entries than slots makes the integrate kernels use the plain list. */
  VK_BANDS          = 8,
  VK_BAND_SLOTS     = 16384,
  VK_CTR_COUNT      = 24 + 2 * 2 * 65536 + 2 * 8192 + 2 * 2048 + 8 + 8 * 16384
};

enum {
  VK_OK              =  0,
  VK_ERR_ARGUMENT    = -1,  /* null pointer / non-positive size */
  VK_ERR_UNSUPPORTED = -2,
  VK_ERR_NO_DEVICE   = -3,
  VK_ERR_TIMEOUT     = -6   /* a bounded wait INSIDE a launch expired (vk_trace_normals_settle); what it says was not
                               written has been recomputed, or the call must be repeated. (-4, -5: vk_comm.h) */
};

/* Device-memory view of a Volume (ref: include/vulcan/volume.h:89-116).
 * The caller owns every buffer; the library never allocates or frees them. */
typedef struct vk_volume {
  vk_voxel*      voxels;             /* [max_block_count * 512]      voxels_            */
  vk_hash_entry* hash_entries;       /* [max_block_count]            hash_entries_      */
  int32_t*       free_voxel_blocks;  /* [max_block_count]            free_voxel_blocks_ */
  uint8_t*       allocation_types;   /* [main_block_count], 16-byte aligned             allocation_types_  */
  vk_block*      allocation_blocks;  /* [main_block_count], 8-byte aligned  allocation_blocks_ */
  uint8_t*       block_visibility;   /* [max_block_count], 4-byte aligned, the allocation padded to a multiple of
                                        4 bytes (the last bytes are accessed as part of their word)  block_visibility_  */
  int32_t*       visible_blocks;     /* [max_block_count]            visible_blocks_    */
  int32_t*       counters;           /* [VK_CTR_COUNT]               (volume.cu:17-21)  */
  int32_t        main_block_count;
  int32_t        excess_block_count;
  float          voxel_length;       /* volume.cu:376 default 0.008 */
  float          truncation_length;  /* volume.cu:375 default 0.04  */
  float          min_depth;          /* volume.cu:374 default 0.1   */
  float          max_depth;          /*               default 5.0   */
} vk_volume;

/* Depth / colour / normal images of a Frame (ref: include/vulcan/frame.h:11-32)
 * row-major, no pitch; colour and normals are packed 12-byte float3. */
typedef struct vk_frame {
  const float*  depth;        /* [height*width]   */
  const float*  color;        /* [color_height*color_width*3] or NULL */
  const float*  normals;      /* [height*width*3] or NULL */
  int32_t       width;        /* depth_image (and normal_image) size */
  int32_t       height;
  /* color_image size (frame.h:21-25 holds three independent images). 0 = same
   * as the depth image. ColorIntegrator::IntegrateColor bounds-tests and
   * strides with THIS size (color_integrator.cu:183-184); LightIntegrator
   * indexes colour with the depth image's size (light_integrator.cu:333-334,
   * "TODO: handle different depth and color images sizes" :179), which is an
   * out-of-bounds read when they differ: the light entry points return
   * VK_ERR_ARGUMENT for such a frame instead. */
  int32_t       color_width;
  int32_t       color_height;
  vk_projection depth_projection;
  vk_projection color_projection;
  vk_transform  depth_to_world;   /* Twd */
  vk_transform  depth_to_color;   /* Tcd */
  /* Identity of the images' CONTENT, maintained by the caller: a value that changes whenever
   * a pixel of the depth, colour or normal image changes (the class layer stamps its images on
   * every write it can see, image.h). 0 = unknown. Only vk_volume_set_view_prepare /
   * vk_light_prepared look at it: work done ahead for a frame is reused for the same
   * content only, never for "the same pointers". No reference counterpart. */
  uint64_t      content_id;
} vk_frame;

/* -------------------------------------------------------------- test aids -- */

/* Process-wide switches with which the test suites force paths that real input reaches rarely, and two
 * measurement options. Nothing on a call path reads the environment: a caller sets these once (any thread,
 * before the calls they concern); the library reads them with relaxed atomic loads. No reference counterpart. */
typedef struct vk_test_hooks {
  int32_t posted_capacity;    /* buckets the posted list of vk_volume_set_view* holds; < 0: VK_POSTED_SLOTS. A small
                                 list sends the handle pass to the request flags */
  int32_t retry_capacity;     /* distinct keys per retry list; <= 0: VK_RETRY_KEYS. A small list overflows on purpose */
  int32_t set_view_unfused;   /* 1: vk_volume_set_view* as three launches (requests, handle + later rounds,
                                 visibility) instead of two: the second implementation the tests compare with */
  int32_t force_loop_abort;   /* 1: every one-launch Gauss-Newton loop ends at once with VK_TRACK_ABORTED, as after the
                                 exchange timeout, so that the launch-per-stage fallback can be tested */
  int32_t loop_grid_cap;      /* > 0: at most this many workgroups per loop kernel (a device with fewer CUs) */
  int32_t loop_cooperative;   /* 1: loop kernels go through hipLaunchCooperativeKernel (the runtime then refuses a
                                 grid that cannot be resident; +4..6.5 us per Track) */
  int32_t force_normals_expiry; /* 1: the normals workgroups of the NEXT vk_trace_ahead_requests launch find their wait
                                 expired at once (a target no counter reaches, no polls) — then back to 0 by itself: the
                                 error path of vk_trace_normals_settle under test */
} vk_test_hooks;

VK_API int vk_test_hooks_set(const vk_test_hooks* hooks);   /* NULL: everything back to its default */
VK_API int vk_test_hooks_get(vk_test_hooks* out);

/* The library's count of one-launch Gauss-Newton loops (vk_icp_track & co.). Every such launch tags the words its
 * workgroups exchange with 22 bits of this count (1 + count mod (2^22 - 1)), so the tags repeat every 2^22 - 1 launches —
 * about ten minutes of tracking. The library keeps the repeat from ever being seen: a tracker's `workspace` is cleared
 * in front of a launch whenever it is new to the library, was last cleared 2^21 or more launches ago, or was written by
 * a launch-per-stage loop in between (vulcan_amd/csrc/vk_runtime.hip, vk_loop_epoch_begin). What the caller owes: nothing
 * but the library writes a workspace between two calls that use it, and a workspace made of memory that has been back at an
 * allocator since the library last saw that address is zeroed by the caller first (vk_memset) — the library knows areas by
 * their address and cannot see a free (the host layers zero a workspace when they allocate it). This entry point lets a test put the count right
 * in front of a repeat instead of tracking for ten minutes: `set_to` (may be NULL) replaces the count, `count_now` and
 * `clears` (may be NULL) receive the count and the number of clears enqueued so far. No reference counterpart (upstream's
 * loop, src/tracker.cpp:53-63, has no launch tags). */
VK_API int vk_test_hooks_loop_count(const uint64_t* set_to, uint64_t* count_now, uint64_t* clears);

/* ------------------------------------------------------ library / device -- */

VK_API const char* vk_error_string(int code);
VK_API int vk_version(void);                       /* 100*major + minor */

/* The binary interface has changed between rounds (struct fields, the size of the counters block — VK_CTR_COUNT ints,
 * of which vk_volume_read_counters_sync copies VK_CTR_PUBLIC = 24 —, the meaning of vk_frame.content_id), and a caller
 * built against an older vk.h would hand the library buffers that are too small. VK_ABI_VERSION counts those changes;
 * vk_abi_check compares what the CALLER was compiled against with what the LIBRARY was: call it once after loading
 *   vk_abi_check(VK_ABI_VERSION, sizeof(vk_volume), sizeof(vk_frame), VK_CTR_COUNT)
 * and refuse to go on unless it returns VK_OK (VK_ERR_UNSUPPORTED otherwise). The class layer (vulcan_amd/host) and the
 * Python binding do. New struct fields are appended; a change of VK_CTR_COUNT bumps the version. No reference counterpart. */
#define VK_ABI_VERSION 7
VK_API int vk_abi_version(void);
VK_API int vk_abi_check(int header_abi_version, size_t sizeof_vk_volume, size_t sizeof_vk_frame, int ctr_count);
VK_API int vk_device_count(int* count);
VK_API int vk_set_device(int device);
VK_API int vk_device_name(char* out, size_t bytes);

VK_API int vk_stream_create(void** stream);
VK_API int vk_stream_destroy(void* stream);
VK_API int vk_stream_synchronize(void* stream);   /* blocking */

/* Device memory for the host classes (ref: buffer.h:64-103, image.h:85-97). */
VK_API int vk_malloc(void** ptr, size_t bytes);
VK_API int vk_free(void* ptr);
VK_API int vk_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream);  /* blocking */
VK_API int vk_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream);  /* blocking */
VK_API int vk_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream);
VK_API int vk_memset(void* dst, int value, size_t bytes, void* stream);

/* Pinned, device-visible, coherent host memory (hipHostMalloc): the same pointer
 * is valid on the host and in kernels. Used for vk_track_poll.host_state. */
VK_API int vk_malloc_host(void** ptr, size_t bytes);
VK_API int vk_free_host(void* ptr);

/* Timing helpers (hipEvent) so hosts without HIP headers can time kernels. The
 * events carry no system-scope fence (hipEventDisableSystemFence): they order and
 * time work on a stream, they do not publish device writes to the host — use
 * vk_stream_synchronize or a blocking copy for that. */
VK_API int vk_event_create(void** event);
VK_API int vk_event_destroy(void* event);
VK_API int vk_event_record(void* event, void* stream);
VK_API int vk_event_elapsed_ms(void* start, void* stop, float* ms);  /* blocking on stop */

/* The input side of a frame (ref: include/vulcan/image.h:100-123 — Image::Load ends in a blocking cudaMemcpy —
 * called once per camera frame at apps/vulcan/vulcan.cu:220,232). Here a frame can be uploaded while the one
 * before it is being fused: the copy is enqueued on a stream of its own from pinned host memory (vk_malloc_host)
 * and returns at once; an ORDERING event (vk_event_create_ordering: no timing) recorded behind it is what the compute
 * stream waits for (vk_stream_wait_event) before the frame's first kernel, and an event recorded on the compute
 * stream behind the frame's last reader is what the copy stream waits for before it overwrites the image two frames
 * later. `publishes`: 1 when the work in front of the event WROTE what the waiter reads (the upload: default fences);
 * 0 when it only read what the waiter overwrites (the frame's kernels: no system-scope release — with one, every
 * record behind the integrate kernel wrote the device's caches back and the streamed frame took 357 us instead of
 * 114, tools/stream_input_probe.py). vk_event_synchronize blocks the HOST (before it refills a staging buffer).
 * vulcan_amd/host/include/vulcan/upload.h (FrameUploader) and bench.py --stream-input are built from these. */
VK_API int vk_memcpy_h2d_async(void* dst, const void* src_pinned, size_t bytes, void* stream);
VK_API int vk_event_create_ordering(void** event, int publishes);
VK_API int vk_stream_wait_event(void* stream, void* event);
VK_API int vk_event_synchronize(void* event);

/* ----------------------------------------------------------------- volume -- */

/* ref: src/volume.cu:552-627 Volume::Initialize + 9 Create* — fills voxels
 * with Voxel::Empty(), entries with HashEntry(), free list with 0..N-1,
 * types NONE, visibility FALSE; counters: visible=0, voxel_ptr=N-1,
 * excess_ptr=main. */
VK_API int vk_volume_initialize(const vk_volume* v, void* stream);

/* ref: src/volume.cu:465-471 Volume::ResetBlockVisibility — TRUE -> UNKNOWN */
VK_API int vk_volume_reset_block_visibility(const vk_volume* v, void* stream);

/* ref: src/volume.cu:87-301,497-518 Volume::CreateAllocationRequests.
 * One request per main bucket per call, as in the reference; where the
 * reference lets racing threads overwrite each other (last writer wins,
 * volume.cu:200,237) the winner here is the request with the largest packed
 * (pad=type,z,y,x) 64-bit key, chosen with one 64-bit atomic max, so a request
 * cannot tear and the outcome does not depend on timing. */
VK_API int vk_volume_create_allocation_requests(const vk_volume* v,
    const float* depth, int width, int height, const vk_projection* projection,
    const vk_transform* Twd, void* stream);

/* ref: src/volume.cu:304-368,520-535 Volume::HandleAllocationRequests.
 * Pool slots and excess indices are handed out in ascending bucket order
 * (an exclusive scan instead of the reference's atomicAdd/atomicSub order,
 * volume.cu:337,352), i.e. the outcome of running the reference's threads
 * serially; slot numbers are therefore reproducible. */
VK_API int vk_volume_handle_allocation_requests(const vk_volume* v, void* stream);

/* ref: src/volume.cu:25-84,473-495 Volume::UpdateBlockVisibility. Leaves the
 * count in counters[VK_CTR_VISIBLE]; does NOT read it back. The order of
 * visible_blocks is unspecified (as in the reference). */
VK_API int vk_volume_update_block_visibility(const vk_volume* v, int width,
    int height, const vk_projection* projection, const vk_transform* Tdw,
    void* stream);

/* ref: src/volume.cu:430-437 Volume::SetView = the four calls above. */
VK_API int vk_volume_set_view(const vk_volume* v, const vk_frame* frame, void* stream);

/* ref: src/volume.cu:537-543 Volume::GetBufferSize — blocking readback of the first
 * VK_CTR_PUBLIC counters into host memory (int32[VK_CTR_PUBLIC] or larger). */
VK_API int vk_volume_read_counters_sync(const vk_volume* v, int32_t* host_out, void* stream);

/* ------------------------------------------------------------- integrators -- */

/* Integrator parameters (ref: include/vulcan/integrator.h:38-44, integrator.cu:7-13) */
typedef struct vk_integrator {
  float min_depth;            /* depth_range_[0] = 0.1 */
  float max_depth;            /* depth_range_[1] = 5.0 */
  float max_distance_weight;  /* 16 */
  float max_color_weight;     /* 16 */
} vk_integrator;

/* ref: src/depth_integrator.cu:17-80,89-115 DepthIntegrator::Integrate
 * (= ColorIntegrator::IntegrateDepth color_integrator.cu:18-80,150-176
 *  = LightIntegrator::IntegrateDepth light_integrator.cu:106-168,295-321).
 * Visible blocks and their count are read from the volume on the device. */
VK_API int vk_integrate_depth(const vk_volume* v, const vk_integrator* p,
    const vk_frame* frame, void* stream);

/* ref: src/color_integrator.cu:83-135,178-204 ColorIntegrator::IntegrateColor */
VK_API int vk_integrate_color(const vk_volume* v, const vk_integrator* p,
    const vk_frame* frame, void* stream);

/* ref: src/color_integrator.cu:144-148 ColorIntegrator::Integrate — depth then
 * colour in ONE pass over the voxels (same values as the two reference passes:
 * the colour update reads the distance the depth update just wrote). */
VK_API int vk_integrate_depth_color(const vk_volume* v, const vk_integrator* p,
    const vk_frame* frame, void* stream);

/* ref: src/light_integrator.cu:17-103,277-293 LightIntegrator::ComputeFrameMask
 * (keeps the reference's asymmetric window). mask: [h*w] floats. */
VK_API int vk_light_compute_frame_mask(const vk_frame* frame, float depth_threshold,
    float* mask, void* stream);

/* ref: src/light_integrator.cu:277-293 LightIntegrator::ComputeFrameMask plus the part
 * of IntegrateColorKernel that depends on the PIXEL only (:215-225: the mask test and
 * the pixel's normal rotated into the colour camera, Tcd * n). `records`: device
 * float[h*w*4], 16-byte aligned, receives {Tcd * n (3), mask} per pixel, so the colour
 * kernel gathers one record per voxel instead of a mask and three normal components
 * and rotates nothing. `mask` is written as by vk_light_compute_frame_mask. Needs
 * frame->normals. */
VK_API int vk_light_prepare(const vk_frame* frame, float depth_threshold, float* mask,
    float* records, void* stream);

/* ref: src/light_integrator.cu:170-250,323-354 LightIntegrator::IntegrateColor */
VK_API int vk_integrate_light_color(const vk_volume* v, const vk_integrator* p,
    const vk_light* light, const float* mask, const vk_frame* frame, void* stream);

/* ref: src/light_integrator.cu:270-275 LightIntegrator::Integrate minus the
 * mask kernel: depth + light-colour update in one pass. */
VK_API int vk_integrate_depth_light(const vk_volume* v, const vk_integrator* p,
    const vk_light* light, const float* mask, const vk_frame* frame, void* stream);

/* ------------------------------------------------------------------ tracer -- */

/* ref: include/vulcan/tracer.cuh:14-18, src/tracer.cu:13-87,453-466
 * vulcan::ComputePatches. `block_count_dev` (device int*, may be NULL): when
 * non-NULL the number of visible blocks is read from it on the device and
 * `block_count` is only the upper bound used to size the grid. patch_count is
 * a device int the caller has zeroed (tracer.cpp:102-108). Patches are not
 * written past `patch_capacity`. Blocks with no valid depth interval emit no
 * patch (the reference writes patches[-1] there). */
VK_API int vk_trace_compute_patches(const int32_t* indices,
    const vk_hash_entry* entries, const vk_transform* Tcw,
    const vk_projection* projection, float block_length, float min_depth,
    float max_depth, int block_count, const int32_t* block_count_dev,
    int image_width, int image_height, int bounds_width, int bounds_height,
    vk_patch* patches, int patch_capacity, int32_t* patch_count, void* stream);

/* ref: tracer.cuh:20-21, src/tracer.cu:89-112,468-476 vulcan::ComputeBounds.
 * bounds: [bounds_h*bounds_w] float2 (near,far). patch_count_dev as above. */
VK_API int vk_trace_compute_bounds(const vk_patch* patches, float* bounds,
    int bounds_width, int patch_count, const int32_t* patch_count_dev, void* stream);

/* ref: tracer.cuh:32, src/tracer.cu:494-500 vulcan::ResetBoundsBuffer */
VK_API int vk_trace_reset_bounds(float* bounds, int count, void* stream);

/* Fused patches+bounds: every visible block rasterises its cell rectangle
 * straight into `bounds` (min/max are order independent, so the result is
 * bit-identical to ComputePatches + ComputeBounds). Resets bounds first. */
VK_API int vk_trace_compute_block_bounds(const int32_t* indices,
    const vk_hash_entry* entries, const vk_transform* Tcw,
    const vk_projection* projection, float block_length, float min_depth,
    float max_depth, int block_count, const int32_t* block_count_dev,
    int image_width, int image_height, int bounds_width, int bounds_height,
    float* bounds, void* stream);

/* ref: tracer.cuh:23-27, src/tracer.cu:114-451,478-492 vulcan::ComputePoints.
 * block_count = MAIN block count (the hash modulus K, tracer.cpp:85).
 * depths [h*w], colors [h*w*3]. */
VK_API int vk_trace_compute_points(const vk_hash_entry* entries,
    const vk_voxel* voxels, const float* bounds, int block_count,
    float block_length, float voxel_length, float trunc_length,
    const vk_transform* Twc, const vk_projection* projection, float* depths,
    float* colors, int image_width, int image_height, int bounds_width,
    int bounds_height, void* stream);

/* ref: tracer.cuh:29-30, src/frame.cu:9-122,183-192 vulcan::ComputeNormals
 * (= Frame::ComputeNormals frame.cpp:21-36). normals [h*w*3]. */
VK_API int vk_frame_compute_normals(const float* depths,
    const vk_projection* projection, float* normals, int image_width,
    int image_height, void* stream);

/* ref: src/frame.cu:126-181,194-203 vulcan::FilterDepths (7x7 bilateral). */
VK_API int vk_frame_filter_depths(int image_width, int image_height,
    const float* src, float* dst, void* stream);

/* Floats the `bounds` scratch of vk_trace must hold: the bounds grid itself
 * (tracer.cpp:140: 80*60 float2) followed by the per-workgroup private copies
 * the fused bounds pass reduces into and, since VK_ABI_VERSION 5, 2 048 words of
 * row counters for the raycast's normals (vk_trace_ahead_requests; one scratch
 * belongs to one vk_view_bounds record). */
VK_API size_t vk_trace_bounds_floats(int bounds_width, int bounds_height);

/* ref: src/tracer.cpp:41-47 Tracer::Trace — bounds (fused), points, normals.
 * `frame` supplies pose + intrinsics; outputs go to depth/color/normals
 * (all non-const device pointers owned by the caller). `bounds` is the
 * tracer's scratch, vk_trace_bounds_floats(w, h) floats; on return its first
 * w*h float2 hold the (near, far) grid of Tracer::bounds_. */
VK_API int vk_trace(const vk_volume* v, const vk_frame* frame, float min_depth,
    float max_depth, float* bounds, int bounds_width, int bounds_height,
    float* out_depth, float* out_color, float* out_normals, void* stream);

/* ------------------------------------------- light preparation, with SetView -- */

/* LightIntegrator's per-pixel preparation (vk_light_prepare: the frame mask and one
 * {Tcd * normal, mask} record per depth pixel) walks the depth image one lane per pixel,
 * exactly as the request pass of vk_volume_set_view does, and as a launch of its own most
 * of its ~6 us is the launch. An application that integrates the frame it has just set
 * the view from (the reference's frame loop, apps/vulcan/vulcan.cu:316-325) can have it
 * computed by that pass: vk_volume_set_view_prepare fills prep->mask / prep->records and
 * notes which frame they are for; the integrator asks (vk_light_prepared) and skips its
 * own pass on a match. "Which frame" means which CONTENT (vk_frame.content_id): upstream's
 * SetView reads the depth image only, so SetView(frame); frame.ComputeNormals();
 * Integrate(frame) is a legal sequence, and a preparation made before the normals changed
 * must not be used after. A frame whose content_id is 0 is never prepared ahead.
 * No reference counterpart; results are identical with or without it. */
typedef struct vk_light_prep {
  /* set by the caller */
  float  depth_threshold;      /* LightIntegrator::depth_threshold_ (light_integrator.cu:256) */
  float* mask;                 /* device float[capacity]                                       */
  float* records;              /* device float[4 * capacity], 16-byte aligned                  */
  /* One-shot request, reset to NULL by the call: the frame's normal image has NOT been computed yet
   * (Frame::ComputeNormals, frame.cu:9-122, still due) and vk_volume_set_view_prepare / _rounds is to
   * leave it in frame->normals — which this must equal — exactly as vk_frame_compute_normals would:
   * in its request pass when the preparation rides along (the 5-tap stencil reads the depth tile
   * that pass has staged anyway: one launch less per frame), else by that launch. */
  float* normals_out;
  int32_t capacity;            /* pixels the two buffers hold: a larger frame is not prepared  */
  /* set by vk_volume_set_view_prepare, compared by vk_light_prepared */
  int32_t      valid;
  int32_t      width, height;
  const float* depth;
  const float* color;
  const float* normals;
  float        prepared_threshold;
  vk_transform depth_to_color;
  uint64_t     content_id;     /* frame->content_id at the time; a frame with id 0 is never prepared */
} vk_light_prep;

/* ref: src/volume.cu:430-437 Volume::SetView (as vk_volume_set_view) + src/light_integrator.cu:
 * 277-293 — with a `prep` whose buffers are set and a frame that has colour and normals of
 * the depth image's size, the same launches also do vk_light_prepare. prep->valid
 * tells whether they did. (Two launches: the request pass; the handle pass and the visibility
 * pass together, the handle pass working from the list of buckets the request pass posted to.) */
VK_API int vk_volume_set_view_prepare(const vk_volume* v, const vk_frame* frame, vk_light_prep* prep,
    void* stream);

/* ref: apps/vulcan/vulcan.cu:316-318 — the frame loop calls SetView(frame) three times per frame,
 * because a bucket takes one request per call and a block that loses the contest for its bucket
 * (or finds its bucket's main entry taken by the winner) must ask again. This entry point leaves
 * exactly the state of `max_rounds` consecutive vk_volume_set_view(_prepare) calls with the same
 * frame — every buffer, the visible set, VK_CTR_VOXEL_PTR / EXCESS_PTR / REQUESTS / DROPPED — in the
 * two launches of ONE call. What a further call changes is only this: the blocks that lost ask
 * again, the winners among them are committed, their entries become visible. The request pass
 * therefore files the losers in a list, and — only when there are any — the workgroup of the second
 * launch that finishes last replays the later rounds from that list; the rays are walked once.
 * `prep` as in vk_volume_set_view_prepare (may be NULL). max_rounds >= 1. VK_CTR_ROUNDS counts the
 * rounds that ran; VK_CTR_UNSETTLED tells whether one more call would still have work.
 * Two cases end the rounds early, with VK_CTR_UNSETTLED = 1 and the state of as many calls as
 * rounds ran (VK_CTR_ROUNDS): a round that drops a request (pool or excess list exhausted —
 * upstream's calls after that link entries they never write, its state is inconsistent from
 * there on), and more than VK_RETRY_KEYS different blocks losing in one round. */
VK_API int vk_volume_set_view_rounds(const vk_volume* v, const vk_frame* frame, vk_light_prep* prep,
    int max_rounds, void* stream);

/* vk_volume_set_view_rounds with its two launches on two streams (round 4; no reference counterpart — upstream runs everything
 * on stream 0, device.h:40-52; ref: src/volume.cu:430-437 for the call). The request pass (volume.cu:497-518, with whatever rides in it: the light preparation, the
 * frame's normals) reads the depth image, the table and the visibility bytes and writes visibility bytes, request flags and
 * the preparation's buffers: nothing the raycast of the PREVIOUS frame reads or writes (vk_trace_ahead: table, voxels, its own
 * bounds and images). A caller that fuses at poses it knows before the previous frame's raycast has finished — the fusion +
 * raycast benchmark; not the tracking loop, whose pose comes out of that raycast — can therefore put the request pass on a stream
 * of its own and let it fill the tail of the previous raycast, whose last waves leave most of the device idle:
 *   vk_stream_wait_event(request_stream, integrated);   // the previous frame's vk_integrate_* has read the lists, mask, records
 *   vk_volume_set_view_rounds_split(&v, &f, &prep, 3, request_stream, ordering, stream);
 *   vk_integrate_ahead(.., stream);  vk_event_record(integrated, stream);  vk_trace_ahead(.., stream);
 * `ordering_event` (vk_event_create_ordering(&e, 1): the request pass WRITES what the waiter reads — visibility bytes, request
 * flags, the posted list, the preparation's buffers, the normals — so the event is one that publishes; `integrated` above is
 * vk_event_create_ordering(&e, 0): the integrate launch only read what the request pass overwrites) is recorded behind the
 * request pass on request_stream, and `stream` waits
 * for it before the handle + visibility pass, which must follow the previous raycast anyway (it writes table entries).
 * Same state as vk_volume_set_view_rounds, bit for bit (tests/test_gpu_configs.py runs bench.py's step, which uses this). */
VK_API int vk_volume_set_view_rounds_split(const vk_volume* v, const vk_frame* frame, vk_light_prep* prep, int max_rounds,
    void* request_stream, void* ordering_event, void* stream);

/* ref: src/light_integrator.cu:270-275 LightIntegrator::Integrate decides here whether
 * ComputeFrameMask still has to run: 1 if *prep holds the preparation of exactly `frame` (same
 * non-zero content_id, same image pointers and size, same depth->colour transform) at
 * `depth_threshold`, else 0. */
VK_API int vk_light_prepared(const vk_light_prep* prep, const vk_frame* frame, float depth_threshold);

/* ---------------------------------------------------- raycast bounds, ahead -- */

/* The first stage of a raycast (per-cell depth bounds of the visible blocks,
 * tracer.cpp:49-76) depends only on the visible list and the view, not on the
 * voxels, and it is a short latency-bound pass that keeps a few CUs busy. An
 * application that raycasts from the pose it has just integrated (the reference's
 * frame loop, apps/vulcan/vulcan.cu:316-325) can have it computed by a handful of
 * extra workgroups INSIDE the integrate launch, where it costs nothing, and let
 * the raycast skip it. This record carries the result and the view it is for.
 * No reference counterpart; results are identical with or without it. */
typedef struct vk_view_bounds {
  /* set by the caller */
  float*        scratch;          /* device, vk_trace_bounds_floats(bounds_width, bounds_height) floats */
  int32_t       bounds_width;     /* the tracer's grid (80 x 60, tracer.cpp:108-111) */
  int32_t       bounds_height;
  float         min_depth;        /* the tracer's depth range (tracer.cpp:103-106)   */
  float         max_depth;
  /* set by vk_integrate_ahead / vk_trace_ahead, compared by vk_trace_ahead. The
   * caller zeroes `valid` whenever the visible list changes: after
   * vk_volume_set_view and after the staged visibility calls. */
  int32_t       valid;
  int32_t       width, height;
  float         block_length;
  const void*   visible_blocks;   /* which volume */
  vk_projection projection;
  vk_transform  depth_to_world;
  /* kept by vk_trace_ahead / vk_trace_ahead_requests (the caller only zeroes them with the record): the raycast's normals
   * are computed by trailing workgroups of the SAME launch, which wait per 8-pixel row of tiles for the counters behind
   * the bounds in `scratch`; these say for which scratch and image size the counters count, and how many launches far */
  const float*  counted_scratch;
  int32_t       counted_width, counted_height;
  uint32_t      trace_launches;
  int32_t       pad_;
  /* (ABI 6) The waiting normals workgroups' wait is BOUNDED (a launch whose counters were left behind by an aborted
   * one must not hang the device). A group whose wait expires stores nothing, sets a word behind the counters in
   * `scratch` and — when the caller gave one — `*late_host`: one int32 of pinned host memory (vk_malloc_host), zeroed by
   * the caller once; the value stored is the number of the launch (1, 2, ... as this record counts them since its counters
   * were last zeroed, never 0) whose group expired (round 6). The library looks at it, without synchronising anything, at the start of every vk_trace_ahead* call
   * with this record, and vk_trace_normals_settle does after synchronising; see there for what happens then. The
   * `last_*` fields say which images the riding normals of the last launch belong to (set by the library).
   * ONE stream at a time per record: the counters' target counts the launches in stream order, so two traces in
   * flight on two streams through one record would wait for the wrong count. A call that names another stream than the
   * last one re-zeroes the counters on the new stream — the caller makes sure the old stream's launch has completed. */
  int32_t*      late_host;
  const float*  last_depths;
  float*        last_normals;
  int32_t       last_width, last_height;
  vk_projection last_projection;
  const void*   counted_stream;
} vk_view_bounds;

/* vk_integrate_depth / _depth_color / _depth_light (color_mode 0 / 1 / 2; `light`
 * and `mask` as in vk_integrate_depth_light for mode 2; `light_records`: optional
 * output of vk_light_prepare for the same frame, NULL = gather mask and normals
 * separately) which, when `ahead` is given, also computes the raycast bounds of
 * `frame`'s own view into ahead->scratch and records the view in *ahead.
 * ref: as those three. */
VK_API int vk_integrate_ahead(const vk_volume* v, const vk_integrator* p,
    const vk_frame* frame, int color_mode, const vk_light* light, const float* mask,
    const float* light_records, vk_view_bounds* ahead, void* stream);

/* A measurement aid (round 5; no reference counterpart — upstream times nothing on the device; ref: src/depth_integrator.cu:
 * 83-120, src/light_integrator.cu:296-340 for the launches it times): the NEXT pipelined integrate launch issued from this
 * host thread (vk_integrate_depth / _depth_color / _depth_light / vk_integrate_ahead) records `start_event` and `stop_event`
 * (vk_event_create) as the begin and the end OF THE DISPATCH ITSELF (hipExtLaunchKernelGGL) — the duration rocprofv3's kernel
 * trace reports. A caller that brackets the call with two vk_event_record instead also times the events' own processing and
 * the launch latency behind the first of them: 1.7 - 3.5 us on a 34 us launch, which is what separated bench.py's roofline
 * fraction from the kernel trace's until round 5. One CALL only: the pair is used up by the next vk_integrate_* call of this
 * thread even when that call returns an error before its launch (round 6: events left armed by a failed call were recorded by a
 * later, unrelated launch); (NULL, NULL) cancels. Not a stream operation: nothing is enqueued by this call. */
VK_API int vk_integrate_time_next(void* start_event, void* stop_event);

/* ref: src/tracer.cpp:41-47 Tracer::Trace, as vk_trace with the grid, depth range
 * and scratch taken from *ahead: when *ahead holds the bounds of this very view
 * the bounds pass is skipped, otherwise it runs and *ahead is updated. `normals` may be NULL (round 5): the normal image is
 * then left to the caller — e.g. to the next frame's vk_icp_pyramid_track_frame(.., frame_normals_due = 2 | .., ..). */
VK_API int vk_trace_ahead(const vk_volume* v, const vk_frame* frame,
    vk_view_bounds* ahead, float* depths, float* colors, float* normals, void* stream);

/* The request pass of a frame's SetView done AHEAD, in the previous frame's raycast launch (round 4; no reference
 * counterpart; ref: src/volume.cu:430-437,497-518 and src/tracer.cpp:41-47 for the two calls it joins). A raycast launch is as
 * long as its slowest wave and leaves most of the device idle for its last third; the request pass of the NEXT frame reads
 * and writes nothing the raycast touches (see vk_volume_set_view_rounds_split), so for a caller that knows the next frame — its
 * images and its pose — when it raycasts this one (fusion at given poses; not the tracking loop, whose next pose comes out of
 * this raycast) the pass can ride behind the raycast's workgroups in the SAME launch:
 *   vk_trace_ahead_requests(&v, &key_i, &bounds, depth, color, normals, &frame_next, &prep, &ahead, stream);   // Trace(i) + requests(i+1)
 *   ...
 *   vk_volume_set_view_rounds_ahead(&v, &frame_next, &prep, 3, &ahead, stream);    // SetView(i+1): the handle + visibility pass only
 * `ahead` is a caller-owned record (zero it once) that names the frame the pass was made for; vk_volume_set_view_rounds_ahead
 * uses the pass only for that very frame (same volume, images, size, intrinsics, pose and non-zero content_id, same riding
 * preparation); with a record that is not valid (zeroed, already used, or vk_trace_ahead_requests could not make the pass:
 * no content_id, normals asked for without a riding preparation) it runs the whole call, so the outcome is
 * vk_volume_set_view_rounds', bit for bit (tests/test_gpu_configs.py runs bench.py's step, which uses this). A VALID record
 * for another frame is refused (VK_ERR_ARGUMENT, nothing launched, the record kept): that frame's requests are in the volume,
 * and handled together with this frame's they would allocate in an order no sequence of upstream calls gives — the announced
 * frame has to be fused first. Between the two calls nothing else may run a SetView stage on the volume. `next_prep` as in vk_volume_set_view_prepare (may be NULL): the light preparation — and, with normals_out, the
 * next frame's normals — ride with the pass as they do in SetView. */
typedef struct vk_requests_ahead {
  const void*   counters;          /* the volume the pass was made on */
  const float*  depth;
  const void*   prep;              /* the vk_light_prep that rode along, or NULL */
  int32_t       width, height;
  vk_projection depth_projection;
  vk_transform  depth_to_world;
  uint64_t      content_id;
  int32_t       valid;
  int32_t       normals_made;      /* round 6 (was padding): 1 = the announced frame's normal image (next_frame->normals) was written
                                      with the pass (next_prep->normals_out rode) — or, set by the class layer, by a launch of its
                                      own in front of the announce; 0 = the announce left the normals as they were. A caller
                                      whose next step is "ComputeNormals, then SetView" may skip the normals only on 1 */
  int32_t       pose_on_device;    /* round 6: 1 = the pass was made by vk_volume_requests_at_device_pose — at the pose a tracker had
                                      left on the device, which the host did not have yet; vk_volume_set_view_rounds_ahead then takes
                                      the frame's pose on trust (the caller read it from that same device pose), and depth_to_world
                                      above is the Track's START pose (what vk_requests_ahead_cancel completes after an aborted Track) */
  int32_t       pad_;
} vk_requests_ahead;

VK_API int vk_trace_ahead_requests(const vk_volume* v, const vk_frame* view, vk_view_bounds* ahead, float* out_depth,
    float* out_color, float* out_normals, const vk_frame* next_frame, vk_light_prep* next_prep,
    vk_requests_ahead* requests, void* stream);
VK_API int vk_volume_set_view_rounds_ahead(const vk_volume* v, const vk_frame* frame, vk_light_prep* prep, int max_rounds,
    vk_requests_ahead* requests, void* stream);

/* The request pass of a TRACKED frame's SetView without the host round trip in front of it (round 6; no reference counterpart;
 * ref: src/volume.cu:430-437,497-518 Volume::SetView's first stage, apps/vulcan/vulcan.cu:300-318 for the loop). In the tracking
 * loop the pose of frame i comes out of Tracker::Track, the host waits for it (Tracker::EndSolve), and only then can it enqueue
 * SetView / Integrate / Trace — which need the pose as launch arguments: the device idles for the round trip, ~15 us of a 272 us
 * frame (profiles/r06_tracked_frame_timeline.txt). The request pass is the first launch behind that gap, and it needs the pose
 * for nothing but the rays' origin and direction: here it takes them from *pose_dev — the vk_transform the tracker's launches in
 * front of it in `stream` leave on the device (vk_icp_track*'s Twc_dev) — so the host enqueues it RIGHT BEHIND the Track, before
 * it waits, and has the whole pass (19 us) to pick up the pose and enqueue the rest:
 *   vk_icp_pyramid_track_frame(.., Twc_dev, .., stream);                                 // Track(i), enqueued
 *   vk_volume_requests_at_device_pose(&v, &frame, Twc_dev, &prep, &record, stream);      // requests(i), enqueued: no wait
 *   vk_track_wait(&poll, stream);  frame.depth_to_world = *poll.host_pose;               // Tracker::EndSolve
 *   vk_volume_set_view_rounds_ahead(&v, &frame, &prep, 3, &record, stream);              // handle + visibility only
 * `frame`: as for vk_volume_set_view_prepare, its depth_to_world = the Track's start pose (ignored by the pass; kept in the
 * record for vk_requests_ahead_cancel). `prep` rides as in SetView (the frame's normals must exist: normals_out is refused,
 * VK_ERR_UNSUPPORTED). The same requests as SetView's own pass at the pose the host then reads: the same state, bit for bit
 * (tests/test_gpu_round6.py). A Track that ends with VK_TRACK_ABORTED has left its start pose in *pose_dev: the record then
 * announces a SetView at the start pose — vk_requests_ahead_cancel completes it (a state upstream reaches by calling SetView
 * with the untracked frame), after which the staged Track and the frame's own SetView follow. */
VK_API int vk_volume_requests_at_device_pose(const vk_volume* v, const vk_frame* frame, const vk_transform* pose_dev,
    vk_light_prep* prep, vk_requests_ahead* requests, void* stream);

/* ref: src/volume.cu:430-437 Volume::SetView (x max_rounds, as vk_volume_set_view_rounds) — the WHOLE call at the pose on the
 * device: the request pass as above and the handle + visibility launch with its frustum test's world -> depth transform taken
 * from pose_dev->inv. Nothing of SetView then waits for the host: a tracking loop enqueues it right behind the Track and has
 * both launches (28 us) to pick the pose up (vk_track_wait) for Integrate and Trace — which found the device idle for 6 us even
 * behind the request pass alone (profiles/r06_tracked_frame_timeline.txt). `frame->depth_to_world` is ignored. No record: there
 * is no later SetView call to match. After a Track that ABORTED the volume has seen SetView(frame at the Track's start pose) —
 * a state upstream reaches — and the caller's staged Track and the frame's own SetView follow. VK_ERR_UNSUPPORTED (nothing
 * launched): prep->normals_out set, or vk_test_hooks.set_view_unfused. Same state as vk_volume_set_view_rounds with the pose
 * the host then reads, bit for bit (tests/test_gpu_round6.py). */
VK_API int vk_volume_set_view_at_device_pose(const vk_volume* v, const vk_frame* frame, const vk_transform* pose_dev,
    vk_light_prep* prep, int max_rounds, void* stream);

/* A record that is still valid when vk_trace_ahead_requests is called again names a frame whose requests are in the volume
 * and whose SetView has not run: a second pass on top would mix two frames' requests, retry keys, posted list and touched
 * bits — the state vk_volume_set_view_rounds_ahead refuses. vk_trace_ahead_requests therefore returns VK_ERR_ARGUMENT for a
 * record with valid == 1 and launches nothing (round 5), and the staged SetView stages of the class layers refuse to run
 * while their volume's record is valid. The way out of such a record — one a SetView refused, or a frame the caller no
 * longer wants to fuse — is this call: it completes the ANNOUNCED frame's SetView (the handle + visibility pass with the
 * record's own size, intrinsics and pose: everything that pass needs; `max_rounds` as there) and marks the record used. The
 * volume is then exactly where `max_rounds` upstream SetView calls with the announced frame leave it — a state upstream can
 * reach — and any frame may follow. A record that is not valid: VK_OK, nothing launched. A record of another volume:
 * VK_ERR_ARGUMENT. ref: src/volume.cu:430-437 (the SetView it completes), :520-535, :473-495. */
VK_API int vk_requests_ahead_cancel(const vk_volume* v, vk_requests_ahead* requests, int max_rounds, void* stream);

/* The outcome of the bounded wait of the raycast's riding normals (vk_view_bounds.late_host; round 5; no reference
 * counterpart; ref: include/vulcan/device.h:14-17 CUDA_ASSERT — the contract it keeps is upstream's "a failed device step
 * always surfaces" — and src/frame.cu:9-122 for the normals it recomputes). Synchronises `stream`, then looks at the expiry word (the pinned one,
 * or — without one — the word behind the counters, by a blocking copy). Nothing expired: VK_OK. Otherwise the riding
 * normals of A launch since the last check are incomplete: the stream the counters counted on is drained, the counters and
 * both words are re-zeroed, the launch count starts over, Frame::ComputeNormals of the LAST traced image is enqueued on
 * `stream` as a launch of its own (ahead->last_*), and the call returns VK_ERR_TIMEOUT. What is guaranteed is the error: the
 * caller knows its counters had been left in a bad state. The image repaired is the most recent one; a caller that runs
 * several launches ahead of this check and kept an EARLIER image (the pinned word held the number of the launch that
 * expired until this call zeroed it) recomputes that image's normals itself (vk_frame_compute_normals). Every vk_trace_ahead / vk_trace_ahead_requests call makes the same check first, on the pinned word
 * only and without synchronising; when it finds the word set it repairs in the same way, launches NOTHING of its own and
 * returns VK_ERR_TIMEOUT — the caller repeats the call. */
VK_API int vk_trace_normals_settle(vk_view_bounds* ahead, void* stream);

/* ------------------------------------------------------------------- image -- */

/* ref: src/image.cu:101-165,183-211 Image::Downsample (nearest or 2x2 box) */
VK_API int vk_image_downsample(int src_w, int src_h, const float* src,
    float* dst, int nearest, void* stream);

/* ref: src/image.cu:133-165,234-262 ColorImage::Downsample */
VK_API int vk_color_image_downsample(int src_w, int src_h, const float* src,
    float* dst, int nearest, void* stream);

/* ref: src/frame.cpp:38-51 Frame::Downsample's three image passes in one launch: depth and
 * normals nearest, colour 2x2 box (exactly vk_image_downsample / vk_color_image_downsample).
 * An output that is NULL is skipped; depth and normals have the size frame->width x height,
 * the colour image its own (color_width x color_height, 0 = the same); all sizes even. */
VK_API int vk_frame_downsample(const vk_frame* frame, float* depth_out, float* color_out,
    float* normals_out, void* stream);

/* --------------------------------------------------------------------- ICP -- */

/* One side of the ICP problem (keyframe or frame). */
typedef struct vk_icp_view {
  const float*  depths;      /* [h*w]   */
  const float*  normals;     /* [h*w*3] */
  int32_t       width;
  int32_t       height;
  vk_projection projection;
} vk_icp_view;

/* ref: src/depth_tracker.cu:97-118,272-300 DepthTracker::ComputeResiduals */
VK_API int vk_icp_compute_residuals(const vk_icp_view* keyframe,
    const vk_transform* Twm, const vk_icp_view* frame, const vk_transform* Twc,
    float* residuals, void* stream);

/* ref: src/depth_tracker.cu:120-141,302-336 DepthTracker::ComputeJacobian.
 * jacobian: [h*w*6] floats; columns 3..5 are 0 when !translation_enabled. */
VK_API int vk_icp_compute_jacobian(const vk_icp_view* keyframe,
    const vk_transform* Twm, const vk_icp_view* frame, const vk_transform* Twc,
    int translation_enabled, float* jacobian, void* stream);

/* ref: src/depth_tracker.cu:144-268,338-378 DepthTracker::ComputeSystem.
 * hessian: device float[36] — the first 21 (6 when !translation_enabled) hold
 * the packed lower triangle, row-major (r, c<=r), rest zero; gradient: device
 * float[6]. Both are overwritten (the reference zero-fills then atomically
 * accumulates; this is a two-stage fixed-order reduction, so the sums are
 * reproducible). `workspace`: device float[vk_icp_workspace_floats(w,h)].
 * `Twc_dev`: optional DEVICE pointer to a vk_transform that overrides Twc
 * (lets a device-side Gauss-Newton loop run without host round trips). */
VK_API size_t vk_icp_workspace_floats(int width, int height);
VK_API int vk_icp_compute_system(const vk_icp_view* keyframe,
    const vk_transform* Twm, const vk_icp_view* frame, const vk_transform* Twc,
    const vk_transform* Twc_dev, int translation_enabled, float* workspace,
    float* hessian, float* gradient, void* stream);

/* ref: src/tracker.cpp:124-163 Tracker::ComputeUpdate + src/depth_tracker.cpp:
 * 22-86 DepthTracker::ApplyUpdate, on the device: unpack the packed lower
 * triangle, LDLT-solve x = -H^-1 g, compose Tinc*Twc, re-orthonormalise, write
 * the new transform to *Twc_dev. state_dev: device int[2] = {iteration, done};
 * once ||x|| < 1e-6 (tracker.cpp:162) `done` is set and later calls become
 * no-ops (the pose stops changing), so a fixed-length device-side loop gives
 * the same pose as the reference's early exit.
 * update_dev (optional): device float[6] receiving x. */
VK_API int vk_icp_solve_update(const float* hessian, const float* gradient,
    int translation_enabled, vk_transform* Twc_dev, int32_t* state_dev,
    float* update_dev, void* stream);

/* Called between the system and the solve of every iteration with the packed
 * device system (48 floats: hessian[36], gradient[6], pad): a multi-GPU rig sums
 * it over ranks here (ncclAllReduce on `stream`). Returns 0 on success. */
typedef int (*vk_icp_reduce_fn)(float* system_dev, int count, void* user, void* stream);

/* A reduce hook that changes nothing. Passing it selects the launch-per-stage loop of a tracker
 * on one GPU (ref: src/tracker.cpp:53-63, one ComputeSystem + ComputeUpdate per iteration): what a
 * host falls back to when a one-launch loop ends with VK_TRACK_ABORTED. */
VK_API int vk_reduce_nothing(float* system_dev, int count, void* user, void* stream);

/* *dst_dev = *src_host, stream-ordered and without a host synchronisation (the 128
 * bytes travel as kernel arguments): how a tracker's device-side pose is seeded
 * from frame.depth_to_world_transform (ref: src/tracker.cpp:70-76 BeginSolve). */
VK_API int vk_transform_upload(vk_transform* dst_dev, const vk_transform* src_host, void* stream);

/* Early exit for the device-side Gauss-Newton loops. The reference leaves its loop
 * as soon as |update| < 1e-6 (tracker.cpp:162) because its host sees every update.
 * The loops here are enqueued without a host round trip; steps after convergence
 * are no-ops but still cost their launches. With a vk_track_poll the solve also
 * publishes {iterations, converged} to `host_state` (ONE 64-bit store to pinned host
 * memory, vk_malloc_host), and the enqueuing function looks at it after every
 * `chunk` steps and stops once the loop has converged: the pose is the same, the
 * call blocks for at most one chunk, and no launch follows convergence by more
 * than 2 * chunk - 1 steps. The number of steps enqueued depends only on the step at
 * which the loop converged — not on timing — so all ranks of a multi-GPU rig (whose
 * steps carry an all-reduce each) enqueue the same number. NULL (or chunk <= 0):
 * enqueue every step, never block. */
typedef struct vk_track_poll {
  int32_t* host_state;   /* pinned int32[4], 8-byte aligned, zeroed once by the caller: [0..1] the
                            device's word {steps, converged | call tag}, [2] the library's call counter,
                            [3] the tag of the last call whose pose has arrived in host_pose             */
  int32_t  chunk;        /* steps enqueued between two looks at host_state (0: never look)               */
  vk_transform* host_pose;  /* optional, pinned (vk_malloc_host): every Track also leaves its final pose
                            here (for the colour trackers: depth_to_world); see vk_track_wait            */
} vk_track_poll;

/* ref: src/tracker.cpp:78-82 Tracker::EndSolve — the pose of the last Track issued with `poll` is in
 * poll->host_pose when this returns VK_OK. It watches one word of pinned memory that the last kernel
 * of the Track writes after the pose (a few microseconds after the kernel is done) instead of copying
 * the pose and synchronising the stream. VK_ERR_UNSUPPORTED: `stream` drained without the pose having
 * been left (a Track that failed, or none was issued with this `poll`). */
VK_API int vk_track_wait(const vk_track_poll* poll, void* stream);

/* ref: src/tracker.cpp:53-63 Tracker::Track for DepthTracker — up to `iterations`
 * Gauss-Newton steps without a host round trip, ending early once |update| < 1e-6
 * (tracker.cpp:162). `system`: device float[48]; `state_dev`: device int[2]
 * {steps run, converged}, the caller zeroes it (a state that says "converged" makes
 * the call a no-op); `workspace`: vk_icp_workspace_floats(w, h) floats.
 *
 * Without a `reduce` hook the whole loop is ONE launch: the workgroups exchange
 * their 27 sums inside the launch after every step, each adds all of them in a fixed
 * order and solves the 6x6 system itself (redundantly, hence identically), workgroup 0
 * publishes pose, system and state at the end (a loop of more than 32 steps continues in
 * further launches of 32). `poll` is not
 * needed for an early exit on this path (its mirror still receives the final state).
 * Should the device be unable to hold the launch's workgroups at the same time — not
 * expected: the grid is sized from the occupancy query — the launch gives up after two
 * seconds and leaves state_dev[1] = VK_TRACK_ABORTED; no pose is published then (vk_track_wait
 * returns VK_ERR_UNSUPPORTED), a later level of the same coarse-to-fine Track does not run,
 * and the host's way out is the same call again from the start pose with `reduce` =
 * vk_reduce_nothing, i.e. the launch-per-stage loop, which waits for nobody (the class layer and
 * vulcan_amd/api.py do exactly that). With vk_test_hooks.loop_cooperative set the loop
 * kernels are launched with hipLaunchCooperativeKernel, which refuses a grid that cannot be
 * resident instead of letting the kernel find out. Tracks issued on different streams of
 * one device are run one after the other by the library (a loop launch fills the device); two
 * processes that share a device are not protected from starving each other.
 *
 * With a `reduce` hook (multi-GPU rig) a step is three launches — partial sums, sum,
 * [the hook's all-reduce], solve — and `poll` stops the enqueuing after convergence. */
#define VK_TRACK_ABORTED (-1)
VK_API int vk_icp_track(const vk_icp_view* keyframe, const vk_transform* Twm,
    const vk_icp_view* frame, vk_transform* Twc_dev, int iterations,
    int translation_enabled, float* workspace, float* system, int32_t* state_dev,
    float* update_dev, vk_icp_reduce_fn reduce, void* reduce_user, const vk_track_poll* poll,
    void* stream);

/* A rigid multi-camera rig, one camera per GPU (BASELINE configs[4]; no reference counterpart): the
 * ranks' views contribute to ONE normal system, so after every rank has summed its own view the 27
 * sums are added over ranks and every rank runs the same solve. With `reduce` that costs three
 * launches and an all-reduce per Gauss-Newton step; with a vk_rig_exchange the one-launch loop is
 * kept: one workgroup of every rank writes its 27 {tag, value} words straight into every peer's
 * memory (areas mapped over xGMI by vk_comm_exchange_attach, include/vk_comm.h), all workgroups
 * read their own GPU's area, add the ranks' words in rank order (the same bits everywhere) and
 * solve. Protocol and its proof obligations: vulcan_amd/csrc/vk_rig_protocol.h. */
#define VK_RIG_MAX_RANKS 8
typedef struct vk_rig_exchange {
  unsigned long long* areas[VK_RIG_MAX_RANKS]; /* areas[r]: rank r's area as mapped into THIS process (areas[rank]:
                                                  its own); each of the size vk_rig_area_bytes returns, zeroed once, fine-grained */
  int32_t  rank, world;
  uint32_t sequence;      /* names the Track: the same on every rank, 22 bits, never 0; after EVERY Track it entered
                             (an aborted one too) a rank moves it on with vk_comm_exchange_next_sequence
                             (vk_comm.h): 1, 2, ... 2^22 - 2, 1, ... — the parity alternates across the wrap */
} vk_rig_exchange;

/* bytes of one rank's area (ref: none — the reference has no multi-GPU code) */
VK_API size_t vk_rig_area_bytes(void);

/* ref: src/tracker.cpp:53-63 Tracker::Track for DepthTracker on a rig: vk_icp_track (no `reduce`: the whole
 * loop is one launch per rank) with the ranks' sums added through `rig` after every step. Every rank
 * calls it for the same Track with the same sequence number, iterations (<= 1000) and
 * translation_enabled; world == 1 gives vk_icp_track's bits. A rank that waits two seconds for a peer
 * ends with VK_TRACK_ABORTED. */
VK_API int vk_icp_track_rig(const vk_icp_view* keyframe, const vk_transform* Twm,
    const vk_icp_view* frame, vk_transform* Twc_dev, int iterations,
    int translation_enabled, float* workspace, float* system, int32_t* state_dev,
    float* update_dev, const vk_rig_exchange* rig, const vk_track_poll* poll, void* stream);

/* ref: src/pyramid_tracker.cpp:52-90 PyramidTracker<DepthTracker>::Track — the half-
 * resolution level of both frames (Frame::Downsample, src/frame.cpp:38-58: nearest
 * depth and normals, intrinsics / 2; ONE launch for the four images), vk_icp_track
 * with 15 steps on it, then 20 steps at full resolution from the pose the half
 * level left (one launch per level when there is no `reduce` hook). The quarter level
 * upstream builds and never tracks (:64-77) is not built. `pyramid`: device float[vk_icp_pyramid_floats(...)] for the half-resolution
 * images; `workspace`: vk_icp_workspace_floats of the FULL frame size; the other
 * buffers as in vk_icp_track. state_dev is reset before each level and holds the
 * full-resolution level's {steps, converged} afterwards. Image sizes must be even. */
VK_API size_t vk_icp_pyramid_floats(int key_width, int key_height, int frame_width, int frame_height);
VK_API int vk_icp_pyramid_track(const vk_icp_view* keyframe, const vk_transform* Twm,
    const vk_icp_view* frame, vk_transform* Twc_dev, float* pyramid, float* workspace,
    float* system, int32_t* state_dev, float* update_dev, vk_icp_reduce_fn reduce,
    void* reduce_user, const vk_track_poll* poll, void* stream);

/* vk_icp_pyramid_track for the frame loop (apps/vulcan/vulcan.cu:297-311: frame.ComputeNormals(); tracker->Track(frame)), with
 * two launches of that sequence folded into the pyramid's own: `Twc_start` (host, may be NULL) is stored to Twc_dev by the
 * pyramid launch (instead of vk_transform_upload before the call), and with `frame_normals_due` the frame's normal image
 * frame->normals is COMPUTED by it (ref: src/frame.cu:9-122 Frame::ComputeNormals — vk_frame_compute_normals' bits; the
 * half-resolution normals are computed at the pixels they are sampled from, which is the same normal) instead of read.
 * `frame_normals_due` is a set of bits: 1 = the frame's normal image as above; 2 (round 5) = the KEYFRAME's — the raycast's
 * normal image, which Tracer::Trace otherwise computes with a launch of its own right behind the raycast (src/tracer.cpp:97-100)
 * and which nobody reads before this Track: vk_trace_ahead(.., normals = NULL, ..) leaves it out, and this call writes
 * keyframe->normals (the same bits) on the way. The caller that skips the raycast's normals owes them to whoever reads the
 * key frame before the next Track. Everything else as vk_icp_pyramid_track. */
VK_API int vk_icp_pyramid_track_frame(const vk_icp_view* keyframe, const vk_transform* Twm,
    const vk_icp_view* frame, vk_transform* Twc_dev, const vk_transform* Twc_start, int frame_normals_due,
    float* pyramid, float* workspace, float* system, int32_t* state_dev, float* update_dev,
    vk_icp_reduce_fn reduce, void* reduce_user, const vk_track_poll* poll, void* stream);

/* The NEXT Track's pyramid behind the raycast (round 6; no reference counterpart; ref: src/tracer.cpp:41-47,97-100 for the
 * raycast it rides in, src/pyramid_tracker.cpp:52-62 + src/frame.cpp:21-58 for what it builds). In the tracking loop
 * (apps/vulcan/vulcan.cu:297-325) the launch that builds a Track's pyramid — the input frame's normal image and half-resolution
 * level, the key frame's normal image and half-resolution level — sits between the raycast and the Gauss-Newton loops, 7 us of
 * launch-floor work in a strict chain; most of it needs only the INPUT frame, which a caller has while the previous frame is still
 * being raycast. vk_trace_ahead_pyramid is vk_trace_ahead(v, frame, ahead, depths, colors, normals) with that work done by
 * trailing workgroups of the raycast's own launch: next_frame->normals (written: Frame::ComputeNormals of next_frame->depths),
 * the half-resolution level of next_frame and — behind the raycast's row counters, as the riding normals of
 * vk_trace_ahead_requests — `normals` and the half-resolution level of the traced image, into `pyramid`
 * (vk_icp_pyramid_floats(frame, next_frame) floats, vk_icp_pyramid_track's layout). *built (caller-owned, zeroed once) then names
 * the images and the buffer; vk_icp_pyramid_track_built is vk_icp_pyramid_track_frame that skips its pyramid launch when *built
 * is valid for exactly its keyframe (depths, normals = the traced images), frame and `pyramid` — and is the whole call, with
 * `frame_normals_due` as given, otherwise. A record serves once. Same images, same level, same pose, bit for bit
 * (tests/test_gpu_round6.py). All sizes even. When the launch cannot carry the work (a bounds grid too large for the row
 * counters) it is vk_trace_ahead and *built stays invalid. The key side's wait is bounded like the riding normals': on expiry
 * nothing is stored, ahead->late_host is set and the next vk_trace_ahead* / vk_trace_normals_settle returns VK_ERR_TIMEOUT after
 * recomputing the normal image — and the Track that consumed the level in between is not to be trusted (its key level was
 * incomplete): the caller repeats it from its start pose with vk_icp_pyramid_track_frame(.., frame_normals_due = 1 | 2, ..). */
typedef struct vk_pyramid_ahead {
  const float* key_depths;     /* the traced image: the next Track's keyframe */
  const float* key_normals;
  const float* frame_depths;   /* the next Track's frame */
  const float* frame_normals;
  const float* pyramid;
  int32_t      key_width, key_height, frame_width, frame_height;
  int32_t      valid;
  int32_t      pad_;
} vk_pyramid_ahead;

/* ref: src/tracer.cpp:41-47,97-100 Tracer::Trace + what src/pyramid_tracker.cpp:58-62 builds next (see above) */
VK_API int vk_trace_ahead_pyramid(const vk_volume* v, const vk_frame* frame, vk_view_bounds* ahead, float* depths, float* colors,
    float* normals, const vk_icp_view* next_frame, float* pyramid, vk_pyramid_ahead* built, void* stream);
/* ref: src/pyramid_tracker.cpp:52-90 PyramidTracker<DepthTracker>::Track, as vk_icp_pyramid_track_frame (see above) */
VK_API int vk_icp_pyramid_track_built(const vk_icp_view* keyframe, const vk_transform* Twm, const vk_icp_view* frame,
    vk_transform* Twc_dev, const vk_transform* Twc_start, int frame_normals_due, vk_pyramid_ahead* built, float* pyramid,
    float* workspace, float* system, int32_t* state_dev, float* update_dev, vk_icp_reduce_fn reduce, void* reduce_user,
    const vk_track_poll* poll, void* stream);

/* ------------------------------------------------------------ colour tracker -- */

/* ref: src/image.cu:10-19,235-247 ColorImage::ConvertTo — intensity = (r+g+b)/3.
 * src: [total*3] floats, dst: [total]. */
VK_API int vk_color_image_convert(int total, const float* src, float* dst, void* stream);

/* ref: src/image.cu:21-99,166-179 Image::GetGradients — 3x3 stencil
 * (1/8, 1/4, 1/8 rows), zero padding outside the image. */
VK_API int vk_image_gradients(int width, int height, const float* src,
    float* gradient_x, float* gradient_y, void* stream);

/* One side of the photometric problem. The keyframe needs depths, normals and
 * intensities; the frame also needs the two gradient images. */
typedef struct vk_color_view {
  const float*  depths;        /* [h*w]   */
  const float*  normals;       /* [h*w*3] */
  const float*  intensities;   /* [h*w]   */
  const float*  gradient_x;    /* [h*w], frame only */
  const float*  gradient_y;    /* [h*w], frame only */
  int32_t       width;
  int32_t       height;
  vk_projection projection;    /* the COLOUR camera's, color_tracker.cu:309-310 */
} vk_color_view;

/* ref: src/color_tracker.cu:43-163,296-344 ColorTracker::ComputeResiduals. One
 * residual per KEYFRAME pixel. Tcm = frame_Tcw * keyframe_Tcw^-1 with
 * X_Tcw = X.depth_to_color * X.depth_to_world^-1 (color_tracker.cu:312-320). */
VK_API int vk_color_tracker_compute_residuals(const vk_color_view* keyframe,
    const vk_color_view* frame, const vk_transform* Tcm, float* residuals, void* stream);

/* ref: src/color_tracker.cu:165-204,346-410 ColorTracker::ComputeJacobian;
 * jacobian: [h*w*6] floats, columns 3..5 are 0 when !translation_enabled. */
VK_API int vk_color_tracker_compute_jacobian(const vk_color_view* keyframe,
    const vk_color_view* frame, const vk_transform* Tcm, int translation_enabled,
    float* jacobian, void* stream);

/* ref: src/color_tracker.cu:206-343,412-470 ColorTracker::ComputeSystem; same
 * outputs, workspace and fixed-order reduction as vk_icp_compute_system
 * (workspace: vk_icp_workspace_floats(keyframe w, h) floats). `Tcm_dev`:
 * optional DEVICE transform that overrides Tcm. */
VK_API int vk_color_tracker_compute_system(const vk_color_view* keyframe,
    const vk_color_view* frame, const vk_transform* Tcm, const vk_transform* Tcm_dev,
    int translation_enabled, float* workspace, float* hessian, float* gradient, void* stream);

/* The frame pose as the device-side loop keeps it. */
typedef struct vk_color_pose {
  vk_transform depth_to_world;  /* frame.depth_to_world_transform, updated in place */
  vk_transform Tcm;             /* derived: what the kernels above consume         */
} vk_color_pose;

/* ref: src/tracker.cpp:124-163 Tracker::ComputeUpdate + src/color_tracker.cpp:
 * 34-96 ColorTracker::ApplyUpdate on the device: solve, M = Tinc *
 * depth_to_world^-1, re-orthonormalise, depth_to_world = (T(t) R)^-1, then
 * Tcm = (frame_Tcd * depth_to_world^-1) * keyframe_Twc for the next iteration.
 * `keyframe_Twc` = (keyframe.depth_to_color * keyframe.depth_to_world^-1)^-1.
 * state_dev / update_dev as in vk_icp_solve_update. */
VK_API int vk_color_tracker_solve_update(const float* hessian, const float* gradient,
    int translation_enabled, const vk_transform* frame_Tcd, const vk_transform* keyframe_Twc,
    vk_color_pose* pose_dev, int32_t* state_dev, float* update_dev, void* stream);

/* ref: src/tracker.cpp:65-76 Tracker::BeginSolve + src/color_tracker.cpp:19-25 ColorTracker::
 * BeginSolve (+ src/light_tracker.cpp:34-41 with `frame_mask`) in ONE launch: the keyframe's and
 * the frame's intensity images (vk_color_image_convert), the frame's gradients
 * (vk_image_gradients), the light tracker's frame mask (vk_light_compute_frame_mask at
 * `depth_threshold`; NULL for the colour tracker), the pose upload into
 * pose_dev->depth_to_world (vk_transform_upload; both NULL to skip) and the reset of
 * state_dev (may be NULL; without a pose upload — a later level of a coarse-to-fine Track — a
 * state that says VK_TRACK_ABORTED is kept, so that the Track fails as a whole). Same bits as the staged calls: the jobs are independent once the
 * gradients take their taps from the colour image, and run side by side. */
VK_API int vk_color_tracker_begin(const vk_frame* keyframe, const vk_frame* frame,
    float* keyframe_intensities, float* frame_intensities, float* gradient_x, float* gradient_y,
    float depth_threshold, float* frame_mask, const vk_transform* pose_host,
    vk_color_pose* pose_dev, int32_t* state_dev, void* stream);

/* ref: src/tracker.cpp:53-63 Tracker::Track for ColorTracker: derive Tcm from
 * pose_dev->depth_to_world, then up to `iterations` Gauss-Newton steps without a host
 * round trip — one launch for the whole loop, as vk_icp_track (three launches per step
 * with `reduce`). Buffers, early exit and VK_TRACK_ABORTED as in vk_icp_track; `workspace`:
 * vk_icp_workspace_floats(keyframe w, h). */
VK_API int vk_color_tracker_track(const vk_color_view* keyframe, const vk_color_view* frame,
    const vk_transform* frame_Tcd, const vk_transform* keyframe_Twc, vk_color_pose* pose_dev,
    int iterations, int translation_enabled, float* workspace, float* system,
    int32_t* state_dev, float* update_dev, vk_icp_reduce_fn reduce, void* reduce_user,
    const vk_track_poll* poll, void* stream);

/* ------------------------------------------------------------- light tracker -- */

/* What LightTracker adds to the colour tracker's inputs (light_tracker.h:52-66):
 * the frame's validity mask, the point light, and the frame's depth->colour
 * extrinsics (its normals are stored in the depth frame). The keyframe's
 * `intensities` are read as albedos. */
typedef struct vk_light_terms {
  const float*  frame_mask;   /* [h*w], vk_light_compute_frame_mask of the frame (light_tracker.cu:19-103 is the same kernel) */
  vk_light      light;
  vk_transform  frame_Tcd;
} vk_light_terms;

/* ref: src/light_tracker.cu:133-330,566-608 LightTracker::ComputeResiduals. Where
 * the mask is set the residual is photometric with a shading model,
 * Ic - albedo * light.GetShading(Xcp, n); elsewhere it falls back to the
 * point-to-plane distance (:283-322). The reference's ComputeResiduals reads
 * frame_mask_ without computing it (:569-577); here the mask is an input. */
VK_API int vk_light_tracker_compute_residuals(const vk_color_view* keyframe,
    const vk_color_view* frame, const vk_light_terms* terms, const vk_transform* Tcm,
    float* residuals, void* stream);

/* ref: src/light_tracker.cu:332-371,610-668 LightTracker::ComputeJacobian. The
 * reference evaluates machine-generated expressions (powf / sqrt chains,
 * :233-242); here the same derivative is written in factored form
 * (J_v = grad_p r, J_w = p x grad_p r + n x grad_n r), so values agree with the
 * reference to rounding, not bit for bit — its own finite-difference test
 * (light_tracker_test.cu:454-528, |f - e| < 0.05) is the pin. */
VK_API int vk_light_tracker_compute_jacobian(const vk_color_view* keyframe,
    const vk_color_view* frame, const vk_light_terms* terms, const vk_transform* Tcm,
    int translation_enabled, float* jacobian, void* stream);

/* ref: src/light_tracker.cu:373-531,670-732 LightTracker::ComputeSystem; outputs
 * and workspace as vk_color_tracker_compute_system. */
VK_API int vk_light_tracker_compute_system(const vk_color_view* keyframe,
    const vk_color_view* frame, const vk_light_terms* terms, const vk_transform* Tcm,
    const vk_transform* Tcm_dev, int translation_enabled, float* workspace,
    float* hessian, float* gradient, void* stream);

/* ref: src/tracker.cpp:53-63 Tracker::Track for LightTracker. The pose update is
 * ColorTracker's (light_tracker.cpp:50-114 is the same code as
 * color_tracker.cpp:34-96): vk_color_tracker_solve_update, vk_color_pose. */
VK_API int vk_light_tracker_track(const vk_color_view* keyframe, const vk_color_view* frame,
    const vk_light_terms* terms, const vk_transform* keyframe_Twc, vk_color_pose* pose_dev,
    int iterations, int translation_enabled, float* workspace, float* system,
    int32_t* state_dev, float* update_dev, vk_icp_reduce_fn reduce, void* reduce_user,
    const vk_track_poll* poll, void* stream);

/* ---------------------------------------------------------------- detector -- */

/* ref: include/vulcan/detector.h:10-72, src/detector.cu — box detector over a
 * point cloud (nothing upstream instantiates it).
 * `bounds[a]` = {lo, hi}; an axis with lo > hi is unbounded (detector.cu:214-221
 * starts all three at {1,-1}). */
typedef struct vk_detector {
  float   radius;             /* <= 0: no radius test                          */
  float   origin[3];
  float   bounds[3][2];
  int32_t min_inlier_count;   /* fewer survivors => position is NaN            */
  int32_t bounds_use_own_axis;/* 0 = as upstream: the y and z intervals are
                                 tested against point[0] (detector.cu:27-28);
                                 1 = test point[1] / point[2]                  */
} vk_detector;

/* What the detector leaves on the device after a call (one blocking copy reads
 * it all). Sums are fixed-order (4096-point chunks, 256-way strided partials,
 * binary tree, chunks added in order), so every field is reproducible. */
typedef struct vk_detect_state {
  int32_t filtered_count;     /* after the radius / interval test              */
  int32_t inlier_count;       /* after the 1.5-sigma removal                   */
  int32_t detected;           /* inlier_count >= min_inlier_count              */
  int32_t reserved;
  float   center[3];          /* sum|x| / n of the filtered points (Sasum, :137-139) */
  float   limit;              /* 1.5 * sqrt(sum d^2 / n)  (:177-179)           */
  float   position[3];        /* sum|x| / n of the inliers, or NaN             */
  float   squared_error;
} vk_detect_state;

/* bytes of device scratch for a cloud of `count` points (ref: detector.h:66-68
 * points_ / distances_, which this replaces) */
VK_API size_t vk_detect_workspace_bytes(int32_t count);

/* ref: src/detector.cu:152-188 Detector::Filter — interval/radius filter, then
 * removal of points further than 1.5 sigma from the centroid. `inliers`: device
 * float[3*count], receives the survivors in input order (the reference compacts
 * per thread block through an atomic, so its order varies run to run; the set is
 * the same). Fills every field of *state_dev except position / detected. */
VK_API int vk_detect_filter(const vk_detector* detector, const float* points,
    int32_t count, float* inliers, vk_detect_state* state_dev, void* workspace,
    void* stream);

/* ref: src/detector.cu:120-147 Detector::Detect = Filter + BoxDetected +
 * GetValidPosition / GetInvalidPosition; the answer is state_dev->position. */
VK_API int vk_detect(const vk_detector* detector, const float* points,
    int32_t count, float* inliers, vk_detect_state* state_dev, void* workspace,
    void* stream);

/* --------------------------------------------------------------- compaction -- */

/* ref: include/vulcan/util.cuh:52-140 PrefixSum<N> (both overloads) — the stream-compaction
 * primitive behind visible-block, patch and detector compaction. For every element i with
 * counts[i] > 0, offsets[i] is the first of counts[i] private, consecutive slots of a packed
 * output; elements with counts[i] == 0 get -1 (util.cuh:93-94); *total_dev is increased by
 * the sum (it is the reference's `total` argument: zero it to start a new output). The
 * reference leaves the order of the workgroups' ranges to its atomicAdd; here the ranges
 * are in input order (stable). counts, offsets: device int32[count]; workspace: device,
 * vk_compact_workspace_bytes(count). Three launches, no readback. */
VK_API size_t vk_compact_workspace_bytes(int32_t count);
VK_API int vk_compact_offsets(const int32_t* counts, int32_t count, int32_t* offsets,
    int32_t* total_dev, void* workspace, void* stream);

/* --------------------------------------------------------------- extraction -- */

/* ref: include/vulcan/extractor.h:10-136, src/extractor.cu — Extractor::Extract(DeviceMesh&):
 * the triangle mesh of the zero level set. Upstream stops after the vertices
 * (ExtractVertexIndicesKernel / ExtractFacesKernel are empty, extractor.cu:392-430), works one
 * block per launch with blocking copies in between, drops the far faces of every block and
 * has no test; this entry point is the finished extractor (conventions kept, gaps filled:
 * vulcan_amd/csrc/vk_extract.hip). Blocks come from the visible list as upstream
 * (extractor.cu:455-457) or, with `all_allocated`, from the whole table. `interpolate` = 0
 * places a vertex at its edge's midpoint as upstream's active code does (:361), 1 where
 * the linearly interpolated distance is 0.
 * points: device float[3 * point_capacity]; faces: device int32[3 * face_capacity];
 * counts_dev: device int32[4] = {points, faces, cubes skipped because a vertex they need
 * belongs to a block that is not listed, blocks listed}. Counts are the full totals even when
 * a capacity is too small (nothing is written past a capacity). Order: blocks in list order,
 * cubes by z*64 + y*8 + x, vertices by axis, faces in table order — reproducible.
 * workspace: device, vk_extract_workspace_bytes(main, excess) bytes. Four launches, no readback. */
VK_API size_t vk_extract_workspace_bytes(int32_t main_block_count, int32_t excess_block_count);
VK_API int vk_extract_mesh(const vk_volume* v, int all_allocated, int interpolate, float* points,
    int32_t point_capacity, int32_t* faces, int32_t face_capacity, int32_t* counts_dev,
    void* workspace, void* stream);

#ifdef __cplusplus
}  /* extern "C" */
#endif

#endif  /* VK_H_ */
