/*
 * hvrt.h -- C ABI of libhvrt: MI355X (gfx950) voxel-grid build + primary-ray grid traversal.
 *
 * This is the drop-in boundary for ONE hot path of Archibajl/RayTracer.  The reference has no FFI
 * of its own (it is a single C++ program), so each entry point cites the reference C++ interface
 * it stands in for; paths are relative to the reference's RayTracer/ directory.  The C++ class
 * HipVoxelRayTracer (raytracer_amd/host/HipVoxelRayTracer.hpp) implements the reference's
 * IRayTracer (IRayTracer.h:17-44) on top of exactly these calls; INTEGRATION.md shows the
 * reference-side patch.
 *
 * Conventions: every function returns HVRT_OK (0) or a negative hvrt_status and never throws;
 * hvrt_last_error() gives the message of the last failure on that context (or, with ctx == NULL,
 * of the last failed hvrt_create on this thread).  The caller owns all host buffers; the context
 * owns device memory, its HIP stream and its events.  Signatures use plain pointers and sizes
 * only -- no HIP, torch or C++ types -- so host code compiles without HIP headers.  "Device
 * pointer" arguments are raw addresses in the context's GPU (void*).  A context is bound to one
 * GPU and is not thread-safe; one process per GPU is the intended multi-GPU model.
 *
 * All results are defined by the reference's arithmetic (IEEE fp32, no FMA contraction, NaN/inf
 * control flow included): the voxel CSR is bit-exact, hit ids / t / colours are bit-exact with the
 * CPU restatement in oracle/ (tests/).
 */
#ifndef HVRT_H
#define HVRT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HVRT_MISS 0xFFFFFFFFu /* hit_id value for "no triangle" */

typedef enum hvrt_status {
    HVRT_OK = 0,
    HVRT_ERR_INVALID = -1,   /* bad argument / call order (message says which) */
    HVRT_ERR_NO_DEVICE = -2, /* no usable HIP device */
    HVRT_ERR_HIP = -3,       /* a HIP runtime call failed (message carries hipGetErrorString) */
    HVRT_ERR_OOM = -4,       /* device allocation failed */
    HVRT_ERR_OVERFLOW = -5   /* more than 2^32-1 triangle references (CSR offsets are u32 as in Voxels.h:19-20) */
} hvrt_status;

typedef struct hvrt_ctx hvrt_ctx;

/* Loop-invariant head of OctreeLikeRayTracer::generateRay (OctreeLikeRayTracer.cpp:342-359),
 * computed once on the host (tan() is libm-dependent; everything per pixel is + - * / sqrt). */
typedef struct hvrt_camera_frame {
    float origin[3];
    float lower_left[3];
    float horizontal[3];
    float vertical[3];
} hvrt_camera_frame;

/* Same fields and layout as cg_datastructures::RayHit (RayTracerCommon.h:17-29), 52 bytes;
 * bool hit widened to 32 bits (the reference struct has 3 bytes of padding there). */
typedef struct hvrt_hit {
    uint32_t hit;
    float t;          /* FLT_MAX on a miss */
    float point[3];   /* = triangle.v0, as the reference sets it (OctreeLikeRayTracer.cpp:286) */
    float origin[3];
    float normal[3];  /* raw file normal of the hit triangle */
    uint32_t triangle_index;
    uint32_t voxel_index; /* always 0, as in the reference */
} hvrt_hit;

typedef struct hvrt_grid_info {
    int32_t nx, ny, nz;
    float min_bound[3];
    float max_bound[3];
    float voxel_size[3];
    uint64_t n_voxels;
    uint64_t n_refs;       /* length of the CSR index array */
    uint64_t n_occupied;   /* voxels with count > 0 (VoxelizationResult::occupied_count) */
    uint32_t max_list;     /* longest per-voxel list */
    uint32_t n_triangles;
} hvrt_grid_info;

/* Per-call device timings (HIP events on the context's stream), milliseconds.
 * build_total_ms = count + scan + fill + sort + expand. */
typedef struct hvrt_timings {
    float build_total_ms;   /* hvrt_build_grid, everything */
    float build_count_ms;
    float build_scan_ms;
    float build_fill_ms;
    float build_sort_ms;
    float build_expand_ms;  /* libhvrt's own acceleration data: distance field, face bits, per-reference records */
    float render_ms;        /* last hvrt_render* / hvrt_trace_device: the traversal kernels (the tile order of the next
                             * launch of the same shape is built behind them and is not in this figure) */
    float trace_ms;         /* last hvrt_trace */
} hvrt_timings;

/* ---- context ------------------------------------------------------------------------------- */

/* Replaces: constructing a tracer (OctreeRayTracer.h:30).  device_id = HIP device ordinal. */
int hvrt_create(hvrt_ctx** out, int device_id);
void hvrt_destroy(hvrt_ctx* ctx);
const char* hvrt_last_error(const hvrt_ctx* ctx);
/* The context's HIP stream as an opaque handle (hipStream_t), for callers that interleave their
 * own work (e.g. RCCL collectives) with hvrt calls. */
void* hvrt_stream(hvrt_ctx* ctx);
/* Issue all further work on the caller's HIP stream (e.g. the stream its RCCL collectives run on), so
 * that renders and collectives order without host synchronisation.  NULL restores the context's own
 * stream.  The stream must belong to the context's device and outlive its use here. */
int hvrt_set_stream(hvrt_ctx* ctx, void* hip_stream);
int hvrt_synchronize(hvrt_ctx* ctx);
/* Traversal mode of all later hvrt_render* / hvrt_trace* calls.
 *   0 (default)  the reference's traversal, quirks included (OctreeLikeRayTracer.cpp:155-336): start in the clamped
 *                voxel of the origin, first voxel with any hit, last list entry that hits.
 *   1            nearest-hit mode: enter the grid through the reference's rayBoxIntersection (GeometryUtils.cpp:15-37,
 *                which its own traversal never calls), keep the nearest hit over the visited voxels, stop once it lies
 *                before the current voxel's exit.  An extension for callers who want a right
 *                image rather than the reference's; nothing in the reference pins it -- it is checked against the
 *                CPU restatement and against brute force over all triangles.  Shading, outputs and counters keep
 *                their meaning (a ray that misses the grid visits no voxel). */
int hvrt_set_mode(hvrt_ctx* ctx, int mode);
int hvrt_get_mode(const hvrt_ctx* ctx);
/* Forgets the context's mesh and grid and returns every per-scene setting (mode, overlap test, deferred data) to its
 * default, keeping the device buffers, streams and warmed-up queues: what a pool of contexts calls between two tracers.
 * The reference constructs a tracer per image (TraceImages.cpp:215-229); creating a context per image would pay the
 * runtime's start-up each time. */
int hvrt_clear_scene(hvrt_ctx* ctx);
/* Which voxels of a triangle's bounding-box range list it, for all later hvrt_build_grid calls.
 *   0 (default)  all of them: the reference's voxelizer (SceneCreator.cpp:197-281), the only rule under which the CSR is
 *                the reference's grid bit for bit.
 *   1            those the triangle overlaps, by the separating-axis test the reference declares and never defines
 *                (GeometryUtils.h:19, triangleIntersectsBox): the voxel grown by 2^-10 of its half size against the
 *                triangle -- three box normals, the triangle's normal, nine edge x axis products.  An extension: every
 *                list is a subset of the list rule 0 builds (fewer tests per ray); nothing in the reference pins it, it
 *                is checked against the CPU restatement (bit-identical CSR) and, in nearest-hit mode, against the frames
 *                of a rule-0 grid.  In the reference's own traversal mode (first voxel with any hit) frames may differ
 *                from the reference's: there a hit can be found in a voxel the triangle only shares a bounding box with.
 *                Geometry outside the grid (the reference clamps its voxel indices into the border voxels) is listed
 *                nowhere under this rule. */
int hvrt_set_overlap_test(hvrt_ctx* ctx, int mode);
int hvrt_get_timings(const hvrt_ctx* ctx, hvrt_timings* out);
/* Width in bits (16 or 32) of the per-voxel counters the NEXT hvrt_build_grid counts into first.  A context starts
 * with 16 (two counters per word: half the bytes to clear and scan); a build in which some voxel's list outgrows 16 bits
 * notices (its two reference totals differ), repeats its front with 32-bit counters and leaves the context at 32 for
 * good.  HVRT_BUILD_NARROW=0 and the separating-axis rule start at 32.  The CSR is the same either way.  No reference
 * counterpart (its lists are std::vectors, SceneCreator.cpp:205). */
int hvrt_get_counter_width(const hvrt_ctx* ctx);
/* Durations of the last asynchronous traversals (hvrt_render_part_device, hvrt_render_batch_part_device, the noalloc
 * form), for callers that queue several launches and wait once: waits for the stream, then writes up to max_entries
 * durations in milliseconds (HIP events around each traversal, like hvrt_timings.render_ms), the oldest of them first --
 * at most the last 32 launches are kept.  Returns the number written, or -1 (hvrt_last_error says why).
 * No reference counterpart (its render() is synchronous and logs one elapsed time, OctreeLikeRayTracer.cpp:42-118). */
int hvrt_render_history(hvrt_ctx* ctx, float* ms, int max_entries);
/* The tile cover of the last launch without counters (hvrt_cover.hip): before such a launch a pass over the triangles
 * marks the 8x8-pixel tiles of each frame in which some sample's ray can pass the reference's ray / triangle test
 * (GeometryUtils.cpp:57-97) for some triangle; the pixels of the other tiles receive the reference's result for a miss
 * without a walk.  dims <- {tiles per row, rows of tiles, words per row, triangles whose rectangle of tiles was larger than
 * 4 x 4 (the pass's second kernel handles those)}; bits (nullable; rows x words, bit tx & 31 of
 * word [ty * words + tx / 32], FRAME rows) <- the map of `frame` of that launch.  HVRT_TILE_COVER=0 turns the pass off
 * (then, and after a launch with counters, HVRT_ERR_INVALID).  Waits for the stream.  No reference counterpart
 * (OctreeLikeRayTracer.cpp:88-126 walks every ray). */
int hvrt_get_tile_cover(hvrt_ctx* ctx, int frame, uint32_t* bits, uint64_t capacity_words, uint32_t dims[4]);
/* The durations the last asynchronous traversal recorded for its 8x8-pixel tiles, in units of 16 shader clocks, tile
 * t = (frame * rows of tiles + row) * tiles per row + column of the rendered PART (what the next launch of the same shape
 * orders its work queue by; top bit: the tile is rendered by the 16-wave kernel, the word then holds the sum over its
 * quads).  A diagnostic: waits for the stream.  Returns the number of tiles, or a negative status.  No reference
 * counterpart. */
int hvrt_get_tile_costs(hvrt_ctx* ctx, uint32_t* cost, uint64_t capacity);

/* Callers that only want the voxel grid itself (the CSR that replaces VoxelGrid::voxels, SceneCreator.cpp:197-281)
 * can keep libhvrt's own derived acceleration data -- empty-space distance field, face bits, per-reference records,
 * none of which the reference has -- out of hvrt_build_grid / hvrt_set_grid_csr*: with on != 0 those calls return once
 * the CSR stands, and the derived data is built by the first render / trace / hvrt_get_distance_field, or by
 * hvrt_finish_accel (hvrt_timings.build_expand_ms is then that call's time).  Default: off (built with the grid).
 * No counterpart in the reference (its grid has no derived data). */
int hvrt_set_deferred_accel(hvrt_ctx* ctx, int on);
int hvrt_finish_accel(hvrt_ctx* ctx);

/* ---- mesh ---------------------------------------------------------------------------------- */

/* Replaces: buildTriangleList's output `std::vector<Triangle>` (SceneCreator.cpp:129-170,
 * Voxels.h:10-13).  tri_aos = n_tris x 12 floats (v0, v1, v2, normal), host memory. */
int hvrt_set_mesh(hvrt_ctx* ctx, const float* tri_aos, uint32_t n_tris);
/* Same, from a device pointer in the same layout (used after an RCCL broadcast). */
int hvrt_set_mesh_device(hvrt_ctx* ctx, const void* d_tri_aos, uint32_t n_tris);

/* Replaces: computeMeshBounds (SceneCreator.cpp:49-84) on the GPU: fp32 min/max over n_points
 * xyz triples (host memory) in the reference's sequential semantics (first occurrence wins ties,
 * NaNs ignored).  The reference feeds it the de-duplicated STL vertex list. */
int hvrt_compute_bounds(hvrt_ctx* ctx, const float* points_xyz, uint64_t n_points,
                        float min_out[3], float max_out[3]);
/* Same over the 3*n_tris corners of the resident mesh (device-resident input). */
int hvrt_mesh_bounds(hvrt_ctx* ctx, float min_out[3], float max_out[3]);

/* Replaces: computeVoxelizationParams (SceneCreator.cpp:91-122): the uniform voxel size
 * max(extent) / (float)max(nx,ny,nz).  Pure host arithmetic. */
float hvrt_voxel_size(const float min_bound[3], const float max_bound[3], int nx, int ny, int nz);

/* ---- voxel grid ---------------------------------------------------------------------------- */

/* Replaces: voxelizeTrianglesDirect + createVoxelGrid (SceneCreator.cpp:197-302), and the dead
 * GPU/inline variants (SceneCreator.cu:67-358, InlineVoxelizer.h:17-167).  Builds the CSR
 * voxel -> triangle-index structure of the resident mesh on the GPU: per-triangle AABB voxel
 * range (SceneCreator.cpp:23-37,177-188), counts, exclusive scan in flat order
 * ix + nx*(iy + ny*iz), scatter, per-voxel ascending order.  voxel_size is per axis (the reference
 * builder passes the same value three times). */
int hvrt_build_grid(hvrt_ctx* ctx, int nx, int ny, int nz,
                    const float min_bound[3], const float max_bound[3], const float voxel_size[3]);

/* Replaces: handing a CPU-built VoxelGrid to a tracer (OctreeRayTracer.h:30, VoxelGrid.h:8-42).
 * offsets = n_voxels+1 exclusive-scan entries in flat order, indices = offsets[n_voxels] entries;
 * host memory. */
int hvrt_set_grid_csr(hvrt_ctx* ctx, int nx, int ny, int nz,
                      const float min_bound[3], const float max_bound[3], const float voxel_size[3],
                      const uint32_t* offsets, const uint32_t* indices);
/* Same from device pointers (used after an RCCL broadcast of a grid built on another GPU). */
int hvrt_set_grid_csr_device(hvrt_ctx* ctx, int nx, int ny, int nz,
                             const float min_bound[3], const float max_bound[3], const float voxel_size[3],
                             const void* d_offsets, const void* d_indices, uint64_t n_refs);

int hvrt_get_grid_info(const hvrt_ctx* ctx, hvrt_grid_info* out);
/* Replaces: reading VoxelGrid::voxels[x][y][z].{triangle_start_idx,triangle_count} and
 * VoxelGrid::triangle_indices.  Copies the CSR to host memory; either pointer may be NULL. */
int hvrt_get_csr(hvrt_ctx* ctx, uint32_t* offsets, uint32_t* indices);
/* Copies the empty-space distance field to host memory: one byte per voxel in the CSR's voxel order,
 * min(255, city-block distance in voxels to the nearest occupied voxel), 0 for an occupied voxel.  No
 * reference counterpart: the reference steps through every empty voxel (OctreeLikeRayTracer.cpp:316-334);
 * this array only lets the traversal skip looking at voxels that cannot be occupied.  For tests and tools. */
int hvrt_get_distance_field(hvrt_ctx* ctx, uint8_t* dist);
/* Device addresses of the resident arrays (for RCCL broadcast by the caller). */
int hvrt_get_device_arrays(hvrt_ctx* ctx, void** d_tri_aos, void** d_offsets, void** d_indices);

/* Copies the resident scene into caller-provided device buffers (each nullable): n_tris*48 bytes,
 * (n_voxels+1)*4 bytes, n_refs*4 bytes.  This is the send side of the multi-GPU scene replication
 * (the caller broadcasts the buffers with RCCL; receivers call hvrt_set_mesh_device and
 * hvrt_set_grid_csr_device).  No reference counterpart: the reference is single-process.
 * Asynchronous on the context's stream. */
int hvrt_export_scene_device(hvrt_ctx* ctx, void* d_tri_aos, void* d_offsets, void* d_indices);

/* ---- rendering ----------------------------------------------------------------------------- */

/* Replaces: the loop-invariant part of generateRay (OctreeLikeRayTracer.cpp:342-359).  Host. */
void hvrt_make_camera_frame(const float position[3], const float look_at[3], const float up[3],
                            float fov_degrees, float aspect, hvrt_camera_frame* out);

/* Replaces: OctreeLikeRayTracer::render (OctreeLikeRayTracer.cpp:88-118).  Traces one primary ray
 * per sample through pixel corners (u = x/W, v = y/H; row 0 = bottom of the view), shades
 * (OctreeLikeRayTracer.cpp:27-40, Shaders.cpp:4-22) and returns host arrays indexed y*W + x:
 *   rgb      W*H*3 floats                      (the reference's return value)
 *   hit_id   W*H u32, HVRT_MISS on a miss      (nullable)
 *   t        W*H floats, FLT_MAX on a miss     (nullable)
 *   counters {voxels tested, ray/triangle tests} summed over all rays, counted once per ray
 *            (the reference traces every ray twice, so it logs twice these) (nullable)
 * spp == 1 reproduces the reference; spp > 1 adds counter-hash jitter in [0,1)^2 and averages
 * (an extension; hit_id / t are those of sample 0). */
int hvrt_render(hvrt_ctx* ctx, const hvrt_camera_frame* frame, int width, int height,
                int spp, uint32_t seed,
                float* rgb, uint32_t* hit_id, float* t, uint64_t counters[2]);

/* Replaces: the hit counter of OctreeLikeRayTracer::render (OctreeLikeRayTracer.cpp:106-111, logged by :57-65 as
 * "Ray hits"): the number of pixels of the last hvrt_render / hvrt_render_views_rgb8 call whose (sample-0) ray hit a
 * triangle, counted on the device -- so a caller that only wants the frame and the statistics need not fetch the
 * hit-id image to count them. */
int hvrt_last_hit_count(const hvrt_ctx* ctx, uint64_t* out);

/* Device-resident form, and the multi-GPU partition: the image is cut into row tiles of tile_rows
 * rows; tile k belongs to part (k % n_parts).  This call renders the tiles of `part` into compact
 * device buffers holding hvrt_part_rows() rows of `width` pixels, in tile order.  n_parts == 1
 * renders the whole frame.  Outputs are device pointers (each nullable): d_rgb 3 floats / px,
 * d_hit_id u32 / px, d_t float / px, d_counters 2 x u64 (accumulated into: zero it first).
 * Asynchronous on the context's stream. */
int hvrt_render_part_device(hvrt_ctx* ctx, const hvrt_camera_frame* frame, int width, int height,
                            int spp, uint32_t seed, int tile_rows, int part, int n_parts,
                            void* d_rgb, void* d_hit_id, void* d_t, void* d_counters);

/* hvrt_render_part_device for callers that cannot allocate -- code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  It has no counterpart in the reference (which allocates a std::vector per
 * render, OctreeLikeRayTracer.cpp:90).  Same arguments, same results bit for bit; what hvrt_render_part_device keeps
 * in context-owned, grow-on-demand device buffers (work-queue head, per-tile durations, the camera frame) lives here
 * in a fixed array inside the library's device image: 4 MiB + 256 bytes per GPU, held until the process ends.
 * Calls no hipMalloc / hipFree (nor anything else that allocates).  Limits and caveats:
 *   - at most 2^20 tiles of 8x8 pixels in the rendered part (a 7680x4320 frame has 518 400); larger -> HVRT_ERR_INVALID;
 *   - no tile history is kept or used (every call schedules like a first launch), so it is a few per cent slower
 *     than repeated hvrt_render_part_device calls of one shape, and over-long tiles are not split;
 *   - every call on a GPU, from any context, shares that GPU's array: calls that overlap on one GPU race;
 *   - each call records a pair of timing events around its kernel (hvrt_timings.render_ms, hvrt_render_history).  Under
 *     stream capture those events are captured, not recorded: render_ms then reads -1 and the history entry of such a
 *     call means nothing -- time a captured graph from outside.
 * Asynchronous on the context's stream. */
int hvrt_render_part_device_noalloc(hvrt_ctx* ctx, const hvrt_camera_frame* frame, int width, int height,
                                    int spp, uint32_t seed, int tile_rows, int part, int n_parts,
                                    void* d_rgb, void* d_hit_id, void* d_t, void* d_counters);

/* Batch form of hvrt_render_part_device: n_frames camera frames (1..HVRT_MAX_BATCH_FRAMES, host array), ONE
 * kernel launch whose work queue spans the tiles of every frame.  Frame f's rows of this part go to
 * d_rgb + f * slab_rows * width * 12 bytes (d_hit_id / d_t: * 4 bytes); slab_rows >= hvrt_part_rows().
 * This is what a rank of the multi-GPU batch mode runs per step: a part of one frame alone is too small a
 * launch (its duration is set by its longest tiles, not by throughput).  No reference counterpart (the
 * reference renders one frame at a time on one thread, OctreeLikeRayTracer.cpp:71-118).
 *
 * Scheduling note for both forms: each launch records the duration of every 8x8 tile, and the next launch
 * of the same shape (image size, partition, n_frames, spp) on the same scene pulls its tiles longest-first,
 * provided its (first) camera is within 2.5 degrees of that launch's -- a turn of the central ray, or a move of the eye
 * seen from the grid's centre; HVRT_HISTORY_MAX_DEG changes the limit, 0 removes it.
 * A launch without such history takes the rows of tiles from the middle of the frame outwards.  Only the order
 * of work changes, never a result; HVRT_TILE_ORDER=0 in the environment disables the use of history
 * (HVRT_TILE_SPLIT=0 only the hand-over of over-long tiles to the 16-waves-per-tile kernel). */
#define HVRT_MAX_BATCH_FRAMES 32
int hvrt_render_batch_part_device(hvrt_ctx* ctx, const hvrt_camera_frame* frames, int n_frames,
                                  int width, int height, int spp, uint32_t seed,
                                  int tile_rows, int part, int n_parts, int slab_rows,
                                  void* d_rgb, void* d_hit_id, void* d_t, void* d_counters);

/* Replaces: the per-channel conversion loop of saveToPPM / saveToJPG / saveToPNG (ImageSaver.h:28-33, 48-52,
 * 67-71): byte = (uint8_t)clamp(c * 255.0f, 0.0f, 255.0f), truncating; done on the device so that 3 bytes per
 * pixel leave the GPU instead of 12.  n_values = pixels * 3.  d_rgb 16-byte aligned, d_rgb8 4-byte aligned.
 * Asynchronous on the context's stream. */
int hvrt_quantise_rgb8_device(hvrt_ctx* ctx, const void* d_rgb, uint64_t n_values, void* d_rgb8);

/* Replaces: the three render + save passes of TraceImageMultiView (TraceImages.cpp:77-119) up to the image
 * encoder: n_frames views of the resident scene in ONE launch, quantised on the device, returned as 8-bit
 * RGB (frame-major, rows bottom-up as the reference's framebuffer).  hit_id (n_frames x H x W, nullable) and
 * counters (nullable) as for hvrt_render.  Synchronous. */
int hvrt_render_views_rgb8(hvrt_ctx* ctx, const hvrt_camera_frame* frames, int n_frames, int width, int height,
                           uint8_t* rgb8, uint32_t* hit_id, uint64_t counters[2]);
/* Rows owned by `part` (host arithmetic). */
int hvrt_part_rows(int height, int tile_rows, int part, int n_parts);
/* Reassembles a frame from gathered parts: d_parts holds n_parts slabs of slab_rows x width
 * pixels of bytes_per_pixel bytes each (slab_rows >= hvrt_part_rows of every part); writes
 * height x width pixels to d_frame.  A row (width x bytes_per_pixel bytes) must be a whole number of
 * 4-byte words: 12 (float rgb) and 4 (hit id, t) always are, 3 (the 8-bit rgb of hvrt_quantise_rgb8_device,
 * which is what the multi-GPU exchange moves) when width is a multiple of 4.  Asynchronous on the context's stream. */
int hvrt_assemble_parts_device(hvrt_ctx* ctx, const void* d_parts, int width, int height,
                               int tile_rows, int n_parts, int slab_rows, int bytes_per_pixel,
                               void* d_frame);

/* Replaces: OctreeLikeRayTracer::traceRay (OctreeLikeRayTracer.cpp:124-126), batched.
 * origins / dirs = n x 3 floats, out = n records; host memory. */
int hvrt_trace(hvrt_ctx* ctx, const float* origins, const float* dirs, uint32_t n,
               hvrt_hit* out, uint64_t counters[2]);

/* Device-resident form of hvrt_trace (rays produced on the GPU, e.g. by a caller's own camera or secondary-ray
 * kernel): d_origins / d_dirs n x 3 floats, d_hits n x hvrt_hit, d_counters 2 x u64 accumulated into (nullable).
 * Asynchronous on the context's stream; its kernel time is reported as render_ms after hvrt_synchronize. */
int hvrt_trace_device(hvrt_ctx* ctx, const void* d_origins, const void* d_dirs, uint32_t n,
                      void* d_hits, void* d_counters);

/* Device -> the caller's (pageable) host memory through the context's ring of pinned chunks (what hvrt_render uses for
 * its frames: a plain DMA into pageable memory runs at a third of the link).  Synchronous.  No reference counterpart. */
int hvrt_download(hvrt_ctx* ctx, void* dst, const void* d_src, uint64_t bytes);
/* The same with a scattered destination: the source is one compact run of blocks of block_bytes (the last may be
 * shorter), block j lands at dst + j * dst_stride (dst_stride >= block_bytes).  How a member of a device group puts
 * its row tiles where they belong in the caller's frame.  Synchronous.  No reference counterpart. */
int hvrt_download_blocks(hvrt_ctx* ctx, void* dst, const void* d_src, uint64_t bytes, uint64_t block_bytes, uint64_t dst_stride);

/* ---- several GPUs of one node ---------------------------------------------------------------
 * No counterpart in the reference (one thread on one CPU); the seam a maintainer would use is the tracer factory,
 * TraceImages.cpp:464-484.  One process, one context per device, one host thread per member: the scene is replicated, a
 * frame's 16-row tiles are interleaved over the members (tile k -> member k % n).  A frame wanted in HOST memory
 * (hvrt_group_render -- what IRayTracer::render returns, IRayTracer.h:28) is copied out by the members themselves, each
 * its own tiles over its own PCIe link straight into the caller's arrays; nothing crosses xGMI.  A frame wanted in
 * DEVICE memory (hvrt_group_render_device) is gathered on member 0 -- RCCL (ncclCommInitAll, grouped ncclSend / ncclRecv
 * of the slabs; librccl is opened at run time) or, with HVRT_GROUP_TRANSPORT=copy or when a device is listed twice,
 * device-to-device copies -- and reassembled there.  The scene travels the same way (ncclBroadcast, or copies).
 * Results are bit for bit those of one context.  A group of one device needs no transport at all. */
typedef struct hvrt_group hvrt_group;
typedef struct hvrt_group_timings {
    float create_ms;      /* hvrt_group_create: every member's hvrt_create (concurrent) + the transport's set-up; host clock */
    float replicate_ms;   /* last hvrt_group_replicate_scene; host clock */
    float render_ms;      /* last render: the longest member's traversal (HIP events, as hvrt_timings.render_ms) */
    float download_ms;    /* last hvrt_group_render: the longest member's copy into the caller's arrays; host clock */
    float total_ms;       /* last render, call to return; host clock */
} hvrt_group_timings;
int hvrt_device_count(void);                                  /* visible HIP devices (0 when there is none) */
int hvrt_group_create(hvrt_group** out, const int* device_ids, int n_devices);
void hvrt_group_destroy(hvrt_group* group);
const char* hvrt_group_last_error(const hvrt_group* group);   /* group == NULL: why hvrt_group_create failed (this thread) */
int hvrt_group_size(const hvrt_group* group);
const char* hvrt_group_transport(const hvrt_group* group);    /* "rccl", "copy" or "none" (one member) */
int hvrt_group_get_timings(const hvrt_group* group, hvrt_group_timings* out);
/* Member i's context (owned by the group).  Member 0 is where the caller sets the mesh and builds or ingests the grid. */
hvrt_ctx* hvrt_group_member(hvrt_group* group, int i);
/* Member 0's mesh and CSR to every other member (each derives its own acceleration data, all at once).  Synchronous. */
int hvrt_group_replicate_scene(hvrt_group* group);
/* hvrt_set_mode on every member.  (The renders below also hand member 0's mode to the others, so a caller that only
 * ever talks to member 0's context -- hvrt_set_mode(hvrt_group_member(g, 0), 1) -- gets one mode over the whole frame.) */
int hvrt_group_set_mode(hvrt_group* group, int mode);
/* hvrt_render over the group: rgb W*H*3 floats, hit_id / t W*H (nullable), host memory; counters / hits nullable
 * (summed over the members; hits = pixels whose sample-0 ray hit a triangle, as hvrt_last_hit_count).  Synchronous. */
int hvrt_group_render(hvrt_group* group, const hvrt_camera_frame* frame, int width, int height, int spp, uint32_t seed,
                      float* rgb, uint32_t* hit_id, float* t, uint64_t counters[2], uint64_t* hits);
/* The same frame left in member 0's device memory: *d_rgb <- W*H*3 floats owned by the group, valid until its next
 * render.  Synchronous.  hvrt_group_last_exchange_bytes: what the last render moved between devices (0 after
 * hvrt_group_render). */
int hvrt_group_render_device(hvrt_group* group, const hvrt_camera_frame* frame, int width, int height, int spp, uint32_t seed,
                             void** d_rgb, uint64_t counters[2], uint64_t* hits);
double hvrt_group_last_exchange_bytes(const hvrt_group* group);

#ifdef __cplusplus
}
#endif
#endif /* HVRT_H */
