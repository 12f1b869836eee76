// hvrt_internal.hpp -- shared between the libhvrt translation units (not installed).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hvrt.h"

namespace hvrt {

// Grid geometry as the kernels see it.  Flat voxel index f = ix + nx*(iy + ny*iz)
// (reference: SceneCreator.cpp:40-42).
struct GridParams {
    int nx, ny, nz;
    float mnx, mny, mnz;   // VoxelGrid::minBound
    float vsx, vsy, vsz;   // VoxelGrid::voxelSize (uniform when built by us)
    uint64_t nvox;
    float bxx, bxy, bxz;   // VoxelGrid::maxBound (only the nearest-hit mode's entry test reads it)
};

// HBM layout of the scene (all owned by the context):
//   tri_aos   T x 12 f32   v0,v1,v2,normal  -- the reference's Triangle (Voxels.h:10-13); input of the
//                                              voxelizer, and what is broadcast/returned
//   tri_rec   T x 12 f32   v0,e1,e2,normal  -- traversal record: e1 = v1-v0, e2 = v2-v0 precomputed
//                                              (same fp32 subtractions Moller-Trumbore would do per test)
//   offsets   nvox+1 u32   CSR exclusive scan; lives at cnt_base+3 so that cnt = offsets+1 is 16-B aligned
//   indices   n_refs u32   ascending per voxel
struct SceneDev {
    GridParams g;
    const float4* tri_rec;
    const uint32_t* offsets;
    const uint32_t* indices;
    uint32_t n_tris;
    // ref_rec: one 48-byte record PER REFERENCE, in CSR order: v0, e1, e2 of triangle indices[j], the
    // triangle id in the 10th dword, 2 pad dwords.  A voxel's list is then one contiguous run of records
    // whose addresses do not depend on a loaded index: the index -> record dependent load of the
    // traversal disappears and the next record can be prefetched while the current one is tested.
    // Null when the context runs in indexed mode (HVRT_INLINE_RECORDS=0 or the array would not fit).
    const float4* ref_rec;
    // face_bits: 6 bit arrays of face_words u32 each, one bit per CSR entry j: entry j's triangle is also
    // listed in the voxel's -x / +x / -y / +y / -z / +z neighbour (k_face_bits).  Always present.
    const uint32_t* face_bits;
    uint64_t face_words;
    // dist: one byte per voxel, min(255, city-block distance to the nearest occupied voxel), 0 = occupied
    // (k_dist_x / k_dist_sweep).  Lets the traversal step through empty space without looking at it.
    const uint8_t* dist;
};

struct RenderArgs {
    const hvrt_camera_frame* frames;   // n_frames camera frames, DEVICE memory (read with scalar loads); null: the one frame below
    hvrt_camera_frame frame0;          // a single frame travels as a kernel argument (no copy in front of the launch)
    int n_frames;
    int width, height;       // full frame
    int spp;
    float width_f, height_f, spp_f;   // the same as floats (exact), so that the kernel holds them in scalar registers
    uint32_t seed;
    int nearest;             // 0: the reference's traversal; 1: nearest-hit mode (hvrt_set_mode)
    int tile_rows, part, n_parts;
    int local_rows;          // rows owned by this part
    int slab_rows;           // rows between consecutive frames' outputs (>= local_rows)
    float* rgb;              // n_frames x slab_rows x width x 3 (nullable)
    uint32_t* hit_id;        // nullable
    float* t;                // nullable
    unsigned long long* counters;  // 2 x u64, nullable
    unsigned long long* hits;      // 1 x u64, nullable: += pixels whose sample-0 ray hit a triangle (the reference's "Ray hits" log line)
    // Queue scratch (the context's is 2048 bytes = 512 words, hvrt_create; the entry point that allocates nothing brings 16 bytes):
    //   [0]          the single queue head (n_heads == 1)                      zero before a launch
    //   [1]          number of tiles k_render gave up on (n_abandoned)         zero before a launch
    //   [2]          head of the sample-item queue (sample_rgb)                 zero before a launch
    //   [8 .. 15]    heads of the segment launches (hvrt_render's half frames: launch b uses word 8 + b as ITS [0])
    //   [32 .. 40]   seg_slots: slot ranges of those launches (written by k_order_segments)
    //   [49]         n_front of the order in use (written by the order build, never reset)
    //   [50], [51]   launch_compact_queue's counters (the queue's length, its front tiles)   zero before a launch
    //   [48]         split_started: workgroups of every k_render_split so far that have started -- compared with a running
    //                total on the host, NEVER reset (hvrt_create zeroes the whole block once)
    //   [64 + 32 h]  head h of the eight per-XCD heads (n_heads == 8), 128 bytes apart so that their atomics do not queue up
    //                behind one another in one cache line                          zero before a launch
    // Who zeroes: render_batch_part's memsets ([0, 16) and [64, 320)) when the queue is not known to be clean, otherwise
    // the last kernel of the order build behind the previous launch (launch_order_build's clear_heads).
    uint32_t* tile_counter;
    // n_heads == 8: the queue has eight heads (above), one per XCD: behind the first n_waves slots (a wave's first slot is its
    // own number) the queue is cut into chunks of 2^HVRT_QUEUE_CHUNK_LOG2 slots, chunk c belongs to head c & 7, and head h
    // hands out the slots of its chunks in order (see render_tiles).  1: the single head [0].
    uint32_t n_heads;
    // Work-queue order.  Slot q of the queue is tile order[q] (null: q itself); a tile is 8x8 pixels of one
    // frame, numbered frame-major.  Every launch leaves the duration of each tile (shader clocks / 16) in
    // tile_cost; the next launch of the same shape starts with the long tiles of this one (k_order_*), because
    // the kernel's length is set by when its few longest tiles START, not by throughput alone.
    const uint32_t* order;
    // Tiles that alone would outlast the rest of the launch are taken out of that queue and rendered by
    // k_render_split, 16 waves per tile (split_tiles[0, *n_split), both written by k_order_place; with order).
    const uint32_t* split_tiles;
    const uint32_t* n_split;
    const uint32_t* n_queue;      // (nullable) the queue's length, where only the device knows it (launch_compact_queue)
    const uint32_t* n_front;      // (nullable; with order) how many slots at the queue's head hold tiles put there for their length: pulled one at a time
    uint32_t* split_started;      // (nullable) k_render_split: every workgroup adds 1 when it starts (launch_wait_split_started)
    // k_render gives up on a tile whose rays are still inside long lists abort_ticks after they set out and
    // appends it to abandoned[*n_abandoned++] (zeroed before the launch); a k_render_split launch behind it
    // renders those tiles (split_tiles = abandoned, n_split = n_abandoned).
    uint32_t* abandoned;
    uint32_t* n_abandoned;
    long long abort_ticks;   // in ticks of the constant 100 MHz clock (wall_clock64), per sample of a pixel
    // A launch without history whose queue (n_queue: the marked tiles) holds at most few_tiles tiles gives every tile up at the
    // first occupied voxel its rays reach: the GPU is all but empty, and k_render_split behind it walks each tile with
    // sixteen waves that look ahead (trace_ray_ahead) instead of one that cannot.  0: never.
    uint32_t few_tiles;
    // (nullable) two words of pinned host memory: the give-up variant leaves there whether it took that course, then
    // few_serial -- a host that finds the serial of its last launch knows that the launch left no durations worth an order
    uint32_t* few_report;
    uint32_t few_serial;
    uint32_t* tile_cost;
    unsigned long long* cost_sum;   // sum of tile_cost over this launch (zeroed before it)
    unsigned long long* walked;     // (nullable) += the tiles this launch walked (zeroed before it; the order build reads it at cost_sum[2])
    // A frame rendered as several launches over consecutive ranges of ONE ordered queue (the host-facing hvrt_render:
    // the rows a launch completes travel over PCIe while the next launch traces): this launch takes the slots
    // [seg_slots[seg], seg_slots[seg + 1]) and its queue head starts at 0.  Null: the whole queue.
    const uint32_t* seg_slots;
    int seg;
    // Supersampled launches (spp > 1) render the tiles of split_tiles sample by sample inside k_render instead of handing them
    // to k_render_split (null: they do not): sample_rgb holds sample_cap x spp x 64 colours, sample_done one counter per
    // tile of the list (zero before the launch; the kernel leaves them zero), and word [2] of the queue scratch is the head
    // of the item queue (zero before the launch).
    float4* sample_rgb;
    uint32_t* sample_done;
    uint32_t sample_cap;
    // Tile cover (hvrt_cover.hip; null: every tile is walked): bit tx & 31 of cover[frame * cover_frame_words + (y / 8) *
    // cover_words_x + tx / 32] is clear when no sample of the 8x8 tile at column tile tx and FRAME row y can hit anything --
    // its pixels get the miss result without a walk.  Only the kernels without counters look at it.
    const uint32_t* cover;
    uint32_t cover_words_x, cover_frame_words;
    int debug_start;         // a diagnostic (HVRT_DEBUG_TILE_START=1 at the time of the call): tile_cost <- when each tile started (10-ns ticks, low 32 bits) instead of how long it took
};

// ---- launchers (hvrt_build.hip) ----
#ifndef HVRT_SCAN_CHUNK
#define HVRT_SCAN_CHUNK 32768
#endif
constexpr int SCAN_CHUNK = HVRT_SCAN_CHUNK;  // counters per scan workgroup (64 lanes x 32 per wave; 16 waves)

hipError_t launch_prepare_records(const float* tri_aos, float4* tri_rec, uint32_t n_tris, hipStream_t s);
// Build statistics block (one allocation, zeroed with one memset per build):
//   [0, 8)              stats: 0 total references (exact, 64-bit), 1 occupied voxels, 2 longest list, 3 long-list queue
//                       length, 4 bad indices, 5 "the scan gave up", 6 scan ticket, 7 the count pass's own total
//   [8, 8 + 1024)       vol_sums: 64 words a cache line apart (k_count with 16-bit counters)
//   [1032, 1032 + n)    scan status words, then n aux words (n = chunks of the scan)
constexpr uint64_t STATS_VOL_SUMS = 8, STATS_STATUS = 8 + 1024;
// narrow: cnt is the 16-bit counter array (two per word; zero on entry) and vol_sums collects the exact total
// sat_mask (n_tris words, or null): list a triangle only in the voxels of its range that it overlaps (separating-axis test;
// 32-bit counters only); the verdicts on ranges of up to 32 voxels are left there for launch_fill
// cnt_base[0..3] <- 0 (16-byte aligned) and stats[0..n_words) <- 0, one launch
hipError_t launch_build_reset(uint32_t* cnt_base, uint64_t* stats, uint64_t n_words, hipStream_t s);
hipError_t launch_count(const float* tri_aos, uint32_t n_tris, const GridParams& g, uint32_t* cnt, bool narrow,
                        uint64_t* vol_sums, uint32_t* ranges, uint32_t* sat_mask, hipStream_t s);
// exclusive scan of n_chunks x SCAN_CHUNK counters; status_aux: 2 x n_chunks u64, the first n_chunks zero on entry.
// cnt16 == nullptr: in place over cnt.  Otherwise the counters are read from cnt16 and the offsets written to cnt.
// stats[0] <- total (exact, 64-bit, whatever the 32-bit offsets did), stats[1] += occupied voxels, stats[2] <- longest
// list, stats[5] != 0: the scan gave up (stats[5], stats[6] must be zero on entry), stats[7] <- sum of vol_sums
// (narrow) or stats[0]
// occ_bits (nullable): n_chunks x SCAN_CHUNK / 32 words, one bit per voxel "has a list"
hipError_t launch_scan(uint32_t* cnt, const uint32_t* cnt16, const uint64_t* vol_sums, uint64_t n_chunks, uint64_t* status_aux,
                       uint64_t* stats, uint32_t* occ_bits, hipStream_t s);

// How the kernels behind the scan learn the number of CSR entries.  live == nullptr: the host knows it (n), the arrays
// hold at least that many.  live != nullptr: it is read on the device (live = the stats block: [0] the scan's total,
// [7] the count pass's own); the arrays hold cap entries, launches are sized for cap, and a build that needs more --
// or whose two totals differ -- writes nothing (the host notices at the end).
struct RefCount {
    const uint64_t* live;
    uint64_t n;
    uint64_t cap;
};
uint64_t face_words_for(uint64_t n_refs);    // words per bit plane of face_bits (SceneDev::face_words)
hipError_t launch_fill(const uint32_t* ranges, const float* tri_aos /* read for big ranges when sat_mask */, uint32_t n_tris, const GridParams& g, uint32_t* cnt,
                       uint32_t* indices_unsorted, uint32_t* vox_of, const RefCount& rc, const uint32_t* sat_mask, hipStream_t s);
hipError_t launch_sort(const uint32_t* offsets, const uint32_t* unsorted, const uint32_t* vox_of, const RefCount& rc,
                       uint32_t* indices, uint32_t* long_queue, uint64_t long_queue_capacity, uint64_t* stats,
                       hipStream_t s);
hipError_t launch_expand_records(const uint32_t* indices, const RefCount& rc, const float4* tri_rec, uint32_t n_tris,
                                 float4* ref_rec, uint64_t* stats, hipStream_t s);
hipError_t launch_face_bits(const uint32_t* indices, const RefCount& rc, const uint32_t* offsets, const GridParams& g,
                            uint32_t n_tris, const uint32_t* ranges /* null: search the neighbour lists */,
                            const uint32_t* vox_of /* null: search the offsets */,
                            uint32_t* face_bits /* 6 x face_words_for(entries) */, hipStream_t s);
// occ_bits: the bitmap launch_scan left for THESE offsets, or null (occupancy is then read from the offsets)
hipError_t launch_distance_field(const uint32_t* offsets, const uint32_t* occ_bits, const GridParams& g, uint8_t* dist, hipStream_t s);
// stats[1] += occupied voxels, stats[2] <- longest list, stats[6] += offsets that break the CSR rules (offsets[0] == 0,
// non-decreasing, offsets[nvox] == n_refs), stats[4] += indices >= n_tris (all zero on entry)
hipError_t launch_csr_stats(const uint32_t* offsets, uint64_t nvox, const uint32_t* indices, uint64_t n_refs,
                            uint32_t n_tris, uint64_t* stats, hipStream_t s);
hipError_t launch_bounds(const float* pts, uint64_t n_points, float* partial /* blocks x 6 + idx */,
                         uint64_t* partial_idx, int n_blocks, float* out6, hipStream_t s);
int bounds_blocks(uint64_t n_points);

constexpr int ORDER_BINS = 6;                    // long classes of the tile order (launch_order_build)
constexpr int ORDER_ARRAYS = ORDER_BINS + 2;     // count arrays: split, the long classes, the other walked tiles
// ---- launchers (hvrt_trace.hip) ----
// a full-size dispatch of a kernel with 160 bytes of scratch per lane (a queue's scratch set-up), nothing else
hipError_t launch_scratch_warmup(uint32_t* sink, int n_cus, hipStream_t s);
// abortable: the kernel may give up on over-long tiles (ra.abandoned); a launch_render_split over that list must follow
hipError_t launch_render(const SceneDev& sc, const RenderArgs& ra, int n_cus, bool abortable, hipStream_t s);
// zeroes the queue scratch words a launch starts from ([0, 16), the eight heads), *cost_sum and *walked (nullable)
hipError_t launch_queue_reset(uint32_t* tile_counter, unsigned long long* cost_sum, unsigned long long* walked, hipStream_t s);
uint64_t render_tile_count(int width, int local_rows, int n_frames);
// with ra.cover: the pixels of the tiles the cover does not mark (k_render passes those by); in front of launch_render
hipError_t launch_fill_background(const RenderArgs& ra, int n_cus, hipStream_t s);
// with ra.cover: queue <- the marked tiles of order_in (null: of the natural order) in nearly that order; counters[0] <- how
// many, counters[1] <- how many of them from order_in's first *n_front_in slots (both zero on entry); unmarked tiles get a
// duration of zero.  A launch_render over that queue (order = queue, n_queue = counters, n_front = counters + 1) needs no cover.
hipError_t launch_compact_queue(const RenderArgs& ra, const uint32_t* order_in, const uint32_t* n_front_in, uint32_t* block_marked,
                                uint32_t* queue, uint32_t* counters, hipStream_t s);
// (alone: nothing runs beside it -- the list of tiles a launch gave up on -- so it may fill the SIMDs: six of its waves each;
// rays: 4 or 8 rays per wave, i.e. 16 or 8 waves per tile)
hipError_t launch_render_split(const SceneDev& sc, const RenderArgs& ra, int n_cus, hipStream_t s, uint32_t* n_workgroups = nullptr,
                               uint32_t expected_tiles = 0, bool alone = false, int rays = 4);
// One wave that waits (at most timeout_us) until *started has reached `target` (wrap-around compare): queued in front of
// a k_render whose k_render_split runs beside it on another stream, so that the split kernel's workgroups are resident
// before k_render's persistent waves take every slot there is.
hipError_t launch_wait_split_started(uint32_t* started, uint32_t target, uint32_t timeout_us, hipStream_t s);
// split_tiles <- the tiles for k_render_split; order <- the others: long tiles (cost > 2.5 x mean) first, the
// rest behind them in natural order.  block_count: ORDER_ARRAYS x (ceil(n_tiles / 256) + 1) u32 of scratch; afterwards
// block_count[ceil(n_tiles / 256)] is the number of split tiles (0 when may_split is false).  *wants_split
// (device-visible host memory) <- the number of tiles that qualify for splitting, whatever may_split says.
hipError_t launch_order_build(uint32_t* tile_cost, uint32_t n_tiles, const unsigned long long* cost_sum, int n_cus,
                              bool may_split, uint32_t* block_count, uint32_t* order, uint32_t* split_tiles,
                              uint32_t* wants_split /* 2 words: the count, then `serial` */, uint32_t serial,
                              uint32_t* clear_heads /* 24 words, nullable */,
                              unsigned long long* clear_sum /* nullable */, bool by_sample /* the next launch renders split tiles sample by sample */,
                              hipStream_t s);
// seg_slots[0..n_seg] <- the slot ranges of n_seg launches over the queue launch_order_build left: launch b covers the
// tiles of tile rows [row_bounds[b], row_bounds[b + 1]) that are in k_render's queue behind the long ones; launch 0
// also takes the long tiles, which sit at the queue's head (row_bounds[0] = 0, row_bounds[n_seg] = tile rows; n_seg <= 8)
hipError_t launch_order_segments(const uint32_t* order, const uint32_t* block_count, uint32_t n_blocks, uint32_t n_tiles,
                                 uint32_t tiles_x, const uint32_t* row_bounds, int n_seg, uint32_t* seg_slots, hipStream_t s);
hipError_t launch_trace(const SceneDev& sc, const float* origins, const float* dirs, uint32_t n,
                        hvrt_hit* out, unsigned long long* counters, int nearest, hipStream_t s);
hipError_t launch_quantise_rgb8(const float* rgb /* 16-B aligned */, uint64_t n_values, uint8_t* out /* 4-B aligned */, hipStream_t s);
hipError_t launch_assemble(const void* parts, int width, int height, int tile_rows, int n_parts,
                           int slab_rows, int bytes_per_pixel, void* frame, hipStream_t s);

// ---- launchers (hvrt_cover.hip) ----
struct CoverArgs {
    const hvrt_camera_frame* frames;   // n_frames frames in DEVICE memory, or null: frame0
    hvrt_camera_frame frame0;
    int n_frames;
    int width, height;                 // the full frame
    uint32_t tiles_x, tiles_y;         // 8x8-pixel tiles of the full frame
    uint32_t words_x;                  // words per row of tiles, at every level
    int levels;                        // L: the root block is 2^L x 2^L tiles
    uint32_t frame_words;              // (L + 1) x tiles_y x words_x: one frame's pyramid; level 0 comes first and is the result
    uint32_t* bits;                    // n_frames x frame_words words, ZERO on entry
    uint32_t* slow_count;              // n_frames words, ZERO on entry: triangles queued for the general case, per frame
    uint32_t* slow_queue;              // n_frames x slow_cap triangle indices
    uint32_t slow_cap;
    void* consts;                      // n_frames x 64 bytes: what k_cover_frames works out of the camera frames
};
// fills in the geometry fields (everything but frames / frame0 / the pointers); nonzero: the frame is too large for a pyramid
int cover_geometry(int width, int height, int n_frames, uint32_t n_tris, CoverArgs& ca);
// one allocation of cover_words(ca) words holds everything: cover_bind points bits / slow_count / slow_queue into it; its
// first cover_zero_words(ca) words must be zero before launch_cover
uint64_t cover_words(const CoverArgs& ca);
uint64_t cover_zero_words(const CoverArgs& ca);
void cover_bind(CoverArgs& ca, uint32_t* words);
// marks, in level 0 of ca.bits, every tile of every frame in which some sample's ray may hit some triangle of sc
hipError_t launch_cover(const SceneDev& sc, const CoverArgs& ca, hipStream_t s);

}  // namespace hvrt
