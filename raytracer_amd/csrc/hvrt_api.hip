// hvrt_api.hip -- the C ABI of libhvrt (include/hvrt.h): context, device memory, kernel sequencing.
// Host code (plus the device-side scratch array of the entry point that allocates nothing and the one-wave
// kernel that fills its camera slot); the kernels live in hvrt_build.hip / hvrt_trace.hip.  No CPU fallback
// exists: every entry point either runs on the GPU or returns an error.
#include "hvrt_internal.hpp"
#include <unistd.h>

#include <float.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

using namespace hvrt;

struct hvrt_ctx {
    int device = 0;
    int n_cus = 256;
    hipStream_t stream = nullptr;       // the stream all work is issued on (own_stream unless hvrt_set_stream)
    hipStream_t own_stream = nullptr;
    hipStream_t aux_stream = nullptr;   // k_render_split runs here, beside k_render
    hipStream_t df_stream = nullptr;    // the distance field is built here, beside fill / order / records
    hipStream_t copy_stream = nullptr;  // hvrt_render: the rows one launch completed go to the host here while the next launch traces
    hipEvent_t ev_seg[8] = {};
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool dist_in_flight = false;
    hipEvent_t ev[10] = {};
    // ev[6] / ev[7] bracket the LAST traversal; every asynchronous render takes the next pair of this ring for them, so
    // that a caller who queues several launches and waits once can still read each one's duration (hvrt_render_history)
    static constexpr int RENDER_RING = 32;
    hipEvent_t render_ring[RENDER_RING][2] = {};
    uint64_t renders_recorded = 0;
    std::string error;

    // mesh
    float* tri_aos = nullptr;      // owned unless aos_external
    float4* tri_rec = nullptr;
    uint32_t n_tris = 0;
    uint64_t aos_capacity = 0;     // floats
    uint64_t rec_capacity = 0;     // float4s

    // grid
    bool has_grid = false;
    GridParams g = {};
    float max_bound[3] = {};
    uint32_t* cnt_base = nullptr;  // offsets = cnt_base + 3
    uint64_t cnt_capacity = 0;     // u32 elements
    uint32_t* indices = nullptr;
    uint64_t indices_capacity = 0;
    float4* ref_rec = nullptr;     // per-reference traversal records (n_refs x 48 B), see SceneDev
    uint64_t ref_rec_capacity = 0; // float4s
    bool inline_records = true;    // HVRT_INLINE_RECORDS=0 selects the indexed traversal
    bool ref_rec_valid = false;
    uint32_t* face_bits = nullptr; // 6 x face_words u32, see SceneDev
    uint64_t face_bits_capacity = 0;
    uint64_t face_words = 0;
    uint8_t* dist = nullptr;       // per-voxel empty-space distance, see SceneDev
    uint64_t dist_capacity = 0;
    uint64_t n_refs = 0, n_occupied = 0;
    uint32_t max_list = 0;

    // build scratch
    uint32_t* ranges = nullptr;   uint64_t ranges_capacity = 0;
    uint32_t* cnt16 = nullptr;    uint64_t cnt16_capacity = 0;      // 16-bit counters of the count pass, two per word
    bool wait_for_split = true;            // HVRT_SPLIT_WAIT=0: k_render does not wait for k_render_split's workgroups to be resident
    uint32_t order_serial = 0;             // number of the order build that made the order in use (it leaves it beside its split count, split_hint[1])
    uint32_t split_started_total = 0;      // workgroups of all k_render_split launches that report in (wraps; compared as a difference)
    bool narrow_counters = true;   // HVRT_BUILD_NARROW=0; cleared for good once a list of this context outgrew 16 bits
    uint32_t* occ_bits = nullptr; uint64_t occ_bits_capacity = 0;   // one bit per voxel, left by the scan for the distance field
    bool occ_bits_valid = false;
    uint64_t* stats = nullptr;     // the build statistics block (hvrt_internal.hpp: stats | vol_sums | scan status | aux)
    uint64_t stats_capacity = 0;
    uint32_t* long_queue = nullptr; uint64_t long_queue_capacity = 0;
    uint32_t* fill_scratch = nullptr; uint64_t fill_scratch_capacity = 0;  // unsorted ids | voxel of each entry
    const uint32_t* vox_of = nullptr;  // valid between fill and the end of hvrt_build_grid (or until deferred data is built)
    bool reuse_arrays = true;      // HVRT_BUILD_REUSE=0: every build waits for the reference count and sizes its arrays
    bool defer_accel = false;      // hvrt_set_deferred_accel
    bool accel_pending = false;    // the CSR is final, the derived data is not built yet
    bool accel_built_here = false; // whether the pending data belongs to a grid built here (ranges / vox_of usable)
    uint32_t* sat_mask = nullptr; uint64_t sat_mask_capacity = 0;    // SAT builds: per triangle, which voxels of a small range keep it
    int overlap_mode = 0;          // hvrt_set_overlap_test: 0 = the reference's bounding-box rule, 1 = separating axes (later builds)
    bool lists_follow_ranges = false;   // the grid in place lists a triangle in EVERY voxel of its range (built here, box rule):
                                        // "also listed in that neighbour" is then a compare with the range, not a search
    float* bounds_v = nullptr; uint64_t* bounds_i = nullptr; float* bounds_out = nullptr;

    // render scratch
    uint32_t* tile_counter = nullptr;
    // camera frames of a launch: pinned staging ring -> device ring (the kernel reads them with scalar loads)
    static constexpr int CAM_SLOTS = 4, CAM_MAX = HVRT_MAX_BATCH_FRAMES;
    hvrt_camera_frame* cam_host = nullptr;   // CAM_SLOTS x CAM_MAX, pinned
    hvrt_camera_frame* cam_dev = nullptr;    // CAM_SLOTS x CAM_MAX
    hipEvent_t cam_ev[CAM_SLOTS] = {};
    bool cam_ev_used[CAM_SLOTS] = {};
    int cam_next = 0;
    // longest-first tile order: per-tile durations of the previous launch of the same shape (RenderArgs)
    uint32_t* tile_cost = nullptr;  uint64_t tile_cost_capacity = 0;
    uint32_t* tile_order = nullptr; uint64_t tile_order_capacity = 0;
    uint32_t* order_blocks = nullptr; uint64_t order_blocks_capacity = 0;
    uint32_t* split_tiles = nullptr; uint64_t split_tiles_capacity = 0;
    uint32_t* abandoned = nullptr; uint64_t abandoned_capacity = 0;
    // 0.3 ms of the constant 100 MHz clock; HVRT_TILE_ABORT_US sets it, 0 = never give up on a tile.  (Before k_render_split's walk
    // looked ahead it was 0.5 ms: launches without history, the cover worked out by each, at 1000 / 600 / 500 / 420 / 350 / 250 us: config 2 1.09 / 0.95 / 0.86 / 0.80 / 0.86 / 0.90 ms,
    // config 3 1.86 / 1.81 / 1.70 / 1.99 / 2.58 / 3.78, config 4 16.3 / 14.6 / 13.9 / 15.7 / 15.9 / 18.2)
    long long abort_ticks = 30000;       // HVRT_TILE_ABORT_US: 300 us (see render_batch_part)
    uint32_t few_tiles = 2048;           // HVRT_FEW_TILES (RenderArgs::few_tiles)
    uint32_t few_serial = 0;             // number of the last launch that ran the give-up variant with a report (split_hint[2], [3]); 0: the last launch did not
    uint32_t few_counter = 0;
    hipEvent_t ev_split_fork = nullptr, ev_split_join = nullptr;
    hipEvent_t ev_fill_fork = nullptr, ev_fill_join = nullptr;      // k_fill_background runs beside the traversal, on df_stream
    // pinned, written by k_order_scan: how many tiles the last order build found that qualify for k_render_split.
    // Read without waiting -- a stale answer only delays the switch by a launch.  Starts at "some".
    uint32_t* split_hint = nullptr;
    uint64_t* stats_host = nullptr;        // pinned landing place of a build's eight statistics words (a copy into pageable memory waits inside hipMemcpyAsync)
    uint32_t* split_hint_dev = nullptr;    // the same word as the device addresses it
    unsigned long long* cost_sum = nullptr;   // 4 x u64: this launch's and the previous launch's sum of tile costs [0, 1] and walked-tile counts [2, 3]
    int cost_parity = 0;
    bool adaptive_order = true;    // HVRT_TILE_ORDER=0 keeps the natural order
    bool adaptive_split = true;    // HVRT_TILE_SPLIT=0 leaves every tile to k_render
    // tile cover (hvrt_cover.hip): a pre-pass over the triangles in front of every launch without counters marks the tiles in
    // which a ray can hit something; k_render does not walk the others.  HVRT_TILE_COVER=0 walks every tile.
    bool tile_cover = true;
    uint32_t* cover_bits = nullptr; uint64_t cover_capacity = 0;
    uint32_t* walk_queue = nullptr; uint64_t walk_queue_capacity = 0;      // the marked tiles of a launch, in queue order (launch_compact_queue)
    uint32_t* compact_counts = nullptr; uint64_t compact_counts_capacity = 0;      // marked tiles per 256 slots of the order (launch_compact_queue)
    CoverArgs cover_last = {};     // the last launch that ran the pre-pass (n_frames == 0: none since the scene changed)
    // ... and its cameras: a launch with the same cameras, frame size and mesh finds its cover standing (HVRT_TILE_COVER_CACHE=0:
    // every launch runs the pre-pass)
    hvrt_camera_frame cover_frames[HVRT_MAX_BATCH_FRAMES] = {};
    bool cover_cache = true;
    // supersampled launches render their over-long tiles sample by sample inside k_render (RenderArgs::sample_rgb) instead of
    // handing them to k_render_split; HVRT_SAMPLE_SPLIT=0: the split kernel for every launch
    bool sample_split = true;
    float4* sample_rgb = nullptr; uint64_t sample_rgb_capacity = 0;
    uint32_t* sample_done = nullptr; uint64_t sample_done_capacity = 0;
    bool cost_valid = false;
    int cost_key[8] = {};
    hvrt_camera_frame cost_cam = {};   // the (first) camera of the launch the tile durations come from
    float history_max_rad = 2.5f * 3.14159265f / 180.0f;   // HVRT_HISTORY_MAX_DEG: durations of a view further away than this are not used (0: always used)
    // the order of the NEXT launch of the same shape is built right behind a launch (queued while its kernel runs, so
    // the three small kernels cost their run time and not three launch latencies in front of the next traversal);
    // likewise the reset of the queue head and of the next launch's cost sum
    bool order_ahead = false;      // tile_order / split_tiles hold the order for a launch with cost_key
    bool order_may_split = false;  // ... built with the hand-over of over-long tiles to k_render_split
    bool queue_clean = false;      // tile_counter and cost_sum[cost_parity] are zero
    unsigned long long* counters = nullptr;   // 2 x u64
    void* out_buf[3] = {};  uint64_t out_cap[3] = {};   // rgb / hit / t staging for host-facing render
    void* io_buf[3] = {};   uint64_t io_cap[3] = {};    // trace staging: origins / dirs / hits

    // large downloads go through a ring of pinned bounce buffers (a DMA into pageable memory runs at a third of the
    // link): 8 x 4 MiB, the DMA of chunk k + 8 queued as soon as copy thread (k mod 3) has emptied chunk k
    // (scripts/ubench/host_xfer.hip: 6 GB/s plain, 17 GB/s with two 8-MiB buffers and one copying thread, 40 GB/s so)
    static constexpr size_t PIN_CHUNK = 4u << 20;
    static constexpr int PIN_RING = 8, PIN_THREADS = 3;
    char* pin_ring = nullptr;
    hipEvent_t pin_ev[PIN_RING] = {};
    uint64_t last_hits = 0;        // hvrt_last_hit_count

    int mode = 0;   // hvrt_set_mode

    // this device's copy of the file-scope scratch array (hvrt_render_part_device_noalloc), resolved in hvrt_create
    unsigned char* noalloc_scratch = nullptr;

    hvrt_timings tm = {};
    bool render_pending = false;   // ev[6..7] recorded by an asynchronous render, not yet read
};

// ---- scratch of hvrt_render_part_device_noalloc -------------------------------------------------------------
// Everything that entry point's kernels need beyond the caller's inputs and outputs: the work-queue head, the sum
// of tile durations, the camera frame (the kernel reads it with scalar loads from device memory) and one duration
// word per 8x8 tile.  One copy per GPU, held until the process ends; `scratch` is only a host-side stand-in for it --
// the device address comes from hipGetSymbolAddress (hvrt_create, with the device current) and is what every use
// below goes through.
constexpr uint32_t NOALLOC_MAX_TILES = 1u << 20;     // 8x8-pixel tiles per call: 7680x4320 has 518 400
constexpr size_t NOALLOC_QUEUE_HEAD = 0;             // u32 queue head, u32 unused
constexpr size_t NOALLOC_COST_SUM = 8;               // u64
constexpr size_t NOALLOC_FRAME = 64;                 // hvrt_camera_frame, 48 bytes
constexpr size_t NOALLOC_TILE_COST = 256;            // u32 x NOALLOC_MAX_TILES
constexpr size_t NOALLOC_BYTES = NOALLOC_TILE_COST + 4ull * NOALLOC_MAX_TILES;     // 4 MiB + 256 B
static __device__ __align__(256) unsigned char scratch[NOALLOC_BYTES];

// the camera frame travels as a kernel argument (no staging buffer, no copy from pageable memory)
__global__ __launch_bounds__(64) void k_store_frame(hvrt_camera_frame f, float* __restrict__ dst) {
    const float* src = f.origin;     // the struct is twelve consecutive floats
    if (threadIdx.x < sizeof(hvrt_camera_frame) / sizeof(float)) dst[threadIdx.x] = src[threadIdx.x];
}

namespace {

thread_local std::string g_create_error;

int fail(hvrt_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->error = buf; else g_create_error = buf;
    return code;
}

#define HIP_TRY(c, call)                                                                               \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail((c), e_ == hipErrorOutOfMemory ? HVRT_ERR_OOM : HVRT_ERR_HIP, "%s failed: %s", \
                        #call, hipGetErrorString(e_));                                                 \
    } while (0)

template <typename T>
int ensure(hvrt_ctx* c, T*& p, uint64_t& cap, uint64_t need) {
    if (need <= cap && p) return HVRT_OK;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    if (need == 0) need = 1;
    hipError_t e = hipMalloc((void**)&p, need * sizeof(T));
    if (e != hipSuccess) { p = nullptr; return fail(c, HVRT_ERR_OOM, "hipMalloc of %llu bytes failed: %s",
                                                    (unsigned long long)(need * sizeof(T)), hipGetErrorString(e)); }
    cap = need;
    return HVRT_OK;
}
int ensure_bytes(hvrt_ctx* c, void*& p, uint64_t& cap, uint64_t need) {
    char* q = (char*)p;
    int r = ensure<char>(c, q, cap, need);
    p = q;
    return r;
}

int bind(hvrt_ctx* c) {
    HIP_TRY(c, hipSetDevice(c->device));
    return HVRT_OK;
}

// Device -> caller's (pageable) host memory on stream s.  Small copies are issued directly and complete at the
// caller's next stream synchronisation; large ones are pipelined through the context's ring of pinned chunks -- this
// thread queues the DMAs, PIN_THREADS helper threads copy finished chunks out -- and have completed on return.
// The source is one compact run of bytes; the destination may be cut into blocks: block j (block_bytes of the source,
// the last one possibly shorter) lands at dst + j * dst_stride.  That is how a member of a device group puts its row
// tiles where they belong in the caller's frame (hvrt_group_render); block_bytes >= bytes is the plain copy.
int download(hvrt_ctx* c, void* dst, const void* src, uint64_t bytes, hipStream_t s, uint64_t block_bytes = ~0ull, uint64_t dst_stride = 0) {
    if (bytes == 0) return HVRT_OK;
    constexpr uint64_t CH = hvrt_ctx::PIN_CHUNK;
    constexpr int R = hvrt_ctx::PIN_RING, T = hvrt_ctx::PIN_THREADS;
    const bool blocks = block_bytes < bytes;
    if (!blocks) { block_bytes = bytes; dst_stride = 0; }
    // where the source bytes [off, off + len) go: piece by piece, a piece never crossing a block boundary
    auto scatter = [&](uint64_t off, uint64_t len, auto&& piece) {
        while (len) {
            const uint64_t j = off / block_bytes, within = off - j * block_bytes;
            const uint64_t n = block_bytes - within < len ? block_bytes - within : len;
            piece((char*)dst + j * dst_stride + within, off, n);
            off += n; len -= n;
        }
    };
    // (a scattered copy takes the ring from one chunk on: its blocks would otherwise be as many small DMAs into pageable memory)
    if (bytes < (blocks ? CH : 4 * CH) || !c->pin_ring) {
        hipError_t err = hipSuccess;
        scatter(0, bytes, [&](char* d, uint64_t off, uint64_t n) {
            if (err == hipSuccess) err = hipMemcpyAsync(d, (const char*)src + off, n, hipMemcpyDeviceToHost, s);
        });
        if (err != hipSuccess) return fail(c, HVRT_ERR_HIP, "download: %s", hipGetErrorString(err));
        return HVRT_OK;
    }
    const uint64_t n = (bytes + CH - 1) / CH;
    auto size_of = [&](uint64_t k) { return k + 1 < n ? CH : bytes - k * CH; };
    std::vector<std::atomic<int>> issued(n), emptied(n);
    for (uint64_t k = 0; k < n; ++k) { issued[k].store(0); emptied[k].store(0); }
    std::atomic<int> failed{0};
    const int device = c->device;
    auto copier = [&](int t) {
        (void)hipSetDevice(device);
        for (uint64_t k = (uint64_t)t; k < n; k += T) {
            while (!issued[k].load(std::memory_order_acquire)) { if (failed.load()) return; std::this_thread::yield(); }
            if (hipEventSynchronize(c->pin_ev[k % R]) != hipSuccess) { failed.store(1); return; }
            const char* from = c->pin_ring + (k % R) * CH;
            scatter(k * CH, size_of(k), [&](char* d, uint64_t off, uint64_t len) { memcpy(d, from + (off - k * CH), len); });
            emptied[k].store(1, std::memory_order_release);
        }
    };
    std::thread workers[T];
    for (int t = 0; t < T; ++t) workers[t] = std::thread(copier, t);
    hipError_t err = hipSuccess;
    for (uint64_t k = 0; k < n && err == hipSuccess && !failed.load(); ++k) {
        if (k >= (uint64_t)R)      // the ring slot is free once its previous chunk has been copied out
            while (!emptied[k - R].load(std::memory_order_acquire)) { if (failed.load()) break; std::this_thread::yield(); }
        err = hipMemcpyAsync(c->pin_ring + (k % R) * CH, (const char*)src + k * CH, size_of(k), hipMemcpyDeviceToHost, s);
        if (err == hipSuccess) err = hipEventRecord(c->pin_ev[k % R], s);
        if (err != hipSuccess) { failed.store(1); break; }
        issued[k].store(1, std::memory_order_release);
    }
    for (auto& w : workers) w.join();
    if (err != hipSuccess || failed.load())
        return fail(c, HVRT_ERR_HIP, "download: %s", err != hipSuccess ? hipGetErrorString(err) : "a copy thread failed waiting for its chunk");
    return HVRT_OK;
}

uint32_t* offsets_of(const hvrt_ctx* c) { return c->cnt_base + 3; }
uint32_t* cnt_of(const hvrt_ctx* c) { return c->cnt_base + 4; }   // 16-byte aligned

uint64_t padded_vox(uint64_t nvox) { return (nvox + SCAN_CHUNK - 1) / SCAN_CHUNK * SCAN_CHUNK; }

int check_grid_args(hvrt_ctx* c, int nx, int ny, int nz, const float* mn, const float* mx, const float* vs) {
    if (!mn || !mx || !vs) return fail(c, HVRT_ERR_INVALID, "grid: null bounds / voxel size");
    if (nx < 1 || ny < 1 || nz < 1) return fail(c, HVRT_ERR_INVALID, "grid: dimensions must be >= 1 (got %d x %d x %d)", nx, ny, nz);
    if (nx > 65536 || ny > 65536 || nz > 65536) return fail(c, HVRT_ERR_INVALID, "grid: at most 65536 voxels per axis");
    uint64_t nvox = (uint64_t)nx * ny * nz;
    if (nvox > 0xFFFFFFFFull - 2 * SCAN_CHUNK) return fail(c, HVRT_ERR_INVALID, "grid: at most 2^32 - 65536 voxels");
    return HVRT_OK;
}

void set_grid_params(hvrt_ctx* c, int nx, int ny, int nz, const float* mn, const float* mx, const float* vs) {
    c->g.nx = nx; c->g.ny = ny; c->g.nz = nz;
    c->g.mnx = mn[0]; c->g.mny = mn[1]; c->g.mnz = mn[2];
    c->g.vsx = vs[0]; c->g.vsy = vs[1]; c->g.vsz = vs[2];
    c->g.nvox = (uint64_t)nx * ny * nz;
    memcpy(c->max_bound, mx, 12);
    c->g.bxx = mx[0]; c->g.bxy = mx[1]; c->g.bxz = mx[2];
}

SceneDev scene_of(const hvrt_ctx* c) {
    SceneDev s;
    s.g = c->g;
    s.tri_rec = c->tri_rec;
    s.offsets = offsets_of(c);
    s.indices = c->indices;
    s.n_tris = c->n_tris;
    s.ref_rec = c->ref_rec_valid ? c->ref_rec : nullptr;
    s.face_bits = c->face_bits;
    s.face_words = c->face_words;
    s.dist = c->dist;
    return s;
}

int finish_accel(hvrt_ctx* c);

// may_allocate = false: the caller promised not to allocate (hvrt_render_part_device_noalloc); derived data that was
// deferred must then have been built already (hvrt_finish_accel)
int require_scene(hvrt_ctx* c, bool may_allocate = true) {
    if (!c->n_tris && !c->tri_rec) return fail(c, HVRT_ERR_INVALID, "no mesh: call hvrt_set_mesh first");
    if (!c->has_grid) return fail(c, HVRT_ERR_INVALID, "no grid: call hvrt_build_grid or hvrt_set_grid_csr first");
    if (c->accel_pending) {
        if (!may_allocate)
            return fail(c, HVRT_ERR_INVALID, "the grid's deferred acceleration data is not built yet: call hvrt_finish_accel first");
        return finish_accel(c);
    }
    return HVRT_OK;
}

// Builds the per-reference records (and validates every CSR index against the mesh).  Falls back to the
// indexed traversal when switched off or when the array does not fit in device memory.
// The distance field needs only the finished offsets, so it runs on a stream of its own beside the kernels that
// order the lists and derive face bits and records (small grids: those kernels are a few waves each and
// overlap almost entirely).  fork after the offsets are final, join before the build returns.
int fork_distance_field(hvrt_ctx* c, hipStream_t s) {
    if (int r = ensure<uint8_t>(c, c->dist, c->dist_capacity, c->g.nvox + 64)) return r;
    HIP_TRY(c, hipEventRecord(c->ev_fork, s));
    HIP_TRY(c, hipStreamWaitEvent(c->df_stream, c->ev_fork, 0));
    HIP_TRY(c, launch_distance_field(offsets_of(c), c->occ_bits_valid ? c->occ_bits : nullptr, c->g, c->dist, c->df_stream));
    HIP_TRY(c, hipEventRecord(c->ev_join, c->df_stream));
    c->dist_in_flight = true;
    return HVRT_OK;
}
int join_distance_field(hvrt_ctx* c, hipStream_t s) {
    if (!c->dist_in_flight) return HVRT_OK;
    c->dist_in_flight = false;
    HIP_TRY(c, hipStreamWaitEvent(s, c->ev_join, 0));
    return HVRT_OK;
}

// rc: the number of entries (known, or still on the device: see RefCount).  With a live count the arrays were sized by
// an earlier build (ref_capacity_of) and nothing is allocated here.
int make_ref_records_inner(hvrt_ctx* c, hipStream_t s, bool validate, const RefCount& rc) {
    c->ref_rec_valid = false;
    if (!c->dist_in_flight) if (int r = fork_distance_field(c, s)) return r;
    // face bits: one bit per entry and face, padded by two words so that a 32-entry window read at the
    // last entry stays inside the array
    const uint64_t n_alloc = rc.live ? rc.cap : rc.n;
    if (int r = ensure<uint32_t>(c, c->face_bits, c->face_bits_capacity, 6 * face_words_for(n_alloc))) return r;
    HIP_TRY(c, launch_face_bits(c->indices, rc, offsets_of(c), c->g, c->n_tris, (validate || !c->lists_follow_ranges) ? nullptr : c->ranges,
                                validate ? nullptr : c->vox_of, c->face_bits, s));
    if (n_alloc == 0) return HVRT_OK;
    const bool want = c->inline_records;
    if (want) {
        if (c->ref_rec_capacity < n_alloc * 3) {
            if (c->ref_rec) { (void)hipFree(c->ref_rec); c->ref_rec = nullptr; c->ref_rec_capacity = 0; }
            // + 64 float4s: the traversal fetches a list chunk with one 16-byte load per lane of the wave, and the
            // lanes behind the last record of the last list read (and ignore) what follows it
            hipError_t e = hipMalloc((void**)&c->ref_rec, n_alloc * 48 + 64 * sizeof(float4));
            if (e == hipSuccess) c->ref_rec_capacity = n_alloc * 3;
            else { (void)hipGetLastError(); c->ref_rec = nullptr; }   // does not fit: indexed mode
        }
    }
    if (!want || !c->ref_rec) return HVRT_OK;
    HIP_TRY(c, hipMemsetAsync(c->stats + 4, 0, sizeof(uint64_t), s));
    HIP_TRY(c, launch_expand_records(c->indices, rc, c->tri_rec, c->n_tris, c->ref_rec, c->stats, s));
    c->ref_rec_valid = true;
    return HVRT_OK;
}

// Builds the derived acceleration data; always rejoins the distance-field stream, whatever the outcome.
int make_ref_records(hvrt_ctx* c, hipStream_t s, bool validate, const RefCount& rc) {
    const int r = make_ref_records_inner(c, s, validate, rc);
    const int j = join_distance_field(c, s);
    return r ? r : j;
}

// How many CSR entries the arrays left by earlier builds can take (0: some array is missing): a build may then go on
// past the scan without waiting for the host to read the count back.
uint64_t ref_capacity_of(const hvrt_ctx* c, bool with_derived_data) {
    if (!c->indices || !c->fill_scratch || !c->long_queue) return 0;
    uint64_t cap = c->indices_capacity;
    auto lower = [&](uint64_t v) { if (v < cap) cap = v; };
    lower(c->fill_scratch_capacity / 2);
    lower(c->long_queue_capacity ? (c->long_queue_capacity - 1) * 64 : 0);       // sized total / 64 + 2
    if (with_derived_data) {
        lower(c->face_bits && c->face_bits_capacity >= 18 ? (c->face_bits_capacity / 6 - 2) / 2 * 64 : 0);
        if (c->inline_records) lower(c->ref_rec ? c->ref_rec_capacity / 3 : 0);
    }
    return cap > 0xFFFFFFFFull ? 0xFFFFFFFFull : cap;
}

float ms_between(hipEvent_t a, hipEvent_t b) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, a, b) != hipSuccess) { (void)hipGetLastError(); return -1.0f; }      // (an event never recorded: not an error to carry along)
    return ms;
}

// The derived data of a grid whose build deferred it (hvrt_set_deferred_accel): distance field, face bits, records.
int finish_accel(hvrt_ctx* c) {
    if (!c->accel_pending) return HVRT_OK;
    hipStream_t s = c->stream;
    HIP_TRY(c, hipEventRecord(c->ev[5], s));
    const RefCount rc = { nullptr, c->n_refs, c->n_refs };
    const int r = make_ref_records(c, s, !c->accel_built_here, rc);
    c->vox_of = nullptr;
    if (r) return r;
    HIP_TRY(c, hipEventRecord(c->ev[8], s));
    HIP_TRY(c, hipStreamSynchronize(s));
    c->accel_pending = false;
    c->tm.build_expand_ms = ms_between(c->ev[5], c->ev[8]);
    return HVRT_OK;
}

}  // namespace

extern "C" {

int hvrt_create(hvrt_ctx** out, int device_id) {
    if (!out) return fail(nullptr, HVRT_ERR_INVALID, "hvrt_create: out is null");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, HVRT_ERR_NO_DEVICE, "no HIP device available (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= n)
        return fail(nullptr, HVRT_ERR_INVALID, "device_id %d out of range (0..%d)", device_id, n - 1);
    hvrt_ctx* c = new hvrt_ctx();
    c->device = device_id;
    if (const char* e = getenv("HVRT_INLINE_RECORDS")) c->inline_records = atoi(e) != 0;
    auto bail = [&](const char* what, hipError_t err) {
        fail(nullptr, HVRT_ERR_HIP, "%s failed: %s", what, hipGetErrorString(err));
        hvrt_destroy(c);
        return HVRT_ERR_HIP;
    };
    if ((e = hipSetDevice(device_id)) != hipSuccess) return bail("hipSetDevice", e);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess) return bail("hipGetDeviceProperties", e);
    c->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if ((e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    c->stream = c->own_stream;
    if ((e = hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    {
        // Normal priority unless HVRT_DF_LOW_PRIORITY=1.  The distance field used to run on a lowest-priority stream, to
        // keep out of the scatter's way; but a low-priority hardware queue is what the GPU's scheduler serves last when
        // the card is shared: with three processes holding queues on one MI355X (a test run: pytest and two bench ranks)
        // each build then took 20 to 30 SECONDS, its other phases long done and this stream still "busy".
        int least = 0, greatest = 0;
        const char* low = getenv("HVRT_DF_LOW_PRIORITY");
        if (!(low && atoi(low) != 0) || hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = 0; }
        if ((e = hipStreamCreateWithPriority(&c->df_stream, hipStreamNonBlocking, least)) != hipSuccess) return bail("hipStreamCreate", e);
    }
    if ((e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    for (auto& ev : c->ev_seg)
        if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreateWithFlags(&c->ev_split_fork, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreateWithFlags(&c->ev_fill_fork, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreateWithFlags(&c->ev_fill_join, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreateWithFlags(&c->ev_split_join, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipHostMalloc((void**)&c->split_hint, 128, hipHostMallocDefault)) != hipSuccess) return bail("hipHostMalloc", e);
    c->split_hint[0] = 1u; c->split_hint[1] = 0u; c->split_hint[2] = 0u; c->split_hint[3] = 0u;
    c->stats_host = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(c->split_hint) + 64);      // 8 words of the same pinned block
    if ((e = hipHostGetDevicePointer((void**)&c->split_hint_dev, c->split_hint, 0)) != hipSuccess) return bail("hipHostGetDevicePointer", e);
    for (auto& ev : c->ev)
        if (&ev != &c->ev[6] && &ev != &c->ev[7] && (e = hipEventCreate(&ev)) != hipSuccess) return bail("hipEventCreate", e);
    for (auto& pair : c->render_ring)
        for (auto& ev : pair)
            if ((e = hipEventCreate(&ev)) != hipSuccess) return bail("hipEventCreate", e);
    c->ev[6] = c->render_ring[0][0];        // (aliases: the ring owns them)
    c->ev[7] = c->render_ring[0][1];
    if ((e = hipMalloc((void**)&c->stats, STATS_STATUS * sizeof(uint64_t))) != hipSuccess) return bail("hipMalloc", e);
    c->stats_capacity = STATS_STATUS;
    if ((e = hipMalloc((void**)&c->tile_counter, 2048)) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc((void**)&c->cost_sum, 256)) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc((void**)&c->cam_dev, sizeof(hvrt_camera_frame) * hvrt_ctx::CAM_SLOTS * hvrt_ctx::CAM_MAX)) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipHostMalloc((void**)&c->cam_host, sizeof(hvrt_camera_frame) * hvrt_ctx::CAM_SLOTS * hvrt_ctx::CAM_MAX, hipHostMallocDefault)) != hipSuccess) return bail("hipHostMalloc", e);
    for (auto& ev : c->cam_ev)
        if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    if (const char* v = getenv("HVRT_BUILD_REUSE")) c->reuse_arrays = atoi(v) != 0;
    if (const char* v = getenv("HVRT_BUILD_NARROW")) c->narrow_counters = atoi(v) != 0;
    if (const char* v = getenv("HVRT_SPLIT_WAIT")) c->wait_for_split = atoi(v) != 0;
    if (const char* v = getenv("HVRT_OVERLAP")) c->overlap_mode = strcmp(v, "sat") == 0 ? 1 : 0;      // the default of hvrt_set_overlap_test
    if (const char* v = getenv("HVRT_TILE_ORDER")) c->adaptive_order = atoi(v) != 0;
    if (const char* v = getenv("HVRT_TILE_SPLIT")) c->adaptive_split = atoi(v) != 0;
    if (const char* v = getenv("HVRT_TILE_COVER")) c->tile_cover = atoi(v) != 0;
    if (const char* v = getenv("HVRT_TILE_COVER_CACHE")) c->cover_cache = atoi(v) != 0;
    if (const char* v = getenv("HVRT_SAMPLE_SPLIT")) c->sample_split = atoi(v) != 0;
    if (const char* v = getenv("HVRT_HISTORY_MAX_DEG")) c->history_max_rad = (float)atof(v) * 3.14159265f / 180.0f;
    if (const char* v = getenv("HVRT_TILE_ABORT_US")) {
        const long long us = atoll(v);
        c->abort_ticks = us > 0 ? (us < (1ll << 40) ? us * 100 : (1ll << 47)) : 0;      // 0: the plain kernel is launched
    }
    if (const char* v = getenv("HVRT_FEW_TILES")) { const long long n = atoll(v); c->few_tiles = n > 0 ? (uint32_t)(n < 0x7FFFFFFF ? n : 0x7FFFFFFF) : 0u; }
    if ((e = hipMalloc((void**)&c->counters, 256)) != hipSuccess) return bail("hipMalloc", e);
    // the download ring belongs to the construction of a tracer, not to its first render()
    if ((e = hipHostMalloc((void**)&c->pin_ring, hvrt_ctx::PIN_CHUNK * hvrt_ctx::PIN_RING, hipHostMallocDefault)) != hipSuccess) return bail("hipHostMalloc", e);
    for (auto& ev : c->pin_ev)
        if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    // this device's copy of the scratch array (every GPU holds its own, at its own address)
    if ((e = hipGetSymbolAddress((void**)&c->noalloc_scratch, HIP_SYMBOL(scratch))) != hipSuccess) return bail("hipGetSymbolAddress(scratch)", e);
    if ((e = hipMalloc((void**)&c->bounds_v, 2048 * 6 * sizeof(float))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc((void**)&c->bounds_i, 2048 * 6 * sizeof(uint64_t))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc((void**)&c->bounds_out, 256)) != hipSuccess) return bail("hipMalloc", e);
    // The reference constructs a tracer per image and renders once (TraceImages.cpp:215-229): what a process pays once
    // per hardware queue -- its scratch memory, the first pass through the download ring -- is paid here, not inside
    // that one render() (HVRT_WARMUP=0: not at all).
    // (the queue scratch starts at zero whatever HVRT_WARMUP says: the word the split kernel's workgroups report to is never
    // reset afterwards, it is compared with a running total)
    if ((e = hipMemsetAsync(c->tile_counter, 0, 2048, c->stream)) != hipSuccess) return bail("hipMemsetAsync", e);
    if ((e = hipMemsetAsync(c->cost_sum, 0, 256, c->stream)) != hipSuccess) return bail("hipMemsetAsync", e);
    const char* warm = getenv("HVRT_WARMUP");
    if (!warm || atoi(warm) != 0) {
        // the download ring on the copy stream WHILE the main stream is busy with the launches -- that is how a frame
        // comes back (hvrt_render's two-launch path), and streams that are busy at the same time for the first time cost
        // the process a hardware queue's set-up -- then once on the main stream
        const size_t wb = 4 * hvrt_ctx::PIN_CHUNK;
        void* d_tmp = nullptr;
        std::vector<char> h_tmp(wb);
        if ((e = hipMalloc(&d_tmp, wb)) != hipSuccess) return bail("hipMalloc", e);
        if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return bail("hipStreamSynchronize", e);
        e = launch_scratch_warmup(c->tile_counter, c->n_cus, c->stream);
        if (e != hipSuccess) { (void)hipFree(d_tmp); return bail("warm-up launches", e); }
        // (the two other streams -- k_render_split beside k_render, the distance field beside the fill -- each with work
        // of its own in flight at the same time)
        // -- and the scratch of the second stream's queue)
        e = launch_scratch_warmup((uint32_t*)d_tmp, c->n_cus, c->aux_stream);
        if (e != hipSuccess) { (void)hipFree(d_tmp); return bail("warm-up launches", e); }
        (void)hipMemsetAsync((char*)c->counters + 192, 0, 64, c->df_stream);
        // (the copy stream waits for an event of the main stream first, as it does for a half frame)
        (void)hipEventRecord(c->ev_seg[0], c->stream);
        (void)hipStreamWaitEvent(c->copy_stream, c->ev_seg[0], 0);
        int rc = download(c, h_tmp.data(), d_tmp, wb, c->copy_stream);
        if (rc == HVRT_OK && hipStreamSynchronize(c->copy_stream) != hipSuccess) rc = HVRT_ERR_HIP;
        if (rc == HVRT_OK && (hipStreamSynchronize(c->aux_stream) != hipSuccess || hipStreamSynchronize(c->df_stream) != hipSuccess)) rc = HVRT_ERR_HIP;
        if (rc == HVRT_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = HVRT_ERR_HIP;
        if (rc == HVRT_OK) rc = download(c, h_tmp.data(), d_tmp, wb, c->stream);
        if (rc == HVRT_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = HVRT_ERR_HIP;
        (void)hipFree(d_tmp);
        if (rc != HVRT_OK) { g_create_error = "warm-up of the download ring failed: " + c->error; hvrt_destroy(c); return rc; }
    }
    *out = c;
    return HVRT_OK;
}

void hvrt_destroy(hvrt_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->own_stream) (void)hipStreamSynchronize(c->stream);
    if (c->aux_stream) (void)hipStreamSynchronize(c->aux_stream);
    if (c->df_stream) (void)hipStreamSynchronize(c->df_stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    void* bufs[] = { c->tri_aos, c->tri_rec, c->cnt_base, c->indices, c->ref_rec, c->face_bits, c->dist, c->ranges, c->sat_mask, c->cnt16, c->occ_bits, c->stats, c->long_queue, c->fill_scratch,
                     c->bounds_v, c->bounds_i, c->bounds_out, c->tile_counter, c->counters, c->cost_sum, c->order_blocks, c->cam_dev, c->tile_cost, c->tile_order, c->split_tiles, c->abandoned, c->cover_bits, c->walk_queue, c->compact_counts, c->sample_rgb, c->sample_done,
                     c->out_buf[0], c->out_buf[1], c->out_buf[2], c->io_buf[0], c->io_buf[1], c->io_buf[2] };
    for (void* b : bufs) if (b) (void)hipFree(b);
    c->ev[6] = c->ev[7] = nullptr;          // aliases of ring entries
    for (auto& ev : c->ev) if (ev) (void)hipEventDestroy(ev);
    for (auto& pair : c->render_ring) for (auto& ev : pair) if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : c->cam_ev) if (ev) (void)hipEventDestroy(ev);
    if (c->cam_host) (void)hipHostFree(c->cam_host);
    if (c->split_hint) (void)hipHostFree(c->split_hint);
    if (c->pin_ring) (void)hipHostFree(c->pin_ring);
    for (auto& ev : c->pin_ev) if (ev) (void)hipEventDestroy(ev);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
    if (c->df_stream) (void)hipStreamDestroy(c->df_stream);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    for (auto& ev : c->ev_seg) if (ev) (void)hipEventDestroy(ev);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_split_fork) (void)hipEventDestroy(c->ev_split_fork);
    if (c->ev_fill_fork) (void)hipEventDestroy(c->ev_fill_fork);
    if (c->ev_fill_join) (void)hipEventDestroy(c->ev_fill_join);
    if (c->ev_split_join) (void)hipEventDestroy(c->ev_split_join);
    delete c;
}

const char* hvrt_last_error(const hvrt_ctx* c) { return c ? c->error.c_str() : g_create_error.c_str(); }

void* hvrt_stream(hvrt_ctx* c) { return c ? (void*)c->stream : nullptr; }

int hvrt_set_stream(hvrt_ctx* c, void* hip_stream) {
    if (!c) return HVRT_ERR_INVALID;
    if (int r = bind(c)) return r;
    HIP_TRY(c, hipStreamSynchronize(c->stream));   // nothing of ours may still run on the old stream
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return HVRT_OK;
}

static void next_render_events(hvrt_ctx* c) {
    const int k = (int)(c->renders_recorded++ % hvrt_ctx::RENDER_RING);
    c->ev[6] = c->render_ring[k][0];
    c->ev[7] = c->render_ring[k][1];
}

int hvrt_synchronize(hvrt_ctx* c) {
    if (!c) return HVRT_ERR_INVALID;
    if (int r = bind(c)) return r;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->render_pending) { c->tm.render_ms = ms_between(c->ev[6], c->ev[7]); c->render_pending = false; }
    return HVRT_OK;
}

int hvrt_set_overlap_test(hvrt_ctx* c, int mode) {
    if (!c) return HVRT_ERR_INVALID;
    if (mode != 0 && mode != 1) return fail(c, HVRT_ERR_INVALID, "set_overlap_test: mode %d is neither 0 (bounding box, the reference's rule) nor 1 (separating axes)", mode);
    c->overlap_mode = mode;
    return HVRT_OK;
}

int hvrt_set_mode(hvrt_ctx* c, int mode) {
    if (!c) return HVRT_ERR_INVALID;
    if (mode != 0 && mode != 1) return fail(c, HVRT_ERR_INVALID, "set_mode: mode %d is neither 0 (reference) nor 1 (nearest hit)", mode);
    if (mode != c->mode) c->cost_valid = false;     // tile durations of the other mode say nothing about this one
    c->mode = mode;
    return HVRT_OK;
}

int hvrt_get_mode(const hvrt_ctx* c) { return c ? c->mode : HVRT_ERR_INVALID; }

int hvrt_get_counter_width(const hvrt_ctx* c) { return c ? (c->narrow_counters && c->overlap_mode == 0 ? 16 : 32) : HVRT_ERR_INVALID; }

int hvrt_clear_scene(hvrt_ctx* c) {
    if (!c) return HVRT_ERR_INVALID;
    if (int r = bind(c)) return r;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    (void)join_distance_field(c, c->stream);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->n_tris = 0;
    c->has_grid = false;
    c->accel_pending = false;
    c->occ_bits_valid = false;
    c->ref_rec_valid = false;
    c->vox_of = nullptr;
    c->n_refs = c->n_occupied = 0;
    c->max_list = 0;
    c->cost_valid = false;
    c->cover_last.n_frames = 0;
    c->order_ahead = false;
    c->mode = 0;
    c->defer_accel = false;
    const char* v = getenv("HVRT_OVERLAP");
    c->overlap_mode = (v && strcmp(v, "sat") == 0) ? 1 : 0;
    c->last_hits = 0;
    c->tm = hvrt_timings{};
    c->error.clear();
    return HVRT_OK;
}

int hvrt_set_deferred_accel(hvrt_ctx* c, int on) {
    if (!c) return HVRT_ERR_INVALID;
    c->defer_accel = on != 0;
    return HVRT_OK;
}

int hvrt_finish_accel(hvrt_ctx* c) {
    if (!c) return HVRT_ERR_INVALID;
    if (int r = bind(c)) return r;
    if (!c->has_grid) return fail(c, HVRT_ERR_INVALID, "finish_accel: no grid");
    return finish_accel(c);
}

int hvrt_get_timings(const hvrt_ctx* c, hvrt_timings* out) {
    if (!c || !out) return HVRT_ERR_INVALID;
    *out = c->tm;
    return HVRT_OK;
}

// ---- mesh ------------------------------------------------------------------------------------

static int adopt_mesh(hvrt_ctx* c, const void* src, uint32_t n_tris, hipMemcpyKind kind) {
    if (int r = bind(c)) return r;
    if (n_tris && !src) return fail(c, HVRT_ERR_INVALID, "set_mesh: null triangle pointer");
    if (kind == hipMemcpyDeviceToDevice && ((uintptr_t)src & 15))
        return fail(c, HVRT_ERR_INVALID, "set_mesh_device: pointer must be 16-byte aligned");
    if (int r = ensure<float>(c, c->tri_aos, c->aos_capacity, (uint64_t)n_tris * 12)) return r;
    if (int r = ensure<float4>(c, c->tri_rec, c->rec_capacity, (uint64_t)n_tris * 3)) return r;
    c->n_tris = n_tris;
    c->has_grid = false;
    c->cost_valid = false;
    c->cover_last.n_frames = 0;
    (void)join_distance_field(c, c->stream);   // left over from a build that failed half-way
    if (n_tris) {
        HIP_TRY(c, hipMemcpyAsync(c->tri_aos, src, (size_t)n_tris * 48, kind, c->stream));
        HIP_TRY(c, launch_prepare_records(c->tri_aos, c->tri_rec, n_tris, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return HVRT_OK;
}

int hvrt_set_mesh(hvrt_ctx* c, const float* tri_aos, uint32_t n_tris) {
    if (!c) return HVRT_ERR_INVALID;
    return adopt_mesh(c, tri_aos, n_tris, hipMemcpyHostToDevice);
}

int hvrt_set_mesh_device(hvrt_ctx* c, const void* d_tri_aos, uint32_t n_tris) {
    if (!c) return HVRT_ERR_INVALID;
    return adopt_mesh(c, d_tri_aos, n_tris, hipMemcpyDeviceToDevice);
}

static int run_bounds(hvrt_ctx* c, const float* d_pts, uint64_t n_points, float mn[3], float mx[3]) {
    const int nb = bounds_blocks(n_points);
    HIP_TRY(c, launch_bounds(d_pts, n_points, c->bounds_v, c->bounds_i, nb, c->bounds_out, c->stream));
    float h[6];
    HIP_TRY(c, hipMemcpyAsync(h, c->bounds_out, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    memcpy(mn, h, 12);
    memcpy(mx, h + 3, 12);
    return HVRT_OK;
}

int hvrt_compute_bounds(hvrt_ctx* c, const float* pts, uint64_t n_points, float mn[3], float mx[3]) {
    if (!c) return HVRT_ERR_INVALID;
    if (!mn || !mx || (n_points && !pts)) return fail(c, HVRT_ERR_INVALID, "compute_bounds: null pointer");
    if (int r = bind(c)) return r;
    if (int r = ensure_bytes(c, c->io_buf[0], c->io_cap[0], n_points * 12)) return r;
    if (n_points) HIP_TRY(c, hipMemcpyAsync(c->io_buf[0], pts, n_points * 12, hipMemcpyHostToDevice, c->stream));
    return run_bounds(c, (const float*)c->io_buf[0], n_points, mn, mx);
}

int hvrt_mesh_bounds(hvrt_ctx* c, float mn[3], float mx[3]) {
    if (!c) return HVRT_ERR_INVALID;
    if (!mn || !mx) return fail(c, HVRT_ERR_INVALID, "mesh_bounds: null pointer");
    if (int r = bind(c)) return r;
    // corners of triangle t are floats [12t, 12t+9): run the point kernel over a strided view by
    // packing them first would cost a pass; instead treat each triangle's first 9 floats as 3 points
    // via a compact copy (device-to-device, 36 of every 48 bytes).
    if (int r = ensure_bytes(c, c->io_buf[0], c->io_cap[0], (uint64_t)c->n_tris * 36)) return r;
    if (c->n_tris)
        HIP_TRY(c, hipMemcpy2DAsync(c->io_buf[0], 36, c->tri_aos, 48, 36, c->n_tris, hipMemcpyDeviceToDevice, c->stream));
    return run_bounds(c, (const float*)c->io_buf[0], (uint64_t)c->n_tris * 3, mn, mx);
}

float hvrt_voxel_size(const float mn[3], const float mx[3], int nx, int ny, int nz) {
    // SceneCreator.cpp:100-109 (std::max(a,b) = (a < b) ? b : a)
    float gx = mx[0] - mn[0], gy = mx[1] - mn[1], gz = mx[2] - mn[2];
    float m = gx < gy ? gy : gx;
    m = m < gz ? gz : m;
    int r = nx < ny ? ny : nx;
    r = r < nz ? nz : r;
    return m / (float)r;
}

// ---- grid ------------------------------------------------------------------------------------

int hvrt_build_grid(hvrt_ctx* c, int nx, int ny, int nz, const float mn[3], const float mx[3], const float vs[3]) {
    if (!c) return HVRT_ERR_INVALID;
    if (int r = bind(c)) return r;
    if (int r = check_grid_args(c, nx, ny, nz, mn, mx, vs)) return r;
    if (!c->tri_aos && c->n_tris) return fail(c, HVRT_ERR_INVALID, "build_grid: no mesh");
    c->has_grid = false;
    c->cost_valid = false;
    c->accel_pending = false;
    const bool sat = c->overlap_mode == 1;
    c->lists_follow_ranges = !sat;          // (read when the derived data is built, inside this call or deferred)
    (void)join_distance_field(c, c->stream);   // left over from a build that failed half-way
    set_grid_params(c, nx, ny, nz, mn, mx, vs);
    const uint64_t nvox = c->g.nvox, npad = padded_vox(nvox), n_chunks = npad / SCAN_CHUNK;

    if (int r = ensure<uint32_t>(c, c->cnt_base, c->cnt_capacity, npad + 8)) return r;
    if (int r = ensure<uint32_t>(c, c->ranges, c->ranges_capacity, (uint64_t)c->n_tris * 3)) return r;
    if (sat) if (int r = ensure<uint32_t>(c, c->sat_mask, c->sat_mask_capacity, (uint64_t)c->n_tris)) return r;
    if (int r = ensure<uint32_t>(c, c->occ_bits, c->occ_bits_capacity, npad / 32)) return r;
    if (c->stats_capacity < STATS_STATUS + 2 * n_chunks) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));            // a render may still be reading nothing of it, but be sure
        if (int r = ensure<uint64_t>(c, c->stats, c->stats_capacity, STATS_STATUS + 2 * n_chunks)) return r;
    }
    c->occ_bits_valid = false;

    hipStream_t s = c->stream;
    uint64_t* const status_aux = c->stats + STATS_STATUS;
    // count + scan.  The counters are 16-bit while no list of this context has outgrown them: 2 bytes per voxel to
    // clear and to read instead of 4, and the 32-bit offsets array is written once instead of cleared, counted into
    // and scanned in place (1024^3: 8.6 GB of traffic instead of 12.9).  A counter that carried shows as a total
    // that differs from the count pass's own (stats[0] != stats[7]); the front is then issued again with wide counters.
    auto front = [&](bool narrow) -> int {
        HIP_TRY(c, hipEventRecord(c->ev[0], s));
        if (narrow) {
            if (int r = ensure<uint32_t>(c, c->cnt16, c->cnt16_capacity, npad / 2)) return r;
            HIP_TRY(c, hipMemsetAsync(c->cnt16, 0, npad / 2 * sizeof(uint32_t), s));
            HIP_TRY(c, launch_build_reset(c->cnt_base, c->stats, STATS_STATUS + n_chunks, s));       // offsets[0] and the statistics block
        } else {
            HIP_TRY(c, hipMemsetAsync(c->cnt_base, 0, (npad + 8) * sizeof(uint32_t), s));
            HIP_TRY(c, hipMemsetAsync(c->stats, 0, (STATS_STATUS + n_chunks) * sizeof(uint64_t), s));
        }
        HIP_TRY(c, launch_count(c->tri_aos, c->n_tris, c->g, narrow ? c->cnt16 : cnt_of(c), narrow, c->stats + STATS_VOL_SUMS, c->ranges, sat ? c->sat_mask : nullptr, s));
        HIP_TRY(c, hipEventRecord(c->ev[1], s));
        HIP_TRY(c, launch_scan(cnt_of(c), narrow ? c->cnt16 : nullptr, c->stats + STATS_VOL_SUMS, n_chunks, status_aux, c->stats, c->occ_bits, s));
        c->occ_bits_valid = true;
        // With the occupancy bitmap the distance field needs nothing else: it starts now, on its own stream, beside
        // the fill (which waits on atomics, not on bandwidth).  Beside the list-ordering kernel -- where it ran before
        // -- its streaming sweeps pushed that kernel's re-read lists out of L2 (3.1 ms for the two together at 1024^3,
        // against 1.2 + 1.9 apart).  Grids whose x extent the bitmap kernel does not take are forked after the fill.
        if (!c->defer_accel && nx % 4 == 0 && nx <= 16384) if (int r = fork_distance_field(c, s)) return r;
        HIP_TRY(c, hipEventRecord(c->ev[2], s));
        return HVRT_OK;
    };

    static const bool trace_build = getenv("HVRT_TRACE_BUILD") != nullptr;      // diagnostics: where a build is, on stderr
    auto note = [&](const char* what) { if (trace_build) { fprintf(stderr, "[hvrt_build_grid %p] %s\n", (void*)c, what); fflush(stderr); } };
    uint64_t h_stats[8];
    auto read_stats = [&]() -> int {
        HIP_TRY(c, hipMemcpyAsync(c->stats_host, c->stats, sizeof h_stats, hipMemcpyDeviceToHost, s));
        note("waiting for the stream");
        if (trace_build) {
            // which phase a build that does not come back is in: the events recorded behind the phases, after 20 s
            const auto t0 = std::chrono::steady_clock::now();
            while (hipStreamQuery(s) == hipErrorNotReady) {
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) {
                    const int ks[] = { 0, 1, 2, 3, 4, 5, 8 };
                    const char* names[] = { "start", "count", "scan", "(fill queued)", "fill", "order", "derived data" };
                    for (int i = 0; i < 7; ++i)
                        fprintf(stderr, "[hvrt_build_grid %p] after 20 s: event behind '%s' %s\n", (void*)c, names[i],
                                hipEventQuery(c->ev[ks[i]]) == hipSuccess ? "reached" : "NOT reached");
                    fprintf(stderr, "[hvrt_build_grid %p] distance-field stream %s\n", (void*)c,
                            hipStreamQuery(c->df_stream) == hipSuccess ? "idle" : "busy");
                    fflush(stderr);
                    break;
                }
                usleep(500);
            }
            (void)hipGetLastError();
        }
        HIP_TRY(c, hipStreamSynchronize(s));
        memcpy(h_stats, c->stats_host, sizeof h_stats);
        note("stream drained");
        if (h_stats[5]) return fail(c, HVRT_ERR_HIP, "build_grid: the offset scan gave up waiting for a chunk (device fault?)");
        // stats[0] is exact in 64 bits; the offsets are 32-bit sums and only meaningful when it fits
        if (h_stats[0] > 0xFFFFFFFFull)
            return fail(c, HVRT_ERR_OVERFLOW, "build_grid: %llu triangle references exceed the u32 CSR", (unsigned long long)h_stats[0]);
        return HVRT_OK;
    };
    // everything behind the scan: scatter, per-voxel order, derived data.  The scanned counters are untouched until the
    // scatter really runs, so a tail that found its arrays too small can simply be issued again.
    auto tail = [&](const RefCount& rc) -> int {
        uint32_t* unsorted = c->fill_scratch;
        uint32_t* vox_of = c->fill_scratch + rc.cap;
        HIP_TRY(c, hipEventRecord(c->ev[3], s));
        HIP_TRY(c, launch_fill(c->ranges, c->tri_aos, c->n_tris, c->g, cnt_of(c), unsorted, vox_of, rc, sat ? c->sat_mask : nullptr, s));
        HIP_TRY(c, hipEventRecord(c->ev[4], s));
        if (!c->defer_accel && !c->dist_in_flight) if (int r = fork_distance_field(c, s)) return r;     // offsets are final from here on
        HIP_TRY(c, launch_sort(offsets_of(c), unsorted, vox_of, rc, c->indices, c->long_queue, c->long_queue_capacity, c->stats, s));
        HIP_TRY(c, hipEventRecord(c->ev[5], s));
        c->vox_of = vox_of;
        if (!c->defer_accel) if (int r = make_ref_records(c, s, false, rc)) return r;
        HIP_TRY(c, hipEventRecord(c->ev[8], s));
        return HVRT_OK;
    };
    auto size_arrays = [&](uint64_t total) -> int {
        // an eighth more than needed: the next build of a scene that moved a little then fits and need not wait
        const uint64_t want = total + total / 8;
        const uint64_t cap = want > 0xFFFFFFFFull ? 0xFFFFFFFFull : want;
        if (int r = ensure<uint32_t>(c, c->indices, c->indices_capacity, cap)) return r;
        if (int r = ensure<uint32_t>(c, c->long_queue, c->long_queue_capacity, cap / 64 + 2)) return r;
        if (int r = ensure<uint32_t>(c, c->fill_scratch, c->fill_scratch_capacity, 2 * cap)) return r;
        return HVRT_OK;
    };

    // Arrays left by an earlier build: go on without the round trip to the host (the kernels read the count where the
    // scan left it).  Otherwise wait for the count and size them.
    uint64_t total = 0;
    for (;;) {
        const bool narrow = c->narrow_counters && !sat;       // (the 16-bit counters' cross-check is the sum of the range volumes)
        (void)join_distance_field(c, s);        // (second round only: the first one's field is being recomputed)
        note("queueing count + scan");
        if (int r = front(narrow)) return r;
        note("count + scan queued");
        const uint64_t live_cap = c->reuse_arrays ? ref_capacity_of(c, !c->defer_accel) : 0;
        bool went_ahead = live_cap > 0;
        if (went_ahead) {
            const RefCount rc = { c->stats, 0, live_cap };
            if (int r = tail(rc)) return r;
            note("scatter, order and derived data queued");
        }
        if (int r = read_stats()) return r;
        if (narrow && h_stats[0] != h_stats[7]) {      // a 16-bit counter carried: nothing was scattered, count again
            c->narrow_counters = false;
            continue;
        }
        total = h_stats[0];
        if (went_ahead && total <= live_cap) break;
        // no arrays yet, or too small (then nothing was scattered): size them and issue the tail (again)
        if (int r = size_arrays(total)) return r;
        const RefCount rc = { nullptr, total, c->fill_scratch_capacity / 2 };
        if (int r = tail(rc)) return r;
        HIP_TRY(c, hipStreamSynchronize(s));
        break;
    }

    c->n_refs = total;
    c->face_words = face_words_for(total);
    c->n_occupied = h_stats[1];
    c->max_list = (uint32_t)h_stats[2];
    c->has_grid = true;
    c->accel_pending = c->defer_accel;
    c->accel_built_here = true;
    if (!c->accel_pending) c->vox_of = nullptr;
    c->tm.build_count_ms = ms_between(c->ev[0], c->ev[1]);
    c->tm.build_scan_ms = ms_between(c->ev[1], c->ev[2]);
    c->tm.build_fill_ms = ms_between(c->ev[3], c->ev[4]);
    c->tm.build_sort_ms = ms_between(c->ev[4], c->ev[5]);
    c->tm.build_expand_ms = c->accel_pending ? 0.0f : ms_between(c->ev[5], c->ev[8]);
    c->tm.build_total_ms = ms_between(c->ev[0], c->ev[8]);
    return HVRT_OK;
}

static int adopt_csr(hvrt_ctx* c, int nx, int ny, int nz, const float mn[3], const float mx[3], const float vs[3],
                     const void* offsets, const void* indices, uint64_t n_refs, hipMemcpyKind kind) {
    if (int r = bind(c)) return r;
    if (int r = check_grid_args(c, nx, ny, nz, mn, mx, vs)) return r;
    if (!offsets || (n_refs && !indices)) return fail(c, HVRT_ERR_INVALID, "set_grid_csr: null array");
    if (n_refs > 0xFFFFFFFFull) return fail(c, HVRT_ERR_OVERFLOW, "set_grid_csr: too many references");
    c->has_grid = false;
    c->accel_pending = false;
    c->occ_bits_valid = false;      // the bitmap belongs to the offsets of the last build
    c->cost_valid = false;
    (void)join_distance_field(c, c->stream);   // left over from a build that failed half-way
    set_grid_params(c, nx, ny, nz, mn, mx, vs);
    const uint64_t nvox = c->g.nvox, npad = padded_vox(nvox);
    if (int r = ensure<uint32_t>(c, c->cnt_base, c->cnt_capacity, npad + 8)) return r;
    if (int r = ensure<uint32_t>(c, c->indices, c->indices_capacity, n_refs)) return r;
    hipStream_t s = c->stream;
    HIP_TRY(c, hipMemcpyAsync(offsets_of(c), offsets, (nvox + 1) * 4, kind, s));
    if (n_refs) HIP_TRY(c, hipMemcpyAsync(c->indices, indices, n_refs * 4, kind, s));
    HIP_TRY(c, hipMemsetAsync(c->stats, 0, 8 * sizeof(uint64_t), s));
    HIP_TRY(c, launch_csr_stats(offsets_of(c), nvox, c->indices, n_refs, c->n_tris, c->stats, s));
    uint64_t h_stats[8];
    HIP_TRY(c, hipMemcpyAsync(h_stats, c->stats, sizeof h_stats, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    // the caller's arrays are checked before anything is derived from them, whatever the record mode
    if (h_stats[6])
        return fail(c, HVRT_ERR_INVALID, "grid: the CSR offsets are malformed (%llu violations of: offsets[0] == 0, non-decreasing, "
                    "offsets[n_voxels] == n_refs = %llu)", (unsigned long long)h_stats[6], (unsigned long long)n_refs);
    if (h_stats[4])
        return fail(c, HVRT_ERR_INVALID, "grid: %llu CSR entries reference a triangle index >= %u",
                    (unsigned long long)h_stats[4], c->n_tris);
    c->n_refs = n_refs;
    c->face_words = face_words_for(n_refs);
    c->n_occupied = h_stats[1];
    c->max_list = (uint32_t)h_stats[2];
    c->accel_pending = c->defer_accel;
    c->accel_built_here = false;
    c->lists_follow_ranges = false;
    c->vox_of = nullptr;
    if (!c->defer_accel) {
        const RefCount rc = { nullptr, n_refs, n_refs };
        if (int r = make_ref_records(c, s, true, rc)) return r;
    }
    c->has_grid = true;
    return HVRT_OK;
}

int hvrt_set_grid_csr(hvrt_ctx* c, int nx, int ny, int nz, const float mn[3], const float mx[3], const float vs[3],
                      const uint32_t* offsets, const uint32_t* indices) {
    if (!c) return HVRT_ERR_INVALID;
    if (!offsets) return fail(c, HVRT_ERR_INVALID, "set_grid_csr: null offsets");
    if (nx < 1 || ny < 1 || nz < 1) return fail(c, HVRT_ERR_INVALID, "grid: dimensions must be >= 1");
    const uint64_t nvox = (uint64_t)nx * ny * nz;
    return adopt_csr(c, nx, ny, nz, mn, mx, vs, offsets, indices, offsets[nvox], hipMemcpyHostToDevice);
}

int hvrt_set_grid_csr_device(hvrt_ctx* c, int nx, int ny, int nz, const float mn[3], const float mx[3],
                             const float vs[3], const void* d_offsets, const void* d_indices, uint64_t n_refs) {
    if (!c) return HVRT_ERR_INVALID;
    return adopt_csr(c, nx, ny, nz, mn, mx, vs, d_offsets, d_indices, n_refs, hipMemcpyDeviceToDevice);
}

int hvrt_get_grid_info(const hvrt_ctx* c, hvrt_grid_info* out) {
    if (!c || !out) return HVRT_ERR_INVALID;
    if (!c->has_grid) return fail(const_cast<hvrt_ctx*>(c), HVRT_ERR_INVALID, "no grid");
    out->nx = c->g.nx; out->ny = c->g.ny; out->nz = c->g.nz;
    out->min_bound[0] = c->g.mnx; out->min_bound[1] = c->g.mny; out->min_bound[2] = c->g.mnz;
    memcpy(out->max_bound, c->max_bound, 12);
    out->voxel_size[0] = c->g.vsx; out->voxel_size[1] = c->g.vsy; out->voxel_size[2] = c->g.vsz;
    out->n_voxels = c->g.nvox;
    out->n_refs = c->n_refs;
    out->n_occupied = c->n_occupied;
    out->max_list = c->max_list;
    out->n_triangles = c->n_tris;
    return HVRT_OK;
}

int hvrt_get_csr(hvrt_ctx* c, uint32_t* offsets, uint32_t* indices) {
    if (!c) return HVRT_ERR_INVALID;
    if (!c->has_grid) return fail(c, HVRT_ERR_INVALID, "get_csr: no grid");
    if (int r = bind(c)) return r;
    if (offsets) if (int r = download(c, offsets, offsets_of(c), (c->g.nvox + 1) * 4, c->stream)) return r;
    if (indices && c->n_refs) if (int r = download(c, indices, c->indices, c->n_refs * 4, c->stream)) return r;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return HVRT_OK;
}

int hvrt_get_distance_field(hvrt_ctx* c, uint8_t* dist) {
    if (!c) return HVRT_ERR_INVALID;
    if (int r = bind(c)) return r;
    if (!c->has_grid) return fail(c, HVRT_ERR_INVALID, "get_distance_field: no grid");
    if (!dist) return fail(c, HVRT_ERR_INVALID, "get_distance_field: null output");
    if (int r = finish_accel(c)) return r;
    if (int r = download(c, dist, c->dist, c->g.nvox, c->stream)) return r;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return HVRT_OK;
}

int hvrt_get_device_arrays(hvrt_ctx* c, void** d_tri_aos, void** d_offsets, void** d_indices) {
    if (!c) return HVRT_ERR_INVALID;
    if (d_tri_aos) *d_tri_aos = c->tri_aos;
    if (d_offsets) *d_offsets = c->has_grid ? (void*)offsets_of(c) : nullptr;
    if (d_indices) *d_indices = c->has_grid ? (void*)c->indices : nullptr;
    return HVRT_OK;
}

int hvrt_export_scene_device(hvrt_ctx* c, void* d_tri_aos, void* d_offsets, void* d_indices) {
    if (!c) return HVRT_ERR_INVALID;
    if (int r = bind(c)) return r;
    if (d_tri_aos && c->n_tris)
        HIP_TRY(c, hipMemcpyAsync(d_tri_aos, c->tri_aos, (size_t)c->n_tris * 48, hipMemcpyDeviceToDevice, c->stream));
    if ((d_offsets || d_indices) && !c->has_grid) return fail(c, HVRT_ERR_INVALID, "export_scene: no grid");
    if (d_offsets)
        HIP_TRY(c, hipMemcpyAsync(d_offsets, offsets_of(c), (c->g.nvox + 1) * 4, hipMemcpyDeviceToDevice, c->stream));
    if (d_indices && c->n_refs)
        HIP_TRY(c, hipMemcpyAsync(d_indices, c->indices, c->n_refs * 4, hipMemcpyDeviceToDevice, c->stream));
    return HVRT_OK;
}

// ---- rendering ---------------------------------------------------------------------------------

void hvrt_make_camera_frame(const float pos[3], const float look_at[3], const float up_hint[3], float fov,
                            float aspect, hvrt_camera_frame* out) {
    // OctreeLikeRayTracer.cpp:342-359; this file is built with -ffp-contract=off
    auto norm = [](float* v) {
        float len = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        if (len > 0.0f) { v[0] = v[0] / len; v[1] = v[1] / len; v[2] = v[2] / len; }
    };
    auto crs = [](const float* a, const float* b, float* r) {
        r[0] = a[1] * b[2] - a[2] * b[1];
        r[1] = a[2] * b[0] - a[0] * b[2];
        r[2] = a[0] * b[1] - a[1] * b[0];
    };
    float fov_rad = fov * 3.14159265f / 180.0f;
    float half_h = tanf(fov_rad / 2.0f);
    float half_w = aspect * half_h;
    float fwd[3] = { look_at[0] - pos[0], look_at[1] - pos[1], look_at[2] - pos[2] };
    norm(fwd);
    float right[3];
    crs(fwd, up_hint, right);
    norm(right);
    float up[3];
    crs(right, fwd, up);
    for (int k = 0; k < 3; ++k) {
        out->origin[k] = pos[k];
        out->horizontal[k] = right[k] * (2.0f * half_w);
        out->vertical[k] = up[k] * (2.0f * half_h);
    }
    for (int k = 0; k < 3; ++k)
        out->lower_left[k] = ((pos[k] - out->horizontal[k] * 0.5f) - out->vertical[k] * 0.5f) + fwd[k];
}

int hvrt_part_rows(int height, int tile_rows, int part, int n_parts) {
    if (height <= 0 || tile_rows <= 0 || n_parts <= 0 || part < 0 || part >= n_parts) return 0;
    const int n_tiles = (height + tile_rows - 1) / tile_rows;
    int rows = 0;
    for (int k = part; k < n_tiles; k += n_parts) {
        int r = height - k * tile_rows;
        rows += r < tile_rows ? r : tile_rows;
    }
    return rows;
}

// A frame rendered as n launches over consecutive row ranges of one ordered queue (hvrt_render's way of overlapping the
// PCIe copy with the traversal): tile rows [row[b], row[b + 1]) are complete when ev_seg[b] has happened.  Filled in
// by render_batch_part when it took that route (used); otherwise the frame is one launch as always.
struct SegPlan { int n = 0; uint32_t row[9] = {}; bool used = false; };

// How far apart two views are, as an angle: the larger of the angle between their central rays and the angle under
// which the move of the eye is seen from the grid's centre.  (NaNs compare false: such a view counts as near.)
static float view_distance(const hvrt_ctx* c, const hvrt_camera_frame& a, const hvrt_camera_frame& b) {
    const GridParams g = scene_of(c).g;
    double da[3], db[3], la = 0, lb = 0, dot = 0, move = 0, dist = 0;
    const double centre[3] = { 0.5 * ((double)g.mnx + g.bxx), 0.5 * ((double)g.mny + g.bxy), 0.5 * ((double)g.mnz + g.bxz) };
    for (int k = 0; k < 3; ++k) {
        da[k] = (double)a.lower_left[k] + 0.5 * a.horizontal[k] + 0.5 * a.vertical[k] - a.origin[k];
        db[k] = (double)b.lower_left[k] + 0.5 * b.horizontal[k] + 0.5 * b.vertical[k] - b.origin[k];
        la += da[k] * da[k]; lb += db[k] * db[k]; dot += da[k] * db[k];
        move += ((double)a.origin[k] - b.origin[k]) * ((double)a.origin[k] - b.origin[k]);
        dist += ((double)a.origin[k] - centre[k]) * ((double)a.origin[k] - centre[k]);
    }
    double turn = 0.0;
    if (la > 0 && lb > 0) { double cs = dot / sqrt(la * lb); cs = cs > 1.0 ? 1.0 : cs < -1.0 ? -1.0 : cs; turn = acos(cs); }
    const double seen = dist > 0 ? atan(sqrt(move / dist)) : (move > 0 ? 1.5707963 : 0.0);
    return (float)(turn > seen ? turn : seen);
}

// hvrt_render_batch_part_device plus d_hits (nullable): one u64 that receives the number of pixels whose sample-0 ray
// hit a triangle (the host-facing entry points keep it for hvrt_last_hit_count)
static int render_batch_part(hvrt_ctx* c, const hvrt_camera_frame* frames, int n_frames, int width, int height,
                             int spp, uint32_t seed, int tile_rows, int part, int n_parts, int slab_rows,
                             void* d_rgb, void* d_hit_id, void* d_t, void* d_counters, void* d_hits, SegPlan* plan = nullptr) {
    if (!c) return HVRT_ERR_INVALID;
    if (int r = bind(c)) return r;
    if (int r = require_scene(c)) return r;
    if (!frames) return fail(c, HVRT_ERR_INVALID, "render: null camera frame");
    if (n_frames < 1 || n_frames > HVRT_MAX_BATCH_FRAMES)
        return fail(c, HVRT_ERR_INVALID, "render: n_frames %d outside 1..%d", n_frames, HVRT_MAX_BATCH_FRAMES);
    if (width < 1 || height < 1 || width > 0x7FFFFFF0 || height > 0x7FFFFFF0 || (uint64_t)width * height > 0x7FFFFFFFull)
        return fail(c, HVRT_ERR_INVALID, "render: bad image size %d x %d", width, height);
    if (spp < 1) return fail(c, HVRT_ERR_INVALID, "render: spp must be >= 1");
    if (tile_rows < 1 || n_parts < 1 || part < 0 || part >= n_parts)
        return fail(c, HVRT_ERR_INVALID, "render: bad partition (tile_rows %d, part %d of %d)", tile_rows, part, n_parts);
    const int local_rows = hvrt_part_rows(height, tile_rows, part, n_parts);
    if (slab_rows < local_rows)
        return fail(c, HVRT_ERR_INVALID, "render: slab_rows %d < the %d rows of part %d", slab_rows, local_rows, part);
    const uint64_t n_tiles = render_tile_count(width, local_rows, n_frames);
    if (n_tiles > 0x7FFFFFFFull) return fail(c, HVRT_ERR_INVALID, "render: too many tiles in one launch");
    hipStream_t s = c->stream;

    // camera frames: a single one travels inside the kernel arguments; a batch goes to the device through pinned
    // staging (so the copy is asynchronous; a slot is reused only after the copy that last used it has run)
    hvrt_camera_frame* ds = nullptr;
    if (n_frames > 1) {
        const int slot = c->cam_next;
        c->cam_next = (c->cam_next + 1) % hvrt_ctx::CAM_SLOTS;
        if (c->cam_ev_used[slot]) HIP_TRY(c, hipEventSynchronize(c->cam_ev[slot]));
        hvrt_camera_frame* hs = c->cam_host + (size_t)slot * hvrt_ctx::CAM_MAX;
        ds = c->cam_dev + (size_t)slot * hvrt_ctx::CAM_MAX;
        memcpy(hs, frames, sizeof(hvrt_camera_frame) * (size_t)n_frames);
        HIP_TRY(c, hipMemcpyAsync(ds, hs, sizeof(hvrt_camera_frame) * (size_t)n_frames, hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipEventRecord(c->cam_ev[slot], s));
        c->cam_ev_used[slot] = true;
    }

    // tile order: longest first, from the previous launch of the same shape
    const int key[8] = { width, height, tile_rows, part, n_parts, n_frames, spp, 1 };
    const bool grew = c->tile_cost_capacity < n_tiles || c->tile_order_capacity < n_tiles;
    const uint64_t order_blocks = (n_tiles + 255) / 256;
    if (int r = ensure<uint32_t>(c, c->tile_cost, c->tile_cost_capacity, n_tiles)) return r;
    if (int r = ensure<uint32_t>(c, c->tile_order, c->tile_order_capacity, n_tiles)) return r;
    if (int r = ensure<uint32_t>(c, c->split_tiles, c->split_tiles_capacity, n_tiles)) return r;
    if (int r = ensure<uint32_t>(c, c->abandoned, c->abandoned_capacity, n_tiles)) return r;
    if (int r = ensure<uint32_t>(c, c->order_blocks, c->order_blocks_capacity, (uint64_t)ORDER_ARRAYS * (order_blocks + 1))) return r;
    // ... and of a view near this one: a queue ordered by the durations of a view a few degrees away is worse than an
    // unordered one (config 2, an orbit in steps of 2 / 3 / 5 / 7 degrees: 1.55 / 1.71 / 1.89 / 2.09 ms per frame with the
    // previous frame's durations, 1.69-1.75 with none; scripts/time_moving.py), so past 2.5 degrees this is a first launch
    const bool view_moved = c->history_max_rad > 0.0f && c->cost_valid && view_distance(c, frames[0], c->cost_cam) > c->history_max_rad;
    // A launch that gave all its tiles up at once because they were few (RenderArgs::few_tiles) leaves durations that are sums
    // over the split kernel's pieces: they rank the tiles poorly (config 2 at 800x600: 0.74 ms for the launch ordered by them,
    // 0.42 for one that gives everything up again).  If that launch has run by now and said so, this one is another of its kind.
    bool last_was_few = false;
    if (c->few_serial != 0u && __atomic_load_n(c->split_hint + 3, __ATOMIC_ACQUIRE) == c->few_serial)
        last_was_few = __atomic_load_n(c->split_hint + 2, __ATOMIC_RELAXED) != 0u;
    c->few_serial = 0u;
    const bool reuse = c->adaptive_order && c->cost_valid && c->order_ahead && !grew && !view_moved && !last_was_few && memcmp(key, c->cost_key, sizeof key) == 0;
    c->order_ahead = false;

    RenderArgs ra;
    ra.frames = ds; ra.frame0 = frames[0]; ra.n_frames = n_frames;
    ra.width = width; ra.height = height; ra.spp = spp; ra.seed = seed;
    ra.width_f = (float)width; ra.height_f = (float)height; ra.spp_f = (float)spp;
    ra.nearest = c->mode;
    ra.tile_rows = tile_rows; ra.part = part; ra.n_parts = n_parts;
    ra.local_rows = local_rows; ra.slab_rows = slab_rows;
    ra.rgb = (float*)d_rgb; ra.hit_id = (uint32_t*)d_hit_id; ra.t = (float*)d_t;
    ra.counters = (unsigned long long*)d_counters;
    ra.hits = (unsigned long long*)d_hits;
    ra.tile_counter = c->tile_counter;
    ra.n_heads = 8;
    ra.order = reuse ? c->tile_order : nullptr;
    ra.split_tiles = c->split_tiles;
    ra.n_split = c->order_blocks + order_blocks;
    ra.n_front = c->tile_counter + 49;
    ra.n_queue = nullptr;
    ra.split_started = nullptr;
    ra.abandoned = c->abandoned;
    ra.n_abandoned = c->tile_counter + 1;
    // (the give-up time grows with the grid: a ray crosses nx + ny + nz voxels at most, and what is "long" for a tile goes with
    // that -- HVRT_TILE_ABORT_US is the time for grids up to 256^3.  A tile given up on is split from then on, so the time is
    // also the bar of the launches that follow: config 3, 512^3, with 500 us for all grids: 1.66 ms without history but 0.92-0.97
    // with, against 1.86 and 0.87)
    {
        const GridParams& g = scene_of(c).g;
        const long long cells = (long long)g.nx + g.ny + g.nz;
        ra.abort_ticks = cells > 768 ? c->abort_ticks * cells / 768 : c->abort_ticks;
        // (Since k_render_split's walk looks ahead, a tile is better off there sooner: launches without history at 500 / 420 /
        // 350 / 300 / 250 / 200 us: config 2 0.82 / 0.74 / 0.68 / 0.63 / 0.60 / 0.78 ms, config 3 1.76 / 1.60 / 1.51 / 1.41 / 1.48 /
        // 1.65, config 4 10.3 / 10.3 / 10.3 / 11.4 / 11.5 / 11.7; with history 250 us costs configs 2 and 3 some 4 %, 300 us
        // nothing.  Supersampled launches, whose long tiles are rendered sample by sample from the second launch on, keep a
        // third more: config 4's cliff lies between 350 and 300.)
        if (spp > 1) ra.abort_ticks += ra.abort_ticks / 3;
    }
    // (the split kernel cuts a tile into sixteen pieces where the grid's lists are long, into eight otherwise: half the tiles
    // make as many pieces -- config 3 at 1080p, some 1 500 marked tiles: 0.9 ms given up at once, 0.75 with history)
    ra.few_tiles = c->max_list > 256u ? c->few_tiles / 2u : c->few_tiles; ra.few_report = nullptr; ra.few_serial = 0u;
    const long long abort_with_cover = ra.abort_ticks;
    ra.abort_ticks *= 2;      // (every tile walked, the GPU full from end to end: twice the time; halved below where a tile cover is in use)
    ra.tile_cost = c->tile_cost;
    ra.cost_sum = c->cost_sum + c->cost_parity;
    ra.walked = ra.cost_sum + 2;
    ra.seg_slots = nullptr; ra.seg = 0;
    ra.cover = nullptr; ra.cover_words_x = 0; ra.cover_frame_words = 0;
    ra.sample_rgb = nullptr; ra.sample_done = nullptr; ra.sample_cap = 0;
    { const char* v = getenv("HVRT_DEBUG_TILE_START"); ra.debug_start = v && atoi(v) != 0; }
    CoverArgs ca;
    const bool use_cover = c->tile_cover && !d_counters && n_tiles != 0 && scene_of(c).n_tris != 0 &&
                           cover_geometry(width, height, n_frames, scene_of(c).n_tris, ca) == 0;
    bool cover_stands = false, fill_beside = false, split_behind_cover = false;
    RenderArgs ra_fill;
    if (use_cover) {
        ra.abort_ticks = abort_with_cover;
        const uint32_t* const before = c->cover_bits;
        if (int r = ensure<uint32_t>(c, c->cover_bits, c->cover_capacity, cover_words(ca))) return r;
        if (int r = ensure<uint32_t>(c, c->walk_queue, c->walk_queue_capacity, n_tiles)) return r;
        if (int r = ensure<uint32_t>(c, c->compact_counts, c->compact_counts_capacity, 2 * order_blocks + 1)) return r;
        const CoverArgs& l = c->cover_last;
        cover_stands = c->cover_cache && before == c->cover_bits && l.n_frames == n_frames && l.width == width && l.height == height &&
                       memcmp(c->cover_frames, frames, sizeof(hvrt_camera_frame) * (size_t)n_frames) == 0;
    }
    next_render_events(c);
    HIP_TRY(c, hipEventRecord(c->ev[6], s));
    if (use_cover) {
        // (inside the timed region: render_ms is the pre-pass and the traversal)
        ca.frames = ds; ca.frame0 = frames[0];
        cover_bind(ca, c->cover_bits);
        if (!cover_stands) {
            HIP_TRY(c, hipMemsetAsync(ca.bits, 0, cover_zero_words(ca) * sizeof(uint32_t), s));
            HIP_TRY(c, launch_cover(scene_of(c), ca, s));
            memcpy(c->cover_frames, frames, sizeof(hvrt_camera_frame) * (size_t)n_frames);
        }
        ra.cover = ca.bits; ra.cover_words_x = ca.words_x; ra.cover_frame_words = ca.frame_words;
        c->cover_last = ca;
        // The tiles that are not marked get their pixels here; k_render's queue becomes the marked tiles alone (every slot a
        // tile to walk, handed out one at a time like before the cover) -- except for a frame in several launches, whose
        // slot ranges are cut from the full order: there k_render passes the unmarked tiles by as it meets them.
        const bool in_segments = reuse && plan && plan->n > 1 && plan->n <= 8 && n_frames == 1;
        uint32_t* const counters = c->tile_counter + 50;      // two words of the queue scratch, zero like the queue's heads
        const RenderArgs ra_fill_now = ra;
        if (!in_segments) {
            HIP_TRY(c, launch_compact_queue(ra, ra.order, reuse ? ra.n_front : nullptr, c->compact_counts, c->walk_queue, counters, s));
            ra.order = c->walk_queue; ra.n_queue = counters; ra.n_front = counters + 1;
            ra.cover = nullptr;
        }
        // (the fill writes other pixels than the traversal: it runs beside it, on the stream builds use for the distance field.
        // It and k_render_split start here, behind the cover pass and the queue's two small kernels, which the traversal waits
        // for: beside those, the two took the wave slots -- the cover pass of an orbit's frame then lasted 290 us instead of
        // 120, the queue's kernels 100 instead of 12)
        ra_fill = ra_fill_now;
        fill_beside = true;
        split_behind_cover = true;
        HIP_TRY(c, hipEventRecord(c->ev_split_fork, s));
    } else {
        c->cover_last.n_frames = 0;
    }
    // (the fill, on its own stream, from the point where this is called: behind k_wait_split_started, or that kernel waits
    // out its 100 us for workgroups that find the fill's waves in their places)
    auto start_fill = [&]() -> int {
        if (!fill_beside) return 0;
        HIP_TRY(c, hipEventRecord(c->ev_fill_fork, s));
        HIP_TRY(c, hipStreamWaitEvent(c->df_stream, c->ev_fill_fork, 0));
        HIP_TRY(c, launch_fill_background(ra_fill, c->n_cus, c->df_stream));
        HIP_TRY(c, hipEventRecord(c->ev_fill_join, c->df_stream));
        return 0;
    };
    const bool queued_behind_ev6 = !c->queue_clean;
    if (!c->queue_clean) {
        // [0] a single head, [1] abandoned count, [8..15] the heads of segment launches; the queue's eight heads; the two sums
        HIP_TRY(c, launch_queue_reset(c->tile_counter, ra.cost_sum, ra.walked, s));
    }
    c->queue_clean = false;
    if (reuse && n_tiles) {
        const bool may_split = c->order_may_split;
        // supersampled: the over-long tiles stay in k_render, one item per sample (not for a frame in several launches,
        // whose launches share the words behind the queue head)
        bool by_sample = may_split && c->sample_split && spp >= 2 && !d_counters && !(plan && plan->n > 1 && n_frames == 1);
        if (by_sample) {
            uint64_t cap = (64ull << 20) / ((uint64_t)spp * 64u * sizeof(float4));
            cap = cap > 4096u ? 4096u : cap;
            if (cap == 0) {
                by_sample = false;
            } else {
                const uint32_t* const before = c->sample_done;
                if (int r = ensure<float4>(c, c->sample_rgb, c->sample_rgb_capacity, cap * (uint64_t)spp * 64u)) return r;
                if (int r = ensure<uint32_t>(c, c->sample_done, c->sample_done_capacity, 4096u)) return r;
                if (before != c->sample_done) HIP_TRY(c, hipMemsetAsync(c->sample_done, 0, 4096u * sizeof(uint32_t), s));
                ra.sample_rgb = c->sample_rgb; ra.sample_done = c->sample_done; ra.sample_cap = (uint32_t)cap;
            }
        }
        if (may_split && !by_sample) {
            // the tiles too long for one wave go first and beside the others, on the second stream
            // (the fork is the launch's own first timing event when nothing was queued behind it: one packet fewer in the stream)
            hipEvent_t fork = c->ev[6];
            if (split_behind_cover) fork = c->ev_split_fork;
            else if (queued_behind_ev6) { fork = c->ev_split_fork; HIP_TRY(c, hipEventRecord(fork, s)); }
            HIP_TRY(c, hipStreamWaitEvent(c->aux_stream, fork, 0));
            RenderArgs rsplit = ra;
            rsplit.split_started = c->wait_for_split ? c->tile_counter + 48 : nullptr;      // a word of the 64-word scratch, never reset
            uint32_t nb = 0;
            // how many tiles that order build found for the split kernel sizes it -- if the build has run by now and left its
            // number; otherwise the count is not known and the kernel is sized for any (a stale small count once gave 563
            // tiles 16 workgroups: 11 ms for a 0.6 ms launch)
            const uint32_t serial_seen = __atomic_load_n(c->split_hint + 1, __ATOMIC_ACQUIRE);
            const uint32_t expected = serial_seen == c->order_serial ? __atomic_load_n(c->split_hint, __ATOMIC_RELAXED) : 0u;
            // (eight rays per wave unless the grid holds lists of hundreds of entries, see k_render_split)
            HIP_TRY(c, launch_render_split(scene_of(c), rsplit, c->n_cus, c->aux_stream, &nb, expected, false, c->max_list > 256u ? 4 : 8));
            HIP_TRY(c, hipEventRecord(c->ev_split_join, c->aux_stream));
            if (rsplit.split_started && nb) {
                // k_render's persistent waves would otherwise take every slot before the split kernel is on (see k_wait_split_started)
                c->split_started_total += nb;
                HIP_TRY(c, launch_wait_split_started(rsplit.split_started, c->split_started_total, 100u, s));
            }
        }
        if (int r = start_fill()) return r;
        if (plan && plan->n > 1 && plan->n <= 8 && n_frames == 1) {
            // several launches over consecutive row ranges of this one queue; the long tiles (the queue's head) and the
            // tiles of k_render_split belong to the first
            uint32_t* seg_slots = c->tile_counter + 32;      // 9 words of the 64-word scratch
            HIP_TRY(c, launch_order_segments(c->tile_order, c->order_blocks, (uint32_t)order_blocks, (uint32_t)n_tiles,
                                             (uint32_t)(width + 7) >> 3, plan->row, plan->n, seg_slots, s));
            for (int b = 0; b < plan->n; ++b) {
                RenderArgs rs = ra;
                rs.seg_slots = seg_slots; rs.seg = b;
                rs.tile_counter = c->tile_counter + 8 + b;
                rs.n_heads = 1;
                HIP_TRY(c, launch_render(scene_of(c), rs, c->n_cus, false, s));
                if (b == 0 && may_split && !by_sample) HIP_TRY(c, hipStreamWaitEvent(s, c->ev_split_join, 0));
                HIP_TRY(c, hipEventRecord(c->ev_seg[b], s));
            }
            plan->used = true;
        } else {
            HIP_TRY(c, launch_render(scene_of(c), ra, c->n_cus, false, s));
            if (may_split && !by_sample) HIP_TRY(c, hipStreamWaitEvent(s, c->ev_split_join, 0));
        }
    } else {
        // no history: the kernel may give up on over-long tiles; k_render_split renders those behind it (none,
        // as a rule: it then finds an empty list and ends)
        const bool abortable = c->adaptive_split && n_tiles != 0 && c->abort_ticks > 0;    // HVRT_TILE_ABORT_US=0: never give up
        if (int r = start_fill()) return r;
        if (abortable && ra.n_queue && ra.few_tiles) {
            if (++c->few_counter == 0u) c->few_counter = 1u;
            c->few_serial = c->few_counter;
            ra.few_report = c->split_hint_dev + 2; ra.few_serial = c->few_serial;
        }
        HIP_TRY(c, launch_render(scene_of(c), ra, c->n_cus, abortable, s));
        if (abortable) {
            RenderArgs rb = ra;
            rb.order = c->tile_order;           // any non-null value: "there is a list"
            rb.split_tiles = c->abandoned;
            rb.n_split = ra.n_abandoned;
            // (eight rays per wave where the grid's lists are short, as beside k_render: since the split kernel's walk looks ahead, the
            // sixteen lanes per ray no longer pay behind a launch either -- config 2 at 800x600, every tile given up for being few:
            // 0.48 -> 0.43 ms, at 4K 0.83 either way)
            HIP_TRY(c, launch_render_split(scene_of(c), rb, c->n_cus, s, nullptr, 0u, true, c->max_list > 256u ? 4 : 8));
        }
    }
    if (fill_beside) HIP_TRY(c, hipStreamWaitEvent(s, c->ev_fill_join, 0));
    HIP_TRY(c, hipEventRecord(c->ev[7], s));       // render_ms: the traversal (k_render, and k_render_split beside or behind it)
    // behind the traversal: the order for the next launch of this shape from the durations this one records, and the
    // clean queue head / cost sum that launch starts from
    if (c->adaptive_order && n_tiles && !ra.debug_start) {      // (the diagnostic's start times are no durations: no order from them)
        const bool may_split = c->adaptive_split && *(volatile uint32_t*)c->split_hint != 0u;
        // (its last kernel also clears the queue heads and the next launch's cost sum: two memsets fewer in the stream)
        HIP_TRY(c, launch_order_build(c->tile_cost, (uint32_t)n_tiles, ra.cost_sum, c->n_cus, may_split, c->order_blocks,
                                      c->tile_order, c->split_tiles, c->split_hint_dev, ++c->order_serial, c->tile_counter,
                                      c->cost_sum + (c->cost_parity ^ 1), c->sample_split && spp >= 2 && !d_counters, s));
        c->order_ahead = true;
        c->order_may_split = may_split;
        c->cost_parity ^= 1;
    } else {
        c->cost_parity ^= 1;
        HIP_TRY(c, hipMemsetAsync(c->tile_counter, 0, 16 * sizeof(uint32_t), s));
        HIP_TRY(c, hipMemsetAsync(c->tile_counter + 64, 0, 8 * 32 * sizeof(uint32_t), s));
        HIP_TRY(c, hipMemsetAsync(c->cost_sum + c->cost_parity, 0, sizeof(unsigned long long), s));
        HIP_TRY(c, hipMemsetAsync(c->cost_sum + c->cost_parity + 2, 0, sizeof(unsigned long long), s));
        HIP_TRY(c, hipMemsetAsync(c->tile_counter + 50, 0, 2 * sizeof(uint32_t), s));
    }
    c->queue_clean = true;
    memcpy(c->cost_key, key, sizeof key);
    c->cost_cam = frames[0];
    c->cost_valid = true;
    c->render_pending = true;
    return HVRT_OK;
}

int hvrt_render_batch_part_device(hvrt_ctx* c, const hvrt_camera_frame* frames, int n_frames, int width, int height,
                                  int spp, uint32_t seed, int tile_rows, int part, int n_parts, int slab_rows,
                                  void* d_rgb, void* d_hit_id, void* d_t, void* d_counters) {
    return render_batch_part(c, frames, n_frames, width, height, spp, seed, tile_rows, part, n_parts, slab_rows,
                             d_rgb, d_hit_id, d_t, d_counters, nullptr);
}

int hvrt_last_hit_count(const hvrt_ctx* c, uint64_t* out) {
    if (!c || !out) return HVRT_ERR_INVALID;
    *out = c->last_hits;
    return HVRT_OK;
}

int hvrt_render_part_device(hvrt_ctx* c, const hvrt_camera_frame* frame, int width, int height, int spp,
                            uint32_t seed, int tile_rows, int part, int n_parts, void* d_rgb, void* d_hit_id,
                            void* d_t, void* d_counters) {
    const int rows = hvrt_part_rows(height, tile_rows, part, n_parts);
    return hvrt_render_batch_part_device(c, frame, 1, width, height, spp, seed, tile_rows, part, n_parts, rows,
                                         d_rgb, d_hit_id, d_t, d_counters);
}

int hvrt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int hvrt_download(hvrt_ctx* c, void* dst, const void* d_src, uint64_t bytes) {
    if (!c) return HVRT_ERR_INVALID;
    if (int r = bind(c)) return r;
    if (bytes && (!dst || !d_src)) return fail(c, HVRT_ERR_INVALID, "download: null pointer");
    if (int r = download(c, dst, d_src, bytes, c->stream)) return r;
    return hvrt_synchronize(c);
}

int hvrt_download_blocks(hvrt_ctx* c, void* dst, const void* d_src, uint64_t bytes, uint64_t block_bytes, uint64_t dst_stride) {
    if (!c) return HVRT_ERR_INVALID;
    if (int r = bind(c)) return r;
    if (bytes && (!dst || !d_src)) return fail(c, HVRT_ERR_INVALID, "download_blocks: null pointer");
    if (block_bytes == 0 || (block_bytes < bytes && dst_stride < block_bytes))
        return fail(c, HVRT_ERR_INVALID, "download_blocks: blocks of %llu bytes %llu bytes apart overlap", (unsigned long long)block_bytes, (unsigned long long)dst_stride);
    if (int r = download(c, dst, d_src, bytes, c->stream, block_bytes, dst_stride)) return r;
    return hvrt_synchronize(c);
}

int hvrt_get_tile_cover(hvrt_ctx* c, int frame, uint32_t* bits, uint64_t capacity_words, uint32_t dims[4]) {
    if (!c || !dims) return HVRT_ERR_INVALID;
    if (int r = bind(c)) return r;
    const CoverArgs& ca = c->cover_last;
    if (ca.n_frames < 1) return fail(c, HVRT_ERR_INVALID, "tile cover: the last launch ran without the pre-pass (counters, HVRT_TILE_COVER=0) or there was none");
    if (frame < 0 || frame >= ca.n_frames) return fail(c, HVRT_ERR_INVALID, "tile cover: frame %d outside 0..%d", frame, ca.n_frames - 1);
    dims[0] = ca.tiles_x; dims[1] = ca.tiles_y; dims[2] = ca.words_x;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(&dims[3], ca.slow_count + frame, sizeof(uint32_t), hipMemcpyDeviceToHost));
    const uint64_t n = (uint64_t)ca.tiles_y * ca.words_x;
    if (!bits) return HVRT_OK;
    if (capacity_words < n) return fail(c, HVRT_ERR_INVALID, "tile cover: %llu words needed, %llu given", (unsigned long long)n, (unsigned long long)capacity_words);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(bits, ca.bits + (uint64_t)frame * ca.frame_words, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return HVRT_OK;
}

int hvrt_debug_tile_slots(hvrt_ctx* c, uint32_t* slots, uint64_t capacity) {      // not in hvrt.h: scripts/tile_starts.py
    if (!c || !slots || !c->cost_valid) return HVRT_ERR_INVALID;
    if (int r = bind(c)) return r;
    const uint64_t n = render_tile_count(c->cost_key[0], hvrt_part_rows(c->cost_key[1], c->cost_key[2], c->cost_key[3], c->cost_key[4]), c->cost_key[5]);
    if (capacity < n) return HVRT_ERR_INVALID;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(slots, c->abandoned, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return HVRT_OK;
}

int hvrt_get_tile_costs(hvrt_ctx* c, uint32_t* cost, uint64_t capacity) {
    if (!c || !cost) return HVRT_ERR_INVALID;
    if (int r = bind(c)) return r;
    if (!c->cost_valid) return fail(c, HVRT_ERR_INVALID, "tile costs: no launch yet");
    const uint64_t n = render_tile_count(c->cost_key[0], hvrt_part_rows(c->cost_key[1], c->cost_key[2], c->cost_key[3], c->cost_key[4]), c->cost_key[5]);
    if (capacity < n) return fail(c, HVRT_ERR_INVALID, "tile costs: %llu words needed, %llu given", (unsigned long long)n, (unsigned long long)capacity);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(cost, c->tile_cost, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return (int)(n > 0x7FFFFFFFull ? 0x7FFFFFFF : n);
}

int hvrt_render_history(hvrt_ctx* c, float* ms, int max_entries) {
    if (!c) return -1;
    if (bind(c)) return -1;
    if (!ms || max_entries < 1) { fail(c, HVRT_ERR_INVALID, "render_history: no room for an entry"); return -1; }
    if (hipStreamSynchronize(c->stream) != hipSuccess) { fail(c, HVRT_ERR_HIP, "render_history: hipStreamSynchronize failed"); return -1; }
    if (c->render_pending) { c->tm.render_ms = ms_between(c->ev[6], c->ev[7]); c->render_pending = false; }
    uint64_t n = c->renders_recorded < (uint64_t)hvrt_ctx::RENDER_RING ? c->renders_recorded : (uint64_t)hvrt_ctx::RENDER_RING;
    if (n > (uint64_t)max_entries) n = (uint64_t)max_entries;
    for (uint64_t i = 0; i < n; ++i) {                 // oldest of the n first
        const int k = (int)((c->renders_recorded - n + i) % hvrt_ctx::RENDER_RING);
        ms[i] = ms_between(c->render_ring[k][0], c->render_ring[k][1]);
    }
    return (int)n;
}

int hvrt_render_part_device_noalloc(hvrt_ctx* c, const hvrt_camera_frame* frame, int width, int height, int spp,
                                    uint32_t seed, int tile_rows, int part, int n_parts, void* d_rgb, void* d_hit_id,
                                    void* d_t, void* d_counters) {
    if (!c) return HVRT_ERR_INVALID;
    if (int r = bind(c)) return r;
    if (int r = require_scene(c, false)) return r;
    if (!frame) return fail(c, HVRT_ERR_INVALID, "render: null camera frame");
    if (width < 1 || height < 1 || width > 0x7FFFFFF0 || height > 0x7FFFFFF0 || (uint64_t)width * height > 0x7FFFFFFFull)
        return fail(c, HVRT_ERR_INVALID, "render: bad image size %d x %d", width, height);
    if (spp < 1) return fail(c, HVRT_ERR_INVALID, "render: spp must be >= 1");
    if (tile_rows < 1 || n_parts < 1 || part < 0 || part >= n_parts)
        return fail(c, HVRT_ERR_INVALID, "render: bad partition (tile_rows %d, part %d of %d)", tile_rows, part, n_parts);
    const int local_rows = hvrt_part_rows(height, tile_rows, part, n_parts);
    const uint64_t n_tiles = render_tile_count(width, local_rows, 1);
    if (n_tiles > NOALLOC_MAX_TILES)
        return fail(c, HVRT_ERR_INVALID, "render (noalloc): %llu tiles of 8x8 pixels exceed the fixed scratch of %u tiles "
                    "(%d x %d pixels in this part); use hvrt_render_part_device", (unsigned long long)n_tiles,
                    NOALLOC_MAX_TILES, width, local_rows);
    if (!c->noalloc_scratch) return fail(c, HVRT_ERR_HIP, "render (noalloc): the scratch array was not resolved");
    hipStream_t s = c->stream;
    unsigned char* sp = c->noalloc_scratch;      // device address of this GPU's array, never the symbol itself
    hvrt_camera_frame* d_frame = (hvrt_camera_frame*)(sp + NOALLOC_FRAME);

    RenderArgs ra = {};
    ra.frames = d_frame; ra.n_frames = 1;
    ra.width = width; ra.height = height; ra.spp = spp; ra.seed = seed;
    ra.width_f = (float)width; ra.height_f = (float)height; ra.spp_f = (float)spp;
    ra.nearest = c->mode;
    ra.tile_rows = tile_rows; ra.part = part; ra.n_parts = n_parts;
    ra.local_rows = local_rows; ra.slab_rows = local_rows;
    ra.rgb = (float*)d_rgb; ra.hit_id = (uint32_t*)d_hit_id; ra.t = (float*)d_t;
    ra.counters = (unsigned long long*)d_counters;
    ra.hits = nullptr;
    ra.tile_counter = (uint32_t*)(sp + NOALLOC_QUEUE_HEAD);
    ra.n_heads = 1;
    ra.order = nullptr;                 // no history is kept between these calls: rows of tiles from the middle outwards
    ra.split_tiles = nullptr; ra.n_split = nullptr;
    ra.abandoned = nullptr; ra.n_abandoned = nullptr; ra.abort_ticks = 0; ra.few_tiles = 0; ra.few_report = nullptr; ra.few_serial = 0;   // the plain kernel: no tile is handed on
    ra.tile_cost = (uint32_t*)(sp + NOALLOC_TILE_COST);
    ra.cost_sum = (unsigned long long*)(sp + NOALLOC_COST_SUM);
    // the array keeps the last call's contents: the queue head and the duration sum must start at zero (the
    // duration words and the camera slot are overwritten before they are read)
    HIP_TRY(c, hipMemsetAsync(sp, 0, 16, s));
    k_store_frame<<<1, 64, 0, s>>>(*frame, (float*)d_frame);
    HIP_TRY(c, hipGetLastError());
    next_render_events(c);
    HIP_TRY(c, hipEventRecord(c->ev[6], s));
    HIP_TRY(c, launch_render(scene_of(c), ra, c->n_cus, false, s));
    HIP_TRY(c, hipEventRecord(c->ev[7], s));
    c->render_pending = true;
    return HVRT_OK;
}

int hvrt_render(hvrt_ctx* c, const hvrt_camera_frame* frame, int width, int height, int spp, uint32_t seed,
                float* rgb, uint32_t* hit_id, float* t, uint64_t counters[2]) {
    if (!c) return HVRT_ERR_INVALID;
    if (int r = bind(c)) return r;
    if (width < 1 || height < 1 || width > 0x7FFFFFF0 || height > 0x7FFFFFF0) return fail(c, HVRT_ERR_INVALID, "render: bad image size %d x %d", width, height);
    const uint64_t npx = (uint64_t)width * height;
    if (rgb) if (int r = ensure_bytes(c, c->out_buf[0], c->out_cap[0], npx * 12)) return r;
    if (hit_id) if (int r = ensure_bytes(c, c->out_buf[1], c->out_cap[1], npx * 4)) return r;
    if (t) if (int r = ensure_bytes(c, c->out_buf[2], c->out_cap[2], npx * 4)) return r;
    hipStream_t s = c->stream;
    HIP_TRY(c, hipMemsetAsync(c->counters, 0, 24, s));      // voxels, tests, hits
    // A large frame is traced as two launches over one queue (rows of the lower half + every long tile, then the
    // rest): the first half crosses PCIe while the second is traced.  Needs the tile history (a first launch of a
    // shape is one launch, copied afterwards).
    SegPlan plan;
    static const bool trace_times = getenv("HVRT_TRACE_TIMES") != nullptr;      // diagnostics: host time stamps of this call
    const auto t_start = std::chrono::steady_clock::now();
    auto stamp = [&](const char* what) {
        if (trace_times) fprintf(stderr, "[hvrt_render] %-28s %8.3f ms\n", what,
                                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
    };
    const uint32_t tiles_y = (uint32_t)(height + 7) >> 3;
    // (only when the colours alone are wanted: with three arrays to fetch the copies are the longer part either way)
    if (npx >= (1u << 21) && tiles_y >= 16 && !hit_id && !t) { plan.n = 2; plan.row[0] = 0; plan.row[1] = tiles_y / 2; plan.row[2] = tiles_y; }
    if (int r = render_batch_part(c, frame, 1, width, height, spp, seed, height, 0, 1, height,
                                  rgb ? c->out_buf[0] : nullptr, hit_id ? c->out_buf[1] : nullptr,
                                  t ? c->out_buf[2] : nullptr, counters ? c->counters : nullptr, c->counters + 2, &plan)) return r;
    stamp("launches queued");
    uint64_t h_cnt[3] = { 0, 0, 0 };
    if (plan.used) {
        for (int b = 0; b < plan.n; ++b) {
            const uint64_t y0 = (uint64_t)plan.row[b] * 8, y1 = b + 1 < plan.n ? (uint64_t)plan.row[b + 1] * 8 : (uint64_t)height;
            const uint64_t p0 = y0 * width, np = (y1 - y0) * width;
            HIP_TRY(c, hipStreamWaitEvent(c->copy_stream, c->ev_seg[b], 0));
            if (rgb) if (int r = download(c, rgb + 3 * p0, (const char*)c->out_buf[0] + 12 * p0, np * 12, c->copy_stream)) return r;
            if (hit_id) if (int r = download(c, hit_id + p0, (const char*)c->out_buf[1] + 4 * p0, np * 4, c->copy_stream)) return r;
            if (t) if (int r = download(c, t + p0, (const char*)c->out_buf[2] + 4 * p0, np * 4, c->copy_stream)) return r;
            stamp("segment downloaded");
        }
        HIP_TRY(c, hipMemcpyAsync(h_cnt, c->counters, sizeof h_cnt, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
    } else {
        HIP_TRY(c, hipMemcpyAsync(h_cnt, c->counters, sizeof h_cnt, hipMemcpyDeviceToHost, s));
        if (rgb) if (int r = download(c, rgb, c->out_buf[0], npx * 12, s)) return r;
        if (hit_id) if (int r = download(c, hit_id, c->out_buf[1], npx * 4, s)) return r;
        if (t) if (int r = download(c, t, c->out_buf[2], npx * 4, s)) return r;
    }
    HIP_TRY(c, hipStreamSynchronize(s));
    stamp("synchronised");
    if (counters) { counters[0] = h_cnt[0]; counters[1] = h_cnt[1]; }
    c->last_hits = h_cnt[2];
    c->tm.render_ms = ms_between(c->ev[6], c->ev[7]);
    c->render_pending = false;
    return HVRT_OK;
}

int hvrt_quantise_rgb8_device(hvrt_ctx* c, const void* d_rgb, uint64_t n_values, void* d_rgb8) {
    if (!c) return HVRT_ERR_INVALID;
    if (int r = bind(c)) return r;
    if (n_values && (!d_rgb || !d_rgb8)) return fail(c, HVRT_ERR_INVALID, "quantise: null pointer");
    if (((uintptr_t)d_rgb & 15) || ((uintptr_t)d_rgb8 & 3)) return fail(c, HVRT_ERR_INVALID, "quantise: d_rgb must be 16-byte and d_rgb8 4-byte aligned");
    HIP_TRY(c, launch_quantise_rgb8((const float*)d_rgb, n_values, (uint8_t*)d_rgb8, c->stream));
    return HVRT_OK;
}

int hvrt_render_views_rgb8(hvrt_ctx* c, const hvrt_camera_frame* frames, int n_frames, int width, int height,
                           uint8_t* rgb8, uint32_t* hit_id, uint64_t counters[2]) {
    if (!c) return HVRT_ERR_INVALID;
    if (int r = bind(c)) return r;
    if (width < 1 || height < 1 || width > 0x7FFFFFF0 || height > 0x7FFFFFF0) return fail(c, HVRT_ERR_INVALID, "render: bad image size %d x %d", width, height);
    if (n_frames < 1 || n_frames > HVRT_MAX_BATCH_FRAMES)
        return fail(c, HVRT_ERR_INVALID, "render: n_frames %d outside 1..%d", n_frames, HVRT_MAX_BATCH_FRAMES);
    if (!rgb8) return fail(c, HVRT_ERR_INVALID, "render: null output");
    const uint64_t npx = (uint64_t)width * height * (uint64_t)n_frames;
    // float frames and their bytes share the first staging buffer (bytes behind the floats, 16-B aligned)
    const uint64_t f_bytes = (npx * 12 + 15) / 16 * 16;
    if (int r = ensure_bytes(c, c->out_buf[0], c->out_cap[0], f_bytes + npx * 3 + 16)) return r;
    if (hit_id) if (int r = ensure_bytes(c, c->out_buf[1], c->out_cap[1], npx * 4)) return r;
    hipStream_t s = c->stream;
    HIP_TRY(c, hipMemsetAsync(c->counters, 0, 24, s));      // voxels, tests, hits
    if (int r = render_batch_part(c, frames, n_frames, width, height, 1, 0, height, 0, 1, height,
                                  c->out_buf[0], hit_id ? c->out_buf[1] : nullptr, nullptr,
                                  counters ? c->counters : nullptr, c->counters + 2)) return r;
    uint8_t* d8 = (uint8_t*)c->out_buf[0] + f_bytes;
    HIP_TRY(c, launch_quantise_rgb8((const float*)c->out_buf[0], npx * 3, d8, s));
    uint64_t h_cnt[3] = { 0, 0, 0 };
    HIP_TRY(c, hipMemcpyAsync(h_cnt, c->counters, sizeof h_cnt, hipMemcpyDeviceToHost, s));
    if (int r = download(c, rgb8, d8, npx * 3, s)) return r;
    if (hit_id) if (int r = download(c, hit_id, c->out_buf[1], npx * 4, s)) return r;
    HIP_TRY(c, hipStreamSynchronize(s));
    if (counters) { counters[0] = h_cnt[0]; counters[1] = h_cnt[1]; }
    c->last_hits = h_cnt[2];
    c->tm.render_ms = ms_between(c->ev[6], c->ev[7]);
    c->render_pending = false;
    return HVRT_OK;
}

int hvrt_assemble_parts_device(hvrt_ctx* c, const void* d_parts, int width, int height, int tile_rows, int n_parts,
                               int slab_rows, int bytes_per_pixel, void* d_frame) {
    if (!c) return HVRT_ERR_INVALID;
    if (int r = bind(c)) return r;
    if (!d_parts || !d_frame) return fail(c, HVRT_ERR_INVALID, "assemble: null pointer");
    // rows are moved as 4-byte words: 12 (float rgb) and 4 (hit id, t) bytes per pixel always qualify, 3 (8-bit rgb)
    // when the row does (width a multiple of 4)
    if (width < 1 || height < 1 || tile_rows < 1 || n_parts < 1 || bytes_per_pixel < 1 ||
        (((int64_t)width * bytes_per_pixel) & 3) || ((uintptr_t)d_parts & 3) || ((uintptr_t)d_frame & 3))
        return fail(c, HVRT_ERR_INVALID, "assemble: bad geometry (a row of %d pixels x %d bytes must be a whole number of "
                    "4-byte words, buffers 4-byte aligned)", width, bytes_per_pixel);
    for (int p = 0; p < n_parts; ++p)
        if (hvrt_part_rows(height, tile_rows, p, n_parts) > slab_rows)
            return fail(c, HVRT_ERR_INVALID, "assemble: slab_rows %d too small for part %d", slab_rows, p);
    HIP_TRY(c, launch_assemble(d_parts, width, height, tile_rows, n_parts, slab_rows, bytes_per_pixel, d_frame, c->stream));
    return HVRT_OK;
}

int hvrt_trace(hvrt_ctx* c, const float* origins, const float* dirs, uint32_t n, hvrt_hit* out, uint64_t counters[2]) {
    if (!c) return HVRT_ERR_INVALID;
    if (int r = bind(c)) return r;
    if (int r = require_scene(c)) return r;
    if (n && (!origins || !dirs || !out)) return fail(c, HVRT_ERR_INVALID, "trace: null pointer");
    if (int r = ensure_bytes(c, c->io_buf[0], c->io_cap[0], (uint64_t)n * 12)) return r;
    if (int r = ensure_bytes(c, c->io_buf[1], c->io_cap[1], (uint64_t)n * 12)) return r;
    if (int r = ensure_bytes(c, c->io_buf[2], c->io_cap[2], (uint64_t)n * sizeof(hvrt_hit))) return r;
    hipStream_t s = c->stream;
    if (counters) HIP_TRY(c, hipMemsetAsync(c->counters, 0, 16, s));
    if (n) {
        HIP_TRY(c, hipMemcpyAsync(c->io_buf[0], origins, (size_t)n * 12, hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipMemcpyAsync(c->io_buf[1], dirs, (size_t)n * 12, hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipEventRecord(c->ev[6], s));
        HIP_TRY(c, launch_trace(scene_of(c), (const float*)c->io_buf[0], (const float*)c->io_buf[1], n,
                                (hvrt_hit*)c->io_buf[2], counters ? c->counters : nullptr, c->mode, s));
        HIP_TRY(c, hipEventRecord(c->ev[7], s));
        if (int r = download(c, out, c->io_buf[2], (uint64_t)n * sizeof(hvrt_hit), s)) return r;
    }
    if (counters) HIP_TRY(c, hipMemcpyAsync(counters, c->counters, 16, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    if (n) c->tm.trace_ms = ms_between(c->ev[6], c->ev[7]);
    return HVRT_OK;
}

int hvrt_trace_device(hvrt_ctx* c, const void* d_origins, const void* d_dirs, uint32_t n, void* d_hits, void* d_counters) {
    if (!c) return HVRT_ERR_INVALID;
    if (int r = bind(c)) return r;
    if (int r = require_scene(c)) return r;
    if (n && (!d_origins || !d_dirs || !d_hits)) return fail(c, HVRT_ERR_INVALID, "trace: null pointer");
    if (!n) return HVRT_OK;
    hipStream_t s = c->stream;
    HIP_TRY(c, hipEventRecord(c->ev[6], s));
    HIP_TRY(c, launch_trace(scene_of(c), (const float*)d_origins, (const float*)d_dirs, n, (hvrt_hit*)d_hits,
                            (unsigned long long*)d_counters, c->mode, s));
    HIP_TRY(c, hipEventRecord(c->ev[7], s));
    c->render_pending = true;      // hvrt_synchronize reads the pair of events into render_ms
    return HVRT_OK;
}

}  // extern "C"

// for hvrt_group.hip: hvrt_render_part_device that also counts the pixels whose (sample-0) ray hit a triangle
namespace hvrt {
int render_part_counting_hits(hvrt_ctx* c, const hvrt_camera_frame* frame, int width, int height, int spp, uint32_t seed,
                              int tile_rows, int part, int n_parts, void* d_rgb, void* d_hit_id, void* d_t, void* d_counters, void* d_hits) {
    const int rows = hvrt_part_rows(height, tile_rows, part, n_parts);
    return render_batch_part(c, frame, 1, width, height, spp, seed, tile_rows, part, n_parts, rows, d_rgb, d_hit_id, d_t,
                             d_counters, d_hits);
}
}  // namespace hvrt
