// hvrt_group.hip -- several GPUs of one node behind the C ABI (include/hvrt.h, "group" section).
//
// One process, one hvrt_ctx per device, one host thread per member (everything that touches a member's device after
// its creation runs on that thread, so the members work -- create, adopt the scene, trace, download -- at the same
// time).  The scene is replicated (member 0 builds it, the others receive mesh and CSR and derive their own
// acceleration data); a frame is cut into row tiles, tile k to member k % n, every member traces its tiles into a
// compact slab.  Where the slabs go depends on who wants the frame:
//   host   (hvrt_group_render)  every member copies its own tiles straight into the caller's arrays -- a 16-row tile is
//          one contiguous byte range of a row-major frame -- over its own PCIe link, through its context's pinned ring.
//          Nothing crosses xGMI, nothing is reassembled.
//   device (hvrt_group_render_device)  the slabs travel to member 0, which reassembles the frame in its memory:
//     rccl   ncclCommInitAll over the members' devices; ncclBroadcast of the scene arrays, grouped ncclSend / ncclRecv
//            of the slabs (xGMI is point to point: n - 1 concurrent sends into member 0 use n - 1 links).  librccl is
//            opened at run time (dlopen), so libhvrt itself carries no dependency on it and a process that already
//            holds a copy -- torch's -- shares that copy.
//     copy   hipMemcpyAsync between the devices (peer access where the hardware offers it), ordered with events.
//            Taken when a device appears twice in the list (two contexts on one GPU: how the choreography is tested
//            on a one-GPU machine) or when HVRT_GROUP_TRANSPORT=copy asks for it.
// No counterpart in the reference (one thread, one CPU); the seam is the tracer factory, TraceImages.cpp:464-484.
#include "hvrt_internal.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace hvrt {
// hvrt_api.hip: hvrt_render_part_device plus the number of pixels whose (sample-0) ray hit a triangle, added to *d_hits (u64)
int render_part_counting_hits(hvrt_ctx* c, const hvrt_camera_frame* frame, int width, int height, int spp, uint32_t seed,
                              int tile_rows, int part, int n_parts, void* d_rgb, void* d_hit_id, void* d_t, void* d_counters, void* d_hits);
}

namespace {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;

    bool load(std::string& why) {
        const char* names[] = { "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so" };
        for (const char* n : names) {
            lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) { why = std::string("cannot open librccl: ") + dlerror(); return false; }
        auto sym = [&](const char* n) { void* p = dlsym(lib, n); if (!p) why = std::string("librccl lacks ") + n; return p; };
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        Broadcast = (decltype(Broadcast))sym("ncclBroadcast");
        Send = (decltype(Send))sym("ncclSend");
        Recv = (decltype(Recv))sym("ncclRecv");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        return CommInitAll && CommDestroy && Broadcast && Send && Recv && GroupStart && GroupEnd && GetErrorString;
    }
};

// A member's host thread.  submit() hands it one job, wait() returns the job's status; jobs of different members run
// side by side, the caller of the group's entry points only hands out work and collects it.
struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, busy = false, quit = false;
    int rc = 0;

    void start() {
        th = std::thread([this] {
            std::unique_lock<std::mutex> lk(mu);
            for (;;) {
                cv.wait(lk, [this] { return has_job || quit; });
                if (!has_job) return;
                std::function<int()> j = std::move(job);
                has_job = false;
                lk.unlock();
                const int r = j();
                lk.lock();
                rc = r;
                busy = false;
                cv.notify_all();
            }
        });
    }
    void submit(std::function<int()> j) {
        std::lock_guard<std::mutex> lk(mu);
        job = std::move(j);
        has_job = busy = true;
        cv.notify_all();
    }
    int wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this] { return !busy; });
        return rc;
    }
    void stop() {
        if (!th.joinable()) return;
        { std::lock_guard<std::mutex> lk(mu); quit = true; cv.notify_all(); }
        th.join();
    }
};

struct Member {
    int device = 0;
    hvrt_ctx* ctx = nullptr;
    hipStream_t stream = nullptr;            // the context's own stream
    hipEvent_t done = nullptr;               // this member's slab is complete
    void* slab[3] = {};                      // rgb (12 B / px), hit id (4), t (4): this member's rows, in tile order
    size_t slab_cap[3] = {};                 // bytes
    unsigned long long* counters = nullptr;  // device: voxels, tests, hits
    unsigned long long h_counters[3] = {};
    float render_ms = 0.0f, download_ms = 0.0f;
    ncclComm_t comm = nullptr;
    std::string error;                       // written by the member's thread, read after wait()
    Worker* worker = nullptr;
};

double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace

struct hvrt_group {
    std::vector<Member> m;
    std::string error;
    enum { NONE, COPY, RCCL } transport = NONE;
    Rccl rccl;
    float* parts = nullptr; size_t parts_cap = 0;      // member 0: n slabs (device-resident frames only)
    float* frame = nullptr; size_t frame_cap = 0;      // member 0: the reassembled frame
    double last_exchange_bytes = 0.0;
    hvrt_group_timings tm = {};
};

namespace {

thread_local std::string g_group_create_error;

int gfail(hvrt_group* g, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (g) g->error = buf; else g_group_create_error = buf;
    return code;
}
// on a member's thread: the message stays with the member until the caller has joined
int mfail(Member& mb, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    mb.error = buf;
    return code;
}

#define G_HIP(g, call)                                                                                         \
    do {                                                                                                       \
        hipError_t e_ = (call);                                                                                \
        if (e_ != hipSuccess) return gfail((g), e_ == hipErrorOutOfMemory ? HVRT_ERR_OOM : HVRT_ERR_HIP,       \
                                           "%s failed: %s", #call, hipGetErrorString(e_));                    \
    } while (0)
#define G_NCCL(g, call)                                                                                        \
    do {                                                                                                       \
        ncclResult_t r_ = (call);                                                                              \
        if (r_ != ncclSuccess) return gfail((g), HVRT_ERR_HIP, "%s failed: %s", #call, (g)->rccl.GetErrorString(r_)); \
    } while (0)
#define G_CTX(g, member, call)                                                                                 \
    do {                                                                                                       \
        int r_ = (call);                                                                                       \
        if (r_ != HVRT_OK) return gfail((g), r_, "member %d: %s", (member), hvrt_last_error((g)->m[(member)].ctx)); \
    } while (0)
#define M_HIP(mb, call)                                                                                        \
    do {                                                                                                       \
        hipError_t e_ = (call);                                                                                \
        if (e_ != hipSuccess) return mfail((mb), e_ == hipErrorOutOfMemory ? HVRT_ERR_OOM : HVRT_ERR_HIP,      \
                                           "%s failed: %s", #call, hipGetErrorString(e_));                    \
    } while (0)
#define M_CTX(mb, call)                                                                                        \
    do {                                                                                                       \
        int r_ = (call);                                                                                       \
        if (r_ != HVRT_OK) return mfail((mb), r_, "%s", hvrt_last_error((mb).ctx));                            \
    } while (0)

template <typename T>
int grow(hvrt_group* g, T*& p, size_t& cap, size_t need) {
    if (need <= cap && p) return HVRT_OK;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    G_HIP(g, hipMalloc((void**)&p, (need ? need : 1) * sizeof(T)));
    cap = need;
    return HVRT_OK;
}
int mgrow(Member& mb, void*& p, size_t& cap, size_t need) {
    if (need <= cap && p) return HVRT_OK;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    M_HIP(mb, hipMalloc(&p, need ? need : 1));
    cap = need;
    return HVRT_OK;
}

// Runs job(i) on the thread of every member in [first, n) and waits for all of them; the first failure (lowest member)
// becomes the group's error.
int on_members(hvrt_group* g, int first, const std::function<int(int)>& job) {
    const int n = (int)g->m.size();
    for (int i = first; i < n; ++i) {
        g->m[i].error.clear();
        g->m[i].worker->submit([&job, i] { return job(i); });
    }
    int rc = HVRT_OK, who = -1;
    for (int i = first; i < n; ++i) {
        const int r = g->m[i].worker->wait();
        if (r != HVRT_OK && rc == HVRT_OK) { rc = r; who = i; }
    }
    if (rc != HVRT_OK) return gfail(g, rc, "member %d: %s", who, g->m[who].error.c_str());
    return HVRT_OK;
}

constexpr int TILE_ROWS = 16;

// Member i traces its row tiles of the frame into its slabs (d_hit / d_t only when wanted) -- on the member's thread.
int member_trace(hvrt_group* g, int i, const hvrt_camera_frame* frame, int width, int height, int spp, uint32_t seed,
                 bool want_rgb, bool want_hit, bool want_t, int mode) {
    Member& mb = g->m[i];
    const int n = (int)g->m.size();
    const size_t px = (size_t)hvrt_part_rows(height, TILE_ROWS, i, n) * width;
    M_HIP(mb, hipSetDevice(mb.device));
    if (hvrt_get_mode(mb.ctx) != mode) M_CTX(mb, hvrt_set_mode(mb.ctx, mode));
    if (want_rgb) if (int r = mgrow(mb, mb.slab[0], mb.slab_cap[0], px * 12)) return r;
    if (want_hit) if (int r = mgrow(mb, mb.slab[1], mb.slab_cap[1], px * 4)) return r;
    if (want_t) if (int r = mgrow(mb, mb.slab[2], mb.slab_cap[2], px * 4)) return r;
    M_HIP(mb, hipMemsetAsync(mb.counters, 0, 3 * sizeof(unsigned long long), mb.stream));
    M_CTX(mb, hvrt::render_part_counting_hits(mb.ctx, frame, width, height, spp, seed, TILE_ROWS, i, n, want_rgb ? mb.slab[0] : nullptr,
                                              want_hit ? mb.slab[1] : nullptr, want_t ? mb.slab[2] : nullptr, mb.counters, mb.counters + 2));
    M_HIP(mb, hipEventRecord(mb.done, mb.stream));
    return HVRT_OK;
}

int member_counters(Member& mb) {
    M_HIP(mb, hipSetDevice(mb.device));
    M_HIP(mb, hipMemcpyAsync(mb.h_counters, mb.counters, sizeof mb.h_counters, hipMemcpyDeviceToHost, mb.stream));
    M_CTX(mb, hvrt_synchronize(mb.ctx));
    hvrt_timings t;
    if (hvrt_get_timings(mb.ctx, &t) == HVRT_OK) mb.render_ms = t.render_ms;
    return HVRT_OK;
}

void sum_statistics(hvrt_group* g, uint64_t counters[2], uint64_t* hits) {
    uint64_t sum[3] = { 0, 0, 0 };
    float longest = 0.0f, dl = 0.0f;
    for (Member& mb : g->m) {
        for (int k = 0; k < 3; ++k) sum[k] += mb.h_counters[k];
        longest = mb.render_ms > longest ? mb.render_ms : longest;
        dl = mb.download_ms > dl ? mb.download_ms : dl;
    }
    if (counters) { counters[0] = sum[0]; counters[1] = sum[1]; }
    if (hits) *hits = sum[2];
    g->tm.render_ms = longest;
    g->tm.download_ms = dl;
}

}  // namespace

extern "C" {

const char* hvrt_group_last_error(const hvrt_group* g) { return g ? g->error.c_str() : g_group_create_error.c_str(); }

void hvrt_group_destroy(hvrt_group* g) {
    if (!g) return;
    for (Member& mb : g->m) if (mb.worker) mb.worker->wait();
    for (Member& mb : g->m) {
        if (!mb.ctx) continue;
        (void)hipSetDevice(mb.device);
        (void)hvrt_synchronize(mb.ctx);
        if (mb.comm) (void)g->rccl.CommDestroy(mb.comm);
        for (void* p : mb.slab) if (p) (void)hipFree(p);
        if (mb.counters) (void)hipFree(mb.counters);
        if (mb.done) (void)hipEventDestroy(mb.done);
    }
    if (!g->m.empty()) {
        (void)hipSetDevice(g->m[0].device);
        if (g->parts) (void)hipFree(g->parts);
        if (g->frame) (void)hipFree(g->frame);
    }
    for (Member& mb : g->m) if (mb.ctx) hvrt_destroy(mb.ctx);
    for (Member& mb : g->m) if (mb.worker) { mb.worker->stop(); delete mb.worker; mb.worker = nullptr; }
    delete g;
}

int hvrt_group_create(hvrt_group** out, const int* device_ids, int n_devices) {
    if (!out) return HVRT_ERR_INVALID;
    *out = nullptr;
    if (!device_ids || n_devices < 1 || n_devices > 64) return gfail(nullptr, HVRT_ERR_INVALID, "group_create: 1..64 devices");
    const auto t0 = std::chrono::steady_clock::now();
    hvrt_group* g = new hvrt_group;
    auto bail = [&](int code) { g_group_create_error = g->error; hvrt_group_destroy(g); return code; };
    bool repeated = false;
    for (int i = 0; i < n_devices; ++i)
        for (int j = 0; j < i; ++j) repeated |= device_ids[i] == device_ids[j];
    g->m.resize(n_devices);
    for (int i = 0; i < n_devices; ++i) {
        g->m[i].device = device_ids[i];
        g->m[i].worker = new Worker;
        g->m[i].worker->start();
    }
    // every member's context at the same time: hvrt_create is 100-200 ms of runtime start-up and warm-up per device
    const int rc = on_members(g, 0, [g](int i) -> int {
        Member& mb = g->m[i];
        if (hvrt_create(&mb.ctx, mb.device) != HVRT_OK) return mfail(mb, HVRT_ERR_HIP, "device %d: %s", mb.device, hvrt_last_error(nullptr));
        mb.stream = (hipStream_t)hvrt_stream(mb.ctx);
        M_HIP(mb, hipSetDevice(mb.device));
        M_HIP(mb, hipEventCreateWithFlags(&mb.done, hipEventDisableTiming));
        M_HIP(mb, hipMalloc((void**)&mb.counters, 3 * sizeof(unsigned long long)));
        return HVRT_OK;
    });
    if (rc != HVRT_OK) return bail(rc);
    // (one device and HVRT_GROUP_TRANSPORT=rccl: the communicator is still created and the scene broadcast runs, in
    // place, on that one rank -- how the loader and the calls are exercised on a one-GPU machine)
    const char* want_env = getenv("HVRT_GROUP_TRANSPORT");
    if (n_devices > 1 || (want_env && strcmp(want_env, "rccl") == 0)) {
        const char* want = want_env;
        const bool copy = repeated || (want && strcmp(want, "copy") == 0);
        if (want && strcmp(want, "copy") != 0 && strcmp(want, "rccl") != 0) {
            gfail(g, HVRT_ERR_INVALID, "HVRT_GROUP_TRANSPORT must be rccl or copy (got %s)", want);
            return bail(HVRT_ERR_INVALID);
        }
        if (repeated && want && strcmp(want, "rccl") == 0) {
            gfail(g, HVRT_ERR_INVALID, "group_create: a communicator cannot hold one device twice; the copy transport can");
            return bail(HVRT_ERR_INVALID);
        }
        if (copy) {
            g->transport = hvrt_group::COPY;
            for (int i = 1; i < n_devices; ++i) {        // best effort: without peer access the copies are staged by the runtime
                int can = 0;
                if (g->m[i].device != g->m[0].device && hipDeviceCanAccessPeer(&can, g->m[0].device, g->m[i].device) == hipSuccess && can) {
                    (void)hipSetDevice(g->m[0].device);
                    if (hipDeviceEnablePeerAccess(g->m[i].device, 0) != hipSuccess) (void)hipGetLastError();
                    (void)hipSetDevice(g->m[i].device);
                    if (hipDeviceEnablePeerAccess(g->m[0].device, 0) != hipSuccess) (void)hipGetLastError();
                }
            }
        } else {
            std::string why;
            if (!g->rccl.load(why)) { gfail(g, HVRT_ERR_HIP, "group_create: %s", why.c_str()); return bail(HVRT_ERR_HIP); }
            std::vector<ncclComm_t> comms(n_devices);
            ncclResult_t r = g->rccl.CommInitAll(comms.data(), n_devices, device_ids);
            if (r != ncclSuccess) { gfail(g, HVRT_ERR_HIP, "ncclCommInitAll: %s", g->rccl.GetErrorString(r)); return bail(HVRT_ERR_HIP); }
            for (int i = 0; i < n_devices; ++i) g->m[i].comm = comms[i];
            g->transport = hvrt_group::RCCL;
        }
    }
    g->tm.create_ms = (float)ms_since(t0);
    *out = g;
    return HVRT_OK;
}

int hvrt_group_size(const hvrt_group* g) { return g ? (int)g->m.size() : 0; }

hvrt_ctx* hvrt_group_member(hvrt_group* g, int i) { return (g && i >= 0 && i < (int)g->m.size()) ? g->m[i].ctx : nullptr; }

const char* hvrt_group_transport(const hvrt_group* g) {
    if (!g) return "";
    return g->transport == hvrt_group::RCCL ? "rccl" : g->transport == hvrt_group::COPY ? "copy" : "none";
}

int hvrt_group_get_timings(const hvrt_group* g, hvrt_group_timings* out) {
    if (!g || !out) return HVRT_ERR_INVALID;
    *out = g->tm;
    return HVRT_OK;
}

double hvrt_group_last_exchange_bytes(const hvrt_group* g) { return g ? g->last_exchange_bytes : 0.0; }

int hvrt_group_set_mode(hvrt_group* g, int mode) {
    if (!g) return HVRT_ERR_INVALID;
    for (int i = 0; i < (int)g->m.size(); ++i) G_CTX(g, i, hvrt_set_mode(g->m[i].ctx, mode));
    return HVRT_OK;
}

int hvrt_group_replicate_scene(hvrt_group* g) {
    if (!g) return HVRT_ERR_INVALID;
    const auto t0 = std::chrono::steady_clock::now();
    const int n = (int)g->m.size();
    Member& root = g->m[0];
    hvrt_grid_info gi;
    G_CTX(g, 0, hvrt_get_grid_info(root.ctx, &gi));
    void *aos = nullptr, *off = nullptr, *idx = nullptr;
    G_CTX(g, 0, hvrt_get_device_arrays(root.ctx, &aos, &off, &idx));
    G_CTX(g, 0, hvrt_synchronize(root.ctx));
    g->tm.replicate_ms = 0.0f;
    if (g->transport == hvrt_group::NONE) return HVRT_OK;
    const size_t b_tri = (size_t)gi.n_triangles * 48, b_off = ((size_t)gi.n_voxels + 1) * 4, b_idx = (size_t)gi.n_refs * 4;
    if (g->transport == hvrt_group::COPY) {
        // the members copy straight out of member 0's arrays (a device-to-device copy across devices) and derive their
        // own acceleration data, all at the same time
        const int rc = on_members(g, 1, [&](int i) -> int {
            Member& mb = g->m[i];
            M_HIP(mb, hipSetDevice(mb.device));
            M_CTX(mb, hvrt_set_mesh_device(mb.ctx, aos, gi.n_triangles));
            M_CTX(mb, hvrt_set_grid_csr_device(mb.ctx, gi.nx, gi.ny, gi.nz, gi.min_bound, gi.max_bound, gi.voxel_size, off, idx, gi.n_refs));
            return HVRT_OK;
        });
        g->tm.replicate_ms = (float)ms_since(t0);
        return rc;
    }
    // rccl: one broadcast per array into staging buffers on every other member, which then adopts them
    std::vector<char*> st_tri(n, nullptr), st_off(n, nullptr), st_idx(n, nullptr);
    auto release = [&]() {
        for (int i = 1; i < n; ++i) {
            (void)hipSetDevice(g->m[i].device);
            if (st_tri[i]) (void)hipFree(st_tri[i]);
            if (st_off[i]) (void)hipFree(st_off[i]);
            if (st_idx[i]) (void)hipFree(st_idx[i]);
        }
    };
    int rc = [&]() -> int {
        for (int i = 1; i < n; ++i) {
            G_HIP(g, hipSetDevice(g->m[i].device));
            G_HIP(g, hipMalloc((void**)&st_tri[i], b_tri ? b_tri : 16));
            G_HIP(g, hipMalloc((void**)&st_off[i], b_off));
            G_HIP(g, hipMalloc((void**)&st_idx[i], b_idx ? b_idx : 16));
        }
        struct { const void* src; std::vector<char*>* dst; size_t bytes; } arrays[3] = { { aos, &st_tri, b_tri }, { off, &st_off, b_off }, { idx, &st_idx, b_idx } };
        for (auto& a : arrays) {
            if (a.bytes == 0) continue;
            G_NCCL(g, g->rccl.GroupStart());
            ncclResult_t first = ncclSuccess;         // (a group that was opened is closed whatever its calls returned)
            for (int i = 0; i < n && first == ncclSuccess; ++i) {
                void* recv = i == 0 ? const_cast<void*>(a.src) : (void*)(*a.dst)[i];       // in place on the root
                first = g->rccl.Broadcast(i == 0 ? a.src : recv, recv, a.bytes, ncclChar, 0, g->m[i].comm, g->m[i].stream);
            }
            const ncclResult_t end = g->rccl.GroupEnd();
            G_NCCL(g, first);
            G_NCCL(g, end);
        }
        // every member adopts its copy (and derives distance field, face bits and records) on its own thread
        if (int r = on_members(g, 1, [&](int i) -> int {
                Member& mb = g->m[i];
                M_HIP(mb, hipSetDevice(mb.device));
                M_HIP(mb, hipStreamSynchronize(mb.stream));
                M_CTX(mb, hvrt_set_mesh_device(mb.ctx, st_tri[i], gi.n_triangles));
                M_CTX(mb, hvrt_set_grid_csr_device(mb.ctx, gi.nx, gi.ny, gi.nz, gi.min_bound, gi.max_bound, gi.voxel_size, st_off[i], st_idx[i], gi.n_refs));
                return HVRT_OK;
            })) return r;
        G_HIP(g, hipSetDevice(root.device));
        G_HIP(g, hipStreamSynchronize(root.stream));
        return HVRT_OK;
    }();
    release();
    g->tm.replicate_ms = (float)ms_since(t0);
    return rc;
}

int hvrt_group_render(hvrt_group* g, const hvrt_camera_frame* frame, int width, int height, int spp, uint32_t seed,
                      float* rgb, uint32_t* hit_id, float* t, uint64_t counters[2], uint64_t* hits) {
    if (!g) return HVRT_ERR_INVALID;
    if (!frame || !rgb) return gfail(g, HVRT_ERR_INVALID, "group_render: null frame or output");
    if (width < 1 || height < 1 || width > 0x7FFFFFF0 || (uint64_t)width * height > 0x7FFFFFFFull) return gfail(g, HVRT_ERR_INVALID, "group_render: bad image size %d x %d", width, height);
    const auto t0 = std::chrono::steady_clock::now();
    const int n = (int)g->m.size();
    const int mode = hvrt_get_mode(g->m[0].ctx);      // member 0 is the one the caller talks to (hvrt_set_mode on it alone is enough)
    g->last_exchange_bytes = 0.0;                      // nothing travels between the devices on this path
    // every member, on its own thread: trace its row tiles, then copy them to where they belong in the caller's arrays --
    // tile j of member i is global tile j * n + i, i.e. 16 whole rows starting at row (j * n + i) * 16
    const int rc = on_members(g, 0, [&](int i) -> int {
        Member& mb = g->m[i];
        if (int r = member_trace(g, i, frame, width, height, spp, seed, true, hit_id != nullptr, t != nullptr, mode)) return r;
        const uint64_t px = (uint64_t)hvrt_part_rows(height, TILE_ROWS, i, n) * width, tile_px = (uint64_t)TILE_ROWS * width;
        const auto td = std::chrono::steady_clock::now();
        struct { void* dst; int k; uint64_t bpp; } outs[3] = { { rgb, 0, 12 }, { hit_id, 1, 4 }, { t, 2, 4 } };
        for (auto& o : outs) {
            if (!o.dst || !px) continue;
            M_CTX(mb, hvrt_download_blocks(mb.ctx, (char*)o.dst + (uint64_t)i * tile_px * o.bpp, mb.slab[o.k], px * o.bpp, tile_px * o.bpp, (uint64_t)n * tile_px * o.bpp));
        }
        if (int r = member_counters(mb)) return r;
        mb.download_ms = (float)ms_since(td);
        return HVRT_OK;
    });
    if (rc != HVRT_OK) return rc;
    sum_statistics(g, counters, hits);
    g->tm.total_ms = (float)ms_since(t0);
    return HVRT_OK;
}

int hvrt_group_render_device(hvrt_group* g, const hvrt_camera_frame* frame, int width, int height, int spp, uint32_t seed,
                             void** d_rgb, uint64_t counters[2], uint64_t* hits) {
    if (!g) return HVRT_ERR_INVALID;
    if (!frame || !d_rgb) return gfail(g, HVRT_ERR_INVALID, "group_render_device: null frame or output");
    if (width < 1 || height < 1 || width > 0x7FFFFFF0 || (uint64_t)width * height > 0x7FFFFFFFull) return gfail(g, HVRT_ERR_INVALID, "group_render_device: bad image size %d x %d", width, height);
    const auto t0 = std::chrono::steady_clock::now();
    const int n = (int)g->m.size();
    const int mode = hvrt_get_mode(g->m[0].ctx);
    int slab_rows = 0;
    for (int i = 0; i < n; ++i) { const int r = hvrt_part_rows(height, TILE_ROWS, i, n); slab_rows = r > slab_rows ? r : slab_rows; }
    const size_t slab_floats = (size_t)slab_rows * width * 3, frame_floats = (size_t)height * width * 3;
    Member& root = g->m[0];
    if (int r = on_members(g, 0, [&](int i) { return member_trace(g, i, frame, width, height, spp, seed, true, false, false, mode); })) return r;
    const float* result = (const float*)root.slab[0];       // one member: its slab is the frame
    g->last_exchange_bytes = 0.0;
    if (n > 1) {
        G_HIP(g, hipSetDevice(root.device));
        if (int r = grow(g, g->parts, g->parts_cap, slab_floats * n)) return r;
        if (int r = grow(g, g->frame, g->frame_cap, frame_floats)) return r;
        auto floats_of = [&](int i) { return (size_t)hvrt_part_rows(height, TILE_ROWS, i, n) * width * 3; };
        // member 0's own slab never leaves its device
        if (floats_of(0)) G_HIP(g, hipMemcpyAsync(g->parts, root.slab[0], floats_of(0) * 4, hipMemcpyDeviceToDevice, root.stream));
        if (g->transport == hvrt_group::RCCL) {
            G_NCCL(g, g->rccl.GroupStart());
            ncclResult_t first = ncclSuccess;
            for (int i = 1; i < n && first == ncclSuccess; ++i) {
                if (!floats_of(i)) continue;
                first = g->rccl.Send(g->m[i].slab[0], floats_of(i), ncclFloat, 0, g->m[i].comm, g->m[i].stream);
                if (first == ncclSuccess) first = g->rccl.Recv(g->parts + slab_floats * i, floats_of(i), ncclFloat, i, root.comm, root.stream);
                g->last_exchange_bytes += floats_of(i) * 4.0;
            }
            const ncclResult_t end = g->rccl.GroupEnd();
            G_NCCL(g, first);
            G_NCCL(g, end);
        } else {
            for (int i = 1; i < n; ++i) {
                if (!floats_of(i)) continue;
                G_HIP(g, hipStreamWaitEvent(root.stream, g->m[i].done, 0));
                G_HIP(g, hipMemcpyAsync(g->parts + slab_floats * i, g->m[i].slab[0], floats_of(i) * 4, hipMemcpyDefault, root.stream));
                g->last_exchange_bytes += floats_of(i) * 4.0;
            }
        }
        G_CTX(g, 0, hvrt_assemble_parts_device(root.ctx, g->parts, width, height, TILE_ROWS, n, slab_rows, 12, g->frame));
        result = g->frame;
    }
    // (member 0 last: its stream also carries the gather and the reassembly)
    for (int i = n - 1; i >= 0; --i) {
        Member& mb = g->m[i];
        mb.download_ms = 0.0f;
        if (member_counters(mb) != HVRT_OK) return gfail(g, HVRT_ERR_HIP, "member %d: %s", i, mb.error.c_str());
    }
    sum_statistics(g, counters, hits);
    *d_rgb = const_cast<float*>(result);
    g->tm.total_ms = (float)ms_since(t0);
    return HVRT_OK;
}

}  // extern "C"
