// hvrt_group.hip -- several GPUs of one node behind the C ABI (include/hvrt.h, "group" section).
//
// One process, one hvrt_ctx per device.  The scene is replicated (member 0 builds it, the others receive mesh and
// CSR and derive their own acceleration data); a frame is cut into row tiles, tile k to member k % n, every member
// traces its tiles into a compact slab, the slabs travel to member 0, which reassembles the frame and hands it to
// the host.  Two transports move the bytes between devices:
//   rccl   ncclCommInitAll over the members' devices; ncclBroadcast of the scene arrays, grouped ncclSend / ncclRecv
//          of the slabs (xGMI is point to point: n - 1 concurrent sends into member 0 use n - 1 links).  librccl is
//          opened at run time (dlopen), so libhvrt itself carries no dependency on it and a process that already
//          holds a copy -- torch's -- shares that copy.
//   copy   hipMemcpyAsync between the devices (peer access where the hardware offers it), ordered with events.
//          Taken when a device appears twice in the list (two contexts on one GPU: how the choreography is tested
//          on a one-GPU machine) or when HVRT_GROUP_TRANSPORT=copy asks for it.
// No counterpart in the reference (one thread, one CPU); the seam is the tracer factory, TraceImages.cpp:464-484.
#include "hvrt_internal.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

namespace hvrt {
// hvrt_api.hip: hvrt_render_part_device plus the number of pixels whose (sample-0) ray hit, added to *d_hits (u64)
int render_part_counting_hits(hvrt_ctx* c, const hvrt_camera_frame* frame, int width, int height, int spp, uint32_t seed,
                              int tile_rows, int part, int n_parts, void* d_rgb, void* d_counters, void* d_hits);
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

struct Member {
    int device = 0;
    hvrt_ctx* ctx = nullptr;
    hipStream_t stream = nullptr;            // the context's own stream
    hipEvent_t done = nullptr;               // this member's slab is complete
    float* slab = nullptr; size_t slab_cap = 0;                 // floats
    unsigned long long* counters = nullptr;  // voxels, tests, hits
    ncclComm_t comm = nullptr;
};

}  // namespace

struct hvrt_group {
    std::vector<Member> m;
    std::string error;
    enum { NONE, COPY, RCCL } transport = NONE;
    Rccl rccl;
    float* parts = nullptr; size_t parts_cap = 0;      // member 0: n slabs
    float* frame = nullptr; size_t frame_cap = 0;      // member 0: the reassembled frame
    double last_exchange_bytes = 0.0;
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

template <typename T>
int grow(hvrt_group* g, T*& p, size_t& cap, size_t need) {
    if (need <= cap && p) return HVRT_OK;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    G_HIP(g, hipMalloc((void**)&p, (need ? need : 1) * sizeof(T)));
    cap = need;
    return HVRT_OK;
}

}  // namespace

extern "C" {

const char* hvrt_group_last_error(const hvrt_group* g) { return g ? g->error.c_str() : g_group_create_error.c_str(); }

void hvrt_group_destroy(hvrt_group* g) {
    if (!g) return;
    for (Member& mb : g->m) {
        if (!mb.ctx) continue;
        (void)hipSetDevice(mb.device);
        (void)hvrt_synchronize(mb.ctx);
        if (mb.comm) (void)g->rccl.CommDestroy(mb.comm);
        if (mb.slab) (void)hipFree(mb.slab);
        if (mb.counters) (void)hipFree(mb.counters);
        if (mb.done) (void)hipEventDestroy(mb.done);
    }
    if (!g->m.empty()) {
        (void)hipSetDevice(g->m[0].device);
        if (g->parts) (void)hipFree(g->parts);
        if (g->frame) (void)hipFree(g->frame);
    }
    for (Member& mb : g->m) if (mb.ctx) hvrt_destroy(mb.ctx);
    delete g;
}

int hvrt_group_create(hvrt_group** out, const int* device_ids, int n_devices) {
    if (!out) return HVRT_ERR_INVALID;
    *out = nullptr;
    if (!device_ids || n_devices < 1 || n_devices > 64) return gfail(nullptr, HVRT_ERR_INVALID, "group_create: 1..64 devices");
    hvrt_group* g = new hvrt_group;
    auto bail = [&](int code) { g_group_create_error = g->error; hvrt_group_destroy(g); return code; };
    bool repeated = false;
    for (int i = 0; i < n_devices; ++i)
        for (int j = 0; j < i; ++j) repeated |= device_ids[i] == device_ids[j];
    g->m.resize(n_devices);
    for (int i = 0; i < n_devices; ++i) {
        Member& mb = g->m[i];
        mb.device = device_ids[i];
        if (hvrt_create(&mb.ctx, mb.device) != HVRT_OK) {
            gfail(g, HVRT_ERR_HIP, "group_create: device %d: %s", mb.device, hvrt_last_error(nullptr));
            return bail(HVRT_ERR_HIP);
        }
        mb.stream = (hipStream_t)hvrt_stream(mb.ctx);
        hipError_t e = hipSetDevice(mb.device);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&mb.done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipMalloc((void**)&mb.counters, 3 * sizeof(unsigned long long));
        if (e != hipSuccess) { gfail(g, HVRT_ERR_HIP, "group_create: device %d: %s", mb.device, hipGetErrorString(e)); return bail(HVRT_ERR_HIP); }
    }
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
    *out = g;
    return HVRT_OK;
}

int hvrt_group_size(const hvrt_group* g) { return g ? (int)g->m.size() : 0; }

hvrt_ctx* hvrt_group_member(hvrt_group* g, int i) { return (g && i >= 0 && i < (int)g->m.size()) ? g->m[i].ctx : nullptr; }

const char* hvrt_group_transport(const hvrt_group* g) {
    if (!g) return "";
    return g->transport == hvrt_group::RCCL ? "rccl" : g->transport == hvrt_group::COPY ? "copy" : "none";
}

int hvrt_group_replicate_scene(hvrt_group* g) {
    if (!g) return HVRT_ERR_INVALID;
    const int n = (int)g->m.size();
    Member& root = g->m[0];
    hvrt_grid_info gi;
    G_CTX(g, 0, hvrt_get_grid_info(root.ctx, &gi));
    void *aos = nullptr, *off = nullptr, *idx = nullptr;
    G_CTX(g, 0, hvrt_get_device_arrays(root.ctx, &aos, &off, &idx));
    G_CTX(g, 0, hvrt_synchronize(root.ctx));
    if (g->transport == hvrt_group::NONE) return HVRT_OK;
    const size_t b_tri = (size_t)gi.n_triangles * 48, b_off = ((size_t)gi.n_voxels + 1) * 4, b_idx = (size_t)gi.n_refs * 4;
    if (g->transport == hvrt_group::COPY) {
        // the members copy straight out of member 0's arrays (a device-to-device copy across devices)
        for (int i = 1; i < n; ++i) {
            G_HIP(g, hipSetDevice(g->m[i].device));
            G_CTX(g, i, hvrt_set_mesh_device(g->m[i].ctx, aos, gi.n_triangles));
            G_CTX(g, i, hvrt_set_grid_csr_device(g->m[i].ctx, gi.nx, gi.ny, gi.nz, gi.min_bound, gi.max_bound, gi.voxel_size, off, idx, gi.n_refs));
        }
        return HVRT_OK;
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
        for (int i = 1; i < n; ++i) {
            G_HIP(g, hipSetDevice(g->m[i].device));
            G_HIP(g, hipStreamSynchronize(g->m[i].stream));
            G_CTX(g, i, hvrt_set_mesh_device(g->m[i].ctx, st_tri[i], gi.n_triangles));
            G_CTX(g, i, hvrt_set_grid_csr_device(g->m[i].ctx, gi.nx, gi.ny, gi.nz, gi.min_bound, gi.max_bound, gi.voxel_size, st_off[i], st_idx[i], gi.n_refs));
        }
        G_HIP(g, hipSetDevice(root.device));
        G_HIP(g, hipStreamSynchronize(root.stream));
        return HVRT_OK;
    }();
    release();
    return rc;
}

int hvrt_group_render(hvrt_group* g, const hvrt_camera_frame* frame, int width, int height, int spp, uint32_t seed,
                      float* rgb, uint64_t counters[2], uint64_t* hits) {
    if (!g) return HVRT_ERR_INVALID;
    if (!frame || !rgb) return gfail(g, HVRT_ERR_INVALID, "group_render: null frame or output");
    if (width < 1 || height < 1 || (uint64_t)width * height > 0x7FFFFFFFull) return gfail(g, HVRT_ERR_INVALID, "group_render: bad image size %d x %d", width, height);
    const int n = (int)g->m.size();
    constexpr int TILE_ROWS = 16;
    int slab_rows = 0;
    for (int i = 0; i < n; ++i) { const int r = hvrt_part_rows(height, TILE_ROWS, i, n); slab_rows = r > slab_rows ? r : slab_rows; }
    const size_t slab_floats = (size_t)slab_rows * width * 3, frame_floats = (size_t)height * width * 3;
    Member& root = g->m[0];
    // every member traces its row tiles into its slab
    for (int i = 0; i < n; ++i) {
        Member& mb = g->m[i];
        G_HIP(g, hipSetDevice(mb.device));
        if (int r = grow(g, mb.slab, mb.slab_cap, slab_floats)) return r;
        G_HIP(g, hipMemsetAsync(mb.counters, 0, 3 * sizeof(unsigned long long), mb.stream));
        G_CTX(g, i, hvrt::render_part_counting_hits(mb.ctx, frame, width, height, spp, seed, TILE_ROWS, i, n, mb.slab, mb.counters, mb.counters + 2));
        G_HIP(g, hipEventRecord(mb.done, mb.stream));
    }
    const float* result = root.slab;       // one member: its slab is the frame
    g->last_exchange_bytes = 0.0;
    if (n > 1) {
        G_HIP(g, hipSetDevice(root.device));
        if (int r = grow(g, g->parts, g->parts_cap, slab_floats * n)) return r;
        if (int r = grow(g, g->frame, g->frame_cap, frame_floats)) return r;
        auto floats_of = [&](int i) { return (size_t)hvrt_part_rows(height, TILE_ROWS, i, n) * width * 3; };
        // member 0's own slab never leaves its device
        if (floats_of(0)) G_HIP(g, hipMemcpyAsync(g->parts, root.slab, floats_of(0) * 4, hipMemcpyDeviceToDevice, root.stream));
        if (g->transport == hvrt_group::RCCL) {
            G_NCCL(g, g->rccl.GroupStart());
            ncclResult_t first = ncclSuccess;
            for (int i = 1; i < n && first == ncclSuccess; ++i) {
                if (!floats_of(i)) continue;
                first = g->rccl.Send(g->m[i].slab, floats_of(i), ncclFloat, 0, g->m[i].comm, g->m[i].stream);
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
                G_HIP(g, hipMemcpyAsync(g->parts + slab_floats * i, g->m[i].slab, floats_of(i) * 4, hipMemcpyDefault, root.stream));
                g->last_exchange_bytes += floats_of(i) * 4.0;
            }
        }
        G_CTX(g, 0, hvrt_assemble_parts_device(root.ctx, g->parts, width, height, TILE_ROWS, n, slab_rows, 12, g->frame));
        result = g->frame;
    }
    G_HIP(g, hipSetDevice(root.device));
    G_CTX(g, 0, hvrt_download(root.ctx, rgb, result, frame_floats * 4));
    uint64_t sum[3] = { 0, 0, 0 };
    for (int i = 0; i < n; ++i) {
        unsigned long long h[3];
        G_HIP(g, hipSetDevice(g->m[i].device));
        G_HIP(g, hipMemcpyAsync(h, g->m[i].counters, sizeof h, hipMemcpyDeviceToHost, g->m[i].stream));
        G_HIP(g, hipStreamSynchronize(g->m[i].stream));
        for (int k = 0; k < 3; ++k) sum[k] += h[k];
    }
    if (counters) { counters[0] = sum[0]; counters[1] = sum[1]; }
    if (hits) *hits = sum[2];
    return HVRT_OK;
}

}  // extern "C"
