// HipVoxelRayTracer.cpp -- see the header.  Plain C++17, links libhvrt; no HIP headers.
#include "HipVoxelRayTracer.hpp"

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>

#include "Logger.hpp"
#include "hvrt.h"

using namespace cg_datastructures;

#include <type_traits>

// What the C ABI and the .voxgrid format rely on, checked against whichever headers are in use (this repository's
// rt_types.hpp / IRayTracer.hpp, or the reference's own with -DHVRT_USE_REFERENCE_HEADERS): SURVEY.md 8(a) a1-a3.
static_assert(sizeof(Triangle) == 48 && sizeof(Vec3) == 12, "Triangle must be the 12-float AoS record");
static_assert(sizeof(Voxel) == 28, "Voxel: bool + 3 pad, u32 start, u32 count, f32 density, Vec3 color (Voxels.h:17-23)");
static_assert(sizeof(Ray) == 24 && sizeof(RayHit) == 52 && sizeof(Camera) == 44, "RayTracerCommon.h:9-43 layouts");
static_assert(sizeof(RayHit) == sizeof(hvrt_hit), "hvrt_hit mirrors RayHit");
static_assert(std::is_abstract<IRayTracer>::value && std::is_base_of<IRayTracer, HipVoxelRayTracer>::value &&
              !std::is_abstract<HipVoxelRayTracer>::value, "HipVoxelRayTracer implements every pure virtual of IRayTracer");

namespace {
[[noreturn]] void raise(hvrt_ctx* c, const char* what) {
    throw std::runtime_error(std::string(what) + ": " + hvrt_last_error(c));
}
inline void ck(hvrt_ctx* c, int rc, const char* what) {
    if (rc != HVRT_OK) raise(c, what);
}
}  // namespace

// ------------------------------------------------------------------------------------------ HipScene

namespace {

// Contexts and device groups outlive the scenes that use them.  The reference constructs a tracer per image
// (TraceImages.cpp:215-229) and HipVoxelRayTracer(const VoxelGrid&) makes a scene per tracer: a context per scene would pay
// hvrt_create -- 100-200 ms of runtime start-up and queue warm-up per device, plus the communicator of a group -- for
// every image.  A scene takes an idle context (or group) of the same device list from here, cleared of its last scene
// (hvrt_clear_scene: the device buffers stay, grow-only), and hands it back when it dies.  HVRT_CONTEXT_CACHE=0: no
// pool, every scene creates and destroys its own.
class ContextPool {
public:
    static ContextPool& get() {
        (void)hvrt_device_count();      // the HIP runtime starts BEFORE this object exists, so it is torn down after it
        static ContextPool p;
        return p;
    }
    hvrt_ctx* takeContext(int device) {
        std::lock_guard<std::mutex> lk(mu_);
        auto it = ctxs_.find(device);
        if (it == ctxs_.end()) return nullptr;
        hvrt_ctx* c = it->second;
        ctxs_.erase(it);
        return c;
    }
    hvrt_group* takeGroup(const std::string& key) {
        std::lock_guard<std::mutex> lk(mu_);
        auto it = groups_.find(key);
        if (it == groups_.end()) return nullptr;
        hvrt_group* g = it->second;
        groups_.erase(it);
        return g;
    }
    void give(int device, hvrt_ctx* c) {
        if (!enabled_) { hvrt_destroy(c); return; }
        std::lock_guard<std::mutex> lk(mu_);
        ctxs_.emplace(device, c);
    }
    void give(const std::string& key, hvrt_group* g) {
        if (!enabled_) { hvrt_group_destroy(g); return; }
        std::lock_guard<std::mutex> lk(mu_);
        groups_.emplace(key, g);
    }
    void clear() {
        std::lock_guard<std::mutex> lk(mu_);
        for (auto& kv : ctxs_) hvrt_destroy(kv.second);
        for (auto& kv : groups_) hvrt_group_destroy(kv.second);
        ctxs_.clear();
        groups_.clear();
    }
    ~ContextPool() { clear(); }

private:
    ContextPool() {
        if (const char* e = std::getenv("HVRT_CONTEXT_CACHE")) enabled_ = std::atoi(e) != 0;
    }
    std::mutex mu_;
    std::multimap<int, hvrt_ctx*> ctxs_;
    std::multimap<std::string, hvrt_group*> groups_;
    bool enabled_ = true;
};

}  // namespace

void HipScene::releaseCachedContexts() { ContextPool::get().clear(); }

HipScene::HipScene(int device_id) {
    // which devices: HVRT_DEVICES ("0,1,2", "all") asks for a group; unset, the scene lives on device_id alone, however
    // many GPUs are visible (one process per GPU -- rank r constructing HipScene(r) -- is the other multi-GPU model and
    // must get exactly its device).  A device may be listed twice (two contexts on one GPU, the copy transport).
    std::vector<int> ids;
    if (const char* e = std::getenv("HVRT_DEVICES")) {
        const std::string v(e);
        if (v == "all") {
            const int n = hvrt_device_count();
            for (int i = 0; i < n; ++i) ids.push_back(i);
        } else {
            size_t p = 0;
            while (p < v.size()) {
                size_t q = v.find(',', p);
                if (q == std::string::npos) q = v.size();
                if (q > p) ids.push_back(std::atoi(v.substr(p, q - p).c_str()));
                p = q + 1;
            }
        }
    }
    if (ids.empty()) ids.push_back(device_id);
    if (const char* e = std::getenv("HVRT_GROUP_MIN_PIXELS")) group_min_pixels_ = (size_t)std::strtoull(e, nullptr, 10);
    ContextPool& pool = ContextPool::get();
    if (ids.size() > 1) {
        for (int id : ids) pool_key_ += std::to_string(id) + ",";
        group_ = pool.takeGroup(pool_key_);
        if (group_) {
            for (int i = 0; i < hvrt_group_size(group_); ++i) (void)hvrt_clear_scene(hvrt_group_member(group_, i));
            LOG_INFO("[HipVoxelGrid] scene spans {} devices ({} transport; group taken from the pool)", ids.size(), hvrt_group_transport(group_));
        } else {
            if (hvrt_group_create(&group_, ids.data(), (int)ids.size()) != HVRT_OK)
                throw std::runtime_error(std::string("hvrt_group_create: ") + hvrt_group_last_error(nullptr));
            hvrt_group_timings gt{};
            hvrt_group_get_timings(group_, &gt);
            LOG_INFO("[HipVoxelGrid] scene spans {} devices ({} transport; group created in {:.1f} ms)", ids.size(),
                     hvrt_group_transport(group_), gt.create_ms);
        }
        ctx_ = hvrt_group_member(group_, 0);
        device_ = ids[0];
        return;
    }
    device_ = ids[0];
    ctx_ = pool.takeContext(device_);
    if (ctx_) { (void)hvrt_clear_scene(ctx_); return; }
    if (hvrt_create(&ctx_, device_) != HVRT_OK) raise(nullptr, "hvrt_create");
}

HipScene::~HipScene() {
    if (group_) ContextPool::get().give(pool_key_, group_);     // owns the contexts
    else if (ctx_) ContextPool::get().give(device_, ctx_);
}

void HipScene::replicate() {
    // (lazy: the other members receive the scene when the first frame large enough for the group is asked for --
    // the reference's default 800x600 frame never is, and its flow builds a scene per image)
    if (!group_ || replicated_) return;
    if (hvrt_group_replicate_scene(group_) != HVRT_OK)
        throw std::runtime_error(std::string("hvrt_group_replicate_scene: ") + hvrt_group_last_error(group_));
    replicated_ = true;
    hvrt_group_timings gt{};
    hvrt_group_get_timings(group_, &gt);
    LOG_INFO("[HipVoxelGrid] scene replicated to {} devices in {:.1f} ms", hvrt_group_size(group_), gt.replicate_ms);
}

bool HipScene::spansDevices(int width, int height) const {
    return group_ && (size_t)width * (size_t)height >= group_min_pixels_;
}

int HipScene::deviceCount() const { return group_ ? hvrt_group_size(group_) : 1; }

void HipScene::setTriangles(const std::vector<Triangle>& tris) { setTriangles(std::vector<Triangle>(tris)); }

void HipScene::setTriangles(std::vector<Triangle>&& tris_in) {
    tris_ = std::move(tris_in);
    const std::vector<Triangle>& tris = tris_;
    replicated_ = false;
    ck(ctx_, hvrt_set_mesh(ctx_, tris.empty() ? nullptr : &tris[0].v0.x, (uint32_t)tris.size()), "hvrt_set_mesh");
}

void HipScene::buildGrid(int nx, int ny, int nz, const Vec3& mn, const Vec3& mx, const Vec3& vs) {
    ck(ctx_, hvrt_build_grid(ctx_, nx, ny, nz, &mn.x, &mx.x, &vs.x), "hvrt_build_grid");
    replicated_ = false;
}

void HipScene::computeBounds(const float* xyz, size_t n, Vec3& mn, Vec3& mx) {
    ck(ctx_, hvrt_compute_bounds(ctx_, xyz, n, &mn.x, &mx.x), "hvrt_compute_bounds");
}

void HipScene::setGrid(const VoxelGrid& g) {
    setTriangles(g.triangles);
    const size_t nvox = (size_t)g.nx * g.ny * g.nz;
    size_t refs = 0;
    for (int x = 0; x < g.nx; ++x)
        for (int y = 0; y < g.ny; ++y)
            for (int z = 0; z < g.nz; ++z) refs += g.voxels[x][y][z].triangle_count;
    if (refs > 0 && g.triangle_indices.empty()) {
        // a grid read back from a .voxgrid file has no index array (TraceImages.cpp:447-448): the CSR is
        // a deterministic function of (triangles, bounds, voxel size), so rebuild it on the GPU.
        LOG_WARN("[HipVoxelGrid] grid has no triangle_indices; rebuilding the CSR on the GPU");
        buildGrid(g.nx, g.ny, g.nz, g.minBound, g.maxBound, g.voxelSize);
        return;
    }
    // voxels[x][y][z] -> flat order ix + nx*(iy + ny*iz); copy each list so that any start layout works
    std::vector<uint32_t> offsets(nvox + 1), indices;
    indices.reserve(refs);
    size_t f = 0;
    for (int z = 0; z < g.nz; ++z)
        for (int y = 0; y < g.ny; ++y)
            for (int x = 0; x < g.nx; ++x, ++f) {
                const Voxel& v = g.voxels[x][y][z];
                offsets[f] = (uint32_t)indices.size();
                if ((size_t)v.triangle_start_idx + v.triangle_count > g.triangle_indices.size())
                    throw std::runtime_error("VoxelGrid: voxel list runs past triangle_indices");
                indices.insert(indices.end(), g.triangle_indices.begin() + v.triangle_start_idx,
                               g.triangle_indices.begin() + v.triangle_start_idx + v.triangle_count);
            }
    offsets[nvox] = (uint32_t)indices.size();
    ck(ctx_, hvrt_set_grid_csr(ctx_, g.nx, g.ny, g.nz, &g.minBound.x, &g.maxBound.x, &g.voxelSize.x, offsets.data(),
                               indices.empty() ? nullptr : indices.data()), "hvrt_set_grid_csr");
    replicated_ = false;
}

VoxelGrid HipScene::downloadGrid() const {
    hvrt_grid_info gi;
    ck(ctx_, hvrt_get_grid_info(ctx_, &gi), "hvrt_get_grid_info");
    std::vector<uint32_t> offsets(gi.n_voxels + 1), indices(gi.n_refs);
    ck(ctx_, hvrt_get_csr(ctx_, offsets.data(), indices.empty() ? nullptr : indices.data()), "hvrt_get_csr");
    std::vector<std::vector<std::vector<Voxel>>> vox(gi.nx, std::vector<std::vector<Voxel>>(gi.ny, std::vector<Voxel>(gi.nz)));
    size_t f = 0;
    for (int z = 0; z < gi.nz; ++z)
        for (int y = 0; y < gi.ny; ++y)
            for (int x = 0; x < gi.nx; ++x, ++f) {
                Voxel& v = vox[x][y][z];
                v.triangle_start_idx = offsets[f];
                v.triangle_count = offsets[f + 1] - offsets[f];
                v.occupied = v.triangle_count > 0;
            }
    Vec3 mn{ gi.min_bound[0], gi.min_bound[1], gi.min_bound[2] }, mx{ gi.max_bound[0], gi.max_bound[1], gi.max_bound[2] };
    Vec3 vs{ gi.voxel_size[0], gi.voxel_size[1], gi.voxel_size[2] };
    std::vector<Triangle> tris = tris_;
    std::vector<unsigned int> idx(indices.begin(), indices.end());
    return VoxelGrid(std::move(vox), std::move(idx), gi.nx, gi.ny, gi.nz, mn, mx, vs, std::move(tris));
}

double HipScene::lastBuildMs() const {
    hvrt_timings t;
    hvrt_get_timings(ctx_, &t);
    return t.build_total_ms;
}
double HipScene::lastRenderMs() const {
    hvrt_timings t;
    hvrt_get_timings(ctx_, &t);
    return t.render_ms;
}

// --------------------------------------------------------------------------------- HipVoxelRayTracer

HipVoxelRayTracer::HipVoxelRayTracer(const VoxelGrid& grid, int, int) : scene_(std::make_shared<HipScene>(0)) {
    const auto t0 = std::chrono::steady_clock::now();
    LOG_INFO("[{}] Uploading voxel grid {}x{}x{} ({} triangles) to the GPU", getMethodName(), grid.nx, grid.ny, grid.nz,
             grid.triangles.size());
    scene_->setGrid(grid);
    LOG_INFO("[{}] grid resident after {:.1f} ms", getMethodName(),
             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
}

HipVoxelRayTracer::HipVoxelRayTracer(std::shared_ptr<HipScene> scene) : scene_(std::move(scene)) {
    if (!scene_) throw std::runtime_error("HipVoxelRayTracer: null scene");
}

std::vector<Vec3> HipVoxelRayTracer::render(const Camera& cam, int width, int height) {
    if (width <= 0 || height <= 0) throw std::runtime_error("render: width and height must be positive");
    hvrt_camera_frame fr;
    hvrt_make_camera_frame(&cam.position.x, &cam.lookAt.x, &cam.up.x, cam.fov, cam.aspectRatio, &fr);
    std::vector<Vec3> frame((size_t)width * height);
    // the hit-id image is fetched only when the caller asked for it (keepHitIds): it is a third more bytes over
    // PCIe, and the reference's render() returns colours only; the hit COUNT it logs comes from the device
    if (keep_hit_ids_) hit_ids_.assign((size_t)width * height, HVRT_MISS); else hit_ids_.clear();
    uint64_t counters[2] = { 0, 0 };
    uint64_t hits = 0;
    double kernel_ms = 0.0;
    if (scene_->spansDevices(width, height)) {
        // row tiles over every device of the scene, each member copying its own tiles into `frame` (include/hvrt.h, group section)
        scene_->replicate();
        LOG_INFO("[{}] Rendering {}x{} image on {} devices...", getMethodName(), width, height, scene_->deviceCount());
        if (hvrt_group_render(scene_->group(), &fr, width, height, 1, 0, &frame[0].x, keep_hit_ids_ ? hit_ids_.data() : nullptr,
                              nullptr, counters, &hits) != HVRT_OK)
            throw std::runtime_error(std::string("hvrt_group_render: ") + hvrt_group_last_error(scene_->group()));
        hvrt_group_timings gt{};
        hvrt_group_get_timings(scene_->group(), &gt);
        kernel_ms = gt.render_ms;
        LOG_INFO("[{}] group render {:.3f} ms in all (longest traversal {:.3f} ms, longest copy-out {:.3f} ms)", getMethodName(),
                 gt.total_ms, gt.render_ms, gt.download_ms);
    } else {
        LOG_INFO("[{}] Rendering {}x{} image...", getMethodName(), width, height);
        ck(scene_->handle(), hvrt_render(scene_->handle(), &fr, width, height, 1, 0, &frame[0].x,
                                         keep_hit_ids_ ? hit_ids_.data() : nullptr, nullptr, counters), "hvrt_render");
        ck(scene_->handle(), hvrt_last_hit_count(scene_->handle(), &hits), "hvrt_last_hit_count");
        kernel_ms = scene_->lastRenderMs();
    }
    voxels_ = counters[0];
    tests_ = counters[1];
    const double total = (double)width * height;
    // same statistics lines as OctreeLikeRayTracer.cpp:57-65 (counted once per ray, not twice)
    LOG_INFO("[{}] Rendering complete! kernel {:.3f} ms", getMethodName(), kernel_ms);
    LOG_INFO("[{}] Ray hits: {} / {} ({:.1f}%)", getMethodName(), hits, (size_t)total, 100.0 * hits / total);
    LOG_INFO("[{}] Voxels traversed: {}, Avg per ray: {:.1f}", getMethodName(), voxels_, voxels_ / total);
    LOG_INFO("[{}] Triangle tests: {}, Avg per ray: {:.1f}", getMethodName(), tests_, tests_ / total);
    return frame;
}

std::vector<std::vector<unsigned char>> HipVoxelRayTracer::renderViewsRgb8(const std::vector<Camera>& cams, int width, int height, bool wantHitIds) {
    if (width <= 0 || height <= 0) throw std::runtime_error("render: width and height must be positive");
    if (cams.empty()) return {};
    std::vector<hvrt_camera_frame> fr(cams.size());
    for (size_t i = 0; i < cams.size(); ++i)
        hvrt_make_camera_frame(&cams[i].position.x, &cams[i].lookAt.x, &cams[i].up.x, cams[i].fov, cams[i].aspectRatio, &fr[i]);
    const size_t npx = (size_t)width * height;
    std::vector<unsigned char> all(npx * 3 * cams.size());
    std::vector<unsigned int> hits(wantHitIds ? npx * cams.size() : 0);
    uint64_t counters[2] = { 0, 0 };
    LOG_INFO("[{}] Rendering {} view{} of {}x{} in one launch...", getMethodName(), cams.size(), cams.size() == 1 ? "" : "s", width, height);
    ck(scene_->handle(), hvrt_render_views_rgb8(scene_->handle(), fr.data(), (int)fr.size(), width, height, all.data(),
                                                wantHitIds ? hits.data() : nullptr, counters), "hvrt_render_views_rgb8");
    voxels_ = counters[0];
    tests_ = counters[1];
    if (wantHitIds) hit_ids_.assign(hits.end() - (std::ptrdiff_t)npx, hits.end()); else hit_ids_.clear();
    const double total = (double)npx * cams.size();
    LOG_INFO("[{}] Rendering complete! kernel {:.3f} ms", getMethodName(), scene_->lastRenderMs());
    LOG_INFO("[{}] Voxels traversed: {}, Avg per ray: {:.1f}", getMethodName(), voxels_, voxels_ / total);
    LOG_INFO("[{}] Triangle tests: {}, Avg per ray: {:.1f}", getMethodName(), tests_, tests_ / total);
    std::vector<std::vector<unsigned char>> out(cams.size());
    for (size_t i = 0; i < cams.size(); ++i) out[i].assign(all.begin() + (std::ptrdiff_t)(i * npx * 3), all.begin() + (std::ptrdiff_t)((i + 1) * npx * 3));
    return out;
}

void HipVoxelRayTracer::setNearestHit(bool on) {
    // (a group: every member, or the frame would be striped -- hvrt_group_render also hands member 0's mode on)
    if (scene_->group()) {
        if (hvrt_group_set_mode(scene_->group(), on ? 1 : 0) != HVRT_OK)
            throw std::runtime_error(std::string("hvrt_group_set_mode: ") + hvrt_group_last_error(scene_->group()));
    } else {
        ck(scene_->handle(), hvrt_set_mode(scene_->handle(), on ? 1 : 0), "hvrt_set_mode");
    }
}

std::vector<RayHit> HipVoxelRayTracer::traceRays(const std::vector<Ray>& rays) {
    const size_t n = rays.size();
    std::vector<float> o(3 * n), d(3 * n);
    for (size_t i = 0; i < n; ++i) {
        std::memcpy(&o[3 * i], &rays[i].origin.x, 12);
        std::memcpy(&d[3 * i], &rays[i].direction.x, 12);
    }
    std::vector<hvrt_hit> h(n);
    uint64_t counters[2] = { 0, 0 };
    ck(scene_->handle(), hvrt_trace(scene_->handle(), o.data(), d.data(), (uint32_t)n, h.data(), counters), "hvrt_trace");
    voxels_ += counters[0];
    tests_ += counters[1];
    std::vector<RayHit> out(n);
    for (size_t i = 0; i < n; ++i) {
        RayHit& r = out[i];
        r.hit = h[i].hit != 0;
        r.t = h[i].t;
        r.point = { h[i].point[0], h[i].point[1], h[i].point[2] };
        r.origin = { h[i].origin[0], h[i].origin[1], h[i].origin[2] };
        r.normal = { h[i].normal[0], h[i].normal[1], h[i].normal[2] };
        r.triangleIndex = h[i].triangle_index;
        r.voxelIndex = h[i].voxel_index;
    }
    return out;
}

RayHit HipVoxelRayTracer::traceRay(const Ray& ray) { return traceRays({ ray })[0]; }
