// TraceImages.cpp -- host pipeline around the GPU hot path; see TraceImages.hpp.
#include "TraceImages.hpp"

#include <algorithm>
#include <chrono>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <stdexcept>

#include "ImageWriter.hpp"
#include "Logger.hpp"
#include "hvrt.h"

using namespace cg_datastructures;
namespace fs = std::filesystem;

namespace {
double seconds_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
std::string lower_ext(const std::string& path) {
    std::string e = fs::path(path).extension().string();
    std::transform(e.begin(), e.end(), e.begin(), [](unsigned char c) { return (char)std::tolower(c); });
    return e;
}
}  // namespace

std::string TraceImages::getRayTracingMethodName(RayTracingMethod m) {
    switch (m) {
        case RayTracingMethod::VOXEL_DDA: return "Voxel Grid DDA";
        case RayTracingMethod::ART: return "ARTS (3DDDA)";
        case RayTracingMethod::OCTREE: return "Octree";
        case RayTracingMethod::HIP_VOXEL: return "HIP Voxel Grid (MI355X)";
    }
    return "Unknown";
}

// ---------------------------------------------------------------------------------------- scene build

VoxelGrid TraceImages::buildSceneOnGpu(const stl_reader::StlMesh<float, unsigned int>& mesh, int nx, int ny, int nz,
                                       HipScene& scene) {
    // computeMeshBounds over the de-duplicated vertex list (SceneCreator.cpp:49-84) -- on the GPU
    Vec3 mn{}, mx{};
    scene.computeBounds(mesh.raw_coords(), mesh.num_vrts(), mn, mx);
    // computeVoxelizationParams (SceneCreator.cpp:91-122)
    const float vs = hvrt_voxel_size(&mn.x, &mx.x, nx, ny, nz);
    // buildTriangleList (SceneCreator.cpp:129-170): gather corners + file normal
    std::vector<Triangle> tris(mesh.num_tris());
    for (size_t t = 0; t < tris.size(); ++t) {
        std::memcpy(&tris[t].v0, mesh.tri_corner_coords(t, 0), 12);
        std::memcpy(&tris[t].v1, mesh.tri_corner_coords(t, 1), 12);
        std::memcpy(&tris[t].v2, mesh.tri_corner_coords(t, 2), 12);
        std::memcpy(&tris[t].normal, mesh.tri_normal(t), 12);
    }
    const size_t n_tris = tris.size();
    scene.setTriangles(std::move(tris));
    // voxelizeTrianglesDirect (SceneCreator.cpp:197-281) -- on the GPU
    scene.buildGrid(nx, ny, nz, mn, mx, Vec3{ vs, vs, vs });
    LOG_INFO("Voxel grid built on the GPU in {:.3f} ms ({} triangles)", scene.lastBuildMs(), n_tris);
    return VoxelGrid({}, {}, nx, ny, nz, mn, mx, Vec3{ vs, vs, vs }, {});
}

TraceImages::Loaded TraceImages::loadOrGenerate(const std::string& path, int nx, int ny, int nz, RayTracingMethod method) {
    const std::string ext = lower_ext(path);
    if (ext != ".voxgrid" && ext != ".stl") {
        LOG_ERROR("Unsupported file format: {}. Expected .stl or .voxgrid", ext);
        throw std::runtime_error("Unsupported file format: " + ext);      // TraceImages.cpp:351-352
    }
    if (method != RayTracingMethod::HIP_VOXEL)
        throw std::runtime_error("method '" + getRayTracingMethodName(method) +
                                 "' is the reference's CPU tracer, which this library does not contain; use HIP_VOXEL");
    Loaded l;
    auto t0 = std::chrono::steady_clock::now();
    l.scene = std::make_shared<HipScene>(0);
    lastPhases.context_ms = 1e3 * seconds_since(t0);
    t0 = std::chrono::steady_clock::now();
    if (ext == ".voxgrid") {
        LOG_INFO("Loading pre-saved voxel grid from: {}", path);
        l.grid = VoxelGrid::loadFromFile(path);
        lastPhases.load_ms = 1e3 * seconds_since(t0);
        t0 = std::chrono::steady_clock::now();
        l.scene->setGrid(l.grid);
    } else {
        LOG_INFO("Processing STL file: {}", path);
        stl_reader::StlMesh<float, unsigned int> mesh(path);
        LOG_INFO("Loaded mesh with {} triangles", mesh.num_tris());
        lastPhases.load_ms = 1e3 * seconds_since(t0);
        t0 = std::chrono::steady_clock::now();
        l.grid = buildSceneOnGpu(mesh, nx, ny, nz, *l.scene);
    }
    lastPhases.build_ms = 1e3 * seconds_since(t0);
    return l;
}

// ---------------------------------------------------------------------------------------- cameras

Camera TraceImages::setupCamera(const VoxelGrid& g) { return setupCameraWithView(g, CameraView::FRONT); }

Camera TraceImages::setupCameraWithView(const VoxelGrid& g, CameraView view) {
    // TraceImages.cpp:125-197: distance = 2.5 * largest extent, fov 60, aspect 800/600
    const Vec3 c = g.center;
    const float sx = g.maxBound.x - g.minBound.x, sy = g.maxBound.y - g.minBound.y, sz = g.maxBound.z - g.minBound.z;
    const float dist = std::max({ sx, sy, sz }) * 2.5f;
    Vec3 pos = c, up{ 0.0f, 1.0f, 0.0f };
    switch (view) {
        case CameraView::FRONT: pos.z = c.z - dist; break;
        case CameraView::SIDE: pos.x = c.x + dist; break;
        case CameraView::TOP: pos.y = c.y + dist; up = { 0.0f, 0.0f, 1.0f }; break;
    }
    return Camera(pos, c, up, 60.0f, 800.0f / 600.0f);
}

// ---------------------------------------------------------------------------------------- factory

std::unique_ptr<IRayTracer> TraceImages::createRayTracer(VoxelGrid& voxelGrid, RayTracingMethod method) {
    switch (method) {
        case RayTracingMethod::HIP_VOXEL:
            return std::make_unique<HipVoxelRayTracer>(voxelGrid);
        default:
            throw std::runtime_error("method '" + getRayTracingMethodName(method) +
                                     "' is the reference's CPU tracer, which this library does not contain; use HIP_VOXEL");
    }
}

// ---------------------------------------------------------------------------------------- pipeline

void TraceImages::renderAndSave(Loaded& l, const Camera& cam, const std::string& filename, RayTracingMethod) {
    HipVoxelRayTracer tracer(l.scene);
    tracer.setNearestHit(nearestHit);
    tracer.keepHitIds(keepHitIds);
    auto t0 = std::chrono::steady_clock::now();
    if (!keepPixels && !l.scene->spansDevices(imageWidth, imageHeight)) {
        // the file holds 8-bit pixels: quantise where the frame is (ImageSaver.h:28-33 on the GPU) and fetch a quarter of the bytes
        const auto frames = tracer.renderViewsRgb8({ cam }, imageWidth, imageHeight, keepHitIds);
        lastPixels.clear();
        lastHitIds = tracer.lastHitIds();
        const double s = seconds_since(t0);
        lastPhases.render_ms = 1e3 * s;
        LOG_INFO("Rendered {} pixels in {:.3f}s ({:.0f} rays/sec)", (size_t)imageWidth * imageHeight, s, (imageWidth * (double)imageHeight) / s);
        t0 = std::chrono::steady_clock::now();
        SaveImageBytes(outputDirectory + filename, frames[0].data(), imageWidth, imageHeight);
        lastPhases.save_ms = 1e3 * seconds_since(t0);
        return;
    }
    lastPixels = tracer.render(cam, imageWidth, imageHeight);
    lastHitIds = tracer.lastHitIds();
    const double s = seconds_since(t0);
    lastPhases.render_ms = 1e3 * s;
    LOG_INFO("Rendered {} pixels in {:.3f}s ({:.0f} rays/sec)", lastPixels.size(), s, (imageWidth * (double)imageHeight) / s);
    t0 = std::chrono::steady_clock::now();
    SaveImage(outputDirectory + filename, lastPixels, imageWidth, imageHeight);
    lastPhases.save_ms = 1e3 * seconds_since(t0);
}

void TraceImages::TraceImage(std::string gridFileLocation, std::string outputFileName, RayTracingMethod method) {
    LOG_INFO("========== Starting image generation ==========");
    LOG_INFO("Input: {} | Output: {} | Method: {}", gridFileLocation, outputFileName, getRayTracingMethodName(method));
    auto t0 = std::chrono::steady_clock::now();
    lastOk = false;
    lastPhases = Phases{};
    try {
        Loaded l = loadOrGenerate(gridFileLocation, gridResolution, gridResolution, gridResolution, method);
        LOG_INFO("Voxel grid generated with size: ({:.6f}, {:.6f}, {:.6f})", l.grid.voxelSize.x, l.grid.voxelSize.y, l.grid.voxelSize.z);
        renderAndSave(l, setupCamera(l.grid), outputFileName, method);
        lastPhases.total_ms = 1e3 * seconds_since(t0);
        LOG_INFO("Success! Generated {} in {:.3f}s", outputFileName, seconds_since(t0));
        LOG_INFO("Phases (ms): context {:.2f} | read {:.2f} | upload + build {:.2f} | render + copy {:.2f} | save {:.2f}", lastPhases.context_ms,
                 lastPhases.load_ms, lastPhases.build_ms, lastPhases.render_ms, lastPhases.save_ms);
        lastOk = true;
    } catch (const std::exception& e) {
        // as the reference: log and carry on with the next image (TraceImages.cpp:69-74)
        LOG_ERROR("Failed: {} - {} (after {:.3f}s)", gridFileLocation, e.what(), seconds_since(t0));
    }
}

void TraceImages::TraceImageMultiView(std::string gridFileLocation, std::string baseOutputName, RayTracingMethod method) {
    LOG_INFO("========== Starting multi-view image generation ==========");
    auto t0 = std::chrono::steady_clock::now();
    lastOk = false;
    lastPhases = Phases{};
    try {
        // the grid is built once and stays resident for all three views
        Loaded l = loadOrGenerate(gridFileLocation, gridResolution, gridResolution, gridResolution, method);
        const fs::path base(baseOutputName);
        const std::string ext = base.extension().string(), stem = base.stem().string();
        const std::pair<CameraView, const char*> views[] = { { CameraView::FRONT, "_front" }, { CameraView::SIDE, "_side" }, { CameraView::TOP, "_top" } };
        // one launch for the three views, quantised on the GPU: 3 bytes per pixel come back
        std::vector<Camera> cams;
        for (const auto& vs : views) cams.push_back(setupCameraWithView(l.grid, vs.first));
        HipVoxelRayTracer tracer(l.scene);
        tracer.setNearestHit(nearestHit);
        auto tv = std::chrono::steady_clock::now();
        const auto frames = tracer.renderViewsRgb8(cams, imageWidth, imageHeight);
        lastPixels.clear();
        lastHitIds = tracer.lastHitIds();
        lastPhases.render_ms = 1e3 * seconds_since(tv);
        LOG_INFO("{} views rendered in {:.3f}s", frames.size(), seconds_since(tv));
        tv = std::chrono::steady_clock::now();
        for (size_t i = 0; i < frames.size(); ++i)
            SaveImageBytes(outputDirectory + stem + views[i].second + ext, frames[i].data(), imageWidth, imageHeight);
        lastPhases.save_ms = 1e3 * seconds_since(tv);
        lastPhases.total_ms = 1e3 * seconds_since(t0);
        LOG_INFO("Success! Generated all views in {:.3f}s", seconds_since(t0));
        lastOk = true;
    } catch (const std::exception& e) {
        LOG_ERROR("Failed: {} - {} (after {:.3f}s)", gridFileLocation, e.what(), seconds_since(t0));
    }
}

// The reference encodes .jpg / .png / .ppm by extension (TraceImages.cpp:272-317, stb_image_write); this library
// has a PNG and a PPM writer and no JPEG encoder (image codecs are outside the hot path).  A name with any other
// extension -- the reference's own driver passes ".jpg" for every output (RayTracer.cpp:24-27) -- is therefore
// written as PPM under the same name with the extension replaced by ".ppm": the bytes never contradict the name, and
// the log names the file that was really written.
static std::string writable_name(const std::string& filename, std::string& ext) {
    ext = lower_ext(filename);
    if (ext == ".png" || ext == ".ppm") return filename;
    fs::path p(filename);
    p.replace_extension(".ppm");
    LOG_WARN("No encoder for '{}' in this library: writing {} (PPM) instead of {}", ext.empty() ? "(no extension)" : ext, p.string(), filename);
    ext = ".ppm";
    return p.string();
}

void TraceImages::SaveImageBytes(const std::string& filename, const unsigned char* rgb8, int width, int height) {
    std::string ext;
    const std::string out = writable_name(filename, ext);
    const fs::path p(out);
    if (!p.parent_path().empty() && !fs::exists(p.parent_path())) fs::create_directories(p.parent_path());
    const bool ok = ext == ".png" ? saveBytesToPNG(out, rgb8, width, height) : saveBytesToPPM(out, rgb8, width, height);
    if (ok) LOG_INFO("Image saved: {}", out); else LOG_ERROR("Failed to save: {}", out);
}

void TraceImages::SaveImage(const std::string& filename, const std::vector<Vec3>& pixels, int width, int height) {
    std::string ext;
    const std::string out = writable_name(filename, ext);
    const fs::path p(out);
    if (!p.parent_path().empty() && !fs::exists(p.parent_path())) fs::create_directories(p.parent_path());
    const bool ok = ext == ".png" ? saveToPNG(out, pixels, width, height) : saveToPPM(out, pixels, width, height);
    if (ok) LOG_INFO("Image saved: {}", out); else LOG_ERROR("Failed to save: {}", out);
}

// ---------------------------------------------------------------------------------------- .voxgrid

// Byte layout of TraceImages.cpp:360-457: int nx,ny,nz; Vec3 min,max,voxelSize,center; Voxel (28 raw bytes)
// for x { y { z } }; int triangle_count; Triangle[...] (48 bytes each).  triangle_indices are NOT in the file.
bool VoxelGrid::saveToFile(const std::string& filename) const {
    std::ofstream f(filename, std::ios::binary);
    if (!f) return false;
    auto put = [&](const void* p, size_t n) { f.write(static_cast<const char*>(p), (std::streamsize)n); };
    put(&nx, 4); put(&ny, 4); put(&nz, 4);
    put(&minBound, 12); put(&maxBound, 12); put(&voxelSize, 12); put(&center, 12);
    for (int x = 0; x < nx; ++x)
        for (int y = 0; y < ny; ++y)
            for (int z = 0; z < nz; ++z) put(&voxels[x][y][z], sizeof(Voxel));
    const int n = (int)triangles.size();
    put(&n, 4);
    put(triangles.data(), (size_t)n * sizeof(Triangle));
    return (bool)f;
}

VoxelGrid VoxelGrid::loadFromFile(const std::string& filename) {
    std::ifstream f(filename, std::ios::binary);
    if (!f) throw std::runtime_error("Failed to open file: " + filename);
    auto get = [&](void* p, size_t n) {
        f.read(static_cast<char*>(p), (std::streamsize)n);
        if ((size_t)f.gcount() != n) throw std::runtime_error("Truncated voxel grid file: " + filename);
    };
    int nx, ny, nz;
    Vec3 mn, mx, vs, center;
    get(&nx, 4); get(&ny, 4); get(&nz, 4);
    if (nx < 1 || ny < 1 || nz < 1 || (long long)nx * ny * nz > (1ll << 32)) throw std::runtime_error("Bad voxel grid dimensions in " + filename);
    get(&mn, 12); get(&mx, 12); get(&vs, 12); get(&center, 12);
    std::vector<std::vector<std::vector<Voxel>>> vox(nx, std::vector<std::vector<Voxel>>(ny, std::vector<Voxel>(nz)));
    for (int x = 0; x < nx; ++x)
        for (int y = 0; y < ny; ++y)
            for (int z = 0; z < nz; ++z) get(&vox[x][y][z], sizeof(Voxel));
    int n = 0;
    get(&n, 4);
    if (n < 0) throw std::runtime_error("Bad triangle count in " + filename);
    std::vector<Triangle> tris((size_t)n);
    get(tris.data(), (size_t)n * sizeof(Triangle));
    return VoxelGrid(std::move(vox), {}, nx, ny, nz, mn, mx, vs, std::move(tris));
}
