// HipVoxelRayTracer.hpp -- IRayTracer on the MI355X through libhvrt's C ABI (include/hvrt.h).
//
// Drop-in for OctreeLikeRayTracer (OctreeRayTracer.h:28-79): same constructor shape, same three
// virtuals, same results.  Errors from the C ABI become std::runtime_error, which TraceImage catches
// and logs exactly like any other exception (TraceImages.cpp:69-74).  This header needs no HIP header.
#pragma once

#include <memory>
#include <string>
#include <vector>

#ifdef HVRT_USE_REFERENCE_HEADERS   // built inside the reference tree: its own IRayTracer.h:17-44 and VoxelGrid.h:8-42
#include "IRayTracer.h"            // (-> RayTracerCommon.h -> Voxels.h)
#include "VoxelGrid.h"
#else                              // standalone: this repository's restatement of the same types
#include "IRayTracer.hpp"
#endif

struct hvrt_ctx;
struct hvrt_group;

// A scene resident on the GPU(s): triangles + CSR grid.  Shared by tracers so that TraceImageMultiView's
// three views (TraceImages.cpp:77-119) reuse one upload / one build.
// Devices: a scene lives on device_id alone unless HVRT_DEVICES in the environment lists several ("0,1,2,3", or "all")
// -- a device group is opt-in: one process per GPU, rank r constructing HipScene(r), is the other multi-GPU model and
// gets exactly its device.  With a group the scene is built on the first device, replicated when the first frame of at
// least HVRT_GROUP_MIN_PIXELS pixels (default 2^22) is asked for (hvrt_group_replicate_scene), and such frames are traced
// by all members, each copying its own row tiles into the caller's frame (hvrt_group_render); smaller frames, batched
// views and single rays stay on the first device.  Contexts and groups are pooled per process (see the .cpp).
class HipScene {
public:
    explicit HipScene(int device_id = 0);
    ~HipScene();
    HipScene(const HipScene&) = delete;
    HipScene& operator=(const HipScene&) = delete;

    // Upload triangles and build the grid ON THE GPU (replaces voxelizeTrianglesDirect,
    // SceneCreator.cpp:197-281).  minBound / maxBound / voxelSize as computeVoxelizationParams gives.
    void setTriangles(const std::vector<cg_datastructures::Triangle>& tris);
    void setTriangles(std::vector<cg_datastructures::Triangle>&& tris);      // (takes the vector: no second host copy of the mesh)
    void buildGrid(int nx, int ny, int nz, const cg_datastructures::Vec3& mn, const cg_datastructures::Vec3& mx,
                   const cg_datastructures::Vec3& voxelSize);
    // Ingest a CPU-built VoxelGrid (flattens voxels[x][y][z] to the x-fastest CSR).  A grid whose
    // triangle_indices are missing (a .voxgrid file, TraceImages.cpp:447-448) is rebuilt on the GPU.
    void setGrid(const VoxelGrid& grid);
    // GPU bounds of a vertex list (computeMeshBounds, SceneCreator.cpp:49-84).
    void computeBounds(const float* xyz, size_t n_points, cg_datastructures::Vec3& mn, cg_datastructures::Vec3& mx);
    // Copy the GPU-built CSR back into the reference's container (voxels[x][y][z] + triangle_indices).
    VoxelGrid downloadGrid() const;

    // Destroys the idle contexts / groups the process-wide pool holds (they are otherwise kept until exit).
    static void releaseCachedContexts();

    hvrt_ctx* handle() const { return ctx_; }
    // the device group when the scene spans several GPUs (nullptr otherwise), and whether a frame of this size uses it
    hvrt_group* group() const { return group_; }
    bool spansDevices(int width, int height) const;
    int deviceCount() const;
    double lastBuildMs() const;
    double lastRenderMs() const;

    void replicate();                  // group: hands the scene to the other members if they do not hold it yet

private:
    hvrt_ctx* ctx_ = nullptr;          // member 0 of the group when there is one
    hvrt_group* group_ = nullptr;
    int device_ = 0;
    std::string pool_key_;             // the group's device list
    bool replicated_ = false;          // the other members hold the current mesh and grid
    size_t group_min_pixels_ = (size_t)1 << 22;
    std::vector<cg_datastructures::Triangle> tris_;   // host copy (downloadGrid returns them)
};

class HipVoxelRayTracer : public IRayTracer {
public:
    // Same shape as OctreeLikeRayTracer(const VoxelGrid&, int, int): uploads the CPU-built grid.
    explicit HipVoxelRayTracer(const VoxelGrid& grid, int maxDepth = 8, int maxTrianglesPerNode = 10);
    // Scene already resident (built on the GPU): no upload.
    explicit HipVoxelRayTracer(std::shared_ptr<HipScene> scene);
    ~HipVoxelRayTracer() override = default;

    std::vector<cg_datastructures::Vec3> render(const cg_datastructures::Camera& camera, int width, int height) override;
    cg_datastructures::RayHit traceRay(const cg_datastructures::Ray& ray) override;
    std::string getMethodName() const override { return "HipVoxelGrid"; }

    // Several views of the resident scene in ONE launch, quantised to 8-bit RGB on the GPU
    // (ImageSaver.h's conversion): what TraceImageMultiView needs, at 3 bytes per pixel over PCIe.
    // Returns cameras.size() frames of width*height*3 bytes, rows bottom-up like render().
    // wantHitIds: also fetch the hit-id image of the last view (lastHitIds(); a third more bytes over PCIe)
    std::vector<std::vector<unsigned char>> renderViewsRgb8(const std::vector<cg_datastructures::Camera>& cameras,
                                                            int width, int height, bool wantHitIds = true);
    const HipScene& scene() const { return *scene_; }

    // false (default): the reference's traversal.  true: nearest-hit mode (hvrt_set_mode(1)), an extension for
    // callers who want a right image rather than the reference's; see include/hvrt.h.
    void setNearestHit(bool on);

    // Batched traceRay and the statistics the reference logs (OctreeLikeRayTracer.cpp:57-65).
    std::vector<cg_datastructures::RayHit> traceRays(const std::vector<cg_datastructures::Ray>& rays);
    unsigned long long voxelsTraversed() const { return voxels_; }
    unsigned long long triangleTests() const { return tests_; }
    // per-pixel hit triangle ids of the last render() (HVRT_MISS = 0xFFFFFFFF for none): fetched only after
    // keepHitIds(true); renderViewsRgb8 always keeps those of its last view
    void keepHitIds(bool on) { keep_hit_ids_ = on; }
    const std::vector<unsigned int>& lastHitIds() const { return hit_ids_; }

private:
    std::shared_ptr<HipScene> scene_;
    unsigned long long voxels_ = 0, tests_ = 0;
    std::vector<unsigned int> hit_ids_;
    bool keep_hit_ids_ = false;
};
