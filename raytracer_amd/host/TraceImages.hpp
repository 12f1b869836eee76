// TraceImages.hpp -- the reference's pipeline entry (TraceImages.h:16-89) with the HIP method added.
//
// TraceImage(stl, png, method) keeps its signature and behaviour: load (.stl or .voxgrid) -> 50^3 voxel
// grid -> tracer from the factory -> default front camera, 800x600 -> image file; exceptions are caught,
// logged and swallowed (TraceImages.cpp:69-74).  New: RayTracingMethod::HIP_VOXEL builds the grid and
// traces on the MI355X.  The three CPU enumerators keep their numeric values; this standalone library
// carries no CPU tracer (that is the reference's own OctreeLikeRayTracer), so selecting one of them
// here throws std::runtime_error -- which TraceImage logs like any other failure.
#pragma once

#include <memory>
#include <string>
#include <vector>

#include "HipVoxelRayTracer.hpp"
#include "IRayTracer.hpp"
#include "StlMesh.hpp"
#include "rt_types.hpp"

enum class RayTracingMethod {
    VOXEL_DDA,   // reference CPU tracer (TraceImages.h:17)
    ART,         // reference CPU tracer (TraceImages.h:18)
    OCTREE,      // reference CPU tracer (TraceImages.h:19)
    HIP_VOXEL    // NEW: libhvrt on the GPU -- grid build + traversal
};

enum class CameraView { FRONT, SIDE, TOP };

class TraceImages {
public:
    void TraceImage(std::string gridFileLocation, std::string outputFileName,
                    RayTracingMethod method = RayTracingMethod::VOXEL_DDA);
    void TraceImageMultiView(std::string gridFileLocation, std::string baseOutputName,
                             RayTracingMethod method = RayTracingMethod::VOXEL_DDA);

    // Knobs the reference hard-codes (TraceImages.cpp:58, 223, 227); defaults are its literals, except
    // the output directory (the reference's is a Windows path): default = current directory.
    int gridResolution = 50;
    int imageWidth = 800, imageHeight = 600;
    std::string outputDirectory = "";
    // false: the reference's traversal.  true: HipVoxelRayTracer::setNearestHit (an extension, off by default)
    bool nearestHit = false;
    // last frame rendered (for callers / tests that want the pixels, not the file)
    std::vector<cg_datastructures::Vec3> lastPixels;
    std::vector<unsigned int> lastHitIds;   // of the last image; single-view renders fetch them only if keepHitIds
    bool keepHitIds = false;
    // false (default): TraceImage quantises the frame on the GPU (ImageSaver.h's conversion) and fetches 3 bytes per pixel
    // -- lastPixels stays empty; true: the float frame comes back as IRayTracer::render returns it (12 bytes per pixel) and
    // is quantised on the host, as in the reference's flow (TraceImages.cpp:215-229).  Same bytes on disk either way.
    bool keepPixels = false;
    bool lastOk = false;
    // where the last TraceImage / TraceImageMultiView call spent its time, host clock, milliseconds (the reference logs
    // grid generation and rendering, TraceImages.cpp:38-44, 54-67): taking a context (created, or from the process-wide
    // pool), reading the file, upload + voxel build, render() incl. the copy to the host, writing the image
    struct Phases { double context_ms = 0, load_ms = 0, build_ms = 0, render_ms = 0, save_ms = 0, total_ms = 0; };
    Phases lastPhases;

    // the factory seam (TraceImages.cpp:464-484), public here so that tests can reach it
    std::unique_ptr<IRayTracer> createRayTracer(VoxelGrid& voxelGrid, RayTracingMethod method);
    cg_datastructures::Camera setupCamera(const VoxelGrid& voxelGrid);
    cg_datastructures::Camera setupCameraWithView(const VoxelGrid& voxelGrid, CameraView view);
    static std::string getRayTracingMethodName(RayTracingMethod method);

    // GPU replacement of BuildVoxelGridFromStlMesh (SceneCreator.cpp:308-329): bounds, parameters,
    // triangle list, CSR -- the grid stays resident in `scene`; meta-only VoxelGrid (no voxel array) out.
    static VoxelGrid buildSceneOnGpu(const stl_reader::StlMesh<float, unsigned int>& mesh, int nx, int ny, int nz,
                                     HipScene& scene);

private:
    struct Loaded {
        VoxelGrid grid;                       // meta (+ voxels / indices when read from a .voxgrid file)
        std::shared_ptr<HipScene> scene;      // resident GPU scene
    };
    Loaded loadOrGenerate(const std::string& path, int nx, int ny, int nz, RayTracingMethod method);
    void renderAndSave(Loaded& l, const cg_datastructures::Camera& cam, const std::string& filename, RayTracingMethod method);
    void SaveImage(const std::string& filename, const std::vector<cg_datastructures::Vec3>& pixels, int width, int height);
    void SaveImageBytes(const std::string& filename, const unsigned char* rgb8, int width, int height);
};
