// rt_types.hpp -- host-side data types of the drop-in, source-compatible with the reference's
// Voxels.h:6-23, VoxelGrid.h:8-42 and RayTracerCommon.h:9-43 (same namespace, names and field order, so
// code written against the reference's headers compiles against these).  Own code; nothing is copied.
#pragma once

#include <limits>
#include <string>
#include <utility>
#include <vector>

namespace cg_datastructures {

struct Vec3 { float x, y, z; };                      // 12 bytes

struct Triangle { Vec3 v0, v1, v2, normal; };        // 48 bytes: what hvrt_set_mesh takes, as is

struct Voxel {                                       // 28 bytes (bool + 3 pad, u32, u32, f32, Vec3)
    bool occupied = false;
    unsigned int triangle_start_idx = 0;
    unsigned int triangle_count = 0;
    float density = 0.0f;
    Vec3 color = { 0.0f, 0.0f, 0.0f };
};

struct Ray {
    Vec3 origin, direction;
    Ray(const Vec3& o, const Vec3& d) : origin(o), direction(d) {}
};

struct RayHit {                                      // 52 bytes
    bool hit = false;
    float t = std::numeric_limits<float>::max();
    Vec3 point{ 0, 0, 0 };
    Vec3 origin{ 0, 0, 0 };
    Vec3 normal{ 0, 0, 0 };
    unsigned int triangleIndex = 0;
    unsigned int voxelIndex = 0;
};

struct Camera {
    Vec3 position, lookAt, up;
    float fov;          // degrees
    float aspectRatio;  // width / height
    Camera(const Vec3& p, const Vec3& target, const Vec3& upVec, float fovDeg, float aspect)
        : position(p), lookAt(target), up(upVec), fov(fovDeg), aspectRatio(aspect) {}
};

}  // namespace cg_datastructures

// The reference's grid container: voxels[x][y][z] + flat index array + triangles.
struct VoxelGrid {
    std::vector<std::vector<std::vector<cg_datastructures::Voxel>>> voxels;
    std::vector<unsigned int> triangle_indices;
    int nx = 0, ny = 0, nz = 0;
    cg_datastructures::Vec3 minBound{}, maxBound{}, voxelSize{}, center{};
    std::vector<cg_datastructures::Triangle> triangles;

    VoxelGrid() = default;
    VoxelGrid(std::vector<std::vector<std::vector<cg_datastructures::Voxel>>>&& v, std::vector<unsigned int>&& idx,
              int nx_, int ny_, int nz_, const cg_datastructures::Vec3& mn, const cg_datastructures::Vec3& mx,
              const cg_datastructures::Vec3& vs, std::vector<cg_datastructures::Triangle>&& tris)
        : voxels(std::move(v)), triangle_indices(std::move(idx)), nx(nx_), ny(ny_), nz(nz_),
          minBound(mn), maxBound(mx), voxelSize(vs),
          center{ (mn.x + mx.x) * 0.5f, (mn.y + mx.y) * 0.5f, (mn.z + mx.z) * 0.5f },   // VoxelGrid.h:30-32
          triangles(std::move(tris)) {}

    // .voxgrid cache (TraceImages.cpp:360-457): same byte layout as the reference writes / reads.
    bool saveToFile(const std::string& filename) const;
    static VoxelGrid loadFromFile(const std::string& filename);
};
