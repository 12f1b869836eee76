// oracle/ref_bridge.cpp -- C-ABI bridge onto the REAL reference functions.  TEST INFRASTRUCTURE ONLY.
//
// This file is compiled together with the reference's own, unmodified sources where they lie
// (/root/reference/RayTracer/GeometryUtils.cpp and Shaders.cpp, plus the header-only
// STLReader.h / VectorMath.h) into oracle/_ref/libref.so by oracle/Makefile.  It adds nothing
// but extern "C" entry points so that tests can call the reference's functions through ctypes.
// Only the parts of the reference that compile from its own files with the toolchain in this
// image are bridged; SceneCreator.cpp and OctreeLikeRayTracer.cpp include Logger.h -> spdlog,
// which the image lacks, so they are treated as unbuildable (no stand-in headers are written).
//
// Mandatory flags: -include stdlib.h so that the unqualified abs() calls in
// GeometryUtils.cpp:78,84 bind to the float overload as they do on the reference's MSVC platform.
#include <cstring>
#include <string>
#include <vector>

#include "GeometryUtils.h"
#include "Shaders.h"
#include "STLReader.h"
#include "VectorMath.h"

using cg_datastructures::RayHit;
using cg_datastructures::Triangle;
using cg_datastructures::Vec3;

static Triangle make_tri(const float* t) {
    Triangle tr;
    tr.v0 = { t[0], t[1], t[2] };
    tr.v1 = { t[3], t[4], t[5] };
    tr.v2 = { t[6], t[7], t[8] };
    tr.normal = { t[9], t[10], t[11] };
    return tr;
}

extern "C" {

// GeometryUtils::rayTriangleIntersection (GeometryUtils.cpp:57-97)
int ref_ray_triangle(const float* tri12, const float* dir, const float* org, float* t) {
    Triangle tr = make_tri(tri12);
    Vec3 d = { dir[0], dir[1], dir[2] }, o = { org[0], org[1], org[2] };
    float tt = 0.0f;
    bool hit = GeometryUtils::rayTriangleIntersection(tr, d, o, tt);
    *t = tt;
    return hit ? 1 : 0;
}

// batch form: n rays against n triangles
void ref_ray_triangle_batch(const float* tris, const float* dirs, const float* orgs, unsigned n,
                            int* hit, float* t) {
    for (unsigned i = 0; i < n; ++i) hit[i] = ref_ray_triangle(tris + 12ull * i, dirs + 3ull * i, orgs + 3ull * i, t + i);
}

// GeometryUtils::rayBoxIntersection (GeometryUtils.cpp:15-37); n rays against one box
void ref_ray_box_batch(const float* orgs, const float* dirs, const float* bmin, const float* bmax, unsigned n,
                       int* hit, float* t_near, float* t_far) {
    const Vec3 lo = { bmin[0], bmin[1], bmin[2] }, hi = { bmax[0], bmax[1], bmax[2] };
    for (unsigned i = 0; i < n; ++i) {
        const Vec3 o = { orgs[3 * i], orgs[3 * i + 1], orgs[3 * i + 2] }, d = { dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2] };
        const Ray r(o, d);
        float tn = 0.0f, tf = 0.0f;
        hit[i] = GeometryUtils::rayBoxIntersection(r, lo, hi, tn, tf) ? 1 : 0;
        t_near[i] = tn;
        t_far[i] = tf;
    }
}

// GeometryUtils::computeTriangleMinMax (GeometryUtils.cpp:43-51)
void ref_triangle_min_max(const float* tri12, float* mn, float* mx) {
    Triangle tr = make_tri(tri12);
    Vec3 a, b;
    GeometryUtils::computeTriangleMinMax(tr, a, b);
    mn[0] = a.x; mn[1] = a.y; mn[2] = a.z;
    mx[0] = b.x; mx[1] = b.y; mx[2] = b.z;
}

// Shaders::LambertianShader (Shaders.cpp:4-22) called as OctreeLikeRayTracer::computeShading does
// (OctreeLikeRayTracer.cpp:30-34): r=g=b=1, hit.point = triangle.v0.
void ref_lambert(const float* tri12, const float* ray_origin, float* rgb) {
    Triangle tr = make_tri(tri12);
    RayHit h;
    h.hit = true;
    h.t = 1.0f;
    h.point = tr.v0;
    h.origin = { ray_origin[0], ray_origin[1], ray_origin[2] };
    h.normal = tr.normal;
    Shaders::Pixel px = Shaders::LambertianShader(h, 1.0l, 1.0l, 1.0l, tr);
    rgb[0] = px.color.x; rgb[1] = px.color.y; rgb[2] = px.color.z;
}

// vector_math (VectorMath.h:9-43)
void ref_normalize(const float* v, float* out) {
    Vec3 r = vector_math::normalize(Vec3{ v[0], v[1], v[2] });
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void ref_cross(const float* a, const float* b, float* out) {
    Vec3 r = vector_math::cross(Vec3{ a[0], a[1], a[2] }, Vec3{ b[0], b[1], b[2] });
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
float ref_dot(const float* a, const float* b) {
    return vector_math::dot(Vec3{ a[0], a[1], a[2] }, Vec3{ b[0], b[1], b[2] });
}

// stl_reader::StlMesh<float, unsigned> (STLReader.h:234-330): load, then expose the deduped
// vertices, the kept triangles' corner indices and the per-triangle file normals.
struct RefMesh {
    stl_reader::StlMesh<float, unsigned int> mesh;
    std::string error;
};

void* ref_stl_open(const char* path) {
    RefMesh* m = new RefMesh();
    try {
        m->mesh.read_file(path);
    } catch (std::exception& e) {
        m->error = e.what();
    }
    return m;
}
const char* ref_stl_error(void* h) { return static_cast<RefMesh*>(h)->error.c_str(); }
unsigned long long ref_stl_num_vrts(void* h) { return static_cast<RefMesh*>(h)->mesh.num_vrts(); }
unsigned long long ref_stl_num_tris(void* h) { return static_cast<RefMesh*>(h)->mesh.num_tris(); }
void ref_stl_copy(void* h, float* coords, unsigned* tris, float* normals) {
    auto& mesh = static_cast<RefMesh*>(h)->mesh;
    for (size_t i = 0; i < mesh.num_vrts(); ++i) std::memcpy(coords + 3 * i, mesh.vrt_coords(i), 12);
    for (size_t t = 0; t < mesh.num_tris(); ++t) {
        std::memcpy(tris + 3 * t, mesh.tri_corner_inds(t), 12);
        std::memcpy(normals + 3 * t, mesh.tri_normal(t), 12);
    }
}
void ref_stl_close(void* h) { delete static_cast<RefMesh*>(h); }

}  // extern "C"
