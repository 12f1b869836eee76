/*
 * oracle/cpu_ref.h -- CPU restatement of the reference's voxel-grid build and
 * primary-ray traversal path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  Nothing under raytracer_amd/ links,
 * loads or calls it; the product path fails loudly if its HIP library is missing.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference/RayTracer/).  All arithmetic is IEEE fp32, round-to-nearest-even,
 * no FMA contraction (build with -ffp-contract=off), comparisons with NaN are false.
 *
 * Pinning status ("Oracle"):
 *   - ray/triangle test, triangle min/max, vector math, Lambert shader, STL ingest:
 *     pinned bit-for-bit against the real reference sources compiled in place
 *     (oracle/_ref, built from GeometryUtils.cpp, Shaders.cpp, STLReader.h, VectorMath.h).
 *   - voxelizer CSR order, DDA init/advance, ray generation, render loop:
 *     SceneCreator.cpp and OctreeLikeRayTracer.cpp need spdlog, which this image lacks,
 *     so they are unbuildable here without a stand-in: PARITY UNPINNED by live execution;
 *     pinned only by the reference run outputs recorded in SURVEY.md Appendix C / section 6
 *     (tests/test_oracle_pinning.py: both recorded runs; tests/test_gpu_recorded_runs.py holds the
 *     HIP path to the same recorded numbers directly).
 */
#ifndef ORACLE_CPU_REF_H
#define ORACLE_CPU_REF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z; } orc_vec3;

/* Host-computed camera frame: the loop-invariant part of generateRay
 * (OctreeLikeRayTracer.cpp:342-359), hoisted. */
typedef struct {
    orc_vec3 origin;      /* camera.position */
    orc_vec3 lower_left;  /* :358-359 */
    orc_vec3 horizontal;  /* :354 */
    orc_vec3 vertical;    /* :355 */
} orc_camera_frame;

/* Flat scene: x-fastest CSR grid + 48-byte AoS triangles (v0,v1,v2,normal: Voxels.h:10-13). */
typedef struct {
    int nx, ny, nz;
    orc_vec3 min;          /* VoxelGrid::minBound */
    orc_vec3 vs;           /* VoxelGrid::voxelSize (the builder makes it uniform) */
    const float* tris;     /* T x 12 floats */
    uint32_t n_tris;
    const uint32_t* offsets; /* nx*ny*nz + 1, exclusive scan, flat index ix + nx*(iy + ny*iz) */
    const uint32_t* indices; /* offsets[nvox] entries, ascending per voxel */
    orc_vec3 max;          /* VoxelGrid::maxBound (used by mode 1 only) */
    int mode;              /* 0: the reference's traversal.  1: nearest-hit mode (an extension,
                              nothing in the reference pins it): enter the grid through rayBoxIntersection
                              (GeometryUtils.cpp:15-37), keep the nearest hit over all visited voxels, stop once
                              it lies before the current voxel's exit. */
} orc_scene;

/* Mirrors cg_datastructures::RayHit (RayTracerCommon.h:17-29), 52 bytes. */
typedef struct {
    uint32_t hit;
    float t;
    float point[3];
    float origin[3];
    float normal[3];
    uint32_t triangle_index;
    uint32_t voxel_index;
} orc_hit;

/* GeometryUtils.cpp:15-37 (rayBoxIntersection, unused by the reference's own traversal). */
int orc_ray_box(const float org[3], const float dir[3], const float bmin[3], const float bmax[3],
                float* t_near, float* t_far);

/* SceneCreator.cpp:49-84 (computeMeshBounds) over n points (xyz triples), in order. */
void orc_mesh_bounds(const float* pts, uint64_t n, float mn[3], float mx[3]);

/* SceneCreator.cpp:91-122 (computeVoxelizationParams): uniform voxel size. */
float orc_voxel_size(const float mn[3], const float mx[3], int nx, int ny, int nz);

/* SceneCreator.cpp:197-281 (voxelizeTrianglesDirect) as a flat two-pass CSR build.
 * Pass 1 (counts_only != 0, or indices == NULL): fills offsets[nvox+1] only.
 * Returns the number of references. indices must hold offsets[nvox] entries. */
uint64_t orc_voxelize(const float* tris, uint32_t n_tris, int nx, int ny, int nz,
                      const float mn[3], const float vs[3],
                      uint32_t* offsets, uint32_t* indices);
/* The separating-axis option of the voxelizer (hvrt_set_overlap_test(ctx, 1)).  NOT the reference's rule and pinned by
 * nothing in it (GeometryUtils.h:19 declares triangleIntersectsBox, no file defines it): parity unpinned. */
uint64_t orc_voxelize_sat(const float* tris, uint32_t n_tris, int nx, int ny, int nz,
                          const float mn[3], const float vs[3], uint32_t* offsets, uint32_t* indices);
int orc_triangle_box_overlap(const float* tri, const float mn[3], const float vs[3], int ix, int iy, int iz);

/* OctreeLikeRayTracer.cpp:342-359, the loop-invariant head of generateRay. */
void orc_camera_frame_make(const float pos[3], const float look_at[3], const float up[3],
                           float fov_deg, float aspect, orc_camera_frame* out);

/* OctreeLikeRayTracer.cpp:361-364: direction through (u,v). */
void orc_generate_ray(const orc_camera_frame* f, float u, float v, float dir[3]);

/* GeometryUtils.cpp:57-97. tri = 12 floats (normal ignored). Returns 1 on hit. */
int orc_ray_triangle(const float* tri, const float dir[3], const float org[3], float* t);

/* GeometryUtils.cpp:43-51. */
void orc_triangle_min_max(const float* tri, float mn[3], float mx[3]);

/* OctreeLikeRayTracer.cpp:309-336 (traverseVoxelGrid) with :155-195, :201-260, :267-295.
 * counters[0] += voxels tested, counters[1] += ray/triangle tests (may be NULL). */
void orc_trace(const orc_scene* s, const float org[3], const float dir[3],
               orc_hit* out, uint64_t counters[2]);

/* OctreeLikeRayTracer.cpp:27-40 + Shaders.cpp:4-22. v is the pixel's v coordinate. */
void orc_shade(const orc_scene* s, const orc_hit* h, float v, float rgb[3]);

/* Counter-based jitter used by the spp>1 extension (no reference counterpart). */
void orc_jitter(uint32_t x, uint32_t y, uint32_t s, uint32_t seed, float* jx, float* jy);

/* OctreeLikeRayTracer.cpp:88-118 (render) + :71-82 (renderPixel), rows [y0,y1), one trace
 * per sample (the reference's second, statistics-only trace per pixel is not repeated, so
 * counters are half of what the reference logs).  Outputs are full-frame arrays indexed
 * y*W+x; any of rgb / hit_id / t / counters may be NULL.  spp==1 is the reference path. */
void orc_render_rows(const orc_scene* s, const orc_camera_frame* f, int W, int H,
                     int y0, int y1, int spp, uint32_t seed,
                     float* rgb, uint32_t* hit_id, float* t, uint64_t counters[2]);

#ifdef __cplusplus
}
#endif
#endif
