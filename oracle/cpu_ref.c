/*
 * oracle/cpu_ref.c -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY
 * (see cpu_ref.h for who may load it and for the pinning status of each part).
 *
 * Build: gcc -std=c11 -O2 -ffp-contract=off -fno-fast-math -fPIC -shared (oracle/Makefile).
 * Citations are file:line under /root/reference/RayTracer/.
 */
#include "cpu_ref.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- vector math */
/* VectorMath.h:9-43.  Evaluation order is the C order of the reference expressions:
 * a*b + c*d + e*f == ((a*b) + (c*d)) + (e*f). */

static inline orc_vec3 v_sub(orc_vec3 a, orc_vec3 b) { orc_vec3 r = { a.x - b.x, a.y - b.y, a.z - b.z }; return r; }
static inline orc_vec3 v_add(orc_vec3 a, orc_vec3 b) { orc_vec3 r = { a.x + b.x, a.y + b.y, a.z + b.z }; return r; }
static inline orc_vec3 v_mul(orc_vec3 a, float s)   { orc_vec3 r = { a.x * s, a.y * s, a.z * s }; return r; }
static inline float v_dot(orc_vec3 a, orc_vec3 b)   { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline orc_vec3 v_cross(orc_vec3 a, orc_vec3 b) {
    orc_vec3 r = { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x };
    return r;
}
/* VectorMath.h:9-15: divide by sqrt; a zero (or NaN) length leaves the vector unchanged. */
static inline orc_vec3 v_normalize(orc_vec3 v) {
    float len = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    if (len > 0.0f) { orc_vec3 r = { v.x / len, v.y / len, v.z / len }; return r; }
    return v;
}

/* static_cast<int>(float) as the reference's platform does it (x86-64 cvttss2si):
 * NaN and out-of-range values give INT_MIN ("integer indefinite").  C leaves this
 * undefined, so it is spelled out here and in the HIP kernels. */
static inline int f2i(float f) {
    if (!(f < 2147483648.0f) || f < -2147483648.0f) return INT_MIN;
    return (int)f;
}
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ---------------------------------------------------------------- bounds / params */

/* SceneCreator.cpp:49-84: start at +FLT_MAX / lowest, std::min(cur, x) = (x < cur) ? x : cur. */
void orc_mesh_bounds(const float* p, uint64_t n, float mn[3], float mx[3]) {
    for (int k = 0; k < 3; ++k) { mn[k] = FLT_MAX; mx[k] = -FLT_MAX; }
    for (uint64_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            float c = p[3 * i + k];
            if (c < mn[k]) mn[k] = c;
            if (mx[k] < c) mx[k] = c;
        }
}

/* SceneCreator.cpp:100-109. */
float orc_voxel_size(const float mn[3], const float mx[3], int nx, int ny, int nz) {
    float gx = mx[0] - mn[0], gy = mx[1] - mn[1], gz = mx[2] - mn[2];
    float m = gx < gy ? gy : gx;            /* std::max(gx, gy) */
    m = m < gz ? gz : m;
    int r = nx < ny ? ny : nx;
    r = r < nz ? nz : r;
    return m / (float)r;
}

/* ---------------------------------------------------------------- voxelizer */

/* GeometryUtils.cpp:43-51: std::min(std::min(a,b),c); std::min(a,b) = (b < a) ? b : a. */
static inline float min2(float a, float b) { return b < a ? b : a; }
static inline float max2(float a, float b) { return a < b ? b : a; }

void orc_triangle_min_max(const float* t, float mn[3], float mx[3]) {
    for (int k = 0; k < 3; ++k) {
        mn[k] = min2(min2(t[k], t[3 + k]), t[6 + k]);
        mx[k] = max2(max2(t[k], t[3 + k]), t[6 + k]);
    }
}

/* SceneCreator.cpp:23-37 + :177-188: inclusive voxel index range of a triangle's AABB. */
static inline void tri_range(const float* t, const int n[3], const float mn[3], const float vs[3],
                             int lo[3], int hi[3]) {
    float a[3], b[3];
    orc_triangle_min_max(t, a, b);
    for (int k = 0; k < 3; ++k) {
        lo[k] = clampi(f2i((a[k] - mn[k]) / vs[k]), 0, n[k] - 1);
        hi[k] = clampi(f2i((b[k] - mn[k]) / vs[k]), 0, n[k] - 1);
    }
}

/* Triangle / box overlap by separating axes (Akenine-Moller 2001), for the SAT option of the voxelizer
 * (hvrt_set_overlap_test(ctx, 1)).  NOT in the reference: GeometryUtils.h:19 declares triangleIntersectsBox and no
 * file defines it, so nothing pins this -- parity unpinned; it restates the published test for the voxel grown by
 * 2^-10 of its half size, in plain fp32 operations in the order libhvrt uses, and the tests compare the two CSRs. */
static inline float abs1(float x) { return x < 0.0f ? -x : x; }
static inline int separated(float p0, float p1, float p2, float r) {
    const float lo = min2(min2(p0, p1), p2), hi = max2(max2(p0, p1), p2);
    return lo > r || hi < -r;
}
static inline int edge_separates(const float e[3], const float h[3], const float a[3], const float b[3], const float d[3]) {
    const float fx = abs1(e[0]), fy = abs1(e[1]), fz = abs1(e[2]);
    if (separated(e[2] * a[1] - e[1] * a[2], e[2] * b[1] - e[1] * b[2], e[2] * d[1] - e[1] * d[2], h[1] * fz + h[2] * fy)) return 1;
    if (separated(e[0] * a[2] - e[2] * a[0], e[0] * b[2] - e[2] * b[0], e[0] * d[2] - e[2] * d[0], h[0] * fz + h[2] * fx)) return 1;
    if (separated(e[1] * a[0] - e[0] * a[1], e[1] * b[0] - e[0] * b[1], e[1] * d[0] - e[0] * d[1], h[0] * fy + h[1] * fx)) return 1;
    return 0;
}
int orc_triangle_box_overlap(const float* t, const float mn[3], const float vs[3], int ix, int iy, int iz) {
    const float GROW = 1.0009765625f;
    const int idx[3] = { ix, iy, iz };
    float h[3], a[3], b[3], d[3], e0[3], e1[3], e2[3];
    for (int k = 0; k < 3; ++k) {
        const float c = mn[k] + ((float)idx[k] + 0.5f) * vs[k];
        h[k] = (0.5f * vs[k]) * GROW;
        a[k] = t[k] - c; b[k] = t[3 + k] - c; d[k] = t[6 + k] - c;
    }
    for (int k = 0; k < 3; ++k) if (separated(a[k], b[k], d[k], h[k])) return 0;
    for (int k = 0; k < 3; ++k) { e0[k] = b[k] - a[k]; e1[k] = d[k] - b[k]; e2[k] = a[k] - d[k]; }
    if (edge_separates(e0, h, a, b, d) || edge_separates(e1, h, a, b, d) || edge_separates(e2, h, a, b, d)) return 0;
    const float nx = e0[1] * e1[2] - e0[2] * e1[1], ny = e0[2] * e1[0] - e0[0] * e1[2], nz = e0[0] * e1[1] - e0[1] * e1[0];
    const float dist = (nx * a[0] + ny * a[1]) + nz * a[2];
    const float r = (h[0] * abs1(nx) + h[1] * abs1(ny)) + h[2] * abs1(nz);
    if (dist > r || dist < -r) return 0;
    return 1;
}

/* SceneCreator.cpp:197-281.  The reference appends t (ascending) to a vector per voxel and then
 * flattens the vectors in flat order ix + nx*(iy + ny*iz); a count / exclusive-scan / fill in
 * ascending t produces the same two arrays. */
static uint64_t voxelize(const float* tris, uint32_t T, int nx, int ny, int nz, const float mn[3], const float vs[3],
                         uint32_t* offsets, uint32_t* indices, int sat);
uint64_t orc_voxelize(const float* tris, uint32_t T, int nx, int ny, int nz,
                      const float mn[3], const float vs[3],
                      uint32_t* offsets, uint32_t* indices) {
    return voxelize(tris, T, nx, ny, nz, mn, vs, offsets, indices, 0);
}
/* The SAT option (not the reference's rule): a voxel of the range keeps the triangle only if the two overlap. */
uint64_t orc_voxelize_sat(const float* tris, uint32_t T, int nx, int ny, int nz,
                          const float mn[3], const float vs[3],
                          uint32_t* offsets, uint32_t* indices) {
    return voxelize(tris, T, nx, ny, nz, mn, vs, offsets, indices, 1);
}
static uint64_t voxelize(const float* tris, uint32_t T, int nx, int ny, int nz, const float mn[3], const float vs[3],
                         uint32_t* offsets, uint32_t* indices, int sat) {
    const int n[3] = { nx, ny, nz };
    const uint64_t nvox = (uint64_t)nx * ny * nz;
    memset(offsets, 0, (nvox + 1) * sizeof(uint32_t));
    for (uint32_t t = 0; t < T; ++t) {
        int lo[3], hi[3];
        tri_range(tris + 12ull * t, n, mn, vs, lo, hi);
        for (int iz = lo[2]; iz <= hi[2]; ++iz)
            for (int iy = lo[1]; iy <= hi[1]; ++iy)
                for (int ix = lo[0]; ix <= hi[0]; ++ix)
                    if (!sat || orc_triangle_box_overlap(tris + 12ull * t, mn, vs, ix, iy, iz))
                        offsets[1 + (uint64_t)ix + (uint64_t)nx * ((uint64_t)iy + (uint64_t)ny * iz)]++;
    }
    /* offsets[v+1] holds count(v); turn it into start(v) stored at offsets[v+1] temporarily. */
    uint64_t run = 0;
    for (uint64_t v = 0; v < nvox; ++v) { uint32_t c = offsets[v + 1]; offsets[v + 1] = (uint32_t)run; run += c; }
    if (indices) {
        for (uint32_t t = 0; t < T; ++t) {
            int lo[3], hi[3];
            tri_range(tris + 12ull * t, n, mn, vs, lo, hi);
            for (int iz = lo[2]; iz <= hi[2]; ++iz)
                for (int iy = lo[1]; iy <= hi[1]; ++iy)
                    for (int ix = lo[0]; ix <= hi[0]; ++ix) {
                        if (sat && !orc_triangle_box_overlap(tris + 12ull * t, mn, vs, ix, iy, iz)) continue;
                        uint64_t f = (uint64_t)ix + (uint64_t)nx * ((uint64_t)iy + (uint64_t)ny * iz);
                        indices[offsets[f + 1]++] = t;
                    }
        }
        /* offsets[v+1] is now end(v) == start(v+1): exactly the CSR offsets array. */
    } else {
        /* counts only: shift starts into place (offsets[v] = start(v), offsets[nvox] = total). */
        for (uint64_t v = 0; v + 1 < nvox + 1; ++v) offsets[v] = offsets[v + 1];
        offsets[nvox] = (uint32_t)run;
    }
    return run;
}

/* ---------------------------------------------------------------- camera / rays */

/* OctreeLikeRayTracer.cpp:342-359. */
void orc_camera_frame_make(const float pos[3], const float look_at[3], const float up_hint[3],
                           float fov, float aspect, orc_camera_frame* out) {
    orc_vec3 p = { pos[0], pos[1], pos[2] };
    orc_vec3 la = { look_at[0], look_at[1], look_at[2] };
    orc_vec3 uh = { up_hint[0], up_hint[1], up_hint[2] };
    float fov_rad = fov * 3.14159265f / 180.0f;
    float half_h = tanf(fov_rad / 2.0f);
    float half_w = aspect * half_h;
    orc_vec3 fwd = v_normalize(v_sub(la, p));
    orc_vec3 right = v_normalize(v_cross(fwd, uh));
    orc_vec3 up = v_cross(right, fwd);
    orc_vec3 hor = v_mul(right, 2.0f * half_w);
    orc_vec3 ver = v_mul(up, 2.0f * half_h);
    orc_vec3 ll = v_add(v_sub(v_sub(p, v_mul(hor, 0.5f)), v_mul(ver, 0.5f)), fwd);
    out->origin = p; out->lower_left = ll; out->horizontal = hor; out->vertical = ver;
}

/* OctreeLikeRayTracer.cpp:361-362. */
void orc_generate_ray(const orc_camera_frame* f, float u, float v, float dir[3]) {
    orc_vec3 d = v_normalize(v_sub(v_add(v_add(f->lower_left, v_mul(f->horizontal, u)),
                                         v_mul(f->vertical, v)), f->origin));
    dir[0] = d.x; dir[1] = d.y; dir[2] = d.z;
}

/* ---------------------------------------------------------------- ray / triangle */

/* GeometryUtils.cpp:57-97 (Moller-Trumbore with FLT_EPSILON tolerances; abs is the float
 * overload, as on the reference's MSVC platform -- SURVEY.md F8). */
static inline int ray_tri(orc_vec3 v0, orc_vec3 v1, orc_vec3 v2, orc_vec3 d, orc_vec3 o, float* t_out) {
    const float EPS = FLT_EPSILON;
    orc_vec3 e1 = v_sub(v1, v0);
    orc_vec3 e2 = v_sub(v2, v0);
    orc_vec3 p = v_cross(d, e2);
    float det = v_dot(e1, p);
    if (det > -EPS && det < EPS) return 0;
    float inv = (float)(1.0 / (double)det);                 /* :74, double divide narrowed */
    orc_vec3 s = v_sub(o, v0);
    float u = inv * v_dot(s, p);
    if ((u < 0 && fabsf(u) > EPS) || (u > 1 && fabsf(u - 1) > EPS)) return 0;
    orc_vec3 q = v_cross(s, e1);
    float v = inv * v_dot(d, q);
    if ((v < 0 && fabsf(v) > EPS) || (u + v > 1 && fabsf(u + v - 1) > EPS)) return 0;
    float t = inv * v_dot(e2, q);
    *t_out = t;                                              /* :88 writes t before the test */
    return t > EPS;
}

int orc_ray_triangle(const float* tri, const float dir[3], const float org[3], float* t) {
    orc_vec3 v0 = { tri[0], tri[1], tri[2] }, v1 = { tri[3], tri[4], tri[5] }, v2 = { tri[6], tri[7], tri[8] };
    orc_vec3 d = { dir[0], dir[1], dir[2] }, o = { org[0], org[1], org[2] };
    return ray_tri(v0, v1, v2, d, o, t);
}

/* ---------------------------------------------------------------- traversal */

/* rayBoxIntersection, GeometryUtils.cpp:15-37: reciprocal directions, products, nested std::max / std::min
 * (a NaN or an empty slab falls out through the final compares). */
int orc_ray_box(const float org[3], const float dir[3], const float bmin[3], const float bmax[3],
                float* t_near, float* t_far) {
    float inv[3], t1[3], t2[3];
    for (int k = 0; k < 3; ++k) {
        inv[k] = 1.0f / dir[k];
        t1[k] = (bmin[k] - org[k]) * inv[k];
        t2[k] = (bmax[k] - org[k]) * inv[k];
    }
    *t_near = max2(max2(min2(t1[0], t2[0]), min2(t1[1], t2[1])), min2(t1[2], t2[2]));
    *t_far = min2(min2(max2(t1[0], t2[0]), max2(t1[1], t2[1])), max2(t1[2], t2[2]));
    return *t_far >= *t_near && *t_far >= 0.0f;
}

void orc_trace(const orc_scene* s, const float org[3], const float dir[3],
               orc_hit* out, uint64_t counters[2]) {
    const int n[3] = { s->nx, s->ny, s->nz };
    const float mn[3] = { s->min.x, s->min.y, s->min.z };
    const float vs[3] = { s->vs.x, s->vs.y, s->vs.z };
    int i[3], step[3];
    float t_max[3], t_delta[3];

    /* RayHit() + traverseVoxelGrid prologue, OctreeLikeRayTracer.cpp:310-312. */
    memset(out, 0, sizeof(*out));
    out->hit = 0;
    out->t = FLT_MAX;

    float start[3] = { org[0], org[1], org[2] };
    if (s->mode == 1) {
        /* nearest-hit mode: the traversal starts where the ray enters the grid (or at the origin inside it) */
        const float bmin[3] = { s->min.x, s->min.y, s->min.z }, bmax[3] = { s->max.x, s->max.y, s->max.z };
        float tn, tf;
        if (!orc_ray_box(org, dir, bmin, bmax, &tn, &tf)) return;
        const float t0 = max2(tn, 0.0f);
        for (int k = 0; k < 3; ++k) { const float m = dir[k] * t0; start[k] = org[k] + m; }
    }
    /* initializeTraversal, :155-195 (start voxel is the CLAMPED voxel of the origin). */
    for (int k = 0; k < 3; ++k) {
        i[k] = clampi(f2i((start[k] - mn[k]) / vs[k]), 0, n[k] - 1);    /* :132-135 */
        step[k] = (dir[k] > 0) ? 1 : -1;                                 /* :170-172 */
        t_delta[k] = fabsf(vs[k] / dir[k]);                              /* :175-177 */
        int b = (dir[k] > 0) ? (i[k] + 1) : i[k];                        /* :181 */
        float boundary = mn[k] + (float)b * vs[k];                       /* :182 */
        t_max[k] = (boundary - org[k]) / dir[k];                         /* :183 */
    }

    orc_vec3 o = { org[0], org[1], org[2] }, d = { dir[0], dir[1], dir[2] };
    uint64_t nv = 0, nt = 0;
    const int max_steps = s->nx + s->ny + s->nz;                         /* :318 */
    for (int it = 0; it < max_steps; ++it) {
        if (!(i[0] >= 0 && i[0] < n[0] && i[1] >= 0 && i[1] < n[1] && i[2] >= 0 && i[2] < n[2]))
            break;                                                       /* :145-149, :321 */
        /* testVoxelTriangles, :267-295: every success overwrites the record, so the last
         * list entry that hits wins; no nearest-t, no in-voxel test. */
        ++nv;
        uint64_t f = (uint64_t)i[0] + (uint64_t)n[0] * ((uint64_t)i[1] + (uint64_t)n[1] * i[2]);
        uint32_t b0 = s->offsets[f], b1 = s->offsets[f + 1];
        int found = 0;
        for (uint32_t j = b0; j < b1; ++j) {
            uint32_t ti = s->indices[j];
            const float* tr = s->tris + 12ull * ti;
            orc_vec3 v0 = { tr[0], tr[1], tr[2] }, v1 = { tr[3], tr[4], tr[5] }, v2 = { tr[6], tr[7], tr[8] };
            float t;
            ++nt;
            if (ray_tri(v0, v1, v2, d, o, &t) && (s->mode == 0 || t < out->t)) {
                out->hit = 1; out->t = t;
                out->normal[0] = tr[9]; out->normal[1] = tr[10]; out->normal[2] = tr[11];
                out->point[0] = tr[0]; out->point[1] = tr[1]; out->point[2] = tr[2];  /* = v0, :286 */
                out->origin[0] = org[0]; out->origin[1] = org[1]; out->origin[2] = org[2];
                out->triangle_index = ti;
                found = 1;
            }
        }
        if (s->mode == 0) {
            if (found) break;                                            /* :326-329 */
        } else {
            /* the exit of this voxel is the smallest tMax as advanceToNextVoxel picks it; a hit at or before it
             * cannot be beaten by anything in a later voxel (whose list would also be in a visited voxel's) */
            const float t_next = (t_max[0] < t_max[1]) ? ((t_max[0] < t_max[2]) ? t_max[0] : t_max[2])
                                                       : ((t_max[1] < t_max[2]) ? t_max[1] : t_max[2]);
            if (out->hit && out->t <= t_next) break;
        }
        /* advanceToNextVoxel, :201-260: literal nested strict '<'; ties and NaNs fall to else. */
        if (t_max[0] < t_max[1]) {
            if (t_max[0] < t_max[2]) { i[0] += step[0]; t_max[0] += t_delta[0]; }
            else                     { i[2] += step[2]; t_max[2] += t_delta[2]; }
        } else {
            if (t_max[1] < t_max[2]) { i[1] += step[1]; t_max[1] += t_delta[1]; }
            else                     { i[2] += step[2]; t_max[2] += t_delta[2]; }
        }
    }
    if (counters) { counters[0] += nv; counters[1] += nt; }
}

/* ---------------------------------------------------------------- shading */

/* OctreeLikeRayTracer.cpp:27-40 + Shaders.cpp:4-22.  With r=g=b=kd=1 and inShadow=0 every
 * multiply by one is exact, so each channel is exactly 0.1f + max(0, N.L). */
void orc_shade(const orc_scene* s, const orc_hit* h, float v, float rgb[3]) {
    (void)s;
    if (h->hit) {
        orc_vec3 P = { h->point[0], h->point[1], h->point[2] };
        orc_vec3 O = { h->origin[0], h->origin[1], h->origin[2] };
        orc_vec3 Nn = { h->normal[0], h->normal[1], h->normal[2] };
        orc_vec3 L = v_normalize(v_sub(O, P));
        orc_vec3 N = v_normalize(Nn);
        float ndl = v_dot(N, L);
        float k = (0.0f < ndl) ? ndl : 0.0f;               /* std::max(0.0f, dot) */
        float diffuse = (1.0f * 1.0f) * k;                 /* multiply(scale(kd, light), dotNL) */
        float ambient = 1.0f * 0.1f;
        float c = ambient + (1.0f - 0.0f) * diffuse;
        rgb[0] = c; rgb[1] = c; rgb[2] = c;
    } else {
        rgb[0] = 0.5f + 0.3f * v; rgb[1] = 0.5f + 0.3f * v; rgb[2] = 0.7f + 0.3f * v;
    }
}

/* ---------------------------------------------------------------- jitter (extension) */

static inline uint32_t mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x7feb352dU; h ^= h >> 15; h *= 0x846ca68bU; h ^= h >> 16;
    return h;
}

void orc_jitter(uint32_t x, uint32_t y, uint32_t s, uint32_t seed, float* jx, float* jy) {
    uint32_t h = mix32(x + 0x9E3779B9U * (y + 0x85EBCA6BU * (s + 0xC2B2AE35U * seed)));
    uint32_t a = mix32(h ^ 0x68E31DA4U), b = mix32(h ^ 0xB5297A4DU);
    *jx = (float)(a >> 8) * (1.0f / 16777216.0f);
    *jy = (float)(b >> 8) * (1.0f / 16777216.0f);
}

/* ---------------------------------------------------------------- render */

void orc_render_rows(const orc_scene* s, const orc_camera_frame* f, int W, int H,
                     int y0, int y1, int spp, uint32_t seed,
                     float* rgb, uint32_t* hit_id, float* t_out, uint64_t counters[2]) {
    const float org[3] = { f->origin.x, f->origin.y, f->origin.z };
    for (int y = y0; y < y1; ++y) {
        for (int x = 0; x < W; ++x) {
            float acc[3] = { 0.0f, 0.0f, 0.0f };
            for (int sidx = 0; sidx < spp; ++sidx) {
                float jx = 0.0f, jy = 0.0f;
                if (spp > 1) orc_jitter((uint32_t)x, (uint32_t)y, (uint32_t)sidx, seed, &jx, &jy);
                /* renderPixel, :73-74: u = x/W, v = y/H (pixel corner; jitter is 0 at spp 1). */
                float u = ((float)x + jx) / (float)W;
                float v = ((float)y + jy) / (float)H;
                float dir[3];
                orc_generate_ray(f, u, v, dir);
                orc_hit h;
                orc_trace(s, org, dir, &h, counters);
                float c[3];
                orc_shade(s, &h, v, c);
                if (spp == 1) { acc[0] = c[0]; acc[1] = c[1]; acc[2] = c[2]; }
                else { acc[0] += c[0]; acc[1] += c[1]; acc[2] += c[2]; }
                if (sidx == 0) {
                    size_t p = (size_t)y * W + x;
                    if (hit_id) hit_id[p] = h.hit ? h.triangle_index : 0xFFFFFFFFu;
                    if (t_out) t_out[p] = h.t;
                }
            }
            if (rgb) {
                size_t p = ((size_t)y * W + x) * 3;
                if (spp == 1) { rgb[p] = acc[0]; rgb[p + 1] = acc[1]; rgb[p + 2] = acc[2]; }
                else { rgb[p] = acc[0] / (float)spp; rgb[p + 1] = acc[1] / (float)spp; rgb[p + 2] = acc[2] / (float)spp; }
            }
        }
    }
}
