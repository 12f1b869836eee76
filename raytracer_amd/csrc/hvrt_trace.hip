// hvrt_trace.hip -- primary-ray grid traversal kernels for gfx950 (MI355X).
//
// Replaces OctreeLikeRayTracer::render / traceRay (OctreeLikeRayTracer.cpp:88-126) with a
// persistent-thread kernel: waves pull 8x8 pixel tiles from a work-queue head, every lane runs
// the reference's Amanatides-Woo 3D-DDA (OctreeLikeRayTracer.cpp:155-260, 309-336) and
// Moller-Trumbore test (GeometryUtils.cpp:57-97) with the reference's exact fp32 operation order,
// NaN/inf control flow included, then shades (OctreeLikeRayTracer.cpp:27-40,
// Shaders.cpp:4-22).  Compile with -ffp-contract=off: a*b+c must stay v_mul + v_add.
#include "hvrt_internal.hpp"

#include <stddef.h>
#include <stdlib.h>

#include <float.h>
#include <limits.h>

namespace hvrt {
namespace {

constexpr float EPS = FLT_EPSILON;   // GeometryUtils.cpp:64
#if defined(HVRT_DIAG) && (HVRT_DIAG == 33 || HVRT_DIAG == 34)
constexpr bool HVRT_DIAG_NO_SPLIT_COUNTS = true;
#else
constexpr bool HVRT_DIAG_NO_SPLIT_COUNTS = false;
#endif
#ifdef HVRT_DIAG
constexpr bool HVRT_DIAG_BUILD = true;
#else
constexpr bool HVRT_DIAG_BUILD = false;
#endif

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// std::min / std::max as libstdc++ / MSVC define them (GeometryUtils.cpp:33-34 nests them).
__device__ __forceinline__ float min2(float a, float b) { return b < a ? b : a; }
__device__ __forceinline__ float max2(float a, float b) { return a < b ? b : a; }

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
__device__ __forceinline__ V3 add(V3 a, V3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
__device__ __forceinline__ V3 mul(V3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }   // ((ax*bx)+(ay*by))+(az*bz)
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x };
}
// VectorMath.h:9-15
__device__ __forceinline__ V3 normalize(V3 v) {
    float len = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    if (len > 0.0f) return { v.x / len, v.y / len, v.z / len };
    return v;
}

// GeometryUtils.cpp:57-97 on a (v0, e1, e2) record.
//
// The reference divides first (inv_det = 1/det) and then tests u, v, u+v against [0,1] with FLT_EPSILON
// slack.  Most tests fail those bounds by far, so the IEEE divide (11 instructions) is deferred: an
// approximate reciprocal (v_rcp_f32, 1 ulp) gives u~, v~ within 2^-21 relative of the exact values, and a
// test is rejected early only when u~ < -1e-3, u~ > 1.001, v~ < -1e-3 or u~ + v~ > 1.002 -- each of which
// implies the reference's own rejection.  Everything else (including every NaN / inf case, which fails the
// early compares) takes the exact path below, so hit / miss and t are bit-identical.
// on_hit(t) runs where the hit is established (nested ifs, not early returns + a flag tested by the caller): a
// rejected test then skips straight to the loop's end through one exec-mask restore -- every mask operation in
// this loop is an instruction of the CU's one scalar unit.
// (ray_triangle_s: the same test given s = o - v0 instead of o and v0 -- the subtraction is the reference's own,
// GeometryUtils.cpp:73, whoever performs it: where all rays of a wave share their origin and the record comes from the
// wave's LDS window, s is the same number in every lane and is worked out once per record, not once per lane and test)
// (dc: diagnostic builds only -- two counters for scripts/diag_counts.py)
#if defined(HVRT_DIAG) && (HVRT_DIAG == 25 || HVRT_DIAG == 26)
#define HVRT_DIAG_STAGE(n, slot) if (dc && (HVRT_DIAG == (n)) && (int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) dc[slot] += 64;
#else
#define HVRT_DIAG_STAGE(n, slot)
#endif
template <class OnHit>
__device__ __forceinline__ void ray_triangle_s(V3 s, V3 e1, V3 e2, V3 d, OnHit on_hit, uint32_t* dc = nullptr) {
    (void)dc;
    HVRT_DIAG_STAGE(25, 0)      // 25: wave-level tests | of them, those in which some lane goes on to v
    V3 p = cross(d, e2);
    float det = dot(e1, p);
    float su = dot(s, p);
    const float r = __builtin_amdgcn_rcpf(det);
    const float ua = su * r;
    // one exit for "parallel" (the reference's first test) and "u far outside": no branch sits between the
    // record's loads, so they are issued together instead of v0 waiting for the determinant
    if (!((det > -EPS && det < EPS) | (ua < -1e-3f) | (ua > 1.001f))) {
        HVRT_DIAG_STAGE(25, 1)
        V3 q = cross(s, e1);
        float sv = dot(d, q);
        const float va = sv * r;
        if (!(va < -1e-3f || ua + va > 1.002f)) {
            HVRT_DIAG_STAGE(26, 0)      // 26: wave-level tests that reach the exact divide | that hit
            float inv = 1.0f / det;   // == (float)(1.0 / (double)det): double rounding is innocuous for 53 >= 2*24+2
            float u = inv * su;
            if (!((u < 0.0f && fabsf(u) > EPS) || (u > 1.0f && fabsf(u - 1.0f) > EPS))) {
                float v = inv * sv;
                if (!((v < 0.0f && fabsf(v) > EPS) || (u + v > 1.0f && fabsf(u + v - 1.0f) > EPS))) {
                    float t = inv * dot(e2, q);
                    if (t > EPS) { HVRT_DIAG_STAGE(26, 1) on_hit(t); }
                }
            }
        }
    }
}
template <class OnHit>
__device__ __forceinline__ void ray_triangle(V3 v0, V3 e1, V3 e2, V3 o, V3 d, OnHit on_hit) {
    ray_triangle_s(sub(o, v0), e1, e2, d, on_hit);
}

// the mask of lanes for which b holds, straight from the compare (HIP's __ballot goes through an integer)
__device__ __forceinline__ unsigned long long any_lane(bool b) { return __builtin_amdgcn_ballot_w64(b); }

// The lane number, worked out where it is needed (two instructions) rather than kept in a register for the whole
// kernel: k_render has none to spare.  The opaque zero keeps the compiler from hoisting it back out of the loops.
__device__ __forceinline__ int lane_now() {
    unsigned z = 0;
    asm volatile("" : "+v"(z));
    return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
}

// Selects on a lane mask held in a scalar register pair, in the VOP3 encoding.  Why by hand: for `c ? a : b` with c
// the AND of two compares the compiler forms the mask with s_and_b64 into vcc and selects with the VOP2
// v_cndmask_b32, and on gfx950 a VOP2 v_cndmask_b32 whose vcc was last written by the SCALAR unit blocks its SIMD for
// ~19 clocks, however many waves are resident (scripts/ubench/valu_rate.hip: 19.5 clk for s_mov vcc + v_cndmask,
// 6.1 for v_cmp vcc + v_cndmask, 4.0 for the VOP3 form with any mask).  The DDA step had one such select.
typedef unsigned long long lanemask;
__device__ __forceinline__ float sel(lanemask m, float if_set, float if_clear) {
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(m));
    return r;
}
__device__ __forceinline__ uint32_t sel(lanemask m, uint32_t if_set, uint32_t if_clear) {
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(m));
    return r;
}
// static_cast<int> as the reference platform (x86 cvttss2si) does it: INT_MIN for NaN and out-of-range values
// (v_cvt_i32_f32 saturates instead); the OR of the two compares is a scalar-unit mask, hence the VOP3 select
__device__ __forceinline__ int f2i(float f) {
    const lanemask bad = __builtin_amdgcn_ballot_w64(!(f < 2147483648.0f)) | __builtin_amdgcn_ballot_w64(f < -2147483648.0f);
    return (int)sel(bad, 0x80000000u, (uint32_t)(int)f);
}
// v - 1 where the lane's bit is set (one VOP3 instruction; the borrow out goes nowhere)
__device__ __forceinline__ int dec_where(lanemask m, int v) {
    lanemask borrow;
    asm("v_subbrev_co_u32_e64 %0, %1, 0, %0, %2" : "+v"(v), "=s"(borrow) : "s"(m));
    return v;
}
// advanceToNextVoxel (:201-260) on explicit lane masks.  cxy picks the x-or-y candidate, c2 compares that candidate
// with z: the nested ifs of :204-259 (ties and NaNs fall to the else arms exactly as written there).
// (Tried: the step in assembly, each v_cmp followed at once by VOP2 selects on its fresh vcc -- 2 clocks each instead of
// 4 -- and the masks copied out of vcc for the write-backs: 42 clocks of issue on paper instead of 52, but config 4 went
// from 73.5 to 79.0 ms; the scalar copies of vcc wait for the vector pipeline twice per step.)
#define HVRT_STEP()                                                             \
    do {                                                                        \
        const lanemask cxy = any_lane(tmx < tmy);                               \
        const lanemask c2 = any_lane(sel(cxy, tmx, tmy) < tmz);                 \
        const lanemask stx = cxy & c2, sty = c2 & ~cxy;                         \
        tmx = sel(stx, tmx + tdx, tmx);                                         \
        tmy = sel(sty, tmy + tdy, tmy);                                         \
        tmz = sel(c2, tmz, tmz + tdz);                                          \
        rx = dec_where(stx, rx); ry = dec_where(sty, ry);                       \
        last = sel(c2, sel(cxy, fsx, fsy), fsz);                                \
        f += last;                                                              \
    } while (0)

// A run of k empty-space steps during which one axis (the "far" one, C) is known not to step: the other two (A
// before B in x, y, z order) then advance by `tmA < tmB ? A : B` alone -- a compare, two adds, two selects on the
// compare's own vcc and a borrow, a third of the general step's instruction time.  Which of the reference's nested
// compares (:204-259) that is, per far axis:
//   z far: `(tmx < tmy ? tmx : tmy) < tmz` holds at every step  -> step x or y by tmx < tmy           (A = x, B = y)
//   y far: `tmx < tmy` holds at every step                       -> step x or z by tmx < tmz           (A = x, B = z)
//   x far: `tmx < tmy` fails at every step                       -> step y or z by tmy < tmz           (A = y, B = z)
// The values compared only grow along a run (tDelta >= 0; a NaN fails every check), so each condition holds at
// every step of the run iff it holds at the LAST step, which `OK` tests on the values that step compared (preA,
// preB, c).  The run is speculative: the caller picks the far axis from a cheap estimate, OK decides exactly; when it
// fails for any lane the state is rolled back and the general steps are taken.  On success the flat index and
// the counters are brought up to date from the number of A steps (only rA is kept per step).
#if defined(HVRT_DIAG) && HVRT_DIAG == 24   // steps of two-axis runs in which every lane steps ONE axis only | the other two-axis steps
#define HVRT_DIAG24_HOOK if (any_lane(!(dA == 0u || dB == 0u)) == 0ull) r.nv += (uint32_t)k; else r.nt += (uint32_t)k;
#else
#define HVRT_DIAG24_HOOK
#endif
#define HVRT_TWO_AXIS_RUN(tmA, tdA, rA, fsA, tmB, tdB, fsB, OK, AFTER)                               \
    do {                                                                                             \
        const float saveA = tmA, saveB = tmB;                                                        \
        const int save_rA = rA;                                                                      \
        for (int i = 1; i < k; ++i) {                                                                \
            const bool c = tmA < tmB;                                                                \
            const float na = tmA + tdA, nb = tmB + tdB;                                              \
            tmA = c ? na : tmA; tmB = c ? tmB : nb; rA -= c ? 1 : 0;                                 \
        }                                                                                            \
        const float preA = tmA, preB = tmB;                                                          \
        const bool c = tmA < tmB;                                                                    \
        { const float na = tmA + tdA, nb = tmB + tdB; tmA = c ? na : tmA; tmB = c ? tmB : nb; rA -= c ? 1 : 0; } \
        (void)preA; (void)preB;                                                                      \
        if (any_lane(!(OK)) == 0ull) {                                                               \
            const uint32_t dA = (uint32_t)(save_rA - rA), dB = (uint32_t)k - dA;                     \
            HVRT_DIAG24_HOOK                                                                         \
            f += dA * fsA + dB * fsB;                                                                \
            last = c ? fsA : fsB;                                                                    \
            AFTER;                                                                                   \
            ran = true;                                                                              \
        } else {                                                                                     \
            tmA = saveA; tmB = saveB; rA = save_rA;                                                  \
        }                                                                                            \
    } while (0)

struct TraceResult {
    uint32_t tri;   // HVRT_MISS when nothing was hit
    float t;
    uint32_t nv, nt;
};

// traverseVoxelGrid, OctreeLikeRayTracer.cpp:309-336, in "while-while" form: every lane first steps
// through empty voxels, so that the lanes of a wave reach their next occupied voxel together and the
// triangle loop runs with most lanes active.  The visited voxel sequence, the tests and their order are
// the reference's; only the bookkeeping differs:
//   * the start voxel is in bounds (it is clamped) and every step moves one axis by its fixed +-1, so
//     a ray visits at most nx+ny+nz-2 voxels: the reference's step cap (:318) can never fire and only
//     the stepped axis needs a bounds check -- kept as "steps left before leaving the grid" counters;
//   * the flat voxel index ix + nx*(iy + ny*iz) is updated incrementally (32-bit: Nvox < 2^32);
//   * advanceToNextVoxel's nested strict '<' (:201-260) becomes three compares and selects with the same
//     truth table (ties and NaNs fall to the else arms exactly as written there).
//
// NEAREST (hvrt_set_mode(1); an extension, SURVEY 8f row 4 -- the reference has no such mode): the ray enters
// the grid through the reference's own rayBoxIntersection (GeometryUtils.cpp:15-37, unused by its traversal),
// the nearest hit over all visited voxels is kept, and the walk stops once that hit lies at or before the
// exit of the current voxel -- nothing in a later voxel can beat it, because a triangle hit before that exit
// is listed in a voxel already visited.  The skip of entries tested one step ago stays valid (their result
// is remembered in the running minimum).  Same kernels, same data.
//
// RAYS < 64 (k_render_split, the tiles that alone would outlast the rest of the launch): the wave holds RAYS rays,
// each on 64 / RAYS lanes ("slices": lane = slice * RAYS + ray).  The slices of a ray walk in lockstep and share
// out the entries of every list (entry k of a chunk goes to slice k mod 64/RAYS); after the list, the slices'
// candidates are reduced to the one the sequential loop would have kept -- the highest entry that hit (mode 0),
// the smallest t, the lower entry on a tie (nearest-hit mode) -- so the result is the same bit for bit.
//
// ABORT (k_render): a walk that is still inside a list of more than 32 entries when the 100 MHz clock passes `deadline`
// gives up (tri = HVRT_ABORTED).  The clock is a scalar, so all lanes that are in such a list give up together;
// the kernel then hands the whole tile to k_render_split.  (trace_ray_wave also looks at the clock at every occupied voxel
// it reaches.)
constexpr uint32_t HVRT_ABORTED = 0xFFFFFFFEu;
constexpr uint32_t COST_SPLIT_FLAG = 0x80000000u;   // top bit of a tile's cost word: the tile goes to k_render_split
template <bool COUNT, bool INLINE, bool NEAREST, int RAYS = 64, bool ABORT = false>
__device__ __forceinline__ TraceResult trace_ray(const SceneDev& sc, V3 o, V3 d, long long deadline = 0) {
    const GridParams& g = sc.g;
    TraceResult r;
    r.tri = HVRT_MISS; r.t = FLT_MAX; r.nv = 0; r.nt = 0;
    bool aborted = false;

    V3 st = o;
    if (NEAREST) {
        const float ivx = 1.0f / d.x, ivy = 1.0f / d.y, ivz = 1.0f / d.z;
        const float tx1 = (g.mnx - o.x) * ivx, tx2 = (g.bxx - o.x) * ivx;
        const float ty1 = (g.mny - o.y) * ivy, ty2 = (g.bxy - o.y) * ivy;
        const float tz1 = (g.mnz - o.z) * ivz, tz2 = (g.bxz - o.z) * ivz;
        const float tn = max2(max2(min2(tx1, tx2), min2(ty1, ty2)), min2(tz1, tz2));
        const float tf = min2(min2(max2(tx1, tx2), max2(ty1, ty2)), max2(tz1, tz2));
        if (!(tf >= tn && tf >= 0.0f)) return r;
        const float t0 = max2(tn, 0.0f);
        st = { o.x + d.x * t0, o.y + d.y * t0, o.z + d.z * t0 };
    }
    // initializeTraversal, :155-195 (start voxel = CLAMPED voxel of the origin, no entry test)
    const int ix = clampi(f2i((st.x - g.mnx) / g.vsx), 0, g.nx - 1);
    const int iy = clampi(f2i((st.y - g.mny) / g.vsy), 0, g.ny - 1);
    const int iz = clampi(f2i((st.z - g.mnz) / g.vsz), 0, g.nz - 1);
    const bool px = d.x > 0.0f, py = d.y > 0.0f, pz = d.z > 0.0f;   // stepX = (dir > 0) ? 1 : -1
    const float tdx = fabsf(g.vsx / d.x), tdy = fabsf(g.vsy / d.y), tdz = fabsf(g.vsz / d.z);
    float tmx = ((g.mnx + (float)(px ? ix + 1 : ix) * g.vsx) - o.x) / d.x;
    float tmy = ((g.mny + (float)(py ? iy + 1 : iy) * g.vsy) - o.y) / d.y;
    float tmz = ((g.mnz + (float)(pz ? iz + 1 : iz) * g.vsz) - o.z) / d.z;
    int rx = px ? g.nx - 1 - ix : ix;      // steps left on each axis before the ray leaves the grid
    int ry = py ? g.ny - 1 - iy : iy;
    // z has no counter of its own: z steps = all steps - x steps - y steps, so its steps-left value is
    // zsum - tot - rx - ry (tot = steps taken so far; a run of k steps adds k once, not 1 k times)
    const int zsum = (pz ? g.nz - 1 - iz : iz) + rx + ry;
    int tot = 0;
    const uint32_t plane = (uint32_t)g.nx * (uint32_t)g.ny;
    const uint32_t fsx = px ? 1u : 0xFFFFFFFFu;
    const uint32_t fsy = py ? (uint32_t)g.nx : 0u - (uint32_t)g.nx;
    const uint32_t fsz = pz ? plane : 0u - plane;
    uint32_t f = (uint32_t)ix + (uint32_t)g.nx * (uint32_t)iy + plane * (uint32_t)iz;

    const uint32_t* __restrict__ offs = sc.offsets;
    // face through which the current voxel was entered: 0 -x, 1 +x, 2 -y, 3 +y, 4 -z, 5 +z neighbour
    // (kept as the last step's index increment; decoded only when a voxel has a list to test)
    // (kept as the word offset of that face's bit array, so that no multiply sits in the loop:
    // 6 * face_words < 2^30 because there are fewer than 2^32 entries)
    const uint32_t fw = (uint32_t)sc.face_words;
    const uint32_t facex = (px ? 0u : 1u) * fw, facey = (py ? 2u : 3u) * fw, facez = (pz ? 4u : 5u) * fw;
    uint32_t last = 0;   // 0 = no step taken yet (an increment is never 0)
    uint32_t b0, b1;
#ifndef HVRT_DIAG
    if (COUNT) r.nv = 1;
#endif

    // the steps-left counters only ever decrease: once one is negative the ray has left the grid for good
#define HVRT_IN_GRID() ((rx | ry | (zsum - tot - rx - ry)) >= 0)
    // nearest-hit mode: is the kept hit at or before the exit of the current voxel (the tMax the next step takes)?
#define HVRT_NEAREST_DONE()                                                     \
    (r.tri != HVRT_MISS && r.t <= (((tmx < tmy ? tmx : tmy) < tmz) ? (tmx < tmy ? tmx : tmy) : tmz))
#define HVRT_LOAD_VOXEL()                                                       \
    do {                                                                        \
        const uint32_t* pv = offs + f;                                          \
        b0 = pv[0]; b1 = pv[1];                                                 \
    } while (0)

    const uint8_t* __restrict__ dist = sc.dist;
    for (;;) {
        // phase 1: walk to the next occupied voxel (testVoxelTriangles on an unoccupied voxel does nothing).
        // dist[f] = city-block distance to the nearest occupied voxel (0 = occupied): a ray moves one face
        // per step, so the next dist-1 voxels are empty whatever the ray does and are stepped through without
        // being looked at.  The look-ups stay wave-synchronous -- all searching lanes read, then all take
        // k <= min(dist) steps (k a power of two, found with ballots: one compare near a surface, where
        // k = 1, a few in open space) -- because a vector load costs the L1 a slot per quad of lanes, however
        // few of them want it.
        bool search = true;      // every lane that gets here is in the grid
        bool done = false;
        for (;;) {
            uint32_t dd = 255u;
            if (search) dd = dist[f];
            search = search && dd != 0u;
            if (any_lane(search) == 0ull) break;
            const uint32_t ds = search ? dd : 255u;
            int k = 1;
            if (any_lane(ds < 2u) == 0ull) {
                k = 2;
                if (any_lane(ds < 4u) == 0ull) {
                    k = 4;
                    if (any_lane(ds < 8u) == 0ull) {
                        k = 8;
                        if (any_lane(ds < 16u) == 0ull) k = any_lane(ds < 32u) == 0ull ? 32 : 16;
                    }
                }
            }
            // The k steps run without per-step branches: a lane that leaves the grid (or, in nearest-hit mode,
            // is done) on the way keeps stepping -- its state is not used again -- and only remembers that it
            // did (the counters and `done` are sticky).  Per-step exec-mask bookkeeping was two fifths of the
            // kernel's scalar instructions, and the scalar unit is shared by the CU's four SIMDs.
            // (The steps-left counters only ever decrease, so "still in the grid" needs one look after the k
            // steps, not one per step, unless the steps are being counted.)
            if (search) {
                if (COUNT || NEAREST) {
                    bool going = true;
                    for (int i = 0; i < k; ++i) {
                        if (NEAREST) { done = done || HVRT_NEAREST_DONE(); going = going && !done; }
                        HVRT_STEP();
                        ++tot;
                        going = going && HVRT_IN_GRID();
#ifndef HVRT_DIAG
                        if (COUNT) r.nv += going ? 1u : 0u;
#endif
                    }
                    search = going;
                } else {
                    for (int i = 0; i < k; ++i) HVRT_STEP();
                    tot += k;
                    search = HVRT_IN_GRID();
                }
            }
        }
        if (NEAREST && done) break;
        if (!HVRT_IN_GRID()) break;
        HVRT_LOAD_VOXEL();
        // a surviving x step implies nx > 1 (it would have left a 1-wide grid), so fsx, fsy, fsz are told
        // apart by testing the axes in this order with their extent
        const bool entered = last != 0u;          // false: the ray's first voxel, nothing was tested before
        const uint32_t face = sel(any_lane(last == fsx) & (g.nx > 1 ? ~0ull : 0ull), facex, sel(any_lane(last == fsy) & (g.ny > 1 ? ~0ull : 0ull), facey, facez));
        // phase 2: testVoxelTriangles, :267-295 -- every entry is tested, the LAST one that hits wins
        uint32_t bkey = 0;   // RAYS < 64: 1 + list position of this slice's candidate, 0 = none from this list
        {
            // Entries that are also in the list of the voxel the ray just left were tested there one step
            // ago with this very ray and missed (a hit would have ended it), so they are not tested again:
            // sc.face_bits holds, per face, one bit per CSR entry ("also listed in that face neighbour"),
            // read as a 32-entry window starting at b0.  Skipping never changes a result; it halves the
            // records a lane has to fetch (the loop is bound by L1 data return, 40 B per lane and test).
            const uint32_t n = b1 - b0;
            constexpr uint32_t SLICES = 64u / (uint32_t)RAYS;
            constexpr uint32_t SLICE_BITS = SLICES == 16u ? 0x00010001u : SLICES == 8u ? 0x01010101u : SLICES == 4u ? 0x11111111u : 0xFFFFFFFFu;
            static_assert(RAYS == 64 || RAYS == 16 || RAYS == 8 || RAYS == 4, "slice masks are written for 1, 4, 8 and 16 slices");
            const uint32_t slice_mask = RAYS == 64 ? 0xFFFFFFFFu : SLICE_BITS << ((threadIdx.x & 63u) / (uint32_t)RAYS);
            for (uint32_t c = 0; c < n; c += 32) {
                if (ABORT && c != 0u && wall_clock64() > (unsigned long long)deadline) { aborted = true; break; }
                const uint32_t base = b0 + c;
                uint32_t skip = 0;
                if (entered) {
                    const uint32_t* __restrict__ fb = sc.face_bits + (face + (base >> 5));
                    skip = __builtin_amdgcn_alignbit(fb[1], fb[0], base & 31u);
                }
                const uint32_t left = n - c;
                uint32_t todo = ~skip & (left >= 32 ? 0xFFFFFFFFu : ((1u << left) - 1u));
                if (RAYS < 64) todo &= slice_mask;
                while (todo) {
                    const uint32_t k = (uint32_t)__builtin_ctz(todo);
                    todo &= todo - 1;
                    const uint32_t j = base + k;
#ifdef HVRT_DIAG   // diagnostic build only: nt = tests executed, nv = 64 x wave-level test iterations
#if HVRT_DIAG == 2   // nt = 64 x wave-level iterations in which every active lane tests the SAME entry
                    if (COUNT) {
                        const bool lead = (int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1;
                        const bool same = __ballot(j != (uint32_t)__builtin_amdgcn_readfirstlane((int)j)) == 0ull;
                        if (lead) { r.nv += 64; if (same) r.nt += 64; }
                    }
#else
                    if (COUNT) { ++r.nt; if ((int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) r.nv += 64; }
#endif
#endif
                    if (INLINE) {
                        const float4* __restrict__ rec = sc.ref_rec + 3ull * j;
                        // 36 of the record's 48 bytes; the triangle id (10th dword) is fetched only on a hit
                        const float4 c0 = rec[0], c1 = rec[1];
                        const float e2z = reinterpret_cast<const float*>(rec + 2)[0];
                        ray_triangle({ c0.x, c0.y, c0.z }, { c0.w, c1.x, c1.y }, { c1.z, c1.w, e2z }, o, d, [&](float t) {
                            if (!NEAREST || t < r.t) {
                                r.tri = __float_as_uint(reinterpret_cast<const float*>(rec + 2)[1]); r.t = t;
                                if (RAYS < 64) bkey = c + k + 1u;
                            }
                        });
                    } else {
                        const uint32_t ti = sc.indices[j];
                        const float4 r0 = sc.tri_rec[3ull * ti], r1 = sc.tri_rec[3ull * ti + 1];
                        const float e2z = reinterpret_cast<const float*>(sc.tri_rec + 3ull * ti + 2)[0];
                        ray_triangle({ r0.x, r0.y, r0.z }, { r0.w, r1.x, r1.y }, { r1.z, r1.w, e2z }, o, d, [&](float t) {
                            if (!NEAREST || t < r.t) {
                                r.tri = ti; r.t = t;
                                if (RAYS < 64) bkey = c + k + 1u;
                            }
                        });
                    }
                }
            }
        }
        if (ABORT && aborted) { r.tri = HVRT_ABORTED; break; }
        if (RAYS < 64) {
#pragma unroll
            for (int m = RAYS; m < 64; m <<= 1) {
                const uint32_t ok = (uint32_t)__shfl_xor((int)bkey, m), otri = (uint32_t)__shfl_xor((int)r.tri, m);
                const float ot = __shfl_xor(r.t, m);
                const bool take = NEAREST ? (ot < r.t || (ot == r.t && ok < bkey)) : ok > bkey;
                if (take) { bkey = ok; r.tri = otri; r.t = ot; }
            }
        }
#ifndef HVRT_DIAG
        if (COUNT) r.nt += b1 - b0;
#endif
        if (NEAREST) { if (HVRT_NEAREST_DONE()) break; }
        else if (r.tri != HVRT_MISS) break;   // :326-329 (no flag to carry through the loops: tri is MISS until a hit)
        HVRT_STEP();
        ++tot;
        if (!HVRT_IN_GRID()) break;
#ifndef HVRT_DIAG
        if (COUNT) ++r.nv;
#endif
    }
#undef HVRT_IN_GRID
#undef HVRT_NEAREST_DONE
#undef HVRT_LOAD_VOXEL
    return r;
}

// ---- the same walk, looking ahead (k_render_split, mode 0, no counters) ---------------------------------------
// What a launch waits for in the end is a handful of ray packets that graze the occupied shell: some hundred voxels one
// after the other, each a look-up that depends on the step before it and, where the voxel holds a list, a second one for
// its records -- 0.5 to 1 us apiece with the GPU three quarters idle (scripts/tile_starts.py: the longest 4x2-pixel piece of
// config 2's front view takes 350 us, whatever the frame size).  But a ray's voxel sequence does not depend on what the
// voxels hold: the DDA's steps are arithmetic on the ray alone.  So the walk takes HVRT_LOOKAHEAD steps first, remembers
// the voxels (and the face each was entered through), fetches all their list bounds with loads that are in flight
// together, and then looks at the voxels in the ray's order: the first one with a hit ends the ray, exactly as in trace_ray
// -- same voxel sequence, same tests in the same order, same result bit for bit; the steps taken beyond the voxel that
// ends a ray are thrown away with the ray's state.  The distance field is consulted for the last voxel of a batch only:
// dist - 1 further voxels are empty whatever the ray does, and are stepped through without being looked at (32 at most,
// as in trace_ray).
#ifndef HVRT_LOOKAHEAD
#define HVRT_LOOKAHEAD 2      // (0: off.  Configs 2 / 3 with 0 / 2 / 4 / 8 voxels and the split bar where it is best for each: 0.44 / 0.415 / 0.42 / 0.43 and 0.875 / 0.80 / 0.82 / 0.91 ms)
#endif
template <bool INLINE, int RAYS>
__device__ __forceinline__ TraceResult trace_ray_ahead(const SceneDev& sc, V3 o, V3 d) {
    static_assert(RAYS == 16 || RAYS == 8 || RAYS == 4, "the slices of a ray share out every list");
    constexpr int N = HVRT_LOOKAHEAD > 0 ? HVRT_LOOKAHEAD : 1;      // (0: render_pixel does not call this)
    const GridParams& g = sc.g;
    TraceResult r;
    r.tri = HVRT_MISS; r.t = FLT_MAX; r.nv = 0; r.nt = 0;
    // initializeTraversal, :155-195, as in trace_ray
    const int ix = clampi(f2i((o.x - g.mnx) / g.vsx), 0, g.nx - 1);
    const int iy = clampi(f2i((o.y - g.mny) / g.vsy), 0, g.ny - 1);
    const int iz = clampi(f2i((o.z - g.mnz) / g.vsz), 0, g.nz - 1);
    const bool px = d.x > 0.0f, py = d.y > 0.0f, pz = d.z > 0.0f;
    const float tdx = fabsf(g.vsx / d.x), tdy = fabsf(g.vsy / d.y), tdz = fabsf(g.vsz / d.z);
    float tmx = ((g.mnx + (float)(px ? ix + 1 : ix) * g.vsx) - o.x) / d.x;
    float tmy = ((g.mny + (float)(py ? iy + 1 : iy) * g.vsy) - o.y) / d.y;
    float tmz = ((g.mnz + (float)(pz ? iz + 1 : iz) * g.vsz) - o.z) / d.z;
    int rx = px ? g.nx - 1 - ix : ix;
    int ry = py ? g.ny - 1 - iy : iy;
    const int zsum = (pz ? g.nz - 1 - iz : iz) + rx + ry;
    int tot = 0;
    const uint32_t plane = (uint32_t)g.nx * (uint32_t)g.ny;
    const uint32_t fsx = px ? 1u : 0xFFFFFFFFu;
    const uint32_t fsy = py ? (uint32_t)g.nx : 0u - (uint32_t)g.nx;
    const uint32_t fsz = pz ? plane : 0u - plane;
    uint32_t f = (uint32_t)ix + (uint32_t)g.nx * (uint32_t)iy + plane * (uint32_t)iz;
    const uint32_t* __restrict__ offs = sc.offsets;
    const uint8_t* __restrict__ dist = sc.dist;
    const uint32_t fw = (uint32_t)sc.face_words;
    const uint32_t facex = (px ? 0u : 1u) * fw, facey = (py ? 2u : 3u) * fw, facez = (pz ? 4u : 5u) * fw;
    uint32_t last = 0;
    constexpr uint32_t SLICES = 64u / (uint32_t)RAYS;
    constexpr uint32_t SLICE_BITS = SLICES == 16u ? 0x00010001u : SLICES == 8u ? 0x01010101u : 0x11111111u;
    const uint32_t slice_mask = SLICE_BITS << ((threadIdx.x & 63u) / (uint32_t)RAYS);
#define HVRT_IN_GRID() ((rx | ry | (zsum - tot - rx - ry)) >= 0)
    bool in_grid = true;      // the start voxel is clamped into the grid
    for (;;) {
        // the ray's next N voxels, from the steps alone (a voxel behind the grid's edge is remembered as such, not looked at)
        uint32_t qf[N], ql[N];
        bool qv[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            qf[i] = f; ql[i] = last; qv[i] = in_grid;
            HVRT_STEP();
            ++tot;
            in_grid = in_grid && HVRT_IN_GRID();
        }
        uint32_t qb0[N], qb1[N], dd = 0u;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            qb0[i] = 0u; qb1[i] = 0u;
            if (qv[i]) { const uint32_t* pv = offs + qf[i]; qb0[i] = pv[0]; qb1[i] = pv[1]; }
        }
        if (in_grid) dd = dist[qf[N - 1]];      // (in_grid: voxel N - 1 and the one behind it are in the grid)
        bool ended = false;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (ended) break;
            if (!qv[i]) { ended = true; break; }          // the ray left the grid before this voxel
            const uint32_t b0 = qb0[i], n = qb1[i] - qb0[i];
            if (n == 0u) continue;
            // testVoxelTriangles, :267-295, as in trace_ray: every entry is tested, the LAST one that hits wins
            const bool entered = ql[i] != 0u;
            const uint32_t face = sel(any_lane(ql[i] == fsx) & (g.nx > 1 ? ~0ull : 0ull), facex, sel(any_lane(ql[i] == fsy) & (g.ny > 1 ? ~0ull : 0ull), facey, facez));
            uint32_t bkey = 0;
            for (uint32_t c = 0; c < n; c += 32) {
                const uint32_t base = b0 + c;
                uint32_t skip = 0;
                if (entered) {
                    const uint32_t* __restrict__ fb = sc.face_bits + (face + (base >> 5));
                    skip = __builtin_amdgcn_alignbit(fb[1], fb[0], base & 31u);
                }
                const uint32_t left = n - c;
                uint32_t todo = ~skip & (left >= 32 ? 0xFFFFFFFFu : ((1u << left) - 1u)) & slice_mask;
                while (todo) {
                    const uint32_t k = (uint32_t)__builtin_ctz(todo);
                    todo &= todo - 1;
                    const uint32_t j = base + k;
                    if (INLINE) {
                        const float4* __restrict__ rec = sc.ref_rec + 3ull * j;
                        const float4 c0 = rec[0], c1 = rec[1];
                        const float e2z = reinterpret_cast<const float*>(rec + 2)[0];
                        ray_triangle({ c0.x, c0.y, c0.z }, { c0.w, c1.x, c1.y }, { c1.z, c1.w, e2z }, o, d, [&](float t) {
                            r.tri = __float_as_uint(reinterpret_cast<const float*>(rec + 2)[1]); r.t = t;
                            bkey = c + k + 1u;
                        });
                    } else {
                        const uint32_t ti = sc.indices[j];
                        const float4 r0 = sc.tri_rec[3ull * ti], r1 = sc.tri_rec[3ull * ti + 1];
                        const float e2z = reinterpret_cast<const float*>(sc.tri_rec + 3ull * ti + 2)[0];
                        ray_triangle({ r0.x, r0.y, r0.z }, { r0.w, r1.x, r1.y }, { r1.z, r1.w, e2z }, o, d, [&](float t) {
                            r.tri = ti; r.t = t;
                            bkey = c + k + 1u;
                        });
                    }
                }
            }
            // the slices' candidates: the highest entry that hit is the one the sequential loop would have kept
#pragma unroll
            for (int m = RAYS; m < 64; m <<= 1) {
                const uint32_t ok = (uint32_t)__shfl_xor((int)bkey, m), otri = (uint32_t)__shfl_xor((int)r.tri, m);
                const float ot = __shfl_xor(r.t, m);
                if (ok > bkey) { bkey = ok; r.tri = otri; r.t = ot; }
            }
            if (r.tri != HVRT_MISS) ended = true;      // :326-329
        }
        if (ended || !in_grid) break;
        // open space behind the batch's last voxel: dd - 1 voxels that hold nothing
        const int run = dd > 33u ? 32 : (int)dd - 1;
        for (int i = 0; i < run; ++i) { HVRT_STEP(); }
        if (run > 0) { tot += run; in_grid = HVRT_IN_GRID(); if (!in_grid) break; }
    }
#undef HVRT_IN_GRID
    return r;
}

// ---- wave-synchronous traversal on per-reference records (k_render, k_trace) ---------------------------
// The same walk as trace_ray<..., INLINE = true, ..., RAYS = 64>, written so that the wave's control flow is
// UNIFORM: the outer loop, the list loop and the chunk loop branch on ballots, and a ray that has ended (or a
// lane that never had one) stays in them with alive == false.  Every lane therefore still executes at the point
// where a list chunk is fetched, which is what the fast path needs:
//
//   In nine of ten test iterations of a coherent 8x8 tile all lanes test the SAME list entry (the rays of a tile
//   leave from one clamped voxel and stay together), and a vector load costs the CU's texture path the same
//   ~16 clocks per wave whether its 64 lanes name 64 addresses or one (scripts/ubench/l1_rate.hip): three loads
//   per entry and wave kept that path 97 % busy.  When all lanes that have work are in the same chunk of the same
//   list, the chunk's records (21 x 48 B = 63 lanes x 16 B, consecutive) are fetched by ONE coalesced 16-byte
//   load over the wave, and each entry's nine floats and id are then broadcast to scalar registers with
//   v_readlane and used as scalar operands of the test: one vector load per chunk instead of three per entry.
//   The entries are taken in ascending order over the union of the lanes' to-do bits (face-bit skips differ
//   between lanes that entered the voxel through different faces), so "the last entry that hits wins" holds.
//   Lanes in different voxels fall back to the per-lane loads for that round.
//
// Arithmetic, voxel sequence, test order and counters are trace_ray's (and so the reference's); only who fetches
// the bytes differs.  ABORT gives up for the whole wave at once (the clock is a scalar).
constexpr uint32_t COOP_CHUNK = 21;     // entries per cooperative chunk: 63 lanes x 16 B = 21 records of 48 B
#ifndef HVRT_SPEC_OFFSETS
#define HVRT_SPEC_OFFSETS 1      // a consult of the distance field fetches the voxel's list bounds in the same breath
#endif
#ifndef HVRT_CULL_DIRECT
#define HVRT_CULL_DIRECT 0
#endif
#ifndef HVRT_GROUP_SERIAL
#define HVRT_GROUP_SERIAL 0
#endif

// ---- tile-level rejection of list entries (k_render: all rays of a wave leave from one point) ---------------------
// Measured on config 2 (make diag DIAG=25 / 27): of the 102 wave-level ray / triangle tests per ray slot, 0.6 get past the
// first rejection in ANY lane, and 101 are rejected by every direction of the 8x8 tile on the same side of u with one
// sign of the determinant.  Such an entry need not be tested by 64 lanes one after the other: with a common origin o the
// reference's first rejection (GeometryUtils.cpp:64-78) depends on the ray only through two linear forms of the direction,
//      det = e1 . (d x e2) = d . (e2 x e1) =: d . N          s . (d x e2) = d . (e2 x s) =: d . M        (s = o - v0)
// and u = (d . M) / (d . N) is outside [0, 1] for every d of a convex set on which d . N keeps its sign as soon as
// d . (M + m N) or d . (M - (1 + m) N) keeps the right sign there.  The set is the bounding box of the 64 lanes' actual
// float directions (centre c, half extents h, worked out once per tile and sample: tile_box), over which a linear form
// d . X ranges within c . X +- h . |X|.  So ONE lane per record decides for the whole tile, 21 records at a time.
//
// It must PROVE the reference's rejection for the floats the reference computes, not for the real numbers:
//   * the reference's det and s . p differ from d . N and d . M (reals, d the lane's float direction) by at most
//     7 u |e1| |e2| |d| and 7 u |s| |e2| |d|, u = 2^-24 (each component of p = d x e2 is off by <= 2 u |d| |e2|, a
//     three-term dot adds <= 3 u |a| |b|);
//   * N, M, the two dots and the sums below are computed here in fp32, another <= 20 u of the same magnitudes, and the
//     box centre / half extents carry <= 2^-22 per component;
//   * all of it is covered by slacks of 2^-16 |e1|_1 |e2|_1 and 2^-16 |s|_1 |e2|_1 (1-norms: upper bounds of the
//     Euclidean ones) -- a thousand times the sum of the above -- plus 1e-30 against gradual underflow;
//   * u is rejected with a margin of 2^-11 on either side, far more than the two roundings between s . p / det and the
//     reference's u = inv_det * (s . p) and than its FLT_EPSILON tolerance: "(d . M) / (d . N) < -2^-11" implies
//     "u < 0 && |u| > EPS", "> 1 + 2^-11" implies "u > 1 && |u - 1| > EPS";
//   * every condition is a comparison that is false on NaN, so infinities and NaNs anywhere cull nothing; a tile with a
//     non-finite direction component gets an infinite box.
// An entry that is culled would have been rejected by every lane (hit / miss, t and the reference's counters are
// untouched: those count list entries, not executed tests); an entry that is not culled is tested as before.
constexpr float CULL_MARGIN = 4.8828125e-4f;     // 2^-11
constexpr float CULL_SLACK = 1.52587890625e-5f;  // 2^-16
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_of(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
// min and max over the 64 lanes (all active, no NaNs), in every lane
__device__ __forceinline__ void wave_minmax(float v, float& lo, float& hi) {
    float a = v, b = v;
    a = __builtin_fminf(a, dpp_f<0xB1>(a)); b = __builtin_fmaxf(b, dpp_f<0xB1>(b));      // quad_perm [1,0,3,2]
    a = __builtin_fminf(a, dpp_f<0x4E>(a)); b = __builtin_fmaxf(b, dpp_f<0x4E>(b));      // quad_perm [2,3,0,1]
    a = __builtin_fminf(a, dpp_f<0x141>(a)); b = __builtin_fmaxf(b, dpp_f<0x141>(b));    // row_half_mirror
    a = __builtin_fminf(a, dpp_f<0x140>(a)); b = __builtin_fmaxf(b, dpp_f<0x140>(b));    // row_mirror
    lo = __builtin_fminf(__builtin_fminf(lane_of(a, 0), lane_of(a, 16)), __builtin_fminf(lane_of(a, 32), lane_of(a, 48)));
    hi = __builtin_fmaxf(__builtin_fmaxf(lane_of(b, 0), lane_of(b, 16)), __builtin_fmaxf(lane_of(b, 32), lane_of(b, 48)));
}
// The bounding box of the wave's 64 ray directions -> box[0] = centre, box[1] = half extents (LDS, read back by
// cull_chunk).  Called by all 64 lanes, with or without a live ray (every lane has a direction).
__device__ __forceinline__ void tile_box(V3 d, float4* __restrict__ box) {
    const bool finite = any_lane(!(fabsf(d.x) <= FLT_MAX && fabsf(d.y) <= FLT_MAX && fabsf(d.z) <= FLT_MAX)) == 0ull;
    float lx, hx, ly, hy, lz, hz;
    wave_minmax(d.x, lx, hx); wave_minmax(d.y, ly, hy); wave_minmax(d.z, lz, hz);
    const float pad = finite ? 4.76837158203125e-7f : __builtin_inff();       // 2^-21 per component: the roundings of centre and half extent
    if (lane_now() == 0) {
        box[0] = { 0.5f * lx + 0.5f * hx, 0.5f * ly + 0.5f * hy, 0.5f * lz + 0.5f * hz, 0.0f };
        box[1] = { (0.5f * hx - 0.5f * lx) + pad, (0.5f * hy - 0.5f * ly) + pad, (0.5f * hz - 0.5f * lz) + pad, 0.0f };
    }
}
// Which of the chunk's records (win[3 k .. 3 k + 2] = v0, e1, e2 of record k, k < 21) every direction of the tile rejects:
// lane k decides for record k (s = o - v0 is the reference's own subtraction, GeometryUtils.cpp:73, done once per record
// here); returns the mask of culled records.
// (HVRT_CULL_DIRECT: lane k reads its record straight from the record array -- three loads a record apart, all inside the
// chunk's kilobyte -- instead of the wave fetching the chunk with one coalesced load, parking it in LDS and the lanes
// reading it back: the common case, nothing of the chunk left, then never touches LDS for the records)
__device__ __forceinline__ uint32_t cull_chunk(const float4* __restrict__ recs, const float4* __restrict__ win, int lane, V3 o) {
    const int k = lane < (int)COOP_CHUNK ? lane : (int)COOP_CHUNK - 1;
    const float4* __restrict__ rk = recs + 3 * k;
    const float4 q0 = rk[0], q1 = rk[1];
    const float e2z = reinterpret_cast<const float*>(rk + 2)[0];
    const float4 bc = win[64], bh = win[65];
    const V3 sv = sub(o, { q0.x, q0.y, q0.z }), e1 = { q0.w, q1.x, q1.y }, e2 = { q1.z, q1.w, e2z };
    const float l1e2 = (fabsf(e2.x) + fabsf(e2.y)) + fabsf(e2.z);
    const float slack_n = CULL_SLACK * (((fabsf(e1.x) + fabsf(e1.y)) + fabsf(e1.z)) * l1e2) + 1e-30f;
    const float slack_m = CULL_SLACK * (((fabsf(sv.x) + fabsf(sv.y)) + fabsf(sv.z)) * l1e2) + 1e-30f;
    const V3 N = cross(e2, e1);
    const float n0 = (bc.x * N.x + bc.y * N.y) + bc.z * N.z;
    const float rn = (bh.x * fabsf(N.x) + bh.y * fabsf(N.y)) + bh.z * fabsf(N.z);
    const V3 M = cross(e2, sv);
    const float m0 = (bc.x * M.x + bc.y * M.y) + bc.z * M.z;
    const float rm = (bh.x * fabsf(M.x) + bh.y * fabsf(M.y)) + bh.z * fabsf(M.z);
    const float an0 = fabsf(n0);
    const float sm0 = __uint_as_float(__float_as_uint(m0) ^ (__float_as_uint(n0) & 0x80000000u));      // m0 * sign(n0)
    const bool one_sign = an0 - rn > slack_n;                       // d . N keeps n0's sign, and so does the reference's det
    const float reach = an0 + rn;                                   // >= |d . N| anywhere in the box
    // u < -margin everywhere:  sign * d . (M + margin N) <= sm0 + margin an0 + rm + margin rn
    const bool below = (sm0 + rm) + CULL_MARGIN * reach < -(slack_m + slack_n);
    // u > 1 + margin everywhere:  sign * d . (M - (1 + margin) N) >= sm0 - rm - (1 + margin) (an0 + rn)
    const bool above = (sm0 - rm) - (1.0f + CULL_MARGIN) * reach > slack_m + 2.0f * slack_n;
    return (uint32_t)any_lane(one_sign && (below || above)) & ((1u << COOP_CHUNK) - 1u);
}

__device__ __forceinline__ uint32_t wave_or(uint32_t v) {
    // OR over the 64 lanes (all must be active): two quad swaps, the two row mirrors, then the four rows' values
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);   // row_half_mirror
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);   // row_mirror
    return (uint32_t)(__builtin_amdgcn_readlane((int)v, 0) | __builtin_amdgcn_readlane((int)v, 16) |
                      __builtin_amdgcn_readlane((int)v, 32) | __builtin_amdgcn_readlane((int)v, 48));
}
__device__ __forceinline__ float lane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// SAME_ORIGIN: every lane's ray starts at the same point (k_render: the camera), so the start voxel is a scalar
template <bool COUNT, bool NEAREST, bool ABORT, bool SAME_ORIGIN = false>
__device__ __forceinline__ TraceResult trace_ray_wave(const SceneDev& sc, V3 o, V3 d, bool alive, float4* __restrict__ win, long long deadline = 0) {
    const GridParams& g = sc.g;
    TraceResult r;
    r.tri = HVRT_MISS; r.t = FLT_MAX; r.nv = 0; r.nt = 0;
#if defined(HVRT_ABLATE) && HVRT_ABLATE >= 3      // 3 = no walk: what a tile costs before and after its rays (4: and no pixel stores, 5: and no duration store, 6: and no ray set-up)
    alive = false;
#endif

    V3 st = o;
    // (a scalar start point is taken through an opaque copy, so that the start voxel is worked out here, per ray and
    // into scalar registers, and not hoisted out of the caller's loops into vector registers that must be spilled)
    if (SAME_ORIGIN && !NEAREST) asm volatile("" : "+v"(st.x), "+v"(st.y), "+v"(st.z));
    if (NEAREST) {
        const float ivx = 1.0f / d.x, ivy = 1.0f / d.y, ivz = 1.0f / d.z;
        const float tx1 = (g.mnx - o.x) * ivx, tx2 = (g.bxx - o.x) * ivx;
        const float ty1 = (g.mny - o.y) * ivy, ty2 = (g.bxy - o.y) * ivy;
        const float tz1 = (g.mnz - o.z) * ivz, tz2 = (g.bxz - o.z) * ivz;
        const float tn = max2(max2(min2(tx1, tx2), min2(ty1, ty2)), min2(tz1, tz2));
        const float tf = min2(min2(max2(tx1, tx2), max2(ty1, ty2)), max2(tz1, tz2));
        alive = alive && (tf >= tn && tf >= 0.0f);
        const float t0 = max2(tn, 0.0f);
        st = { o.x + d.x * t0, o.y + d.y * t0, o.z + d.z * t0 };
    }
    int ix = clampi(f2i((st.x - g.mnx) / g.vsx), 0, g.nx - 1);
    int iy = clampi(f2i((st.y - g.mny) / g.vsy), 0, g.ny - 1);
    int iz = clampi(f2i((st.z - g.mnz) / g.vsz), 0, g.nz - 1);
    if (SAME_ORIGIN && !NEAREST) {      // (the nearest-hit mode starts where each ray enters the grid)
        ix = __builtin_amdgcn_readfirstlane(ix); iy = __builtin_amdgcn_readfirstlane(iy); iz = __builtin_amdgcn_readfirstlane(iz);
    }
    const bool px = d.x > 0.0f, py = d.y > 0.0f, pz = d.z > 0.0f;
    float tdx = fabsf(g.vsx / d.x), tdy = fabsf(g.vsy / d.y), tdz = fabsf(g.vsz / d.z);
    // (the absolute values are taken here, once: left to the compiler they become |x| source modifiers on every
    // `tMax + tDelta` of the walk, which forces those adds into the VOP3 encoding -- 4 clocks of issue instead of 2)
    asm volatile("" : "+v"(tdx), "+v"(tdy), "+v"(tdz));
    float tmx = ((g.mnx + (float)(px ? ix + 1 : ix) * g.vsx) - o.x) / d.x;
    float tmy = ((g.mny + (float)(py ? iy + 1 : iy) * g.vsy) - o.y) / d.y;
    float tmz = ((g.mnz + (float)(pz ? iz + 1 : iz) * g.vsz) - o.z) / d.z;
    int rx = px ? g.nx - 1 - ix : ix;
    int ry = py ? g.ny - 1 - iy : iy;
    const int zsum = (pz ? g.nz - 1 - iz : iz) + rx + ry;
    int tot = 0;
    const uint32_t plane = (uint32_t)g.nx * (uint32_t)g.ny;
    const uint32_t fsx = px ? 1u : 0xFFFFFFFFu;
    const uint32_t fsy = py ? (uint32_t)g.nx : 0u - (uint32_t)g.nx;
    const uint32_t fsz = pz ? plane : 0u - plane;
    uint32_t f = (uint32_t)ix + (uint32_t)g.nx * (uint32_t)iy + plane * (uint32_t)iz;
    const uint32_t* __restrict__ offs = sc.offsets;
    const uint32_t fw = (uint32_t)sc.face_words;
    uint32_t last = 0;
#ifndef HVRT_DIAG
    if (COUNT) r.nv = alive ? 1u : 0u;
#endif

#define HVRT_IN_GRID() ((rx | ry | (zsum - tot - rx - ry)) >= 0)
#define HVRT_NEAREST_DONE()                                                     \
    (r.tri != HVRT_MISS && r.t <= (((tmx < tmy ? tmx : tmy) < tmz) ? (tmx < tmy ? tmx : tmy) : tmz))

    const uint8_t* __restrict__ dist = sc.dist;
    for (;;) {
        // phase 1 (as in trace_ray): walk to the next occupied voxel, k empty-space steps at a time
        bool search = alive;
        bool done = false;
#if HVRT_SPEC_OFFSETS
        uint32_t sb0 = 0, sb1 = 0;       // the list bounds of the voxel a lane last looked at, fetched WITH its distance byte
#endif
        for (;;) {
            uint32_t dd = 255u;
#if HVRT_SPEC_OFFSETS
            if (search) { dd = dist[f]; const uint32_t* pv = offs + f; sb0 = pv[0]; sb1 = pv[1]; }
#else
            if (search) dd = dist[f];
#endif
            search = search && dd != 0u;
            if (any_lane(search) == 0ull) break;
            const uint32_t ds = search ? dd : 255u;
            int k = 1;
            if (any_lane(ds < 2u) == 0ull) {
                k = 2;
                if (any_lane(ds < 4u) == 0ull) {
                    k = 4;
                    if (any_lane(ds < 8u) == 0ull) {
                        k = 8;
                        if (any_lane(ds < 16u) == 0ull) k = any_lane(ds < 32u) == 0ull ? 32 : 16;
                    }
                }
            }
            if (search) {
                if ((COUNT && !HVRT_DIAG_BUILD) || NEAREST) {      // (a diagnostic build counts other things)
                    bool going = true;
                    for (int i = 0; i < k; ++i) {
                        if (NEAREST) { done = done || HVRT_NEAREST_DONE(); going = going && !done; }
                        HVRT_STEP();
                        ++tot;
                        going = going && HVRT_IN_GRID();
#ifndef HVRT_DIAG
                        if (COUNT) r.nv += going ? 1u : 0u;
#endif
                    }
                    search = going;
                } else {
                    bool ran = false;
                    if (k >= 4) {
                        // which axis, if any, stays out of this run for every lane?  (estimates; the run itself checks)
                        const float kf = (float)k;
                        const float ex = tmx + kf * tdx, ey = tmy + kf * tdy;
                        if (any_lane(!(ex < tmz && ey < tmz)) == 0ull)
                            HVRT_TWO_AXIS_RUN(tmx, tdx, rx, fsx, tmy, tdy, fsy, (c ? preA : preB) < tmz, ry -= (int)dB);
                        else if (any_lane(!(ex < tmy)) == 0ull)
                            HVRT_TWO_AXIS_RUN(tmx, tdx, rx, fsx, tmz, tdz, fsz, preA < tmy, (void)0);
                        else if (any_lane(tmx < ey) == 0ull)
                            HVRT_TWO_AXIS_RUN(tmy, tdy, ry, fsy, tmz, tdz, fsz, !(tmx < preA), (void)0);
                    }
#if defined(HVRT_DIAG) && HVRT_DIAG == 23   // general steps of runs with k >= 4: every lane has SOME far axis by the estimate | not
                    // (config 2: 0.16 | 29.3 per ray -- the runs that fall back hold a lane in which all three axes step, so
                    // a far axis chosen per lane would convert next to nothing; runs cut into quarters were tried: slower)
                    if (!ran && k >= 4) {
                        const float kf2 = (float)k;
                        const float ex2 = tmx + kf2 * tdx, ey2 = tmy + kf2 * tdy;
                        const bool some = (ex2 < tmz && ey2 < tmz) || (ex2 < tmy) || !(tmx < ey2);
                        if (any_lane(!some) == 0ull) r.nv += (uint32_t)k; else r.nt += (uint32_t)k;
                    }
#elif defined(HVRT_DIAG) && HVRT_DIAG == 21   // steps of two-axis runs | general steps of runs with k >= 4
                    if (ran) r.nv += (uint32_t)k; else if (k >= 4) r.nt += (uint32_t)k;
#elif defined(HVRT_DIAG) && HVRT_DIAG == 22   // general steps of runs with k < 4 | runs (consults that stepped)
                    if (k < 4) r.nv += (uint32_t)k;
                    r.nt += 1;
#endif
                    if (!ran) for (int i = 0; i < k; ++i) HVRT_STEP();
                    tot += k;
                    search = HVRT_IN_GRID();
                }
            }
        }
        if (NEAREST) alive = alive && !done;
        alive = alive && HVRT_IN_GRID();
        if (any_lane(alive) == 0ull) break;
        // (the give-up variant looks at the clock at every occupied voxel it reaches: a tile of grazing rays is long by the
        // number of its voxels, whatever the length of their lists)
        if (ABORT && wall_clock64() > (unsigned long long)deadline) { r.tri = HVRT_ABORTED; return r; }
        uint32_t b0 = 0, b1 = 0;
#if !defined(HVRT_ABLATE) || HVRT_ABLATE < 2      // ablation builds (timing experiments only): 2 = no lists at all
#if HVRT_SPEC_OFFSETS
        if (alive) { b0 = sb0; b1 = sb1; }
#else
        if (alive) { const uint32_t* pv = offs + f; b0 = pv[0]; b1 = pv[1]; }
#endif
#endif
        const bool entered = last != 0u;
        // phase 2: testVoxelTriangles, :267-295 -- every entry is tested, the LAST one that hits wins
        const uint32_t n = b1 - b0;       // 0 for lanes without a ray
        uint32_t c = 0;
        // Which entries of the chunk at `base` this lane has to test: all `left` of them (at most COOP_CHUNK) but those
        // that are also listed in the voxel the ray came from -- it tested them there, one step ago, and missed.
        // The face through which the voxel was entered (word offset of its bit array: 0 -x, 1 +x, 2 -y, 3 +y, 4 -z,
        // 5 +z neighbour) comes from the last step's increment -- the walking direction is in the increments' signs, so
        // nothing but `last` is kept for it.  A surviving x step implies nx > 1, so the axes are told apart in this
        // order with their extent (in a 1-wide grid fsx == fsy or fsy == fsz).
        // (Worked out only where a test may really run: the tile-level rejection below leaves nothing of most chunks.)
        auto todo_of = [&](uint32_t base, bool work) -> uint32_t {
            uint32_t todo = 0;
            if (work) {
                uint32_t skip = 0;
                if (entered) {
                    const uint32_t face = sel(any_lane(last == fsx) & (g.nx > 1 ? ~0ull : 0ull), sel(any_lane(fsx == 1u), 0u, fw),
                                              sel(any_lane(last == fsy) & (g.ny > 1 ? ~0ull : 0ull), sel(any_lane(fsy == (uint32_t)g.nx), 2u * fw, 3u * fw),
                                                  sel(any_lane(fsz == plane), 4u * fw, 5u * fw)));
                    const uint32_t* __restrict__ fb = sc.face_bits + (face + (base >> 5));
                    skip = __builtin_amdgcn_alignbit(fb[1], fb[0], base & 31u);
                }
                const uint32_t left = n - c;
                todo = ~skip & (left >= COOP_CHUNK ? ((1u << COOP_CHUNK) - 1u) : ((1u << left) - 1u));
            }
            return todo;
        };
        for (;;) {
            const bool work = c < n;
            const unsigned long long wm = any_lane(work);
            if (wm == 0ull) break;
            if (ABORT) {
                // the wave gives the tile up as a whole: nothing of it is kept
                if (any_lane(work && c != 0u) != 0ull && wall_clock64() > (unsigned long long)deadline) { r.tri = HVRT_ABORTED; return r; }
            }
            const uint32_t base = b0 + c;
            const int first = __builtin_ctzll(wm);
            const uint32_t ubase = (uint32_t)__builtin_amdgcn_readlane((int)base, first);
            // SAME_ORIGIN: the lanes that stand in the first working lane's voxel are served together; those in other voxels
            // wait for a later round of this loop (HVRT_GROUP_SERIAL)
            const bool mine = work && base == ubase;
            // (HVRT_GROUP_SERIAL: 1 = always, 2 = when that voxel holds at least half of the working lanes)
            const unsigned long long gm = any_lane(mine);
            const bool serve_group = SAME_ORIGIN && (HVRT_GROUP_SERIAL == 1 || (HVRT_GROUP_SERIAL == 2 && 2 * __builtin_popcountll(gm) >= __builtin_popcountll(wm)));
            if (serve_group || gm == wm) {
                // every lane with work is in this chunk of this list (same voxel, so same list length): one coalesced
                // fetch for the wave, parked in the wave's LDS window; every lane then tests the entries that ANY lane
                // has to test (an entry one lane skips was tested by that ray one voxel ago and missed, so testing it
                // again changes nothing), reading each record from LDS at a wave-uniform address (a broadcast)
                const int lane = lane_now();
                uint32_t um;
                if (SAME_ORIGIN) {
                    // entries that every direction of the tile rejects are not tested at all (see cull_chunk); what the lanes
                    // may skip is looked up only for a chunk of which something is left
                    const uint32_t uleft = (uint32_t)__builtin_amdgcn_readlane((int)(n - c), first);
                    um = uleft >= COOP_CHUNK ? ((1u << COOP_CHUNK) - 1u) : ((1u << uleft) - 1u);
#if HVRT_CULL_DIRECT
                    const uint32_t culled = cull_chunk(sc.ref_rec + 3ull * ubase, win, lane, o);
#else
                    win[lane] = sc.ref_rec[3ull * ubase + (uint32_t)lane];
                    const uint32_t culled = cull_chunk(win, win, lane, o);
#endif
#if defined(HVRT_DIAG) && HVRT_DIAG == 28   // entries of cooperative chunks culled | left (before the lanes' skip bits)
                    if (COUNT && lane == 0) { r.nv += 64 * (uint32_t)__builtin_popcount(um & culled); r.nt += 64 * (uint32_t)__builtin_popcount(um & ~culled); }
#endif
                    um &= ~culled;
                    if (um) {
#if HVRT_CULL_DIRECT
                        win[lane] = sc.ref_rec[3ull * ubase + (uint32_t)lane];      // something is left to test: the chunk goes to LDS after all
#endif
                        um &= wave_or(todo_of(base, mine));
                    }
                } else {
                    um = wave_or(todo_of(base, mine));
                    win[lane] = sc.ref_rec[3ull * ubase + (uint32_t)lane];
                }
#ifdef HVRT_DIAG   // diagnostic build only: what the two counters mean depends on HVRT_DIAG (scripts/diag_counts.py)
                if (COUNT && lane == 0 && HVRT_DIAG == 11) r.nv += 64;                                    // cooperative chunks
                if (COUNT && lane == 0 && HVRT_DIAG == 12) r.nv += 64 * (uint32_t)__builtin_popcount(um);   // cooperative test iterations
#endif
#if defined(HVRT_ABLATE) && HVRT_ABLATE == 1      // 1 = lists fetched, no entry tested
                um = 0;
#endif
                if (mine) {
                    while (um) {
                        const int k = __builtin_ctz(um);
                        um &= um - 1u;
                        const float4* __restrict__ rec = win + 3 * k;
                        const float4 c0 = rec[0], c1 = rec[1];
                        const float2 c2 = *reinterpret_cast<const float2*>(rec + 2);
                        auto keep = [&](float t) { if (!NEAREST || t < r.t) { r.tri = __float_as_uint(c2.y); r.t = t; } };
#if defined(HVRT_DIAG) && (HVRT_DIAG == 25 || HVRT_DIAG == 26)
                        uint32_t dcs[2] = { 0, 0 };
                        if (SAME_ORIGIN) { ray_triangle_s(sub(o, { c0.x, c0.y, c0.z }), { c0.w, c1.x, c1.y }, { c1.z, c1.w, c2.x }, d, keep, COUNT ? dcs : nullptr); r.nv += dcs[0]; r.nt += dcs[1]; } else
#endif
                        ray_triangle({ c0.x, c0.y, c0.z }, { c0.w, c1.x, c1.y }, { c1.z, c1.w, c2.x }, o, d, keep);
                    }
                }
            } else {
#ifdef HVRT_DIAG
                if (COUNT && HVRT_DIAG == 11 && lane_now() == 0) r.nt += 64;                  // per-lane (fallback) chunks
#endif
                uint32_t todo = todo_of(base, work);
                while (todo) {
                    const uint32_t k = (uint32_t)__builtin_ctz(todo);
                    todo &= todo - 1;
#ifdef HVRT_DIAG
                    if (COUNT && HVRT_DIAG == 12 && (int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) r.nt += 64;   // fallback wave iterations
#endif
                    const float4* __restrict__ rec = sc.ref_rec + 3ull * (base + k);
                    const float4 c0 = rec[0], c1 = rec[1];
                    const float e2z = reinterpret_cast<const float*>(rec + 2)[0];
                    ray_triangle({ c0.x, c0.y, c0.z }, { c0.w, c1.x, c1.y }, { c1.z, c1.w, e2z }, o, d, [&](float t) {
                        if (!NEAREST || t < r.t) { r.tri = __float_as_uint(reinterpret_cast<const float*>(rec + 2)[1]); r.t = t; }
                    });
                }
            }
            if ((serve_group || gm == wm) ? mine : work) c += COOP_CHUNK;
        }
#ifndef HVRT_DIAG
        if (COUNT) r.nt += n;
#endif
        if (NEAREST) { if (alive && HVRT_NEAREST_DONE()) alive = false; }
        else if (r.tri != HVRT_MISS) alive = false;      // :326-329
        if (alive) {
            HVRT_STEP();
            ++tot;
            alive = HVRT_IN_GRID();
#ifndef HVRT_DIAG
            if (COUNT) r.nv += alive ? 1u : 0u;
#endif
        }
        if (any_lane(alive) == 0ull) break;
    }
#undef HVRT_IN_GRID
#undef HVRT_NEAREST_DONE
    return r;
}

// computeShading (OctreeLikeRayTracer.cpp:27-40) + LambertianShader (Shaders.cpp:4-22): with
// r=g=b=kd=1 and inShadow=0 each channel is exactly 0.1f + max(0, N.L).
__device__ __forceinline__ V3 shade(const SceneDev& sc, const TraceResult& r, V3 o, float v) {
    if (r.tri != HVRT_MISS) {
        const float4 r0 = sc.tri_rec[3ull * r.tri], r2 = sc.tri_rec[3ull * r.tri + 2];
        V3 L = normalize(sub(o, { r0.x, r0.y, r0.z }));   // hit.point = triangle.v0 (:286)
        V3 N = normalize({ r2.y, r2.z, r2.w });
        float ndl = dot(N, L);
        float k = (0.0f < ndl) ? ndl : 0.0f;
        float c = 0.1f + k;
        return { c, c, c };
    }
    return { 0.5f + 0.3f * v, 0.5f + 0.3f * v, 0.7f + 0.3f * v };
}

__device__ __forceinline__ uint32_t mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x7feb352dU; h ^= h >> 15; h *= 0x846ca68bU; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ void jitter(uint32_t x, uint32_t y, uint32_t s, uint32_t seed, float& jx, float& jy) {
    uint32_t h = mix32(x + 0x9E3779B9U * (y + 0x85EBCA6BU * (s + 0xC2B2AE35U * seed)));
    uint32_t a = mix32(h ^ 0x68E31DA4U), b = mix32(h ^ 0xB5297A4DU);
    jx = (float)(a >> 8) * (1.0f / 16777216.0f);
    jy = (float)(b >> 8) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ uint64_t wave_sum64(uint64_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d);
    return v;
}

// What both render kernels do for one pixel: the spp samples of pixel (x, y) of frame `cf` -- renderPixel
// (OctreeLikeRayTracer.cpp:73-74; pixel corner, jitter is 0 at spp 1), generateRay (:361-364), the traversal and
// the shading.  Returns false when ABORT is set and a walk gave up (nothing of the pixel is valid then).
// Pixel of lane `lane` in tile (tx, ty) of a part: column x, row lr inside the part's slab, row y of the frame (tile k
// of the image's row tiles belongs to part k % n_parts).  Worked out wherever it is needed -- before the walk, per
// sample for the jitter, after the walk for the stores -- from the two scalars tx, ty instead of being carried through
// the walk in vector registers: k_render has no register to spare at seven waves per SIMD (the asm statements keep
// the compiler from hoisting the lane-only parts out of the tile loop, where they would have to be spilled).
__device__ __forceinline__ void pixel_coords(const RenderArgs& ra, uint32_t tx, uint32_t ty, int lane, int& x, int& lr, int& y) {
    if (lane < 0) lane = lane_now();      // k_render: not even the lane number is carried (it would be spilled)
    else asm volatile("" : "+v"(lane));
    x = (int)(tx * 8u) + (lane & 7);
    lr = (int)(ty * 8u) + (lane >> 3);
    if (ra.n_parts == 1) {
        y = lr;
    } else {
        int tr = ra.tile_rows;
        asm volatile("" : "+s"(tr));
        const int kl = lr / tr;
        y = (kl * ra.n_parts + ra.part) * tr + (lr - kl * tr);
    }
}

struct PixelResult {
    V3 rgb;              // mean of the samples
    uint32_t hit; float t;   // sample 0's
    uint64_t nv, nt;     // counters over all samples
};
template <bool COUNT, bool INLINE, bool NEAREST, int RAYS, bool ABORT>
__device__ __forceinline__ bool render_pixel(const SceneDev& sc, const RenderArgs& ra, const hvrt_camera_frame* __restrict__ cf,
                                             uint32_t tx, uint32_t ty, int lane, int px_x, int px_y, bool alive,
                                             float4* __restrict__ win, long long wall0, PixelResult& out, int only = -1) {
    // only >= 0: that sample alone, its colour as it is (out.hit / out.t are sample 0's or a miss)
    // The frame is wave-uniform and nothing writes it while the kernel runs, but the compiler cannot see the second
    // part (the tile loop stores pixels) and would fetch it with vector loads into twelve vector registers that stay
    // live across the walk -- four of them in scratch at 64 registers.  Read through the constant address space it
    // arrives by scalar loads in scalar registers; and it is read again for every sample (the opaque copy of the
    // pointer below), because nine scalar registers held across the walk are spilled into lanes of a vector register
    // and come back through v_readlane inside the walk's loops.
    typedef const __attribute__((address_space(4))) float* const_floats;
    static_assert(sizeof(hvrt_camera_frame) == 48 && offsetof(hvrt_camera_frame, lower_left) == 12 &&
                  offsetof(hvrt_camera_frame, horizontal) == 24 && offsetof(hvrt_camera_frame, vertical) == 36, "twelve consecutive floats");
    const uint64_t cf_bits = (uint64_t)(uintptr_t)cf;       // the same for every lane: say so
    const uint64_t cf_uniform = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)cf_bits) |
                                ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(cf_bits >> 32)) << 32);
    // (ra.frames == nullptr: the launch's single frame is RenderArgs::frame0, i.e. it lies in the kernel-argument segment --
    // every kernel that gets here has the parameter list (SceneDev, RenderArgs) -- and is read from there the same way)
    typedef const __attribute__((address_space(4))) char* const_bytes;
    constexpr size_t RA_IN_KERNARG = (sizeof(SceneDev) + alignof(RenderArgs) - 1) / alignof(RenderArgs) * alignof(RenderArgs);
    const_floats cfs = ra.frames ? (const_floats)cf_uniform
                          : (const_floats)((const_bytes)__builtin_amdgcn_kernarg_segment_ptr() + RA_IN_KERNARG + offsetof(RenderArgs, frame0));
    V3 org = { cfs[0], cfs[1], cfs[2] };
    // (the origin is an operand of every ray / triangle test: in vector registers, where the walk can use it without a
    // second scalar operand in one instruction)
    asm volatile("" : "+v"(org.x), "+v"(org.y), "+v"(org.z));
    V3 acc = { 0.0f, 0.0f, 0.0f };
    out.hit = HVRT_MISS; out.t = FLT_MAX; out.nv = 0; out.nt = 0;
    for (int s = only < 0 ? 0 : only; s < (only < 0 ? ra.spp : only + 1); ++s) {
        // RAYS == 64 (k_render): the pixel's coordinates are worked out again per sample (see pixel_coords);
        // k_render_split's lanes share pixels and pass theirs in
        int x = px_x, y = px_y, lr_unused = 0;
        if (RAYS == 64) pixel_coords(ra, tx, ty, lane, x, lr_unused, y);
        float jx = 0.0f, jy = 0.0f;
        if (ra.spp > 1) jitter((uint32_t)x, (uint32_t)y, (uint32_t)s, ra.seed, jx, jy);
        const float u = ((float)x + jx) / ra.width_f;
        const float v = ((float)y + jy) / ra.height_f;
        asm volatile("" : "+s"(cfs));
        const V3 ll = { cfs[3], cfs[4], cfs[5] }, hor = { cfs[6], cfs[7], cfs[8] }, ver = { cfs[9], cfs[10], cfs[11] };
        const V3 d = normalize(sub(add(add(ll, mul(hor, u)), mul(ver, v)), org));
        TraceResult r;
        // the wave-synchronous walk wants every lane of the wave in the call, with or without a ray (alive);
        // the per-lane walks are only ever called by lanes that have one
        if constexpr (INLINE && RAYS == 64) {
            tile_box(d, win + 64);
            r = trace_ray_wave<COUNT, NEAREST, ABORT, true>(sc, org, d, alive, win, wall0 + ra.abort_ticks * (s + 1));
        }
        else if constexpr (!COUNT && !NEAREST && !ABORT && RAYS < 64 && HVRT_LOOKAHEAD > 0) r = trace_ray_ahead<INLINE, RAYS>(sc, org, d);
        else r = trace_ray<COUNT, INLINE, NEAREST, RAYS, ABORT>(sc, org, d, wall0 + ra.abort_ticks * (s + 1));
        if (ABORT && r.tri == HVRT_ABORTED) return false;
        const V3 c = shade(sc, r, org, v);
        if (ra.spp == 1 || only >= 0) acc = c; else acc = add(acc, c);
        if (s == 0) { out.hit = r.tri; out.t = r.t; }
        if (COUNT) { out.nv += r.nv; out.nt += r.nt; }
    }
    if (ra.spp > 1 && only < 0) { const float n = ra.spp_f; acc = { acc.x / n, acc.y / n, acc.z / n }; }
    out.rgb = acc;
    return true;
}
// render_pixel for a pixel none of whose samples can hit anything (RenderArgs::cover): what it leaves for samples that all
// miss -- the background of every sample's own v, accumulated and averaged in the same order -- without rays or walks
__device__ __forceinline__ V3 background_colour(const RenderArgs& ra, int x, int y) {
    V3 acc = { 0.0f, 0.0f, 0.0f };
    for (int s = 0; s < ra.spp; ++s) {
        float jx = 0.0f, jy = 0.0f;
        if (ra.spp > 1) jitter((uint32_t)x, (uint32_t)y, (uint32_t)s, ra.seed, jx, jy);
        const float v = ((float)y + jy) / ra.height_f;
        const V3 c = { 0.5f + 0.3f * v, 0.5f + 0.3f * v, 0.7f + 0.3f * v };      // shade() on a miss
        if (ra.spp == 1) acc = c; else acc = add(acc, c);
    }
    if (ra.spp > 1) { const float n = ra.spp_f; acc = { acc.x / n, acc.y / n, acc.z / n }; }
    return acc;
}
__device__ __forceinline__ void background_pixel(const RenderArgs& ra, uint32_t tx, uint32_t ty, PixelResult& out) {
    out.hit = HVRT_MISS; out.t = FLT_MAX; out.nv = 0; out.nt = 0;
    int x, lr_unused, y;
    pixel_coords(ra, tx, ty, -1, x, lr_unused, y);
    out.rgb = background_colour(ra, x, y);
}
__device__ __forceinline__ void store_pixel(const RenderArgs& ra, uint32_t frame, int local_row, int x, const PixelResult& px) {
    const size_t p = ((size_t)frame * ra.slab_rows + local_row) * ra.width + x;
    if (ra.rgb) { ra.rgb[3 * p] = px.rgb.x; ra.rgb[3 * p + 1] = px.rgb.y; ra.rgb[3 * p + 2] = px.rgb.z; }
    if (ra.hit_id) ra.hit_id[p] = px.hit;
    if (ra.t) ra.t[p] = px.t;
}

// Frame row of row lr of this part's slab (pixel_coords' rule, for any lane or scalar)
__device__ __forceinline__ int frame_row(const RenderArgs& ra, int lr) {
    if (ra.n_parts == 1) return lr;
    const int kl = lr / ra.tile_rows;
    return (kl * ra.n_parts + ra.part) * ra.tile_rows + (lr - kl * ra.tile_rows);
}
// Does the tile cover mark tile (tx, ty) of frame fi of this part -- any of the frame's tiles its rows lie in?  (With row
// tiles of a multiple of eight rows, or a single part, the tile's eight rows lie in one tile row of the frame.)
__device__ __forceinline__ bool tile_covered(const RenderArgs& ra, uint32_t fi, uint32_t tx, uint32_t ty) {
    const uint32_t* __restrict__ words = ra.cover + (uint64_t)fi * ra.cover_frame_words + (tx >> 5);
    if (ra.n_parts == 1 || (ra.tile_rows & 7) == 0) {
        const int y = frame_row(ra, (int)(ty * 8u));
        return (words[(uint64_t)((uint32_t)y >> 3) * ra.cover_words_x] >> (tx & 31u)) & 1u;
    }
    bool any = false;
    for (int r = 0; r < 8; ++r) {
        const int lr = (int)(ty * 8u) + r;
        if (lr >= ra.local_rows) break;
        const int y = frame_row(ra, lr);
        if (y < ra.height) any = any || ((words[(uint64_t)((uint32_t)y >> 3) * ra.cover_words_x] >> (tx & 31u)) & 1u);
    }
    return any;
}

// The pixels of the tiles the cover does not mark: the reference's result for a miss.  A wave takes eight tiles side by side
// and writes them row by row -- 64 neighbouring pixels per store, where a wave per tile wrote eight pieces of 96 bytes (config
// 2: 146 us beside the traversal, and 32 400 workgroups in the way of everything else that wanted a wave slot) -- and the
// grid strides over the frame.
__global__ __launch_bounds__(256) void k_fill_background(RenderArgs ra) {
    const uint32_t tiles_x = (uint32_t)(ra.width + 7) >> 3;
    const uint32_t tiles_y = (uint32_t)(ra.local_rows + 7) >> 3;
    const uint32_t groups_x = (tiles_x + 7u) >> 3;
    const uint32_t units_frame = groups_x * tiles_y;
    const uint32_t n_units = units_frame * (uint32_t)ra.n_frames;
    const int lane = (int)(threadIdx.x & 63u);
    for (uint32_t unit = blockIdx.x * 4u + (threadIdx.x >> 6); unit < n_units; unit += gridDim.x * 4u) {
        const uint32_t fi = unit / units_frame, uf = unit - fi * units_frame;
        const uint32_t ty = uf / groups_x, gx = uf - ty * groups_x;
        const int x = (int)(gx * 64u) + lane;
        if (x >= ra.width) continue;
        if (tile_covered(ra, fi, (uint32_t)x >> 3, ty)) continue;
        for (int r = 0; r < 8; ++r) {
            const int lr = (int)(ty * 8u) + r;
            if (lr >= ra.local_rows) break;
            const int y = frame_row(ra, lr);
            if (y >= ra.height) continue;
            PixelResult px;
            px.hit = HVRT_MISS; px.t = FLT_MAX; px.nv = 0; px.nt = 0;
            px.rgb = background_colour(ra, x, y);
            store_pixel(ra, fi, lr, x, px);
        }
    }
}

// Natural order of a launch without history: rows of tiles from the middle of the frame outwards.  The reference's cameras
// look at the centre of the grid (TraceImages.cpp:125-197), so the silhouette -- where the long tiles are -- surrounds the
// image centre and starts early.
__device__ __forceinline__ uint32_t natural_tile(uint32_t sl, uint32_t tiles_x, uint32_t tiles_y) {
    const uint32_t tiles_frame = tiles_x * tiles_y;
    const uint32_t fi = sl / tiles_frame, tf = sl - fi * tiles_frame;
    const uint32_t q = tf / tiles_x, tx = tf - q * tiles_x;
    const uint32_t mid = tiles_y / 2, h = (q + 1) / 2;
    const uint32_t ty = (q & 1u) ? (mid >= h ? mid - h : tiles_y - 1 - (h - mid - 1))
                                 : (mid + h < tiles_y ? mid + h : (tiles_y - 1 - mid) + mid - h);
    return fi * tiles_frame + ty * tiles_x + tx;
}

// With a tile cover: k_render's queue is the marked tiles only, in the order they have in order_in (null: the natural order)
// -- exactly: the long tiles at the head of order_in stay at the head, so that they start with the launch.  (One atomic per
// wave for a place in the queue was simpler and let any wave's tiles land anywhere: the longest tiles of config 2 started
// 170 us into a 450 us launch.)  Two kernels without a wait: k_compact_count leaves every workgroup's number of marked
// tiles, k_compact_queue adds up the numbers before its own and places.  counters[0] <- the queue's length, counters[1] <-
// how many tiles make up the queue's head (the waves pull those one at a time); the tiles that are not marked get a duration
// of zero.
#ifndef HVRT_RIM_TILES
#define HVRT_RIM_TILES 3
#endif
__device__ __forceinline__ bool compact_slot(const RenderArgs& ra, const uint32_t* __restrict__ order_in, uint32_t slot, uint32_t n_in, uint32_t nf,
                                             uint32_t tiles_x, uint32_t tiles_y, uint32_t& tile, bool& head, bool clear_cost) {
    const uint32_t tiles_frame = tiles_x * tiles_y;
    const uint32_t n_tiles = tiles_frame * (uint32_t)ra.n_frames;
    tile = 0u;
    if (slot >= n_in) return false;
    tile = order_in ? order_in[slot] : natural_tile(slot, tiles_x, tiles_y);
    if (tile >= n_tiles) return false;
    const uint32_t fi = tile / tiles_frame, tf = tile - fi * tiles_frame;
    const uint32_t ty = tf / tiles_x, tx = tf - ty * tiles_x;
    const bool walk = tile_covered(ra, fi, tx, ty);
    if (clear_cost && !walk && !ra.debug_start) ra.tile_cost[tile] = 0u;
    // Which marked tiles go to the queue's head.  With an order: those behind its front (no duration known).  Without: the
    // rim of the cover -- a tile not marked within RIM tiles of this one.  Rays that graze the mesh cross the most occupied
    // voxels, so a launch without history finds its longest tiles a little inside the outline of what the cover marks,
    // and they had better start with the launch.
    head = false;
    if (walk) {
        if (order_in) head = slot >= nf;
        else {
            constexpr int RIM = HVRT_RIM_TILES;
            for (int dy = -RIM; dy <= RIM && !head; ++dy) {
                const int y = (int)ty + dy;
                if (y < 0 || y >= (int)tiles_y) continue;
                for (int dx = -RIM; dx <= RIM; dx += RIM) {      // (the square's corners and edge midpoints: enough for an outline that is a curve)
                    const int x = (int)tx + dx;
                    if (x < 0 || x >= (int)tiles_x) continue;
                    if (!tile_covered(ra, fi, (uint32_t)x, (uint32_t)y)) { head = true; break; }
                }
            }
        }
    }
    return walk;
}
// (block_marked: two arrays of one word per workgroup -- its marked tiles, and those of them behind order_in's first
// *n_front_in slots)
__global__ __launch_bounds__(256) void k_compact_count(RenderArgs ra, const uint32_t* __restrict__ order_in, const uint32_t* __restrict__ n_front_in,
                                                      uint32_t* __restrict__ block_marked) {
    const uint32_t tiles_x = (uint32_t)(ra.width + 7) >> 3;
    const uint32_t tiles_y = (uint32_t)(ra.local_rows + 7) >> 3;
    const uint32_t n_tiles = tiles_x * tiles_y * (uint32_t)ra.n_frames;
    const uint32_t n_in = order_in ? n_tiles - *ra.n_split : n_tiles;
    const uint32_t nf = (order_in && n_front_in) ? *n_front_in : 0u;
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    uint32_t tile;
    bool head;
    const bool walk = compact_slot(ra, order_in, slot, n_in, nf, tiles_x, tiles_y, tile, head, true);
    __shared__ uint32_t w[2][4];
    const unsigned long long m = __ballot(walk), mr = __ballot(walk && head);
    if ((threadIdx.x & 63u) == 0u) { w[0][threadIdx.x >> 6] = (uint32_t)__popcll(m); w[1][threadIdx.x >> 6] = (uint32_t)__popcll(mr); }
    __syncthreads();
    if (threadIdx.x < 2) block_marked[threadIdx.x * gridDim.x + blockIdx.x] = w[threadIdx.x][0] + w[threadIdx.x][1] + w[threadIdx.x][2] + w[threadIdx.x][3];
}
// Frames by the dozen make tens of thousands of workgroups, and every one adding up all the counts before its own is
// quadratic (32 frames of 8K: 8 G loads); from COMPACT_SCAN_FROM workgroups on, one launch in between turns the two arrays
// into exclusive prefixes (one workgroup each) and leaves the second array's total in word [2 n].
constexpr uint32_t COMPACT_SCAN_FROM = 1024u;
__global__ __launch_bounds__(1024) void k_compact_scan(uint32_t* __restrict__ block_marked, uint32_t n_blocks) {
    uint32_t* __restrict__ a = block_marked + blockIdx.x * n_blocks;
    __shared__ uint32_t sh[1024];
    const uint32_t per = (n_blocks + 1023u) / 1024u;
    const uint32_t b0 = threadIdx.x * per, b = b0 < n_blocks ? b0 : n_blocks, e = b + per < n_blocks ? b + per : n_blocks;
    uint32_t sum = 0;
    for (uint32_t i = b; i < e; ++i) sum += a[i];
    sh[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const uint32_t o = threadIdx.x >= (unsigned)d ? sh[threadIdx.x - d] : 0;
        __syncthreads();
        sh[threadIdx.x] += o;
        __syncthreads();
    }
    uint32_t run = sh[threadIdx.x] - sum;
    for (uint32_t i = b; i < e; ++i) { const uint32_t v = a[i]; a[i] = run; run += v; }
    if (threadIdx.x == 1023 && blockIdx.x == 1) block_marked[2u * n_blocks] = sh[1023];
}
// The queue: first the marked tiles from behind order_in's front -- tiles the launch that left the order did not walk, or
// hardly: with a camera that moves they are the new silhouette, the longest tiles of the frame for all anyone knows, and at
// the queue's end they were what an orbit's launches ended with (config 2: k_render 556 us with them last, the rest of
// the queue done after 355) -- then the front's marked tiles, both in order_in's order.
__global__ __launch_bounds__(256) void k_compact_queue(RenderArgs ra, const uint32_t* __restrict__ order_in, const uint32_t* __restrict__ n_front_in,
                                                      const uint32_t* __restrict__ block_marked, bool scanned, uint32_t* __restrict__ queue, uint32_t* __restrict__ counters) {
    const uint32_t tiles_x = (uint32_t)(ra.width + 7) >> 3;
    const uint32_t tiles_y = (uint32_t)(ra.local_rows + 7) >> 3;
    const uint32_t n_tiles = tiles_x * tiles_y * (uint32_t)ra.n_frames;
    const uint32_t n_in = order_in ? n_tiles - *ra.n_split : n_tiles;
    const uint32_t nf = (order_in && n_front_in) ? *n_front_in : 0u;
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    uint32_t tile;
    bool rest;      // to the queue's head
    const bool walk = compact_slot(ra, order_in, slot, n_in, nf, tiles_x, tiles_y, tile, rest, false);
    // the marked tiles of the workgroups before this one (all of them, and those behind the front), and the latter's total
    uint32_t before = 0u, before_rest = 0u, all_rest = 0u;
    if (scanned) {
        // (k_compact_scan has been over the two arrays: exclusive prefixes, the second array's total behind it)
        if (threadIdx.x == 0) { before = block_marked[blockIdx.x]; before_rest = block_marked[gridDim.x + blockIdx.x]; all_rest = block_marked[2u * gridDim.x]; }
    } else {
        for (uint32_t i = threadIdx.x; i < gridDim.x; i += 256u) {
            const uint32_t r = block_marked[gridDim.x + i];
            all_rest += r;
            if (i < blockIdx.x) { before += block_marked[i]; before_rest += r; }
        }
    }
    before = (uint32_t)wave_sum64(before); before_rest = (uint32_t)wave_sum64(before_rest); all_rest = (uint32_t)wave_sum64(all_rest);      // (lane 0 holds the sums)
    __shared__ uint32_t w[2][4], wb[3][4];
    const unsigned long long m = __ballot(walk), mr = __ballot(walk && rest);
    const int lane = (int)(threadIdx.x & 63u), wv = (int)(threadIdx.x >> 6);
    if (lane == 0) { w[0][wv] = (uint32_t)__popcll(m); w[1][wv] = (uint32_t)__popcll(mr); wb[0][wv] = before; wb[1][wv] = before_rest; wb[2][wv] = all_rest; }
    __syncthreads();
    uint32_t b_all = wb[0][0] + wb[0][1] + wb[0][2] + wb[0][3], b_rest = wb[1][0] + wb[1][1] + wb[1][2] + wb[1][3];
    const uint32_t t_rest = wb[2][0] + wb[2][1] + wb[2][2] + wb[2][3];
    for (int k = 0; k < wv; ++k) { b_all += w[0][k]; b_rest += w[1][k]; }
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint32_t at_all = b_all + (uint32_t)__popcll(m & below);        // the marked slots before this one
    const uint32_t at_rest = b_rest + (uint32_t)__popcll(mr & below);     // ... those of them behind the front
    if (walk) queue[rest ? at_rest : t_rest + (at_all - at_rest)] = tile;
    if (slot == nf) counters[1] = t_rest + at_all;                        // (no marked slot behind the front lies before slot nf)
    if (slot + 1u == gridDim.x * 256u) {
        const uint32_t total = at_all + (walk ? 1u : 0u);
        counters[0] = total;
        if (nf >= gridDim.x * 256u) counters[1] = total;
    }
}

// Persistent threads: every wave loops pulling 8x8 pixel tiles until the queue is empty; the
// exit condition (tile >= n_tiles) is reached by every wave whatever the grid size.
// ABORT: the variant for launches without tile history (see "gives up" below); launches with history run the
// plain one -- the bookkeeping costs 4 % of a config 2 frame, and history already sends long tiles elsewhere.
// Eight waves per SIMD (64 VGPRs): the kernel is bound by VALU issue with long dependent chains, and every further
// resident wave fills issue slots -- 5 / 6 / 7 / 8 waves: 91.4 / 81.8 / 78.4 / 73.6 ms on config 4.  The plain kernel
// fits in 63 registers with no scratch (scripts/kernel_resources.sh) since the camera frame stopped living in twelve
// vector registers across the walk (render_pixel); the counting variants keep 24-40 bytes per lane there.
#ifndef HVRT_RENDER_WAVES
#define HVRT_RENDER_WAVES 8
#endif
#ifndef HVRT_SPLIT_WAVES
#define HVRT_SPLIT_WAVES 6           // k_render_split: 80 vector registers instead of 84 (88 allocated) -- four of its waves and three of k_render's share a SIMD's 512
#endif
#ifndef HVRT_RENDER_BLOCK
#define HVRT_RENDER_BLOCK 256          // lanes per workgroup of k_render / k_render_segment
#endif
constexpr uint32_t RENDER_BLOCK_WAVES = HVRT_RENDER_BLOCK / 64;
template <bool COUNT, bool INLINE, bool NEAREST, bool ABORT>
__device__ __forceinline__ void render_tiles(const SceneDev& sc, const RenderArgs& ra) {
    __shared__ float4 lds_win[RENDER_BLOCK_WAVES][66];     // per wave: the list chunk its lanes are testing (trace_ray_wave), [64..65] the tile's direction box
    // (the lane number is worked out where it is needed, lane_now(): a register held for it would be spilled)
    float4* const win = lds_win[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))];
    __shared__ uint32_t lds_tiles[RENDER_BLOCK_WAVES][64];       // per wave: the tiles of the pull in hand, one per slot
    uint32_t* const my_tiles = lds_tiles[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))];
    const uint32_t tiles_x = (uint32_t)(ra.width + 7) >> 3;
    const uint32_t tiles_y = (uint32_t)(ra.local_rows + 7) >> 3;
    const uint32_t tiles_frame = tiles_x * tiles_y;
    const uint32_t n_tiles = tiles_frame * (uint32_t)ra.n_frames;
    // with an order, the tiles that k_render_split takes are not in this kernel's queue; a launch may be given a
    // range of the queue's slots only (RenderArgs::seg_slots)
    const uint32_t slot0 = ra.seg_slots ? ra.seg_slots[ra.seg] : 0u;
    const uint32_t n_queue = ra.n_queue ? *ra.n_queue : ra.seg_slots ? ra.seg_slots[ra.seg + 1] : (ra.order ? n_tiles - *ra.n_split : n_tiles);
    const uint32_t n_front = ra.order && ra.n_front ? *ra.n_front : 0u;      // the slots of the tiles the order put in front for their length
    uint64_t tot_nv = 0, tot_nt = 0;
#if defined(HVRT_DIAG) && HVRT_DIAG == 32       // a wave's lifetime in the kernel | the clocks it spent inside tiles (16-clock units x 16)
    const long long life0 = clock64();
#endif
#if defined(HVRT_DIAG) && (HVRT_DIAG >= 34 && HVRT_DIAG <= 41)   // 35: the hit count becomes the sum of the waves' lifetimes (10-ns ticks), any variant; 36: of their start delays behind workgroup 0
    const unsigned long long life_wall0 = wall_clock64();
#if HVRT_DIAG >= 36
    if (blockIdx.x == 0 && threadIdx.x == 0) ra.abandoned[0] = (uint32_t)life_wall0;
#endif
#endif
    if (ABORT && ra.few_report && blockIdx.x == 0 && threadIdx.x == 0) {
        // (the verdict, then the launch's number: see RenderArgs::few_report)
        __hip_atomic_store(ra.few_report, ra.n_queue && n_queue <= ra.few_tiles ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(ra.few_report + 1, ra.few_serial, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    uint32_t tot_hits = 0;                // wave-uniform
#if defined(HVRT_DIAG) && HVRT_DIAG == 41
    uint32_t n_done = 0;
#endif
    unsigned long long wave_cost = 0;     // wave-uniform
    uint32_t n_walked = 0;                // wave-uniform: tiles this wave walked (all of them without a tile cover)
    uint32_t given_up = 0xFFFFFFFFu;      // wave-uniform: the tile the previous iteration gave up on (see below)
    // The queue head is ONE word, and one word takes about 88 returning atomics per microsecond from the whole chip
    // (MI355X_MICROARCH.md, "dequeue"): 129 600 single-tile pulls cannot finish in under 1.5 ms, whatever the tiles
    // cost -- a 4K frame of config 2 was bound by exactly that (a build whose rays do not walk at all took the
    // same 1.77 ms).  So a wave pulls `batch` consecutive slots per atomic and sizes the next batch from what the last
    // one cost per tile: about PULL_CLOCKS of work per pull (7000 waves then stay well under the word's rate), one
    // tile at a time where tiles are long (the head of an ordered queue, the silhouette, supersampled frames),
    // and never more than the waves' even share of what is left of the queue, so that the launch does not end on a
    // wave with a full batch in hand.
    // Without an order (a first launch) the long tiles lie where the image has them -- side by side along the top and
    // bottom of a silhouette -- and a wave that sized its batch on cheap background tiles walks into eight of them in a
    // row: config 2 from 10 degrees to the side took 3.7 ms with no tile above 1.2 ms and a balanced launch of 0.63
    // (scripts/time_cold_views.py).  Such launches pull one tile at a time (2.4 ms; two at a time 2.8).
#ifndef HVRT_COLD_BATCH
#define HVRT_COLD_BATCH 1
#endif
#ifndef HVRT_PULL_CLOCKS
#define HVRT_PULL_CLOCKS 200000
#endif
    constexpr long long PULL_CLOCKS = HVRT_PULL_CLOCKS;
#ifndef HVRT_MAX_BATCH
#define HVRT_MAX_BATCH 8
#endif
    constexpr uint32_t MAX_BATCH = HVRT_MAX_BATCH;
    // (with a tile cover most slots hold tiles that are passed by: up to 64 at a time, one per lane; a launch without history
    // takes its tiles where the image has them, HVRT_COLD_BATCH at a time -- unless most of them are passed by)
    const uint32_t max_batch = (!COUNT && ra.cover) ? 64u : MAX_BATCH;
    const uint32_t cold_batch = (!COUNT && ra.cover) ? 16u : (uint32_t)HVRT_COLD_BATCH;
    // Eight heads (RenderArgs::n_heads), one per XCD (workgroups go to the XCDs round-robin, so blockIdx.x & 7 is the
    // XCD's number): behind the first n_waves slots the queue is cut into chunks of HVRT_QUEUE_CHUNK slots, chunk c belongs to
    // head c & 7, and head h hands out the slots of its chunks in order.  The waves of one XCD then work on runs of
    // neighbouring tiles -- their voxels, lists and records in that XCD's L2 -- instead of every XCD touching every row.
    // A wave whose head is used up goes on to the next (at most eight rounds: every wave leaves when all are used up).
#ifndef HVRT_QUEUE_CHUNK_LOG2
#define HVRT_QUEUE_CHUNK_LOG2 9
#endif
    uint32_t batch = 1, slot = 0, g = 0, g0 = 0;      // wave-uniform: g: the next slot's number at its head, g0: the first one of the pull in hand
    unsigned long long pend = 0ull;                   // wave-uniform: bit i: slot i of the pull in hand holds a tile that is still to be walked
    const uint32_t heads = ra.n_heads > 1u ? ra.n_heads : 1u;              // a power of two
    // (the eight heads lie 128 bytes apart, tile_counter[64 + 32 h]: atomics on one cache line queue up behind one another)
    uint32_t* const head0 = heads > 1u ? ra.tile_counter + 64 : ra.tile_counter;
    const uint32_t head_stride = heads > 1u ? 32u : 0u;
    uint32_t head = blockIdx.x & (heads - 1u);
    bool own_slot = false;
    long long batch_clk0 = 0;
    const uint32_t n_waves = gridDim.x * RENDER_BLOCK_WAVES;
    // A wave's FIRST slot is its own number: 8192 waves asking the head for one slot each at the same moment is 93 us
    // of the word's atomic rate, 46 us of waiting for the mean wave, in a 1.7-ms launch.  The head hands out what lies
    // behind those first n_waves slots.
    bool first_pull = true;
    // slot number g of the wave's current head (its chunks, in order; a head's slots rise with g)
    auto slot_of = [&](uint32_t gg) -> unsigned long long {
        return heads > 1u
            ? (unsigned long long)slot0 + n_waves + ((((unsigned long long)(gg >> HVRT_QUEUE_CHUNK_LOG2) * heads + head) << HVRT_QUEUE_CHUNK_LOG2) | (gg & ((1u << HVRT_QUEUE_CHUNK_LOG2) - 1u)))
            : (unsigned long long)slot0 + n_waves + gg;
    };
    // size of the next pull: PULL_CLOCKS of work at the last pull's cost per slot, halved while it exceeds the waves' even
    // share of the slots that are left (last: the last slot of the pull that has just been finished)
    auto size_next_pull = [&](unsigned long long last) {
        const long long per_tile = (clock64() - batch_clk0) / (long long)batch;
        const uint32_t grown = batch * 2u;
        batch = per_tile * 4 > PULL_CLOCKS ? 1u : per_tile * 8 > PULL_CLOCKS ? 2u : per_tile * 16 > PULL_CLOCKS ? 4u : per_tile * 32 > PULL_CLOCKS ? 8u : max_batch;
        // One cheap tile says little about the next ones: a pull at most doubles, and among the tiles the order put in front
        // for their length it stays at one.  (With a tile cover a tile of that kind can turn out not to be walked at all --
        // its wave then took eight supersampled silhouette tiles in one pull and ended a 7-ms launch at 24 ms.)
        if (batch > grown) batch = grown;
        if (n_front && last < (unsigned long long)n_front) batch = 1u;
        if (!ra.order && batch > cold_batch) batch = cold_batch;
        const unsigned long long left = (unsigned long long)n_queue > last ? (unsigned long long)n_queue - last : 0ull;
        while (batch > 1u && (unsigned long long)batch * n_waves > left) batch >>= 1;
    };

    // ---- tiles too long for one wave, sample by sample (RenderArgs::sample_rgb): a supersampled tile is spp independent
    // pieces of work but for the order in which the pixel's samples are summed.  Every sample of such a tile is an item of
    // its own; its colours go to a scratch array, and the wave that finishes the tile's last sample sums them in sample
    // order -- the sum render_pixel forms -- and stores the pixels.  (Tiles past the scratch array's room are items as a
    // whole.)  An item's first number is the wave's own, like a slot's below.
    if constexpr (!ABORT) {
        if (ra.sample_rgb) {
            const uint32_t n_sp = *ra.n_split, spp_u = (uint32_t)ra.spp;
            const uint32_t n_by_sample = n_sp < ra.sample_cap ? n_sp : ra.sample_cap;
            const uint32_t n_items = n_by_sample * spp_u + (n_sp - n_by_sample);
            uint32_t item = blockIdx.x * RENDER_BLOCK_WAVES + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
            while (item < n_items) {
                const long long clk0 = clock64();
                const uint32_t sl = item < n_by_sample * spp_u ? item / spp_u : n_by_sample + (item - n_by_sample * spp_u);
                const int smp = item < n_by_sample * spp_u ? (int)(item - sl * spp_u) : -1;
                const uint32_t tile = ra.split_tiles[sl];
                const uint32_t fi = tile / tiles_frame, tf = tile - fi * tiles_frame;
                const uint32_t ty = tf / tiles_x, tx = tf - ty * tiles_x;
                int x, lr, y;
                pixel_coords(ra, tx, ty, -1, x, lr, y);
                const bool inside = x < ra.width && lr < ra.local_rows && y < ra.height;
                PixelResult px;
                px.nv = px.nt = 0;
                if constexpr (INLINE) render_pixel<COUNT, INLINE, NEAREST, 64, false>(sc, ra, ra.frames ? ra.frames + fi : nullptr, tx, ty, -1, 0, 0, inside, win, 0, px, smp);
                else if (inside) render_pixel<COUNT, INLINE, NEAREST, 64, false>(sc, ra, ra.frames ? ra.frames + fi : nullptr, tx, ty, -1, 0, 0, true, nullptr, 0, px, smp);
                pixel_coords(ra, tx, ty, -1, x, lr, y);
                if (smp < 0) {
                    if (inside) store_pixel(ra, fi, lr, x, px);
                } else {
                    const size_t p = ((size_t)fi * ra.slab_rows + lr) * ra.width + x;
                    if (smp == 0 && inside) { if (ra.hit_id) ra.hit_id[p] = px.hit; if (ra.t) ra.t[p] = px.t; }
                    if (ra.rgb) {
                        float4* const mine = ra.sample_rgb + ((size_t)sl * spp_u + (uint32_t)smp) * 64u + (uint32_t)lane_now();
                        __hip_atomic_store(&mine->x, px.rgb.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&mine->y, px.rgb.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&mine->z, px.rgb.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __threadfence();
                        uint32_t before = 0u;
                        if (lane_now() == 0) before = __hip_atomic_fetch_add(ra.sample_done + sl, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                        before = (uint32_t)__builtin_amdgcn_readfirstlane((int)before);
                        if (before == spp_u - 1u) {
                            // the tile's last sample: every other sample's colours were released before its count
                            __threadfence();
                            V3 acc = { 0.0f, 0.0f, 0.0f };
                            for (uint32_t k = 0; k < spp_u; ++k) {
                                const float4* const q = ra.sample_rgb + ((size_t)sl * spp_u + k) * 64u + (uint32_t)lane_now();
                                const V3 c = { __hip_atomic_load(&q->x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                               __hip_atomic_load(&q->y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                               __hip_atomic_load(&q->z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) };
                                acc = k == 0u ? c : add(acc, c);
                            }
                            const float n = ra.spp_f;
                            if (inside) { ra.rgb[3 * p] = acc.x / n; ra.rgb[3 * p + 1] = acc.y / n; ra.rgb[3 * p + 2] = acc.z / n; }
                            if (lane_now() == 0) __hip_atomic_store(ra.sample_done + sl, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // as the next launch expects it
                        }
                    }
                }
                const unsigned long long dt = (unsigned long long)(clock64() - clk0) >> 4;
                const uint32_t dc = dt > 0x00FFFFFFull ? 0x00FFFFFFu : (uint32_t)dt;
                if (lane_now() == 0) atomicAdd(&ra.tile_cost[tile], dc);      // the tile's cost: the sum over its samples (its word holds the bare flag)
                wave_cost += dc;
                if (smp <= 0) {
                    ++n_walked;
                    if (ra.hits) tot_hits += (uint32_t)__builtin_popcountll(any_lane(inside && px.hit != HVRT_MISS));
                }
                if (COUNT) { tot_nv += px.nv; tot_nt += px.nt; }
                uint32_t got = 0u;
                if (lane_now() == 0) got = atomicAdd(ra.tile_counter + 2, 1u);
                item = n_waves + (uint32_t)__builtin_amdgcn_readlane((int)got, 0);
            }
        }
    }

    for (;;) {
        if (ABORT && lane_now() == 0 && given_up != 0xFFFFFFFFu) ra.abandoned[atomicAdd(ra.n_abandoned, 1u)] = given_up;
        given_up = 0xFFFFFFFFu;
        if (pend == 0ull) {
#if defined(HVRT_DIAG) && HVRT_DIAG == 31       // clocks spent waiting for the queue head | clocks in the loop, per wave
            const long long pull0 = clock64();
#endif
            if (first_pull) {
                first_pull = false;
                own_slot = true;
                batch = 1u;
                slot = slot0 + blockIdx.x * RENDER_BLOCK_WAVES + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
                if (slot >= n_queue) break;
            } else {
                own_slot = false;
                bool found = false;
                for (uint32_t tries = 0; tries < heads; ++tries) {
                    // (a look before the atomic once a head has failed: 8192 waves x 8 used-up heads would otherwise end
                    // the launch with 65 000 atomics)
                    uint32_t got = 0;
                    if (lane_now() == 0) {
                        uint32_t* const hp = head0 + head * head_stride;
                        got = tries == 0u ? atomicAdd(hp, batch) : __hip_atomic_load(hp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    g = (uint32_t)__builtin_amdgcn_readlane((int)got, 0);
                    const unsigned long long at = heads > 1u
                        ? (unsigned long long)slot0 + n_waves + ((((unsigned long long)(g >> HVRT_QUEUE_CHUNK_LOG2) * heads + head) << HVRT_QUEUE_CHUNK_LOG2) | (g & ((1u << HVRT_QUEUE_CHUNK_LOG2) - 1u)))
                        : (unsigned long long)slot0 + n_waves + g;
                    if (at < (unsigned long long)n_queue) {
                        if (tries == 0u) { found = true; break; }
                        // the look says this head has slots left: claim some (it may have run out meanwhile -- then on)
                        uint32_t got2 = 0;
                        if (lane_now() == 0) got2 = atomicAdd(head0 + head * head_stride, batch);
                        g = (uint32_t)__builtin_amdgcn_readlane((int)got2, 0);
                        const unsigned long long at2 = (unsigned long long)slot0 + n_waves + ((((unsigned long long)(g >> HVRT_QUEUE_CHUNK_LOG2) * heads + head) << HVRT_QUEUE_CHUNK_LOG2) | (g & ((1u << HVRT_QUEUE_CHUNK_LOG2) - 1u)));
                        if (at2 < (unsigned long long)n_queue) { found = true; break; }
                    }
                    head = (head + 1u) & (heads - 1u);
                    // (another head may still be among its long tiles, whatever this wave's last tiles cost: one at a time
                    // again -- eight supersampled silhouette tiles in one pull once ended a 7-ms launch at 24 ms)
                    batch = 1u;
                }
                if (!found) break;
            }
#if defined(HVRT_DIAG) && HVRT_DIAG == 31
            if (COUNT && lane_now() == 0) { tot_nv += (uint64_t)(clock64() - pull0); tot_nt += 1; }
#endif
            batch_clk0 = clock64();
            // ---- what the pull holds: lane i looks up slot i's tile and, with a tile cover, whether it is marked.  The tiles
            // that are not (their pixels are k_fill_background's) are passed by here, up to 64 of them at the price of one.
            const uint32_t ln = (uint32_t)lane_now();
            const unsigned long long my_at = own_slot ? (unsigned long long)slot : slot_of(g + ln);
            bool valid = ln < batch && my_at < (unsigned long long)n_queue;
            uint32_t my_tile = 0xFFFFFFFFu;
            if (valid) {
                if (ra.order) {
                    my_tile = ra.order[my_at];
                } else {
                    my_tile = natural_tile((uint32_t)my_at, tiles_x, tiles_y);      // no history for this shape yet
                }
            }
            valid = valid && my_tile < n_tiles;      // (always, with a well-formed order)
            bool walk = valid;
            if constexpr (!COUNT) {
                if (ra.cover && valid) {
                    const uint32_t fi = my_tile / tiles_frame, tf = my_tile - fi * tiles_frame;
                    const uint32_t ty = tf / tiles_x, tx = tf - ty * tiles_x;
                    walk = tile_covered(ra, fi, tx, ty);
                    if (!walk && !ra.debug_start) ra.tile_cost[my_tile] = 0u;
                }
            }
            my_tiles[ln] = my_tile;
            pend = any_lane(walk);
            g0 = g;
            g += batch;
            if (pend == 0ull) {
                if (own_slot) continue;      // (its one tile is not walked: on to the queue's heads, one tile at first)
                size_next_pull(slot_of(g0 + batch - 1u));
                continue;
            }
        }
        const uint32_t bit = (uint32_t)__builtin_ctzll(pend);
        pend &= pend - 1ull;
        if (!own_slot) slot = (uint32_t)slot_of(g0 + bit);
        const uint32_t tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)my_tiles[bit]);   // wave-uniform (an SGPR)
        const long long clk0 = clock64();
        // the give-up deadline runs on the constant 100 MHz clock (s_memrealtime): the shader clock, which prices the
        // tiles, moves with the chip's power state
        // (few tiles in the whole queue: a deadline that has passed, see RenderArgs::few_tiles)
        const long long wall0 = ABORT ? (ra.n_queue && n_queue <= ra.few_tiles ? 0ll : (long long)wall_clock64()) : 0;
        const unsigned long long debug_wall0 = ra.debug_start ? wall_clock64() : 0ull;      // (a diagnostic: when the tile started, 10-ns ticks)
        const uint32_t fi = tile / tiles_frame, tf = tile - fi * tiles_frame;
        const uint32_t ty = tf / tiles_x, tx = tf - ty * tiles_x;
        int x, lr, y;
        pixel_coords(ra, tx, ty, -1, x, lr, y);
        // A tile whose rays are still inside long lists a millisecond after they set out is not finished here:
        // the wave drops it (nothing of it is written or counted) and k_render_split, launched behind this
        // kernel, renders it with 16 waves.  This is what bounds a launch that has no history of the scene yet;
        // from the next launch on the recorded durations decide which tiles go to k_render_split from the start.
        bool gave_up = false;
        PixelResult px;
        px.nv = px.nt = 0;
        const bool inside = x < ra.width && lr < ra.local_rows && y < ra.height;
        ++n_walked;
        if constexpr (INLINE) {
            // wave-synchronous walk: lanes outside the image come along without a ray; giving up is wave-wide
#if defined(HVRT_ABLATE) && HVRT_ABLATE >= 6
            px.rgb = { 0.5f, 0.5f, (float)x }; px.hit = HVRT_MISS; px.t = FLT_MAX;
#else
            gave_up = !render_pixel<COUNT, INLINE, NEAREST, 64, ABORT>(sc, ra, ra.frames ? ra.frames + fi : nullptr, tx, ty, -1, 0, 0, inside, win, wall0, px);
#endif
#if !defined(HVRT_ABLATE) || (HVRT_ABLATE != 4 && HVRT_ABLATE != 7)
            int x2, lr2, y2;
            pixel_coords(ra, tx, ty, -1, x2, lr2, y2);
            if (inside && (!ABORT || !gave_up)) store_pixel(ra, fi, lr2, x2, px);
#endif
        } else if (inside) {
            gave_up = !render_pixel<COUNT, INLINE, NEAREST, 64, ABORT>(sc, ra, ra.frames ? ra.frames + fi : nullptr, tx, ty, -1, 0, 0, true, nullptr, wall0, px);
            // the wave decides; a lane that finished its pixel holds its stores back until then
            if (!ABORT || any_lane(gave_up) == 0ull) store_pixel(ra, fi, lr, x, px);
        }
        // every lane is back here (the block above is structured); the clock is a scalar, so all lanes store
        // the same word -- no lane-0 branch at the loop's end, which keeps the loop's control flow simple
        {
            const bool dropped = ABORT && any_lane(gave_up) != 0ull;
            const unsigned long long dt = (unsigned long long)(clock64() - clk0) >> 4;
            const uint32_t dc = dt > 0x00FFFFFFull ? 0x00FFFFFFu : (uint32_t)dt;
#if !defined(HVRT_ABLATE) || (HVRT_ABLATE != 5 && HVRT_ABLATE != 7)
            ra.tile_cost[tile] = ra.debug_start ? (uint32_t)debug_wall0 : dc;      // a dropped tile: k_render_split adds its quads' durations to this
            if (!ABORT && ra.debug_start) ra.abandoned[tile] = slot | ((blockIdx.x & 7u) << 28) | (own_slot ? 0x80000000u : 0u);      // (the diagnostic: which slot, whose wave)
#endif
            wave_cost += dc;
#if defined(HVRT_DIAG) && HVRT_DIAG == 41
            ++n_done;
#endif
            if (dropped) { given_up = tile; --n_walked; }        // appended to the list by the next iteration; k_render_split counts it among the walked
#if !defined(HVRT_DIAG) || (HVRT_DIAG != 31 && HVRT_DIAG != 32 && HVRT_DIAG != 33 && HVRT_DIAG != 34)
            else if (COUNT) { tot_nv += px.nv; tot_nt += px.nt; }
#endif
            if (ra.hits && !dropped) tot_hits += (uint32_t)__builtin_popcountll(any_lane(inside && px.hit != HVRT_MISS));
        }
        if (pend == 0ull) size_next_pull(own_slot ? slot : slot_of(g0 + batch - 1u));
    }
    if (ABORT && lane_now() == 0 && given_up != 0xFFFFFFFFu) ra.abandoned[atomicAdd(ra.n_abandoned, 1u)] = given_up;
#if defined(HVRT_DIAG) && HVRT_DIAG == 32
    if (COUNT) { tot_nv = lane_now() == 0 ? (uint64_t)(clock64() - life0) : 0ull; tot_nt = lane_now() == 0 ? (uint64_t)wave_cost << 4 : 0ull; }
#endif
#if defined(HVRT_DIAG) && (HVRT_DIAG == 33 || HVRT_DIAG == 34)   // 33: sum | max of the waves' end times, 34: of their start times (100 MHz wall clock)
    if (COUNT && ra.counters) {
        if (lane_now() == 0) {
#if HVRT_DIAG == 33
            const unsigned long long tw = wall_clock64();
#else
            const unsigned long long tw = life_wall0;
#endif
            atomicAdd(&ra.counters[0], tw & 0xFFFFFFFFull);
            atomicMax(&ra.counters[1], HVRT_DIAG == 33 ? (tw & 0xFFFFFFFFull) : (0xFFFFFFFFull - (tw & 0xFFFFFFFFull)));
        }
        tot_nv = tot_nt = 0;
    }
#endif
    const int lane = lane_now();
    if (lane == 0 && wave_cost) atomicAdd(ra.cost_sum, wave_cost);
    if (lane == 0 && n_walked && ra.walked) atomicAdd(ra.walked, (unsigned long long)n_walked);
#if defined(HVRT_DIAG) && HVRT_DIAG == 35
    if (lane == 0 && ra.hits) atomicAdd(ra.hits, wall_clock64() - life_wall0);
#elif defined(HVRT_DIAG) && HVRT_DIAG == 40     // waves that started more than 0.5 ms behind workgroup 0 (and, << 32, more than 20 us)
    if (lane == 0 && ra.hits) {
        const int d = (int)((uint32_t)life_wall0 - ra.abandoned[0]);
        atomicAdd(ra.hits, (unsigned long long)(d > 50000 ? 1 : 0) | ((unsigned long long)(d > 2000 ? 1 : 0) << 32));
    }
#elif defined(HVRT_DIAG) && HVRT_DIAG == 41     // tiles rendered by waves that started more than 0.5 ms late
    if (lane == 0 && ra.hits) { const int d = (int)((uint32_t)life_wall0 - ra.abandoned[0]); atomicAdd(ra.hits, (unsigned long long)(d > 50000 ? n_done : 0u)); }
#elif defined(HVRT_DIAG) && HVRT_DIAG >= 36 && HVRT_DIAG <= 39    // sum / max of start delays (36 / 37), sum / max of end times (38 / 39), behind workgroup 0's start
    if (lane == 0 && ra.hits) {
        const uint32_t tw = HVRT_DIAG <= 37 ? (uint32_t)life_wall0 : (uint32_t)wall_clock64();
        const int d = (int)(tw - ra.abandoned[0]);
        const unsigned long long v = (unsigned long long)(d > 0 ? d : 0);
        if (HVRT_DIAG == 36 || HVRT_DIAG == 38) atomicAdd(ra.hits, v); else atomicMax(ra.hits, v);
    }
#else
    if (lane == 0 && tot_hits) atomicAdd(ra.hits, (unsigned long long)tot_hits);
#endif
    if (COUNT && ra.counters) {
        tot_nv = wave_sum64(tot_nv);
        tot_nt = wave_sum64(tot_nt);
        if (lane == 0) {
            atomicAdd(&ra.counters[0], (unsigned long long)tot_nv);
            atomicAdd(&ra.counters[1], (unsigned long long)tot_nt);
        }
    }
}

template <bool COUNT, bool INLINE, bool NEAREST, bool ABORT>
__global__ __launch_bounds__(HVRT_RENDER_BLOCK) __attribute__((amdgpu_waves_per_eu(HVRT_RENDER_WAVES, HVRT_RENDER_WAVES))) void k_render(SceneDev sc, RenderArgs ra) {
    render_tiles<COUNT, INLINE, NEAREST, ABORT>(sc, ra);
}
// The same code under a second name, for launches over a range of the queue's slots (RenderArgs::seg_slots, the host
// path's half frames): a kernel trace then lists whole-frame launches and part launches separately instead of
// averaging them into one row.
template <bool COUNT, bool INLINE, bool NEAREST>
__global__ __launch_bounds__(HVRT_RENDER_BLOCK) __attribute__((amdgpu_waves_per_eu(HVRT_RENDER_WAVES, HVRT_RENDER_WAVES))) void k_render_segment(SceneDev sc, RenderArgs ra) {
    render_tiles<COUNT, INLINE, NEAREST, false>(sc, ra);
}

// ---- tiles too long for one wave ------------------------------------------------------------------------
// A launch cannot end before its longest tile does, and a tile is one wave walking serially dependent look-ups:
// on the skewed 10 M-triangle scene a few tiles cross voxels with thousands of entries and take 7 ms, whatever
// the frame size, while the rest of the frame is done in 3.  Tiles that the previous launch measured longer than
// the whole launch would take on a perfectly filled GPU are therefore rendered by this kernel instead: 16 waves per tile, each with a 2x2 pixel quad of it, every ray on 16 lanes that share out the entries
// of each list (trace_ray, RAYS = 4) -- or 8 waves per tile with 4x2 pixels each, eight lanes per ray (SR = 8): half the items,
// so that config 2's 380 tiles are one round of the kernel's waves instead of one and a half (0.46 -> 0.44 ms); where
// lists run to hundreds of entries and more the sixteen lanes are worth more (config 3, lists of up to 1 076: the front
// view 0.86 -> 0.81 ms with eight rays per wave, but the view 20 degrees beside it, across the dense patch, 0.97 -> 1.29;
// a 9.8 M-triangle scene with lists of up to 3 700: 0.68 -> 0.72), and so they are behind a launch that gave up on
// tiles, where the kernel has the GPU to itself (config 2: 0.86 against 0.94 ms).  The caller chooses by the grid's
// longest list.  It runs beside k_render on a second stream.  Items are dealt round-robin
// with a bounded loop (no queue): they are few and of one kind.
template <bool COUNT, bool INLINE, bool NEAREST, int SR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HVRT_SPLIT_WAVES, HVRT_SPLIT_WAVES))) void k_render_split(SceneDev sc, RenderArgs ra) {
    const int lane = threadIdx.x & 63;
    // SR, rays per wave: 4 (2 x 2 pixels, 16 waves per tile) or 8 (4 x 2 pixels, 8 waves per tile)
    constexpr uint32_t PER_TILE = 64u / (uint32_t)SR;    // items per tile
    const int ray = lane & (SR - 1), slice = lane / SR;
    const uint32_t tiles_x = (uint32_t)(ra.width + 7) >> 3;
    const uint32_t tiles_y = (uint32_t)(ra.local_rows + 7) >> 3;
    const uint32_t tiles_frame = tiles_x * tiles_y;
    if (ra.split_started && threadIdx.x == 0) atomicAdd(ra.split_started, 1u);      // "this workgroup has a place on the GPU"
    const uint32_t n_items = *ra.n_split * PER_TILE;
    if (ra.walked && blockIdx.x == 0 && threadIdx.x == 0 && n_items) atomicAdd(ra.walked, (unsigned long long)(n_items / PER_TILE));
    const uint32_t wave0 = blockIdx.x * 4u + (threadIdx.x >> 6), n_waves = gridDim.x * 4u;
    uint64_t tot_nv = 0, tot_nt = 0;
    unsigned long long wave_cost = 0;
    for (uint32_t item = wave0; item < n_items; item += n_waves) {
        const long long clk0 = clock64();
        const long long wall0 = 0;      // this kernel never gives up
        const uint32_t tile = ra.split_tiles[item / PER_TILE], quad = item % PER_TILE;
        const uint32_t fi = tile / tiles_frame, tf = tile - fi * tiles_frame;
        const uint32_t ty = tf / tiles_x, tx = tf - ty * tiles_x;
        const int x = SR == 4 ? (int)(tx * 8u) + (int)(quad & 3u) * 2 + (ray & 1) : (int)(tx * 8u) + (int)(quad & 1u) * 4 + (ray & 3);
        const int lr = SR == 4 ? (int)(ty * 8u) + (int)(quad >> 2) * 2 + (ray >> 1) : (int)(ty * 8u) + (int)(quad >> 1) * 2 + (ray >> 2);
        const int kl = lr / ra.tile_rows;
        const int y = (kl * ra.n_parts + ra.part) * ra.tile_rows + (lr - kl * ra.tile_rows);
        if (x < ra.width && lr < ra.local_rows && y < ra.height) {
            PixelResult px;
            render_pixel<COUNT, INLINE, NEAREST, SR, false>(sc, ra, ra.frames ? ra.frames + fi : nullptr, 0u, 0u, lane, x, y, true, nullptr, wall0, px);
            if (slice == 0) {
                store_pixel(ra, fi, lr, x, px);
                if (COUNT) { tot_nv += px.nv; tot_nt += px.nt; }
            }
            if (ra.hits) {
                const uint32_t nh = (uint32_t)__builtin_popcountll(any_lane(slice == 0 && px.hit != HVRT_MISS));
                if (nh && lane == (int)__builtin_ctzll(any_lane(true))) atomicAdd(ra.hits, (unsigned long long)nh);   // the first lane that is here
            }
        }
        const unsigned long long dt = (unsigned long long)(clock64() - clk0) >> 4;
        const uint32_t dc = dt > 0x00FFFFFFull ? 0x00FFFFFFu : (uint32_t)dt;
        // the tile's cost: the sum over its 16 quads
        if (lane == 0) atomicAdd(&ra.tile_cost[tile], dc);
        wave_cost += dc;
    }
    // (the quads' durations go into the launch's sum, 16 per tile: the more tiles are split, the higher "the mean walked tile"
    // and with it the bar for the next one -- without that, every frame of an orbit added its long tiles to the split ones,
    // which stay split: config 2's orbit 0.71 -> 0.86 ms per frame)
    if (lane == 0 && wave_cost) atomicAdd(ra.cost_sum, wave_cost);
    if (COUNT && ra.counters && !(HVRT_DIAG_BUILD && HVRT_DIAG_NO_SPLIT_COUNTS)) {
        tot_nv = wave_sum64(tot_nv);
        tot_nt = wave_sum64(tot_nt);
        if (lane == 0 && (tot_nv | tot_nt)) {
            atomicAdd(&ra.counters[0], (unsigned long long)tot_nv);
            atomicAdd(&ra.counters[1], (unsigned long long)tot_nt);
        }
    }
}

// ---- order of the tile queue: the few long tiles first, the rest as they lie ----------------------------
// Three classes from the previous launch's tile durations (sum = their total, n tiles, w wave slots on the GPU):
//   split  longer than sum / w, the duration of a perfectly balanced launch (a shorter tile that starts first is
//          not what the launch waits for, and splitting it only adds work: measured): to k_render_split.  The
//          class is sticky (top bit of the tile's cost word) for as long as the launch shape stays the same, so
//          that a tile does not flip between the kernels from one launch to the next;
//   long   more than 2.5 x the mean tile (about 0.5 % of a frame: the silhouette, where rays graze the occupied
//          shell): to the head of k_render's queue;
//   the others keep their natural order behind them -- neighbours in the queue stay neighbours in the image,
//   which is what keeps the L1 / L2 hit rates (a full sort by cost was measured 7 % slower than no order).
// Stable three-way partition: per-block counts of the first two classes, scan, placement.

// (walked: the tiles the launch walked -- with a tile cover most tiles of a frame cost next to nothing, and "the mean tile"
// is the mean over the walked ones: sum[2], beside the sum of the durations; 0 = not known, all n of them)
__device__ __forceinline__ uint32_t walked_tiles(const unsigned long long* __restrict__ sum, uint32_t n) {
    const unsigned long long w = sum[2];
    return w == 0ull || w > (unsigned long long)n ? n : (uint32_t)w;
}
// What the order build goes by: the wave slots k_render keeps resident, ten times the second split rule's factor, and the
// lower bounds of the long classes in hundredths of the mean walked tile, highest first.
// (ten times the factor; with the queue in class order, configs 2 / 3 / 4 at 1.2 / 1.5 / 1.8 / 2.0 / 2.2 / 2.5:
// 0.91 / 0.55 / 0.52 / 0.52 / 0.53 / 0.52, 0.96 / 0.91 / 0.90 / 0.92 / 0.94 / 1.00 and - / - / 9.9 / 9.6 / 9.5 / 9.4 ms: flat from 1.8 to 2.2.
// The mean includes k_render_split's quads, see there; with a mean over k_render's tiles alone and the factor at 2.6 the
// configs stood at 0.49 / 0.82 / 8.6 ms and an orbit lost a fifth)
#ifndef HVRT_SPLIT_X10
#define HVRT_SPLIT_X10 18
#endif
#ifndef HVRT_ORDER_CLASSES
#define HVRT_ORDER_CLASSES ORDER_BINS      // long classes in use (1: the tiles above the first bound only)
#endif
constexpr uint32_t SPLIT_MEAN_X10 = HVRT_SPLIT_X10;
struct OrderRule { uint32_t slots, split_x10, bin_x100[ORDER_BINS]; };
__device__ __forceinline__ bool tile_wants_split(uint32_t c, unsigned long long sum, const OrderRule& r, uint32_t walked) {
    // longer than the balanced launch, and (where the walked tiles are hardly more than the wave slots, so that nearly
    // every one of them is "longer than the balanced launch") longer than split_x10 / 10 mean walked tiles
    return (c & COST_SPLIT_FLAG) || ((unsigned long long)c * r.slots > sum && (unsigned long long)c * walked * 10ull > (unsigned long long)r.split_x10 * sum);
}
// Classes, in the order of the queue: ORDER_BINS long classes (the longest tiles first: what starts with the launch is
// resident from its first microsecond, what comes later waits for a wave slot -- a launch with a tile cover ends with the
// tile that was both long and late), then the other walked tiles (more than 2.5 x the mean over ALL n tiles: with a tile
// cover, where most tiles cost next to nothing, that is every walked tile; without one walked == n and the class is
// empty), then the rest; each class in natural order -- neighbours in the queue stay neighbours in the image.
// Returns the index of the class's count array: 0 split, 1 .. ORDER_BINS the long classes, ORDER_BINS + 1 walked; -1 the rest.
__device__ __forceinline__ int tile_class(uint32_t c, unsigned long long sum, uint32_t n, uint32_t walked, const OrderRule& r, bool may_split) {
    if (may_split && tile_wants_split(c, sum, r, walked)) return 0;
    const unsigned long long cw = (unsigned long long)(c & ~COST_SPLIT_FLAG);
    // (most tiles walked -- no tile cover: one long class, more than 2.5 mean tiles, about 0.5 % of a frame; the others stay
    // where they lie: a queue in cost order loses the caches what it gains in balance -- config 3 without a cover 2.50 ms
    // with the one class, 2.93 with all of them)
    if ((unsigned long long)walked * 2ull > (unsigned long long)n) return cw * 2ull * walked > 5ull * sum ? 1 : -1;
#pragma unroll
    for (int j = 0; j < HVRT_ORDER_CLASSES; ++j)
        if (cw * walked * 100ull > (unsigned long long)r.bin_x100[j] * sum) return 1 + j;
    return cw * 2ull * n > 5ull * sum ? ORDER_BINS + 1 : -1;
}
// block_count: ORDER_ARRAYS arrays of n_blocks + 1 words: per block of 256 tiles, the tiles that qualify for splitting
// (allowed or not), the tiles of each long class, the other walked tiles
__global__ __launch_bounds__(256) void k_order_count(const uint32_t* __restrict__ cost, uint32_t n,
                                                     const unsigned long long* __restrict__ sum, OrderRule rule,
                                                     bool may_split, uint32_t* __restrict__ block_count) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t walked = walked_tiles(sum, n);
    const bool wants = i < n && tile_wants_split(cost[i], *sum, rule, walked);
    const int cls = i < n ? tile_class(cost[i], *sum, n, walked, rule, may_split) : -1;
    __shared__ uint32_t w[ORDER_ARRAYS][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < ORDER_ARRAYS; ++a) {
        const unsigned long long m = __ballot(a == 0 ? wants : cls == a);
        if (lane == 0) w[a][wv] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    if (threadIdx.x < ORDER_ARRAYS) block_count[threadIdx.x * (gridDim.x + 1) + blockIdx.x] = w[threadIdx.x][0] + w[threadIdx.x][1] + w[threadIdx.x][2] + w[threadIdx.x][3];
}
// each of the count arrays <- its exclusive scan, [n_blocks] <- its total (one workgroup per array).  The
// number of tiles that qualify for splitting goes to *wants_split (pinned host memory, one store; the host reads
// it, without waiting, to decide whether the NEXT launch gets a k_render_split); when this launch may not split,
// the first array is cleared instead -- nothing is split.
__global__ __launch_bounds__(1024) void k_order_scan(uint32_t* __restrict__ block_counts, uint32_t n_blocks, bool may_split,
                                                     uint32_t* __restrict__ wants_split, uint32_t serial) {
    uint32_t* __restrict__ block_count = block_counts + blockIdx.x * (n_blocks + 1);
    const bool keep = blockIdx.x != 0 || may_split;
    __shared__ uint32_t sh[1024];
    const uint32_t per = (n_blocks + 1023u) / 1024u;
    const uint32_t b0 = threadIdx.x * per, b = b0 < n_blocks ? b0 : n_blocks, e = b + per < n_blocks ? b + per : n_blocks;
    uint32_t sum = 0;
    for (uint32_t i = b; i < e; ++i) sum += block_count[i];
    sh[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const uint32_t o = threadIdx.x >= (unsigned)d ? sh[threadIdx.x - d] : 0;
        __syncthreads();
        sh[threadIdx.x] += o;
        __syncthreads();
    }
    uint32_t run = sh[threadIdx.x] - sum;
    for (uint32_t i = b; i < e; ++i) { const uint32_t v = block_count[i]; block_count[i] = keep ? run : 0u; run += v; }
    if (threadIdx.x == 1023) {
        block_count[n_blocks] = keep ? sh[1023] : 0u;
        // (the count, then the number of the order build it belongs to: a host that reads them in the other order and finds
        // the number it expects has the count that goes with it)
        if (blockIdx.x == 0) {
            __hip_atomic_store(wants_split, sh[1023], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(wants_split + 1, serial, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
// (the tiles at the queue's head: the long classes and the other walked tiles)
__device__ __forceinline__ uint32_t order_front_tiles(const uint32_t* __restrict__ block_count, uint32_t n_blocks) {
    uint32_t f = 0;
#pragma unroll
    for (int a = 1; a < ORDER_ARRAYS; ++a) f += block_count[a * (n_blocks + 1) + n_blocks];
    return f;
}
// order[0, n - n_split): k_render's queue (the classes above, each in natural order); split_tiles[0, n_split):
// k_render_split's tiles, whose cost words are reset to the bare flag (that kernel adds its quads' durations to them)
__global__ __launch_bounds__(256) void k_order_place(uint32_t* __restrict__ cost, uint32_t n,
                                                     const unsigned long long* __restrict__ sum, OrderRule rule, bool may_split,
                                                     const uint32_t* __restrict__ block_count, uint32_t n_blocks,
                                                     uint32_t* __restrict__ order, uint32_t* __restrict__ split_tiles,
                                                     uint32_t* __restrict__ clear_heads, unsigned long long* __restrict__ clear_sum) {
    // (clear_heads[49], never cleared: the number of long and walked tiles at the queue's head, RenderArgs::n_front)
    if (blockIdx.x == 0 && threadIdx.x == 26 && clear_heads) clear_heads[49] = order_front_tiles(block_count, n_blocks);
    if (blockIdx.x == 0 && (threadIdx.x == 27 || threadIdx.x == 28) && clear_heads) clear_heads[50 + (threadIdx.x - 27)] = 0u;      // launch_compact_queue's counters
    // (nullable: the 24 queue-head words and the cost sum the NEXT launch starts from -- not the sum read here)
    if (blockIdx.x == 0 && threadIdx.x < 16 && clear_heads) clear_heads[threadIdx.x] = 0u;
    if (blockIdx.x == 0 && threadIdx.x >= 16 && threadIdx.x < 24 && clear_heads) clear_heads[64 + 32 * (threadIdx.x - 16)] = 0u;
    if (blockIdx.x == 0 && (threadIdx.x == 24 || threadIdx.x == 25) && clear_sum) clear_sum[2 * (threadIdx.x - 24)] = 0ull;      // the sum and its walked count
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const int cls = i < n ? tile_class(cost[i], *sum, n, walked_tiles(sum, n), rule, may_split) : -1;
    __shared__ uint32_t w[ORDER_ARRAYS][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t mine = 0;          // tiles of my class before me in my wave
    uint32_t classed = 0;       // tiles of any class before me in my workgroup (wave part)
#pragma unroll
    for (int a = 0; a < ORDER_ARRAYS; ++a) {
        const unsigned long long m = __ballot(cls == a);
        if (lane == 0) w[a][wv] = (uint32_t)__popcll(m);
        if (cls == a) mine = (uint32_t)__popcll(m & below);
        classed += (uint32_t)__popcll(m & below);
    }
    __syncthreads();
    if (i >= n) return;
    if (cls >= 0) {
        uint32_t at = block_count[cls * (n_blocks + 1) + blockIdx.x] + mine;
        for (int k = 0; k < wv; ++k) at += w[cls][k];
        if (cls == 0) { split_tiles[at] = i; cost[i] = COST_SPLIT_FLAG; return; }
        for (int a = 1; a < cls; ++a) at += block_count[a * (n_blocks + 1) + n_blocks];      // the classes in front of mine
        order[at] = i;
    } else {
        // the rest: my index less the classed tiles before me
        uint32_t before = classed;
        for (int a = 0; a < ORDER_ARRAYS; ++a) {
            before += block_count[a * (n_blocks + 1) + blockIdx.x];
            for (int k = 0; k < wv; ++k) before += w[a][k];
        }
        order[order_front_tiles(block_count, n_blocks) + (i - before)] = i;
    }
}

// Slot ranges for a frame rendered as several launches over one queue (RenderArgs::seg_slots): behind the long tiles
// the queue is in natural (tile index) order, so the first slot of tile row r is found by bisection.
struct SegBounds { uint32_t row[9]; };
__global__ __launch_bounds__(64) void k_order_segments(const uint32_t* __restrict__ order, const uint32_t* __restrict__ block_count,
                                                        uint32_t n_blocks, uint32_t n_tiles, uint32_t tiles_x, SegBounds b, int n_seg,
                                                        uint32_t* __restrict__ seg_slots) {
    const int i = (int)threadIdx.x;
    if (i > n_seg) return;
    const uint32_t n_split = block_count[n_blocks], n_long = order_front_tiles(block_count, n_blocks);   // long and walked: the queue's head
    const uint32_t n_queue = n_tiles - n_split;
    if (i == 0) { seg_slots[0] = 0; return; }
    if (i == n_seg) { seg_slots[i] = n_queue; return; }
    const uint32_t first_tile = b.row[i] * tiles_x;
    uint32_t lo = n_long, hi = n_queue;                  // first slot in [n_long, n_queue) whose tile is >= first_tile
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (order[mid] < first_tile) lo = mid + 1; else hi = mid;
    }
    seg_slots[i] = lo;
}

// traceRay batch: one lane per ray, RayHit records out (OctreeLikeRayTracer.cpp:124-126, :283-288).
template <bool INLINE, bool NEAREST>
__global__ __launch_bounds__(256) void k_trace(SceneDev sc, const float* __restrict__ origins,
                                               const float* __restrict__ dirs, uint32_t n,
                                               hvrt_hit* __restrict__ out, unsigned long long* counters) {
    __shared__ float4 lds_win[4][64];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    uint64_t nv = 0, nt = 0;
    V3 o = { 0.0f, 0.0f, 0.0f }, d = { 0.0f, 0.0f, 0.0f };
    if (i < n) {
        o = { origins[3ull * i], origins[3ull * i + 1], origins[3ull * i + 2] };
        d = { dirs[3ull * i], dirs[3ull * i + 1], dirs[3ull * i + 2] };
    }
    TraceResult r;
    r.tri = HVRT_MISS; r.t = FLT_MAX; r.nv = 0; r.nt = 0;
    if constexpr (INLINE) r = trace_ray_wave<true, NEAREST, false>(sc, o, d, i < n, lds_win[threadIdx.x >> 6]);    // all 64 lanes take part
    else if (i < n) r = trace_ray<true, INLINE, NEAREST>(sc, o, d);
    if (i < n) {
        hvrt_hit h;
        h.hit = r.tri != HVRT_MISS;
        h.t = r.t;
        h.point[0] = h.point[1] = h.point[2] = 0.0f;
        h.origin[0] = h.origin[1] = h.origin[2] = 0.0f;   // the reference leaves origin unset on a miss; we zero it
        h.normal[0] = h.normal[1] = h.normal[2] = 0.0f;
        h.triangle_index = 0;
        h.voxel_index = 0;
        if (h.hit) {
            const float4 r0 = sc.tri_rec[3ull * r.tri], r2 = sc.tri_rec[3ull * r.tri + 2];
            h.point[0] = r0.x; h.point[1] = r0.y; h.point[2] = r0.z;
            h.origin[0] = o.x; h.origin[1] = o.y; h.origin[2] = o.z;
            h.normal[0] = r2.y; h.normal[1] = r2.z; h.normal[2] = r2.w;
            h.triangle_index = r.tri;
        }
        out[i] = h;
        nv = r.nv; nt = r.nt;
    }
    if (counters) {
        nv = wave_sum64(nv);
        nt = wave_sum64(nt);
        if ((threadIdx.x & 63) == 0 && (nv | nt)) {
            atomicAdd(&counters[0], (unsigned long long)nv);
            atomicAdd(&counters[1], (unsigned long long)nt);
        }
    }
}

// ImageSaver.h:18-75: byte = static_cast<uint8_t>(std::clamp(c * 255.0f, 0.0f, 255.0f)) -- truncation; a NaN
// passes std::clamp unchanged and its conversion is undefined there (0 on the reference's platform): 0 here.
__device__ __forceinline__ uint32_t quantise_channel(float c) {
    const float v = c * 255.0f;
    const float k = v < 0.0f ? 0.0f : (255.0f < v ? 255.0f : v);
    return k == k ? (uint32_t)(int)k : 0u;
}
__global__ __launch_bounds__(256) void k_quantise_rgb8(const float* __restrict__ rgb, uint64_t n, uint8_t* __restrict__ out) {
    const uint64_t i = ((uint64_t)blockIdx.x * 256u + threadIdx.x) * 4u;     // four channels -> one dword
    if (i + 4 <= n) {
        const float4 v = *reinterpret_cast<const float4*>(rgb + i);
        *reinterpret_cast<uint32_t*>(out + i) = quantise_channel(v.x) | (quantise_channel(v.y) << 8) |
                                                (quantise_channel(v.z) << 16) | (quantise_channel(v.w) << 24);
    } else {
        for (uint64_t k = i; k < n; ++k) out[k] = (uint8_t)quantise_channel(rgb[k]);
    }
}

// frame[y][x] <- parts[part(y)][local_row(y)][x], 4-byte words
__global__ __launch_bounds__(256) void k_assemble(const uint32_t* __restrict__ parts, int words_per_row, int height,
                                                  int tile_rows, int n_parts, int slab_rows,
                                                  uint32_t* __restrict__ frame) {
    const int y = blockIdx.y;
    const int k = y / tile_rows;
    const int part = k % n_parts;
    const int lr = (k / n_parts) * tile_rows + (y - k * tile_rows);
    const uint32_t* src = parts + ((size_t)part * slab_rows + lr) * words_per_row;
    uint32_t* dst = frame + (size_t)y * words_per_row;
    for (int w = blockIdx.x * 256 + threadIdx.x; w < words_per_row; w += gridDim.x * 256) dst[w] = src[w];
    (void)height;
}

}  // namespace

uint64_t render_tile_count(int width, int local_rows, int n_frames) {
    const uint64_t tiles_x = (uint64_t)(width + 7) >> 3, tiles_y = (uint64_t)(local_rows + 7) >> 3;
    return tiles_x * tiles_y * (uint64_t)n_frames;
}

hipError_t launch_fill_background(const RenderArgs& ra, int n_cus, hipStream_t s) {
    const uint64_t n_tiles = render_tile_count(ra.width, ra.local_rows, ra.n_frames);
    if (n_tiles == 0 || !ra.cover) return hipSuccess;
    const uint64_t units = (uint64_t)((((ra.width + 7) >> 3) + 7) >> 3) * (uint64_t)((ra.local_rows + 7) >> 3) * (uint64_t)ra.n_frames;
    uint64_t blocks = (units + 3) / 4;
    if (blocks > (uint64_t)n_cus * 4u) blocks = (uint64_t)n_cus * 4u;      // four waves per SIMD at most
    k_fill_background<<<(unsigned)blocks, 256, 0, s>>>(ra);
    return hipGetLastError();
}
hipError_t launch_compact_queue(const RenderArgs& ra, const uint32_t* order_in, const uint32_t* n_front_in, uint32_t* block_marked,
                                uint32_t* queue, uint32_t* counters, hipStream_t s) {
    const uint64_t n_tiles = render_tile_count(ra.width, ra.local_rows, ra.n_frames);
    if (n_tiles == 0 || !ra.cover || !block_marked) return hipErrorInvalidValue;
    const unsigned nb = (unsigned)((n_tiles + 255) / 256);
    k_compact_count<<<nb, 256, 0, s>>>(ra, order_in, n_front_in, block_marked);
    const bool scan = nb >= COMPACT_SCAN_FROM;
    if (scan) k_compact_scan<<<2, 1024, 0, s>>>(block_marked, nb);
    k_compact_queue<<<nb, 256, 0, s>>>(ra, order_in, n_front_in, block_marked, scan, queue, counters);
    return hipGetLastError();
}

hipError_t launch_render(const SceneDev& sc, const RenderArgs& ra, int n_cus, bool abortable, hipStream_t s) {
    const uint64_t n_tiles = render_tile_count(ra.width, ra.local_rows, ra.n_frames);
    if (n_tiles == 0) return hipSuccess;
    uint64_t blocks = (n_tiles + RENDER_BLOCK_WAVES - 1) / RENDER_BLOCK_WAVES;
    const uint64_t cap = (uint64_t)n_cus * 4u * HVRT_RENDER_WAVES / RENDER_BLOCK_WAVES;   // workgroups that fill the CUs' wave slots
    if (blocks > cap) blocks = cap;
    const bool il = sc.ref_rec != nullptr;
    const unsigned nb = (unsigned)blocks;
#define HVRT_LAUNCH3(C, I, N) do { if (abortable) k_render<C, I, N, true><<<nb, HVRT_RENDER_BLOCK, 0, s>>>(sc, ra); \
                                   else if (ra.seg_slots) k_render_segment<C, I, N><<<nb, HVRT_RENDER_BLOCK, 0, s>>>(sc, ra); \
                                   else k_render<C, I, N, false><<<nb, HVRT_RENDER_BLOCK, 0, s>>>(sc, ra); } while (0)
#define HVRT_LAUNCH(C, I) do { if (ra.nearest) HVRT_LAUNCH3(C, I, true); else HVRT_LAUNCH3(C, I, false); } while (0)
    if (ra.counters) { if (il) HVRT_LAUNCH(true, true); else HVRT_LAUNCH(true, false); }
    else             { if (il) HVRT_LAUNCH(false, true); else HVRT_LAUNCH(false, false); }
#undef HVRT_LAUNCH
#undef HVRT_LAUNCH3
    return hipGetLastError();
}

// k_render and k_render_split become eligible at the same moment (an event on k_render's stream releases the other).
// Launched one by one with the host waiting in between, the split kernel happened to reach the GPU some 20 us ahead:
// its workgroups took their places, those without work left at once, and k_render's filled in around the rest.  Queued
// back to back, k_render wins the race by 6 us, its persistent waves take every wave slot, and the split kernel only
// gets on when they retire -- it then ended 130 us AFTER k_render (config 2: 1.78 ms per launch instead of 1.65).  This
// wave holds k_render's stream until every workgroup of the split kernel has reported in (or timeout_us have passed --
// an exit every run reaches).
__global__ __launch_bounds__(64) void k_wait_split_started(uint32_t* started, uint32_t target, uint32_t timeout_ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    while ((int)(atomicAdd(started, 0u) - target) < 0) {
        if (wall_clock64() - t0 > (unsigned long long)timeout_ticks) break;
        __builtin_amdgcn_s_sleep(16);
    }
}
hipError_t launch_wait_split_started(uint32_t* started, uint32_t target, uint32_t timeout_us, hipStream_t s) {
    k_wait_split_started<<<1, 64, 0, s>>>(started, target, timeout_us * 100u);      // wall_clock64 ticks at 100 MHz
    return hipGetLastError();
}

hipError_t launch_render_split(const SceneDev& sc, const RenderArgs& ra, int n_cus, hipStream_t s, uint32_t* n_workgroups,
                               uint32_t expected_tiles, bool alone, int rays) {
    if (n_workgroups) *n_workgroups = 0;
    const uint64_t n_tiles = render_tile_count(ra.width, ra.local_rows, ra.n_frames);
    if (n_tiles == 0 || !ra.order) return hipSuccess;
    const uint64_t WG_PER_TILE = rays == 8 ? 2u : 4u;      // 64 / rays waves per tile, four waves per workgroup
    uint64_t blocks = n_tiles * WG_PER_TILE;               // were every tile split
    // (workgroups per CU at most: 2 / 3 / 4 / 6: config 2 0.63 / 0.51 / 0.49 / 0.51 ms, config 3 0.97 / 0.93 / 0.87 / 0.86,
    // config 4 8.8 throughout)
#ifndef HVRT_SPLIT_WGS
#define HVRT_SPLIT_WGS 4
#endif
    const uint64_t cap = (uint64_t)n_cus * (alone ? (uint64_t)HVRT_SPLIT_WAVES : (uint64_t)(rays == 8 ? HVRT_SPLIT_WGS : 4));
    if (blocks > cap) blocks = cap;
    // (the caller's idea of how many tiles there are, when it has one: twice the workgroups they need, 16 at least --
    // the items are dealt round-robin, so a count that was too low costs time, not tiles)
    if (expected_tiles && (uint64_t)expected_tiles * 2u * WG_PER_TILE < blocks) blocks = expected_tiles * 2u * WG_PER_TILE < 16u ? 16u : expected_tiles * 2u * WG_PER_TILE;
    const bool il = sc.ref_rec != nullptr;
    const unsigned nb = (unsigned)blocks;
    if (n_workgroups) *n_workgroups = nb;
#define HVRT_LAUNCH(C, I, N) do { if (rays == 8) k_render_split<C, I, N, 8><<<nb, 256, 0, s>>>(sc, ra); else k_render_split<C, I, N, 4><<<nb, 256, 0, s>>>(sc, ra); } while (0)
    if (ra.nearest) {
        if (ra.counters) { if (il) HVRT_LAUNCH(true, true, true); else HVRT_LAUNCH(true, false, true); }
        else             { if (il) HVRT_LAUNCH(false, true, true); else HVRT_LAUNCH(false, false, true); }
    } else {
        if (ra.counters) { if (il) HVRT_LAUNCH(true, true, false); else HVRT_LAUNCH(true, false, false); }
        else             { if (il) HVRT_LAUNCH(false, true, false); else HVRT_LAUNCH(false, false, false); }
    }
#undef HVRT_LAUNCH
    return hipGetLastError();
}

// Warm-up of a hardware queue (hvrt_create): the runtime gives a queue its scratch memory when a dispatch first needs
// some, and again when a later dispatch needs more -- 4 to 5 ms each time, which the first render() of a process paid
// for its counting kernel and the second for the variant it runs.  One full-size dispatch of a kernel that needs more
// scratch per lane than any traversal kernel (176 bytes against at most 116) settles that per queue, and no k_render
// launch of a profiled run is an empty one.  (Launching the traversal kernels themselves empty brought nothing more.)
__global__ __launch_bounds__(256) void k_warm_scratch(uint32_t* sink, uint32_t n) {
    volatile uint32_t spill[40];
    for (uint32_t i = 0; i < 40; ++i) spill[i] = i * n + threadIdx.x;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < 40; ++i) acc += spill[(i * 7u + n) % 40u];
    if (n == 0xFFFFFFFFu) sink[0] = acc;        // never: n is the grid size
}
hipError_t launch_scratch_warmup(uint32_t* sink, int n_cus, hipStream_t s) {
    const unsigned nb = (unsigned)n_cus * 8u;
    k_warm_scratch<<<nb, 256, 0, s>>>(sink, nb);
    return hipGetLastError();
}

// what a launch's queue must find zero (RenderArgs::tile_counter's word map), where no order build has cleared it: one
// launch instead of four memsets (5 us each on the stream, in front of every launch without history)
__global__ __launch_bounds__(64) void k_queue_reset(uint32_t* __restrict__ tile_counter, unsigned long long* __restrict__ cost_sum,
                                                   unsigned long long* __restrict__ walked) {
    const uint32_t i = threadIdx.x;
    if (i < 16u) tile_counter[i] = 0u;
    else if (i < 24u) tile_counter[64u + 32u * (i - 16u)] = 0u;
    else if (i == 24u) *cost_sum = 0ull;
    else if (i == 25u && walked) *walked = 0ull;
}
hipError_t launch_queue_reset(uint32_t* tile_counter, unsigned long long* cost_sum, unsigned long long* walked, hipStream_t s) {
    k_queue_reset<<<1, 64, 0, s>>>(tile_counter, cost_sum, walked);
    return hipGetLastError();
}

hipError_t launch_order_build(uint32_t* tile_cost, uint32_t n_tiles, const unsigned long long* cost_sum, int n_cus,
                              bool may_split, uint32_t* block_count, uint32_t* order, uint32_t* split_tiles,
                              uint32_t* wants_split, uint32_t serial, uint32_t* clear_heads, unsigned long long* clear_sum, bool by_sample, hipStream_t s) {
    const unsigned blocks = (n_tiles + 255u) / 256u;
    OrderRule rule;
    rule.slots = (uint32_t)n_cus * 4u * (uint32_t)HVRT_RENDER_WAVES;     // the waves k_render keeps resident
    // (tiles rendered sample by sample cost no more than in one piece: the bar is lower)
    // (config 4, the bar at 1 / 2 / 4 / 8 times the slots: 11.8 / 10.6 / 11.0 / 11.1 ms)
    if (by_sample) rule.slots *= 2u;
    rule.split_x10 = SPLIT_MEAN_X10;
    // the long classes: from the split rule's bar down to the mean walked tile, in equal ratios (178, 158, 141, 125, 112, 100
    // hundredths).  Configs 2 / 3 / 4 with the classes' lower end at 0.6 / 0.8 / 1.0 mean walked tiles:
    // 0.55 / 0.53 / 0.53, 0.69 / 0.69 / 0.68 and 9.7 / 9.5 / 9.4 ms; with ONE long class (more than 2.5 mean walked
    // tiles, none once the split rule's bar is there) 0.68 / 0.89 / 10.9
    for (int j = 0; j < ORDER_BINS; ++j)
        rule.bin_x100[j] = (uint32_t)(100.0 * pow((double)SPLIT_MEAN_X10 / 10.0, (double)(ORDER_BINS - 1 - j) / (double)ORDER_BINS));
    may_split = may_split && n_tiles <= 0x0FFFFFFFu;       // 16 items per tile in a 32-bit count
    k_order_count<<<blocks, 256, 0, s>>>(tile_cost, n_tiles, cost_sum, rule, may_split, block_count);
    k_order_scan<<<ORDER_ARRAYS, 1024, 0, s>>>(block_count, blocks, may_split, wants_split, serial);
    k_order_place<<<blocks, 256, 0, s>>>(tile_cost, n_tiles, cost_sum, rule, may_split, block_count, blocks, order, split_tiles, clear_heads, clear_sum);
    return hipGetLastError();
}

hipError_t launch_order_segments(const uint32_t* order, const uint32_t* block_count, uint32_t n_blocks, uint32_t n_tiles,
                                 uint32_t tiles_x, const uint32_t* row_bounds, int n_seg, uint32_t* seg_slots, hipStream_t s) {
    if (n_seg < 1 || n_seg > 8) return hipErrorInvalidValue;
    SegBounds b;
    for (int i = 0; i <= n_seg; ++i) b.row[i] = row_bounds[i];
    k_order_segments<<<1, 64, 0, s>>>(order, block_count, n_blocks, n_tiles, tiles_x, b, n_seg, seg_slots);
    return hipGetLastError();
}

hipError_t launch_trace(const SceneDev& sc, const float* origins, const float* dirs, uint32_t n, hvrt_hit* out,
                        unsigned long long* counters, int nearest, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const unsigned nb = (n + 255) / 256;
    if (nearest) {
        if (sc.ref_rec) k_trace<true, true><<<nb, 256, 0, s>>>(sc, origins, dirs, n, out, counters);
        else k_trace<false, true><<<nb, 256, 0, s>>>(sc, origins, dirs, n, out, counters);
    } else {
        if (sc.ref_rec) k_trace<true, false><<<nb, 256, 0, s>>>(sc, origins, dirs, n, out, counters);
        else k_trace<false, false><<<nb, 256, 0, s>>>(sc, origins, dirs, n, out, counters);
    }
    return hipGetLastError();
}

hipError_t launch_quantise_rgb8(const float* rgb, uint64_t n_values, uint8_t* out, hipStream_t s) {
    if (n_values == 0) return hipSuccess;
    k_quantise_rgb8<<<(unsigned)((n_values + 1023) / 1024), 256, 0, s>>>(rgb, n_values, out);
    return hipGetLastError();
}

hipError_t launch_assemble(const void* parts, int width, int height, int tile_rows, int n_parts, int slab_rows,
                           int bytes_per_pixel, void* frame, hipStream_t s) {
    const int words = (int)(((int64_t)width * bytes_per_pixel) / 4);
    if (words == 0 || height == 0) return hipSuccess;
    int bx = (words + 255) / 256;
    if (bx > 64) bx = 64;
    k_assemble<<<dim3(bx, height), 256, 0, s>>>((const uint32_t*)parts, words, height, tile_rows, n_parts, slab_rows,
                                                (uint32_t*)frame);
    return hipGetLastError();
}

}  // namespace hvrt
