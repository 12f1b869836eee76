// hvrt_build.hip -- voxel-grid build kernels for gfx950 (MI355X).
//
// Replaces the reference's voxelizeTrianglesDirect (SceneCreator.cpp:197-281) and its dead GPU /
// inline variants (SceneCreator.cu:67-358, InlineVoxelizer.h:17-167) with:
//   k_count       one lane per triangle: AABB -> inclusive voxel range (SceneCreator.cpp:23-37,
//                 177-188, GeometryUtils.cpp:43-51), atomic count per voxel with run-aggregated
//                 atomics; big ranges are spread over the whole wave
//   k_scan        exclusive scan of the counts in flat order ix + nx*(iy + ny*iz) in ONE pass over
//                 the array (chained scan with decoupled look-back, 32768 counters per workgroup);
//                 also leaves the one-bit-per-voxel occupancy map for the distance field and the exact
//                 64-bit number of references in stats[0] (the offsets themselves are 32-bit: the host,
//                 or the kernels behind the scan, refuse to scatter when that number does not fit)
//   k_fill        scatter of triangle ids through atomic cursors (the scanned counts themselves:
//                 after the fill cnt[v] == start(v+1), i.e. the buffer IS the CSR offsets array
//                 shifted by one, so no separate cursor array and no copy)
//   k_rank_sort   per-voxel ascending order (the reference appends in ascending triangle order,
//                 SceneCreator.cpp:212-230): an entry's final position is the number of smaller ids
//                 in its list (lists of up to 96 entries); longer lists are queued for
//   k_sort_long   an all-ascending bitonic network in LDS (<= 4096 and <= 32768 entries) or in
//                 global memory beyond that
//   k_dist_*, k_face_bits, k_expand_records   derived acceleration data of the traversal (empty-space
//                 distance field, "also listed in that face neighbour" bits, per-reference records)
//   k_csr_stats, k_check_indices   statistics and validation of a caller-supplied CSR
// All integer / byte work; HBM-bound; no MFMA.  Compile with -ffp-contract=off.
#include "hvrt_internal.hpp"

#include <float.h>
#include <limits.h>

namespace hvrt {
namespace {

// static_cast<int>(float) with x86 cvttss2si semantics (NaN / out of range -> INT_MIN): the
// reference's platform behaviour, spelled out because v_cvt_i32_f32 saturates instead.
__device__ __forceinline__ int f2i(float f) {
    return (!(f < 2147483648.0f) || f < -2147483648.0f) ? INT_MIN : (int)f;
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
// std::min / std::max as libstdc++ / MSVC define them (GeometryUtils.cpp:44-50 nests them).
__device__ __forceinline__ float min2(float a, float b) { return b < a ? b : a; }
__device__ __forceinline__ float max2(float a, float b) { return a < b ? b : a; }

__device__ __forceinline__ uint64_t flat_index(const GridParams& g, int ix, int iy, int iz) {
    return (uint64_t)ix + (uint64_t)g.nx * ((uint64_t)iy + (uint64_t)g.ny * (uint64_t)iz);
}

// The number of CSR entries as the kernels behind the scan learn it (RefCount, hvrt_internal.hpp): either the host
// read it back and sized the arrays (live == nullptr), or the build goes on without that round trip and every kernel
// reads the scan's 64-bit total itself -- and does nothing when the arrays it would write (sized by an earlier
// build) are too small; the host sees the same number at the end of the build and repeats the tail with new arrays.
__device__ __forceinline__ bool ref_count(const RefCount& r, uint64_t& n) {
    n = r.live ? r.live[0] : r.n;
    // live[7]: the count pass's independent total (k_scan_stats); differs when a 16-bit counter carried
    return n <= r.cap && (!r.live || r.live[7] == n);
}
__device__ __forceinline__ uint64_t face_words_of(uint64_t n_refs) { return (n_refs + 63) / 64 * 2 + 2; }

// ------------------------------------------------------------------------------------------
// triangle / box overlap by separating axes (hvrt_set_overlap_test(ctx, 1); an extension)
// ------------------------------------------------------------------------------------------
// The reference lists a triangle in every voxel of its bounding box (SceneCreator.cpp:197-281; triangleIntersectsBox
// is declared in GeometryUtils.h:19 and never defined).  With the SAT rule a voxel of that range keeps the triangle
// only if the two really overlap: Akenine-Moller's test -- the box's three normals, the triangle's normal, the nine
// products of an edge with a coordinate axis.  The box is the voxel grown by 2^-10 of its half size, so that rounding
// can add a voxel to a list but not drop one; every compare that involves a NaN is false, i.e. "not separated".
// Plain fp32 operations in a fixed order (no contraction), so that a CPU restatement gives the same lists bit for bit.
// Geometry outside the grid overlaps no voxel: where the reference's clamped indices put it into the border voxels
// (a non-cubic grid under a round mesh), this rule lists it nowhere.
struct Tri9 { float ax, ay, az, bx, by, bz, cx, cy, cz; };
__device__ __forceinline__ float abs1(float x) { return x < 0.0f ? -x : x; }
__device__ __forceinline__ bool separated(float p0, float p1, float p2, float r) {
    const float lo = min2(min2(p0, p1), p2), hi = max2(max2(p0, p1), p2);
    return lo > r || hi < -r;
}
__device__ __forceinline__ bool edge_separates(float ex, float ey, float ez, float hx, float hy, float hz,
                                               float ax, float ay, float az, float bx, float by, float bz,
                                               float cx, float cy, float cz) {
    const float fx = abs1(ex), fy = abs1(ey), fz = abs1(ez);
    // e x (1,0,0) = (0, ez, -ey) up to sign: p = ez * y - ey * z
    if (separated(ez * ay - ey * az, ez * by - ey * bz, ez * cy - ey * cz, hy * fz + hz * fy)) return true;
    // e x (0,1,0): p = ex * z - ez * x
    if (separated(ex * az - ez * ax, ex * bz - ez * bx, ex * cz - ez * cx, hx * fz + hz * fx)) return true;
    // e x (0,0,1): p = ey * x - ex * y
    if (separated(ey * ax - ex * ay, ey * bx - ex * by, ey * cx - ex * cy, hx * fy + hy * fx)) return true;
    return false;
}
__device__ __forceinline__ bool tri_box_overlap(const GridParams& g, const Tri9& t, int ix, int iy, int iz) {
    const float GROW = 1.0009765625f;       // 1 + 2^-10
    const float cx = g.mnx + ((float)ix + 0.5f) * g.vsx, cy = g.mny + ((float)iy + 0.5f) * g.vsy, cz = g.mnz + ((float)iz + 0.5f) * g.vsz;
    const float hx = (0.5f * g.vsx) * GROW, hy = (0.5f * g.vsy) * GROW, hz = (0.5f * g.vsz) * GROW;
    const float ax = t.ax - cx, ay = t.ay - cy, az = t.az - cz;
    const float bx = t.bx - cx, by = t.by - cy, bz = t.bz - cz;
    const float dx = t.cx - cx, dy = t.cy - cy, dz = t.cz - cz;
    if (separated(ax, bx, dx, hx) || separated(ay, by, dy, hy) || separated(az, bz, dz, hz)) return false;
    const float e0x = bx - ax, e0y = by - ay, e0z = bz - az;
    const float e1x = dx - bx, e1y = dy - by, e1z = dz - bz;
    const float e2x = ax - dx, e2y = ay - dy, e2z = az - dz;
    if (edge_separates(e0x, e0y, e0z, hx, hy, hz, ax, ay, az, bx, by, bz, dx, dy, dz)) return false;
    if (edge_separates(e1x, e1y, e1z, hx, hy, hz, ax, ay, az, bx, by, bz, dx, dy, dz)) return false;
    if (edge_separates(e2x, e2y, e2z, hx, hy, hz, ax, ay, az, bx, by, bz, dx, dy, dz)) return false;
    const float nx = e0y * e1z - e0z * e1y, ny = e0z * e1x - e0x * e1z, nz = e0x * e1y - e0y * e1x;
    const float d = (nx * ax + ny * ay) + nz * az;
    const float r = (hx * abs1(nx) + hy * abs1(ny)) + hz * abs1(nz);
    if (d > r || d < -r) return false;
    return true;
}

// ------------------------------------------------------------------------------------------
// records: v0, e1, e2, normal
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prepare_records(const float4* __restrict__ aos,
                                                         float4* __restrict__ rec, uint32_t T) {
    uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= T) return;
    float4 a = aos[3ull * t], b = aos[3ull * t + 1], c = aos[3ull * t + 2];
    // a = v0.xyz v1.x | b = v1.yz v2.xy | c = v2.z n.xyz
    float e1x = a.w - a.x, e1y = b.x - a.y, e1z = b.y - a.z;   // subtract(v1, v0), GeometryUtils.cpp:66
    float e2x = b.z - a.x, e2y = b.w - a.y, e2z = c.x - a.z;   // subtract(v2, v0), GeometryUtils.cpp:67
    rec[3ull * t]     = make_float4(a.x, a.y, a.z, e1x);
    rec[3ull * t + 1] = make_float4(e1y, e1z, e2x, e2y);
    rec[3ull * t + 2] = make_float4(e2z, c.y, c.z, c.w);
}

// ------------------------------------------------------------------------------------------
// count / fill
// ------------------------------------------------------------------------------------------
constexpr uint32_t SMALL_VOL = 32;  // ranges up to this many voxels are walked by the owning lane

// what a visit to a voxel does: count into 32-bit counters, count into 16-bit counters (two per word; a list of
// 65536 or more entries carries into its neighbour -- the build notices, see k_scan_stats, and counts again in 32
// bits), or take the next slot of the voxel's list (the cursors are the scanned counters, always 32-bit)
enum { COUNT32 = 0, FILL = 1, COUNT16 = 2 };

template <int MODE>
__device__ __forceinline__ uint32_t bump(uint32_t* cnt, uint64_t f, uint32_t by) {
    if (MODE == COUNT16) { atomicAdd(&cnt[f >> 1], by << (16u * (uint32_t)(f & 1ull))); return 0u; }
    if (MODE == COUNT32) { atomicAdd(&cnt[f], by); return 0u; }
    return atomicAdd(&cnt[f], by);
}

template <int MODE>
__device__ __forceinline__ void touch(uint32_t* cnt, uint32_t* indices, uint32_t* vox_of, uint64_t f, uint32_t t) {
    const uint32_t p = bump<MODE>(cnt, f, 1u);
    if (MODE == FILL) {
        indices[p] = t;
        vox_of[p] = (uint32_t)f;
    }
}

// One voxel per lane, touched with wave-aggregated atomics: mesh order is spatial order more often than not,
// so neighbouring lanes tend to name the same voxel.  Each run of adjacent active lanes with equal voxel
// issues ONE atomicAdd of the run length (its first lane), and in the fill the others take consecutive
// slots behind it.  A dense cluster of small triangles (thousands per voxel) otherwise serialises on a
// handful of L2 atomic units.  Every lane of the wave must call this (ballot / shuffle inside).
template <int MODE>
__device__ __forceinline__ void touch_runs(uint32_t* cnt, uint32_t* indices, uint32_t* vox_of, bool act, uint32_t f, uint32_t t) {
    const int lane = threadIdx.x & 63;
    const unsigned long long S = __ballot(act);
    if (S == 0) return;
    // the word a lane adds to: the voxel's counter -- or, with 16-bit counters, the word that holds it and its x
    // neighbour's, so that a run of lanes walking along x is still ONE atomic (two different voxels of one word
    // would otherwise be two atomics on the same address, which the L2 serialises)
    const uint32_t key = MODE == COUNT16 ? f >> 1 : f;
    const uint32_t kprev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x138, 0xF, 0xF, false);      // wave_shr:1 (lane - 1's key; lane 0 does not look)
    const bool prev_act = lane > 0 && ((S >> (lane - 1)) & 1ull);
    const unsigned long long H = __ballot(act && (!prev_act || kprev != key));
    // my run's first lane: the highest head at or below me
    const unsigned long long below = H & (~0ull >> (63 - lane));
    const int hl = act ? 63 - __builtin_clzll(below) : lane;
    const unsigned long long odd = MODE == COUNT16 ? __ballot(act && (f & 1u)) : 0ull;
    uint32_t base = 0;
    if (act && hl == lane) {
        const unsigned long long stop = (H | ~S) & ~(~0ull >> (63 - lane));   // next head or inactive lane above me
        const int end = stop ? __builtin_ctzll(stop) : 64;
        const uint32_t len = (uint32_t)(end - lane);
        if (MODE == COUNT16) {
            const unsigned long long run = (end == 64 ? ~0ull : ((1ull << end) - 1ull)) & ~((1ull << lane) - 1ull);
            const uint32_t hi = (uint32_t)__builtin_popcountll(odd & run);
            atomicAdd(&cnt[key], (len - hi) | (hi << 16));
        } else {
            base = bump<MODE>(cnt, f, len);
        }
    }
    if (MODE == FILL) {
        base = __shfl(base, hl);
        if (act) {
            const uint32_t p = base + (uint32_t)(lane - hl);
            indices[p] = t;
            vox_of[p] = f;
        }
    }
}

// SAT: `mask` carries the verdicts on a small range (bit k: the k-th voxel of the walk keeps the triangle) from the count
// pass, which works them out, to the scatter, which then needs neither the triangle nor the test again.
template <int MODE, bool SAT = false>
__device__ __forceinline__ void walk_range(const GridParams& g, uint32_t* cnt, uint32_t* indices, uint32_t* vox_of, uint32_t t,
                                           bool valid, int lx, int ly, int lz, int hx, int hy, int hz, const Tri9& tri, uint32_t& mask) {
    const uint32_t dx = valid ? (uint32_t)(hx - lx + 1) : 0u;
    const uint32_t dy = valid ? (uint32_t)(hy - ly + 1) : 0u;
    const uint32_t dz = valid ? (uint32_t)(hz - lz + 1) : 0u;
    const uint64_t vol = (uint64_t)dx * dy * dz;
    const bool big = vol > SMALL_VOL;
    // Small ranges: the wave steps through "k-th voxel of my range" together (x fastest, like the
    // reference's loops) until the longest range of the wave is done.
    {
        const bool small = valid && !big;
        int ix = lx, iy = ly, iz = lz;
        bool act = small;
        uint32_t step = 0;          // < SMALL_VOL = 32 while act
        while (__ballot(act)) {
            bool listed = act;
            if (SAT && act) {
                if (MODE == FILL) listed = (mask >> step) & 1u;
                else { listed = tri_box_overlap(g, tri, ix, iy, iz); mask |= (listed ? 1u : 0u) << step; }
            }
            ++step;
            touch_runs<MODE>(cnt, indices, vox_of, listed, (uint32_t)flat_index(g, ix, iy, iz), t);
            if (act) {
                if (++ix > hx) { ix = lx; if (++iy > hy) { iy = ly; if (++iz > hz) act = false; } }
            }
        }
    }
    // Triangles that span many voxels: the whole wave walks each one (every lane reaches here).
    unsigned long long m = __ballot(big);
    const int lane = threadIdx.x & 63;
    while (m) {
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1;
        const int blx = __shfl(lx, src), bly = __shfl(ly, src), blz = __shfl(lz, src);
        const uint32_t bdx = __shfl(dx, src), bdy = __shfl(dy, src), bdz = __shfl(dz, src);
        const uint32_t bt = __shfl(t, src);
        Tri9 btri = tri;
        if (SAT) {
            btri.ax = __shfl(tri.ax, src); btri.ay = __shfl(tri.ay, src); btri.az = __shfl(tri.az, src);
            btri.bx = __shfl(tri.bx, src); btri.by = __shfl(tri.by, src); btri.bz = __shfl(tri.bz, src);
            btri.cx = __shfl(tri.cx, src); btri.cy = __shfl(tri.cy, src); btri.cz = __shfl(tri.cz, src);
        }
        const uint64_t bvol = (uint64_t)bdx * bdy * bdz;
        const uint64_t plane = (uint64_t)bdx * bdy;
        for (uint64_t i = lane; i < bvol; i += 64) {
            const uint32_t iz = (uint32_t)(i / plane);
            const uint32_t r = (uint32_t)(i - (uint64_t)iz * plane);
            const uint32_t iy = r / bdx, ix = r - iy * bdx;
            if (SAT && !tri_box_overlap(g, btri, blx + (int)ix, bly + (int)iy, blz + (int)iz)) continue;
            touch<MODE>(cnt, indices, vox_of, flat_index(g, blx + (int)ix, bly + (int)iy, blz + (int)iz), bt);
        }
    }
}

__device__ __forceinline__ uint64_t wave_sum64(uint64_t v);

// vol_sums (COUNT16 only): 64 words a cache line apart; every workgroup adds the number of references its triangles
// make (the sum of their range volumes, exact in 64 bits) to one of them -- the independent total that tells a
// carried 16-bit counter from a sound one
template <int MODE, bool SAT = false>
__global__ __launch_bounds__(256) void k_count(const float4* __restrict__ aos, uint32_t T, GridParams g,
                                               uint32_t* __restrict__ cnt, uint32_t* __restrict__ ranges,
                                               uint64_t* __restrict__ vol_sums, uint32_t* __restrict__ sat_mask) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const bool valid = t < T;
    int lx = 0, ly = 0, lz = 0, hx = -1, hy = -1, hz = -1;
    Tri9 tri = {};
    if (valid) {
        float4 a = aos[3ull * t], b = aos[3ull * t + 1], c = aos[3ull * t + 2];
        if (SAT) tri = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x };
        // computeTriangleMinMax, GeometryUtils.cpp:43-51
        float mnx = min2(min2(a.x, a.w), b.z), mny = min2(min2(a.y, b.x), b.w), mnz = min2(min2(a.z, b.y), c.x);
        float mxx = max2(max2(a.x, a.w), b.z), mxy = max2(max2(a.y, b.x), b.w), mxz = max2(max2(a.z, b.y), c.x);
        // worldToVoxelIndex, SceneCreator.cpp:23-37: IEEE divide, truncate, clamp
        lx = clampi(f2i((mnx - g.mnx) / g.vsx), 0, g.nx - 1);
        ly = clampi(f2i((mny - g.mny) / g.vsy), 0, g.ny - 1);
        lz = clampi(f2i((mnz - g.mnz) / g.vsz), 0, g.nz - 1);
        hx = clampi(f2i((mxx - g.mnx) / g.vsx), 0, g.nx - 1);
        hy = clampi(f2i((mxy - g.mny) / g.vsy), 0, g.ny - 1);
        hz = clampi(f2i((mxz - g.mnz) / g.vsz), 0, g.nz - 1);
        // A NaN corner can make lo > hi: the reference's loops then run zero times; keep that.
        ranges[3ull * t]     = (uint32_t)lx | ((uint32_t)hx << 16);
        ranges[3ull * t + 1] = (uint32_t)ly | ((uint32_t)hy << 16);
        ranges[3ull * t + 2] = (uint32_t)lz | ((uint32_t)hz << 16);
    }
    const bool nonempty = valid && lx <= hx && ly <= hy && lz <= hz;
    if (MODE == COUNT16) {
        __shared__ uint64_t s_vol[4];
        const uint64_t vol = wave_sum64(nonempty ? (uint64_t)(hx - lx + 1) * (uint64_t)(hy - ly + 1) * (uint64_t)(hz - lz + 1) : 0ull);
        if ((threadIdx.x & 63u) == 0u) s_vol[threadIdx.x >> 6] = vol;
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint64_t all = s_vol[0] + s_vol[1] + s_vol[2] + s_vol[3];
            if (all) atomicAdd((unsigned long long*)&vol_sums[16u * (blockIdx.x & 63u)], (unsigned long long)all);
        }
    }
    uint32_t mask = 0;
    walk_range<MODE, SAT>(g, cnt, nullptr, nullptr, t, nonempty, lx, ly, lz, hx, hy, hz, tri, mask);
    if (SAT && valid) sat_mask[t] = mask;
}

template <bool SAT>
__global__ __launch_bounds__(256) void k_fill(const uint32_t* __restrict__ ranges, const float4* __restrict__ aos, uint32_t T, GridParams g,
                                              uint32_t* __restrict__ cnt, uint32_t* __restrict__ indices,
                                              uint32_t* __restrict__ vox_of, RefCount rc, const uint32_t* __restrict__ sat_mask) {
    uint64_t n_refs;
    if (!ref_count(rc, n_refs)) return;      // uniform: the arrays are too small for this build, nothing is scattered
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const bool valid = t < T;
    int lx = 0, ly = 0, lz = 0, hx = -1, hy = -1, hz = -1;
    if (valid) {
        uint32_t rx = ranges[3ull * t], ry = ranges[3ull * t + 1], rz = ranges[3ull * t + 2];
        lx = (int)(rx & 0xFFFFu); hx = (int)(rx >> 16);
        ly = (int)(ry & 0xFFFFu); hy = (int)(ry >> 16);
        lz = (int)(rz & 0xFFFFu); hz = (int)(rz >> 16);
    }
    const bool nonempty = valid && lx <= hx && ly <= hy && lz <= hz;
    Tri9 tri = {};
    uint32_t mask = 0;
    if (SAT && nonempty) {
        // a small range brings its verdicts along; a big one (the whole wave walks it) is tested again
        if ((uint64_t)(hx - lx + 1) * (uint64_t)(hy - ly + 1) * (uint64_t)(hz - lz + 1) > SMALL_VOL) {
            const float4 a = aos[3ull * t], b = aos[3ull * t + 1], c = aos[3ull * t + 2];
            tri = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x };
        } else {
            mask = sat_mask[t];
        }
    }
    walk_range<FILL, SAT>(g, cnt, indices, vox_of, t, nonempty, lx, ly, lz, hx, hy, hz, tri, mask);
}

// ------------------------------------------------------------------------------------------
// scan (wave-level primitives)
// ------------------------------------------------------------------------------------------
// Inclusive scan over the 64 lanes (all must be active) in six DPP adds: shifts by 1, 2, 4, 8 inside each row of 16 (a
// lane without a source adds 0), then lane 15 of row 0 / 2 into rows 1 / 3 and lane 31 into rows 2 and 3.  The shuffle
// version was six ds_bpermute round trips with an address, a compare and a select each -- and, eight of them unrolled
// side by side in k_scan, a good part of that kernel's registers.
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);     // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);     // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);     // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);     // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);     // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);     // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ uint64_t wave_sum64(uint64_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d);
    return v;
}
__device__ __forceinline__ uint32_t wave_max32(uint32_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { uint32_t o = __shfl_down(v, d); v = o > v ? o : v; }
    return v;
}

// sum / maximum over the 64 lanes (all active), the same six DPP steps; the result is wave-uniform
__device__ __forceinline__ uint32_t wave_total32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan(v), 63); }
__device__ __forceinline__ uint32_t wave_max32_dpp(uint32_t v) {
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// Exclusive scan of the counters, in place, in ONE pass over the array (chained scan with decoupled look-back):
// a workgroup takes the next chunk (a ticket, so that every chunk before it has already started), scans its 32768
// counters in registers, publishes the chunk total, adds up the totals of the chunks before it -- looking back
// 64 chunks at a time until it meets one whose inclusive prefix is already known -- publishes its own inclusive
// prefix and writes the offsets.  4 bytes read and 4 written per voxel; the reduce / scan-of-sums / apply scheme
// it replaces read the array twice and had a single-workgroup kernel in the middle.
// Each wave owns 2048 consecutive counters and walks them in eight fully coalesced 1-KiB rows (lane i holds
// counters 4i..4i+3 of the row), scanning each row with wave shuffles and carrying the row total forward.
// status[c]: bits 63..62 = 0 nothing yet, 1 chunk total, 2 inclusive prefix; bits 61..0 the value (one 64-bit
// word, so flag and value arrive together).  A wait is bounded: a chunk that never shows up ends the wait,
// raises stats[5] and the build is reported as failed -- the kernel cannot hang.
constexpr uint64_t ST_AGG = 1ull << 62, ST_PFX = 2ull << 62, ST_VALUE = (1ull << 62) - 1ull;
constexpr uint32_t ST_SPIN_LIMIT = 1u << 22;
#ifndef HVRT_SCAN_ATTR
#define HVRT_SCAN_ATTR
#endif
constexpr int SCAN_WAVES = SCAN_CHUNK / 2048;       // 16 waves of 2048 counters each
// occ_bits (nullable): one bit per voxel, "has a list" -- the counters are in registers here anyway, and the
// distance field's x pass then reads 1/32 of what the offsets would cost it.
// NARROW: the counters come from the 16-bit array (four per 8-byte load) and the offsets go to cnt4, which nothing
// has written before (out of place: no memset of the 32-bit array, and half the bytes to read).  (Eight counters per
// 16-byte load and lane -- rows of 512 -- were slower, 2.12 against 1.70 ms at 1024^3: a lane then stores two
// neighbouring 16-byte pieces, and a wave's store instruction covers its 2 KiB with gaps.)
template <bool NARROW>
__global__ __launch_bounds__(SCAN_WAVES * 64) HVRT_SCAN_ATTR void k_scan(uint4* __restrict__ cnt4, const uint2* __restrict__ cnt16,
                                                          uint64_t* __restrict__ status,
                                                          uint64_t* __restrict__ aux, uint64_t* __restrict__ stats,
                                                          uint32_t* __restrict__ occ_bits) {
    __shared__ uint32_t s_chunk, s_w[SCAN_WAVES], s_occ[SCAN_WAVES], s_mx[SCAN_WAVES];
    __shared__ uint64_t s_base, s_w64[SCAN_WAVES];
    if (threadIdx.x == 0) s_chunk = atomicAdd(reinterpret_cast<uint32_t*>(stats + 6), 1u);
    __syncthreads();
    const uint64_t c = s_chunk;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint4* p = cnt4 + c * (SCAN_CHUNK / 4) + w * 512 + lane;
    // The narrow counters stay packed (two registers per row instead of four) and are unpacked where they are used, twice:
    // with them unpacked the kernel took 93 registers, ONE 16-wave workgroup fitted a CU, and its four phases -- load,
    // scan, look back, store -- had nothing to overlap with (1.70 ms at 1024^3 where a plain kernel with the same loads
    // and stores takes 1.22, scripts/ubench/hbm_rate.hip).
    uint4 v[NARROW ? 1 : 8];
    uint2 h[NARROW ? 8 : 1];
    uint32_t excl[8];
    if (NARROW) {
        const uint2* q = cnt16 + c * (SCAN_CHUNK / 4) + w * 512 + lane;
#pragma unroll
        for (int k = 0; k < 8; ++k) h[k] = q[k * 64];
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p[k * 64];
    }
    auto row = [&](int k) -> uint4 {
        if (NARROW) return make_uint4(h[k].x & 0xFFFFu, h[k].x >> 16, h[k].y & 0xFFFFu, h[k].y >> 16);
        return v[k];
    };
    uint32_t carry = 0, occ = 0, mx = 0;
    uint64_t exact = 0;     // my counters added up in 64 bits: the 32-bit sums below wrap beyond 2^32 - 1 references
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint4 r4 = row(k);
        const uint32_t tot = r4.x + r4.y + r4.z + r4.w;
        if (NARROW) exact += tot;       // (four 16-bit counters: no wrap)
        else exact += (uint64_t)r4.x + (uint64_t)r4.y + (uint64_t)r4.z + (uint64_t)r4.w;
        const uint32_t inc = wave_inclusive_scan(tot);
        excl[k] = carry + inc - tot;
        carry += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        occ += (r4.x > 0) + (r4.y > 0) + (r4.z > 0) + (r4.w > 0);
        if (occ_bits) {
            // my four voxels' bits, then eight lanes' nibbles into one word (lanes 0, 8, 16, ... store)
            uint32_t bits = (r4.x > 0 ? 1u : 0u) | (r4.y > 0 ? 2u : 0u) | (r4.z > 0 ? 4u : 0u) | (r4.w > 0 ? 8u : 0u);
            // (row_shl: lane i reads lane i + 1 / 2 / 4 of its row of 16 -- a storing lane's seven neighbours are in its row)
            bits |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bits, 0x101, 0xF, 0xF, false) << 4;
            bits |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bits, 0x102, 0xF, 0xF, false) << 8;
            bits |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bits, 0x104, 0xF, 0xF, false) << 16;
            if ((lane & 7) == 0) occ_bits[(c * SCAN_CHUNK + (uint64_t)w * 2048u + (uint64_t)k * 256u) / 32u + (uint32_t)(lane >> 3)] = bits;
        }
        const uint32_t m = r4.x > r4.y ? r4.x : r4.y, n = r4.z > r4.w ? r4.z : r4.w;
        const uint32_t mm = m > n ? m : n;
        mx = mx > mm ? mx : mm;
    }
    occ = wave_total32(occ);
    mx = wave_max32_dpp(mx);
    exact = NARROW ? (uint64_t)wave_total32((uint32_t)exact) : wave_sum64(exact);      // (narrow: 2048 x 65535 < 2^32)
    if (lane == 0) { s_w[w] = carry; s_occ[w] = occ; s_mx[w] = mx; s_w64[w] = exact; }
    __syncthreads();
    if (w == 0) {
        // wave 0: the waves' totals -> their exclusive prefixes (left in s_w), chunk total, statistics
        const uint32_t wt = lane < SCAN_WAVES ? s_w[lane] : 0u;
        const uint32_t winc = wave_inclusive_scan(wt);
        // the chunk's total as the chain carries it: exact (at most 32768 x 2^32 < 2^62, the width of a status value)
        const uint64_t total = wave_sum64(lane < SCAN_WAVES ? s_w64[lane] : 0ull);      // valid in lane 0, the only reader
        const uint32_t tocc = (uint32_t)wave_sum64(lane < SCAN_WAVES ? s_occ[lane] : 0u);
        const uint32_t tmx = wave_max32(lane < SCAN_WAVES ? s_mx[lane] : 0u);
        if (lane < SCAN_WAVES) s_w[lane] = winc - wt;
        if (lane == 0) {
            // occupied count and longest list of the chunk, reduced by k_scan_stats (tens of thousands of
            // same-address atomics from here would serialise in L2)
            aux[c] = (uint64_t)tocc | ((uint64_t)tmx << 32);
            __hip_atomic_store(&status[c], (c == 0 ? ST_PFX : ST_AGG) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        uint64_t before = 0;          // sum of the chunks before this one (lane 0)
        bool gave_up = false;
        if (c > 0) {
            long long pos = (long long)c - 1;
            for (;;) {
                const long long i = pos - lane;
                uint64_t word = ST_PFX;                       // in front of the array: an empty prefix
                if (i >= 0) {
                    uint32_t spins = 0;
                    for (;;) {
                        word = __hip_atomic_load(&status[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((word >> 62) != 0ull) break;
                        if (++spins >= ST_SPIN_LIMIT) { gave_up = true; word = ST_PFX; break; }
                        __builtin_amdgcn_s_sleep(2);
                    }
                }
                const unsigned long long pfx = __ballot((word >> 62) == 2ull);
                const int first = pfx ? __ffsll((long long)pfx) - 1 : 64;      // nearest chunk with a known prefix
                before += wave_sum64(lane <= first ? (word & ST_VALUE) : 0ull);
                if (pfx) break;
                pos -= 64;
            }
        }
        if (__ballot(gave_up) != 0ull && lane == 0) atomicOr((unsigned long long*)&stats[5], 1ull);
        if (lane == 0) {
            if (c > 0) __hip_atomic_store(&status[c], ST_PFX | (before + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (c == gridDim.x - 1) stats[0] = before + total;
            s_base = before;
        }
    }
    __syncthreads();
    const uint32_t base = (uint32_t)s_base + s_w[w];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint4 r4 = row(k);
        uint32_t run = base + excl[k];
        uint4 o;
        o.x = run; run += r4.x;
        o.y = run; run += r4.y;
        o.z = run; run += r4.z;
        o.w = run;
        p[k * 64] = o;
    }
}
// occupied voxels and longest list: reduction of the per-chunk pairs.  (Round 3 tried to fold this launch into k_scan --
// the last workgroup to publish its pair adding them all up behind a fence, or every workgroup adding its own pair to the
// statistics words where there was something to add: 1.48 -> 2.2 / 2.1 ms at 1024^3.  One 16-wave workgroup fits a CU, so
// whatever wave 0 waits for at the end -- an agent-scope fence, the round trip of one more atomic -- the whole CU waits
// for, 128 times over.  The launch of its own costs a few microseconds once.)
// vol_sums (nullable): the count pass's own total, striped (k_count<COUNT16>) -> stats[7]; without it stats[7] is
// the scan's total, i.e. "nothing to compare against, the counters were wide".  stats[0] != stats[7] at the end:
// a 16-bit counter carried, nothing built from these offsets may be used.
__global__ __launch_bounds__(256) void k_scan_stats(const uint64_t* __restrict__ aux, uint64_t n, uint64_t* __restrict__ stats,
                                                    const uint64_t* __restrict__ vol_sums) {
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        uint64_t sum = wave_sum64(vol_sums ? vol_sums[16u * threadIdx.x] : 0ull);
        if (threadIdx.x == 0) stats[7] = vol_sums ? sum : stats[0];
    }
    uint64_t occ = 0; uint32_t mx = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256u) {
        const uint64_t a = aux[i];
        occ += a & 0xFFFFFFFFull;
        const uint32_t m = (uint32_t)(a >> 32);
        mx = m > mx ? m : mx;
    }
    occ = wave_sum64(occ);
    mx = wave_max32(mx);
    if ((threadIdx.x & 63) == 0) {
        if (occ) atomicAdd((unsigned long long*)&stats[1], (unsigned long long)occ);
        if (mx) atomicMax((unsigned long long*)&stats[2], (unsigned long long)mx);
    }
}

// ------------------------------------------------------------------------------------------
// per-voxel ascending order
// ------------------------------------------------------------------------------------------
constexpr uint32_t SORT_SMALL = 96;      // lists up to here are ordered by rank counting, one lane per entry
constexpr uint32_t SORT_MID = 4096;      // static-LDS bitonic (16 KiB)
constexpr uint32_t SORT_BIG = 32768;     // dynamic-LDS bitonic (128 KiB of the CU's 160 KiB)

// One lane per CSR entry: the entry's final position inside its voxel's list is the number of entries
// of that list with a smaller triangle id (ids are unique within a list).  A list of at most SORT_SMALL entries
// that contains entry j lies inside [j - SORT_SMALL + 1, j + SORT_SMALL - 1], so a workgroup first brings the ids of
// its 256 entries and of SORT_SMALL - 1 neighbours either side into LDS -- one coalesced pass -- and every lane then
// counts over its list there (lanes of one list read the same word: a broadcast).  Reading the lists from global
// memory instead made the kernel a chain of L1 wave-loads, one per list entry and lane (1.5 ms at 1024^3 for 109 M
// entries, against 0.4 ms of traffic).  The result is written out of place.  Longer lists are copied through
// unchanged and queued once (by their first entry) for the bitonic kernels.
__global__ __launch_bounds__(256) void k_rank_sort(const uint32_t* __restrict__ unsorted, const uint32_t* __restrict__ vox_of,
                                                   const uint32_t* __restrict__ offsets, RefCount rc,
                                                   uint32_t* __restrict__ sorted, uint32_t* __restrict__ queue,
                                                   uint64_t queue_cap, uint64_t* __restrict__ stats) {
    constexpr uint32_t HALO = SORT_SMALL - 1;
    __shared__ uint32_t s_key[256 + 2 * HALO];
    uint64_t n_refs;
    if (!ref_count(rc, n_refs)) return;
    const uint64_t j0 = (uint64_t)blockIdx.x * 256u;
    if (j0 >= n_refs) return;                                  // uniform over the workgroup
    const uint64_t w0 = j0 >= HALO ? j0 - HALO : 0;            // first entry of the window
    const uint64_t w1 = j0 + 256u + HALO < n_refs ? j0 + 256u + HALO : n_refs;
    for (uint64_t q = w0 + threadIdx.x; q < w1; q += 256u) s_key[q - w0] = unsorted[q];
    __syncthreads();
    const uint64_t j = j0 + threadIdx.x;
    if (j >= n_refs) return;
    const uint32_t v = vox_of[j];
    const uint32_t b0 = offsets[v], b1 = offsets[(uint64_t)v + 1];
    const uint32_t key = s_key[j - w0];
    const uint32_t len = b1 - b0;
    if (len <= SORT_SMALL) {
        const uint32_t* __restrict__ list = s_key + (b0 - (uint32_t)w0);
        uint32_t rank = 0, q = 0;
        for (; q + 4 <= len; q += 4) {
            const uint32_t a = list[q], b = list[q + 1], c = list[q + 2], d = list[q + 3];
            rank += (a < key) + (b < key) + (c < key) + (d < key);
        }
        for (; q < len; ++q) rank += list[q] < key;
        sorted[b0 + rank] = key;
    } else {
        sorted[j] = key;
        if ((uint32_t)j == b0) {
            unsigned long long slot = atomicAdd((unsigned long long*)&stats[3], 1ull);
            if (slot < queue_cap) queue[slot] = v;
        }
    }
}

// Bitonic network for arbitrary n with every comparator ascending ("flip" first stage of each
// merge): elements at positions >= n act as +inf and never move, so no padding is stored.
template <typename Ptr>
__device__ __forceinline__ void bitonic_any_n(Ptr a, uint32_t n, uint32_t tid, uint32_t nthreads) {
    uint32_t P = 1;
    while (P < n) P <<= 1;
    const uint32_t half = P >> 1;
    for (uint32_t k = 2; k <= P; k <<= 1) {
        // flip: i and its mirror inside blocks of k
        for (uint32_t c = tid; c < half; c += nthreads) {
            uint32_t grp = c / (k >> 1), w = c % (k >> 1);
            uint32_t i = grp * k + w, j = grp * k + (k - 1 - w);
            if (j < n) { uint32_t x = a[i], y = a[j]; if (x > y) { a[i] = y; a[j] = x; } }
        }
        __syncthreads();
        for (uint32_t d = k >> 2; d > 0; d >>= 1) {
            for (uint32_t c = tid; c < half; c += nthreads) {
                uint32_t grp = c / d, w = c % d;
                uint32_t i = grp * 2 * d + w, j = i + d;
                if (j < n) { uint32_t x = a[i], y = a[j]; if (x > y) { a[i] = y; a[j] = x; } }
            }
            __syncthreads();
        }
    }
}

// TIER 0: SORT_SMALL+1..4096 (static LDS), 1: 4097..32768 (dynamic LDS) and above (in global memory) -- one launch for
// the two rare kinds: every launch costs a build its few microseconds, with or without lists to sort
template <int TIER, int NT>
__global__ __launch_bounds__(NT) void k_sort_long(const uint32_t* __restrict__ offsets, uint32_t* __restrict__ indices,
                                                  const uint32_t* __restrict__ queue, uint64_t queue_cap,
                                                  const uint64_t* __restrict__ stats) {
    extern __shared__ uint32_t s_dyn[];
    __shared__ uint32_t s_static[TIER == 0 ? SORT_MID : 1];
    uint32_t* lds = TIER == 0 ? s_static : s_dyn;
    uint64_t qn = stats[3];
    if (qn > queue_cap) qn = queue_cap;
    for (uint64_t q = blockIdx.x; q < qn; q += gridDim.x) {
        const uint32_t v = queue[q];
        const uint32_t b0 = offsets[v], b1 = offsets[(uint64_t)v + 1];
        const uint32_t len = b1 - b0;
        const bool mine = TIER == 0 ? (len <= SORT_MID) : (len > SORT_MID);
        if (!mine) continue;   // uniform over the workgroup
        uint32_t* a = indices + b0;
        if (TIER == 1 && len > SORT_BIG) {
            bitonic_any_n(a, len, threadIdx.x, NT);
        } else {
            for (uint32_t i = threadIdx.x; i < len; i += NT) lds[i] = a[i];
            __syncthreads();
            bitonic_any_n(lds, len, threadIdx.x, NT);
            for (uint32_t i = threadIdx.x; i < len; i += NT) a[i] = lds[i];
            __syncthreads();
        }
    }
}

// per-reference traversal records (SceneDev::ref_rec); also validates ingested indices (stats[4] = #bad)
__global__ __launch_bounds__(256) void k_expand_records(const uint32_t* __restrict__ indices, RefCount rc,
                                                        const float4* __restrict__ tri_rec, uint32_t n_tris,
                                                        float4* __restrict__ ref_rec, uint64_t* __restrict__ stats) {
    uint64_t n_refs;
    if (!ref_count(rc, n_refs)) return;
    const uint64_t j = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (j >= n_refs) return;
    const uint32_t ti = indices[j];
    float4 a = make_float4(0, 0, 0, 0), b = a, c = a;
    if (ti < n_tris) {
        a = tri_rec[3ull * ti]; b = tri_rec[3ull * ti + 1]; c = tri_rec[3ull * ti + 2];
    } else {
        atomicAdd((unsigned long long*)&stats[4], 1ull);
    }
    ref_rec[3 * j] = a;
    ref_rec[3 * j + 1] = b;
    ref_rec[3 * j + 2] = make_float4(c.x, __uint_as_float(ti), 0.0f, 0.0f);
}

// face_bits (SceneDev): for CSR entry j in voxel (ix,iy,iz), bit f says that the entry's triangle is also
// in the list of face neighbour f (0 -x, 1 +x, 2 -y, 3 +y, 4 -z, 5 +z).  A clear bit is always safe.  For
// grids built here a list is exactly the set of triangles whose AABB voxel range covers the voxel, so the
// bits come from the stored ranges; for an ingested CSR they come from a binary search of the neighbour's
// (ascending) list.  One lane per entry; a wave's 64 consecutive entries become one 64-bit word per face
// (__ballot), so no atomics are needed.
__device__ __forceinline__ bool list_contains(const uint32_t* __restrict__ indices, uint32_t b0, uint32_t b1, uint32_t ti) {
    while (b0 < b1) {
        const uint32_t m = b0 + ((b1 - b0) >> 1);
        const uint32_t v = indices[m];
        if (v == ti) return true;
        if (v < ti) b0 = m + 1; else b1 = m;
    }
    return false;
}

__global__ __launch_bounds__(256) void k_face_bits(const uint32_t* __restrict__ indices, RefCount rc,
                                                   const uint32_t* __restrict__ offsets, GridParams g, uint32_t n_tris,
                                                   const uint32_t* __restrict__ ranges, const uint32_t* __restrict__ vox_of,
                                                   uint32_t* __restrict__ face_bits) {
    uint64_t n_refs;
    if (!ref_count(rc, n_refs)) return;
    const uint64_t face_words = face_words_of(n_refs);      // the stride between the six bit planes (SceneDev::face_words)
    const uint64_t j = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    uint32_t mask = 0;
    if (j < n_refs) {
        const uint32_t ti = indices[j];
        if (ti < n_tris) {
            // voxel of entry j: recorded by the fill, else the last v with offsets[v] <= j
            uint64_t lo = 0, hi = g.nvox;            // invariant: offsets[lo] <= j < offsets[hi]
            if (vox_of) lo = vox_of[j];
            else while (hi - lo > 1) {
                const uint64_t m = lo + ((hi - lo) >> 1);
                if (offsets[m] <= (uint32_t)j) lo = m; else hi = m;
            }
            const uint64_t v = lo;
            const uint32_t plane = (uint32_t)g.nx * (uint32_t)g.ny;
            const int iz = (int)(v / plane);
            const uint32_t rem = (uint32_t)(v - (uint64_t)iz * plane);
            const int iy = (int)(rem / (uint32_t)g.nx), ix = (int)(rem - (uint32_t)iy * (uint32_t)g.nx);
            if (ranges) {
                const uint32_t rx = ranges[3ull * ti], ry = ranges[3ull * ti + 1], rz = ranges[3ull * ti + 2];
                mask |= ((int)(rx & 0xFFFFu) < ix) ? 1u : 0u;
                mask |= ((int)(rx >> 16) > ix) ? 2u : 0u;
                mask |= ((int)(ry & 0xFFFFu) < iy) ? 4u : 0u;
                mask |= ((int)(ry >> 16) > iy) ? 8u : 0u;
                mask |= ((int)(rz & 0xFFFFu) < iz) ? 16u : 0u;
                mask |= ((int)(rz >> 16) > iz) ? 32u : 0u;
            } else {
                const int64_t sv = (int64_t)v;
                const int64_t nb[6] = { sv - 1, sv + 1, sv - g.nx, sv + g.nx, sv - (int64_t)plane, sv + (int64_t)plane };
                const bool ok[6] = { ix > 0, ix + 1 < g.nx, iy > 0, iy + 1 < g.ny, iz > 0, iz + 1 < g.nz };
#pragma unroll
                for (int k = 0; k < 6; ++k)
                    if (ok[k] && list_contains(indices, offsets[nb[k]], offsets[nb[k] + 1], ti)) mask |= 1u << k;
            }
        }
    }
    const uint64_t w = j >> 6;   // this wave's 64 entries
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const unsigned long long bits = __ballot((mask >> k) & 1u);
        if ((threadIdx.x & 63) == 0 && 2 * w < face_words) {
            uint32_t* dst = face_bits + (uint64_t)k * face_words + 2 * w;
            dst[0] = (uint32_t)bits;
            if (2 * w + 1 < face_words) dst[1] = (uint32_t)(bits >> 32);
        }
    }
}

// occupancy statistics of an ingested CSR (hvrt_set_grid_csr*), and its validation: everything downstream (face
// bits, distance field, records, the traversal) trusts that offsets[0] == 0, that the offsets never decrease, that
// offsets[nvox] == n_refs and that every index names a triangle of the mesh.  stats[6] counts the offsets that break
// a rule, stats[4] (k_check_indices) the indices that do; the host rejects the grid when either is non-zero,
// before any derived data is built and whatever the record mode.
__global__ __launch_bounds__(256) void k_csr_stats(const uint32_t* __restrict__ offsets, uint64_t nvox, uint64_t n_refs,
                                                   uint64_t* __restrict__ stats) {
    uint64_t occ = 0, bad = 0; uint32_t mx = 0;
    for (uint64_t v = (uint64_t)blockIdx.x * 256u + threadIdx.x; v < nvox; v += (uint64_t)gridDim.x * 256u) {
        const uint32_t a = offsets[v], b = offsets[v + 1];
        bad += (b < a) + (v == 0 && a != 0u) + (v + 1 == nvox && (uint64_t)b != n_refs);
        const uint32_t len = b < a ? 0u : b - a;
        occ += len > 0;
        mx = len > mx ? len : mx;
    }
    occ = wave_sum64(occ);
    bad = wave_sum64(bad);
    mx = wave_max32(mx);
    if ((threadIdx.x & 63) == 0) {
        if (occ) atomicAdd((unsigned long long*)&stats[1], (unsigned long long)occ);
        if (mx) atomicMax((unsigned long long*)&stats[2], (unsigned long long)mx);
        if (bad) atomicAdd((unsigned long long*)&stats[6], (unsigned long long)bad);
    }
}
__global__ __launch_bounds__(256) void k_check_indices(const uint32_t* __restrict__ indices, uint64_t n_refs, uint32_t n_tris,
                                                       uint64_t* __restrict__ stats) {
    uint64_t bad = 0;
    for (uint64_t j = (uint64_t)blockIdx.x * 256u + threadIdx.x; j < n_refs; j += (uint64_t)gridDim.x * 256u)
        bad += indices[j] >= n_tris;
    bad = wave_sum64(bad);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd((unsigned long long*)&stats[4], (unsigned long long)bad);
}

// ------------------------------------------------------------------------------------------
// empty-space distance field (SceneDev::dist)
// ------------------------------------------------------------------------------------------
// dist[f] = min(255, city-block distance in voxels from voxel f to the nearest occupied voxel); 0 = occupied.
// The traversal moves one face per step, so from a voxel with dist D the next D-1 voxels of ANY ray are
// empty and need not be looked at.  The city-block distance separates per axis:
//   D1(x,y,z) = min_x' |x-x'| + [occupied(x',y,z) ? 0 : inf],  D2 = min_y' |y-y'| + D1(x,y',z),  D = min_z' ... D2
// and each 1-D pass is a forward and a backward sweep.  All values saturate at 255 (a saturated value only
// shortens a skip).

// x pass: one wave per x line, 64 voxels per step; occupancy comes straight from the CSR offsets, and the
// distance to the previous / next occupied voxel of the line from the wave's ballot word.
__global__ __launch_bounds__(256) void k_dist_x(const uint32_t* __restrict__ offsets, uint8_t* __restrict__ dist,
                                                uint32_t nx, uint64_t n_lines) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave0 = ((uint64_t)blockIdx.x * 256u + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * 256u) >> 6;
    for (uint64_t line = wave0; line < n_lines; line += n_waves) {
        const uint32_t* o = offsets + line * nx;
        uint8_t* p = dist + line * nx;
        uint32_t carry = 255;   // distance from the voxel before this chunk back to the last occupied one
        for (uint32_t c = 0; c < nx; c += 64) {
            const uint32_t i = c + lane;
            const bool in = i < nx;
            const bool occ = in && o[i + 1] != o[i];
            const unsigned long long m = __ballot(occ);
            const unsigned long long mm = m & (~0ull >> (63 - lane));
            uint32_t d = mm ? (uint32_t)(lane - (63 - __builtin_clzll(mm))) : carry + (uint32_t)lane + 1u;
            d = d > 255u ? 255u : d;
            if (in) p[i] = (uint8_t)d;
            carry = m ? (uint32_t)__builtin_clzll(m) : carry + 64u;
            carry = carry > 255u ? 255u : carry;
        }
        carry = 255;            // distance from the voxel after this chunk on to the next occupied one
        const uint32_t last = (nx - 1) / 64 * 64;
        for (uint32_t c = last;; c -= 64) {
            const uint32_t i = c + lane;
            const bool in = i < nx;
            const uint32_t fw = in ? p[i] : 255u;
            const unsigned long long m = __ballot(in && fw == 0);
            const unsigned long long mm = m & (~0ull << lane);
            uint32_t d = mm ? (uint32_t)(__builtin_ctzll(mm) - lane) : carry + (uint32_t)(64 - lane);
            d = d < fw ? d : fw;
            if (in && d != fw) p[i] = (uint8_t)d;
            carry = m ? (uint32_t)__builtin_ctzll(m) : carry + 64u;
            carry = carry > 255u ? 255u : carry;
            if (c == 0) break;
        }
    }
}

// x pass, fast form (nx a multiple of 4, at most 16384): a lane takes four neighbouring voxels, a wave 256 per
// step, and the line is read twice (the second time from cache) instead of being written twice: the first
// loop leaves the first / last occupied voxel of chunk c in lane c, two wave scans turn those into "last
// occupied before" / "first occupied after" each chunk, and the second loop writes the finished bytes, one
// dword per lane.
// BITS: occupancy comes from the bitmap k_scan left (one bit per voxel) instead of from the offsets.
template <bool BITS>
__global__ __launch_bounds__(256) void k_dist_x4(const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ occ_bits,
                                                 uint8_t* __restrict__ dist, uint32_t nx, uint64_t n_lines) {
    constexpr int FAR = 1 << 30;
    const int lane = threadIdx.x & 63;
    const uint64_t wave0 = ((uint64_t)blockIdx.x * 256u + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * 256u) >> 6;
    const uint32_t n_chunks = (nx + 255u) / 256u;
    for (uint64_t line = wave0; line < n_lines; line += n_waves) {
        const uint32_t* o = offsets + line * nx;
        uint8_t* p = dist + line * nx;
        auto nibble = [&](uint32_t c) -> uint32_t {
            const uint32_t i0 = c * 256u + 4u * (uint32_t)lane;
            uint32_t nib = 0;
            if (BITS) {
                const uint64_t gv = line * nx + i0;                      // a multiple of 4
                if (i0 < nx) nib = (occ_bits[gv >> 5] >> (uint32_t)(gv & 31u)) & 0xFu;
            } else if (i0 < nx) {
                uint32_t v[5];
#pragma unroll
                for (int k = 0; k < 5; ++k) v[k] = o[i0 + k];
#pragma unroll
                for (int k = 0; k < 4; ++k) nib |= (v[k + 1] != v[k] ? 1u : 0u) << k;
            }
            return nib;
        };
        int my_first = FAR, my_last = -FAR;
        for (uint32_t c = 0; c < n_chunks; ++c) {
            const uint32_t nib = nibble(c);
            const unsigned long long H = __ballot(nib != 0);
            const int lo = nib ? __builtin_ctz(nib) : 0, hi = nib ? 31 - __builtin_clz(nib) : 0;
            if (H) {
                const int fl = __builtin_ctzll(H), ll = 63 - __builtin_clzll(H);
                const int first = (int)(c * 256u) + 4 * fl + __shfl(lo, fl);
                const int last = (int)(c * 256u) + 4 * ll + __shfl(hi, ll);
                if (lane == (int)c) { my_first = first; my_last = last; }
            }
        }
        int inc_last = my_last, inc_first = my_first;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int a = __shfl_up(inc_last, d), b = __shfl_down(inc_first, d);
            if (lane >= d) inc_last = a > inc_last ? a : inc_last;
            if (lane + d < 64) inc_first = b < inc_first ? b : inc_first;
        }
        int before = __shfl_up(inc_last, 1), after = __shfl_down(inc_first, 1);
        if (lane == 0) before = -FAR;
        if (lane == 63) after = FAR;
        for (uint32_t c = 0; c < n_chunks; ++c) {
            const uint32_t nib = nibble(c);
            const unsigned long long H = __ballot(nib != 0);
            const int lo = nib ? __builtin_ctz(nib) : 0, hi = nib ? 31 - __builtin_clz(nib) : 0;
            const int left = __shfl(before, (int)c), right = __shfl(after, (int)c);
            const unsigned long long Hb = H & ((1ull << lane) - 1ull);        // lanes below me with an occupied voxel
            const unsigned long long Ha = H & ~((2ull << lane) - 1ull);       // lanes above me
            const int lb = Hb ? 63 - __builtin_clzll(Hb) : lane, la = Ha ? __builtin_ctzll(Ha) : lane;
            const int hi_b = __shfl(hi, lb), lo_a = __shfl(lo, la);
            const int cbase = (int)(c * 256u);
            const int prev_other = Hb ? cbase + 4 * lb + hi_b : left;
            const int next_other = Ha ? cbase + 4 * la + lo_a : right;
            const int base = cbase + 4 * lane;
            uint32_t word = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t mlo = nib & ((2u << k) - 1u), mhi = nib >> k;
                const int prev = mlo ? base + 31 - __builtin_clz(mlo) : prev_other;
                const int next = mhi ? base + k + __builtin_ctz(mhi) : next_other;
                const int i = base + k;
                int dmin = i - prev < next - i ? i - prev : next - i;
                dmin = dmin > 255 ? 255 : dmin;
                word |= (uint32_t)dmin << (8 * k);
            }
            if ((uint32_t)base < nx) *reinterpret_cast<uint32_t*>(p + base) = word;
        }
    }
}

// x pass for wide grids built here (nx a multiple of 16, at least 1024; occupancy from the scan's bitmap): a lane
// takes SIXTEEN neighbouring voxels -- one halfword of the bitmap in, one 16-byte store out -- so a wave covers
// 1024 voxels per step and the wave-level work (two ballots, two shuffles per step) is shared by four times as
// many voxels as in k_dist_x4; a 1024-wide line is a single step and needs none of the cross-chunk scans.
__global__ __launch_bounds__(256) void k_dist_x16(const uint32_t* __restrict__ occ_bits, uint8_t* __restrict__ dist,
                                                  uint32_t nx, uint64_t n_lines) {
    constexpr int FAR = 1 << 30;
    const int lane = threadIdx.x & 63;
    const uint64_t wave0 = ((uint64_t)blockIdx.x * 256u + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * 256u) >> 6;
    const uint32_t n_chunks = (nx + 1023u) / 1024u;              // <= 16
    const uint16_t* __restrict__ half = reinterpret_cast<const uint16_t*>(occ_bits);
    for (uint64_t line = wave0; line < n_lines; line += n_waves) {
        const uint64_t line0 = line * nx;                        // a multiple of 16
        auto bits16 = [&](uint32_t c) -> uint32_t {
            const uint32_t i0 = c * 1024u + 16u * (uint32_t)lane;
            return i0 < nx ? (uint32_t)half[(line0 + i0) >> 4] : 0u;
        };
        // first / last occupied voxel of chunk c -> lane c, then "last before" / "first after" every chunk
        int before = -FAR, after = FAR;
        if (n_chunks > 1) {
            int my_first = FAR, my_last = -FAR;
            for (uint32_t c = 0; c < n_chunks; ++c) {
                const uint32_t m = bits16(c);
                const unsigned long long H = __ballot(m != 0);
                if (H) {
                    const int fl = __builtin_ctzll(H), ll = 63 - __builtin_clzll(H);
                    const int lo = m ? __builtin_ctz(m) : 0, hi = m ? 31 - __builtin_clz(m) : 0;
                    const int first = (int)(c * 1024u) + 16 * fl + __shfl(lo, fl);
                    const int last = (int)(c * 1024u) + 16 * ll + __shfl(hi, ll);
                    if (lane == (int)c) { my_first = first; my_last = last; }
                }
            }
            int inc_last = my_last, inc_first = my_first;
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) {
                const int a = __shfl_up(inc_last, d), b = __shfl_down(inc_first, d);
                if (lane >= d) inc_last = a > inc_last ? a : inc_last;
                if (lane + d < 64) inc_first = b < inc_first ? b : inc_first;
            }
            before = __shfl_up(inc_last, 1); after = __shfl_down(inc_first, 1);
            if (lane == 0) before = -FAR;
            if (lane == 63) after = FAR;
        }
        for (uint32_t c = 0; c < n_chunks; ++c) {
            const uint32_t m = bits16(c);
            const unsigned long long H = __ballot(m != 0);
            const int lo = m ? __builtin_ctz(m) : 0, hi = m ? 31 - __builtin_clz(m) : 0;
            const int left = n_chunks > 1 ? __shfl(before, (int)c) : -FAR, right = n_chunks > 1 ? __shfl(after, (int)c) : FAR;
            const unsigned long long Hb = H & ((1ull << lane) - 1ull);        // lanes below me with an occupied voxel
            const unsigned long long Ha = H & ~((2ull << lane) - 1ull);       // lanes above me
            const int lb = Hb ? 63 - __builtin_clzll(Hb) : lane, la = Ha ? __builtin_ctzll(Ha) : lane;
            const int hi_b = __shfl(hi, lb), lo_a = __shfl(lo, la);
            const int cbase = (int)(c * 1024u);
            const int prev_other = Hb ? cbase + 16 * lb + hi_b : left;
            const int next_other = Ha ? cbase + 16 * la + lo_a : right;
            const int base = cbase + 16 * lane;
            uint32_t word[4] = { 0, 0, 0, 0 };
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const uint32_t mlo = m & ((2u << k) - 1u), mhi = m >> k;
                const int prev = mlo ? base + 31 - __builtin_clz(mlo) : prev_other;
                const int next = mhi ? base + k + __builtin_ctz(mhi) : next_other;
                const int i = base + k;
                int dmin = i - prev < next - i ? i - prev : next - i;
                dmin = dmin > 255 ? 255 : dmin;
                word[k >> 2] |= (uint32_t)dmin << (8 * (k & 3));
            }
            if ((uint32_t)base < nx) *reinterpret_cast<uint4*>(dist + line0 + (uint32_t)base) = make_uint4(word[0], word[1], word[2], word[3]);
        }
    }
}

// y and z passes: one lane per line (lines are indexed so that neighbouring lanes are neighbouring x: every
// access of a wave is one contiguous run); sixteen loads in flight per lane.  With W = 4 a lane carries four
// neighbouring lines in one dword (needs the line starts and the stride to be multiples of 4).
template <int W> struct DistWord;
template <> struct DistWord<1> { using T = uint8_t; };
template <> struct DistWord<4> { using T = uint32_t; };

template <int W>
__device__ __forceinline__ uint32_t dist_relax(uint32_t (&run)[W], uint32_t v) {
    uint32_t out = 0;
#pragma unroll
    for (int b = 0; b < W; ++b) {
        const uint32_t vb = (v >> (8 * b)) & 0xFFu;
        run[b] = run[b] + 1u < vb ? run[b] + 1u : vb;
        out |= run[b] << (8 * b);
    }
    return out;
}

template <int W>
__global__ __launch_bounds__(256) void k_dist_sweep(uint8_t* __restrict__ dist, uint64_t n_lines, uint64_t inner,
                                                    uint64_t outer_stride, uint64_t stride, uint32_t n) {
    using T = typename DistWord<W>::T;
    constexpr uint32_t U = 16;   // loads in flight per lane: small grids have few lines, and each is a serial chain
    const uint64_t line = (uint64_t)blockIdx.x * 256u + threadIdx.x;   // in units of W lines
    if (line >= n_lines) return;
    const uint64_t first = line * W;
    uint8_t* p = dist + (first % inner) + (first / inner) * outer_stride;
    auto at = [&](uint32_t i) -> T& { return *reinterpret_cast<T*>(p + (uint64_t)i * stride); };
    uint32_t run[W];
#pragma unroll
    for (int b = 0; b < W; ++b) run[b] = 255;
    uint32_t i = 0;
    for (; i + U <= n; i += U) {
        uint32_t v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = at(i + k);
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const uint32_t o = dist_relax<W>(run, v[k]);
            if (o != v[k]) at(i + k) = (T)o;
        }
    }
    for (; i < n; ++i) {
        const uint32_t v = at(i);
        const uint32_t o = dist_relax<W>(run, v);
        if (o != v) at(i) = (T)o;
    }
#pragma unroll
    for (int b = 0; b < W; ++b) run[b] = 255;
    i = n;
    for (; i >= U; i -= U) {
        uint32_t v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = at(i - 1 - k);
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const uint32_t o = dist_relax<W>(run, v[k]);
            if (o != v[k]) at(i - 1 - k) = (T)o;
        }
    }
    for (; i > 0; --i) {
        const uint32_t v = at(i - 1);
        const uint32_t o = dist_relax<W>(run, v);
        if (o != v) at(i - 1) = (T)o;
    }
}

// The same sweep with the line cut into four segments, one per wave of the workgroup (the four waves share their 64
// lines): a one-dimensional city-block transform is d[i] = min_j (v[j] + |i - j|), a min-reduction, so what a segment
// contributes to the lines' other segments is two numbers -- min_j (v[j] + distance to the segment's last / first
// voxel) -- found in one pass of independent loads; after a barrier every wave knows what arrives at its segment from
// the left and from the right and runs the forward / backward chains over its own quarter only.  Four times the waves
// and a quarter of the chain: at 256^3 a sweep is 65 536 lines, 1024 waves of 256-step chains as one wave per line set.
template <int W>
__global__ __launch_bounds__(256) void k_dist_sweep_seg(uint8_t* __restrict__ dist, uint64_t n_lines, uint64_t inner,
                                                        uint64_t outer_stride, uint64_t stride, uint32_t n) {
    using T = typename DistWord<W>::T;
    constexpr uint32_t U = 16, FAR = 1u << 20;
    constexpr int S = 4;
    __shared__ uint32_t s_last[S][W][64], s_first[S][W][64];
    const int seg = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t line = (uint64_t)blockIdx.x * 64u + (uint32_t)lane;     // in units of W lines
    const bool in = line < n_lines;
    const uint64_t first = (in ? line : 0) * W;
    uint8_t* p = dist + (first % inner) + (first / inner) * outer_stride;
    auto at = [&](uint32_t i) -> T& { return *reinterpret_cast<T*>(p + (uint64_t)i * stride); };
    const uint32_t len = (n + S - 1) / S;
    const uint32_t i0 = (uint32_t)seg * len < n ? (uint32_t)seg * len : n, i1 = i0 + len < n ? i0 + len : n;
    // what this segment sends on: to its last voxel (for the segments after it) and to its first (for those before)
    uint32_t to_last[W], to_first[W];
#pragma unroll
    for (int b = 0; b < W; ++b) { to_last[b] = FAR; to_first[b] = FAR; }
    if (in) {
        uint32_t i = i0;
        for (; i + U <= i1; i += U) {
            uint32_t v[U];
#pragma unroll
            for (int k = 0; k < U; ++k) v[k] = at(i + k);
#pragma unroll
            for (int k = 0; k < U; ++k)
#pragma unroll
                for (int b = 0; b < W; ++b) {
                    const uint32_t vb = (v[k] >> (8 * b)) & 0xFFu;
                    const uint32_t a = vb + (i1 - 1u - (i + k)), c = vb + (i + k - i0);
                    to_last[b] = a < to_last[b] ? a : to_last[b];
                    to_first[b] = c < to_first[b] ? c : to_first[b];
                }
        }
        for (; i < i1; ++i) {
            const uint32_t v = at(i);
#pragma unroll
            for (int b = 0; b < W; ++b) {
                const uint32_t vb = (v >> (8 * b)) & 0xFFu;
                const uint32_t a = vb + (i1 - 1u - i), c = vb + (i - i0);
                to_last[b] = a < to_last[b] ? a : to_last[b];
                to_first[b] = c < to_first[b] ? c : to_first[b];
            }
        }
    }
#pragma unroll
    for (int b = 0; b < W; ++b) { s_last[seg][b][lane] = to_last[b]; s_first[seg][b][lane] = to_first[b]; }
    __syncthreads();
    if (!in || i0 >= i1) return;
    auto seg_len = [&](int k) { const uint32_t a = (uint32_t)k * len < n ? (uint32_t)k * len : n; return (a + len < n ? a + len : n) - a; };
    uint32_t run[W];
    // forward: what reaches my first voxel from the left, then the chain over my segment
#pragma unroll
    for (int b = 0; b < W; ++b) {
        uint32_t c = FAR;
        for (int k = 0; k < seg; ++k) {
            const uint32_t through = c + seg_len(k), from = s_last[k][b][lane] + 1u;
            c = through < from ? through : from;
        }
        run[b] = (c > 256u ? 256u : c) - 1u;        // so that run + 1 is the arriving value (at least 1)
    }
    uint32_t i = i0;
    for (; i + U <= i1; i += U) {
        uint32_t v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = at(i + k);
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const uint32_t o = dist_relax<W>(run, v[k]);
            if (o != v[k]) at(i + k) = (T)o;
        }
    }
    for (; i < i1; ++i) {
        const uint32_t v = at(i);
        const uint32_t o = dist_relax<W>(run, v);
        if (o != v) at(i) = (T)o;
    }
    // backward
#pragma unroll
    for (int b = 0; b < W; ++b) {
        uint32_t c = FAR;
        for (int k = S - 1; k > seg; --k) {
            const uint32_t through = c + seg_len(k), from = s_first[k][b][lane] + 1u;
            c = through < from ? through : from;
        }
        run[b] = (c > 256u ? 256u : c) - 1u;
    }
    i = i1;
    for (; i >= i0 + U; i -= U) {
        uint32_t v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = at(i - 1 - k);
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const uint32_t o = dist_relax<W>(run, v[k]);
            if (o != v[k]) at(i - 1 - k) = (T)o;
        }
    }
    for (; i > i0; --i) {
        const uint32_t v = at(i - 1);
        const uint32_t o = dist_relax<W>(run, v);
        if (o != v) at(i - 1) = (T)o;
    }
}

// ------------------------------------------------------------------------------------------
// mesh bounds (computeMeshBounds, SceneCreator.cpp:49-84)
// ------------------------------------------------------------------------------------------
// The reference folds sequentially with `if (c < mn) mn = c` from +FLT_MAX (and the mirror image for
// max from -FLT_MAX): NaNs and infinities never enter, and among equal minima (+0 / -0) the first
// in vertex order wins.  Carrying (value, first index) makes that an order-free reduction.
struct MinMax6 {
    float v[6];
    uint64_t i[6];
};
__device__ __forceinline__ void mm_init(MinMax6& m) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { m.v[k] = FLT_MAX; m.v[3 + k] = -FLT_MAX; m.i[k] = ~0ull; m.i[3 + k] = ~0ull; }
}
__device__ __forceinline__ void mm_take(MinMax6& m, int k, float v, uint64_t i) {
    if (k < 3) { if (v < m.v[k] || (v == m.v[k] && i < m.i[k])) { m.v[k] = v; m.i[k] = i; } }
    else       { if (m.v[k] < v || (v == m.v[k] && i < m.i[k])) { m.v[k] = v; m.i[k] = i; } }
}
__device__ __forceinline__ void mm_wave_reduce(MinMax6& m) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            float ov = __shfl_down(m.v[k], d);
            uint64_t oi = __shfl_down(m.i[k], d);
            mm_take(m, k, ov, oi);
        }
    }
}
__device__ __forceinline__ void mm_block_reduce_store(MinMax6& m, float* out_v, uint64_t* out_i) {
    __shared__ float s_v[4][6];
    __shared__ uint64_t s_i[4][6];
    mm_wave_reduce(m);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) { s_v[w][k] = m.v[k]; s_i[w][k] = m.i[k]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int ww = 1; ww < 4; ++ww)
            for (int k = 0; k < 6; ++k) mm_take(m, k, s_v[ww][k], s_i[ww][k]);
        for (int k = 0; k < 6; ++k) { out_v[k] = m.v[k]; out_i[k] = m.i[k]; }
    }
}

__global__ __launch_bounds__(256) void k_bounds_partial(const float* __restrict__ pts, uint64_t n_points,
                                                        float* __restrict__ part_v, uint64_t* __restrict__ part_i) {
    MinMax6 m;
    mm_init(m);
    const uint64_t n_floats = n_points * 3;
    for (uint64_t e = (uint64_t)blockIdx.x * 256u + threadIdx.x; e < n_floats; e += (uint64_t)gridDim.x * 256u) {
        const float c = pts[e];
        const uint64_t pi = e / 3;
        const int k = (int)(e - pi * 3);
        const float cmin = (c < FLT_MAX) ? c : FLT_MAX;      // what a sequential fold could ever hold
        const float cmax = (-FLT_MAX < c) ? c : -FLT_MAX;
        if (k == 0)      { mm_take(m, 0, cmin, pi); mm_take(m, 3, cmax, pi); }
        else if (k == 1) { mm_take(m, 1, cmin, pi); mm_take(m, 4, cmax, pi); }
        else             { mm_take(m, 2, cmin, pi); mm_take(m, 5, cmax, pi); }
    }
    mm_block_reduce_store(m, part_v + 6ull * blockIdx.x, part_i + 6ull * blockIdx.x);
}

__global__ __launch_bounds__(256) void k_bounds_final(const float* __restrict__ part_v, const uint64_t* __restrict__ part_i,
                                                      int n_parts, float* __restrict__ out6) {
    MinMax6 m;
    mm_init(m);
    for (int p = threadIdx.x; p < n_parts; p += 256)
        for (int k = 0; k < 6; ++k) mm_take(m, k, part_v[6 * p + k], part_i[6 * p + k]);
    __shared__ uint64_t dummy_i[6];
    mm_block_reduce_store(m, out6, dummy_i);
}

}  // namespace

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
hipError_t launch_prepare_records(const float* tri_aos, float4* tri_rec, uint32_t T, hipStream_t s) {
    if (T == 0) return hipSuccess;
    k_prepare_records<<<(T + 255) / 256, 256, 0, s>>>((const float4*)tri_aos, tri_rec, T);
    return hipGetLastError();
}

// the small clears in front of a build with 16-bit counters, in one launch: offsets[0] (with its three pad words) and the
// statistics block (the counters themselves are cleared by the runtime's fill, the fastest writer there is)
__global__ __launch_bounds__(256) void k_build_reset(uint4* __restrict__ cnt_base4, uint64_t* __restrict__ stats, uint64_t n_words) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i == 0) *cnt_base4 = make_uint4(0u, 0u, 0u, 0u);
    if (i < n_words) stats[i] = 0ull;
}
hipError_t launch_build_reset(uint32_t* cnt_base, uint64_t* stats, uint64_t n_words, hipStream_t s) {
    k_build_reset<<<(unsigned)((n_words + 255) / 256), 256, 0, s>>>((uint4*)cnt_base, stats, n_words);
    return hipGetLastError();
}

hipError_t launch_count(const float* tri_aos, uint32_t T, const GridParams& g, uint32_t* cnt, bool narrow, uint64_t* vol_sums,
                        uint32_t* ranges, uint32_t* sat_mask, hipStream_t s) {
    if (T == 0) return hipSuccess;
    const bool sat = sat_mask != nullptr;
    if (sat && narrow) return hipErrorInvalidValue;       // (the 16-bit counters' cross-check is the sum of the range volumes)
    if (sat) k_count<COUNT32, true><<<(T + 255) / 256, 256, 0, s>>>((const float4*)tri_aos, T, g, cnt, ranges, nullptr, sat_mask);
    else if (narrow) k_count<COUNT16><<<(T + 255) / 256, 256, 0, s>>>((const float4*)tri_aos, T, g, cnt, ranges, vol_sums, nullptr);
    else k_count<COUNT32><<<(T + 255) / 256, 256, 0, s>>>((const float4*)tri_aos, T, g, cnt, ranges, nullptr, nullptr);
    return hipGetLastError();
}

hipError_t launch_scan(uint32_t* cnt, const uint32_t* cnt16, const uint64_t* vol_sums, uint64_t n_chunks, uint64_t* status_aux,
                       uint64_t* stats, uint32_t* occ_bits, hipStream_t s) {
    // status words must start empty (the caller zeroed them with the counters); the ticket (stats[6]) and the failure
    // flag (stats[5]) were zeroed with stats
    if (cnt16) k_scan<true><<<(unsigned)n_chunks, SCAN_WAVES * 64, 0, s>>>((uint4*)cnt, (const uint2*)cnt16, status_aux, status_aux + n_chunks, stats, occ_bits);
    else k_scan<false><<<(unsigned)n_chunks, SCAN_WAVES * 64, 0, s>>>((uint4*)cnt, nullptr, status_aux, status_aux + n_chunks, stats, occ_bits);
    const uint64_t blocks = (n_chunks + 2047) / 2048;
    k_scan_stats<<<(unsigned)(blocks > 256 ? 256 : blocks), 256, 0, s>>>(status_aux + n_chunks, n_chunks, stats, cnt16 ? vol_sums : nullptr);
    return hipGetLastError();
}

hipError_t launch_fill(const uint32_t* ranges, const float* tri_aos, uint32_t T, const GridParams& g, uint32_t* cnt, uint32_t* indices,
                       uint32_t* vox_of, const RefCount& rc, const uint32_t* sat_mask, hipStream_t s) {
    if (T == 0) return hipSuccess;
    if (sat_mask) k_fill<true><<<(T + 255) / 256, 256, 0, s>>>(ranges, (const float4*)tri_aos, T, g, cnt, indices, vox_of, rc, sat_mask);
    else k_fill<false><<<(T + 255) / 256, 256, 0, s>>>(ranges, nullptr, T, g, cnt, indices, vox_of, rc, nullptr);
    return hipGetLastError();
}

hipError_t launch_sort(const uint32_t* offsets, const uint32_t* unsorted, const uint32_t* vox_of, const RefCount& rc,
                       uint32_t* indices, uint32_t* queue, uint64_t queue_cap, uint64_t* stats, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_sort_long<1, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)(SORT_BIG * sizeof(uint32_t)));
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const uint64_t n_launch = rc.live ? rc.cap : rc.n;       // lanes to launch: the capacity when the number is still on the device
    if (n_launch == 0) return hipSuccess;
    k_rank_sort<<<(unsigned)((n_launch + 255) / 256), 256, 0, s>>>(unsorted, vox_of, offsets, rc, indices, queue, queue_cap, stats);
    k_sort_long<0, 256><<<2048, 256, 0, s>>>(offsets, indices, queue, queue_cap, stats);
    k_sort_long<1, 1024><<<256, 1024, SORT_BIG * sizeof(uint32_t), s>>>(offsets, indices, queue, queue_cap, stats);
    return hipGetLastError();
}

hipError_t launch_expand_records(const uint32_t* indices, const RefCount& rc, const float4* tri_rec, uint32_t n_tris,
                                 float4* ref_rec, uint64_t* stats, hipStream_t s) {
    const uint64_t n_launch = rc.live ? rc.cap : rc.n;
    if (n_launch == 0) return hipSuccess;
    k_expand_records<<<(unsigned)((n_launch + 255) / 256), 256, 0, s>>>(indices, rc, tri_rec, n_tris, ref_rec, stats);
    return hipGetLastError();
}

uint64_t face_words_for(uint64_t n_refs) { return (n_refs + 63) / 64 * 2 + 2; }

hipError_t launch_face_bits(const uint32_t* indices, const RefCount& rc, const uint32_t* offsets, const GridParams& g,
                            uint32_t n_tris, const uint32_t* ranges, const uint32_t* vox_of, uint32_t* face_bits, hipStream_t s) {
    // one lane per entry, rounded up so that every 64-bit word of the arrays (incl. the tail) is written
    const uint64_t lanes = (face_words_for(rc.live ? rc.cap : rc.n) * 32 + 255) / 256 * 256;
    k_face_bits<<<(unsigned)(lanes / 256), 256, 0, s>>>(indices, rc, offsets, g, n_tris, ranges, vox_of, face_bits);
    return hipGetLastError();
}

hipError_t launch_distance_field(const uint32_t* offsets, const uint32_t* occ_bits, const GridParams& g, uint8_t* dist, hipStream_t s) {
    const uint64_t nx = (uint64_t)g.nx, ny = (uint64_t)g.ny, nz = (uint64_t)g.nz;
    {
        const uint64_t lines = ny * nz;
        uint64_t blocks = (lines + 3) / 4;
        if (blocks > 65536) blocks = 65536;
        if (nx % 16 == 0 && nx >= 1024 && nx <= 16384 && occ_bits) k_dist_x16<<<(unsigned)blocks, 256, 0, s>>>(occ_bits, dist, (uint32_t)nx, lines);
        else if (nx % 4 == 0 && nx <= 16384 && occ_bits) k_dist_x4<true><<<(unsigned)blocks, 256, 0, s>>>(offsets, occ_bits, dist, (uint32_t)nx, lines);
        else if (nx % 4 == 0 && nx <= 16384) k_dist_x4<false><<<(unsigned)blocks, 256, 0, s>>>(offsets, nullptr, dist, (uint32_t)nx, lines);
        else k_dist_x<<<(unsigned)blocks, 256, 0, s>>>(offsets, dist, (uint32_t)nx, lines);
    }
    // four lines per lane where line starts and strides are multiples of 4 (dist itself is 256-B aligned) -- but
    // only when that still leaves every SIMD a few waves: a sweep is a serial chain per lane, and at 256^3 four
    // lines per lane are 256 waves for 1024 SIMDs
    // lines of at least SWEEP_SEG_MIN voxels are swept in four segments by the four waves of a workgroup -- while one
    // wave per 64 line units would leave the SIMDs short of waves; beyond that the whole-line kernel already runs at
    // the memory's pace and reads every voxel once less (1024^3: 4.2 against 4.5 ms for the derived data)
    constexpr uint64_t SWEEP_WIDE_MIN = 4 * 64 * 1024, SWEEP_SEG_MIN = 64, SWEEP_SEG_MAX_UNITS = 4 * 64 * 1024;
    if (ny > 1) {
        const uint64_t lines = nx * nz;
        const bool wide = nx % 4 == 0 && lines >= SWEEP_WIDE_MIN;
        const uint64_t units = wide ? lines / 4 : lines;
        if (ny >= SWEEP_SEG_MIN && units < SWEEP_SEG_MAX_UNITS) {
            if (wide) k_dist_sweep_seg<4><<<(unsigned)((units + 63) / 64), 256, 0, s>>>(dist, units, nx, nx * ny, nx, (uint32_t)ny);
            else k_dist_sweep_seg<1><<<(unsigned)((units + 63) / 64), 256, 0, s>>>(dist, units, nx, nx * ny, nx, (uint32_t)ny);
        } else if (wide) k_dist_sweep<4><<<(unsigned)((units + 255) / 256), 256, 0, s>>>(dist, units, nx, nx * ny, nx, (uint32_t)ny);
        else k_dist_sweep<1><<<(unsigned)((units + 255) / 256), 256, 0, s>>>(dist, units, nx, nx * ny, nx, (uint32_t)ny);
    }
    if (nz > 1) {
        const uint64_t lines = nx * ny;
        const bool wide = lines % 4 == 0 && lines >= SWEEP_WIDE_MIN;
        const uint64_t units = wide ? lines / 4 : lines;
        if (nz >= SWEEP_SEG_MIN && units < SWEEP_SEG_MAX_UNITS) {
            if (wide) k_dist_sweep_seg<4><<<(unsigned)((units + 63) / 64), 256, 0, s>>>(dist, units, lines, 0, lines, (uint32_t)nz);
            else k_dist_sweep_seg<1><<<(unsigned)((units + 63) / 64), 256, 0, s>>>(dist, units, lines, 0, lines, (uint32_t)nz);
        } else if (wide) k_dist_sweep<4><<<(unsigned)((units + 255) / 256), 256, 0, s>>>(dist, units, lines, 0, lines, (uint32_t)nz);
        else k_dist_sweep<1><<<(unsigned)((units + 255) / 256), 256, 0, s>>>(dist, units, lines, 0, lines, (uint32_t)nz);
    }
    return hipGetLastError();
}

hipError_t launch_csr_stats(const uint32_t* offsets, uint64_t nvox, const uint32_t* indices, uint64_t n_refs,
                            uint32_t n_tris, uint64_t* stats, hipStream_t s) {
    uint64_t blocks = (nvox + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks == 0) blocks = 1;
    k_csr_stats<<<(unsigned)blocks, 256, 0, s>>>(offsets, nvox, n_refs, stats);
    if (n_refs) {
        blocks = (n_refs + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        k_check_indices<<<(unsigned)blocks, 256, 0, s>>>(indices, n_refs, n_tris, stats);
    }
    return hipGetLastError();
}

int bounds_blocks(uint64_t n_points) {
    uint64_t b = (n_points * 3 + 256 * 16 - 1) / (256 * 16);
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (int)b;
}

hipError_t launch_bounds(const float* pts, uint64_t n_points, float* part_v, uint64_t* part_i, int n_blocks,
                         float* out6, hipStream_t s) {
    k_bounds_partial<<<n_blocks, 256, 0, s>>>(pts, n_points, part_v, part_i);
    k_bounds_final<<<1, 256, 0, s>>>(part_v, part_i, n_blocks, out6);
    return hipGetLastError();
}

}  // namespace hvrt
