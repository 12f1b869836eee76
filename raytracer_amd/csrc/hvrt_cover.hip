// hvrt_cover.hip -- which 8x8-pixel tiles of a frame can hold a hit at all: a pre-pass of k_render.
//
// The reference's camera rule looks at the whole grid from outside, and most primary rays of such a view cross the grid
// without hitting anything (configs 2-4: 94.5 % of them) -- 186 voxels each, every one a miss.  k_render walks a tile only
// if this pass could NOT prove that every ray of the tile misses every triangle; the pixels of the other tiles get the
// reference's result for a miss (background colour, HVRT_MISS, FLT_MAX) without a walk.  The pass has no counterpart in
// the reference (OctreeLikeRayTracer.cpp:88-126 walks every ray); what it must guarantee is that the reference's
// ray / triangle test (GeometryUtils.cpp:57-97), fed with the floats k_render feeds it, rejects every triangle for every
// sample of a tile it leaves unmarked -- whichever voxels the walk would have visited.
//
// Geometry.  With s = o - v0 (the reference's own float subtraction) the test depends on the ray through three linear
// forms of its direction d:   det = d . N,  s . p = d . M,  d . q = d . Q   with N = e2 x e1, M = e2 x s, Q = s x e1,
// and u = d.M / d.N, v = d.Q / d.N, w := 1 - u - v = d.(N - M - Q) / d.N.  A hit needs u, v, w in [0, 1] (up to the
// reference's FLT_EPSILON tolerances), and "c outside [0, 1]" is  C (C - D) > 0  for a coordinate c = C / D -- no case
// distinction on the sign of D.  A direction is d = g / |g| (+ rounding) with g = A + a hor + b ver, A = lower_left -
// origin, (a, b) = ((x + jx) / W, (y + jy) / H): every form is AFFINE in (a, b), so over the rectangle of (a, b) that a
// block of pixels spans it ranges within  centre +- (|C_a| h_a + |C_b| h_b).  A block is rejected for a triangle when one
// of the coordinates is provably outside [0, 1] on the same side for the whole rectangle:
//      (+)  min C       - m max|D| > S_c   and   min (C - D) - m max|D| > S_c      or
//      (-)  max C       + m max|D| < -S_c  and   max (C - D) + m max|D| < -S_c
// (m = 2^-11 relative margin, S_c the coordinate's absolute slack, below), or when |det| < FLT_EPSILON on all of it (the reference's
// "parallel" exit), or when the hit parameter t = (-s . N) / det is negative on all of it.
//
// Why (+) implies the reference's rejection for the floats it computes (the other cases mirror it).  In direction space
// (divide by |g| > 0): U - m |D| > S_d and U - D - m |D| > S_d, with det_f = D + dD, (s.p)_f = U + dU, |dD| <= Rn :=
// 7 ulp |e1||e2|, |dU| <= Rm := 7 ulp |s||e2| (each component of p = d x e2 is off by <= 2 ulp |d||e2|, a three-term dot
// adds <= 3 ulp |a||b|; |d| = 1), S_d >= 2 (Rn + Rm + Rq):
//   * |det_f| < EPS: the reference returns "no hit" itself;
//   * det_f > 0: (s.p)_f - det_f > m |D| + Rn >= m det_f, so (s.p)_f / det_f > 1 + m and u_f = fl(fl(1 / det_f) (s.p)_f)
//     > (1 + m)(1 - 2^-22) > 1 + 2 EPS: rejected by "u > 1 && |u - 1| > EPS";
//   * det_f < 0: (s.p)_f > m |D| + Rn >= m |det_f|, so u_f < -m / 2: rejected by "u < 0 && |u| > EPS".
//   For v the same with Q ("v < -m" is the reference's v test; "v > 1 + m" leaves u + v > 1 + m / 2 unless u was already
//   rejected); for w, "w < -m" is u + v > 1 + m and "w > 1 + m" is u + v < -m, i.e. u < -m / 2 or v < -m / 2.
//   The t test: e2 . q = -s . N =: E does not depend on the ray; where |E| exceeds its own rounding (2^-19 |s|_1 |e1|_1
//   |e2|_1) and D has the opposite sign by more than S, t_f = fl(1 / det_f) (e2 . q)_f <= 0 fails "t > EPS".
// The slack, in the space of g (direction-space terms times |g| <= gmax, the largest |g| at the screen's corners); lengths
// are Euclidean, P_u := |s||e2| + |e1||e2| the magnitudes behind u's two forms (P_v alike with e1, P_w = P_u + P_v,
// P_D = |e1||e2| for D alone):
//   * the reference's roundings above: <= 14 ulp P_u;
//   * the direction k_render really uses is fl(normalize(fl(ll + hor a + ver b - org))), a and b rounded too.  Component i
//     of the sum is off by <= ulp B_i, B_i = 2 |ll_i| + 5 |hor_i| + 4 |ver_i| + (|A_i| + |hor_i| + |ver_i|) (each operation
//     rounds a number no larger than the partial sum it produces; a and b carry 2 ulp each); normalising turns an error
//     delta of g into a direction error of at most 2 |delta| / |g| and adds 3.5 ulp of its own: eps_d := (2 |B| / gmin +
//     4) ulp, gmin = the eye's distance from the screen's plane <= |g|.  It moves a form d . X by <= eps_d |X|, and
//     |a x b| <= |a||b|;
//   * this kernel's own fp32 evaluation of N, M, Q (3 ulp |a||b| each), of their dots with A, hor, ver (another 6 ulp of
//     |X| times the vector's length) and of the value at a point (3 more roundings): <= 16 ulp G2 P_u, G2 := |A| + |hor|
//     + |ver|; rectangles and bounds are padded by 2^-20 (relative to the coordinates in play) against the roundings of
//     x / W here and there;
//   all of it inside  S_u = K P_u + 1e-30,  K = 2 x (1.01 gmax (14 ulp + eps_d) + 16 ulp G2)  -- twice the sum of the
//   above.  Every decision is a comparison that is false on NaN, so non-finite numbers anywhere reject nothing; a frame or
//   a triangle this analysis does not cover (non-finite, degenerate or absurdly scaled frame, overflowing products) marks
//   the whole frame.
// Marking more than necessary is always allowed; the walk then finds the misses itself.
//
// The bounding rectangle.  At a single point the tests above read, with all forms multiplied by the sign of E (so that a
// hit in front of the eye has D > 0):  the ray is rejected if  f_T := D + S_D < 0  (t test),  or  f_u := U + m D + 3 S_u < 0
// (for D >= -S_D that is case (-) of u: |D| <= D + 2 S_D and S_D <= S_u), or the same for v or w.  (f_T also takes in the
// reference's "parallel" exit where that reaches further, see the code.)  All four are affine in (a, b), so any
// combination h = l_i f_i + l_j f_j with weights >= 0 is affine too, and wherever h < 0 one of the two is negative -- the
// ray is rejected.  Choosing the weights so that h does not depend on b (l_i = |df_j/db|, l_j = |df_i/db| when the two
// slopes have opposite signs) gives "every ray with a < bound is rejected" or "... a > bound ..." for ALL b: the left or
// right end of the region where the two constraints can hold together -- for a triangle seen from the front, the a of
// the vertex where two of its (slightly grown) edges meet.  The tightest such bounds over the six pairs and the four
// single constraints, and the same with a and b exchanged, bound the pixels that can hit the triangle by a rectangle; the
// weights need not be exact (any weights >= 0 give a valid h; what is left of the slope in b and the roundings of the
// combination are added to h before the bound is taken).  Most triangles of a fine mesh end here, with one to four tiles.
//
// Marking.  A rectangle of at most 4 x 4 tiles is marked as it is (most triangles of a fine mesh end there: the wave's
// lanes put their bits together first and one lane marks for all).  Larger rectangles -- large, close or grazing
// triangles -- are queued for a second kernel that gives each sixteen lanes: lane i tests tile i of the rectangle with the
// block form of the test above, 16 tiles at a time; a rectangle of more than 1024 tiles is tested in blocks of 2^k x 2^k
// tiles and marked at that level of a bit pyramid, which k_cover_finish ORs down into level 0, the bitmap k_render reads.
#include "hvrt_internal.hpp"

#include <float.h>
#include <math.h>

namespace hvrt {
namespace {

constexpr float COVER_MARGIN = 4.8828125e-4f;      // 2^-11
constexpr float COVER_PAD = 9.5367431640625e-7f;   // 2^-20

struct F3 { float x, y, z; };
__device__ __forceinline__ F3 crs(F3 a, F3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
__device__ __forceinline__ float dt3(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

struct FrameConsts {
    F3 A, hor, ver, org;
    float K;            // S_c = K x P_c + 1e-30
    float det_scale;    // (max |D| over a block (g space) + S_D) x det_scale + det_round x |e1||e2| bounds the reference's |det|
    float det_round;
    int ok;
};

__device__ void frame_consts(const hvrt_camera_frame& f, FrameConsts& fc) {
    double A[3], h[3], v[3], Bv2 = 0.0;
    bool finite = true;
    for (int k = 0; k < 3; ++k) {
        const double ll = f.lower_left[k], o = f.origin[k];
        h[k] = f.horizontal[k]; v[k] = f.vertical[k];
        A[k] = ll - o;
        finite = finite && fabs(ll) <= 1e18 && fabs(o) <= 1e18 && fabs(h[k]) <= 1e18 && fabs(v[k]) <= 1e18;    // false on NaN
        const double Bk = 2.0 * fabs(ll) + 5.0 * fabs(h[k]) + 4.0 * fabs(v[k]) + (fabs(A[k]) + fabs(h[k]) + fabs(v[k]));
        Bv2 += Bk * Bk;
    }
    auto len = [](const double* p) { return sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]); };
    const double n[3] = { h[1] * v[2] - h[2] * v[1], h[2] * v[0] - h[0] * v[2], h[0] * v[1] - h[1] * v[0] };
    const double nn = len(n), lh = len(h), lv = len(v), lA = len(A);
    const double g2 = lA + lh + lv;
    bool ok = finite && nn > 1e-9 * lh * lv && lh > 0.0 && lv > 0.0;
    double gmin = 0.0, gmax = 0.0;
    if (ok) {
        gmin = fabs(A[0] * n[0] + A[1] * n[1] + A[2] * n[2]) / nn * (1.0 - 1e-9);
        ok = gmin > 1e-6 * g2 && gmin > 1e-18;
        for (int c = 0; c < 4; ++c) {
            const double g[3] = { A[0] + (c & 1) * h[0] + (c >> 1) * v[0], A[1] + (c & 1) * h[1] + (c >> 1) * v[1], A[2] + (c & 1) * h[2] + (c >> 1) * v[2] };
            gmax = fmax(gmax, len(g));
        }
    }
    const double ulp = 5.9604644775390625e-8;     // 2^-24
    const double eps_d = ok ? (2.0 * sqrt(Bv2) / gmin + 4.0) * ulp : 0.0;
    ok = ok && eps_d < 1e-3;
    fc.A = { (float)A[0], (float)A[1], (float)A[2] };
    fc.hor = { f.horizontal[0], f.horizontal[1], f.horizontal[2] };
    fc.ver = { f.vertical[0], f.vertical[1], f.vertical[2] };
    fc.org = { f.origin[0], f.origin[1], f.origin[2] };
    fc.K = (float)(2.0 * (1.01 * gmax * (14.0 * ulp + eps_d) + 16.0 * ulp * g2) * 1.001);
    fc.det_scale = ok ? (float)(1.001 / gmin) : 0.0f;
    fc.det_round = (float)((eps_d + 8.0 * ulp) * 1.01);
    fc.ok = ok ? 1 : 0;
}

__device__ __forceinline__ void mark(uint32_t* __restrict__ bits, const CoverArgs& ca, uint32_t k, uint32_t x, uint32_t y) {
    const uint32_t nx = x >> k, ny = y >> k;
    uint32_t* const w = bits + ((uint64_t)k * ca.tiles_y + ny) * ca.words_x + (nx >> 5);
    const uint32_t b = 1u << (nx & 31u);
    atomicOr(w, b);       // (no look first: a wave that waits for the answer before every mark spends its time waiting)
}

__device__ __forceinline__ void or_word(uint32_t* __restrict__ w, uint32_t b, const CoverArgs& ca) {
    // (no look first: the wave has already put its lanes' bits together, and nothing waits for the result; with a look the
    // first pass over config 2 took 90 us instead of 62)
    atomicOr(w, b);
}
// OR over the 64 lanes (all must be active): two quad swaps, the two row mirrors, then the four rows' values
__device__ __forceinline__ uint32_t wave_or(uint32_t v) {
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);   // row_half_mirror
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);   // row_mirror
    return (uint32_t)(__builtin_amdgcn_readlane((int)v, 0) | __builtin_amdgcn_readlane((int)v, 16) |
                      __builtin_amdgcn_readlane((int)v, 32) | __builtin_amdgcn_readlane((int)v, 48));
}

// SLOW = false: one lane per triangle; the triangles whose rectangle is larger than 4 x 4 tiles are appended to the frame's
// queue (as far as it has room).  SLOW = true: sixteen lanes per queue entry, which share out the rectangle's tiles.
template <bool SLOW>
__device__ __forceinline__ void cover_triangle(const float4* __restrict__ tri_rec, const CoverArgs& ca, const FrameConsts& fc_s,
                                               uint32_t* __restrict__ bits, uint32_t L, uint32_t fi, uint32_t tri, bool have) {
    // what is left to do for this lane's triangle: 0 nothing, 1 the tiles of a rectangle of at most 4 x 4, 2 the general case
    int todo = have ? 2 : 0;
    const F3 A = fc_s.A, hor = fc_s.hor, ver = fc_s.ver;
    const float4 r0 = tri_rec[3ull * tri], r1 = tri_rec[3ull * tri + 1];
    const float e2z = reinterpret_cast<const float*>(tri_rec + 3ull * tri + 2)[0];
    const F3 v0 = { r0.x, r0.y, r0.z }, e1 = { r0.w, r1.x, r1.y }, e2 = { r1.z, r1.w, e2z };
    const F3 s = { fc_s.org.x - v0.x, fc_s.org.y - v0.y, fc_s.org.z - v0.z };      // GeometryUtils.cpp:73
    // Euclidean lengths, rounded up
    const float ls = sqrtf(dt3(s, s)) * 1.000001f, l1 = sqrtf(dt3(e1, e1)) * 1.000001f, l2 = sqrtf(dt3(e2, e2)) * 1.000001f;
    const float PD = l1 * l2, PU = ls * l2 + PD, PV = ls * l1 + PD;
    const float K = fc_s.K;
    const float SD = K * PD + 1e-30f, SU = K * PU + 1e-30f, SV = K * PV + 1e-30f, SW = SU + SV;
    if (todo && !(SW < 1e30f)) { mark(bits, ca, L, 0u, 0u); todo = 0; }      // overflow / NaN in the record: not covered by the analysis
    const F3 N = crs(e2, e1), M = crs(e2, s), Q = crs(s, e1);
    // the forms as affine functions of (a, b): value = X0 + Xa a + Xb b
    const float D0 = dt3(N, A), Da = dt3(N, hor), Db = dt3(N, ver);
    const float U0 = dt3(M, A), Ua = dt3(M, hor), Ub = dt3(M, ver);
    const float V0 = dt3(Q, A), Va = dt3(Q, hor), Vb = dt3(Q, ver);
    const float UDa = Ua - Da, UDb = Ub - Db, VDa = Va - Da, VDb = Vb - Db;      // u - 1, v - 1 (times D)
    const float Wa = (Da - Ua) - Va, Wb = (Db - Ub) - Vb;                        // w
    const float UVa = Ua + Va, UVb = Ub + Vb;                                    // -(w - 1)
    const float det_round = fc_s.det_round * PD, det_scale = fc_s.det_scale;
    // the sign of t's numerator, where it is beyond its own rounding
    const float E = -dt3(s, N), e_round = 1.9073486328125e-6f * (ls * l1 * l2);  // 2^-19
    const float esign = E > e_round ? 1.0f : (E < -e_round ? -1.0f : 0.0f);
    const float inv_w = 1.0f / (float)ca.width, inv_h = 1.0f / (float)ca.height;

    // ---- the rectangle of tiles outside which every ray is rejected (see "The bounding rectangle" above)
    // (where the sign of E is not known -- the eye lies in the triangle's plane, within rounding -- both signs are taken in
    // turn, each without what the other one covers, and the rectangle is the hull of the two)
    uint32_t tx_lo = ca.tiles_x, tx_hi = 0u, ty_lo = ca.tiles_y, ty_hi = 0u;      // empty
    auto rectangle = [&](const float sg) {
        const float m = COVER_MARGIN;
        const float W0 = (D0 - U0) - V0;
        // f_T, f_u, f_v, f_w as c0 + ca a + cb b
        // f_T: the ray is rejected where D < -S_D (t test) and where |det| < EPS (the reference's first exit), which holds for
        // |D| < p_thr (see rej_par below); the two ranges join when p_thr >= S_D, and then "D < p_thr" says it all
        const float p_thr = (0.99f * FLT_EPSILON - det_round) / det_scale - SD;
        const float t0 = p_thr >= SD ? -p_thr : SD;
        const float c0[4] = { sg * D0 + t0, sg * (U0 + m * D0) + 3.0f * SU, sg * (V0 + m * D0) + 3.0f * SV, sg * (W0 + m * D0) + 3.0f * SW };
        const float ka[4] = { sg * Da, sg * (Ua + m * Da), sg * (Va + m * Da), sg * (Wa + m * Da) };
        const float kb[4] = { sg * Db, sg * (Ub + m * Db), sg * (Vb + m * Db), sg * (Wb + m * Db) };
        float mag[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) mag[i] = COVER_PAD * ((fabsf(c0[i]) + fabsf(ka[i])) + fabsf(kb[i]));
        float a_lo = 0.0f, a_hi = 1.0f, b_lo = 0.0f, b_hi = 1.0f;
        // h = h0 + hp p + (what is left of the other coordinate's slope, at its worst in [0, 1]) < 0  =>  rejected
        auto bound = [](float h0, float hp, float rest, float err, float& lo, float& hi) {
            const float H = h0 + fmaxf(rest, 0.0f) + err;
            const float q = -H / hp;
            if (hp > 0.0f) { if (q > lo) lo = q; }
            else if (hp < 0.0f) { if (q < hi) hi = q; }
            else if (H < 0.0f) lo = __builtin_inff();
        };
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bound(c0[i], ka[i], kb[i], mag[i], a_lo, a_hi);
            bound(c0[i], kb[i], ka[i], mag[i], b_lo, b_hi);
#pragma unroll
            for (int j = i + 1; j < 4; ++j) {
                if (kb[i] * kb[j] < 0.0f) {        // slopes in b of opposite signs: a bound on a
                    const float li = fabsf(kb[j]), lj = fabsf(kb[i]);
                    bound(li * c0[i] + lj * c0[j], li * ka[i] + lj * ka[j], li * kb[i] + lj * kb[j], li * mag[i] + lj * mag[j], a_lo, a_hi);
                }
                if (ka[i] * ka[j] < 0.0f) {
                    const float li = fabsf(ka[j]), lj = fabsf(ka[i]);
                    bound(li * c0[i] + lj * c0[j], li * kb[i] + lj * kb[j], li * ka[i] + lj * ka[j], li * mag[i] + lj * mag[j], b_lo, b_hi);
                }
            }
        }
        // pixels x with [x, x + 1] / W inside [a_lo, a_hi] (a pixel's samples lie in that interval, ends included), padded
        const float wf = (float)ca.width, hf = (float)ca.height;
        const float x_lo = fmaxf(a_lo * wf - 1.0f - (0.015625f + COVER_PAD * 4.0f * wf), 0.0f), x_hi = fminf(a_hi * wf + (0.015625f + COVER_PAD * 4.0f * wf), wf);
        const float y_lo = fmaxf(b_lo * hf - 1.0f - (0.015625f + COVER_PAD * 4.0f * hf), 0.0f), y_hi = fminf(b_hi * hf + (0.015625f + COVER_PAD * 4.0f * hf), hf);
        if (x_lo <= x_hi && y_lo <= y_hi) {
            const uint32_t xl = (uint32_t)x_lo >> 3, yl = (uint32_t)y_lo >> 3;
            uint32_t xh = (uint32_t)x_hi >> 3, yh = (uint32_t)y_hi >> 3;
            xh = xh < ca.tiles_x - 1u ? xh : ca.tiles_x - 1u; yh = yh < ca.tiles_y - 1u ? yh : ca.tiles_y - 1u;
            if (xl <= xh && yl <= yh) {
                tx_lo = xl < tx_lo ? xl : tx_lo; tx_hi = xh > tx_hi ? xh : tx_hi;
                ty_lo = yl < ty_lo ? yl : ty_lo; ty_hi = yh > ty_hi ? yh : ty_hi;
            }
        }
    };
    if (todo) rectangle(esign != 0.0f ? esign : 1.0f);
    if (todo && esign == 0.0f) rectangle(-1.0f);
    if (todo) {
        if (tx_lo > tx_hi || ty_lo > ty_hi) todo = 0;      // no pixel at all
        else if (tx_hi - tx_lo < 4u && ty_hi - ty_lo < 4u) todo = 1;
    }

    // ---- the small rectangles of the wave's 64 triangles, put together: neighbouring triangles mark the same few tiles, and
    // 64 lanes x 4 atomics on one or two words queue up at the L2 (config 2: 0.45 ms for the pass that way, 0.1 without marks).
    // Row by row (a rectangle has at most four): the lanes whose bits fall into the same pair of words OR them together and
    // one lane marks for all.
#pragma unroll 1
    for (uint32_t r = 0; r < 4u; ++r) {
        const uint32_t yy = ty_lo + r;
        const bool has = todo == 1 && yy <= ty_hi;
        uint64_t pend = __builtin_amdgcn_ballot_w64(has);
        if (!pend) break;                                               // rows only get fewer
        const uint32_t key = yy * ca.words_x + (tx_lo >> 5);         // level 0: word (tx_lo / 32) of row yy
        const uint64_t m64 = (uint64_t)((1u << (tx_hi - tx_lo + 1u)) - 1u) << (tx_lo & 31u);
        uint32_t my_lo = 0u, my_hi = 0u;      // a leader's words: what its group marks
        while (pend) {
            const int leader = __builtin_ctzll(pend);
            const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)key, leader);
            const bool mine = has && key == k0;
            const uint32_t lo = wave_or(mine ? (uint32_t)m64 : 0u), hi = wave_or(mine ? (uint32_t)(m64 >> 32) : 0u);
            if ((int)(threadIdx.x & 63u) == leader) { my_lo = lo; my_hi = hi; }
            pend &= ~__builtin_amdgcn_ballot_w64(mine);
        }
        // the leaders mark, all at once
        if (my_lo) or_word(bits + key, my_lo, ca);
        if (my_hi) or_word(bits + key + 1u, my_hi, ca);
    }
    if (!SLOW) {
        // the general case goes to the queue, one reservation per wave; what does not fit stays here
        const uint64_t slow = __builtin_amdgcn_ballot_w64(todo == 2);
        if (slow) {
            const int lane = (int)(threadIdx.x & 63u), first = __builtin_ctzll(slow);
            uint32_t base = 0u;
            if (lane == first) base = atomicAdd(ca.slow_count + fi, (uint32_t)__builtin_popcountll(slow));
            base = (uint32_t)__builtin_amdgcn_readlane((int)base, first);
            const uint32_t at = base + (uint32_t)__builtin_popcountll(slow & ((1ull << lane) - 1ull));
            if (todo == 2 && at < ca.slow_cap) { ca.slow_queue[(uint64_t)fi * ca.slow_cap + at] = tri; todo = 0; }
        }
    }
    if (!SLOW && todo != 2) return;      // (the second pass's lanes stay: they put their marks together, like above)

    // ---- may a ray through the block of tiles [x, x + 2^k) x [y, y + 2^k) hit the triangle?
    auto block_test = [&](uint32_t x, uint32_t y, uint32_t k) -> bool {
        const float half = (float)(4u << k);
        const float ca_ = ((float)(x * 8u) + half) * inv_w, cb_ = ((float)(y * 8u) + half) * inv_h;
        // (padded against the roundings of x / W here and in k_render: 2^-20 of the largest coordinate in play)
        float ha = half * inv_w, hb = half * inv_h;
        ha += COVER_PAD * (1.0f + ca_ + ha); hb += COVER_PAD * (1.0f + cb_ + hb);
        const float cD = D0 + (Da * ca_ + Db * cb_), rD = fabsf(Da) * ha + fabsf(Db) * hb;
        const float cU = U0 + (Ua * ca_ + Ub * cb_), rU = fabsf(Ua) * ha + fabsf(Ub) * hb;
        const float cV = V0 + (Va * ca_ + Vb * cb_), rV = fabsf(Va) * ha + fabsf(Vb) * hb;
        const float cUD = cU - cD, rUD = fabsf(UDa) * ha + fabsf(UDb) * hb;
        const float cVD = cV - cD, rVD = fabsf(VDa) * ha + fabsf(VDb) * hb;
        const float cW = (cD - cU) - cV, rW = fabsf(Wa) * ha + fabsf(Wb) * hb;
        const float cUV = cU + cV, rUV = fabsf(UVa) * ha + fabsf(UVb) * hb;
        const float maxD = fabsf(cD) + rD;
        const float mD = COVER_MARGIN * maxD;
        const float mu = mD + SU, mv = mD + SV, mw = mD + SW;
        const bool rej_u = (cU - rU > mu && cUD - rUD > mu) || (cU + rU < -mu && cUD + rUD < -mu);
        const bool rej_v = (cV - rV > mv && cVD - rVD > mv) || (cV + rV < -mv && cVD + rVD < -mv);
        const bool rej_w = (cW - rW > mw && -cUV - rUV > mw) || (cW + rW < -mw && -cUV + rUV < -mw);
        const bool rej_par = (maxD + SD) * det_scale + det_round < 0.99f * FLT_EPSILON;
        const bool rej_t = esign != 0.0f && esign * cD + rD < -SD;
        return !(rej_u || rej_v || rej_w || rej_par || rej_t);
    };
    // the rectangle in blocks of 2^k x 2^k tiles, k the smallest for which there are at most `room` of them
    const uint32_t room = SLOW ? 1024u : 64u;
    uint32_t k = 0u, bx0 = 0u, by0 = 0u, bw = 1u, n_blocks = 0u;
    if (todo == 2) {
        while (((tx_hi >> k) - (tx_lo >> k) + 1u) * ((ty_hi >> k) - (ty_lo >> k) + 1u) > room) ++k;
        bx0 = tx_lo >> k; by0 = ty_lo >> k; bw = (tx_hi >> k) - bx0 + 1u; n_blocks = bw * ((ty_hi >> k) - by0 + 1u);
    }
    if (SLOW) {
        // the sixteen lanes' triangle: lane i of them takes blocks i, i + 16, ... -- neighbours in a row of the rectangle, so
        // their marks fall into the same word or two, and the wave's four triangles are neighbours as well: the lanes whose
        // bits belong to one word OR them together and one lane marks for all.  (One atomic per block that passes was 48 us
        // for config 2's 11 000 queued triangles from the front and 100 us from 8 degrees beside it, with the same number of
        // triangles and three times the marks: the silhouette's few hundred words took them one at a time.)
        for (uint32_t t = threadIdx.x & 15u; __builtin_amdgcn_ballot_w64(t < n_blocks) != 0ull; t += 16u) {
            bool pass = false;
            uint32_t key = 0u, bit = 0u;
            if (t < n_blocks) {
                const uint32_t bx = bx0 + t % bw, by = by0 + t / bw;
                pass = block_test(bx << k, by << k, k);
                key = (k * ca.tiles_y + by) * ca.words_x + (bx >> 5);      // (mark()'s word, bx and by being block numbers of level k)
                bit = 1u << (bx & 31u);
            }
            uint64_t pend = __builtin_amdgcn_ballot_w64(pass);
            uint32_t my = 0u;
            while (pend) {
                const int leader = __builtin_ctzll(pend);
                const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)key, leader);
                const bool mine = pass && key == k0;
                const uint32_t m = wave_or(mine ? bit : 0u);
                if ((int)(threadIdx.x & 63u) == leader) my = m;
                pend &= ~__builtin_amdgcn_ballot_w64(mine);
            }
            if (my) or_word(bits + key, my, ca);
        }
    } else {
        // (the queue was full: the lane's triangle, block by block)
        for (uint32_t t = 0; t < n_blocks; ++t) {
            const uint32_t bx = bx0 + t % bw, by = by0 + t / bw;
            if (block_test(bx << k, by << k, k)) mark(bits, ca, k, bx << k, by << k);
        }
    }
}

// the frames' constants, once per launch (double-precision square roots and divides: not once per workgroup)
__global__ __launch_bounds__(64) void k_cover_frames(CoverArgs ca) {
    const uint32_t fi = blockIdx.x * 64u + threadIdx.x;
    if (fi < (uint32_t)ca.n_frames) frame_consts(ca.frames ? ca.frames[fi] : ca.frame0, reinterpret_cast<FrameConsts*>(ca.consts)[fi]);
}

template <bool SLOW>
__global__ __launch_bounds__(256) void k_cover(const float4* __restrict__ tri_rec, uint32_t n_tris, CoverArgs ca) {
    const uint32_t fi = blockIdx.y;
    uint32_t n_queued = 0u;
    if (SLOW) {
        n_queued = ca.slow_count[fi];
        if (n_queued > ca.slow_cap) n_queued = ca.slow_cap;
        if (blockIdx.x * 16u >= n_queued) return;
    }
    const FrameConsts& fc_s = reinterpret_cast<const FrameConsts*>(ca.consts)[fi];
    uint32_t* const bits = ca.bits + (uint64_t)fi * ca.frame_words;
    const uint32_t L = (uint32_t)ca.levels;
    if (!fc_s.ok) {
        if (!SLOW && blockIdx.x == 0 && threadIdx.x == 0) mark(bits, ca, L, 0u, 0u);
        return;
    }
    if (SLOW) {
        // sixteen lanes per queue entry (they work out the same triangle and share out its rectangle's tiles), four entries per
        // wave, the grid's waves striding over the queue
        for (uint32_t first = (blockIdx.x * 4u + (threadIdx.x >> 6)) * 4u; first < n_queued; first += gridDim.x * 16u) {
            const uint32_t own = first + ((threadIdx.x & 63u) >> 4);
            uint32_t tri = own < n_queued ? ca.slow_queue[(uint64_t)fi * ca.slow_cap + own] : 0u;
            if (tri >= n_tris) tri = n_tris - 1u;
            cover_triangle<true>(tri_rec, ca, fc_s, bits, L, fi, tri, own < n_queued);
        }
    } else {
        // a lane per triangle (no lane leaves before the wave has put its marks together; a lane without a triangle computes
        // on the last one)
        const uint32_t own = blockIdx.x * 256u + threadIdx.x;
        cover_triangle<false>(tri_rec, ca, fc_s, bits, L, fi, own < n_tris ? own : n_tris - 1u, own < n_tris);
    }
}

// level 0 of each frame's pyramid |= every level above it
__global__ __launch_bounds__(256) void k_cover_finish(CoverArgs ca) {
    const uint32_t fi = blockIdx.y;
    const uint32_t n_words = ca.tiles_y * ca.words_x;
    const uint32_t wi = blockIdx.x * 256u + threadIdx.x;
    if (wi >= n_words) return;
    uint32_t* const bits = ca.bits + (uint64_t)fi * ca.frame_words;
    const uint32_t ty = wi / ca.words_x, wx = wi - ty * ca.words_x;
    uint32_t acc = bits[wi];
    for (uint32_t k = 1; k <= (uint32_t)ca.levels && acc != 0xFFFFFFFFu; ++k) {
        const uint32_t* const row = bits + ((uint64_t)k * ca.tiles_y + (ty >> k)) * ca.words_x;
        const uint32_t first = (wx * 32u) >> k;             // the block of this word's first tile at level k
        if (k >= 5u) {
            if ((row[first >> 5] >> (first & 31u)) & 1u) acc = 0xFFFFFFFFu;
        } else {
            // 32 >> k blocks, inside one word (first is a multiple of 32 >> k)
            const uint32_t src = row[first >> 5] >> (first & 31u);
            for (uint32_t b = 0; b < 32u; ++b) acc |= ((src >> (b >> k)) & 1u) << b;
        }
    }
    bits[wi] = acc;
}

}  // namespace

int cover_geometry(int width, int height, int n_frames, uint32_t n_tris, CoverArgs& ca) {
    ca.width = width; ca.height = height; ca.n_frames = n_frames;
    ca.tiles_x = ((uint32_t)width + 7u) >> 3;
    ca.tiles_y = ((uint32_t)height + 7u) >> 3;
    ca.words_x = (ca.tiles_x + 31u) >> 5;
    int L = 0;
    while ((1u << L) < ca.tiles_x || (1u << L) < ca.tiles_y) ++L;
    ca.levels = L;
    const uint64_t frame_words = (uint64_t)(L + 1) * ca.tiles_y * ca.words_x;
    if (L > 20 || frame_words > 0x7FFFFFFFull) return -1;
    ca.frame_words = (uint32_t)frame_words;
    // the queue of triangles for the general case: an eighth of the mesh per frame, within 4 K .. 1 M entries (what does not
    // fit is handled where it arises)
    uint32_t cap = n_tris / 8u;
    cap = cap < 4096u ? 4096u : (cap > (1u << 20) ? (1u << 20) : cap);
    ca.slow_cap = cap;
    return 0;
}
static_assert(sizeof(FrameConsts) == 64, "16 words per frame in the cover's allocation");
uint64_t cover_words(const CoverArgs& ca) { return cover_zero_words(ca) + (uint64_t)ca.n_frames * ca.slow_cap + (uint64_t)ca.n_frames * 16u + 16u; }
uint64_t cover_zero_words(const CoverArgs& ca) { return (uint64_t)ca.n_frames * ca.frame_words + (((uint64_t)ca.n_frames + 63u) & ~63ull); }
void cover_bind(CoverArgs& ca, uint32_t* words) {
    ca.bits = words;
    ca.slow_count = words + (uint64_t)ca.n_frames * ca.frame_words;
    ca.slow_queue = words + cover_zero_words(ca);
    uint32_t* const behind = ca.slow_queue + (uint64_t)ca.n_frames * ca.slow_cap;
    ca.consts = behind + ((16u - (uint32_t)((uintptr_t)behind / 4u % 16u)) % 16u);      // 64-byte aligned
}

hipError_t launch_cover(const SceneDev& sc, const CoverArgs& ca, hipStream_t s) {
    if (sc.n_tris == 0 || ca.n_frames < 1) return hipSuccess;
    const dim3 grid((sc.n_tris + 255u) / 256u, (unsigned)ca.n_frames);
    k_cover_frames<<<((unsigned)ca.n_frames + 63u) / 64u, 64, 0, s>>>(ca);
    k_cover<false><<<grid, 256, 0, s>>>(sc.tri_rec, sc.n_tris, ca);
    const uint32_t slow_blocks = (ca.slow_cap + 15u) / 16u;      // sixteen lanes per queue entry, at most 2048 workgroups striding over the queue
    const dim3 grid_slow(slow_blocks < 2048u ? slow_blocks : 2048u, (unsigned)ca.n_frames);
    k_cover<true><<<grid_slow, 256, 0, s>>>(sc.tri_rec, sc.n_tris, ca);
    const uint32_t n_words = ca.tiles_y * ca.words_x;
    const dim3 grid2((n_words + 255u) / 256u, (unsigned)ca.n_frames);
    k_cover_finish<<<grid2, 256, 0, s>>>(ca);
    return hipGetLastError();
}

}  // namespace hvrt
