// Micro-benchmark: cost of one vector load per wave in the CU's texture path (TA -> L1 -> TD) as a function
// of width (4/8/16 B per lane) and of how many distinct addresses the 64 lanes name.  Footprint stays
// L1/L2 resident.  Prints clocks per wave-load per CU (all 4 SIMDs loaded).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int W> struct Vec;
template <> struct Vec<1> { using T = uint32_t; };
template <> struct Vec<2> { using T = uint2; };
template <> struct Vec<4> { using T = uint4; };
__device__ inline uint32_t fold(uint32_t v) { return v; }
__device__ inline uint32_t fold(uint2 v) { return v.x ^ v.y; }
__device__ inline uint32_t fold(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

// groups: lanes are split into `groups` runs of adjacent lanes; every run names one address per iteration.
// active: only lanes with (lane % 64) < active issue loads
template <int W>
__global__ __launch_bounds__(256) void k(const uint32_t* __restrict__ buf, uint32_t mask_words, int groups, int active,
                                         int iters, uint32_t stride_words, uint32_t* out) {
    using T = typename Vec<W>::T;
    const int lane = threadIdx.x & 63;
    const int grp = lane * groups / 64;
    uint32_t acc = 0;
    uint32_t pos = (blockIdx.x * 977u + (threadIdx.x >> 6) * 131u) * 64u + (uint32_t)grp * stride_words;
    if (lane < active) {
        for (int i = 0; i < iters; ++i) {
            const uint32_t w = (pos & mask_words) & ~3u;
            T v = *reinterpret_cast<const T*>(buf + w);
            acc ^= fold(v);
            pos += 12u + (acc & 0u);   // 48-byte records; dependent on the loaded value only formally
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const size_t words = 1u << 18;   // 1 MiB
    uint32_t *buf, *out;
    hipMalloc(&buf, words * 4 + 64); hipMalloc(&out, 64);
    hipMemset(buf, 1, words * 4 + 64);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double mhz = p.clockRate / 1000.0;
    printf("CUs %d clock %.0f MHz\n", cus, mhz);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 4000;
    const int blocks = cus * 8;   // 32 waves per CU
    auto run = [&](int W, int groups, int active, uint32_t stride) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            if (W == 1) k<1><<<blocks, 256>>>(buf, words - 1, groups, active, iters, stride, out);
            if (W == 2) k<2><<<blocks, 256>>>(buf, words - 1, groups, active, iters, stride, out);
            if (W == 4) k<4><<<blocks, 256>>>(buf, words - 1, groups, active, iters, stride, out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        const double wave_loads_per_cu = 32.0 * iters;
        const double clk = best * 1e-3 * mhz * 1e6 / wave_loads_per_cu;
        printf("W=%2d B  groups=%2d active=%2d stride=%5u B : %.3f ms  %.1f clk per wave-load per CU\n", W * 4, groups, active,
               stride * 4, best, clk);
    };
    for (int W : {1, 2, 4})
        for (int groups : {1, 4, 9, 16, 64}) run(W, groups, 64, 1200);
    for (int W : {1, 4})
        for (int active : {16, 32}) run(W, 64, active, 1200);
    // contiguous lanes (classic coalesced): stride = W words
    for (int W : {1, 2, 4}) run(W, 64, 64, (uint32_t)W);
    // sparse: every 4th lane active is not expressible with `active`; emulate with groups=64 and stride making quads share a line
    for (int W : {1, 4}) run(W, 64, 64, 12);   // adjacent lanes 48 B apart (consecutive records)
    return 0;
}
