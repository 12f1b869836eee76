// Micro-benchmark: what a plain streaming kernel reaches on this GPU's HBM for the access mixes of the voxel build --
// write only (the counters' memset), read only, copy, and the scan's mix (2 bytes read, 4 written per element, the
// same loads and stores per lane as k_scan<NARROW>: eight uint2 loads up front, eight uint4 stores).  The yardstick for
// "HBM-bound" kernels: the 8 TB/s of the data sheet is not what a kernel sees.
//   hipcc -O3 --offload-arch=gfx950 -o hbm_rate hbm_rate.hip && ./hbm_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_write(uint4* __restrict__ out, uint64_t n4) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * 256u)
        out[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ in, uint64_t n4, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * 256u) {
        const uint4 v = in[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ in, uint4* __restrict__ out, uint64_t n4) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * 256u) out[i] = in[i];
}
// one workgroup per 32768 elements, 16 waves of 2048, eight rows of 256 per wave: lane i holds elements 4i..4i+3 of a row
__global__ __launch_bounds__(1024) void k_widen(const uint2* __restrict__ in16, uint4* __restrict__ out32) {
    const uint64_t c = blockIdx.x;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint2* q = in16 + c * 8192u + w * 512 + lane;
    uint4* p = out32 + c * 8192u + w * 512 + lane;
    uint2 h[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) h[k] = q[k * 64];
#pragma unroll
    for (int k = 0; k < 8; ++k) p[k * 64] = make_uint4(h[k].x & 0xFFFFu, h[k].x >> 16, h[k].y & 0xFFFFu, h[k].y >> 16);
}

int main() {
    const uint64_t n = 1ull << 30;                       // elements (a 1024^3 grid)
    uint4 *a = nullptr, *b = nullptr;
    uint32_t* sink = nullptr;
    CK(hipMalloc((void**)&a, n * 4));                    // 4.3 GB
    CK(hipMalloc((void**)&b, n * 4));
    CK(hipMalloc((void**)&sink, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const unsigned grid = (unsigned)cus * 32u;
    auto timeit = [&](const char* what, double bytes, auto launch) -> int {
        float best = 1e30f;
        for (int r = 0; r < 6; ++r) {
            CK(hipEventRecord(e0, 0));
            launch();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0 && ms < best) best = ms;
        }
        printf("%-58s %7.3f ms  %6.2f TB/s\n", what, best, bytes / (best * 1e-3) / 1e12);
        return 0;
    };
    if (timeit("write only, 4.3 GB (uint4 stores)", n * 4.0, [&] { k_write<<<grid, 256>>>(a, n / 4); })) return 1;
    if (timeit("write only, 2.1 GB", n * 2.0, [&] { k_write<<<grid, 256>>>(a, n / 8); })) return 1;
    if (timeit("hipMemsetAsync, 2.1 GB", n * 2.0, [&] { (void)hipMemsetAsync(a, 0, n * 2, 0); })) return 1;
    if (timeit("read only, 4.3 GB (uint4 loads)", n * 4.0, [&] { k_read<<<grid, 256>>>(a, n / 4, sink); })) return 1;
    if (timeit("copy, 4.3 GB read + 4.3 GB written", n * 8.0, [&] { k_copy<<<grid, 256>>>(a, b, n / 4); })) return 1;
    if (timeit("the scan's mix: 2.1 GB read (uint2) + 4.3 GB written (uint4)", n * 6.0,
               [&] { k_widen<<<(unsigned)(n / 32768), 1024>>>((const uint2*)a, b); })) return 1;
    return 0;
}
