// Micro-benchmark for the host-facing render path (hvrt_render): how a 4K float frame (99.5 MB) gets from HBM into the
// caller's pageable buffer fastest.  Measures: hipHostRegister / Unregister of that buffer, a direct DMA into it once
// registered, the staged path (pinned ring + host memcpy) with 1..8 copy threads, and plain hipMemcpy to pageable memory.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t bytes = 3840ull * 2160 * 12;
    char* d;
    { double t0 = now(); hipMalloc(&d, bytes); double t1 = now(); printf("hipMalloc of %.1f MB: %.2f ms (first)\n", bytes / 1e6, t1 - t0);
      char* d2; t0 = now(); hipMalloc(&d2, bytes); t1 = now(); printf("hipMalloc of %.1f MB: %.2f ms (second)\n", bytes / 1e6, t1 - t0);
      t0 = now(); hipFree(d2); t1 = now(); printf("hipFree: %.2f ms\n", t1 - t0);
      char* d3; t0 = now(); hipMalloc(&d3, 33u << 20); t1 = now(); printf("hipMalloc of 34.6 MB: %.2f ms\n", t1 - t0);
      char* hp; t0 = now(); hipHostMalloc((void**)&hp, 32u << 20, hipHostMallocDefault); t1 = now(); printf("hipHostMalloc of 32 MiB: %.2f ms\n", t1 - t0); }
    hipMemset(d, 1, bytes);
    hipStream_t s; hipStreamCreate(&s);
    for (int round = 0; round < 2; ++round) {
        std::vector<char> host(bytes);            // touched (zero-filled) pageable memory, like a std::vector<Vec3>
        double t0 = now(); hipMemcpy(host.data(), d, bytes, hipMemcpyDeviceToHost); double t1 = now();
        printf("round %d: plain hipMemcpy to pageable: %.2f ms (%.1f GB/s)\n", round, t1 - t0, bytes / (t1 - t0) / 1e6);
        t0 = now(); hipError_t e = hipHostRegister(host.data(), bytes, hipHostRegisterDefault); t1 = now();
        printf("  hipHostRegister: %.2f ms (%s)\n", t1 - t0, hipGetErrorString(e));
        for (int k = 0; k < 2; ++k) {
            t0 = now(); hipMemcpyAsync(host.data(), d, bytes, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); t1 = now();
            printf("  DMA into the registered buffer: %.2f ms (%.1f GB/s)\n", t1 - t0, bytes / (t1 - t0) / 1e6);
        }
        t0 = now(); hipHostUnregister(host.data()); t1 = now();
        printf("  hipHostUnregister: %.2f ms\n", t1 - t0);
    }
    // staged: ring of pinned chunks, T copy threads
    const size_t CH = 4u << 20; const int R = 8;
    char* pin; hipHostMalloc((void**)&pin, CH * R, hipHostMallocDefault);
    hipEvent_t ev[R]; for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    std::vector<char> host(bytes);
    for (int T : {1, 2, 4, 8}) {
        for (int rep = 0; rep < 2; ++rep) {
            const size_t n = (bytes + CH - 1) / CH;
            double t0 = now();
            // each thread owns chunks k % T == t; the issuing thread keeps R chunks in flight
            std::vector<std::thread> th;
            std::vector<int> issued(n, 0);
            size_t next_issue = 0;
            auto issue = [&](size_t k) { size_t sz = k + 1 < n ? CH : bytes - k * CH; hipMemcpyAsync(pin + (k % R) * CH, d + k * CH, sz, hipMemcpyDeviceToHost, s); hipEventRecord(ev[k % R], s); };
            for (; next_issue < (size_t)R && next_issue < n; ++next_issue) issue(next_issue);
            // simple scheme: main thread waits for chunk k, hands the memcpy to T threads by splitting the chunk in T pieces
            for (size_t k = 0; k < n; ++k) {
                hipEventSynchronize(ev[k % R]);
                size_t sz = k + 1 < n ? CH : bytes - k * CH;
                if (T == 1) memcpy(host.data() + k * CH, pin + (k % R) * CH, sz);
                else {
                    th.clear();
                    size_t piece = (sz + T - 1) / T;
                    for (int t = 0; t < T; ++t) { size_t o = t * piece; if (o >= sz) break; size_t l = std::min(piece, sz - o);
                        th.emplace_back([=, &host] { memcpy(host.data() + k * CH + o, pin + (k % R) * CH + o, l); }); }
                    for (auto& x : th) x.join();
                }
                if (next_issue < n) { issue(next_issue); ++next_issue; }
            }
            double t1 = now();
            if (rep) printf("staged, %d copy thread(s): %.2f ms (%.1f GB/s)\n", T, t1 - t0, bytes / (t1 - t0) / 1e6);
        }
    }
    return 0;
}
