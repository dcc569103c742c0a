// d2h_bw: device-to-host copy rate into page-locked memory, 1 / 8 / 64 MB per copy (the zoom stage brings 1 MB of bit mask per 4K frame to the host)
//   hipcc --offload-arch=gfx950 -O2 scratch/d2h_bw.hip -o scratch/d2h_bw && scratch/d2h_bw
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
int main() {
    const size_t MAXB = 64u << 20;
    void *d, *h;
    hipMalloc(&d, MAXB); hipHostMalloc(&h, MAXB, hipHostMallocDefault); hipMemset(d, 1, MAXB);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (size_t b : {1u << 20, 8u << 20, 64u << 20}) {
        const int n = (int)((512u << 20) / b);
        hipMemcpyAsync(h, d, b, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; i++) hipMemcpyAsync((char*)h + (i * b) % MAXB % (MAXB - b + 1), d, b, hipMemcpyDeviceToHost, st);
        hipStreamSynchronize(st);
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%3zu MB per copy: %.1f GB/s (%.1f us per copy)\n", b >> 20, n * (double)b / s / 1e9, s / n * 1e6);
    }
    return 0;
}
