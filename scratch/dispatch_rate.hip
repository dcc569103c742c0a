// dispatch_rate.hip: how fast does the GPU start workgroups?  hipcc --offload-arch=gfx950 -O3 -o dispatch_rate dispatch_rate.hip
// Kernels that do (almost) nothing, launched with the warp kernels' grid (32 640 workgroups of 256 threads) and their
// LDS / register footprint: the floor under any one-tile-per-workgroup kernel of that grid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int LDS, int NT>
__global__ __launch_bounds__(NT) void k_empty(uint32_t* out, int mode) {
    __shared__ uint32_t s[LDS / 4 > 0 ? LDS / 4 : 1];
    if (mode == 99) { s[threadIdx.x] = 1; __syncthreads(); out[0] = s[0]; }    // keeps the LDS allocation
}
template <int LDS, int NT>
__global__ __launch_bounds__(NT) void k_store(uint32_t* out, int rows) {        // rows x (NT x 4 bytes) per workgroup, streaming stores
    __shared__ uint32_t s[LDS / 4 > 0 ? LDS / 4 : 1];
    if (rows == 99) { s[threadIdx.x] = 1; __syncthreads(); out[0] = s[0]; }
    uint32_t* p = out + (size_t)blockIdx.x * NT * rows + threadIdx.x;
    for (int r = 0; r < rows; r++) __builtin_nontemporal_store((uint32_t)r, p + (size_t)r * NT);
}
template <int LDS, int NT>
__global__ __launch_bounds__(NT) void k_copy(const uint4* in, uint4* out, int chunks) {   // chunks x 16 bytes per lane, load then store
    __shared__ uint32_t s[LDS / 4 > 0 ? LDS / 4 : 1];
    if (chunks == 99) { s[threadIdx.x] = 1; __syncthreads(); out[0].x = s[0]; }
    const size_t base = (size_t)blockIdx.x * NT * chunks + threadIdx.x;
    uint4 v[8];
    for (int r = 0; r < chunks; r++) v[r] = in[base + (size_t)r * NT];
    for (int r = 0; r < chunks; r++) out[base + (size_t)r * NT] = v[r];
}
template <typename F> float timeit(F f, int iters = 20) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) f();
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; i++) f();
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / iters;
}
int main() {
    uint32_t* buf; hipMalloc(&buf, (size_t)1 << 30);
    uint32_t* buf2; hipMalloc(&buf2, (size_t)1 << 30);
    hipMemset(buf, 1, (size_t)1 << 30); hipMemset(buf2, 1, (size_t)1 << 30);
    const int G = 32640;
    printf("grid %d workgroups\n", G);
    printf("empty, 256 threads, no LDS      : %7.1f us\n", timeit([&] { hipLaunchKernelGGL((k_empty<0, 256>), dim3(G), dim3(256), 0, 0, buf, 0); }));
    printf("empty, 256 threads, 22 KB LDS   : %7.1f us\n", timeit([&] { hipLaunchKernelGGL((k_empty<22528, 256>), dim3(G), dim3(256), 0, 0, buf, 0); }));
    printf("empty, 64 threads x 4x grid     : %7.1f us\n", timeit([&] { hipLaunchKernelGGL((k_empty<0, 64>), dim3(4 * G), dim3(64), 0, 0, buf, 0); }));
    printf("empty, 1024 threads, grid/4     : %7.1f us\n", timeit([&] { hipLaunchKernelGGL((k_empty<0, 1024>), dim3(G / 4), dim3(1024), 0, 0, buf, 0); }));
    for (int rows : {8, 24}) {
        printf("store %2d x 1 KB per WG (%5.1f MB), 256 thr, no LDS  : %7.1f us\n", rows, G * rows * 1024 / 1e6, timeit([&] { hipLaunchKernelGGL((k_store<0, 256>), dim3(G), dim3(256), 0, 0, buf, rows); }));
        printf("store %2d x 1 KB per WG, 22 KB LDS                   : %7.1f us\n", rows, timeit([&] { hipLaunchKernelGGL((k_store<22528, 256>), dim3(G), dim3(256), 0, 0, buf, rows); }));
    }
    printf("store 32 x 1 KB per WG, grid/4 (same bytes as 8 rows): %7.1f us\n", timeit([&] { hipLaunchKernelGGL((k_store<0, 256>), dim3(G / 4), dim3(256), 0, 0, buf, 32); }));
    for (int ch : {2, 4, 8}) {
        const int g = (int)(((size_t)132 << 20) / (256 * 16 * ch));
        printf("copy 132 MB: %d chunks of 16 B per lane, %6d WGs, no LDS: %7.1f us", ch, g, timeit([&] { hipLaunchKernelGGL((k_copy<0, 256>), dim3(g), dim3(256), 0, 0, (const uint4*)buf2, (uint4*)buf, ch); }));
        printf("   22 KB LDS: %7.1f us\n", timeit([&] { hipLaunchKernelGGL((k_copy<22528, 256>), dim3(g), dim3(256), 0, 0, (const uint4*)buf2, (uint4*)buf, ch); }));
    }
    return 0;
}
