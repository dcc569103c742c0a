#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
struct __attribute__((aligned(4))) U3 { unsigned a, b, c; };
__global__ void copy16(const uint4* __restrict__ s, uint4* __restrict__ d, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) d[i] = s[i];
}
__global__ void copy12(const U3* __restrict__ s, U3* __restrict__ d, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) d[i] = s[i];
}
// tile pattern: WG = 128 px x 16 rows of a 1920x1080 BGR frame; lane = 4 px (12 B), 2 rows
__global__ void copy_tile(const unsigned char* __restrict__ s, unsigned char* __restrict__ d, int w, int h, size_t fb) {
    const unsigned char* sf = s + blockIdx.z * fb; unsigned char* df = d + blockIdx.z * fb;
    int tx = threadIdx.x % 32, ty = threadIdx.x / 32;
    int x = blockIdx.x * 128 + tx * 4;
    for (int r = 0; r < 2; r++) {
        int y = blockIdx.y * 16 + ty + 8 * r;
        if (y < h && x < w) {
            size_t o = (size_t)y * w * 3 + (size_t)x * 3;
            *(U3*)(df + o) = *(const U3*)(sf + o);
        }
    }
}
// wide tile: WG = 512 px x 4 rows
__global__ void copy_tile_wide(const unsigned char* __restrict__ s, unsigned char* __restrict__ d, int w, int h, size_t fb) {
    const unsigned char* sf = s + blockIdx.z * fb; unsigned char* df = d + blockIdx.z * fb;
    int tx = threadIdx.x % 128, ty = threadIdx.x / 128;
    int x = blockIdx.x * 512 + tx * 4;
    for (int r = 0; r < 2; r++) {
        int y = blockIdx.y * 4 + ty + 2 * r;
        if (y < h && x < w) {
            size_t o = (size_t)y * w * 3 + (size_t)x * 3;
            *(U3*)(df + o) = *(const U3*)(sf + o);
        }
    }
}
int main(int argc, char** argv) {
    const int W = 1920, H = 1080, B = argc > 1 ? atoi(argv[1]) : 16; size_t fb = (size_t)W * H * 3, total = fb * B;
    unsigned char *s, *d; hipMalloc(&s, total); hipMalloc(&d, total); hipMemset(s, 1, total);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto fn) {
        for (int i = 0; i < 5; i++) fn();
        hipEventRecord(e0); for (int i = 0; i < 50; i++) fn(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); double us = ms * 1e3 / 50;
        printf("%-16s %.2f us  %.1f GB/s (r+w)\n", name, us, 2.0 * total / us / 1e3);
    };
    timeit("copy16 grid2048", [&] { copy16<<<2048, 256>>>((uint4*)s, (uint4*)d, total / 16); });
    timeit("copy16 full", [&] { copy16<<<(unsigned)(total / 16 / 256), 256>>>((uint4*)s, (uint4*)d, total / 16); });
    timeit("copy12 full", [&] { copy12<<<(unsigned)(total / 12 / 256), 256>>>((U3*)s, (U3*)d, total / 12); });
    timeit("tile128x16", [&] { copy_tile<<<dim3(15, 68, B), 256>>>(s, d, W, H, fb); });
    timeit("tile512x4", [&] { copy_tile_wide<<<dim3(4, 270, B), 256>>>(s, d, W, H, fb); });
    return 0;
}
