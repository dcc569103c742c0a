// probe: shader clock under a VALU-heavy load on all CUs (s_memtime cycles per 100 MHz wall-clock tick)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out, unsigned long long* clk, int iters) {
    unsigned a = threadIdx.x, b = blockIdx.x * 7 + 1, c = 3;
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 64; j++) { a = __umul24(a, b) + c; c = __builtin_amdgcn_perm(a, c, 0x05010400u); }
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + c;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}
int main() {
    const int G = 256 * 8, T = 256;
    unsigned* o; unsigned long long* c;
    hipMalloc(&o, G * T * 4); hipMalloc(&c, G * 16);
    unsigned long long* h = new unsigned long long[2 * G];
    for (int iters : {200, 2000, 20000}) {
        hipLaunchKernelGGL(k, dim3(G), dim3(T), 0, 0, o, c, iters);
        hipDeviceSynchronize();
        hipMemcpy(h, c, G * 16, hipMemcpyDeviceToHost);
        double s = 0, w = 0;
        for (int i = 0; i < G; i++) { s += h[2 * i]; w += h[2 * i + 1]; }
        // 128 VALU instructions per inner iteration
        printf("iters %d: clock64/wall = %.3f -> %.0f MHz; cycles per VALU instr per wave = %.2f\n", iters, s / w, s / w * 100.0,
               (s / G) / (iters * 128.0));
    }
    return 0;
}
