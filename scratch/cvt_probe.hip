// probe: does v_cvt_pk_u8_f32 round to nearest even and saturate like cvRound + saturate_cast<uchar>?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const float* in, unsigned* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0, 0);
}
int main() {
    const int n = 4096 * 8 + 16;
    float* h = new float[n]; unsigned* o = new unsigned[n];
    int m = 0;
    for (int i = -1024 * 8; i < 3072 * 8; i++) h[m++] = i / 8.0f;      // -1024 .. 3072 in steps of 1/8
    h[m++] = 1e9f; h[m++] = -1e9f; h[m++] = 254.5f; h[m++] = 255.5f; h[m++] = 0.49999997f; h[m++] = 0.50000006f;
    h[m++] = 2.5f; h[m++] = 3.5f; h[m++] = -0.5f; h[m++] = -0.4999f;
    float* d; unsigned* d2;
    hipMalloc(&d, m * 4); hipMalloc(&d2, m * 4);
    hipMemcpy(d, h, m * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((m + 255) / 256), dim3(256), 0, 0, d, d2, m);
    hipMemcpy(o, d2, m * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < m; i++) {
        long r = lrintf(h[i]); r = r < 0 ? 0 : r > 255 ? 255 : r;
        if ((long)o[i] != r) { if (bad < 10) printf("x=%g got %u want %ld\n", h[i], o[i], r); bad++; }
    }
    printf("cvt_pk_u8_f32 mismatches: %d of %d\n", bad, m);
    return 0;
}
