// probe: (1) rate of v_fma_f32 / v_mul_f32 when one operand is a denormal (an integer bit pattern used as a float),
// (2) exactness of the vertical lerp written in fp32 on such operands against the integer form
//     ((t*wy0 + b*wy1 + 512) >> 10 for t, b <= 8160, wy0 + wy1 = 32).
// build: hipcc -O3 --offload-arch=gfx950 -w -ffp-contract=off -o denorm_probe denorm_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP>
__global__ __launch_bounds__(256) void rate(float* out, unsigned long long* clk, int iters) {
    float a0 = 1.f, a1 = 2.f, a2 = 3.f, a3 = 4.f;
    const float den = __uint_as_float(threadIdx.x + 1u);          // denormal
    const float nrm = (float)(threadIdx.x + 1u);
    const float W = OP == 0 || OP == 2 ? 0x1p100f : 0x1p-20f;
    const float x = (OP == 0 || OP == 2) ? den : nrm;
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
        if (OP < 2)
            asm volatile(".rept 32\n v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3\n .endr"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(W));
        else
            asm volatile(".rept 32\n v_mul_f32 %0, %4, %5\n v_mul_f32 %1, %4, %5\n v_mul_f32 %2, %4, %5\n v_mul_f32 %3, %4, %5\n .endr"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(W));
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

// every (t, b, fy): t, b in 0..8160 step chosen to cover all low-bit patterns, fy in 0..31
__global__ void exact(unsigned* bad, unsigned* count) {
    const unsigned id = blockIdx.x * blockDim.x + threadIdx.x;   // t
    if (id > 8160) return;
    const unsigned t = id;
    unsigned nbad = 0, n = 0;
    for (unsigned b = 0; b <= 8160; b += 1 + (b % 7 == 0 ? 0 : 2)) {
        for (unsigned fy = 0; fy < 32; fy++) {
            const unsigned ref = (t * (32 - fy) + b * fy + 512) >> 10;
            // fp32: t, b as denormals (integer bit patterns); weights k * 2^126; bias half an output unit... see DESIGN
            const float W1 = __uint_as_float(fy << 5) * 0x1p127f * 0x1p17f;   // fy*32*2^-149 * 2^144 = fy * 2^0 ... scaled below
            (void)W1;
            const float wy1 = (float)fy * 0x1p121f, wy0 = (float)(32 - fy) * 0x1p121f;   // k * 2^121
            const float td = __uint_as_float(t), bd = __uint_as_float(b);                // t * 2^-149
            // u = t*wy0*2^-28 + 2^-29 (half of one unit of S at scale 2^-28), v = u + b*wy1*2^-28, z = v * 2^18 + ... magic
            float u = __builtin_fmaf(td, wy0, 0x1p-29f);
            float v = __builtin_fmaf(bd, wy1, u);
            // S*2^-28 + 2^-29 ; want RNE(S/1024 + 2^-11)?  (S + 0.5)/1024 -> ties impossible
            float z = v + 0x1p-5f;                 // ulp of 2^-5 is 2^-28: exact so far; use magic with ulp 2^-18 = 1024 * 2^-28: magic 2^5
            (void)z;
            float m = v + 0x1p5f;                  // RNE to multiples of 2^-18 = 1024 units
            const unsigned got = __float_as_uint(m) & 0xFFu;
            n++;
            if (got != ref) nbad++;
        }
    }
    atomicAdd(bad, nbad);
    atomicAdd(count, n);
}

template <int OP>
void run_rate(const char* name, float* o, unsigned long long* c) {
    const int G = 256 * 8, T = 256, iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate<OP>, dim3(G), dim3(T), 0, 0, o, c, 50);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate<OP>, dim3(G), dim3(T), 0, 0, o, c, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * G);
    hipMemcpy(h.data(), c, G * 16, hipMemcpyDeviceToHost);
    double s = 0, w = 0;
    for (int i = 0; i < G; i++) { s += h[2 * i]; w += h[2 * i + 1]; }
    const double mhz = s / w * 100.0;
    const double cyc = ms * 1e-3 * mhz * 1e6 / ((double)G * (T / 64) * iters * 128.0 / 1024.0);
    printf("%-34s %.3f ms, %.0f MHz, %.2f cycles per wave-instruction per SIMD\n", name, ms, mhz, cyc);
}

int main() {
    float* o; unsigned long long* c; unsigned* bad;
    hipMalloc(&o, 256 * 8 * 256 * 4); hipMalloc(&c, 256 * 8 * 16); hipMalloc(&bad, 8); hipMemset(bad, 0, 8);
    run_rate<0>("v_fma_f32, denormal operand", o, c);
    run_rate<1>("v_fma_f32, normal operands", o, c);
    run_rate<2>("v_mul_f32, denormal operand", o, c);
    run_rate<3>("v_mul_f32, normal operands", o, c);
    hipLaunchKernelGGL(exact, dim3((8161 + 255) / 256), dim3(256), 0, 0, bad, bad + 1);
    hipDeviceSynchronize();
    unsigned h[2]; hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost);
    printf("vertical lerp in fp32 on denormal operands: %u mismatches in %u cases\n", h[0], h[1]);
    return 0;
}
