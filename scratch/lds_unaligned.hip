// probe: correctness and rate of LDS reads at byte addresses that are not multiples of the access size
// (ds_read_b32 / ds_read_b64 / ds_read_u16 at base + lane * stride + offset).  gfx950 reports unaligned DS access; this
// measures what it costs.  build: hipcc -O3 --offload-arch=gfx950 -w -o lds_unaligned lds_unaligned.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>

template <int BYTES>
__global__ __launch_bounds__(256) void k(const unsigned char* in, unsigned long long* out, unsigned long long* clk, int stride, int offset, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char s[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) s[i] = in[i];
    __syncthreads();
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned addr = (unsigned)(size_t)(s) + wave * 2048 + lane * stride + offset;   // LDS byte address (low 32 bits of the generic pointer = LDS offset)
    addr = (unsigned)__builtin_amdgcn_readfirstlane(0) + (unsigned)((unsigned long long)(s + wave * 2048 + lane * stride + offset) & 0xffffffffu);
    unsigned long long acc = 0;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        unsigned long long v0 = 0, v1 = 0, v2 = 0, v3 = 0;
        if (BYTES == 4) {
            unsigned a, b, c, d;
            asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:256\n ds_read_b32 %2, %4 offset:512\n ds_read_b32 %3, %4 offset:768\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(addr) : "memory");
            v0 = a; v1 = b; v2 = c; v3 = d;
        } else if (BYTES == 8) {
            asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:256\n ds_read_b64 %2, %4 offset:512\n ds_read_b64 %3, %4 offset:768\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(addr) : "memory");
        } else {
            unsigned a, b, c, d;
            asm volatile("ds_read_u16 %0, %4\n ds_read_u16 %1, %4 offset:256\n ds_read_u16 %2, %4 offset:512\n ds_read_u16 %3, %4 offset:768\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(addr) : "memory");
            v0 = a; v1 = b; v2 = c; v3 = d;
        }
        acc += v0 + (v1 << 1) + (v2 << 2) + (v3 << 3);
    }
    const unsigned long long t1 = clock64();
    // first-iteration values for the correctness check
    unsigned long long first = 0;
    {
        const unsigned char* p = s + wave * 2048 + lane * stride + offset;
        for (int b = 0; b < (BYTES < 8 ? BYTES : 8); b++) first |= (unsigned long long)p[b] << (8 * b);
    }
    unsigned long long got = 0;
    if (BYTES == 4) { unsigned a; asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=&v"(a) : "v"(addr) : "memory"); got = a; }
    else if (BYTES == 8) { asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=&v"(got) : "v"(addr) : "memory"); }
    else { unsigned a; asm volatile("ds_read_u16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=&v"(a) : "v"(addr) : "memory"); got = a; }
    out[(blockIdx.x * blockDim.x + threadIdx.x) * 2] = got ^ first;     // 0 when correct
    out[(blockIdx.x * blockDim.x + threadIdx.x) * 2 + 1] = acc;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int BYTES>
void run(const unsigned char* d_in, unsigned long long* d_out, unsigned long long* d_clk, int stride, int offset) {
    const int G = 256 * 8, T = 256, iters = 2000;
    hipLaunchKernelGGL(k<BYTES>, dim3(G), dim3(T), 0, 0, d_in, d_out, d_clk, stride, offset, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(G * T * 2), c(G);
    hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), d_clk, G * 8, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int i = 0; i < G * T; i++) bad += h[2 * i] != 0;
    double cyc = 0;
    for (int i = 0; i < G; i++) cyc += c[i];
    cyc /= G;
    // per CU: 8 workgroups x 4 waves x iters x 4 reads share one LDS; a workgroup's loop time covers the reads of all 32 waves
    printf("ds_read_%s stride %2d offset %d: %s, %.2f shader cycles per wave-instruction per CU\n", BYTES == 4 ? "b32" : BYTES == 8 ? "b64" : "u16",
           stride, offset, bad ? "WRONG DATA" : "data ok", cyc / (iters * 4.0 * 32.0));
}

int main() {
    std::vector<unsigned char> in(16384);
    for (size_t i = 0; i < in.size(); i++) in[i] = (unsigned char)(i * 7 + (i >> 8) * 13 + 1);
    unsigned char* d_in; unsigned long long *d_out, *d_clk;
    hipMalloc(&d_in, in.size()); hipMalloc(&d_out, 256 * 8 * 256 * 16); hipMalloc(&d_clk, 256 * 8 * 8);
    hipMemcpy(d_in, in.data(), in.size(), hipMemcpyHostToDevice);
    for (int off = 0; off < 4; off++) run<4>(d_in, d_out, d_clk, 4, off);
    for (int off = 0; off < 2; off++) run<4>(d_in, d_out, d_clk, 1, off);
    for (int off = 0; off < 2; off++) run<4>(d_in, d_out, d_clk, 3, off);
    run<4>(d_in, d_out, d_clk, 12, 0); run<4>(d_in, d_out, d_clk, 12, 3);
    for (int off = 0; off < 8; off += 1) run<8>(d_in, d_out, d_clk, 8, off);
    run<8>(d_in, d_out, d_clk, 3, 0); run<8>(d_in, d_out, d_clk, 3, 1); run<8>(d_in, d_out, d_clk, 4, 0); run<8>(d_in, d_out, d_clk, 1, 0);
    run<2>(d_in, d_out, d_clk, 2, 0); run<2>(d_in, d_out, d_clk, 2, 1); run<2>(d_in, d_out, d_clk, 1, 0);
    return 0;
}
