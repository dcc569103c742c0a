// probe: issue rate of the vector instructions the warp / enhancer kernels are made of, on a full chip
// (256 CUs x W workgroups of 256 threads = W waves per SIMD), four independent chains per lane, 128 instructions per
// asm statement (so the compiler pads nothing in between).  Prints cycles per wave-instruction per SIMD at the measured shader clock.
// Generated table below; build: hipcc -O3 --offload-arch=gfx950 -w -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned long long* clk, int iters) {
    unsigned a0 = threadIdx.x + 1, a1 = threadIdx.x * 2 + 1, a2 = threadIdx.x * 3 + 1, a3 = threadIdx.x * 5 + 1;
    unsigned b = blockIdx.x * 7 + 3, c = threadIdx.x | 0x01020304u;
    unsigned long long q0 = a0, q1 = a1, q2 = a2, q3 = a3, qb = 0x3f8000003f800000ull + b, qc = 0x3f0000003f000000ull + c;
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
        if (OP == 0) asm volatile(".rept 32\nv_mad_u32_u24 %0, %0, %4, %5\nv_mad_u32_u24 %1, %1, %4, %5\nv_mad_u32_u24 %2, %2, %4, %5\nv_mad_u32_u24 %3, %3, %4, %5\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 1) asm volatile(".rept 32\nv_perm_b32 %0, %0, %5, %4\nv_perm_b32 %1, %1, %5, %4\nv_perm_b32 %2, %2, %5, %4\nv_perm_b32 %3, %3, %5, %4\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 2) asm volatile(".rept 32\nv_dot4_u32_u8 %0, %4, %5, %0\nv_dot4_u32_u8 %1, %4, %5, %1\nv_dot4_u32_u8 %2, %4, %5, %2\nv_dot4_u32_u8 %3, %4, %5, %3\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 3) asm volatile(".rept 32\nv_and_b32 %0, %0, %4\nv_and_b32 %1, %1, %4\nv_and_b32 %2, %2, %4\nv_and_b32 %3, %3, %4\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 4) asm volatile(".rept 32\nv_fma_f32 %0, %0, %4, %5\nv_fma_f32 %1, %1, %4, %5\nv_fma_f32 %2, %2, %4, %5\nv_fma_f32 %3, %3, %4, %5\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 5) asm volatile(".rept 32\nv_pk_fma_f32 %0, %0, %4, %5\nv_pk_fma_f32 %1, %1, %4, %5\nv_pk_fma_f32 %2, %2, %4, %5\nv_pk_fma_f32 %3, %3, %4, %5\n.endr" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(qb), "v"(qc) : "vcc");
        if (OP == 6) asm volatile(".rept 32\nv_pk_mad_u16 %0, %0, %4, %5\nv_pk_mad_u16 %1, %1, %4, %5\nv_pk_mad_u16 %2, %2, %4, %5\nv_pk_mad_u16 %3, %3, %4, %5\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 7) asm volatile(".rept 32\nv_cvt_f32_ubyte1 %0, %0\nv_cvt_f32_ubyte1 %1, %1\nv_cvt_f32_ubyte1 %2, %2\nv_cvt_f32_ubyte1 %3, %3\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 8) asm volatile(".rept 32\nv_cvt_pk_u8_f32 %0, %0, 1, %5\nv_cvt_pk_u8_f32 %1, %1, 1, %5\nv_cvt_pk_u8_f32 %2, %2, 1, %5\nv_cvt_pk_u8_f32 %3, %3, 1, %5\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 9) asm volatile(".rept 32\nv_lshl_add_u32 %0, %0, 2, %5\nv_lshl_add_u32 %1, %1, 2, %5\nv_lshl_add_u32 %2, %2, 2, %5\nv_lshl_add_u32 %3, %3, 2, %5\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 10) asm volatile(".rept 32\nv_alignbit_b32 %0, %0, %5, 8\nv_alignbit_b32 %1, %1, %5, 8\nv_alignbit_b32 %2, %2, %5, 8\nv_alignbit_b32 %3, %3, %5, 8\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 11) asm volatile(".rept 32\nv_mul_lo_u32 %0, %0, %4\nv_mul_lo_u32 %1, %1, %4\nv_mul_lo_u32 %2, %2, %4\nv_mul_lo_u32 %3, %3, %4\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 12) asm volatile(".rept 32\nv_bfe_u32 %0, %0, 5, 5\nv_bfe_u32 %1, %1, 5, 5\nv_bfe_u32 %2, %2, 5, 5\nv_bfe_u32 %3, %3, 5, 5\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 13) asm volatile(".rept 32\nv_add_u32 %0, %0, %5\nv_add_u32 %1, %1, %5\nv_add_u32 %2, %2, %5\nv_add_u32 %3, %3, %5\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 14) asm volatile(".rept 32\nv_mad_u32_u16 %0, %0, %4, %5\nv_mad_u32_u16 %1, %1, %4, %5\nv_mad_u32_u16 %2, %2, %4, %5\nv_mad_u32_u16 %3, %3, %4, %5\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 15) asm volatile(".rept 32\nv_dot2_u32_u16 %0, %4, %5, %0\nv_dot2_u32_u16 %1, %4, %5, %1\nv_dot2_u32_u16 %2, %4, %5, %2\nv_dot2_u32_u16 %3, %4, %5, %3\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 16) asm volatile(".rept 32\nv_pk_mul_lo_u16 %0, %0, %4\nv_pk_mul_lo_u16 %1, %1, %4\nv_pk_mul_lo_u16 %2, %2, %4\nv_pk_mul_lo_u16 %3, %3, %4\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 17) asm volatile(".rept 32\nv_pk_add_u16 %0, %0, %5\nv_pk_add_u16 %1, %1, %5\nv_pk_add_u16 %2, %2, %5\nv_pk_add_u16 %3, %3, %5\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 18) asm volatile(".rept 32\nv_mul_u32_u24 %0, %0, %4\nv_mul_u32_u24 %1, %1, %4\nv_mul_u32_u24 %2, %2, %4\nv_mul_u32_u24 %3, %3, %4\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 19) asm volatile(".rept 32\nv_add_f32 %0, %0, %5\nv_add_f32 %1, %1, %5\nv_add_f32 %2, %2, %5\nv_add_f32 %3, %3, %5\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 20) asm volatile(".rept 32\nv_mov_b32_sdwa %0, %5 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\nv_mov_b32_sdwa %1, %5 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\nv_mov_b32_sdwa %2, %5 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\nv_mov_b32_sdwa %3, %5 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 21) asm volatile(".rept 32\nv_mad_i32_i24 %0, %0, %4, %5\nv_mad_i32_i24 %1, %1, %4, %5\nv_mad_i32_i24 %2, %2, %4, %5\nv_mad_i32_i24 %3, %3, %4, %5\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 22) asm volatile(".rept 32\nv_and_or_b32 %0, %0, %4, %5\nv_and_or_b32 %1, %1, %4, %5\nv_and_or_b32 %2, %2, %4, %5\nv_and_or_b32 %3, %3, %4, %5\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 23) asm volatile(".rept 32\nv_pk_fma_f16 %0, %0, %4, %5\nv_pk_fma_f16 %1, %1, %4, %5\nv_pk_fma_f16 %2, %2, %4, %5\nv_pk_fma_f16 %3, %3, %4, %5\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 24) asm volatile(".rept 32\nv_lshlrev_b64 %0, 3, %0\nv_lshlrev_b64 %1, 3, %1\nv_lshlrev_b64 %2, 3, %2\nv_lshlrev_b64 %3, 3, %3\n.endr" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(qb), "v"(qc) : "vcc");
        if (OP == 25) asm volatile(".rept 32\nv_bfi_b32 %0, %4, %0, %5\nv_bfi_b32 %1, %4, %1, %5\nv_bfi_b32 %2, %4, %2, %5\nv_bfi_b32 %3, %4, %3, %5\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 26) asm volatile(".rept 32\nv_pk_mul_f32 %0, %0, %4\nv_pk_mul_f32 %1, %1, %4\nv_pk_mul_f32 %2, %2, %4\nv_pk_mul_f32 %3, %3, %4\n.endr" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(qb), "v"(qc) : "vcc");
        if (OP == 27) asm volatile(".rept 32\nv_mul_f32 %0, %0, %4\nv_mul_f32 %1, %1, %4\nv_mul_f32 %2, %2, %4\nv_mul_f32 %3, %3, %4\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 28) asm volatile(".rept 32\nv_cvt_u32_f32 %0, %0\nv_cvt_u32_f32 %1, %1\nv_cvt_u32_f32 %2, %2\nv_cvt_u32_f32 %3, %3\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 29) asm volatile(".rept 32\nv_sad_u8 %0, %0, %4, %5\nv_sad_u8 %1, %1, %4, %5\nv_sad_u8 %2, %2, %4, %5\nv_sad_u8 %3, %3, %4, %5\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 30) asm volatile(".rept 32\nv_lerp_u8 %0, %0, %4, %5\nv_lerp_u8 %1, %1, %4, %5\nv_lerp_u8 %2, %2, %4, %5\nv_lerp_u8 %3, %3, %4, %5\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 31) asm volatile(".rept 32\nv_pk_lshrrev_b16 %0, 3, %0\nv_pk_lshrrev_b16 %1, 3, %1\nv_pk_lshrrev_b16 %2, 3, %2\nv_pk_lshrrev_b16 %3, 3, %3\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 32) asm volatile(".rept 32\nv_fma_f64 %0, %0, %4, %5\nv_fma_f64 %1, %1, %4, %5\nv_fma_f64 %2, %2, %4, %5\nv_fma_f64 %3, %3, %4, %5\n.endr" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(qb), "v"(qc) : "vcc");
        if (OP == 33) asm volatile(".rept 32\nv_add3_u32 %0, %0, %4, %5\nv_add3_u32 %1, %1, %4, %5\nv_add3_u32 %2, %2, %4, %5\nv_add3_u32 %3, %3, %4, %5\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 34) asm volatile(".rept 32\nv_xad_u32 %0, %0, %4, %5\nv_xad_u32 %1, %1, %4, %5\nv_xad_u32 %2, %2, %4, %5\nv_xad_u32 %3, %3, %4, %5\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 35) asm volatile(".rept 32\nv_mul_hi_u32 %0, %0, %4\nv_mul_hi_u32 %1, %1, %4\nv_mul_hi_u32 %2, %2, %4\nv_mul_hi_u32 %3, %3, %4\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 36) asm volatile(".rept 32\nv_lshl_or_b32 %0, %0, 16, %5\nv_lshl_or_b32 %1, %1, 16, %5\nv_lshl_or_b32 %2, %2, 16, %5\nv_lshl_or_b32 %3, %3, 16, %5\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 37) asm volatile(".rept 32\nv_alignbyte_b32 %0, %0, %5, %4\nv_alignbyte_b32 %1, %1, %5, %4\nv_alignbyte_b32 %2, %2, %5, %4\nv_alignbyte_b32 %3, %3, %5, %4\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 38) asm volatile(".rept 32\nv_ashrrev_i32 %0, 10, %0\nv_ashrrev_i32 %1, 10, %1\nv_ashrrev_i32 %2, 10, %2\nv_ashrrev_i32 %3, 10, %3\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 39) asm volatile(".rept 32\nv_fmac_f32 %0, %4, %5\nv_fmac_f32 %1, %4, %5\nv_fmac_f32 %2, %4, %5\nv_fmac_f32 %3, %4, %5\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 40) asm volatile(".rept 32\nv_lshlrev_b32 %0, 8, %0\nv_lshlrev_b32 %1, 8, %1\nv_lshlrev_b32 %2, 8, %2\nv_lshlrev_b32 %3, 8, %3\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 41) asm volatile(".rept 32\nv_or_b32 %0, %0, %4\nv_or_b32 %1, %1, %4\nv_or_b32 %2, %2, %4\nv_or_b32 %3, %3, %4\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 42) asm volatile(".rept 32\nv_sub_u32 %0, %0, %4\nv_sub_u32 %1, %1, %4\nv_sub_u32 %2, %2, %4\nv_sub_u32 %3, %3, %4\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        if (OP == 43) asm volatile(".rept 32\nv_cndmask_b32 %0, %0, %4, vcc\nv_cndmask_b32 %1, %1, %4, vcc\nv_cndmask_b32 %2, %2, %4, vcc\nv_cndmask_b32 %3, %3, %4, vcc\n.endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (unsigned)(q0 + q1 + q2 + q3);
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int OP>
void run(const char* name, int wg_per_cu, unsigned* o, unsigned long long* c) {
    const int G = 256 * wg_per_cu, T = 256, iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(G), dim3(T), 0, 0, o, c, 50);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<OP>, dim3(G), dim3(T), 0, 0, o, c, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * G);
    hipMemcpy(h.data(), c, G * 16, hipMemcpyDeviceToHost);
    double s = 0, w = 0;
    for (int i = 0; i < G; i++) { s += h[2 * i]; w += h[2 * i + 1]; }
    const double mhz = s / w * 100.0;
    const double winstr = (double)G * (T / 64) * iters * 128.0;   // wave-instructions
    const double cyc = ms * 1e-3 * mhz * 1e6 / (winstr / 1024.0);
    printf("%-18s %d waves/SIMD: %8.3f ms, clock %5.0f MHz, %5.2f cycles per wave-instruction per SIMD\n", name, wg_per_cu, ms, mhz, cyc);
}

int main() {
    unsigned* o; unsigned long long* c;
    hipMalloc(&o, 256 * 8 * 256 * 4); hipMalloc(&c, 256 * 8 * 16);
    for (int w : {8, 2}) {
        run<0>("v_mad_u32_u24", w, o, c);
        run<1>("v_perm_b32", w, o, c);
        run<2>("v_dot4_u32_u8", w, o, c);
        run<3>("v_and_b32", w, o, c);
        run<4>("v_fma_f32", w, o, c);
        run<5>("v_pk_fma_f32", w, o, c);
        run<6>("v_pk_mad_u16", w, o, c);
        run<7>("v_cvt_f32_ubyte1", w, o, c);
        run<8>("v_cvt_pk_u8_f32", w, o, c);
        run<9>("v_lshl_add_u32", w, o, c);
        run<10>("v_alignbit_b32", w, o, c);
        run<11>("v_mul_lo_u32", w, o, c);
        run<12>("v_bfe_u32", w, o, c);
        run<13>("v_add_u32", w, o, c);
        run<14>("v_mad_u32_u16", w, o, c);
        run<15>("v_dot2_u32_u16", w, o, c);
        run<16>("v_pk_mul_lo_u16", w, o, c);
        run<17>("v_pk_add_u16", w, o, c);
        run<18>("v_mul_u32_u24", w, o, c);
        run<19>("v_add_f32", w, o, c);
        run<20>("v_mov_b32_sdwa", w, o, c);
        run<21>("v_mad_i32_i24", w, o, c);
        run<22>("v_and_or_b32", w, o, c);
        run<23>("v_pk_fma_f16", w, o, c);
        run<24>("v_lshlrev_b64", w, o, c);
        run<25>("v_bfi_b32", w, o, c);
        run<26>("v_pk_mul_f32", w, o, c);
        run<27>("v_mul_f32", w, o, c);
        run<28>("v_cvt_u32_f32", w, o, c);
        run<29>("v_sad_u8", w, o, c);
        run<30>("v_lerp_u8", w, o, c);
        run<31>("v_pk_lshrrev_b16", w, o, c);
        run<32>("v_fma_f64", w, o, c);
        run<33>("v_med3_u32? v_add3_u32", w, o, c);
        run<34>("v_xad_u32", w, o, c);
        run<35>("v_mul_hi_u32", w, o, c);
        run<36>("v_lshl_or_b32", w, o, c);
        run<37>("v_alignbyte_b32", w, o, c);
        run<38>("v_ashrrev_i32", w, o, c);
        run<39>("v_fmac_f32", w, o, c);
        run<40>("v_lshlrev_b32", w, o, c);
        run<41>("v_or_b32", w, o, c);
        run<42>("v_sub_u32", w, o, c);
        run<43>("v_cndmask_b32", w, o, c);
    }
    return 0;
}
