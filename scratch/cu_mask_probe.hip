// cu_mask_probe: which compute units does bit k of a hipExtStreamCreateWithCUMask mask select on this device?
// For k = 0 .. 39 (and a few ranges): a stream whose mask has ONLY bit k set runs 512 single-wave workgroups that record
// (XCC_ID, SE_ID, SH_ID, CU_ID) from the hardware registers; the program prints the distinct places seen per mask.
//   hipcc --offload-arch=gfx950 -O2 scratch/cu_mask_probe.hip -o scratch/cu_mask_probe && scratch/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <set>
#include <vector>

__global__ void where(uint32_t* out) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // spin a little so that the workgroups of a launch spread over every CU the mask allows
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) {}
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc & 0xF) << 16 | (hw & 0xFFFF);
}

static void run(const char* name, const std::vector<uint32_t>& mask, uint32_t* d, std::vector<uint32_t>& h) {
    hipStream_t st;
    if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: create failed\n", name); return; }
    hipLaunchKernelGGL(where, dim3(512), dim3(64), 0, st, d);
    hipStreamSynchronize(st);
    hipMemcpy(h.data(), d, 512 * 4, hipMemcpyDeviceToHost);
    std::set<uint32_t> places;
    for (uint32_t v : h) places.insert((v >> 16) << 16 | ((v >> 13) & 7) << 12 | ((v >> 12) & 1) << 8 | ((v >> 8) & 0xF));   // xcc, se, sh, cu
    printf("%-14s %3zu places:", name, places.size());
    int n = 0;
    for (uint32_t p : places) { if (n++ < 12) printf(" x%u.se%u.sh%u.cu%u", p >> 16, (p >> 12) & 7, (p >> 8) & 1, p & 0xF); }
    printf("%s\n", places.size() > 12 ? " ..." : "");
    hipStreamDestroy(st);
}

int main() {
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    printf("%s: %d CUs\n", pr.name, pr.multiProcessorCount);
    uint32_t* d;
    hipMalloc(&d, 512 * 4);
    std::vector<uint32_t> h(512);
    const int words = (pr.multiProcessorCount + 31) / 32;
    for (int k = 0; k < 40; k++) {
        std::vector<uint32_t> m(words, 0u);
        m[k / 32] |= 1u << (k % 32);
        char name[32];
        snprintf(name, sizeof name, "bit %d", k);
        run(name, m, d, h);
    }
    { std::vector<uint32_t> m(words, 0u); m[0] = 0xFFu; run("bits 0..7", m, d, h); }
    { std::vector<uint32_t> m(words, 0xFFFFFFFFu); m[0] = 0xFFFFFF00u; run("all but 0..7", m, d, h); }
    { std::vector<uint32_t> m(words, 0u); m[0] = 0xFFFFFFFFu; run("bits 0..31", m, d, h); }
    { std::vector<uint32_t> m(words, 0xFFFFFFFFu); run("all", m, d, h); }
    hipFree(d);
    return 0;
}
