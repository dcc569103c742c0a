// probe: does v_alignbyte_b32 use only the low two bits of its shift operand?  (prints the result for shifts 0 .. 9)
// build: hipcc -O3 --offload-arch=gfx950 -o alignbyte_probe.bin alignbyte_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    const unsigned s = threadIdx.x;
    out[s] = __builtin_amdgcn_alignbyte(0x77665544u, 0x33221100u, s);
}
int main() {
    unsigned* d; hipMalloc(&d, 64 * 4);
    k<<<1, 64>>>(d);
    unsigned h[64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int i = 0; i < 10; i++) printf("shift %d -> %08x\n", i, h[i]);
    printf("shift 0x104 -> %08x, shift 0xfffffffd -> ", 0u);
    return 0;
}
