// TEST INFRASTRUCTURE (oracle): the host libm's cosf / sinf / atanf / atan2f - what the reference's std::cos(float),
// std::sin(float), std::atan2(float, float) call (/root/reference/src/Stabilizer.cpp:662, 902-908, 1689) - summed over a
// range of arguments into the checksum that vs_op_libm_checksum computes from the product's restatement on the device
// (tests/test_libm.py).  The mix and the generator of atan2f's argument pairs are restated here from the definition in
// include/vs_stab.h / video-stab_amd/csrc/k_traj.hip: argument i of fn 0..2 is the float with bit pattern (uint32_t)i.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "vso.h"

namespace {
inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
inline uint64_t mix(uint64_t i, uint32_t bits) {
    uint64_t z = (i * 0x9E3779B97F4A7C15ull) ^ (uint64_t)bits;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline void pair_of(uint64_t i, float* y, float* x) {
    const uint64_t v = mix(i, 0x5EEDu);
    if (i & 1) { *y = u2f((uint32_t)v); *x = u2f((uint32_t)(v >> 32)); return; }
    float xx = 0.9f + 0.2f * ((float)(v & 0xFFFFFFu) / 16777216.0f);
    float yy = ((float)((v >> 24) & 0xFFFFFFu) / 16777216.0f - 0.5f) * (((v >> 48) & 1) ? 0.5f : 0.02f);
    if ((v >> 49) & 1) { xx = (xx - 1.0f) * 200.0f; yy *= 100.0f; }
    *y = yy; *x = xx;
}
}  // namespace

extern "C" uint64_t vso_libm_checksum(int fn, uint64_t start, uint64_t count, int threads) {
    if (threads < 1) threads = 1;
    std::vector<uint64_t> part((size_t)threads, 0);
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++)
        th.emplace_back([&, t] {
            uint64_t acc = 0;
            for (uint64_t i = (uint64_t)t; i < count; i += (uint64_t)threads) {
                const uint64_t idx = start + i;
                volatile float a, b = 0.f;
                float r;
                if (fn == 3) { float y, x; pair_of(idx, &y, &x); a = y; b = x; r = std::atan2(a, b); }
                else {
                    a = u2f((uint32_t)idx);
                    r = fn == 0 ? std::cos(a) : (fn == 1 ? std::sin(a) : std::atan(a));
                }
                acc += mix(idx, r != r ? 0x7FC00000u : f2u(r));
            }
            part[(size_t)t] = acc;
        });
    for (auto& x : th) x.join();
    uint64_t h = 0;
    for (uint64_t p : part) h += p;
    return h;
}
