// libm_check [cos|sin|atan|atan2] : the product's restatements of glibc's cosf / sinf / atan2f (video-stab_amd/csrc/vs_libm.h,
// host build) against the host's own libm.  cos / sin / atan: every float; atan2: 2^30 random bit patterns + 2^30 pairs shaped like
// the stabilizer's (x near 1, y small; pixel translations).  Prints "<name> <values> <mismatches>" per function; exit code 1 on
// any mismatch.  Built and run by tests/test_libm.py.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../../video-stab_amd/csrc/vs_libm.h"

using vslibm::f2u;
using vslibm::u2f;

static inline uint64_t rng(uint64_t& s) { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return s * 0x2545F4914F6CDD1DULL; }
static bool same(float a, float b) { return f2u(a) == f2u(b) || (a != a && b != b); }

int main(int argc, char** argv) {
    const char* what = argc > 1 ? argv[1] : "all";
    // "quick": every 256th float and 2^22 pairs - the sanitizer run of `make asan`
    const bool quick = !strcmp(what, "quick");
    if (quick) what = "all";
    const uint64_t fstep = quick ? 256 : 1;
    int NT = (int)std::thread::hardware_concurrency();
    if (NT < 1) NT = 1;
    if (NT > 16) NT = 16;
    std::atomic<long> bad[4] = {{0}, {0}, {0}, {0}};
    const bool all = !strcmp(what, "all");
    std::vector<std::thread> th;
    for (int t = 0; t < NT; t++) th.emplace_back([&, t] {
        long b[4] = {0, 0, 0, 0};
        const bool dc = all || !strcmp(what, "cos"), ds = all || !strcmp(what, "sin"), da = all || !strcmp(what, "atan");
        if (dc || ds || da)
            for (uint64_t u = (uint64_t)t * fstep; u < (1ull << 32); u += (uint64_t)NT * fstep) {
                volatile float x = u2f((uint32_t)u);
                if (dc && !same(cosf(x), vslibm::cosf_ref(x))) b[0]++;
                if (ds && !same(sinf(x), vslibm::sinf_ref(x))) b[1]++;
                if (da && !same(atanf(x), vslibm::atanf_ref(x))) b[2]++;
            }
        if (all || !strcmp(what, "atan2")) {
            uint64_t s = 0x9E3779B97F4A7C15ull * (uint64_t)(t + 1);
            const long n = (quick ? (1l << 21) : (1l << 30)) / NT;
            for (long i = 0; i < n; i++) {
                const uint64_t v = rng(s);
                volatile float y = u2f((uint32_t)v), x = u2f((uint32_t)(v >> 32));
                if (!same(atan2f(y, x), vslibm::atan2f_ref(y, x))) b[3]++;
            }
            for (long i = 0; i < n; i++) {
                const uint64_t v = rng(s);
                float xx = 0.9f + 0.2f * ((v & 0xFFFFFF) / 16777216.0f), yy = (((v >> 24) & 0xFFFFFF) / 16777216.0f - 0.5f) * (((v >> 48) & 1) ? 0.5f : 0.02f);
                if ((v >> 49) & 1) { xx = (xx - 1.0f) * 200.0f; yy *= 100.0f; }
                volatile float y = yy, x = xx;
                if (!same(atan2f(y, x), vslibm::atan2f_ref(y, x))) b[3]++;
            }
        }
        for (int i = 0; i < 4; i++) bad[i] += b[i];
    });
    for (auto& x : th) x.join();
    const char* names[4] = {"cosf", "sinf", "atanf", "atan2f"};
    long total = 0;
    for (int i = 0; i < 4; i++) {
        const bool ran = all || !strcmp(what, i == 0 ? "cos" : i == 1 ? "sin" : i == 2 ? "atan" : "atan2");
        if (ran) printf("%s %s %ld\n", names[i], quick ? "sampled" : (i < 3 ? "4294967296" : "2147483648"), bad[i].load());
        total += bad[i].load();
    }
    return total ? 1 : 0;
}
