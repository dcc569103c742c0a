// Host-side robustness check of the AutoZoomCrop contour logic (k_azc.hip: BitFrame scan, border following, spans,
// shrink loop): the file is compiled host-only with ASan + UBSan and vs_azc_crop_from_mask is fed random masks.
// Build line in scratch/README.md.  Not part of the product.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "vs_stab.h"

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 20000;
    std::mt19937 rng(777);
    long valid = 0, contours = 0;
    std::vector<uint8_t> m, filled;
    for (int it = 0; it < rounds; it++) {
        const int w = 1 + rng() % (it % 7 == 0 ? 300 : 90), h = 1 + rng() % (it % 11 == 0 ? 150 : 60);
        const size_t stride = w + rng() % 5;
        m.assign(stride * h, 0);
        const int mode = rng() % 5;
        const unsigned dens = rng() % 101;
        if (mode == 0) {
            for (auto& v : m) v = (rng() % 100 < dens) ? 255 : 0;
        } else if (mode == 1) {                       // blocks
            const int bw = 1 + rng() % 40, bh = 1 + rng() % 20;
            std::vector<uint8_t> cell(((w + bw - 1) / bw) * ((h + bh - 1) / bh));
            for (auto& c : cell) c = rng() % 100 < dens;
            for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) m[y * stride + x] = cell[(y / bh) * ((w + bw - 1) / bw) + x / bw] ? 200 : 0;
        } else if (mode == 2) {                       // all content with a few holes
            for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) m[y * stride + x] = 1;
            for (int k = rng() % 6; k > 0; k--) m[(rng() % h) * stride + rng() % w] = 0;
        } else if (mode == 3) {                       // rings
            for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
                const int d = std::min(std::min(x, w - 1 - x), std::min(y, h - 1 - y));
                m[y * stride + x] = (d / (1 + (int)(dens % 4))) % 2 == 0 ? 255 : 0;
            }
        } else {                                      // one-pixel lines and diagonals
            for (int k = 0; k < 6; k++) {
                int x = rng() % w, y = rng() % h;
                const int dx = (int)(rng() % 3) - 1, dy = (int)(rng() % 3) - 1;
                for (int s = 0; s < 80 && x >= 0 && x < w && y >= 0 && y < h; s++, x += dx, y += dy) m[y * stride + x] = 9;
            }
        }
        int32_t info[8];
        filled.assign((size_t)w * h, 7);
        const int st = vs_azc_crop_from_mask(m.data(), w, h, stride, info, (it & 1) ? filled.data() : nullptr);
        if (st != VS_OK) { fprintf(stderr, "status %d at round %d\n", st, it); return 1; }
        if (info[7]) {
            valid++;
            if (info[2] < 0 || info[3] < 0 || info[2] + info[4] > w || info[3] + info[5] > h || info[4] <= 0 || info[5] <= 0) {
                fprintf(stderr, "rectangle outside the image at round %d\n", it);
                return 1;
            }
        }
        contours += info[0];
    }
    printf("%d masks: %ld with a crop rectangle, %ld contours followed\n", rounds, valid, contours);
    return 0;
}
