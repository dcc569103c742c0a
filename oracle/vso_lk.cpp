// CPU ORACLE (test infrastructure) - cv::calcOpticalFlowPyrLK restated
// (reference call site src/Stabilizer.cpp:611-619: win 15x15, maxLevel 2,
// TermCriteria(COUNT+EPS, 20, 0.03); Stabilizer_legacy.cpp:218-224: 21x21/3).
//
// Structure follows OpenCV 4.11 video/src/lkpyramid.cpp: both pyramids are
// rebuilt per call (pyrDown, REFLECT_101 padding of winSize), Scharr
// derivatives per level (zero padding), per point coarse-to-fine Newton
// iterations with 14-bit bilinear window weights.
//
// Definition choice: OpenCV accumulates the integer products
// (ix*ix, ix*iy, iy*iy, diff*ix, diff*iy) in float, in an order that depends
// on the SIMD width of the build (scalar / SSE / AVX / NEON differ in the last
// bits).  The oracle takes the EXACT integer sum (int64) and converts to float
// once, which every OpenCV build approximates to within float rounding.
#include "vso_internal.h"

#include <algorithm>
#include <thread>

namespace vso {

struct Level {
    int w, h;
    std::vector<uint8_t> img;     // w*h
    std::vector<int16_t> deriv;   // w*h*2 (prev only)
};

static inline int refl(int p, int len) { return border_interpolate(p, len, VS_BORDER_REFLECT_101); }

// image sample with REFLECT_101 padding (buildOpticalFlowPyramid pyrBorder)
static inline int IMG(const Level& L, int x, int y) {
    return L.img[(size_t)refl(y, L.h) * L.w + refl(x, L.w)];
}
// derivative sample with constant-0 padding (derivBorder = BORDER_CONSTANT)
static inline int DER(const Level& L, int x, int y, int c) {
    if ((unsigned)x >= (unsigned)L.w || (unsigned)y >= (unsigned)L.h) return 0;
    return L.deriv[((size_t)y * L.w + x) * 2 + c];
}

static inline int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

static int build_pyramid(const uint8_t* img, int w, int h, size_t stride, int win, int max_level,
                         std::vector<Level>& pyr) {
    pyr.clear();
    pyr.resize(max_level + 1);
    pyr[0].w = w; pyr[0].h = h;
    pyr[0].img.resize((size_t)w * h);
    for (int y = 0; y < h; y++)
        std::copy(img + (size_t)y * stride, img + (size_t)y * stride + w, pyr[0].img.begin() + (size_t)y * w);
    int sw = w, sh = h;
    for (int level = 0; level <= max_level; level++) {
        if (level != 0) {
            Level& L = pyr[level];
            const Level& P = pyr[level - 1];
            L.w = sw; L.h = sh;
            L.img.resize((size_t)sw * sh);
            pyr_down(P.img.data(), P.w, P.h, P.w, L.img.data(), sw);
        }
        sw = (sw + 1) / 2;
        sh = (sh + 1) / 2;
        if (sw <= win || sh <= win) {
            pyr.resize(level + 1);
            return level;
        }
    }
    return max_level;
}

struct LKArgs {
    const Level* I;
    const Level* J;
    const float* prev_pts;
    float* next_pts;
    uint8_t* status;
    float* err;
    int win, level, max_level, max_count;
    float eps2;  // criteria.epsilon^2 as double->compared via ddot
    double eps2d;
};

static void track_range(const LKArgs& a, int p0, int p1) {
    const int win = a.win;
    const float halfWin = (win - 1) * 0.5f;
    const Level& I = *a.I;
    const Level& J = *a.J;
    const int W_BITS = 14;
    const float FLT_SCALE = 1.f / (1 << 20);
    const float minEigThreshold = 1e-4f;
    std::vector<int16_t> Iwin((size_t)win * win), dIwin((size_t)win * win * 2);
    for (int pt = p0; pt < p1; pt++) {
        float scale = (float)(1. / (1 << a.level));
        float prevx = a.prev_pts[2 * pt] * scale, prevy = a.prev_pts[2 * pt + 1] * scale;
        float nextx, nexty;
        if (a.level == a.max_level) { nextx = prevx; nexty = prevy; }
        else { nextx = a.next_pts[2 * pt] * 2.f; nexty = a.next_pts[2 * pt + 1] * 2.f; }
        a.next_pts[2 * pt] = nextx;
        a.next_pts[2 * pt + 1] = nexty;

        prevx -= halfWin; prevy -= halfWin;
        int ipx = cv_floor(prevx), ipy = cv_floor(prevy);
        if (ipx < -win || ipx >= I.w || ipy < -win || ipy >= I.h) {
            if (a.level == 0) { a.status[pt] = 0; a.err[pt] = 0; }
            continue;
        }
        float fa = prevx - ipx, fb = prevy - ipy;
        int iw00 = cv_round((1.f - fa) * (1.f - fb) * (1 << W_BITS));
        int iw01 = cv_round(fa * (1.f - fb) * (1 << W_BITS));
        int iw10 = cv_round((1.f - fa) * fb * (1 << W_BITS));
        int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        int64_t sA11 = 0, sA12 = 0, sA22 = 0;
        for (int y = 0; y < win; y++)
            for (int x = 0; x < win; x++) {
                int X = ipx + x, Y = ipy + y;
                int ival = descale(IMG(I, X, Y) * iw00 + IMG(I, X + 1, Y) * iw01 +
                                   IMG(I, X, Y + 1) * iw10 + IMG(I, X + 1, Y + 1) * iw11, W_BITS - 5);
                int ixval = descale(DER(I, X, Y, 0) * iw00 + DER(I, X + 1, Y, 0) * iw01 +
                                    DER(I, X, Y + 1, 0) * iw10 + DER(I, X + 1, Y + 1, 0) * iw11, W_BITS);
                int iyval = descale(DER(I, X, Y, 1) * iw00 + DER(I, X + 1, Y, 1) * iw01 +
                                    DER(I, X, Y + 1, 1) * iw10 + DER(I, X + 1, Y + 1, 1) * iw11, W_BITS);
                Iwin[(size_t)y * win + x] = (int16_t)ival;
                dIwin[((size_t)y * win + x) * 2] = (int16_t)ixval;
                dIwin[((size_t)y * win + x) * 2 + 1] = (int16_t)iyval;
                sA11 += (int64_t)ixval * ixval;
                sA12 += (int64_t)ixval * iyval;
                sA22 += (int64_t)iyval * iyval;
            }
        float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE, A22 = (float)sA22 * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        float minEig = (A22 + A11 - std::sqrt((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) /
                       (float)(2 * win * win);
        if (minEig < minEigThreshold || D < FLT_EPSILON) {
            if (a.level == 0) a.status[pt] = 0;
            continue;
        }
        D = 1.f / D;
        nextx -= halfWin; nexty -= halfWin;
        float pdx = 0, pdy = 0;
        for (int j = 0; j < a.max_count; j++) {
            int inx = cv_floor(nextx), iny = cv_floor(nexty);
            if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) {
                if (a.level == 0) a.status[pt] = 0;
                break;
            }
            fa = nextx - inx; fb = nexty - iny;
            iw00 = cv_round((1.f - fa) * (1.f - fb) * (1 << W_BITS));
            iw01 = cv_round(fa * (1.f - fb) * (1 << W_BITS));
            iw10 = cv_round((1.f - fa) * fb * (1 << W_BITS));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            int64_t sb1 = 0, sb2 = 0;
            for (int y = 0; y < win; y++)
                for (int x = 0; x < win; x++) {
                    int X = inx + x, Y = iny + y;
                    int diff = descale(IMG(J, X, Y) * iw00 + IMG(J, X + 1, Y) * iw01 +
                                       IMG(J, X, Y + 1) * iw10 + IMG(J, X + 1, Y + 1) * iw11, W_BITS - 5) -
                               Iwin[(size_t)y * win + x];
                    sb1 += (int64_t)diff * dIwin[((size_t)y * win + x) * 2];
                    sb2 += (int64_t)diff * dIwin[((size_t)y * win + x) * 2 + 1];
                }
            float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
            float dx = (A12 * b2 - A22 * b1) * D;
            float dy = (A12 * b1 - A11 * b2) * D;
            nextx += dx; nexty += dy;
            a.next_pts[2 * pt] = nextx + halfWin;
            a.next_pts[2 * pt + 1] = nexty + halfWin;
            // Point2f::ddot: double accumulation of the float products
            if ((double)dx * dx + (double)dy * dy <= a.eps2d) break;
            if (j > 0 && std::abs(dx + pdx) < 0.01 && std::abs(dy + pdy) < 0.01) {
                a.next_pts[2 * pt] -= dx * 0.5f;
                a.next_pts[2 * pt + 1] -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        if (a.status[pt] && a.level == 0) {
            float npx = a.next_pts[2 * pt] - halfWin, npy = a.next_pts[2 * pt + 1] - halfWin;
            int inx = cv_floor(npx), iny = cv_floor(npy);
            if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) {
                a.status[pt] = 0;
                continue;
            }
            float aa = npx - inx, bb = npy - iny;
            iw00 = cv_round((1.f - aa) * (1.f - bb) * (1 << W_BITS));
            iw01 = cv_round(aa * (1.f - bb) * (1 << W_BITS));
            iw10 = cv_round((1.f - aa) * bb * (1 << W_BITS));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            int64_t errsum = 0;  // exact (OpenCV: float accumulation of |diff|)
            for (int y = 0; y < win; y++)
                for (int x = 0; x < win; x++) {
                    int X = inx + x, Y = iny + y;
                    int diff = descale(IMG(J, X, Y) * iw00 + IMG(J, X + 1, Y) * iw01 +
                                       IMG(J, X, Y + 1) * iw10 + IMG(J, X + 1, Y + 1) * iw11, W_BITS - 5) -
                               Iwin[(size_t)y * win + x];
                    errsum += std::abs(diff);
                }
            a.err[pt] = (float)errsum * 1.f / (32 * win * win);
        }
    }
}

int pyr_lk(const uint8_t* prev, const uint8_t* next, int w, int h, size_t stride,
           const float* prev_pts, int n, float* next_pts, uint8_t* status, float* err, int win,
           int max_level, int max_iters, double eps, int nthreads) {
    std::vector<Level> P, N;
    max_level = build_pyramid(prev, w, h, stride, win, max_level, P);
    max_level = build_pyramid(next, w, h, stride, win, max_level, N);
    int max_count = std::min(std::max(max_iters, 0), 100);
    double e = std::min(std::max(eps, 0.), 10.);
    e *= e;
    for (int i = 0; i < n; i++) { status[i] = 1; err[i] = 0; }
    for (int level = max_level; level >= 0; level--) {
        Level& L = P[level];
        L.deriv.resize((size_t)L.w * L.h * 2);
        scharr(L.img.data(), L.w, L.h, L.w, L.deriv.data());
        LKArgs a{&P[level], &N[level], prev_pts, next_pts, status, err, win, level, max_level, max_count, (float)e, e};
        if (nthreads <= 1 || n < 2 * nthreads) {
            track_range(a, 0, n);
        } else {
            std::vector<std::thread> th;
            int per = (n + nthreads - 1) / nthreads;
            for (int t = 0; t < nthreads; t++) {
                int p0 = t * per, p1 = std::min(n, p0 + per);
                if (p0 >= p1) break;
                th.emplace_back([=, &a] { track_range(a, p0, p1); });
            }
            for (auto& t : th) t.join();
        }
    }
    return max_level;
}

}  // namespace vso

extern "C" int vso_pyr_lk(const uint8_t* prev, const uint8_t* next, int w, int h, size_t stride,
                          const float* prev_pts, int n, float* next_pts, uint8_t* status,
                          float* err, int win, int max_level, int max_iters, double eps) {
    return vso::pyr_lk(prev, next, w, h, stride, prev_pts, n, next_pts, status, err, win,
                       max_level, max_iters, eps, vso::g_threads);
}
