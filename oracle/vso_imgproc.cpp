// CPU ORACLE (test infrastructure) - image primitives.  See vso.h for the
// "parity unpinned" statement.  Each function restates the OpenCV 4.11 8-bit
// code path that the reference reaches from src/Stabilizer.cpp (cited).
#include "vso_internal.h"

#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

namespace vso {

// cv::borderInterpolate
int border_interpolate(int p, int len, int border) {
    if ((unsigned)p < (unsigned)len) return p;
    switch (border) {
        case VS_BORDER_REPLICATE:
            return p < 0 ? 0 : len - 1;
        case VS_BORDER_REFLECT:
        case VS_BORDER_REFLECT_101: {
            int delta = border == VS_BORDER_REFLECT_101;
            if (len == 1) return 0;
            do {
                if (p < 0) p = -p - 1 + delta;
                else p = len - 1 - (p - len) - delta;
            } while ((unsigned)p >= (unsigned)len);
            return p;
        }
        case VS_BORDER_WRAP:
            if (p < 0) p -= ((p - len + 1) / len) * len;
            if (p >= len) p %= len;
            return p;
        default:
            return -1;  // BORDER_CONSTANT
    }
}

// ---------------------------------------------------------------------------
// cv::resize INTER_LINEAR, CV_8U (Stabilizer.cpp:304,449,602,1121).
//  * exact 2x2 decimation is re-routed to INTER_AREA's fast path
//    ((s00+s01+s10+s11+2)>>2), as cv::resize does for INTER_LINEAR;
//  * otherwise HResizeLinear (11-bit coeffs) + VResizeLinear 8u
//    ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2)>>2.
void resize_linear_u8(const uint8_t* src, int sw, int sh, size_t sstride, int cn,
                      uint8_t* dst, int dw, int dh, size_t dstride) {
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int iscale_x = (int)lrint(scale_x), iscale_y = (int)lrint(scale_y);
    bool is_area_fast = std::abs(scale_x - iscale_x) < DBL_EPSILON &&
                        std::abs(scale_y - iscale_y) < DBL_EPSILON;
    if (is_area_fast && iscale_x == 2 && iscale_y == 2) {
        for (int y = 0; y < dh; y++) {
            const uint8_t* s0 = src + (size_t)(2 * y) * sstride;
            const uint8_t* s1 = s0 + sstride;
            uint8_t* d = dst + (size_t)y * dstride;
            for (int x = 0; x < dw; x++)
                for (int k = 0; k < cn; k++) {
                    int i = 2 * x * cn + k;
                    d[x * cn + k] = (uint8_t)((s0[i] + s0[i + cn] + s1[i] + s1[i + cn] + 2) >> 2);
                }
        }
        return;
    }
    const int SCALE = 2048;  // INTER_RESIZE_COEF_SCALE
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> ialpha(dw * 2), ibeta(dh * 2);
    int xmax = dw;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) {
            xmax = std::min(xmax, dx);
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        }
        xofs[dx] = sx;
        ialpha[dx * 2] = sat_short(cv_round((1.f - fx) * SCALE));
        ialpha[dx * 2 + 1] = sat_short(cv_round(fx * SCALE));
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        yofs[dy] = sy;
        ibeta[dy * 2] = sat_short(cv_round((1.f - fy) * SCALE));
        ibeta[dy * 2 + 1] = sat_short(cv_round(fy * SCALE));
    }
    parallel_rows(dh, [&](int ya, int yb) {
    std::vector<int> rows[2];
    rows[0].resize((size_t)dw * cn);
    rows[1].resize((size_t)dw * cn);
    for (int dy = ya; dy < yb; dy++) {
        for (int k = 0; k < 2; k++) {
            int sy = yofs[dy] + k;
            sy = sy >= 0 ? (sy < sh ? sy : sh - 1) : 0;  // clip()
            const uint8_t* S = src + (size_t)sy * sstride;
            int* D = rows[k].data();
            for (int dx = 0; dx < dw; dx++) {
                int sx = xofs[dx] * cn;
                if (dx < xmax) {
                    for (int c = 0; c < cn; c++)
                        D[dx * cn + c] = S[sx + c] * ialpha[dx * 2] + S[sx + cn + c] * ialpha[dx * 2 + 1];
                } else {
                    for (int c = 0; c < cn; c++) D[dx * cn + c] = S[sx + c] * SCALE;
                }
            }
        }
        int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
        uint8_t* d = dst + (size_t)dy * dstride;
        const int* S0 = rows[0].data();
        const int* S1 = rows[1].data();
        for (int x = 0; x < dw * cn; x++)
            d[x] = (uint8_t)((((b0 * (S0[x] >> 4)) >> 16) + ((b1 * (S1[x] >> 4)) >> 16) + 2) >> 2);
    }
    });
}

// cv::cvtColor BGR2GRAY 8U: (B*3735 + G*19235 + R*9798 + 2^14) >> 15
void bgr2gray(const uint8_t* src, int w, int h, size_t sstride, uint8_t* dst, size_t dstride) {
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src + (size_t)y * sstride;
        uint8_t* d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++)
            d[x] = (uint8_t)((s[3 * x] * 3735 + s[3 * x + 1] * 19235 + s[3 * x + 2] * 9798 + (1 << 14)) >> 15);
    }
}

// cv::pyrDown 8UC1: [1 4 6 4 1] x [1 4 6 4 1], (v + 128) >> 8, REFLECT_101,
// dst size ((sw+1)/2, (sh+1)/2).
void pyr_down(const uint8_t* src, int sw, int sh, size_t sstride, uint8_t* dst, size_t dstride) {
    int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
    std::vector<int> hrow[5];
    for (auto& r : hrow) r.resize(dw);
    for (int y = 0; y < dh; y++) {
        for (int k = 0; k < 5; k++) {
            int sy = border_interpolate(2 * y + k - 2, sh, VS_BORDER_REFLECT_101);
            const uint8_t* s = src + (size_t)sy * sstride;
            for (int x = 0; x < dw; x++) {
                int i0 = border_interpolate(2 * x - 2, sw, VS_BORDER_REFLECT_101);
                int i1 = border_interpolate(2 * x - 1, sw, VS_BORDER_REFLECT_101);
                int i2 = 2 * x;
                int i3 = border_interpolate(2 * x + 1, sw, VS_BORDER_REFLECT_101);
                int i4 = border_interpolate(2 * x + 2, sw, VS_BORDER_REFLECT_101);
                hrow[k][x] = s[i2] * 6 + (s[i1] + s[i3]) * 4 + s[i0] + s[i4];
            }
        }
        uint8_t* d = dst + (size_t)y * dstride;
        for (int x = 0; x < dw; x++)
            d[x] = (uint8_t)((hrow[2][x] * 6 + (hrow[1][x] + hrow[3][x]) * 4 + hrow[0][x] + hrow[4][x] + 128) >> 8);
    }
}

// calcSharrDeriv (lkpyramid.cpp): dx = [3 10 3]^T x [-1 0 1], dy = [-1 0 1]^T x
// [3 10 3]; REFLECT_101 inside the image; int16 interleaved.
void scharr(const uint8_t* src, int w, int h, size_t sstride, int16_t* dst) {
    std::vector<int> t0(w + 2), t1(w + 2);
    for (int y = 0; y < h; y++) {
        const uint8_t* r0 = src + (size_t)(y > 0 ? y - 1 : h > 1 ? 1 : 0) * sstride;
        const uint8_t* r1 = src + (size_t)y * sstride;
        const uint8_t* r2 = src + (size_t)(y < h - 1 ? y + 1 : h > 1 ? h - 2 : 0) * sstride;
        int* a = t0.data() + 1;
        int* b = t1.data() + 1;
        for (int x = 0; x < w; x++) {
            a[x] = (r0[x] + r2[x]) * 3 + r1[x] * 10;
            b[x] = r2[x] - r0[x];
        }
        int x0 = w > 1 ? 1 : 0, x1 = w > 1 ? w - 2 : 0;
        a[-1] = a[x0]; a[w] = a[x1];
        b[-1] = b[x0]; b[w] = b[x1];
        int16_t* d = dst + (size_t)y * w * 2;
        for (int x = 0; x < w; x++) {
            d[2 * x] = (int16_t)(a[x + 1] - a[x - 1]);
            d[2 * x + 1] = (int16_t)((b[x + 1] + b[x - 1]) * 3 + b[x] * 10);
        }
    }
}

void copy_make_border(const uint8_t* src, int w, int h, size_t sstride, int cn, uint8_t* dst,
                      size_t dstride, int b, int border) {
    int ow = w + 2 * b, oh = h + 2 * b;
    for (int y = 0; y < oh; y++) {
        int sy = border_interpolate(y - b, h, border);
        uint8_t* d = dst + (size_t)y * dstride;
        for (int x = 0; x < ow; x++) {
            int sx = border_interpolate(x - b, w, border);
            for (int k = 0; k < cn; k++)
                d[x * cn + k] = (sx >= 0 && sy >= 0) ? src[(size_t)sy * sstride + sx * cn + k] : 0;
        }
    }
}

// ---------------------------------------------------------------------------
// cv::warpAffine(INTER_LINEAR, BORDER_CONSTANT(0)) - classic fixed-point path
// (WarpAffineInvoker + remapBilinear<FixedPtCast<int,uchar,15>>).
//   AB_BITS = 10, INTER_BITS = 5, weights int16 summing to 32768.
struct WarpCoeffs {
    double M[6];  // inverse map
    std::vector<int> adelta, bdelta;
};

void warp_prepare(const double* Mfwd, int dst_w, WarpCoeffs& c) {
    double M[6];
    for (int i = 0; i < 6; i++) M[i] = Mfwd[i];
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11; M[1] *= -D;
    M[3] *= -D; M[4] = A22;
    double b1 = -M[0] * M[2] - M[1] * M[5];
    double b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1; M[5] = b2;
    for (int i = 0; i < 6; i++) c.M[i] = M[i];
    c.adelta.resize(dst_w);
    c.bdelta.resize(dst_w);
    for (int x = 0; x < dst_w; x++) {
        c.adelta[x] = sat_int(M[0] * x * 1024);
        c.bdelta[x] = sat_int(M[3] * x * 1024);
    }
}

// initInterTab2D(INTER_LINEAR, fixpt): exact for all entries except (0,0),
// where 32768 saturates to 32767 and the fix-up adds the missing 1 to tap 3.
static inline void bilinear_tab(int fx, int fy, int w[4]) {
    if (fx == 0 && fy == 0) { w[0] = 32767; w[1] = 0; w[2] = 0; w[3] = 1; return; }
    w[0] = (32 - fy) * (32 - fx) * 32;
    w[1] = (32 - fy) * fx * 32;
    w[2] = fy * (32 - fx) * 32;
    w[3] = fy * fx * 32;
}

void warp_rows(const uint8_t* src, int sw, int sh, size_t sstride, int cn, uint8_t* dst,
               int dw, size_t dstride, const WarpCoeffs& c, int y0, int y1, int border) {
    for (int y = y0; y < y1; y++) {
        int X0 = sat_int((c.M[1] * y + c.M[2]) * 1024) + 16;
        int Y0 = sat_int((c.M[4] * y + c.M[5]) * 1024) + 16;
        uint8_t* d = dst + (size_t)y * dstride;
        for (int x = 0; x < dw; x++) {
            int X = (X0 + c.adelta[x]) >> 5;
            int Y = (Y0 + c.bdelta[x]) >> 5;
            int sx = sat_short(X >> 5), sy = sat_short(Y >> 5);
            int w[4];
            bilinear_tab(X & 31, Y & 31, w);
            if (border != VS_BORDER_BLACK) {
                // remapBilinear, borders other than CONSTANT: every tap coordinate goes through borderInterpolate
                // (REPLICATE: clip()), each axis by itself
                const int sx0 = border_interpolate(sx, sw, border), sx1 = border_interpolate(sx + 1, sw, border);
                const int sy0 = border_interpolate(sy, sh, border), sy1 = border_interpolate(sy + 1, sh, border);
                for (int k = 0; k < cn; k++) {
                    int t = src[(size_t)sy0 * sstride + sx0 * cn + k] * w[0] + src[(size_t)sy0 * sstride + sx1 * cn + k] * w[1] +
                            src[(size_t)sy1 * sstride + sx0 * cn + k] * w[2] + src[(size_t)sy1 * sstride + sx1 * cn + k] * w[3];
                    d[x * cn + k] = sat_u8((t + (1 << 14)) >> 15);
                }
                continue;
            }
            bool x0in = (unsigned)sx < (unsigned)sw, x1in = (unsigned)(sx + 1) < (unsigned)sw;
            bool y0in = (unsigned)sy < (unsigned)sh, y1in = (unsigned)(sy + 1) < (unsigned)sh;
            for (int k = 0; k < cn; k++) {
                int v0 = (x0in && y0in) ? src[(size_t)sy * sstride + sx * cn + k] : 0;
                int v1 = (x1in && y0in) ? src[(size_t)sy * sstride + (sx + 1) * cn + k] : 0;
                int v2 = (x0in && y1in) ? src[(size_t)(sy + 1) * sstride + sx * cn + k] : 0;
                int v3 = (x1in && y1in) ? src[(size_t)(sy + 1) * sstride + (sx + 1) * cn + k] : 0;
                int t = v0 * w[0] + v1 * w[1] + v2 * w[2] + v3 * w[3];
                d[x * cn + k] = sat_u8((t + (1 << 14)) >> 15);
            }
        }
    }
}

// forward matrix in double (cv::warpAffine converts its CV_32F / CV_64F argument to double first)
void warp_affine_d(const uint8_t* src, int sw, int sh, size_t sstride, int cn, uint8_t* dst, int dw, int dh,
                   size_t dstride, const double* M, int border, int nthreads) {
    WarpCoeffs c;
    warp_prepare(M, dw, c);
    if (nthreads <= 1) {
        warp_rows(src, sw, sh, sstride, cn, dst, dw, dstride, c, 0, dh, border);
        return;
    }
    std::vector<std::thread> th;
    int per = (dh + nthreads - 1) / nthreads;
    for (int t = 0; t < nthreads; t++) {
        int y0 = t * per, y1 = std::min(dh, y0 + per);
        if (y0 >= y1) break;
        th.emplace_back([=, &c] { warp_rows(src, sw, sh, sstride, cn, dst, dw, dstride, c, y0, y1, border); });
    }
    for (auto& t : th) t.join();
}

void warp_affine(const uint8_t* src, int w, int h, size_t sstride, int cn, uint8_t* dst,
                 size_t dstride, const float* M, int nthreads) {
    double Md[6];
    for (int i = 0; i < 6; i++) Md[i] = (double)M[i];
    warp_affine_d(src, w, h, sstride, cn, dst, w, h, dstride, Md, VS_BORDER_BLACK, nthreads);
}

}  // namespace vso

using namespace vso;

extern "C" {
void vso_resize_linear_u8(const uint8_t* src, int sw, int sh, size_t sstride, int cn,
                          uint8_t* dst, int dw, int dh, size_t dstride) {
    resize_linear_u8(src, sw, sh, sstride, cn, dst, dw, dh, dstride);
}
void vso_bgr2gray(const uint8_t* src, int w, int h, size_t sstride, uint8_t* dst, size_t dstride) {
    bgr2gray(src, w, h, sstride, dst, dstride);
}
void vso_pyr_down(const uint8_t* src, int sw, int sh, size_t sstride, uint8_t* dst, size_t dstride) {
    pyr_down(src, sw, sh, sstride, dst, dstride);
}
void vso_scharr(const uint8_t* src, int w, int h, size_t sstride, int16_t* dst) {
    scharr(src, w, h, sstride, dst);
}
void vso_copy_make_border(const uint8_t* src, int w, int h, size_t sstride, int cn, uint8_t* dst,
                          size_t dstride, int b, int border) {
    copy_make_border(src, w, h, sstride, cn, dst, dstride, b, border);
}
void vso_warp_affine(const uint8_t* src, int w, int h, size_t sstride, int cn, uint8_t* dst,
                     size_t dstride, const float* M) {
    warp_affine(src, w, h, sstride, cn, dst, dstride, M, 1);
}
void vso_warp_affine_mt(const uint8_t* src, int w, int h, size_t sstride, int cn, uint8_t* dst,
                        size_t dstride, const float* M, int nthreads) {
    if (nthreads <= 0) nthreads = (int)std::thread::hardware_concurrency();
    warp_affine(src, w, h, sstride, cn, dst, dstride, M, nthreads);
}
void vso_warp_affine_nv12(const uint8_t* src, int w, int h, size_t sstride, uint8_t* dst,
                          size_t dstride, const float* M) {
    // Y plane: full matrix.
    warp_affine(src, w, h, sstride, 1, dst, dstride, M, 1);
    // UV plane: (w/2 x h/2) two-channel image; same rotation, translation halved.
    float Mc[6] = {M[0], M[1], M[2] * 0.5f, M[3], M[4], M[5] * 0.5f};
    warp_affine(src + (size_t)h * sstride, w / 2, h / 2, sstride, 2, dst + (size_t)h * dstride,
                dstride, Mc, 1);
}
}
