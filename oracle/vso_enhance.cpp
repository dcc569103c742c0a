// vso_enhance.cpp - CPU ORACLE (test infrastructure only, see vso.h) for
// vs::Enhancer::enhanceImage (/root/reference/src/Enhancer.cpp:138-239) and the OpenCV 4.11
// primitives it calls.  PARITY UNPINNED like the rest of oracle/: the reference has no fixtures
// for this path and cannot be built here, so every primitive below is a restatement of the
// published OpenCV algorithm, pinned by analytic known-answer tests only
// (tests/test_enhance.py).
//
// Definitions fixed where OpenCV's result depends on the build (SIMD body vs scalar tail, FMA):
//  * convertTo / `Mat *= double` on CV_8U (convert_scale.simd.hpp cvt_32f): float a,b and ONE fused
//    multiply-add per sample, round-half-even, saturate (the AVX2 and aarch64 NEON bodies).
//  * addWeighted on CV_8U (arithm.simd.hpp op_add_weighted): fma(a, alpha, fma(b, beta, gamma)) in float.
//  * cvtColor HSV2BGR 8U: the vector body's formulas (v - v*s, v - v*s*h, v - v*s + v*s*h).
//  * CLAHE interpolation and the vibrance loop: plain float multiply/add, no contraction.
//  * Lab tables: float arithmetic with libm powf/cbrt where OpenCV uses softfloat.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <vector>

#include "vso.h"
#include "vso_internal.h"

namespace {

inline int rne(float v) { return (int)lrintf(v); }     // cvRound (default rounding mode: nearest even)
inline int rne(double v) { return (int)lrint(v); }
inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
inline int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }   // CV_DESCALE

// cv::borderInterpolate(p, len, BORDER_REFLECT_101)
inline int reflect101(int p, int len) {
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

// ---------------------------------------------------------------------------------------------
// color tables (color_lab.cpp initLabTabs)
// ---------------------------------------------------------------------------------------------
constexpr int kLabShift = 12, kGammaShift = 3, kLabShift2 = kLabShift + kGammaShift;
constexpr int kCbrtTabSize = 256 * 3 / 2 * (1 << kGammaShift);
constexpr int kBaseShift = 14, kBase = 1 << kBaseShift;
constexpr int kInvGammaShift = 12, kInvGammaTabSize = 1 << kInvGammaShift;
constexpr int kMinAB = -8145;
constexpr int kAbTabSize = kBase * 9 / 4;

struct LabTabs {
    uint16_t gamma[256];              // sRGBGammaTab_b
    uint16_t cbrt_[kCbrtTabSize];     // LabCbrtTab_b
    int fwd[9];                       // RGB2Lab_b coeffs for B,G,R source order
    uint16_t l2yf[512];               // LabToYF_b
    int ab2xz[kAbTabSize];            // abToXZ_b
    uint16_t inv_gamma[kInvGammaTabSize];  // sRGBInvGammaTab_b
    int inv[9];                       // Lab2RGBinteger coeffs for B,G,R destination order
    uint16_t lin_gamma[256];          // linearGammaTab_b (COLOR_LBGR2Lab)
    uint16_t lin_inv_gamma[kInvGammaTabSize];  // linearInvGammaTab_b (COLOR_Lab2LBGR)
};

const LabTabs& lab_tabs() {
    static LabTabs T;
    static std::once_flag once;
    std::call_once(once, [] {
        static const double D65[3] = {0.950456, 1.0, 1.088754};
        static const double rgb2xyz[9] = {0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169,
                                          0.019334, 0.119193, 0.950227};
        static const double xyz2rgb[9] = {3.240479, -1.53715, -0.498535, -0.969256, 1.875991, 0.041556,
                                          0.055648, -0.204043, 1.057311};
        for (int i = 0; i < 256; i++) {
            float x = (float)i / 255.f;
            float g = x <= 0.04045f ? x * (1.f / 12.92f) : powf((x + 0.055f) * (1.f / 1.055f), 2.4f);
            T.gamma[i] = (uint16_t)rne((float)(255 * (1 << kGammaShift)) * g);
        }
        for (int i = 0; i < kCbrtTabSize; i++) {
            float x = (float)i / (float)(255 * (1 << kGammaShift));
            float f = x < 0.008856f ? x * 7.787f + 0.13793103448275862f : (float)cbrt((double)x);
            T.cbrt_[i] = (uint16_t)rne((float)(1 << kLabShift2) * f);
        }
        for (int i = 0; i < 3; i++) {   // source order B,G,R (blueIdx = 0)
            T.fwd[i * 3 + 2] = rne((double)(1 << kLabShift) * rgb2xyz[i * 3 + 0] / D65[i]);
            T.fwd[i * 3 + 1] = rne((double)(1 << kLabShift) * rgb2xyz[i * 3 + 1] / D65[i]);
            T.fwd[i * 3 + 0] = rne((double)(1 << kLabShift) * rgb2xyz[i * 3 + 2] / D65[i]);
        }
        for (int i = 0; i < 256; i++) {
            int y, ify;
            if (i <= 20) {   // L* <= 8
                y = rne((float)(i * kBase * 20 * 9) / (float)(17 * 29 * 29 * 29));
                ify = rne((float)kBase * ((float)16 / (float)116 + (float)(i * 5) / (float)(3 * 17 * 29)));
            } else {
                float fy = (float)(i * 100 * kBase) / (float)(255 * 116) + (float)(16 * kBase) / (float)116;
                ify = rne(fy);
                y = rne(fy * fy * fy / (float)(kBase * kBase));
            }
            T.l2yf[i * 2] = (uint16_t)y;
            T.l2yf[i * 2 + 1] = (uint16_t)ify;
        }
        for (int i = kMinAB; i < kAbTabSize + kMinAB; i++) {
            int v;
            if (i <= 3390) v = i * 108 / 841 - kBase * 16 / 116 * 108 / 841;
            else v = i * i / kBase * i / kBase;
            T.ab2xz[i - kMinAB] = v;
        }
        for (int i = 0; i < kInvGammaTabSize; i++) {
            float x = (float)i / (float)(kInvGammaTabSize - 1);
            float g = x <= 0.0031308f ? x * 12.92f : 1.055f * powf(x, 1.f / 2.4f) - 0.055f;
            T.inv_gamma[i] = (uint16_t)rne(255.f * g);
            T.lin_inv_gamma[i] = (uint16_t)(int)(255.f * x);          // cvTrunc
        }
        for (int i = 0; i < 256; i++) T.lin_gamma[i] = (uint16_t)(i * (1 << kGammaShift));
        for (int i = 0; i < 3; i++) {   // column i of XYZ; rows ordered so that out0 = R, out1 = G, out2 = B
            T.inv[i + 0] = rne((double)(1 << kLabShift) * xyz2rgb[0 * 3 + i] * D65[i]);
            T.inv[i + 3] = rne((double)(1 << kLabShift) * xyz2rgb[1 * 3 + i] * D65[i]);
            T.inv[i + 6] = rne((double)(1 << kLabShift) * xyz2rgb[2 * 3 + i] * D65[i]);
        }
    });
    return T;
}

// RGB2Lab_b::operator() scalar body, one pixel
inline void bgr2lab_px(const LabTabs& T, const uint8_t* s, uint8_t* d, bool srgb = true) {
    const int Lscale = (116 * 255 + 50) / 100;
    const int Lshift = -((16 * 255 * (1 << kLabShift2) + 50) / 100);
    const int* C = T.fwd;
    const uint16_t* gt = srgb ? T.gamma : T.lin_gamma;
    int B = gt[s[0]], G = gt[s[1]], R = gt[s[2]];
    int fX = T.cbrt_[descale(B * C[0] + G * C[1] + R * C[2], kLabShift)];
    int fY = T.cbrt_[descale(B * C[3] + G * C[4] + R * C[5], kLabShift)];
    int fZ = T.cbrt_[descale(B * C[6] + G * C[7] + R * C[8], kLabShift)];
    int L = descale(Lscale * fY + Lshift, kLabShift2);
    int a = descale(500 * (fX - fY) + 128 * (1 << kLabShift2), kLabShift2);
    int b = descale(200 * (fY - fZ) + 128 * (1 << kLabShift2), kLabShift2);
    d[0] = sat_u8(L); d[1] = sat_u8(a); d[2] = sat_u8(b);
}

// Lab2RGBinteger::process + Lab2RGB_b store, one pixel
inline void lab2bgr_px(const LabTabs& T, const uint8_t* s, uint8_t* d, bool srgb = true) {
    const uint16_t* igt = srgb ? T.inv_gamma : T.lin_inv_gamma;
    const int LL = s[0], aa = s[1], bb = s[2];
    int y = T.l2yf[LL * 2], ify = T.l2yf[LL * 2 + 1];
    int adiv = ((5 * aa * 53687 + (1 << 7)) >> 13) - 128 * kBase / 500;
    int bdiv = ((bb * 41943 + (1 << 4)) >> 9) - 128 * kBase / 200 + 1;
    int x = T.ab2xz[ify + adiv - kMinAB];
    int z = T.ab2xz[ify - bdiv - kMinAB];
    const int* C = T.inv;
    const int shift = kLabShift + (kBaseShift - kInvGammaShift);
    int ro = descale(C[0] * x + C[1] * y + C[2] * z, shift);
    int go = descale(C[3] * x + C[4] * y + C[5] * z, shift);
    int bo = descale(C[6] * x + C[7] * y + C[8] * z, shift);
    ro = std::max(0, std::min(kInvGammaTabSize - 1, ro));
    go = std::max(0, std::min(kInvGammaTabSize - 1, go));
    bo = std::max(0, std::min(kInvGammaTabSize - 1, bo));
    d[0] = sat_u8(igt[bo]); d[1] = sat_u8(igt[go]); d[2] = sat_u8(igt[ro]);
}

// ---------------------------------------------------------------------------------------------
// HSV (color_hsv.simd.hpp RGB2HSV_b / HSV2RGB_b, hrange 180)
// ---------------------------------------------------------------------------------------------
struct HsvTabs { int sdiv[256], hdiv[256]; };
const HsvTabs& hsv_tabs() {
    static HsvTabs T;
    static std::once_flag once;
    std::call_once(once, [] {
        const int hsv_shift = 12;
        T.sdiv[0] = T.hdiv[0] = 0;
        for (int i = 1; i < 256; i++) {
            T.sdiv[i] = rne((255 << hsv_shift) / (1. * i));
            T.hdiv[i] = rne((180 << hsv_shift) / (6. * i));
        }
    });
    return T;
}

inline void bgr2hsv_px(const HsvTabs& T, const uint8_t* s, uint8_t* d) {
    const int hsv_shift = 12;
    int b = s[0], g = s[1], r = s[2];
    int v = std::max(b, std::max(g, r)), vmin = std::min(b, std::min(g, r));
    int diff = v - vmin;
    int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
    int sv = (diff * T.sdiv[v] + (1 << (hsv_shift - 1))) >> hsv_shift;
    int h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
    h = (h * T.hdiv[diff] + (1 << (hsv_shift - 1))) >> hsv_shift;
    h += h < 0 ? 180 : 0;
    d[0] = sat_u8(h); d[1] = (uint8_t)sv; d[2] = (uint8_t)v;
}

inline void hsv2bgr_px(const uint8_t* s, uint8_t* d) {
    float h = (float)s[0] * (6.f / 180.f);
    float sf = (float)s[1] * (1.f / 255.f), v = (float)s[2] * (1.f / 255.f);
    float pre = (float)(int)h;          // v_trunc
    h = h - pre;
    float vs = v * sf;
    float t1 = v - vs;
    float t2 = v - vs * h;
    float t3 = (v - vs) + vs * h;
    int sector = (int)pre % 6;
    float b, g, r;
    switch (sector) {                    // sector_data {1,3,0},{1,0,2},{3,0,1},{0,2,1},{0,1,3},{2,1,0} as (b,g,r) of tab[]
        case 0: b = t1; g = t3; r = v; break;
        case 1: b = t1; g = v; r = t2; break;
        case 2: b = t3; g = v; r = t1; break;
        case 3: b = v; g = t2; r = t1; break;
        case 4: b = v; g = t1; r = t3; break;
        default: b = t2; g = t1; r = v; break;
    }
    d[0] = sat_u8(rne(b * 255.f)); d[1] = sat_u8(rne(g * 255.f)); d[2] = sat_u8(rne(r * 255.f));
}

// ---------------------------------------------------------------------------------------------
// GaussianBlur CV_8U, bit-exact fixed-point path (smooth.dispatch.cpp / smooth.simd.hpp)
// ---------------------------------------------------------------------------------------------
// getGaussianKernelBitExact + getGaussianKernelFixedPoint_ED (8 fractional bits, odd n)
std::vector<uint16_t> gaussian_kernel_q8(int n, double sigma) {
    const int n2 = (n - 1) / 2;
    std::vector<double> vals(n2 + 1);
    const double scale2x = -0.125 / (sigma * sigma);
    double sum = 0;
    for (int i = 0, x = 1 - n; i < n2; i++, x += 2) {
        double t = std::exp((double)(x * x) * scale2x);
        vals[i] = t;
        sum += t;
    }
    sum *= 2;
    sum += 1.0;
    const double mul1 = 1.0 / sum;
    std::vector<double> k(n);
    for (int i = 0; i < n2; i++) k[i] = k[n - 1 - i] = vals[i] * mul1;
    k[n2] = mul1;
    std::vector<uint16_t> q(n);
    double err = 0;
    int64_t isum = 0;
    for (int i = 0; i < n2; i++) {
        double adj = k[i] * 256.0 + err;
        int64_t v0 = (int64_t)lrint(adj);
        err = adj - (double)v0;
        int64_t v = std::max<int64_t>(0, std::min<int64_t>(65535, v0));
        isum += v;
        q[i] = q[n - 1 - i] = (uint16_t)v;
    }
    isum *= 2;
    q[n2] = (uint16_t)(256 - isum);
    return q;
}

}  // namespace

extern "C" {

void vso_enh_params_default(vs_enh_params_c* p) {   // Enhancer.h:12-43
    if (!p) return;
    memset(p, 0, sizeof *p);
    p->struct_size = (int32_t)sizeof *p;
    p->brightness = 0.f; p->contrast = 1.f;
    p->enable_white_balance = 0; p->wb_strength = 1.f;
    p->enable_vibrance = 0; p->vibrance_strength = 0.3f;
    p->enable_unsharp = 0; p->sharpness = 0.f; p->blur_sigma = 1.f;
    p->enable_clahe = 0; p->clahe_clip_limit = 2.f; p->clahe_tile_grid_size = 8;
    p->enable_denoise = 0; p->denoise_strength = 10.f;
    p->gamma = 1.f;
    p->use_cuda = 0;
}

// src.convertTo(dst, -1, alpha, beta) for CV_8U as a 256-entry table (Enhancer.cpp:37-39,150)
void vso_convert_scale_lut(double alpha, double beta, uint8_t* lut256) {
    const float a = (float)alpha, b = (float)beta;
    for (int i = 0; i < 256; i++) lut256[i] = sat_u8(rne(fmaf((float)i, a, b)));
}

// whiteBalanceCPU scale factors (Enhancer.cpp:22-36); sums = per-channel pixel sums (B,G,R)
void vso_wb_scales(const uint64_t* sums, uint64_t npix, float alpha, double* scales) {
    double m[3];
    for (int c = 0; c < 3; c++) m[c] = (double)sums[c] / (double)npix;
    double gray = (m[0] + m[1] + m[2]) / 3.0;
    for (int c = 0; c < 3; c++) {
        double s = gray / (m[c] + 1e-6);
        scales[c] = 1.0 + alpha * (s - 1.0);
    }
}

// gamma table (Enhancer.cpp:171-178)
void vso_gamma_lut(float gamma, uint8_t* lut256) {
    for (int i = 0; i < 256; i++) {
        float norm = i / 255.f;
        float corrected = std::pow(norm, gamma);
        lut256[i] = sat_u8(rne(corrected * 255.f));
    }
}

void vso_bgr2hsv(const uint8_t* src, size_t n, uint8_t* dst) {
    const HsvTabs& T = hsv_tabs();
    for (size_t i = 0; i < n; i++) bgr2hsv_px(T, src + 3 * i, dst + 3 * i);
}
void vso_hsv2bgr(const uint8_t* src, size_t n, uint8_t* dst) {
    for (size_t i = 0; i < n; i++) hsv2bgr_px(src + 3 * i, dst + 3 * i);
}
void vso_bgr2lab(const uint8_t* src, size_t n, uint8_t* dst) {
    const LabTabs& T = lab_tabs();
    for (size_t i = 0; i < n; i++) bgr2lab_px(T, src + 3 * i, dst + 3 * i);
}
void vso_lab2bgr(const uint8_t* src, size_t n, uint8_t* dst) {
    const LabTabs& T = lab_tabs();
    for (size_t i = 0; i < n; i++) lab2bgr_px(T, src + 3 * i, dst + 3 * i);
}

// vibranceCPU (Enhancer.cpp:41-57), in place on packed BGR
void vso_vibrance(uint8_t* bgr, size_t n, float alpha) {
    const HsvTabs& T = hsv_tabs();
    for (size_t i = 0; i < n; i++) {
        uint8_t hsv[3];
        bgr2hsv_px(T, bgr + 3 * i, hsv);
        float s = (float)hsv[1];
        s += alpha * (255.f - s);
        hsv[1] = sat_u8(rne(s));
        hsv2bgr_px(hsv, bgr + 3 * i);
    }
}

// ksize chosen by GaussianBlur(Size(0,0), sigma) for CV_8U, and the Q8 kernel
int vso_gaussian_kernel_q8(double sigma, uint16_t* k, int cap) {
    if (!(sigma > 0)) return 0;
    int n = rne(sigma * 3 * 2 + 1) | 1;
    if (k) {
        if (n > cap) return -n;
        auto q = gaussian_kernel_q8(n, sigma);
        memcpy(k, q.data(), n * sizeof(uint16_t));
    }
    return n;
}

// cv::GaussianBlur(src, dst, Size(0,0), sigma) CV_8UC(cn), BORDER_DEFAULT (Enhancer.cpp:160-161)
int vso_gaussian_blur_u8(const uint8_t* src, int w, int h, size_t stride, int cn, double sigma, uint8_t* dst, size_t dstride) {
    if (!(sigma > 0)) return -1;
    const int n = rne(sigma * 3 * 2 + 1) | 1, r = n / 2;
    auto k = gaussian_kernel_q8(n, sigma);
    std::vector<uint16_t> tmp((size_t)w * cn * h);
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src + (size_t)y * stride;
        uint16_t* t = tmp.data() + (size_t)y * w * cn;
        for (int x = 0; x < w; x++)
            for (int c = 0; c < cn; c++) {
                uint32_t acc = 0;
                for (int j = 0; j < n; j++) acc += (uint32_t)k[j] * s[reflect101(x + j - r, w) * cn + c];
                t[x * cn + c] = (uint16_t)std::min<uint32_t>(acc, 65535u);
            }
    }
    for (int y = 0; y < h; y++) {
        uint8_t* d = dst + (size_t)y * dstride;
        for (int i = 0; i < w * cn; i++) {
            uint32_t acc = 0;
            for (int j = 0; j < n; j++) acc += (uint32_t)k[j] * tmp[(size_t)reflect101(y + j - r, h) * w * cn + i];
            d[i] = sat_u8((int)((acc + 32768u) >> 16));
        }
    }
    return n;
}

// cv::addWeighted(a, alpha, b, beta, gamma, dst) CV_8U (Enhancer.cpp:162)
void vso_add_weighted_u8(const uint8_t* a, double alpha, const uint8_t* b, double beta, double gamma, uint8_t* dst, size_t n) {
    const float fa = (float)alpha, fb = (float)beta, fg = (float)gamma;
    for (size_t i = 0; i < n; i++) dst[i] = sat_u8(rne(fmaf((float)a[i], fa, fmaf((float)b[i], fb, fg))));
}

// cv::createCLAHE(clip, Size(tiles,tiles))->apply(src, dst) for one CV_8UC1 plane (clahe.cpp)
int vso_clahe_u8(const uint8_t* src, int w, int h, size_t stride, double clip_limit, int tiles, uint8_t* dst, size_t dstride,
                 uint8_t* lut_out) {
    if (tiles <= 0 || w <= 0 || h <= 0) return -1;
    const int hist_size = 256;
    int ew = w, eh = h;
    if (w % tiles != 0 || h % tiles != 0) { ew = w + (tiles - w % tiles); eh = h + (tiles - h % tiles); }
    const int tw = ew / tiles, th = eh / tiles;
    if (tw <= 0 || th <= 0) return -1;
    const int tile_total = tw * th;
    const float lut_scale = (float)(hist_size - 1) / tile_total;
    int clip = 0;
    if (clip_limit > 0.0) {
        clip = (int)(clip_limit * tile_total / hist_size);
        clip = std::max(clip, 1);
    }
    std::vector<uint8_t> lut((size_t)tiles * tiles * hist_size);
    for (int k = 0; k < tiles * tiles; k++) {
        const int ty = k / tiles, tx = k % tiles;
        int hist[256] = {0};
        for (int y = ty * th; y < (ty + 1) * th; y++) {
            const uint8_t* row = src + (size_t)reflect101(y, h) * stride;   // copyMakeBorder(0, pad, 0, pad, REFLECT_101)
            for (int x = tx * tw; x < (tx + 1) * tw; x++) hist[row[reflect101(x, w)]]++;
        }
        if (clip > 0) {
            int clipped = 0;
            for (int i = 0; i < hist_size; i++)
                if (hist[i] > clip) { clipped += hist[i] - clip; hist[i] = clip; }
            int batch = clipped / hist_size, residual = clipped - batch * hist_size;
            for (int i = 0; i < hist_size; i++) hist[i] += batch;
            if (residual != 0) {
                int step = std::max(hist_size / residual, 1);
                for (int i = 0; i < hist_size && residual > 0; i += step, residual--) hist[i]++;
            }
        }
        int sum = 0;
        uint8_t* tl = lut.data() + (size_t)k * hist_size;
        for (int i = 0; i < hist_size; i++) { sum += hist[i]; tl[i] = sat_u8(rne(sum * lut_scale)); }
    }
    if (lut_out) memcpy(lut_out, lut.data(), lut.size());
    const float inv_tw = 1.0f / tw, inv_th = 1.0f / th;
    for (int y = 0; y < h; y++) {
        float tyf = y * inv_th - 0.5f;
        int ty1 = (int)floorf(tyf), ty2 = ty1 + 1;
        float ya = tyf - ty1, ya1 = 1.0f - ya;
        ty1 = std::max(ty1, 0); ty2 = std::min(ty2, tiles - 1);
        const uint8_t* p1 = lut.data() + (size_t)ty1 * tiles * hist_size;
        const uint8_t* p2 = lut.data() + (size_t)ty2 * tiles * hist_size;
        for (int x = 0; x < w; x++) {
            float txf = x * inv_tw - 0.5f;
            int tx1 = (int)floorf(txf), tx2 = tx1 + 1;
            float xa = txf - tx1, xa1 = 1.0f - xa;
            tx1 = std::max(tx1, 0); tx2 = std::min(tx2, tiles - 1);
            int v = src[(size_t)y * stride + x];
            int i1 = tx1 * hist_size + v, i2 = tx2 * hist_size + v;
            float res = (p1[i1] * xa1 + p1[i2] * xa) * ya1 + (p2[i1] * xa1 + p2[i2] * xa) * ya;
            dst[(size_t)y * dstride + x] = sat_u8(rne(res));
        }
    }
    return 0;
}

// applyClaheCPU (Enhancer.cpp:59-69), in place on packed BGR
int vso_clahe_bgr(uint8_t* bgr, int w, int h, float clip_limit, int tiles) {
    const size_t n = (size_t)w * h;
    std::vector<uint8_t> lab(n * 3), L(n), L2(n);
    vso_bgr2lab(bgr, n, lab.data());
    for (size_t i = 0; i < n; i++) L[i] = lab[3 * i];
    int rc = vso_clahe_u8(L.data(), w, h, w, clip_limit, tiles, L2.data(), w, nullptr);
    if (rc) return rc;
    for (size_t i = 0; i < n; i++) lab[3 * i] = L2[i];
    vso_lab2bgr(lab.data(), n, bgr);
    return 0;
}

// COLOR_LBGR2Lab / COLOR_Lab2LBGR (linear RGB, no sRGB curve): what fastNlMeansDenoisingColored converts with
void vso_lbgr2lab(const uint8_t* src, size_t n, uint8_t* dst) {
    const LabTabs& T = lab_tabs();
    for (size_t i = 0; i < n; i++) bgr2lab_px(T, src + 3 * i, dst + 3 * i, false);
}
void vso_lab2lbgr(const uint8_t* src, size_t n, uint8_t* dst) {
    const LabTabs& T = lab_tabs();
    for (size_t i = 0; i < n; i++) lab2bgr_px(T, src + 3 * i, dst + 3 * i, false);
}

// almost_dist2weight_ of FastNlMeansDenoisingInvoker<.., DistSquared, int> (fast_nlmeans_denoising_invoker.hpp):
// weights for (sum of squared differences over the template) >> shift.  Returns the table length.
int vso_nlm_weights(float h, int cn, int template_size, int search_size, int32_t* table, int cap, int32_t* shift_out) {
    const int tsq = template_size * template_size;
    int shift = 0;
    while ((1 << shift) < tsq) shift++;
    const double mult = (double)(1 << shift) / tsq;                     // almost_dist2actual_dist_multiplier
    const int max_dist = 255 * 255 * cn;
    const int almost_max = (int)(max_dist / mult + 1);
    const int max_est = search_size * search_size * 255;
    const int fixed_point_mult = (int)std::min<long long>(2147483647LL / max_est, 2147483647LL);
    if (shift_out) { shift_out[0] = shift; shift_out[1] = fixed_point_mult; }
    if (table) {
        for (int a = 0; a < almost_max && a < cap; a++) {
            const double dist = a * mult;
            double w = std::exp(-dist / (h * h * cn));                  // float h*h, then * channels
            if (std::isnan(w)) w = 1.0;
            int weight = rne(fixed_point_mult * w);
            if (weight < 0.001 * fixed_point_mult) weight = 0;
            table[a] = weight;
        }
    }
    return almost_max;
}

// cv::fastNlMeansDenoising(src, dst, h, 7, 21) for CV_8UC(cn), cn = 1 or 2 (photo/src/denoising.cpp):
// every pixel becomes the weighted mean of the pixels of its 21x21 search window, weights from the 7x7
// patch distance.  Evaluated by brute force; OpenCV's sliding sums are exact integer arithmetic.
int vso_fast_nl_means(const uint8_t* src, int w, int h, size_t stride, int cn, float hp, int template_size, int search_size,
                      uint8_t* dst, size_t dstride) {
    if (cn < 1 || cn > 2 || w <= 0 || h <= 0 || !(template_size & 1) || !(search_size & 1)) return -1;
    const int th = template_size / 2, sh = search_size / 2, border = th + sh;
    int32_t info[2];
    const int n_tab = vso_nlm_weights(hp, cn, template_size, search_size, nullptr, 0, info);
    std::vector<int32_t> tab(n_tab);
    vso_nlm_weights(hp, cn, template_size, search_size, tab.data(), n_tab, info);
    const int shift = info[0];
    const int ew = w + 2 * border, eh = h + 2 * border;
    std::vector<uint8_t> ext((size_t)ew * eh * cn);                    // copyMakeBorder(BORDER_DEFAULT)
    for (int y = 0; y < eh; y++)
        for (int x = 0; x < ew; x++)
            for (int c = 0; c < cn; c++)
                ext[((size_t)y * ew + x) * cn + c] = src[(size_t)reflect101(y - border, h) * stride + (size_t)reflect101(x - border, w) * cn + c];
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            long long est[2] = {0, 0};
            long long wsum = 0;
            for (int y = -sh; y <= sh; y++)
                for (int x = -sh; x <= sh; x++) {
                    int dist = 0;
                    for (int ty = -th; ty <= th; ty++) {
                        const uint8_t* a = &ext[((size_t)(border + i + ty) * ew + (border + j - th)) * cn];
                        const uint8_t* b = &ext[((size_t)(border + i + y + ty) * ew + (border + j + x - th)) * cn];
                        for (int k = 0; k < template_size * cn; k++) { const int d = (int)a[k] - (int)b[k]; dist += d * d; }
                    }
                    const int wgt = tab[dist >> shift];
                    const uint8_t* p = &ext[((size_t)(border + i + y) * ew + (border + j + x)) * cn];
                    for (int c = 0; c < cn; c++) est[c] += (long long)wgt * p[c];
                    wsum += wgt;
                }
            for (int c = 0; c < cn; c++) {
                const unsigned v = ((unsigned)est[c] + (unsigned)wsum / 2) / (unsigned)wsum;     // divByWeightsSum
                dst[(size_t)i * dstride + (size_t)j * cn + c] = sat_u8((int)v);
            }
        }
    return 0;
}

// cv::fastNlMeansDenoisingColored(img, img, h, hColor, 7, 21) (Enhancer.cpp:165-169), in place on packed BGR
int vso_denoise_colored(uint8_t* bgr, int w, int h, float hl, float hc) {
    const size_t n = (size_t)w * h;
    std::vector<uint8_t> lab(n * 3), L(n), ab(n * 2), L2(n), ab2(n * 2);
    vso_lbgr2lab(bgr, n, lab.data());
    for (size_t i = 0; i < n; i++) { L[i] = lab[3 * i]; ab[2 * i] = lab[3 * i + 1]; ab[2 * i + 1] = lab[3 * i + 2]; }
    if (vso_fast_nl_means(L.data(), w, h, w, 1, hl, 7, 21, L2.data(), w)) return -1;
    if (vso_fast_nl_means(ab.data(), w, h, (size_t)w * 2, 2, hc, 7, 21, ab2.data(), (size_t)w * 2)) return -1;
    for (size_t i = 0; i < n; i++) { lab[3 * i] = L2[i]; lab[3 * i + 1] = ab2[2 * i]; lab[3 * i + 2] = ab2[2 * i + 1]; }
    vso_lab2lbgr(lab.data(), n, bgr);
    return 0;
}


// Enhancer::enhanceImage (Enhancer.cpp:138-239) for a BGR8 frame.  use_cuda selects the stage ORDER of the
// reference's CUDA branch (:183-233); the arithmetic of each stage is the CPU primitive in both cases.
// Returns 0, -1 bad argument.
int vso_enhance(const uint8_t* src, int w, int h, size_t stride, const vs_enh_params_c* p, uint8_t* out, size_t out_stride) {
    if (!src || !out || !p || w <= 0 || h <= 0) return -1;
    const size_t n = (size_t)w * h;
    std::vector<uint8_t> img(n * 3);
    for (int y = 0; y < h; y++) memcpy(img.data() + (size_t)y * w * 3, src + (size_t)y * stride, (size_t)w * 3);

    auto apply_lut3 = [&](const uint8_t* l0, const uint8_t* l1, const uint8_t* l2) {
        for (size_t i = 0; i < n; i++) { img[3 * i] = l0[img[3 * i]]; img[3 * i + 1] = l1[img[3 * i + 1]]; img[3 * i + 2] = l2[img[3 * i + 2]]; }
    };
    auto wb = [&] {   // :22-39 / :71-97
        uint64_t sums[3] = {0, 0, 0};
        for (size_t i = 0; i < n; i++) { sums[0] += img[3 * i]; sums[1] += img[3 * i + 1]; sums[2] += img[3 * i + 2]; }
        double sc[3];
        vso_wb_scales(sums, n, p->wb_strength, sc);
        uint8_t l[3][256];
        for (int c = 0; c < 3; c++) vso_convert_scale_lut(sc[c], 0.0, l[c]);
        apply_lut3(l[0], l[1], l[2]);
    };
    auto cb = [&] {   // :150 / :186
        uint8_t l[256];
        vso_convert_scale_lut((double)p->contrast, (double)p->brightness, l);
        apply_lut3(l, l, l);
    };
    auto unsharp = [&]() -> int {   // :159-163 / :99-106
        std::vector<uint8_t> blurred(n * 3);
        if (vso_gaussian_blur_u8(img.data(), w, h, (size_t)w * 3, 3, (double)p->blur_sigma, blurred.data(), (size_t)w * 3) < 0) return -1;
        vso_add_weighted_u8(img.data(), 1.0 + p->sharpness, blurred.data(), -(double)p->sharpness, 0.0, img.data(), n * 3);
        return 0;
    };
    auto gamma = [&] {   // :171-180 / :214-227
        uint8_t l[256];
        vso_gamma_lut(p->gamma, l);
        apply_lut3(l, l, l);
    };
    const bool do_unsharp = p->enable_unsharp && p->sharpness > 0.f;
    const bool do_denoise = p->enable_denoise && p->denoise_strength > 0.f;
    const bool do_gamma = std::fabs(p->gamma - 1.f) > 1e-3;
    auto denoise = [&]() -> int { return vso_denoise_colored(img.data(), w, h, p->denoise_strength, p->denoise_strength); };   // :165-169 / :108-114
    if (!p->use_cuda) {
        if (p->enable_white_balance) wb();
        cb();
        if (p->enable_clahe && vso_clahe_bgr(img.data(), w, h, p->clahe_clip_limit, p->clahe_tile_grid_size)) return -1;
        if (p->enable_vibrance) vso_vibrance(img.data(), n, p->vibrance_strength);
        if (do_unsharp && unsharp()) return -1;
        if (do_denoise && denoise()) return -1;
        if (do_gamma) gamma();
    } else {
        cb();
        if (do_unsharp && unsharp()) return -1;
        if (do_denoise && denoise()) return -1;
        if (p->enable_white_balance) wb();
        if (p->enable_vibrance) vso_vibrance(img.data(), n, p->vibrance_strength);
        if (p->enable_clahe && vso_clahe_bgr(img.data(), w, h, p->clahe_clip_limit, p->clahe_tile_grid_size)) return -1;
        if (do_gamma) gamma();
    }
    for (int y = 0; y < h; y++) memcpy(out + (size_t)y * out_stride, img.data() + (size_t)y * w * 3, (size_t)w * 3);
    return 0;
}

}  // extern "C"
