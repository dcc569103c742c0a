// CPU ORACLE (test infrastructure) - vs::RollCorrection::autoCorrectRoll restated
// (/root/reference/src/RollCorrection.cpp:16-155; parameters include/video/RollCorrection.h:16-38).
//
// The reference has NO CPU branch for this stage: it calls cv::cuda::resize /
// cvtColor / CannyEdgeDetector / HoughLinesDetector / buildWarpAffineMaps+remap.
// Parity is unpinned (vso.h).  The oracle restates the stage with the CPU
// OpenCV 4.11 definitions of the same operators - cv::resize(INTER_LINEAR),
// cv::cvtColor(BGR2GRAY), cv::Canny(L1 gradient), cv::HoughLines (standard
// transform, lines sorted by votes), cv::getRotationMatrix2D and
// cv::warpAffine(INTER_LINEAR, BORDER_REPLICATE) - i.e. what the stage computes
// where no CUDA module exists.  State (smoothed angle) is per object here; the
// reference keeps it in file-static variables (:13-14).
#include "vso_internal.h"

#include <algorithm>
#include <cstring>

namespace vso {

// cv::Sobel(src, CV_16S, ksize 3, BORDER_REPLICATE): dx = [-1 0 1]x[1 2 1]^T, dy = transpose
void sobel16(const uint8_t* g, int w, int h, size_t stride, int16_t* dx, int16_t* dy) {
    auto P = [&](int y, int x) -> int {
        y = y < 0 ? 0 : (y >= h ? h - 1 : y);
        x = x < 0 ? 0 : (x >= w ? w - 1 : x);
        return g[(size_t)y * stride + x];
    };
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int a = P(y - 1, x - 1), b = P(y - 1, x), c = P(y - 1, x + 1);
            int d = P(y, x - 1), f = P(y, x + 1);
            int g0 = P(y + 1, x - 1), h0 = P(y + 1, x), i = P(y + 1, x + 1);
            dx[(size_t)y * w + x] = (int16_t)((c + 2 * f + i) - (a + 2 * d + g0));
            dy[(size_t)y * w + x] = (int16_t)((g0 + 2 * h0 + i) - (a + 2 * b + c));
        }
}

// cv::Canny(src, edges, low, high, 3, L2gradient=false)
void canny(const uint8_t* g, int w, int h, size_t stride, double low_thresh, double high_thresh, uint8_t* edges) {
    if (low_thresh > high_thresh) std::swap(low_thresh, high_thresh);
    const int low = cv_floor(low_thresh), high = cv_floor(high_thresh);
    std::vector<int16_t> dx((size_t)w * h), dy((size_t)w * h);
    sobel16(g, w, h, stride, dx.data(), dy.data());
    // magnitude with a zero frame of one pixel
    const int mw = w + 2;
    std::vector<int> mag((size_t)mw * (h + 2), 0);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            mag[(size_t)(y + 1) * mw + x + 1] = std::abs((int)dx[(size_t)y * w + x]) + std::abs((int)dy[(size_t)y * w + x]);
    // map: 0 = may be an edge, 1 = not an edge, 2 = edge; framed with 1
    std::vector<uint8_t> map((size_t)mw * (h + 2), 1);
    std::vector<int> stack;
    const int TG22 = 13573;   // tan(22.5 deg) * 2^15 + 0.5
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int* m0 = &mag[(size_t)(y + 1) * mw + x + 1];
            const int m = *m0;
            uint8_t& out = map[(size_t)(y + 1) * mw + x + 1];
            out = 1;
            if (m > low) {
                const int xs = dx[(size_t)y * w + x], ys = dy[(size_t)y * w + x];
                const int ax = std::abs(xs), ay = std::abs(ys) << 15;
                const int tg22x = ax * TG22;
                bool is_max;
                if (ay < tg22x) {
                    is_max = m > m0[-1] && m >= m0[1];
                } else {
                    const int tg67x = tg22x + (ax << 16);
                    if (ay > tg67x) {
                        is_max = m > m0[-mw] && m >= m0[mw];
                    } else {
                        const int s = (xs ^ ys) < 0 ? 1 : -1;
                        is_max = m > m0[-mw - s] && m > m0[mw + s];
                    }
                }
                if (is_max) {
                    if (m > high) { out = 2; stack.push_back((y + 1) * mw + x + 1); }
                    else out = 0;
                }
            }
        }
    // hysteresis: 8-connected growth from the strong edges through the candidates
    while (!stack.empty()) {
        const int p = stack.back();
        stack.pop_back();
        const int nb[8] = {-mw - 1, -mw, -mw + 1, -1, 1, mw - 1, mw, mw + 1};
        for (int k = 0; k < 8; k++)
            if (map[p + nb[k]] == 0) { map[p + nb[k]] = 2; stack.push_back(p + nb[k]); }
    }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) edges[(size_t)y * w + x] = map[(size_t)(y + 1) * mw + x + 1] == 2 ? 255 : 0;
}

// cv::HoughLines(edges, lines, rho, theta, threshold): standard transform
int hough_lines(const uint8_t* edges, int w, int h, size_t stride, float rho, float theta, int threshold,
                std::vector<float>& lines /* rho,theta pairs */) {
    lines.clear();
    const double min_theta = 0, max_theta = CV_PI_D;
    int numangle = cv_floor((max_theta - min_theta) / theta) + 1;
    if (numangle > 1 && std::fabs(CV_PI_D - (numangle - 1) * (double)theta) < (double)theta / 2) --numangle;
    const int max_rho = w + h, min_rho = -max_rho;
    const int numrho = cv_round(((max_rho - min_rho) + 1) / rho);
    const float irho = 1 / rho;
    std::vector<float> tabSin(numangle), tabCos(numangle);
    float ang = (float)min_theta;
    for (int n = 0; n < numangle; ang += theta, n++) {
        tabSin[n] = (float)(std::sin((double)ang) * irho);
        tabCos[n] = (float)(std::cos((double)ang) * irho);
    }
    std::vector<int> accum((size_t)(numangle + 2) * (numrho + 2), 0);
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++)
            if (edges[(size_t)i * stride + j] != 0)
                for (int n = 0; n < numangle; n++) {
                    int r = cv_round(j * tabCos[n] + i * tabSin[n]);
                    r += (numrho - 1) / 2;
                    accum[(size_t)(n + 1) * (numrho + 2) + r + 1]++;
                }
    std::vector<int> sort_buf;
    for (int r = 0; r < numrho; r++)
        for (int n = 0; n < numangle; n++) {
            const int base = (n + 1) * (numrho + 2) + r + 1;
            if (accum[base] > threshold && accum[base] > accum[base - 1] && accum[base] >= accum[base + 1] &&
                accum[base] > accum[base - numrho - 2] && accum[base] >= accum[base + numrho + 2])
                sort_buf.push_back(base);
        }
    std::sort(sort_buf.begin(), sort_buf.end(), [&](int l1, int l2) {
        return accum[l1] > accum[l2] || (accum[l1] == accum[l2] && l1 < l2);
    });
    const double scale = 1. / (numrho + 2);
    for (int idx : sort_buf) {
        const int n = cv_floor(idx * scale) - 1;
        const int r = idx - (n + 1) * (numrho + 2) - 1;
        lines.push_back((r - (numrho - 1) * 0.5f) * rho);
        lines.push_back((float)min_theta + n * theta);
    }
    return (int)(lines.size() / 2);
}

// cv::getRotationMatrix2D(center, angle_deg, 1.0) -> 2x3 double
void rotation_matrix(float cx, float cy, double angle_deg, double M[6]) {
    const double a = angle_deg * CV_PI_D / 180;
    const double alpha = std::cos(a), beta = std::sin(a);
    M[0] = alpha; M[1] = beta; M[2] = (1 - alpha) * cx - beta * cy;
    M[3] = -beta; M[4] = alpha; M[5] = beta * cx + (1 - alpha) * cy;
}

}  // namespace vso

using namespace vso;

struct vso_roll {
    vs_roll_params_c p;
    bool first = true;
    double smoothed = 0.0;
    int last_lines = 0, last_used = 0;
    double last_detected = 0.0;
};

extern "C" {

void vso_roll_params_default(vs_roll_params_c* p) {   // RollCorrection.h:16-38
    memset(p, 0, sizeof *p);
    p->struct_size = (int32_t)sizeof *p;
    p->scale_factor = 0.25;
    p->canny_threshold_low = 50.0;
    p->canny_threshold_high = 150.0;
    p->canny_aperture = 3;
    p->hough_rho = 1.0f;
    p->hough_theta = (float)(CV_PI_D / 180.0f);
    p->hough_threshold = 100;
    p->angle_filter_min = -10.0;
    p->angle_filter_max = 10.0;
    p->angle_smoothing_alpha = 0.1;
    p->angle_decay = 0.995;
    p->max_angle_change_deg = 0.5;
}

void vso_sobel16(const uint8_t* g, int w, int h, size_t stride, int16_t* dx, int16_t* dy) { sobel16(g, w, h, stride, dx, dy); }
void vso_canny(const uint8_t* g, int w, int h, size_t stride, double low, double high, uint8_t* edges) {
    canny(g, w, h, stride, low, high, edges);
}
int vso_hough_lines(const uint8_t* edges, int w, int h, size_t stride, float rho, float theta, int threshold,
                    float* out, int max_lines) {
    std::vector<float> lines;
    int n = hough_lines(edges, w, h, stride, rho, theta, threshold, lines);
    if (n > max_lines) n = max_lines;
    for (int i = 0; i < 2 * n; i++) out[i] = lines[i];
    return n;
}
void vso_warp_affine_d(const uint8_t* src, int w, int h, size_t sstride, int cn, uint8_t* dst, size_t dstride,
                       const double* M, int border) {
    warp_affine_d(src, w, h, sstride, cn, dst, w, h, dstride, M, border, 1);
}

vso_roll* vso_roll_create(const vs_roll_params_c* p) {
    vso_roll* r = new vso_roll();
    r->p = *p;
    return r;
}
void vso_roll_destroy(vso_roll* r) { delete r; }
void vso_roll_get(const vso_roll* r, double* smoothed, double* detected, int* n_lines, int* n_used) {
    if (smoothed) *smoothed = r->smoothed;
    if (detected) *detected = r->last_detected;
    if (n_lines) *n_lines = r->last_lines;
    if (n_used) *n_used = r->last_used;
}

// RollCorrection.cpp:16-155
int vso_roll_correct(vso_roll* r, const uint8_t* data, int w, int h, size_t stride, uint8_t* out, size_t out_stride) {
    if (!data || w <= 0 || h <= 0) return 0;                                    // :21-23
    const vs_roll_params_c& p = r->p;
    if (r->first) { r->first = false; r->smoothed = 0.0; }                         // :24-27
    const int sw = (int)(w * p.scale_factor), sh = (int)(h * p.scale_factor);   // :35-38
    std::vector<uint8_t> small_bgr;
    const uint8_t* sb = data;
    int aw = w, ah = h;
    size_t sstride = stride;
    if (sw > 0 && sh > 0) {                                                       // :40-45
        small_bgr.resize((size_t)sw * sh * 3);
        resize_linear_u8(data, w, h, stride, 3, small_bgr.data(), sw, sh, (size_t)sw * 3);
        sb = small_bgr.data(); aw = sw; ah = sh; sstride = (size_t)sw * 3;
    }
    std::vector<uint8_t> gray((size_t)aw * ah), edges((size_t)aw * ah);
    bgr2gray(sb, aw, ah, sstride, gray.data(), aw);                              // :51
    canny(gray.data(), aw, ah, aw, p.canny_threshold_low, p.canny_threshold_high, edges.data());   // :54-61
    std::vector<float> lines;
    const int n = hough_lines(edges.data(), aw, ah, aw, p.hough_rho, p.hough_theta, p.hough_threshold, lines);   // :66-73
    r->last_lines = n; r->last_used = 0; r->last_detected = 0.0;
    if (n == 0) {
        r->smoothed *= p.angle_decay;                                             // :76-77
    } else {
        double sum = 0.0;
        int count = 0;
        for (int i = 0; i < n; i++) {                                             // :109-119
            const float theta = lines[2 * i + 1];
            const double deg = (theta * 180.0 / CV_PI_D) - 90.0;
            if (deg >= p.angle_filter_min && deg <= p.angle_filter_max) { sum += deg; ++count; }
        }
        r->last_used = count;
        if (count == 0) {
            r->smoothed *= p.angle_decay;                                         // :122-123
        } else {
            const double detected = sum / count;                                  // :125-135
            r->last_detected = detected;
            double na = p.angle_smoothing_alpha * detected + (1.0 - p.angle_smoothing_alpha) * r->smoothed;
            double diff = na - r->smoothed;
            if (std::fabs(diff) > p.max_angle_change_deg && p.max_angle_change_deg > 0.0) {
                diff = (diff > 0) ? p.max_angle_change_deg : -p.max_angle_change_deg;
                na = r->smoothed + diff;
            }
            r->smoothed = na;
        }
    }
    double M[6];
    rotation_matrix(w / 2.0f, h / 2.0f, r->smoothed, M);                          // :141-144
    warp_affine_d(data, w, h, stride, 3, out, w, h, out_stride, M, VS_BORDER_REPLICATE, g_threads);   // :146-149
    return 1;
}

// autoCorrectRoll on an NV12 surface.  The reference has no NV12 path: DEFINED as the BGR operator's geometry applied per plane -
// the line search of :35-119 on the luma plane (resize, then Canny and HoughLines on it: a gray picture needs no cvtColor), the
// angle recurrence unchanged (:76-77, :106-135), the rotation about the picture centre (:141-149, BORDER_REPLICATE) applied to
// the luma plane and, with the translation halved, to the half-size interleaved chroma plane.
int vso_roll_correct_nv12(vso_roll* r, const uint8_t* data, int w, int h, size_t stride, size_t uv_offset, uint8_t* out, size_t out_stride,
                          size_t out_uv_offset) {
    if (!data || w <= 0 || h <= 0) return 0;
    const vs_roll_params_c& p = r->p;
    if (r->first) { r->first = false; r->smoothed = 0.0; }
    const int sw = (int)(w * p.scale_factor), sh = (int)(h * p.scale_factor);
    std::vector<uint8_t> small;
    const uint8_t* g = data;
    int aw = w, ah = h;
    size_t gs = stride;
    if (sw > 0 && sh > 0) {
        small.resize((size_t)sw * sh);
        resize_linear_u8(data, w, h, stride, 1, small.data(), sw, sh, (size_t)sw);
        g = small.data(); aw = sw; ah = sh; gs = (size_t)sw;
    }
    std::vector<uint8_t> edges((size_t)aw * ah);
    canny(g, aw, ah, gs, p.canny_threshold_low, p.canny_threshold_high, edges.data());
    std::vector<float> lines;
    const int n = hough_lines(edges.data(), aw, ah, aw, p.hough_rho, p.hough_theta, p.hough_threshold, lines);
    r->last_lines = n; r->last_used = 0; r->last_detected = 0.0;
    double sum = 0.0;
    int count = 0;
    for (int i = 0; i < n; i++) {
        const float theta = lines[2 * i + 1];
        const double deg = (theta * 180.0 / CV_PI_D) - 90.0;
        if (deg >= p.angle_filter_min && deg <= p.angle_filter_max) { sum += deg; ++count; }
    }
    r->last_used = count;
    if (n == 0 || count == 0) {
        r->smoothed *= p.angle_decay;
    } else {
        const double detected = sum / count;
        r->last_detected = detected;
        double na = p.angle_smoothing_alpha * detected + (1.0 - p.angle_smoothing_alpha) * r->smoothed;
        double diff = na - r->smoothed;
        if (std::fabs(diff) > p.max_angle_change_deg && p.max_angle_change_deg > 0.0) {
            diff = (diff > 0) ? p.max_angle_change_deg : -p.max_angle_change_deg;
            na = r->smoothed + diff;
        }
        r->smoothed = na;
    }
    double M[6];
    rotation_matrix(w / 2.0f, h / 2.0f, r->smoothed, M);
    const double Mc[6] = {M[0], M[1], M[2] * 0.5, M[3], M[4], M[5] * 0.5};
    warp_affine_d(data, w, h, stride, 1, out, w, h, out_stride, M, VS_BORDER_REPLICATE, g_threads);
    warp_affine_d(data + uv_offset, w / 2, h / 2, stride, 2, out + out_uv_offset, w / 2, h / 2, out_stride, Mc, VS_BORDER_REPLICATE, g_threads);
    return 1;
}

}  // extern "C"
