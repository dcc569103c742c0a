// CPU ORACLE (test infrastructure) - shared helpers.
#ifndef VSO_INTERNAL_H
#define VSO_INTERNAL_H

#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdint>
#include <thread>
#include <vector>

#include "vso.h"
// glibc's cosf / sinf / atan2f as a frozen restatement (the header the product evaluates on the device as well): see libm_*
// below.  It is a statement of a third party's published algorithm, checked against the host's own libm on every float by
// tests/test_libm.py; nothing else of the product is visible to the oracle.
#include "../video-stab_amd/csrc/vs_libm.h"

namespace vso {

// std::cos(float) / std::sin(float) / std::atan2(float, float) of the reference (/root/reference/src/Stabilizer.cpp:662,
// 902-908, 1689) are the HOST libm's cosf / sinf / atan2f.  Their last place depends on the libm build (glibc >= 2.41 ships
// correctly rounded atan2f; x86 hosts without FMA take another sincosf variant), and one ulp in a matrix entry moves a pixel
// column.  The oracle's pipeline therefore evaluates the one definition the tests verified against a real glibc (2.35, x86-64
// with FMA: 0 mismatches over every float) wherever the suite runs; tests/test_libm.py remains the check against the host's
// own libm and says so when the host is outside the verified set.
static inline float libm_cosf(float x) { return vslibm::cosf_ref(x); }
static inline float libm_sinf(float x) { return vslibm::sinf_ref(x); }
static inline float libm_atan2f(float y, float x) { return vslibm::atan2f_ref(y, x); }

// cvRound: round-half-to-even (SSE cvtsd2si / lrint in the default FP mode)
static inline int cv_round(double v) { return (int)lrint(v); }
static inline int cv_round(float v) { return (int)lrintf(v); }
static inline int cv_floor(float v) {
    int i = (int)v;
    return i - (i > v);
}
static inline int cv_floor(double v) {
    int i = (int)v;
    return i - (i > v);
}
static inline short sat_short(int v) {
    return (short)(v < SHRT_MIN ? SHRT_MIN : v > SHRT_MAX ? SHRT_MAX : v);
}
static inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
static inline int sat_int(double v) { return cv_round(v); }

int border_interpolate(int p, int len, int border);
void resize_linear_u8(const uint8_t* src, int sw, int sh, size_t sstride, int cn, uint8_t* dst,
                      int dw, int dh, size_t dstride);
void bgr2gray(const uint8_t* src, int w, int h, size_t sstride, uint8_t* dst, size_t dstride);
void pyr_down(const uint8_t* src, int sw, int sh, size_t sstride, uint8_t* dst, size_t dstride);
void scharr(const uint8_t* src, int w, int h, size_t sstride, int16_t* dst);
void copy_make_border(const uint8_t* src, int w, int h, size_t sstride, int cn, uint8_t* dst,
                      size_t dstride, int b, int border);
void warp_affine(const uint8_t* src, int w, int h, size_t sstride, int cn, uint8_t* dst,
                 size_t dstride, const float* M, int nthreads);
void warp_affine_d(const uint8_t* src, int sw, int sh, size_t sstride, int cn, uint8_t* dst, int dw, int dh,
                   size_t dstride, const double* M, int border, int nthreads);
void sobel16(const uint8_t* g, int w, int h, size_t stride, int16_t* dx, int16_t* dy);
void canny(const uint8_t* g, int w, int h, size_t stride, double low, double high, uint8_t* edges);
int hough_lines(const uint8_t* edges, int w, int h, size_t stride, float rho, float theta, int threshold,
                std::vector<float>& lines);
void rotation_matrix(float cx, float cy, double angle_deg, double M[6]);
constexpr double CV_PI_D = 3.1415926535897932384626433832795;

struct Gray {
    int w = 0, h = 0;
    std::vector<uint8_t> d;
    bool empty() const { return d.empty(); }
    void create(int w_, int h_) { w = w_; h = h_; d.assign((size_t)w * h, 0); }
};

int gftt(const uint8_t* gray, int w, int h, size_t stride, int max_corners, double quality,
         double min_distance, int block_size, std::vector<float>& pts, int* n_candidates);
int pyr_lk(const uint8_t* prev, const uint8_t* next, int w, int h, size_t stride,
           const float* prev_pts, int n, float* next_pts, uint8_t* status, float* err, int win,
           int max_level, int max_iters, double eps, int nthreads);
int estimate_affine_partial2d(const float* from, const float* to, int n, double thr,
                              int max_iters, double* model, uint8_t* inliers, int32_t* info);

std::vector<float> box_filter(const std::vector<float>& path, int radius_param, bool drone);
std::vector<float> gaussian_filter(const std::vector<float>& path, float sigma);
std::vector<float> kalman_filter(const std::vector<float>& path);
int adaptive_radius(const std::vector<float>& px, const std::vector<float>& py,
                    const std::vector<float>& pa, int smoothing_radius);
int motion_intent(const std::vector<float>& tr /* n*3 */, const float motion[3], int frame_index);
float adaptive_strength(int intent, const float motion[3]);

struct Pt { int x, y; };
// cv::findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE), in the order of the vector OpenCV returns
void find_contours_external(const uint8_t* mask, int w, int h, size_t stride, std::vector<std::vector<Pt>>& out);

// Virtual canvas (Stabilizer.cpp:1130-1134, 2066-2443): vso_canvas.cpp
struct CanvasState;
CanvasState* canvas_new();
void canvas_delete(CanvasState* c);
// out <- the frame the reference returns for `frame` (w x h BGR) whose correction is t = (dx, dy, da);
// transforms = transforms_ (n * 3) at the time of the call
void canvas_apply(CanvasState* c, const vs_params_c& p, const uint8_t* frame, int w, int h, size_t stride, const float t[3],
                  const std::vector<float>& transforms, uint8_t* out, size_t out_stride);
void canvas_info(const CanvasState* c, int32_t info[8]);

extern int g_threads;

// fn(y0, y1) over [0, n) in g_threads contiguous pieces (the pieces are independent: the result does not depend on
// the number of threads)
template <class F>
static inline void parallel_rows(int n, F fn) {
    const int nt = g_threads < 1 ? 1 : (g_threads > n ? (n > 0 ? n : 1) : g_threads);
    if (nt <= 1) { fn(0, n); return; }
    std::vector<std::thread> th;
    const int per = (n + nt - 1) / nt;
    for (int t = 0; t < nt; t++) {
        const int y0 = t * per, y1 = y0 + per < n ? y0 + per : n;
        if (y0 >= y1) break;
        th.emplace_back([=] { fn(y0, y1); });
    }
    for (auto& t : th) t.join();
}

}  // namespace vso
#endif
