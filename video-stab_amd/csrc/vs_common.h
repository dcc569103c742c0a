// Internal header of libvideo-stab (gfx950).  Not part of the C ABI.
#ifndef VS_COMMON_H
#define VS_COMMON_H

#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>
#include <cfloat>
#include <cmath>
#include <string>

#include "../../include/vs_stab.h"

namespace vsd {

// ---- error plumbing ---------------------------------------------------------
void set_last_error(const std::string& msg);
const char* get_last_error();

#define VS_HIP_TRY(expr)                                                                   \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            vsd::set_last_error(std::string(#expr) + ": " + hipGetErrorString(_e));        \
            return (_e == hipErrorNoDevice || _e == hipErrorInvalidDevice ||               \
                    _e == hipErrorInsufficientDriver || _e == hipErrorNoBinaryForGpu)      \
                       ? VS_ERR_NO_DEVICE                                                  \
                       : VS_ERR_HIP;                                                       \
        }                                                                                  \
    } while (0)

#define VS_TRY(expr)                      \
    do {                                  \
        int _s = (expr);                  \
        if (_s != VS_OK) return _s;       \
    } while (0)

int ensure_device();  // VS_OK when a gfx950-class device is usable

// ---- device helpers: same integer semantics as OpenCV's cvRound/cvFloor -------
#ifdef __HIPCC__
__device__ __forceinline__ int d_round(double v) { return __double2int_rn(v); }
__device__ __forceinline__ int f_round(float v) { return __float2int_rn(v); }
__device__ __forceinline__ int f_floor(float v) { return __float2int_rd(v); }
__device__ __forceinline__ int sat_s16(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }
__device__ __forceinline__ int reflect101(int p, int len) {
    // cv::borderInterpolate(BORDER_REFLECT_101)
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}
#endif

// cv::warpAffine's inversion of the forward 2x3 matrix, in double (imgwarp.cpp).  The same
// IEEE operation sequence on host and device (no FMA contraction in either build).
// The library has one configuration: kernel and schedule variants measured in earlier rounds live in scratch/ as patches, not
// behind switches.  What remains behind VS_LAB=1 are two test hooks (VS_STAB_DEBUG_DELAY_US: a spin kernel that widens a window
// an ordering test looks into; VS_STAB_HOST_HELPER=0: the pipelined host call without its helper thread).  (The documented
// runtime settings - VS_STAB_DEVICE, VS_STAB_HOST_PIPELINE, VS_STAB_HELPER_SPIN_US - are ordinary environment variables.)
inline const char* lab_env(const char* name) {
    const char* e = std::getenv("VS_LAB");
    return (e && e[0] == '1') ? std::getenv(name) : nullptr;
}

#ifdef __HIPCC__
__host__ __device__
#endif
inline void warp_invert(const float* Mf, double* inv) {
    double M[6];
    for (int i = 0; i < 6; i++) M[i] = (double)Mf[i];
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11; M[1] *= -D;
    M[3] *= -D; M[4] = A22;
    double b1 = -M[0] * M[2] - M[1] * M[5];
    double b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1; M[5] = b2;
    for (int i = 0; i < 6; i++) inv[i] = M[i];
}

// ---- stage launchers (device pointers, asynchronous on `st`) -----------------
// d_Minv: batch*6 doubles on the DEVICE: the INVERSE maps (warp_invert of the forward matrices).
// d_tabs: nullptr, or warp_tabs_ints(dw, dh, frames of the launch) ints of device scratch: the coordinate terms are
// then built once per frame by a small launch in front of the warp instead of once per tile inside it.  The scratch
// is read by the warp launch only (stream order), so one buffer per stream and caller is enough.
size_t warp_tabs_ints(int dw, int dh, int frames);
int launch_warp_affine(const uint8_t* d_src, size_t sstride, size_t sframe, int sw, int sh,
                       uint8_t* d_dst, size_t dstride, size_t dframe, int dw, int dh, int cn,
                       const double* d_Minv, int batch, int32_t* d_tabs, hipStream_t st);
// what: the whole launch, or its two halves apart (tables of the frames now, the warp later from the same d_tabs).
enum { VS_WARP_ALL = 0, VS_WARP_TABLES_ONLY = 1, VS_WARP_ONLY = 2 };
// One warp of a multi-job launch (launch_warp_jobs, k_warp.hip): any geometry, inverse map in double on the host.
struct WarpJob {
    const uint8_t* src;
    uint8_t* dst;
    uint32_t sstride, dstride;
    int32_t sw, sh, dw, dh, cn, border;
    double m[6];
};
constexpr int WARP_JOBS_MAX = 16;
int launch_warp_jobs(const WarpJob* jobs, int n, hipStream_t st);
// tab_stride: ints between the tables of consecutive frames in d_tabs (0: packed, warp_tabs_ints(dw, dh, 1))
int launch_warp_affine_list(const uint8_t* const* srcs, uint8_t* const* dsts, int n, size_t sstride, int sw, int sh,
                            size_t dstride, int dw, int dh, int cn, const double* d_Minv, int minv_stride, int32_t* d_tabs,
                            hipStream_t st, int what = VS_WARP_ALL, int tab_stride = 0);
// NV12: both planes of n surfaces in one launch, from table blocks (warp_tab.h) that have been built
int launch_warp_nv12_list(const uint8_t* const* ys, uint8_t* const* yd, int n, size_t sstride, size_t dstride, int w, int h, size_t src_uv,
                          size_t dst_uv, const int32_t* d_tabs, hipStream_t st, int border = VS_BORDER_BLACK);
int launch_resize_gray(const uint8_t* d_src, size_t sstride, int sw, int sh, int fmt,
                       uint8_t* d_dst, size_t dstride, int dw, int dh, hipStream_t st);
// Batched forms (batch mode): the images of `items` frames in one launch; d_pairs = device table of
// (source, destination) pointers, all of one geometry.
struct ImgPair { const void* src; void* dst; };
int launch_resize_gray_batch(const ImgPair* d_pairs, int items, size_t sstride, int sw, int sh, int fmt, size_t dstride,
                             int dw, int dh, int aligned, hipStream_t st);
int launch_pyr_level_batch(const ImgPair* d_scharr_pairs, const ImgPair* d_pyr_pairs, int items, size_t sstride, int w, int h,
                           size_t dstride, hipStream_t st);
int launch_pyr_down(const uint8_t* d_src, size_t sstride, int sw, int sh, uint8_t* d_dst,
                    size_t dstride, hipStream_t st);
int launch_scharr(const uint8_t* d_src, size_t sstride, int w, int h, int16_t* d_dst,
                  hipStream_t st);

struct LKLevel {
    const uint8_t* prev;
    const uint8_t* next;
    const int16_t* deriv;  // prev derivatives, w*h*2
    int w, h;
    size_t stride;
};
// Tracks n points over levels[max_level..0]; n may be a device count (d_n) or host n.
int launch_pyr_lk(const LKLevel* levels, int max_level, const float* d_prev_pts, int n,
                  const int32_t* d_n, float* d_next_pts, uint8_t* d_status, float* d_err, int win,
                  int max_iters, double eps, hipStream_t st);

struct GfttWork {            // device scratch owned by the caller
    float* eig;              // w*h
    uint64_t* cand;          // cap candidates (key = value bits << 32 | index)
    int32_t* counters;       // [0]=ncand [1]=maxbits [2]=overflow [3]=n_out  (+ spare)
    int cap;
};
size_t gftt_work_bytes(int w, int h, int cap);
void gftt_work_carve(void* base, int w, int h, int cap, GfttWork* out);
int launch_gftt(const uint8_t* d_gray, size_t stride, int w, int h, int max_corners,
                double quality, double min_distance, int block_size, const GfttWork& wk,
                float* d_pts, int32_t* d_count, hipStream_t st);

}  // namespace vsd
#endif
