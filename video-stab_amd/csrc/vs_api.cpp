// C ABI of libvideo-stab (include/vs_stab.h): library-level entry points,
// device memory helpers and the stage operators.  The per-stream pipeline
// (vs_stab_*) lives in stabilizer.cpp.
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

#include "vs_common.h"

namespace vsd {

static thread_local std::string g_last_error;
void set_last_error(const std::string& msg) { g_last_error = msg; }
const char* get_last_error() { return g_last_error.c_str(); }

int ensure_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_last_error(std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "count is 0") +
                       " (libvideo-stab has no CPU fallback)");
        return VS_ERR_NO_DEVICE;
    }
    return VS_OK;
}

int launch_warp_nv12_hostM(const uint8_t* d_src, size_t sstride, size_t sframe, uint8_t* d_dst, size_t dstride, size_t dframe, int w, int h,
                           const float* h_M, int batch, hipStream_t st);
int launch_warp_affine_hostM(const uint8_t* d_src, size_t sstride, size_t sframe, int sw, int sh,
                             uint8_t* d_dst, size_t dstride, size_t dframe, int dw, int dh, int cn,
                             const float* h_M, int batch, hipStream_t st);
int run_copy_rate(size_t bytes, int iters, double* gbps);
int run_libm_checksum(int fn, uint64_t start, uint64_t count, uint64_t* result);
int run_estimate_affine_partial2d(const float* d_from, const float* d_to, int n, double thr,
                                  int max_iters, double* d_model, uint8_t* d_inliers,
                                  int32_t* d_info, hipStream_t st);
int run_pyr_lk_op(const uint8_t* d_prev, const uint8_t* d_next, size_t stride, int w, int h,
                  const float* d_prev_pts, int n, float* d_next_pts, uint8_t* d_status,
                  float* d_err, int win, int max_level, int max_iters, double eps, hipStream_t st);
int run_gftt_op(const uint8_t* d_gray, size_t stride, int w, int h, int max_corners, double quality,
                double min_distance, int block_size, float* d_pts, int32_t* d_count, float* d_eig,
                hipStream_t st);

}  // namespace vsd

using namespace vsd;

extern "C" {

int vs_abi_version(void) { return VS_STAB_ABI_VERSION; }
#ifndef VS_BUILD_TAG
#define VS_BUILD_TAG "untagged"
#endif
const char* vs_build_tag(void) { return VS_BUILD_TAG; }

int vs_host_alloc(void** p, size_t bytes) {
    if (!p) return VS_ERR_INVALID_ARG;
    *p = nullptr;
    VS_TRY(vsd::ensure_device());
    VS_HIP_TRY(hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocDefault));
    return VS_OK;
}
int vs_host_register(void* p, size_t bytes) {
    if (!p || !bytes) return VS_ERR_INVALID_ARG;
    VS_TRY(ensure_device());
    const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); set_last_error(std::string("vs_host_register: ") + hipGetErrorString(e)); return VS_ERR_HIP; }
    return VS_OK;
}
int vs_host_unregister(void* p) {
    if (!p) return VS_OK;
    const hipError_t e = hipHostUnregister(p);
    if (e != hipSuccess) { (void)hipGetLastError(); set_last_error(std::string("vs_host_unregister: ") + hipGetErrorString(e)); return VS_ERR_HIP; }
    return VS_OK;
}
void vs_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

const char* vs_build_info(void) {
    return "libvideo-stab gfx950 (CDNA4, wave64) hipcc -ffp-contract=off; warp=classic-fixed-point";
}

int vs_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// Defaults of vs::Stabilizer::Parameters (Stabilizer.h:76-175) and of the
// constants hard-coded at Stabilizer.cpp:611-619, :647-649.
void vs_params_default(vs_params_c* p) {
    if (!p) return;
    memset(p, 0, sizeof *p);
    p->struct_size = (int32_t)sizeof *p;
    p->smoothing_radius = 30;
    p->max_corners = 200;
    p->quality_level = 0.01;
    p->min_distance = 30.0;
    p->block_size = 3;
    p->border_type = VS_BORDER_BLACK;
    p->smoothing_method = VS_SMOOTH_BOX;
    p->gaussian_sigma = 2.0;
    p->min_smoothing_radius = 5;
    p->max_smoothing_radius = 50;
    p->fade_alpha = 0.1f;
    p->fade_duration = 30;
    p->canvas_scale_factor = 1.5f;
    p->temporal_buffer_size = 30;
    p->canvas_blend_weight = 0.7f;
    p->adaptive_canvas_size = 1;
    p->max_canvas_scale = 2.0f;
    p->min_canvas_scale = 1.2f;
    p->edge_blend_radius = 20;
    p->hf_shake_px = 1.5f;
    p->hf_analysis_max_width = 960;
    p->hf_rot_lp_alpha = 0.2f;
    p->enable_conditional_clahe = 1;
    p->hf_dead_zone_threshold = 2.0f;
    p->hf_freeze_duration = 10;
    p->hf_motion_accumulator_decay = 0.9f;
    p->lk_win_size = 15;
    p->lk_max_level = 2;
    p->lk_max_iters = 20;
    p->lk_epsilon = 0.03;
    p->ransac_max_iters = 500;
    p->ransac_threshold = 5.0;
}

const char* vs_status_string(int status) {
    switch (status) {
        case VS_OK: return "ok";
        case VS_ERR_INVALID_ARG: return "invalid argument";
        case VS_ERR_NO_DEVICE: return "no usable HIP device";
        case VS_ERR_HIP: return "HIP runtime error";
        case VS_ERR_UNSUPPORTED: return "unsupported parameter combination";
        case VS_ERR_SIZE_CHANGED: return "frame size changed";
        case VS_ERR_CAPACITY: return "device capacity exceeded";
        default: return "unknown status";
    }
}

const char* vs_last_error(void) { return get_last_error(); }

// ---- device memory helpers ---------------------------------------------------
int vs_dev_set_device(int device) {
    VS_TRY(ensure_device());
    VS_HIP_TRY(hipSetDevice(device));
    return VS_OK;
}
int vs_dev_malloc(void** d_ptr, size_t bytes) {
    if (!d_ptr) return VS_ERR_INVALID_ARG;
    VS_TRY(ensure_device());
    VS_HIP_TRY(hipMalloc(d_ptr, bytes));
    return VS_OK;
}
int vs_dev_free(void* d_ptr) {
    if (!d_ptr) return VS_OK;
    VS_HIP_TRY(hipFree(d_ptr));
    return VS_OK;
}
int vs_dev_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes) {
    VS_TRY(ensure_device());
    VS_HIP_TRY(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    return VS_OK;
}
int vs_dev_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes) {
    VS_TRY(ensure_device());
    VS_HIP_TRY(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return VS_OK;
}
int vs_dev_memset(void* d_dst, int value, size_t bytes) {
    VS_TRY(ensure_device());
    VS_HIP_TRY(hipMemset(d_dst, value, bytes));
    return VS_OK;
}
int vs_dev_sync(void) {
    VS_TRY(ensure_device());
    VS_HIP_TRY(hipDeviceSynchronize());
    return VS_OK;
}
int vs_dev_memcpy_d2d(void* d_dst, const void* d_src, size_t bytes) {
    VS_TRY(ensure_device());
    VS_HIP_TRY(hipMemcpy(d_dst, d_src, bytes, hipMemcpyDeviceToDevice));
    return VS_OK;
}
int vs_op_libm_checksum(int fn, uint64_t start, uint64_t count, uint64_t* result) {
    if (fn < 0 || fn > 3 || !result || count == 0) return VS_ERR_INVALID_ARG;
    VS_TRY(ensure_device());
    return vsd::run_libm_checksum(fn, start, count, result);
}
int vs_dev_copy_rate(size_t bytes, int iters, double* gbytes_per_s) {
    if (!gbytes_per_s || bytes < 4096 || iters < 1 || iters > 1000) return VS_ERR_INVALID_ARG;
    VS_TRY(ensure_device());
    return vsd::run_copy_rate(bytes, iters, gbytes_per_s);
}

// ---- stage operators -----------------------------------------------------------
int vs_op_warp_affine(const void* d_src, size_t src_stride, size_t src_frame_bytes, void* d_dst,
                      size_t dst_stride, size_t dst_frame_bytes, int w, int h, int cn,
                      const float* M, int batch, void* stream) {
    VS_TRY(ensure_device());
    return launch_warp_affine_hostM((const uint8_t*)d_src, src_stride, src_frame_bytes, w, h,
                                    (uint8_t*)d_dst, dst_stride, dst_frame_bytes, w, h, cn, M, batch,
                                    (hipStream_t)stream);
}

int vs_op_warp_affine_nv12(const void* d_src, size_t src_stride, void* d_dst, size_t dst_stride,
                           int w, int h, const float* M, int batch, size_t src_frame_bytes,
                           size_t dst_frame_bytes, void* stream) {
    VS_TRY(ensure_device());
    // luma: the full matrix; chroma: the half-size two-channel plane, same rotation, translation halved
    return launch_warp_nv12_hostM((const uint8_t*)d_src, src_stride, src_frame_bytes, (uint8_t*)d_dst, dst_stride, dst_frame_bytes, w, h, M, batch,
                                  (hipStream_t)stream);
}

int vs_op_resize_gray(const void* d_src, size_t src_stride, int sw, int sh, int fmt, void* d_dst,
                      size_t dst_stride, int dw, int dh, void* stream) {
    VS_TRY(ensure_device());
    return launch_resize_gray((const uint8_t*)d_src, src_stride, sw, sh, fmt, (uint8_t*)d_dst,
                              dst_stride, dw, dh, (hipStream_t)stream);
}

int vs_op_pyr_down(const void* d_src, size_t src_stride, int sw, int sh, void* d_dst,
                   size_t dst_stride, void* stream) {
    VS_TRY(ensure_device());
    return launch_pyr_down((const uint8_t*)d_src, src_stride, sw, sh, (uint8_t*)d_dst, dst_stride,
                           (hipStream_t)stream);
}

int vs_op_scharr(const void* d_src, size_t src_stride, int w, int h, void* d_dst, void* stream) {
    VS_TRY(ensure_device());
    return launch_scharr((const uint8_t*)d_src, src_stride, w, h, (int16_t*)d_dst, (hipStream_t)stream);
}

int vs_op_pyr_level(const void* d_src, size_t src_stride, size_t src_frame_bytes, int w, int h, void* d_der, void* d_next,
                    size_t next_stride, size_t next_frame_bytes, int items, void* stream) {
    VS_TRY(ensure_device());
    if (!d_src || !d_der || items < 1 || items > 4096 || w <= 0 || h <= 0) { set_last_error("pyr_level: invalid argument"); return VS_ERR_INVALID_ARG; }
    // the kernel takes device tables of (source, destination) pairs: built here for the call (an operator for tests and probes)
    std::vector<ImgPair> t(2 * (size_t)items);
    for (int i = 0; i < items; i++) {
        const uint8_t* s = (const uint8_t*)d_src + (size_t)i * src_frame_bytes;
        t[i] = ImgPair{s, (uint8_t*)d_der + (size_t)i * w * h * 4};
        t[items + i] = ImgPair{s, d_next ? (uint8_t*)d_next + (size_t)i * next_frame_bytes : nullptr};
    }
    ImgPair* d_t = nullptr;
    VS_HIP_TRY(hipMalloc((void**)&d_t, sizeof(ImgPair) * t.size()));
    hipStream_t st = (hipStream_t)stream;
    int rc = VS_OK;
    if (hipMemcpyAsync(d_t, t.data(), sizeof(ImgPair) * t.size(), hipMemcpyHostToDevice, st) != hipSuccess) rc = VS_ERR_HIP;
    if (rc == VS_OK) rc = launch_pyr_level_batch(d_t, d_next ? d_t + items : nullptr, items, src_stride, w, h, next_stride, st);
    (void)hipStreamSynchronize(st);
    (void)hipFree(d_t);
    return rc;
}

int vs_op_pyr_lk(const void* d_prev, const void* d_next, size_t stride, int w, int h,
                 const float* d_prev_pts, int n, float* d_next_pts, uint8_t* d_status, float* d_err,
                 int win, int max_level, int max_iters, double eps, void* stream) {
    VS_TRY(ensure_device());
    return run_pyr_lk_op((const uint8_t*)d_prev, (const uint8_t*)d_next, stride, w, h, d_prev_pts, n,
                         d_next_pts, d_status, d_err, win, max_level, max_iters, eps,
                         (hipStream_t)stream);
}

int vs_op_gftt(const void* d_gray, size_t stride, int w, int h, int max_corners, double quality,
               double min_distance, int block_size, float* d_pts, int32_t* d_count, float* d_eig,
               void* stream) {
    VS_TRY(ensure_device());
    return run_gftt_op((const uint8_t*)d_gray, stride, w, h, max_corners, quality, min_distance,
                       block_size, d_pts, d_count, d_eig, (hipStream_t)stream);
}

int vs_op_estimate_affine_partial2d(const float* d_from, const float* d_to, int n, double thr,
                                    int max_iters, double* d_model, uint8_t* d_inliers,
                                    int32_t* d_info, void* stream) {
    VS_TRY(ensure_device());
    return run_estimate_affine_partial2d(d_from, d_to, n, thr, max_iters, d_model, d_inliers, d_info,
                                         (hipStream_t)stream);
}

}  // extern "C"
