/*
 * vso.h - CPU ORACLE for the stabilize() hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This directory is a plain C++ restatement of the algorithm that
 * /root/reference/src/Stabilizer.cpp (CPU branch, useCuda=false) executes per
 * frame, including the OpenCV 4.11 primitives it calls.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product library (video-stab_amd/csrc) never includes or links anything here.
 *
 * PARITY UNPINNED.  The reference ships no tests, fixtures or golden vectors
 *  and cannot be built here (it needs OpenCV 4.11 +
 * CUDA + GStreamer + DeepStream; none are in the image, its binaries are
 * aarch64).  The arithmetic of the path lives in OpenCV 4.11 (un-vendored
 * third-party dependency, evidenced by `NEEDED libopencv_core.so.411` in
 * examples/vs); the OpenCV primitives are restated here from the published
 * algorithm of that version.  What pins this oracle instead: analytic
 * known-answer tests (tests/test_oracle_kat.py) and the reference's own call
 * sites/parameters, cited per function as file:line below.
 *
 * Where OpenCV's result depends on the SIMD width of the build (float
 * accumulation order in calcOpticalFlowPyrLK, FMA use in cornerMinEigenVal)
 * the oracle fixes ONE definition and says so at the function.
 */
#ifndef VSO_H
#define VSO_H

#include <stddef.h>
#include <stdint.h>
#include "../include/vs_stab.h" /* vs_params_c / vs_debug_frame PODs only */

#ifdef __cplusplus
extern "C" {
#endif

/* ---- image primitives ----------------------------------------------------- */
/* cv::resize(src,dst,Size(dw,dh),0,0,INTER_LINEAR) for CV_8UC(cn)
 * (call sites Stabilizer.cpp:304,449,602,1121) */
void vso_resize_linear_u8(const uint8_t* src, int sw, int sh, size_t sstride, int cn,
                          uint8_t* dst, int dw, int dh, size_t dstride);
/* cv::cvtColor(BGR2GRAY) 8U (Stabilizer.cpp:305,450) */
void vso_bgr2gray(const uint8_t* src, int w, int h, size_t sstride, uint8_t* dst, size_t dstride);
/* cv::pyrDown 8UC1, BORDER_REFLECT_101 (inside calcOpticalFlowPyrLK) */
void vso_pyr_down(const uint8_t* src, int sw, int sh, size_t sstride, uint8_t* dst, size_t dstride);
/* calcSharrDeriv: int16 interleaved (dI/dx, dI/dy), h*w*2 */
void vso_scharr(const uint8_t* src, int w, int h, size_t sstride, int16_t* dst);
/* cv::copyMakeBorder (Stabilizer.cpp:982-987); border = vs_border */
void vso_copy_make_border(const uint8_t* src, int w, int h, size_t sstride, int cn,
                          uint8_t* dst, size_t dstride, int b, int border);
/* cv::warpAffine(INTER_LINEAR, BORDER_CONSTANT 0), forward matrix M (float[6])
 * (Stabilizer.cpp:1056-1060) */
void vso_warp_affine(const uint8_t* src, int w, int h, size_t sstride, int cn,
                     uint8_t* dst, size_t dstride, const float* M);
/* same, multi-threaded over rows (cpu_baseline "all cores" row); nthreads<=0: hw */
void vso_warp_affine_mt(const uint8_t* src, int w, int h, size_t sstride, int cn,
                        uint8_t* dst, size_t dstride, const float* M, int nthreads);
/* NV12 policy of this build (no reference path): Y with M,
 * interleaved UV at half size with the translation halved */
void vso_warp_affine_nv12(const uint8_t* src, int w, int h, size_t sstride,
                          uint8_t* dst, size_t dstride, const float* M);

/* ---- features ------------------------------------------------------------- */
/* cv::cornerMinEigenVal(gray, eig, blockSize, 3) */
void vso_min_eigen(const uint8_t* gray, int w, int h, size_t stride, int block_size, float* eig);
/* cv::goodFeaturesToTrack (Stabilizer.cpp:355-357, 740-744); returns count;
 * n_candidates (optional) receives the number of local maxima examined list size */
int vso_gftt(const uint8_t* gray, int w, int h, size_t stride, int max_corners,
             double quality, double min_distance, int block_size, float* out_pts,
             int* n_candidates);
/* cv::calcOpticalFlowPyrLK (Stabilizer.cpp:611-619); returns levels used-1 */
int vso_pyr_lk(const uint8_t* prev, const uint8_t* next, int w, int h, size_t stride,
               const float* prev_pts, int n, float* next_pts, uint8_t* status, float* err,
               int win, int max_level, int max_iters, double eps);

/* ---- model ---------------------------------------------------------------- */
/* cv::RNG(seed).next() stream: fills out[0..n) */
void vso_rng_stream(uint64_t seed, uint32_t* out, int n);
/* cv::estimateAffinePartial2D(from,to,inliers,RANSAC,thr,max_iters,0.99,10)
 * (Stabilizer.cpp:647-649).  model[6] double row-major; info[4] =
 * {ok, best_iter, iters_run, n_inliers}.  returns ok. */
int vso_estimate_affine_partial2d(const float* from, const float* to, int n, double thr,
                                  int max_iters, double* model, uint8_t* inliers, int32_t* info);

/* ---- trajectory (Stabilizer.cpp:1139-1172,1364-1458,1637-1780) ------------- */
void vso_box_filter(const float* path, int n, int radius_param, int drone, float* out);
void vso_gaussian_filter(const float* path, int n, float sigma, float* out);
void vso_kalman_filter(const float* path, int n, float* out);
int  vso_adaptive_radius(const float* px, const float* py, const float* pa, int n, int smoothing_radius);
int  vso_motion_intent(const float* transforms /*n*3*/, int n, int frame_index);

/* ---- vs::Stabilizer restated (Stabilizer.cpp:50-1172) ---------------------- */
typedef struct vso_stab vso_stab;
vso_stab* vso_stab_create(const vs_params_c* p);
void      vso_stab_destroy(vso_stab* s);
void      vso_stab_clean(vso_stab* s);
/* stabilize(): returns 1 and fills out (out_w x out_h, tightly strided by out_stride) */
int vso_stab_push(vso_stab* s, const uint8_t* data, int w, int h, size_t stride, int fmt,
                  uint8_t* out, size_t out_stride);
int vso_stab_flush(vso_stab* s, uint8_t* out, size_t out_stride);
void vso_stab_out_size(const vso_stab* s, int w, int h, int* ow, int* oh);
void vso_stab_get_debug(const vso_stab* s, vs_debug_frame* d);
/* {canvas w, h, scale (float bits), regions, regions filled, temporal index of the last fill, window x, y} */
void vso_stab_canvas_info(const vso_stab* s, int32_t info[8]);
int  vso_stab_get_debug_arrays(const vso_stab* s, float* prev_pts, float* curr_pts,
                               uint8_t* status, uint8_t* inliers, float* detected_pts,
                               uint8_t* gray, int* aw, int* ah);
/* ---- vs::RollCorrection restated (src/RollCorrection.cpp:16-155) --------------- */
void vso_roll_params_default(vs_roll_params_c* p);
void vso_sobel16(const uint8_t* g, int w, int h, size_t stride, int16_t* dx, int16_t* dy);
void vso_canny(const uint8_t* g, int w, int h, size_t stride, double low, double high, uint8_t* edges);
int  vso_hough_lines(const uint8_t* edges, int w, int h, size_t stride, float rho, float theta, int threshold,
                     float* out, int max_lines);
void vso_warp_affine_d(const uint8_t* src, int w, int h, size_t sstride, int cn, uint8_t* dst, size_t dstride,
                       const double* M, int border);
typedef struct vso_roll vso_roll;
vso_roll* vso_roll_create(const vs_roll_params_c* p);
void vso_roll_destroy(vso_roll* r);
int  vso_roll_correct(vso_roll* r, const uint8_t* data, int w, int h, size_t stride, uint8_t* out, size_t out_stride);
/* NV12 surfaces (no reference path: the BGR operators' geometry per plane, see the definitions) */
int  vso_roll_correct_nv12(vso_roll* r, const uint8_t* data, int w, int h, size_t stride, size_t uv_offset, uint8_t* out, size_t out_stride,
                           size_t out_uv_offset);
int  vso_azc_apply_nv12(const uint8_t* src, int w, int h, size_t stride, size_t uv_offset, uint8_t* out, size_t out_stride, size_t out_uv_offset,
                        int32_t* out_w, int32_t* out_h, int32_t* info);
void vso_roll_get(const vso_roll* r, double* smoothed, double* detected, int* n_lines, int* n_used);

/* ---- vs::AutoZoomCrop restated (src/AutoZoomCrop.cpp:10-283) ------------------- */
/* gray -> threshold(>1) -> MORPH_CLOSE 5x5 ellipse  (:111-139) */
void vso_content_mask(const uint8_t* src, int w, int h, size_t stride, int cn, uint8_t* mask);
/* cv::findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) (:146-147); flattened output */
int  vso_find_contours(const uint8_t* mask, int w, int h, size_t stride, int32_t* counts, int max_contours,
                       int32_t* xy, int max_points);
/* cv::drawContours(..., FILLED) of one contour (:167-168) */
void vso_fill_contour(const int32_t* xy, int n, int w, int h, uint8_t* mask);
/* contours -> largest -> interior rectangle -> aspect fix (:141-228);
 * info = {n_contours, contour_points, x, y, w, h, iterations, valid} */
void vso_azc_crop_rect(const uint8_t* content_mask, int w, int h, int32_t* info);
int  vso_azc_apply(const uint8_t* src, int w, int h, size_t stride, int cn, uint8_t* out, int32_t* out_w,
                   int32_t* out_h, int32_t* info);

/* ---- vs::Enhancer restated (src/Enhancer.cpp:19-69,138-239), vso_enhance.cpp ---- */
void vso_enh_params_default(vs_enh_params_c* p);
/* convertTo(-1, alpha, beta) on CV_8U as a table (:37-39,:150) */
void vso_convert_scale_lut(double alpha, double beta, uint8_t* lut256);
/* whiteBalanceCPU factors from the channel sums (:22-36) */
void vso_wb_scales(const uint64_t* sums, uint64_t npix, float alpha, double* scales);
void vso_gamma_lut(float gamma, uint8_t* lut256);                       /* :171-178 */
/* cvtColor 8U, packed 3-channel pixels: BGR2HSV / HSV2BGR (:43,:56), BGR2Lab / Lab2BGR (:61,:68) */
void vso_bgr2hsv(const uint8_t* src, size_t n, uint8_t* dst);
void vso_hsv2bgr(const uint8_t* src, size_t n, uint8_t* dst);
void vso_bgr2lab(const uint8_t* src, size_t n, uint8_t* dst);
void vso_lab2bgr(const uint8_t* src, size_t n, uint8_t* dst);
void vso_vibrance(uint8_t* bgr, size_t n, float alpha);                 /* :41-57 */
/* tap count of GaussianBlur(Size(0,0), sigma) on CV_8U and its 8.8 fixed-point kernel */
int  vso_gaussian_kernel_q8(double sigma, uint16_t* k, int cap);
int  vso_gaussian_blur_u8(const uint8_t* src, int w, int h, size_t stride, int cn, double sigma, uint8_t* dst,
                          size_t dstride);                              /* :160-161 */
void vso_add_weighted_u8(const uint8_t* a, double alpha, const uint8_t* b, double beta, double gamma, uint8_t* dst,
                         size_t n);                                     /* :162 */
/* cv::CLAHE on one 8-bit plane (:64-65); lut_out (optional) tiles*tiles*256 bytes */
int  vso_clahe_u8(const uint8_t* src, int w, int h, size_t stride, double clip_limit, int tiles, uint8_t* dst,
                  size_t dstride, uint8_t* lut_out);
int  vso_clahe_bgr(uint8_t* bgr, int w, int h, float clip_limit, int tiles);   /* :59-69 */
/* cvtColor COLOR_LBGR2Lab / COLOR_Lab2LBGR (linear RGB), as fastNlMeansDenoisingColored uses them */
void vso_lbgr2lab(const uint8_t* src, size_t n, uint8_t* dst);
void vso_lab2lbgr(const uint8_t* src, size_t n, uint8_t* dst);
/* weight table of cv::fastNlMeansDenoising for 8-bit data (table may be NULL: returns the length);
 * info2 = {bin shift, fixed-point multiplier} */
int  vso_nlm_weights(float h, int cn, int template_size, int search_size, int32_t* table, int cap, int32_t* info2);
/* cv::fastNlMeansDenoising(src, dst, h, template, search) for CV_8UC1 / CV_8UC2, brute force */
int  vso_fast_nl_means(const uint8_t* src, int w, int h, size_t stride, int cn, float hp, int template_size,
                       int search_size, uint8_t* dst, size_t dstride);
/* cv::fastNlMeansDenoisingColored(img, img, h, hColor, 7, 21) (:165-169), in place on packed BGR */
int  vso_denoise_colored(uint8_t* bgr, int w, int h, float hl, float hc);
/* enhanceImage (:138-239), BGR8; 0 ok, -1 bad argument */
int  vso_enhance(const uint8_t* src, int w, int h, size_t stride, const vs_enh_params_c* p, uint8_t* out,
                 size_t out_stride);


/* threads used by row/point-parallel stages of vso_stab_push (default 1) */
void vso_set_threads(int n);
/* the host libm's cosf / sinf / atanf (fn 0, 1, 2: argument i = the float with bit pattern (uint32_t)i) or atan2f (fn 3: pair i
 * of the generator in vso_libm.cpp) over [start, start + count), as the checksum vs_op_libm_checksum forms on the device */
uint64_t vso_libm_checksum(int fn, uint64_t start, uint64_t count, int threads);
void vso_params_default(vs_params_c* p);

#ifdef __cplusplus
}
#endif
#endif
