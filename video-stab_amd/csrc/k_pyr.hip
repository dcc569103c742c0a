// Image pyramid and Scharr derivative kernels: the per-call preprocessing of
// cv::calcOpticalFlowPyrLK (/root/reference/src/Stabilizer.cpp:611-619).
//   pyr_down: 5x5 [1 4 6 4 1]^2, (v+128)>>8, BORDER_REFLECT_101, dst=((w+1)/2,(h+1)/2)
//   scharr  : dI/dx = [3 10 3]^T x [-1 0 1], dI/dy = [-1 0 1]^T x [3 10 3],
//             REFLECT_101 inside the image, int16 interleaved (dx,dy)
// Both are small streaming stencils (<= 0.5 MB in); one lane per output pixel,
// taps served by L1/L2.
#include "vs_common.h"

namespace vsd {
namespace {

constexpr int NT = 256;
constexpr int RPB = 4;      // image rows per workgroup: one row each made the launches dispatch-rate bound

__global__ __launch_bounds__(NT) void pyr_down_kernel(const uint8_t* __restrict__ src_, size_t sstride,
                                                      int sw, int sh, uint8_t* __restrict__ dst_,
                                                      size_t dstride, int dw, int dh, const ImgPair* __restrict__ table) {
    // batched launch: blockIdx.z selects the frame's (source, destination) pair
    const uint8_t* __restrict__ src = table ? static_cast<const uint8_t*>(table[blockIdx.z].src) : src_;
    uint8_t* __restrict__ dst = table ? static_cast<uint8_t*>(table[blockIdx.z].dst) : dst_;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= dw) return;
    int xi[5];
#pragma unroll
    for (int k = 0; k < 5; k++) xi[k] = reflect101(2 * x + k - 2, sw);
    for (int y = blockIdx.y * RPB; y < min((int)(blockIdx.y + 1) * RPB, dh); y++) {
        int col[5];
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const uint8_t* s = src + (size_t)reflect101(2 * y + j - 2, sh) * sstride;
            col[j] = s[xi[2]] * 6 + (s[xi[1]] + s[xi[3]]) * 4 + s[xi[0]] + s[xi[4]];
        }
        dst[(size_t)y * dstride + x] = (uint8_t)((col[2] * 6 + (col[1] + col[3]) * 4 + col[0] + col[4] + 128) >> 8);
    }
}

__global__ __launch_bounds__(NT) void scharr_kernel(const uint8_t* __restrict__ src_, size_t sstride, int w,
                                                    int h, int16_t* __restrict__ dst_, const ImgPair* __restrict__ table) {
    const uint8_t* __restrict__ src = table ? static_cast<const uint8_t*>(table[blockIdx.z].src) : src_;
    int16_t* __restrict__ dst = table ? static_cast<int16_t*>(table[blockIdx.z].dst) : dst_;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= w) return;
    for (int y = blockIdx.y * RPB; y < min((int)(blockIdx.y + 1) * RPB, h); y++) {
    const uint8_t* r0 = src + (size_t)(y > 0 ? y - 1 : h > 1 ? 1 : 0) * sstride;
    const uint8_t* r1 = src + (size_t)y * sstride;
    const uint8_t* r2 = src + (size_t)(y < h - 1 ? y + 1 : h > 1 ? h - 2 : 0) * sstride;
    // column x-1 / x+1 of the vertically filtered rows, with the row's own
    // reflect (trow[-1] = trow[1], trow[w] = trow[w-2])
    const int xm = x > 0 ? x - 1 : (w > 1 ? 1 : 0);
    const int xp = x < w - 1 ? x + 1 : (w > 1 ? w - 2 : 0);
    const int a_m = (r0[xm] + r2[xm]) * 3 + r1[xm] * 10;
    const int a_p = (r0[xp] + r2[xp]) * 3 + r1[xp] * 10;
    const int b_m = r2[xm] - r0[xm];
    const int b_c = r2[x] - r0[x];
    const int b_p = r2[xp] - r0[xp];
    short2 o;
    o.x = (short)(a_p - a_m);
    o.y = (short)((b_p + b_m) * 3 + b_c * 10);
    *reinterpret_cast<short2*>(dst + ((size_t)y * w + x) * 2) = o;
    }
}

}  // namespace

int launch_pyr_down(const uint8_t* d_src, size_t sstride, int sw, int sh, uint8_t* d_dst,
                    size_t dstride, hipStream_t st) {
    if (!d_src || !d_dst || sw <= 0 || sh <= 0 || sh > 131070) {
        set_last_error("pyr_down: invalid argument");
        return VS_ERR_INVALID_ARG;
    }
    const int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
    dim3 grid((dw + NT - 1) / NT, (dh + RPB - 1) / RPB);
    hipLaunchKernelGGL(pyr_down_kernel, grid, dim3(NT), 0, st, d_src, sstride, sw, sh, d_dst, dstride, dw, dh, (const ImgPair*)nullptr);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

int launch_scharr(const uint8_t* d_src, size_t sstride, int w, int h, int16_t* d_dst, hipStream_t st) {
    if (!d_src || !d_dst || w <= 0 || h <= 0 || h > 65535) {
        set_last_error("scharr: invalid argument");
        return VS_ERR_INVALID_ARG;
    }
    dim3 grid((w + NT - 1) / NT, (h + RPB - 1) / RPB);
    hipLaunchKernelGGL(scharr_kernel, grid, dim3(NT), 0, st, d_src, sstride, w, h, d_dst, (const ImgPair*)nullptr);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

int launch_pyr_down_batch(const ImgPair* d_pairs, int items, size_t sstride, int sw, int sh, size_t dstride, hipStream_t st) {
    if (!d_pairs || items < 1 || items > 65535 || sw <= 0 || sh <= 0 || sh > 131070) {
        set_last_error("pyr_down_batch: invalid argument");
        return VS_ERR_INVALID_ARG;
    }
    const int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
    dim3 grid((dw + NT - 1) / NT, (dh + RPB - 1) / RPB, items);
    hipLaunchKernelGGL(pyr_down_kernel, grid, dim3(NT), 0, st, (const uint8_t*)nullptr, sstride, sw, sh, (uint8_t*)nullptr, dstride,
                       dw, dh, d_pairs);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

int launch_scharr_batch(const ImgPair* d_pairs, int items, size_t sstride, int w, int h, hipStream_t st) {
    if (!d_pairs || items < 1 || items > 65535 || w <= 0 || h <= 0 || h > 65535) {
        set_last_error("scharr_batch: invalid argument");
        return VS_ERR_INVALID_ARG;
    }
    dim3 grid((w + NT - 1) / NT, (h + RPB - 1) / RPB, items);
    hipLaunchKernelGGL(scharr_kernel, grid, dim3(NT), 0, st, (const uint8_t*)nullptr, sstride, w, h, (int16_t*)nullptr, d_pairs);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

}  // namespace vsd
