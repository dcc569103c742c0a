// Fused cv::resize(INTER_LINEAR) + cv::cvtColor(BGR2GRAY): the analysis-image
// producer of /root/reference/src/Stabilizer.cpp:304-305 (first frame, 480x270)
// and :448-450 (every frame, 960x540), plus the single-channel resize of
// :598-603.  Reads the full-resolution frame exactly once and writes only the
// small gray image (the reference materialises a 3-channel intermediate).
//
// Integer arithmetic of the OpenCV 8-bit paths:
//  * exact 2x decimation -> INTER_AREA fast path: (s00+s01+s10+s11+2)>>2
//  * otherwise 11-bit horizontal coefficients, vertical pass
//      (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2
//  * gray = (B*3735 + G*19235 + R*9798 + 2^14) >> 15
#include "vs_common.h"

namespace vsd {
namespace {

constexpr int NT = 256;

__device__ __forceinline__ uint32_t bgr_to_gray(uint32_t b, uint32_t g, uint32_t r) {
    return (b * 3735u + g * 19235u + r * 9798u + (1u << 14)) >> 15;
}

// ---- exact 2x, BGR -> gray: 4 output pixels per lane --------------------------
__global__ __launch_bounds__(NT) void half_bgr_gray_kernel(const uint8_t* __restrict__ src_,
                                                           size_t sstride, uint8_t* __restrict__ dst_,
                                                           size_t dstride, int dw, int dh, int vec_ok,
                                                           const ImgPair* __restrict__ table) {
    // batched launch: blockIdx.z selects the frame's (source, destination) pair
    const uint8_t* __restrict__ src = table ? static_cast<const uint8_t*>(table[blockIdx.z].src) : src_;
    uint8_t* __restrict__ dst = table ? static_cast<uint8_t*>(table[blockIdx.z].dst) : dst_;
    const int gx = blockIdx.x * blockDim.x + threadIdx.x;  // group of 4 output pixels
    const int y = blockIdx.y;
    const int x = gx * 4;
    if (x >= dw || y >= dh) return;
    const uint8_t* r0 = src + (size_t)(2 * y) * sstride + (size_t)x * 6;
    const uint8_t* r1 = r0 + sstride;
    uint8_t* d = dst + (size_t)y * dstride + x;
    if (vec_ok && x + 3 < dw) {
        // 24 contiguous bytes per row and lane (8-byte aligned)
        uint32_t a[6], b[6];
        const uint2* p0 = reinterpret_cast<const uint2*>(r0);
        const uint2* p1 = reinterpret_cast<const uint2*>(r1);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            uint2 u = p0[i], v = p1[i];
            a[2 * i] = u.x; a[2 * i + 1] = u.y;
            b[2 * i] = v.x; b[2 * i + 1] = v.y;
        }
        auto byte_of = [](const uint32_t* w, int i) -> uint32_t { return (w[i >> 2] >> (8 * (i & 3))) & 255u; };
        uint32_t out = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint32_t c[3];
#pragma unroll
            for (int k = 0; k < 3; k++)
                c[k] = (byte_of(a, 6 * i + k) + byte_of(a, 6 * i + 3 + k) + byte_of(b, 6 * i + k) +
                        byte_of(b, 6 * i + 3 + k) + 2u) >> 2;
            out |= bgr_to_gray(c[0], c[1], c[2]) << (8 * i);
        }
        *reinterpret_cast<uint32_t*>(d) = out;
    } else {
        for (int i = 0; i < 4 && x + i < dw; i++) {
            uint32_t c[3];
            for (int k = 0; k < 3; k++)
                c[k] = (r0[6 * i + k] + r0[6 * i + 3 + k] + r1[6 * i + k] + r1[6 * i + 3 + k] + 2u) >> 2;
            d[i] = (uint8_t)bgr_to_gray(c[0], c[1], c[2]);
        }
    }
}

// ---- exact 2x, BGR -> gray, batched launches: 12 contiguous bytes per lane ----------------------------------------------
// The kernel above gives a lane 24 bytes of each source row as three 8-byte loads with a lane stride of 24: every cache
// line is touched by three load instructions (2.1 TB/s on the 16 re-detecting frames of a batch).  Here a lane takes TWO
// output pixels = 12 source bytes per row (one dwordx3 load; a wave's load is 768 contiguous bytes) of FOUR output rows:
// eight independent loads per lane in flight, 2-byte stores.
constexpr int HG_ROWS = 4;
struct __attribute__((aligned(4))) G3 { uint32_t a, b, c; };
__global__ __launch_bounds__(NT) void half_bgr_gray12_kernel(size_t sstride, size_t dstride, int dw, int dh, const ImgPair* __restrict__ table) {
    const uint8_t* __restrict__ src = static_cast<const uint8_t*>(table[blockIdx.z].src);
    uint8_t* __restrict__ dst = static_cast<uint8_t*>(table[blockIdx.z].dst);
    const int x = (blockIdx.x * NT + threadIdx.x) * 2;             // first of this lane's two output pixels
    const int y0 = blockIdx.y * HG_ROWS;
    if (x >= dw) return;
    G3 t[HG_ROWS], u[HG_ROWS];
#pragma unroll
    for (int r = 0; r < HG_ROWS; r++) {
        const int y = min(y0 + r, dh - 1);                          // rows past the image repeat the last one (not stored)
        const uint8_t* p = src + (size_t)(2 * y) * sstride + (size_t)x * 6;
        if (x + 1 < dw) {
            t[r] = *reinterpret_cast<const G3*>(p);
            u[r] = *reinterpret_cast<const G3*>(p + sstride);
        } else {                                                    // odd width: the last lane has one pixel (6 bytes per row)
            const uint8_t* q = p + sstride;
            t[r] = G3{(uint32_t)p[0] | p[1] << 8 | p[2] << 16 | (uint32_t)p[3] << 24, (uint32_t)p[4] | p[5] << 8, 0u};
            u[r] = G3{(uint32_t)q[0] | q[1] << 8 | q[2] << 16 | (uint32_t)q[3] << 24, (uint32_t)q[4] | q[5] << 8, 0u};
        }
    }
#pragma unroll
    for (int r = 0; r < HG_ROWS; r++) {
        const int y = y0 + r;
        if (y >= dh) break;
        // bytes 0..11 of the two rows: B0 G0 R0 B1 G1 R1 | B2 G2 R2 B3 G3 R3
        auto byte_of = [](const G3& w, int i) -> uint32_t { return ((i < 4 ? w.a : i < 8 ? w.b : w.c) >> (8 * (i & 3))) & 255u; };
        uint32_t g[2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            uint32_t c[3];
#pragma unroll
            for (int k = 0; k < 3; k++)
                c[k] = (byte_of(t[r], 6 * i + k) + byte_of(t[r], 6 * i + 3 + k) + byte_of(u[r], 6 * i + k) + byte_of(u[r], 6 * i + 3 + k) + 2u) >> 2;
            g[i] = bgr_to_gray(c[0], c[1], c[2]);
        }
        uint8_t* d = dst + (size_t)y * dstride + x;
        if (x + 1 < dw) *reinterpret_cast<uint16_t*>(d) = (uint16_t)(g[0] | (g[1] << 8));
        else d[0] = (uint8_t)g[0];
    }
}

// ---- general bilinear (any scale), CN = 3 (-> gray) or 1 ----------------------
template <int CN, bool TO_GRAY>
__global__ __launch_bounds__(NT) void resize_gray_kernel(const uint8_t* __restrict__ src_, size_t sstride,
                                                         int sw, int sh, uint8_t* __restrict__ dst_,
                                                         size_t dstride, int dw, int dh, double scale_x,
                                                         double scale_y, int area2, const ImgPair* __restrict__ table) {
    const uint8_t* __restrict__ src = table ? static_cast<const uint8_t*>(table[blockIdx.z].src) : src_;
    uint8_t* __restrict__ dst = table ? static_cast<uint8_t*>(table[blockIdx.z].dst) : dst_;
    const int dx = blockIdx.x * blockDim.x + threadIdx.x;
    const int dy = blockIdx.y;
    if (dx >= dw || dy >= dh) return;
    uint32_t v[CN];
    if (area2) {
        const uint8_t* r0 = src + (size_t)(2 * dy) * sstride + (size_t)(2 * dx) * CN;
        const uint8_t* r1 = r0 + sstride;
#pragma unroll
        for (int k = 0; k < CN; k++) v[k] = (r0[k] + r0[CN + k] + r1[k] + r1[CN + k] + 2u) >> 2;
    } else {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = f_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        bool edge = false;
        if (sx + 1 >= sw) {
            edge = true;
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        }
        const int a0 = sat_s16(f_round((1.f - fx) * 2048.f)), a1 = sat_s16(f_round(fx * 2048.f));
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = f_floor(fy);
        fy -= sy;
        const int b0 = sat_s16(f_round((1.f - fy) * 2048.f)), b1 = sat_s16(f_round(fy * 2048.f));
        int sy0 = sy, sy1 = sy + 1;
        sy0 = sy0 >= 0 ? (sy0 < sh ? sy0 : sh - 1) : 0;
        sy1 = sy1 >= 0 ? (sy1 < sh ? sy1 : sh - 1) : 0;
        const uint8_t* r0 = src + (size_t)sy0 * sstride + (size_t)sx * CN;
        const uint8_t* r1 = src + (size_t)sy1 * sstride + (size_t)sx * CN;
#pragma unroll
        for (int k = 0; k < CN; k++) {
            int h0, h1;
            if (!edge) {
                h0 = r0[k] * a0 + r0[CN + k] * a1;
                h1 = r1[k] * a0 + r1[CN + k] * a1;
            } else {
                h0 = r0[k] * 2048;
                h1 = r1[k] * 2048;
            }
            v[k] = (uint32_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
        }
    }
    if (TO_GRAY) {
        dst[(size_t)dy * dstride + dx] = (uint8_t)(CN == 3 ? bgr_to_gray(v[0], v[1], v[2]) : v[0]);
    } else {
#pragma unroll
        for (int k = 0; k < CN; k++) dst[(size_t)dy * dstride + (size_t)dx * CN + k] = (uint8_t)v[k];
    }
}

// cv::resize(INTER_LINEAR) of a one-channel picture to exactly a quarter of its width and height (the luma plane of a 3840 x 2160
// NV12 surface -> the 960 x 540 analysis image; the roll stage's x 0.25): the sample position of output pixel dx is
// (dx + 0.5) * 4 - 0.5 = 4 dx + 1.5, so both weights are 1024 of 2048 in both directions and the fixed-point form of
// resize_gray_kernel, ((1024 * ((1024 (p00 + p01)) >> 4)) >> 16 + the same of the row below + 2) >> 2, is the rounded mean of the
// 2 x 2 block at (4 dx + 1, 4 dy + 1).  A lane takes four output pixels = the bytes 1, 2 / 5, 6 / 9, 10 / 13, 14 of two 16-byte loads.
// (The general kernel - a pixel per lane, byte loads, the weights in double - took 2 x 176 us of the 845-us step at 3840 x 2160.)
constexpr int QG_ROWS = 4;      // output rows per workgroup (one per wave)
__global__ __launch_bounds__(NT) void quarter_gray_kernel(size_t sstride, size_t dstride, int dw, int dh, const ImgPair* __restrict__ table) {
    const uint8_t* __restrict__ src = static_cast<const uint8_t*>(table[blockIdx.z].src);
    uint8_t* __restrict__ dst = static_cast<uint8_t*>(table[blockIdx.z].dst);
    const int lane = threadIdx.x & 63, dy = blockIdx.y * QG_ROWS + (threadIdx.x >> 6);
    const int x4 = (blockIdx.x * 64 + lane) * 4;
    if (dy >= dh || x4 >= dw) return;
    const uint8_t* r0 = src + (size_t)(4 * dy + 1) * sstride + (size_t)4 * x4;
    const uint8_t* r1 = r0 + sstride;
    const bool vec = (((uintptr_t)src | sstride) & 15) == 0 && (((uintptr_t)dst | dstride) & 3) == 0 && x4 + 3 < dw;      // (the first two: per picture)
    if (vec) {
        const uint4 a = *reinterpret_cast<const uint4*>(r0), b = *reinterpret_cast<const uint4*>(r1);
        auto mean = [](uint32_t t, uint32_t u) {
            return (((t >> 8) & 255u) + ((t >> 16) & 255u) + ((u >> 8) & 255u) + ((u >> 16) & 255u) + 2u) >> 2;
        };
        *reinterpret_cast<uint32_t*>(dst + (size_t)dy * dstride + x4) =
            mean(a.x, b.x) | mean(a.y, b.y) << 8 | mean(a.z, b.z) << 16 | mean(a.w, b.w) << 24;
    } else {
        for (int i = 0; i < 4 && x4 + i < dw; i++)
            dst[(size_t)dy * dstride + x4 + i] = (uint8_t)((r0[4 * i + 1] + r0[4 * i + 2] + r1[4 * i + 1] + r1[4 * i + 2] + 2u) >> 2);
    }
}

}  // namespace

int launch_resize_gray_batch(const ImgPair* d_pairs, int items, size_t sstride, int sw, int sh, int fmt, size_t dstride,
                             int dw, int dh, int aligned, hipStream_t st) {
    if (!d_pairs || items < 1 || items > 65535 || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0 || dh > 65535 ||
        (fmt != VS_FMT_BGR8 && fmt != VS_FMT_GRAY8)) {
        set_last_error("resize_gray_batch: invalid argument");
        return VS_ERR_INVALID_ARG;
    }
    const double inv_x = (double)dw / sw, inv_y = (double)dh / sh;
    const double scale_x = 1. / inv_x, scale_y = 1. / inv_y;
    const int isx = (int)lrint(scale_x), isy = (int)lrint(scale_y);
    const bool area2 = std::abs(scale_x - isx) < DBL_EPSILON && std::abs(scale_y - isy) < DBL_EPSILON &&
                       isx == 2 && isy == 2;
    const uint8_t* np = nullptr;
    uint8_t* nd = nullptr;
    if (fmt == VS_FMT_BGR8 && area2) {
        const int vec_ok = aligned && (sstride % 8 == 0) && (dstride % 4 == 0);
        if (aligned && sstride % 4 == 0 && dstride % 2 == 0) {
            // (aligned: every frame of the table starts on an 8-byte boundary; the analysis images come from the library's own
            // allocation, 256-byte aligned)
            dim3 grid(((dw + 1) / 2 + NT - 1) / NT, (dh + HG_ROWS - 1) / HG_ROWS, items);
            hipLaunchKernelGGL(half_bgr_gray12_kernel, grid, dim3(NT), 0, st, sstride, dstride, dw, dh, d_pairs);
        } else {
            dim3 grid(((dw + 3) / 4 + NT - 1) / NT, dh, items);
            hipLaunchKernelGGL(half_bgr_gray_kernel, grid, dim3(NT), 0, st, np, sstride, nd, dstride, dw, dh, vec_ok, d_pairs);
        }
    } else if (fmt == VS_FMT_GRAY8 && sw == 4 * dw && sh == 4 * dh) {
        dim3 grid((dw + 255) / 256, (dh + QG_ROWS - 1) / QG_ROWS, items);
        hipLaunchKernelGGL(quarter_gray_kernel, grid, dim3(NT), 0, st, sstride, dstride, dw, dh, d_pairs);
    } else {
        dim3 grid((dw + NT - 1) / NT, dh, items);
        if (fmt == VS_FMT_BGR8)
            hipLaunchKernelGGL((resize_gray_kernel<3, true>), grid, dim3(NT), 0, st, np, sstride, sw, sh, nd, dstride, dw, dh,
                               scale_x, scale_y, area2 ? 1 : 0, d_pairs);
        else
            hipLaunchKernelGGL((resize_gray_kernel<1, true>), grid, dim3(NT), 0, st, np, sstride, sw, sh, nd, dstride, dw, dh,
                               scale_x, scale_y, area2 ? 1 : 0, d_pairs);
    }
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

int launch_resize_gray(const uint8_t* d_src, size_t sstride, int sw, int sh, int fmt,
                       uint8_t* d_dst, size_t dstride, int dw, int dh, hipStream_t st) {
    if (!d_src || !d_dst || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0 || dh > 65535 ||
        (fmt != VS_FMT_BGR8 && fmt != VS_FMT_NV12 && fmt != VS_FMT_GRAY8)) {
        set_last_error("resize_gray: invalid argument");
        return VS_ERR_INVALID_ARG;
    }
    // cv::resize: scale = 1/(dsize/ssize); INTER_LINEAR with an exact 2x2
    // decimation is served by the INTER_AREA fast path.
    const double inv_x = (double)dw / sw, inv_y = (double)dh / sh;
    const double scale_x = 1. / inv_x, scale_y = 1. / inv_y;
    const int isx = (int)lrint(scale_x), isy = (int)lrint(scale_y);
    const bool area2 = std::abs(scale_x - isx) < DBL_EPSILON && std::abs(scale_y - isy) < DBL_EPSILON &&
                       isx == 2 && isy == 2;
    if (fmt == VS_FMT_BGR8 && area2) {
        const int vec_ok = ((uintptr_t)d_src % 8 == 0) && (sstride % 8 == 0) && ((uintptr_t)d_dst % 4 == 0) &&
                           (dstride % 4 == 0);
        dim3 grid(((dw + 3) / 4 + NT - 1) / NT, dh);
        hipLaunchKernelGGL(half_bgr_gray_kernel, grid, dim3(NT), 0, st, d_src, sstride, d_dst, dstride, dw, dh, vec_ok, (const ImgPair*)nullptr);
    } else {
        dim3 grid((dw + NT - 1) / NT, dh);
        if (fmt == VS_FMT_BGR8)
            hipLaunchKernelGGL((resize_gray_kernel<3, true>), grid, dim3(NT), 0, st, d_src, sstride, sw, sh, d_dst,
                               dstride, dw, dh, scale_x, scale_y, area2 ? 1 : 0, (const ImgPair*)nullptr);
        else
            hipLaunchKernelGGL((resize_gray_kernel<1, true>), grid, dim3(NT), 0, st, d_src, sstride, sw, sh, d_dst,
                               dstride, dw, dh, scale_x, scale_y, area2 ? 1 : 0, (const ImgPair*)nullptr);
    }
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

// cv::resize(INTER_LINEAR) keeping the channels (crop-n-zoom, Stabilizer.cpp:1121)
int launch_resize_linear(const uint8_t* d_src, size_t sstride, int sw, int sh, int cn, uint8_t* d_dst,
                         size_t dstride, int dw, int dh, hipStream_t st) {
    if (!d_src || !d_dst || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0 || dh > 65535 || (cn != 1 && cn != 3)) {
        set_last_error("resize_linear: invalid argument");
        return VS_ERR_INVALID_ARG;
    }
    const double inv_x = (double)dw / sw, inv_y = (double)dh / sh;
    const double scale_x = 1. / inv_x, scale_y = 1. / inv_y;
    const int isx = (int)lrint(scale_x), isy = (int)lrint(scale_y);
    const int area2 = std::abs(scale_x - isx) < DBL_EPSILON && std::abs(scale_y - isy) < DBL_EPSILON && isx == 2 && isy == 2;
    dim3 grid((dw + NT - 1) / NT, dh);
    if (cn == 3)
        hipLaunchKernelGGL((resize_gray_kernel<3, false>), grid, dim3(NT), 0, st, d_src, sstride, sw, sh, d_dst, dstride,
                           dw, dh, scale_x, scale_y, area2, (const ImgPair*)nullptr);
    else
        hipLaunchKernelGGL((resize_gray_kernel<1, false>), grid, dim3(NT), 0, st, d_src, sstride, sw, sh, d_dst, dstride,
                           dw, dh, scale_x, scale_y, area2, (const ImgPair*)nullptr);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

}  // namespace vsd
