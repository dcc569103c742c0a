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
                                                      size_t dstride, int dw, int dh) {
    const uint8_t* __restrict__ src = src_;
    uint8_t* __restrict__ dst = dst_;
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
                                                    int h, int16_t* __restrict__ dst_) {
    const uint8_t* __restrict__ src = src_;
    int16_t* __restrict__ dst = dst_;
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

// ---- one pyramid level per launch (batch mode) --------------------------------------------------------------
// Level l of the LK pyramid needs two things of image l: its Scharr derivatives and, if there is a level l+1, its
// pyrDown.  The two stencils above read the image separately (5 launches for 3 levels, 25 + 9 byte loads per output
// pixel through L1).  Here a workgroup stages a 64 x 16 tile of image l with a 2-pixel rim (REFLECT_101 applied while
// staging: both stencils use it on source coordinates) in LDS once - dword loads when the tile is inside the image -
// and produces the tile's 64 x 16 derivative pairs (4 per lane, one 16-byte store) and its 32 x 8 pixels of level l+1
// (1 per lane): 3 launches for 3 levels, every level read once.  The kernel is bound by dependent LDS round trips, not by
// bytes or arithmetic: with one ds_read_u8 per tap (18 + 25 per lane) level 0 of a 32-frame batch took 73 - 81 us, with
// the taps fetched as dwords (9 + 10 reads, bytes picked out in registers, the 1 4 6 4 row of pyrDown as one v_dot4) 25 - 35.
constexpr int FT_W = 64, FT_H = 16;                    // tile of image l
constexpr int FT_PITCH = 72;                           // staged row: 68 bytes used (x0 - 2 .. x0 + 65), dword aligned
constexpr int FT_ROWS = FT_H + 4;                      // y0 - 2 .. y0 + 17

template <bool DOWN>
__global__ __launch_bounds__(NT) void pyr_level_kernel(const ImgPair* __restrict__ sch, const ImgPair* __restrict__ pyr, size_t sstride,
                                                       int w, int h, size_t dstride, int dw, int dh) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[FT_ROWS * FT_PITCH];
    const uint8_t* __restrict__ src = static_cast<const uint8_t*>(sch[blockIdx.z].src);
    int16_t* __restrict__ der = static_cast<int16_t*>(sch[blockIdx.z].dst);
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * FT_W, y0 = blockIdx.y * FT_H;
    // ---- stage rows y0-2 .. y0+17, columns x0-4 .. x0+67 (the dword grid of the image; tile column c = x - (x0 - 4))
    const bool inside = x0 >= 4 && y0 >= 2 && x0 + FT_W + 4 <= w && y0 + FT_H + 2 <= h && (sstride & 3) == 0 && ((uintptr_t)src & 3) == 0;
    // (Both forms: every load of a lane goes out before its first LDS store - addresses first, then the loads without
    // conditions, then the stores.  With a condition or a reflection loop between two loads the compiler waits for each in
    // turn: two round trips per interior tile, seven per border tile, and 18 % of a 960 x 540 image's tiles are border tiles.)
    if (inside) {
        constexpr int ND = (FT_ROWS * (FT_PITCH / 4) + NT - 1) / NT;
        uint32_t v[ND];
#pragma unroll
        for (int k = 0; k < ND; k++) {
            const int i = min(tid + NT * k, FT_ROWS * (FT_PITCH / 4) - 1);
            const int r = i / (FT_PITCH / 4), c4 = i - r * (FT_PITCH / 4);
            v[k] = *reinterpret_cast<const uint32_t*>(src + (size_t)(y0 - 2 + r) * sstride + (x0 - 4) + 4 * c4);
        }
#pragma unroll
        for (int k = 0; k < ND; k++)
            if (tid + NT * k < FT_ROWS * (FT_PITCH / 4)) reinterpret_cast<uint32_t*>(tile)[tid + NT * k] = v[k];
    } else if (tid < 3 * FT_PITCH) {
        // a tile on the image's border: a lane keeps one column (its reflection is computed once) and takes every third row
        constexpr int NB = (FT_ROWS + 2) / 3;
        const int c = tid % FT_PITCH, r0 = tid / FT_PITCH;
        const uint8_t* col = src + reflect101(x0 - 4 + c, w);
        size_t off[NB];
#pragma unroll
        for (int k = 0; k < NB; k++) off[k] = (size_t)reflect101(y0 - 2 + min(r0 + 3 * k, FT_ROWS - 1), h) * sstride;
        uint8_t v[NB];
#pragma unroll
        for (int k = 0; k < NB; k++) v[k] = col[off[k]];
#pragma unroll
        for (int k = 0; k < NB; k++)
            if (r0 + 3 * k < FT_ROWS) tile[(r0 + 3 * k) * FT_PITCH + c] = v[k];
    }
    __syncthreads();
    const uint32_t* T = reinterpret_cast<const uint32_t*>(tile);
    // ---- Scharr: lane -> row tid / 16, columns 4 * (tid % 16) .. + 3.  Three dwords per row (the byte left of the group,
    // the group, the byte right of it) instead of six byte reads: the kernel is bound by dependent LDS round trips
    {
        const int r = tid >> 4, c = (tid & 15) * 4;
        const int y = y0 + r, x = x0 + c;
        if (y < h && x < w) {
            int t[3][6];                                                // rows y-1 .. y+1, columns x-1 .. x+4
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const uint32_t* q = T + (r + 1 + j) * (FT_PITCH / 4) + (c >> 2);     // tile columns c .. c+11
                const uint32_t lo = q[0], mid = q[1], hi = q[2];
                t[j][0] = lo >> 24;
                t[j][1] = mid & 255u; t[j][2] = (mid >> 8) & 255u; t[j][3] = (mid >> 16) & 255u; t[j][4] = mid >> 24;
                t[j][5] = hi & 255u;
            }
            int a[6], b[6];                                             // vertical smoothing / difference per column
#pragma unroll
            for (int k = 0; k < 6; k++) {
                a[k] = (t[0][k] + t[2][k]) * 3 + t[1][k] * 10;
                b[k] = t[2][k] - t[0][k];
            }
            uint32_t o[4];                                              // (dx, dy) as two 16-bit halves
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int dx = a[k + 2] - a[k], dy = (b[k + 2] + b[k]) * 3 + b[k + 1] * 10;
                o[k] = ((uint32_t)dx & 0xFFFFu) | ((uint32_t)dy << 16);
            }
            int16_t* d = der + ((size_t)y * w + x) * 2;
            if (x + 3 < w && (((uintptr_t)d) & 15) == 0) {
                *reinterpret_cast<uint4*>(d) = make_uint4(o[0], o[1], o[2], o[3]);
            } else {
                for (int k = 0; k < 4 && x + k < w; k++) reinterpret_cast<uint32_t*>(d)[k] = o[k];
            }
        }
    }
    // ---- pyrDown: lane -> output (x0 / 2 + tid % 32, y0 / 2 + tid / 32)
    if (DOWN) {
        uint8_t* __restrict__ dst = static_cast<uint8_t*>(pyr[blockIdx.z].dst);
        const int xo = (x0 >> 1) + (tid & 31), yo = (y0 >> 1) + (tid >> 5);
        if (xo < dw && yo < dh) {
            // source rows 2 yo - 2 .. + 2 = tile rows 2 (tid / 32) .. + 4; columns 2 xo - 2 .. + 2 = tile columns o .. o + 4 with
            // o = 2 (tid % 32) + 2: two dwords per row, the five taps as one byte-aligned dword (1 4 6 4 by v_dot4) plus one byte
            const int o = 2 * (tid & 31) + 2, sh = o & 3;              // sh = 0 or 2
            const uint32_t* q = T + 2 * (tid >> 5) * (FT_PITCH / 4) + (o >> 2);
            int col[5];
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const uint32_t lo = q[j * (FT_PITCH / 4)], hi = q[j * (FT_PITCH / 4) + 1];
                const uint32_t four = __builtin_amdgcn_alignbyte(hi, lo, sh);             // taps 0 .. 3
                const uint32_t fifth = (hi >> (8 * sh)) & 255u;                              // tap 4
                col[j] = (int)__builtin_amdgcn_udot4(four, 0x04060401u, fifth, false);
            }
            dst[(size_t)yo * dstride + xo] = (uint8_t)((col[2] * 6 + (col[1] + col[3]) * 4 + col[0] + col[4] + 128) >> 8);
        }
    }
}

}  // namespace

// Derivatives of level l (d_scharr_pairs: image l -> derivative image) and, when d_pyr_pairs is given, level l+1
// (d_pyr_pairs: image l -> image l+1) of `items` frames.  w, h: size of image l (row pitch sstride); dstride: pitch of l+1.
int launch_pyr_level_batch(const ImgPair* d_scharr_pairs, const ImgPair* d_pyr_pairs, int items, size_t sstride, int w, int h,
                           size_t dstride, hipStream_t st) {
    if (!d_scharr_pairs || items < 1 || items > 65535 || w <= 0 || h <= 0 || h > 65535) {
        set_last_error("pyr_level_batch: invalid argument");
        return VS_ERR_INVALID_ARG;
    }
    const int dw = (w + 1) / 2, dh = (h + 1) / 2;
    dim3 grid((w + FT_W - 1) / FT_W, (h + FT_H - 1) / FT_H, items);
    if (d_pyr_pairs) hipLaunchKernelGGL(pyr_level_kernel<true>, grid, dim3(NT), 0, st, d_scharr_pairs, d_pyr_pairs, sstride, w, h, dstride, dw, dh);
    else hipLaunchKernelGGL(pyr_level_kernel<false>, grid, dim3(NT), 0, st, d_scharr_pairs, d_pyr_pairs, sstride, w, h, dstride, dw, dh);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

int launch_pyr_down(const uint8_t* d_src, size_t sstride, int sw, int sh, uint8_t* d_dst,
                    size_t dstride, hipStream_t st) {
    if (!d_src || !d_dst || sw <= 0 || sh <= 0 || sh > 131070) {
        set_last_error("pyr_down: invalid argument");
        return VS_ERR_INVALID_ARG;
    }
    const int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
    dim3 grid((dw + NT - 1) / NT, (dh + RPB - 1) / RPB);
    hipLaunchKernelGGL(pyr_down_kernel, grid, dim3(NT), 0, st, d_src, sstride, sw, sh, d_dst, dstride, dw, dh);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

int launch_scharr(const uint8_t* d_src, size_t sstride, int w, int h, int16_t* d_dst, hipStream_t st) {
    if (!d_src || !d_dst || w <= 0 || h <= 0 || h > 65535) {
        set_last_error("scharr: invalid argument");
        return VS_ERR_INVALID_ARG;
    }
    dim3 grid((w + NT - 1) / NT, (h + RPB - 1) / RPB);
    hipLaunchKernelGGL(scharr_kernel, grid, dim3(NT), 0, st, d_src, sstride, w, h, d_dst);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

}  // namespace vsd
