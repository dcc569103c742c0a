// warpAffine for gfx950: the device counterpart of
//   cv::warpAffine(src, dst, T, size, INTER_LINEAR, BORDER_CONSTANT)
// as called at /root/reference/src/Stabilizer.cpp:1056-1060 (and, with a scale
// matrix, src/AutoZoomCrop.cpp:270).  Integer fixed-point throughout:
//   inverse map in double (cv::warpAffine), AB_BITS=10 coordinates rounded per
//   row/column, 1/32-px fractions, 4-tap bilinear with weights summing to 2^15.
// The 4-tap sum is evaluated in its exactly-equivalent separable form
//   ((v00*(32-fx)+v01*fx)*(32-fy) + (v10*(32-fx)+v11*fx)*fy + 512) >> 10
// (the (0,0) table entry {32767,0,0,1} gives the same 8-bit result).
//
// HBM-bound stage: 2 x frame bytes of algorithmic traffic per frame.  One
// workgroup = one 128x16 output tile:
//   1. 144 lanes evaluate the double-precision column/row coordinate terms of
//      the tile once (adelta/bdelta per column, X0/Y0 per row) into LDS;
//   2. the source bounding box of the tile is staged into LDS with coalesced
//      12-byte/lane loads, one dword per pixel, zeros outside the image
//      (= BORDER_CONSTANT), so taps need no bounds logic;
//   3. each lane produces 4 consecutive pixels of 2 rows: taps from LDS,
//      horizontal lerps with v_dot4_u32_u8 on byte-permuted tap pairs, vertical
//      lerp in 24-bit multiplies, one 12-byte store per 4 pixels.
// Tiles whose bounding box does not fit the LDS budget (large rotation /
// scale) take a direct global-load path with the same arithmetic.
#include "vs_common.h"

namespace vsd {

namespace {

constexpr int TW = 128;      // output tile width  (pixels)
constexpr int TH = 16;       // output tile height (rows)
constexpr int PX = 4;        // consecutive output pixels per lane
constexpr int NT = 256;      // threads per workgroup
constexpr int TXN = TW / PX; // 32 lanes along x
constexpr int TYN = NT / TXN;// 8 lane-rows
constexpr int LDS_PX = 3072; // 12 KiB of staged pixels per workgroup
constexpr int MAXB = 16;     // matrices passed by value per launch
constexpr int SGW = 48;      // staging: 4-pixel groups handled per row pass (fast mapping)
constexpr int SROWS = 5;     // staging: row passes held in registers (5 rows per pass)

struct WarpArgs {
    const uint8_t* src;
    uint8_t* dst;
    size_t sstride, sframe, dstride, dframe;
    int sw, sh, dw, dh;
    int src_aligned, dst_aligned;
    const double* Minv_dev;      // batch*6 doubles on the device (inverse maps), or nullptr
    double Minv_val[MAXB * 6];   // used when Minv_dev == nullptr
    int border;                  // VS_BORDER_BLACK (constant 0) or VS_BORDER_REPLICATE
};

// One source pixel as a dword; outside the image: 0 (BORDER_CONSTANT) or the nearest
// edge pixel (BORDER_REPLICATE, remapBilinear's clip()).
template <int CN>
__device__ __forceinline__ uint32_t load_px_checked(const uint8_t* src, size_t sstride, int sw,
                                                    int sh, int sx, int sy, int border = VS_BORDER_BLACK) {
    if (border == VS_BORDER_REPLICATE) {
        sx = sx < 0 ? 0 : (sx >= sw ? sw - 1 : sx);
        sy = sy < 0 ? 0 : (sy >= sh ? sh - 1 : sy);
    } else if ((unsigned)sx >= (unsigned)sw || (unsigned)sy >= (unsigned)sh) return 0u;
    const uint8_t* p = src + (size_t)sy * sstride + (size_t)sx * CN;
    uint32_t v = p[0];
    if (CN > 1) v |= (uint32_t)p[1] << 8;
    if (CN > 2) v |= (uint32_t)p[2] << 16;
    return v;
}

// p00/p01 = taps of the upper row, p10/p11 of the lower row (one pixel per dword,
// channel c in byte c).  Horizontal lerps as byte dot products, vertical in 24 bits.
template <int CN>
__device__ __forceinline__ uint32_t blend(uint32_t p00, uint32_t p01, uint32_t p10, uint32_t p11,
                                          uint32_t fx, uint32_t fy) {
    const uint32_t wlo = (32u - fx) | (fx << 8);     // weights against bytes 0,1
    const uint32_t whi = wlo << 16;                    // weights against bytes 2,3
    const uint32_t wy0 = 32u - fy, wy1 = fy;
    // (c0_a, c0_b, c1_a, c1_b): channels 0 and 1 of the two taps side by side
    const uint32_t x0 = __builtin_amdgcn_perm(p01, p00, 0x05010400u);
    const uint32_t x1 = __builtin_amdgcn_perm(p11, p10, 0x05010400u);
    uint32_t out;
    {
        const uint32_t t = __builtin_amdgcn_udot4(x0, wlo, 0u, false);
        const uint32_t b = __builtin_amdgcn_udot4(x1, wlo, 0u, false);
        out = (__umul24(t, wy0) + __umul24(b, wy1) + 512u) >> 10;
    }
    if (CN > 1) {
        const uint32_t t = __builtin_amdgcn_udot4(x0, whi, 0u, false);
        const uint32_t b = __builtin_amdgcn_udot4(x1, whi, 0u, false);
        out |= ((__umul24(t, wy0) + __umul24(b, wy1) + 512u) >> 10) << 8;
    }
    if (CN > 2) {
        const uint32_t y0 = __builtin_amdgcn_perm(p01, p00, 0x0C0C0602u);   // (c2_a, c2_b, 0, 0)
        const uint32_t y1 = __builtin_amdgcn_perm(p11, p10, 0x0C0C0602u);
        const uint32_t t = __builtin_amdgcn_udot4(y0, wlo, 0u, false);
        const uint32_t b = __builtin_amdgcn_udot4(y1, wlo, 0u, false);
        out |= ((__umul24(t, wy0) + __umul24(b, wy1) + 512u) >> 10) << 16;
    }
    return out;
}

struct __attribute__((aligned(4))) U3 { uint32_t a, b, c; };

template <int CN>
__device__ __forceinline__ uint4 stage_group(const WarpArgs& a, const uint8_t* __restrict__ src, int sx, int sy) {
    uint4 px = make_uint4(0u, 0u, 0u, 0u);
    if ((unsigned)sy < (unsigned)a.sh || a.border == VS_BORDER_REPLICATE) {
        if (a.src_aligned && sx >= 0 && sx + 3 < a.sw && (unsigned)sy < (unsigned)a.sh) {
            const uint8_t* p = src + (size_t)sy * a.sstride + (size_t)sx * CN;
            if (CN == 3) {
                const U3 d = *reinterpret_cast<const U3*>(p);
                px.x = d.a & 0xFFFFFFu;
                px.y = (d.a >> 24) | ((d.b & 0xFFFFu) << 8);
                px.z = (d.b >> 16) | ((d.c & 0xFFu) << 16);
                px.w = d.c >> 8;
            } else if (CN == 1) {
                const uint32_t d = *reinterpret_cast<const uint32_t*>(p);
                px.x = d & 255u; px.y = (d >> 8) & 255u; px.z = (d >> 16) & 255u; px.w = d >> 24;
            } else {
                const uint2 d = *reinterpret_cast<const uint2*>(p);
                px.x = d.x & 0xFFFFu; px.y = d.x >> 16; px.z = d.y & 0xFFFFu; px.w = d.y >> 16;
            }
        } else {
            px.x = load_px_checked<CN>(src, a.sstride, a.sw, a.sh, sx, sy, a.border);
            px.y = load_px_checked<CN>(src, a.sstride, a.sw, a.sh, sx + 1, sy, a.border);
            px.z = load_px_checked<CN>(src, a.sstride, a.sw, a.sh, sx + 2, sy, a.border);
            px.w = load_px_checked<CN>(src, a.sstride, a.sw, a.sh, sx + 3, sy, a.border);
        }
    }
    return px;
}

template <int CN, bool USE_LDS>
__device__ __forceinline__ void emit_rows(const WarpArgs& a, const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                          const uint32_t* tile, const int* s_ad, const int* s_bd, const int* s_x0,
                                          const int* s_y0, int x0, int y0, int x1, int y1, int bx0a, int by0, int bw) {
    const int tid = threadIdx.x;
    const int tx = tid % TXN, ty = tid / TXN;
    const int x = x0 + PX * tx;
    if (x > x1) return;
    int ad[PX], bd[PX];
#pragma unroll
    for (int i = 0; i < PX; i++) { ad[i] = s_ad[PX * tx + i]; bd[i] = s_bd[PX * tx + i]; }
    uint32_t o[TH / TYN][PX];
#pragma unroll
    for (int r = 0; r < TH / TYN; r++) {
        const int yl = ty + TYN * r;
        const int X0 = s_x0[yl], Y0 = s_y0[yl];
        uint32_t p00[PX], p01[PX], p10[PX], p11[PX], fx[PX], fy[PX];
#pragma unroll
        for (int i = 0; i < PX; i++) {
            const int X = (X0 + ad[i]) >> 5, Y = (Y0 + bd[i]) >> 5;
            const int sx = sat_s16(X >> 5), sy = sat_s16(Y >> 5);
            fx[i] = X & 31; fy[i] = Y & 31;
            if (USE_LDS) {
                // rows beyond the tile (yl past y1) still index inside the staged box of a
                // full tile only; clamp them to the first row so the reads stay in bounds
                const int idx = y0 + yl <= y1 ? __mul24(sy - by0, bw) + (sx - bx0a) : 0;
                p00[i] = tile[idx]; p01[i] = tile[idx + 1];
                p10[i] = tile[idx + bw]; p11[i] = tile[idx + bw + 1];
            } else {
                p00[i] = load_px_checked<CN>(src, a.sstride, a.sw, a.sh, sx, sy, a.border);
                p01[i] = load_px_checked<CN>(src, a.sstride, a.sw, a.sh, sx + 1, sy, a.border);
                p10[i] = load_px_checked<CN>(src, a.sstride, a.sw, a.sh, sx, sy + 1, a.border);
                p11[i] = load_px_checked<CN>(src, a.sstride, a.sw, a.sh, sx + 1, sy + 1, a.border);
            }
        }
#pragma unroll
        for (int i = 0; i < PX; i++) o[r][i] = blend<CN>(p00[i], p01[i], p10[i], p11[i], fx[i], fy[i]);
    }
#pragma unroll
    for (int r = 0; r < TH / TYN; r++) {
        const int y = y0 + ty + TYN * r;
        if (y > y1) break;
        uint8_t* d = dst + (size_t)y * a.dstride + (size_t)x * CN;
        if (a.dst_aligned && x + PX - 1 <= x1) {
            if (CN == 3) {
                U3 v;
                v.a = o[r][0] | (o[r][1] << 24);
                v.b = (o[r][1] >> 8) | (o[r][2] << 16);
                v.c = (o[r][2] >> 16) | (o[r][3] << 8);
                *reinterpret_cast<U3*>(d) = v;
            } else if (CN == 1) {
                *reinterpret_cast<uint32_t*>(d) = o[r][0] | (o[r][1] << 8) | (o[r][2] << 16) | (o[r][3] << 24);
            } else {
                *reinterpret_cast<uint2*>(d) = make_uint2(o[r][0] | (o[r][1] << 16), o[r][2] | (o[r][3] << 16));
            }
        } else {
#pragma unroll
            for (int i = 0; i < PX; i++) {
                if (x + i > x1) break;
#pragma unroll
                for (int c = 0; c < CN; c++) d[i * CN + c] = (uint8_t)(o[r][i] >> (8 * c));
            }
        }
    }
}

// hal::warpAffine / WarpAffineInvoker coordinate terms (1/1024 px; +16 = round_delta)
__device__ __forceinline__ void col_terms(const double* m, int x, int& ad, int& bd) {
    ad = d_round(m[0] * x * 1024);
    bd = d_round(m[3] * x * 1024);
}
__device__ __forceinline__ void row_terms(const double* m, int y, int& X0, int& Y0) {
    X0 = d_round((m[1] * y + m[2]) * 1024) + 16;
    Y0 = d_round((m[4] * y + m[5]) * 1024) + 16;
}

template <int CN>
__global__ __launch_bounds__(NT) void warp_affine_kernel(WarpArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t tile[LDS_PX];
    __shared__ int s_ad[TW], s_bd[TW], s_x0[TH], s_y0[TH];
    const int bz = blockIdx.z;
    const uint8_t* __restrict__ src = a.src + (size_t)bz * a.sframe;
    uint8_t* __restrict__ dst = a.dst + (size_t)bz * a.dframe;
    double m[6];   // inverse map of this frame (wave-uniform)
    {
        const double* mp = a.Minv_dev ? a.Minv_dev + 6 * bz : a.Minv_val + 6 * bz;
#pragma unroll
        for (int i = 0; i < 6; i++) m[i] = mp[i];
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int x1 = min(x0 + TW, a.dw) - 1, y1 = min(y0 + TH, a.dh) - 1;

    // ---- 1. source bounding box.  The maps are monotone in x and in y separately, so the
    // extremes are at the tile corners: lane 0 evaluates (x0,y0), lane 1 (x1,y1), and the
    // eight terms are broadcast with v_readlane (no LDS, no barrier before the loads).
    int bx0, bx1, by0, by1;
    {
        int ad, bd, X0, Y0;
        col_terms(m, (lane & 1) ? x1 : x0, ad, bd);
        row_terms(m, (lane & 1) ? y1 : y0, X0, Y0);
        const int ad0 = __builtin_amdgcn_readlane(ad, 0), ad1 = __builtin_amdgcn_readlane(ad, 1);
        const int bd0 = __builtin_amdgcn_readlane(bd, 0), bd1 = __builtin_amdgcn_readlane(bd, 1);
        const int Xa = __builtin_amdgcn_readlane(X0, 0), Xb = __builtin_amdgcn_readlane(X0, 1);
        const int Ya = __builtin_amdgcn_readlane(Y0, 0), Yb = __builtin_amdgcn_readlane(Y0, 1);
        const int sx00 = sat_s16((Xa + ad0) >> 10), sx01 = sat_s16((Xa + ad1) >> 10);
        const int sx10 = sat_s16((Xb + ad0) >> 10), sx11 = sat_s16((Xb + ad1) >> 10);
        const int sy00 = sat_s16((Ya + bd0) >> 10), sy01 = sat_s16((Ya + bd1) >> 10);
        const int sy10 = sat_s16((Yb + bd0) >> 10), sy11 = sat_s16((Yb + bd1) >> 10);
        bx0 = min(min(sx00, sx01), min(sx10, sx11));
        bx1 = max(max(sx00, sx01), max(sx10, sx11)) + 1;
        by0 = min(min(sy00, sy01), min(sy10, sy11));
        by1 = max(max(sy00, sy01), max(sy10, sy11)) + 1;
    }
    const int bx0a = bx0 & ~3;                       // 4-pixel (12-byte) aligned start
    const int bw = (bx1 - bx0a + 1 + 3) & ~3;        // staged width, multiple of 4
    const int bh = by1 - by0 + 1;
    const bool use_lds = (long long)bw * bh <= LDS_PX;
    const int gpr = bw >> 2;                         // 4-pixel groups per staged row
    const bool fast_stage = use_lds && gpr <= SGW && bh <= 5 * SROWS;

    // ---- 2. staging loads are issued first (5 rows x 48 groups per pass, no runtime division) ...
    const int ly = tid / SGW, lx = tid - ly * SGW;
    const bool stager = tid < 5 * SGW && lx < gpr;
    uint4 staged[SROWS];
    if (fast_stage && stager) {
#pragma unroll
        for (int k = 0; k < SROWS; k++) {
            const int row = ly + 5 * k;
            staged[k] = row < bh ? stage_group<CN>(a, src, bx0a + 4 * lx, by0 + row) : make_uint4(0u, 0u, 0u, 0u);
        }
    }
    // ---- ... the per-column / per-row coordinate terms of the tile are computed while those
    // loads are in flight (adelta/bdelta per column, X0/Y0 per row, once per tile) ...
    if (tid < TW) {
        int ad, bd;
        col_terms(m, x0 + tid, ad, bd);
        s_ad[tid] = ad; s_bd[tid] = bd;
    } else if (tid < TW + TH) {
        int X0, Y0;
        row_terms(m, y0 + (tid - TW), X0, Y0);
        s_x0[tid - TW] = X0; s_y0[tid - TW] = Y0;
    }
    // ---- ... and then the staged pixels go to LDS (one dword per pixel, zeros outside the image)
    if (fast_stage) {
        if (stager) {
#pragma unroll
            for (int k = 0; k < SROWS; k++) {
                const int row = ly + 5 * k;
                if (row < bh) *reinterpret_cast<uint4*>(&tile[row * bw + 4 * lx]) = staged[k];
            }
        }
    } else if (use_lds) {
        const int total = gpr * bh;
        for (int g = tid; g < total; g += NT) {
            const int row = g / gpr, gx = g - row * gpr;
            const uint4 px = stage_group<CN>(a, src, bx0a + 4 * gx, by0 + row);
            *reinterpret_cast<uint4*>(&tile[row * bw + 4 * gx]) = px;
        }
    }
    __syncthreads();

    // ---- 3. output: 4 consecutive pixels x 2 rows per lane (the LDS / direct choice is
    // tile-uniform: two straight-line bodies, so all taps of a lane are in flight together)
    if (use_lds) emit_rows<CN, true>(a, src, dst, tile, s_ad, s_bd, s_x0, s_y0, x0, y0, x1, y1, bx0a, by0, bw);
    else emit_rows<CN, false>(a, src, dst, tile, s_ad, s_bd, s_x0, s_y0, x0, y0, x1, y1, bx0a, by0, bw);
}

template <int CN>
void launch_one(const WarpArgs& a, dim3 grid, hipStream_t st) {
    hipLaunchKernelGGL(warp_affine_kernel<CN>, grid, dim3(NT), 0, st, a);
}

void fill_common(WarpArgs& a, const uint8_t* d_src, size_t sstride, size_t sframe, int sw, int sh, uint8_t* d_dst,
                 size_t dstride, size_t dframe, int dw, int dh, int cn) {
    a.src = d_src; a.dst = d_dst;
    a.sstride = sstride; a.sframe = sframe; a.dstride = dstride; a.dframe = dframe;
    a.sw = sw; a.sh = sh; a.dw = dw; a.dh = dh;
    a.border = VS_BORDER_BLACK;
    const int galign = cn == 2 ? 8 : 4;
    a.src_aligned = ((uintptr_t)d_src % galign == 0) && (sstride % galign == 0) && (sframe % galign == 0);
    a.dst_aligned = ((uintptr_t)d_dst % galign == 0) && (dstride % galign == 0) && (dframe % galign == 0);
}

bool bad_args(const void* d_src, const void* d_dst, const void* M, size_t sstride, int sw, int sh, size_t dstride,
              int dw, int dh, int cn, int batch) {
    return !d_src || !d_dst || !M || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0 || batch <= 0 ||
           (cn != 1 && cn != 2 && cn != 3) || sstride < (size_t)sw * cn || dstride < (size_t)dw * cn ||
           dh > 65535 * TH;
}

}  // namespace

int launch_warp_affine(const uint8_t* d_src, size_t sstride, size_t sframe, int sw, int sh,
                       uint8_t* d_dst, size_t dstride, size_t dframe, int dw, int dh, int cn,
                       const double* d_Minv, int batch, hipStream_t st) {
    if (bad_args(d_src, d_dst, d_Minv, sstride, sw, sh, dstride, dw, dh, cn, batch) || batch > 65535) {
        set_last_error("warp_affine: invalid argument");
        return VS_ERR_INVALID_ARG;
    }
    WarpArgs a;
    fill_common(a, d_src, sstride, sframe, sw, sh, d_dst, dstride, dframe, dw, dh, cn);
    a.Minv_dev = d_Minv;
    dim3 grid((dw + TW - 1) / TW, (dh + TH - 1) / TH, batch);
    if (cn == 3) launch_one<3>(a, grid, st);
    else if (cn == 1) launch_one<1>(a, grid, st);
    else launch_one<2>(a, grid, st);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

// Host-matrix form used by vs_op_warp_affine: matrices travel as kernel
// arguments (MAXB per launch), so no staging buffer or sync is needed.
int launch_warp_affine_hostM(const uint8_t* d_src, size_t sstride, size_t sframe, int sw, int sh,
                             uint8_t* d_dst, size_t dstride, size_t dframe, int dw, int dh, int cn,
                             const float* h_M, int batch, hipStream_t st) {
    if (bad_args(d_src, d_dst, h_M, sstride, sw, sh, dstride, dw, dh, cn, batch)) {
        set_last_error("warp_affine: invalid argument");
        return VS_ERR_INVALID_ARG;
    }
    for (int b0 = 0; b0 < batch; b0 += MAXB) {
        const int nb = batch - b0 < MAXB ? batch - b0 : MAXB;
        WarpArgs a;
        fill_common(a, d_src + (size_t)b0 * sframe, sstride, sframe, sw, sh, d_dst + (size_t)b0 * dframe, dstride,
                    dframe, dw, dh, cn);
        a.Minv_dev = nullptr;
        for (int b = 0; b < MAXB; b++) {
            if (b < nb) warp_invert(h_M + (size_t)(b0 + b) * 6, a.Minv_val + 6 * b);   // cv::warpAffine inverts on the host too
            else for (int i = 0; i < 6; i++) a.Minv_val[6 * b + i] = 0.;
        }
        dim3 grid((dw + TW - 1) / TW, (dh + TH - 1) / TH, nb);
        if (cn == 3) launch_one<3>(a, grid, st);
        else if (cn == 1) launch_one<1>(a, grid, st);
        else launch_one<2>(a, grid, st);
        VS_HIP_TRY(hipGetLastError());
    }
    return VS_OK;
}

// One frame, inverse map given on the host in double, selectable border: used by the
// roll-correction rotate (cv::warpAffine(..., BORDER_REPLICATE)) and AutoZoomCrop's scale.
int launch_warp_affine_inv(const uint8_t* d_src, size_t sstride, int sw, int sh, uint8_t* d_dst, size_t dstride,
                           int dw, int dh, int cn, const double* h_Minv, int border, hipStream_t st) {
    if (bad_args(d_src, d_dst, h_Minv, sstride, sw, sh, dstride, dw, dh, cn, 1) ||
        (border != VS_BORDER_BLACK && border != VS_BORDER_REPLICATE)) {
        set_last_error("warp_affine: invalid argument");
        return VS_ERR_INVALID_ARG;
    }
    WarpArgs a;
    fill_common(a, d_src, sstride, 0, sw, sh, d_dst, dstride, 0, dw, dh, cn);
    a.Minv_dev = nullptr;
    a.border = border;
    for (int i = 0; i < MAXB * 6; i++) a.Minv_val[i] = i < 6 ? h_Minv[i] : 0.;
    dim3 grid((dw + TW - 1) / TW, (dh + TH - 1) / TH, 1);
    if (cn == 3) launch_one<3>(a, grid, st);
    else if (cn == 1) launch_one<1>(a, grid, st);
    else launch_one<2>(a, grid, st);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

}  // namespace vsd
