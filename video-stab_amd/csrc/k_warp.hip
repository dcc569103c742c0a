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
// HBM-bound kernel (12 B in / 12 B out per BGR pixel-pair... 2 x frame bytes
// per frame).  One workgroup = one 128x16 output tile: the source bounding box
// of the tile is staged once into LDS with coalesced 12-byte/lane loads
// (one dword per pixel, zeros outside the image = BORDER_CONSTANT), every
// output pixel then takes its 4 taps from LDS and each lane writes 4 pixels
// with one 12-byte store.  Tiles whose bounding box does not fit the LDS
// budget (large rotation / scale) take a direct global-load path.
#include "vs_common.h"

namespace vsd {

namespace {

constexpr int TW = 128;      // output tile width  (pixels)
constexpr int TH = 16;       // output tile height (rows)
constexpr int PX = 4;        // consecutive output pixels per lane
constexpr int NT = 256;      // threads per workgroup
constexpr int TXN = TW / PX; // 32 lanes along x
constexpr int TYN = NT / TXN;// 8 lane-rows
constexpr int LDS_PX = 6144; // 24 KiB of staged pixels per workgroup
constexpr int MAXB = 16;     // matrices passed by value per launch

struct WarpArgs {
    const uint8_t* src;
    uint8_t* dst;
    size_t sstride, sframe, dstride, dframe;
    int sw, sh, dw, dh;
    int src_aligned, dst_aligned;
    const float* M_dev;          // batch*6 floats on the device, or nullptr
    float M_val[MAXB * 6];       // used when M_dev == nullptr
};

struct InvMap { double m[6]; };

// cv::warpAffine: invert the forward matrix in double.
__device__ __forceinline__ InvMap invert(const float* Mf) {
    double M[6];
#pragma unroll
    for (int i = 0; i < 6; i++) M[i] = (double)Mf[i];
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11; M[1] *= -D;
    M[3] *= -D; M[4] = A22;
    double b1 = -M[0] * M[2] - M[1] * M[5];
    double b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1; M[5] = b2;
    InvMap r;
#pragma unroll
    for (int i = 0; i < 6; i++) r.m[i] = M[i];
    return r;
}

// hal::warpAffine / WarpAffineInvoker coordinate generation (1/32 px units).
__device__ __forceinline__ void row_base(const InvMap& iv, int y, int& X0, int& Y0) {
    X0 = d_round((iv.m[1] * y + iv.m[2]) * 1024) + 16;
    Y0 = d_round((iv.m[4] * y + iv.m[5]) * 1024) + 16;
}
__device__ __forceinline__ void col_delta(const InvMap& iv, int x, int& ad, int& bd) {
    ad = d_round(iv.m[0] * x * 1024);
    bd = d_round(iv.m[3] * x * 1024);
}

template <int CN>
__device__ __forceinline__ uint32_t load_px_checked(const uint8_t* src, size_t sstride, int sw,
                                                    int sh, int sx, int sy) {
    if ((unsigned)sx >= (unsigned)sw || (unsigned)sy >= (unsigned)sh) return 0u;
    const uint8_t* p = src + (size_t)sy * sstride + (size_t)sx * CN;
    uint32_t v = p[0];
    if (CN > 1) v |= (uint32_t)p[1] << 8;
    if (CN > 2) v |= (uint32_t)p[2] << 16;
    return v;
}

template <int CN>
__device__ __forceinline__ uint32_t blend(uint32_t p00, uint32_t p01, uint32_t p10, uint32_t p11,
                                          int fx, int fy) {
    const uint32_t wx1 = fx, wx0 = 32 - fx, wy1 = fy, wy0 = 32 - fy;
    uint32_t out = 0;
#pragma unroll
    for (int c = 0; c < CN; c++) {
        uint32_t v00 = (p00 >> (8 * c)) & 255u, v01 = (p01 >> (8 * c)) & 255u;
        uint32_t v10 = (p10 >> (8 * c)) & 255u, v11 = (p11 >> (8 * c)) & 255u;
        uint32_t t = v00 * wx0 + v01 * wx1;
        uint32_t b = v10 * wx0 + v11 * wx1;
        uint32_t r = (t * wy0 + b * wy1 + 512u) >> 10;
        out |= r << (8 * c);
    }
    return out;
}

struct __attribute__((aligned(4))) U3 { uint32_t a, b, c; };

template <int CN>
__global__ __launch_bounds__(NT) void warp_affine_kernel(WarpArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t tile[LDS_PX];
    const int bz = blockIdx.z;
    const uint8_t* __restrict__ src = a.src + (size_t)bz * a.sframe;
    uint8_t* __restrict__ dst = a.dst + (size_t)bz * a.dframe;
    const float* Mf = a.M_dev ? a.M_dev + 6 * bz : a.M_val + 6 * bz;
    const InvMap iv = invert(Mf);

    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int x1 = min(x0 + TW, a.dw) - 1, y1 = min(y0 + TH, a.dh) - 1;

    // Source bounding box of the tile: the coordinate maps are monotone in x
    // and in y separately, so the extremes are at the tile corners.
    int bx0, bx1, by0, by1;
    {
        int Xa, Ya, Xb, Yb, ad0, bd0, ad1, bd1;
        row_base(iv, y0, Xa, Ya);
        row_base(iv, y1, Xb, Yb);
        col_delta(iv, x0, ad0, bd0);
        col_delta(iv, x1, ad1, bd1);
        int sx00 = sat_s16((Xa + ad0) >> 10), sx01 = sat_s16((Xa + ad1) >> 10);
        int sx10 = sat_s16((Xb + ad0) >> 10), sx11 = sat_s16((Xb + ad1) >> 10);
        int sy00 = sat_s16((Ya + bd0) >> 10), sy01 = sat_s16((Ya + bd1) >> 10);
        int sy10 = sat_s16((Yb + bd0) >> 10), sy11 = sat_s16((Yb + bd1) >> 10);
        bx0 = min(min(sx00, sx01), min(sx10, sx11));
        bx1 = max(max(sx00, sx01), max(sx10, sx11)) + 1;
        by0 = min(min(sy00, sy01), min(sy10, sy11));
        by1 = max(max(sy00, sy01), max(sy10, sy11)) + 1;
    }
    const int bx0a = bx0 & ~3;                       // 4-pixel (12-byte) aligned start
    const int bw = (bx1 - bx0a + 1 + 3) & ~3;        // staged width, multiple of 4
    const int bh = by1 - by0 + 1;
    const bool use_lds = (long long)bw * bh <= LDS_PX;

    if (use_lds) {
        const int gpr = bw >> 2;                     // 4-pixel groups per staged row
        const int total = gpr * bh;
        for (int g = tid; g < total; g += NT) {
            const int row = g / gpr, gx = g - row * gpr;
            const int sy = by0 + row, sx = bx0a + 4 * gx;
            uint4 px = make_uint4(0u, 0u, 0u, 0u);
            if ((unsigned)sy < (unsigned)a.sh) {
                if (a.src_aligned && sx >= 0 && sx + 3 < a.sw) {
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
                    px.x = load_px_checked<CN>(src, a.sstride, a.sw, a.sh, sx, sy);
                    px.y = load_px_checked<CN>(src, a.sstride, a.sw, a.sh, sx + 1, sy);
                    px.z = load_px_checked<CN>(src, a.sstride, a.sw, a.sh, sx + 2, sy);
                    px.w = load_px_checked<CN>(src, a.sstride, a.sw, a.sh, sx + 3, sy);
                }
            }
            *reinterpret_cast<uint4*>(&tile[row * bw + 4 * gx]) = px;
        }
        __syncthreads();
    }

    const int tx = tid % TXN, ty = tid / TXN;
    const int x = x0 + PX * tx;
    if (x > x1) return;
    int ad[PX], bd[PX];
#pragma unroll
    for (int i = 0; i < PX; i++) col_delta(iv, x + i, ad[i], bd[i]);

#pragma unroll
    for (int r = 0; r < TH / TYN; r++) {
        const int y = y0 + ty + TYN * r;
        if (y > y1) break;
        int X0, Y0;
        row_base(iv, y, X0, Y0);
        uint32_t o[PX];
#pragma unroll
        for (int i = 0; i < PX; i++) {
            const int X = (X0 + ad[i]) >> 5, Y = (Y0 + bd[i]) >> 5;
            const int sx = sat_s16(X >> 5), sy = sat_s16(Y >> 5);
            const int fx = X & 31, fy = Y & 31;
            uint32_t p00, p01, p10, p11;
            if (use_lds) {
                const int idx = (sy - by0) * bw + (sx - bx0a);
                p00 = tile[idx]; p01 = tile[idx + 1];
                p10 = tile[idx + bw]; p11 = tile[idx + bw + 1];
            } else {
                p00 = load_px_checked<CN>(src, a.sstride, a.sw, a.sh, sx, sy);
                p01 = load_px_checked<CN>(src, a.sstride, a.sw, a.sh, sx + 1, sy);
                p10 = load_px_checked<CN>(src, a.sstride, a.sw, a.sh, sx, sy + 1);
                p11 = load_px_checked<CN>(src, a.sstride, a.sw, a.sh, sx + 1, sy + 1);
            }
            o[i] = blend<CN>(p00, p01, p10, p11, fx, fy);
        }
        uint8_t* d = dst + (size_t)y * a.dstride + (size_t)x * CN;
        if (a.dst_aligned && x + PX - 1 <= x1) {
            if (CN == 3) {
                U3 v;
                v.a = o[0] | (o[1] << 24);
                v.b = (o[1] >> 8) | (o[2] << 16);
                v.c = (o[2] >> 16) | (o[3] << 8);
                *reinterpret_cast<U3*>(d) = v;
            } else if (CN == 1) {
                *reinterpret_cast<uint32_t*>(d) = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);
            } else {
                *reinterpret_cast<uint2*>(d) = make_uint2(o[0] | (o[1] << 16), o[2] | (o[3] << 16));
            }
        } else {
#pragma unroll
            for (int i = 0; i < PX; i++) {
                if (x + i > x1) break;
#pragma unroll
                for (int c = 0; c < CN; c++) d[i * CN + c] = (uint8_t)(o[i] >> (8 * c));
            }
        }
    }
}

}  // namespace

int launch_warp_affine(const uint8_t* d_src, size_t sstride, size_t sframe, int sw, int sh,
                       uint8_t* d_dst, size_t dstride, size_t dframe, int dw, int dh, int cn,
                       const float* d_M, int batch, hipStream_t st) {
    if (!d_src || !d_dst || !d_M || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0 || batch <= 0 ||
        batch > 65535 || (cn != 1 && cn != 2 && cn != 3) || sstride < (size_t)sw * cn ||
        dstride < (size_t)dw * cn) {
        set_last_error("warp_affine: invalid argument");
        return VS_ERR_INVALID_ARG;
    }
    WarpArgs a;
    a.src = d_src; a.dst = d_dst;
    a.sstride = sstride; a.sframe = sframe; a.dstride = dstride; a.dframe = dframe;
    a.sw = sw; a.sh = sh; a.dw = dw; a.dh = dh;
    const int galign = cn == 3 ? 4 : cn == 2 ? 8 : 4;
    a.src_aligned = ((uintptr_t)d_src % galign == 0) && (sstride % galign == 0) && (sframe % galign == 0);
    a.dst_aligned = ((uintptr_t)d_dst % galign == 0) && (dstride % galign == 0) && (dframe % galign == 0);
    a.M_dev = d_M;
    dim3 grid((dw + TW - 1) / TW, (dh + TH - 1) / TH, batch);
    if (cn == 3) hipLaunchKernelGGL(warp_affine_kernel<3>, grid, dim3(NT), 0, st, a);
    else if (cn == 1) hipLaunchKernelGGL(warp_affine_kernel<1>, grid, dim3(NT), 0, st, a);
    else hipLaunchKernelGGL(warp_affine_kernel<2>, grid, dim3(NT), 0, st, a);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

// Host-matrix form used by vs_op_warp_affine: matrices travel as kernel
// arguments (MAXB per launch), so no staging buffer or sync is needed.
int launch_warp_affine_hostM(const uint8_t* d_src, size_t sstride, size_t sframe, int sw, int sh,
                             uint8_t* d_dst, size_t dstride, size_t dframe, int dw, int dh, int cn,
                             const float* h_M, int batch, hipStream_t st) {
    if (!d_src || !d_dst || !h_M || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0 || batch <= 0 ||
        (cn != 1 && cn != 2 && cn != 3) || sstride < (size_t)sw * cn || dstride < (size_t)dw * cn) {
        set_last_error("warp_affine: invalid argument");
        return VS_ERR_INVALID_ARG;
    }
    for (int b0 = 0; b0 < batch; b0 += MAXB) {
        const int nb = batch - b0 < MAXB ? batch - b0 : MAXB;
        WarpArgs a;
        a.src = d_src + (size_t)b0 * sframe; a.dst = d_dst + (size_t)b0 * dframe;
        a.sstride = sstride; a.sframe = sframe; a.dstride = dstride; a.dframe = dframe;
        a.sw = sw; a.sh = sh; a.dw = dw; a.dh = dh;
        const int galign = cn == 3 ? 4 : cn == 2 ? 8 : 4;
        a.src_aligned = ((uintptr_t)a.src % galign == 0) && (sstride % galign == 0) && (sframe % galign == 0);
        a.dst_aligned = ((uintptr_t)a.dst % galign == 0) && (dstride % galign == 0) && (dframe % galign == 0);
        a.M_dev = nullptr;
        for (int i = 0; i < nb * 6; i++) a.M_val[i] = h_M[(size_t)b0 * 6 + i];
        for (int i = nb * 6; i < MAXB * 6; i++) a.M_val[i] = 0.f;
        dim3 grid((dw + TW - 1) / TW, (dh + TH - 1) / TH, nb);
        if (cn == 3) hipLaunchKernelGGL(warp_affine_kernel<3>, grid, dim3(NT), 0, st, a);
        else if (cn == 1) hipLaunchKernelGGL(warp_affine_kernel<1>, grid, dim3(NT), 0, st, a);
        else hipLaunchKernelGGL(warp_affine_kernel<2>, grid, dim3(NT), 0, st, a);
        VS_HIP_TRY(hipGetLastError());
    }
    return VS_OK;
}

}  // namespace vsd
