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
//   1. every wave evaluates, in ONE double-precision pass, its share of the tile's
//      coordinate terms (adelta/bdelta of 32 columns, X0/Y0 of 4 rows -> LDS) and, in four
//      spare lanes, the terms of the tile corners, from which the source bounding box of
//      the tile follows by v_readlane (the maps are monotone in x and in y);
//   2. the box is staged into LDS with coalesced 12-byte/lane loads, one dword per pixel;
//      tiles whose box lies inside the image take a branch-free version of this;
//   3. fast path (box at most 136 x 28, i.e. rotations up to ~4 degrees at scale ~1): lane L
//      of a 32-lane row blends the pixels L, L+32, L+64, L+96 (neighbouring lanes read
//      neighbouring LDS dwords: no bank conflicts), taps at immediate offsets of one
//      multiply-add address, horizontal lerps with v_dot4_u32_u8 on byte-permuted tap pairs,
//      vertical lerp in 24-bit multiply-adds whose byte 2 is the rounded result; the row is
//      transposed through a per-wave LDS buffer so that each lane stores 12 contiguous bytes;
//   4. other tiles: a generic LDS path (variable pitch) or, when the box does not fit the
//      LDS budget (large rotation / scale), direct global loads - the same arithmetic.
#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "vs_common.h"
#include "warp_tab.h"

namespace vsd {

namespace {

constexpr int TW = WT_TW;    // output tile width  (pixels)
constexpr int TH = WT_TH;    // output tile height (rows)
constexpr int PX = 4;        // consecutive output pixels per lane
constexpr int NT = 256;      // threads per workgroup
constexpr int TXN = TW / PX; // 32 lanes along x
constexpr int TYN = NT / TXN;// 8 lane-rows
constexpr int MAXB = 32;     // frames per launch (frame pointers and, for host matrices, the maps travel as kernel arguments)
// fast path (near-identity maps): the source box of a tile is staged with a FIXED row pitch, so
// the lower taps sit at an immediate offset and the tap address is one multiply-add
constexpr int FDATA = 136;   // staged pixels of a row (128 + tap + shear + 12-byte alignment slack)
constexpr int FPITCH = 136;  // staged row pitch in pixels (= dwords).  (A pitch of 160 - a multiple of the 32 LDS banks, so that the lanes the
                             // map's rotation moves to the next source row keep their banks: what the plane kernels do - costs the
                             // BGR kernel its eighth workgroup per CU: 90.8 instead of 85 us per 32 frames, not kept.)
constexpr int FROWS = 25;    // staged rows (rotations up to ~3.5 degrees at scale ~1)
constexpr int LDS_PX = FPITCH * FROWS;   // 13.3 KiB of staged pixels per workgroup
constexpr int SG = FDATA / 4;            // 34 four-pixel groups per staged row
constexpr int SR = 7;                    // rows per staging pass (34 x 7 = 238 lanes)
constexpr int SPASS = (FROWS + SR - 1) / SR;   // 4 staging passes held in registers (the last one is partial)
constexpr int LUT_STRIDE = 32;           // bytes between the entries of the weight table (index = coordinate & 0x3E0)
constexpr int OBUF = (NT / 64) * 2 * TW; // per wave: two output rows, one dword per pixel

struct WarpCore {                // what the staging and output code needs of a launch (kept in scalar registers)
    size_t sstride, dstride;
    int sw, sh, dw, dh;
    int src_aligned, dst_aligned;
    int border;                  // VS_BORDER_BLACK (constant 0) or VS_BORDER_REPLICATE
};

struct WarpArgs {
    const uint8_t* src;
    uint8_t* dst;
    size_t sframe, dframe;
    WarpCore c;
    const double* Minv_dev;      // inverse maps on the device (minv_stride doubles apart), or nullptr
    double Minv_val[MAXB * 6];   // used when Minv_dev == nullptr
    int minv_stride;             // doubles between the maps of consecutive frames in Minv_dev
    int use_list;                // frames given one by one (srcs/dsts) instead of base + k*frame
    int32_t* tabs;               // coordinate tables of the frames of this launch (warp_tables_kernel), or nullptr
    int tab_stride;              // ints between the tables of consecutive frames
    int tab_row, tab_ad;         // offsets inside a frame's table (TabLayout)
    const uint8_t* srcs[MAXB];
    uint8_t* dsts[MAXB];
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
struct __attribute__((aligned(4))) U4 { uint32_t a, b, c, d; };

// 12 bytes with the non-temporal hint (global_store_dwordx3 ... nt)
__device__ __forceinline__ void store_nt3(uint8_t* p, const U3& v) {
    typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
    __builtin_nontemporal_store(u32x3{v.a, v.b, v.c}, reinterpret_cast<u32x3*>(p));
}

template <int CN>
__device__ __forceinline__ uint4 stage_group(const WarpCore& c, const uint8_t* __restrict__ src, int sx, int sy) {
    uint4 px = make_uint4(0u, 0u, 0u, 0u);
    if ((unsigned)sy < (unsigned)c.sh || c.border == VS_BORDER_REPLICATE) {
        if (c.src_aligned && sx >= 0 && sx + 3 < c.sw && (unsigned)sy < (unsigned)c.sh) {
            const uint8_t* p = src + (size_t)sy * c.sstride + (size_t)sx * CN;
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
            px.x = load_px_checked<CN>(src, c.sstride, c.sw, c.sh, sx, sy, c.border);
            px.y = load_px_checked<CN>(src, c.sstride, c.sw, c.sh, sx + 1, sy, c.border);
            px.z = load_px_checked<CN>(src, c.sstride, c.sw, c.sh, sx + 2, sy, c.border);
            px.w = load_px_checked<CN>(src, c.sstride, c.sw, c.sh, sx + 3, sy, c.border);
        }
    }
    return px;
}

template <int CN, bool USE_LDS>
__device__ __forceinline__ void emit_rows(const WarpCore& c, const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                          const uint32_t* tile, const int* s_ad, const int* s_bd, const int* s_x0,
                                          const int* s_y0, int x0, int y0, int x1, int y1, int bx0a, int by0, int bw, int tid) {
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
                p00[i] = load_px_checked<CN>(src, c.sstride, c.sw, c.sh, sx, sy, c.border);
                p01[i] = load_px_checked<CN>(src, c.sstride, c.sw, c.sh, sx + 1, sy, c.border);
                p10[i] = load_px_checked<CN>(src, c.sstride, c.sw, c.sh, sx, sy + 1, c.border);
                p11[i] = load_px_checked<CN>(src, c.sstride, c.sw, c.sh, sx + 1, sy + 1, c.border);
            }
        }
#pragma unroll
        for (int i = 0; i < PX; i++) o[r][i] = blend<CN>(p00[i], p01[i], p10[i], p11[i], fx[i], fy[i]);
    }
#pragma unroll
    for (int r = 0; r < TH / TYN; r++) {
        const int y = y0 + ty + TYN * r;
        if (y > y1) break;
        uint8_t* d = dst + (size_t)y * c.dstride + (size_t)x * CN;
        if (c.dst_aligned && x + PX - 1 <= x1) {
            if (CN == 3) {
                U3 v;
                v.a = o[r][0] | (o[r][1] << 24);
                v.b = (o[r][1] >> 8) | (o[r][2] << 16);
                v.c = (o[r][2] >> 16) | (o[r][3] << 8);
                store_nt3(d, v);
            } else if (CN == 1) {
                __builtin_nontemporal_store(o[r][0] | (o[r][1] << 8) | (o[r][2] << 16) | (o[r][3] << 24), reinterpret_cast<uint32_t*>(d));
            } else {
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                __builtin_nontemporal_store(u32x2{o[r][0] | (o[r][1] << 16), o[r][2] | (o[r][3] << 16)}, reinterpret_cast<u32x2*>(d));
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

// ---- fast path (BGR8, near-identity maps) ------------------------------------------------------------------
// Staged layout: one dword per source pixel x, E(x) = [B(x), B(x+1), G(x), R(x)].  The horizontal lerp of the B
// channel is then ONE byte dot product on the staged dword itself, and one byte permute of (E(x), E(x+1)) gives
// [G(x), G(x+1), R(x), R(x+1)] for the other two (two permutes per tap pair with a plain [B,G,R,-] layout).
//
// Weight table in LDS, 32 entries (one per 1/32-px fraction f) of 32 bytes: {wlo, whi, W0, W1}:
//   wlo = (32-f) | f << 8, whi = wlo << 16   byte weights of the horizontal lerp (against bytes 0,1 / 2,3)
//   W0 = (32-f) * 2^121, W1 = f * 2^121      float weights of the vertical lerp
// indexed with (coordinate & 0x3E0) - the fraction bits in place - so a pixel's weights cost two 2-cycle ANDs and
// two LDS reads instead of a bit-field extract, two multiply-adds, a shift and a subtraction.
//
// Vertical lerp in fp32.  gfx950 issues v_fma_f32 / v_add_f32 at twice the rate of the integer multiply-adds.  The
// horizontal sums t, b <= 8160 are used as they are: an integer bit pattern read as a float is the denormal
// t * 2^-149 (denormals run at full rate), so
//   u = fma(t, W0, 2^-29)       = (t*(32-f)     + 1/2) * 2^-28     exact (< 2^20 units of 2^-29)
//   v = fma(b, W1, u)           = (S            + 1/2) * 2^-28     S = t*(32-f) + b*f, exact
//   m = v + 32.0f               rounds to multiples of 2^-18 = 1024 * 2^-28: RNE((S + 1/2) / 1024), never a tie,
//                               = floor((S + 512) / 1024) = cv::warpAffine's (sum + 2^14) >> 15 of the 4-tap form
// and the 8-bit result is the low byte of m's bit pattern (checked exhaustively: scratch/denorm_probe.hip).
struct __attribute__((aligned(8))) LutX { uint32_t wlo, whi; };
struct __attribute__((aligned(8))) LutY { float w0, w1; };

__device__ __forceinline__ float vlerp(uint32_t t, uint32_t b, LutY w) {
    const float u = __builtin_fmaf(__uint_as_float(t), w.w0, 0x1p-29f);
    const float v = __builtin_fmaf(__uint_as_float(b), w.w1, u);
    return v + 32.0f;
}

__device__ __forceinline__ uint32_t blend3_fast(uint32_t e00, uint32_t e01, uint32_t e10, uint32_t e11, LutX wx, LutY wy) {
    const uint32_t q0 = __builtin_amdgcn_perm(e01, e00, 0x07030602u);   // [G00 G01 R00 R01]
    const uint32_t q1 = __builtin_amdgcn_perm(e11, e10, 0x07030602u);
    const float mb = vlerp(__builtin_amdgcn_udot4(e00, wx.wlo, 0u, false), __builtin_amdgcn_udot4(e10, wx.wlo, 0u, false), wy);
    const float mg = vlerp(__builtin_amdgcn_udot4(q0, wx.wlo, 0u, false), __builtin_amdgcn_udot4(q1, wx.wlo, 0u, false), wy);
    const float mr = vlerp(__builtin_amdgcn_udot4(q0, wx.whi, 0u, false), __builtin_amdgcn_udot4(q1, wx.whi, 0u, false), wy);
    const uint32_t bg = __builtin_amdgcn_perm(__float_as_uint(mg), __float_as_uint(mb), 0x0C0C0400u);   // (B, G, 0, 0)
    return __builtin_amdgcn_perm(__float_as_uint(mr), bg, 0x0C040100u);                                  // (B, G, R, 0)
}

// Staged dword of pixel x from the pixel itself and its right neighbour, both as [B, G, R, -].
__device__ __forceinline__ uint32_t pair_b(uint32_t cur, uint32_t next) { return __builtin_amdgcn_perm(next, cur, 0x02010400u); }

// Fast-path output (BGR8).  Lane L of a 32-lane row handles the pixels L, L+32, L+64, L+96 of its row, so that
// neighbouring lanes read neighbouring LDS dwords (no bank conflicts; with 4 consecutive pixels per lane the taps of a
// wave fall on 8 of the 32 banks).  The results are transposed through a small per-wave LDS buffer so that every lane
// still stores 4 consecutive pixels = 12 contiguous bytes.  `base` = -(by0*FPITCH + bx0a).
__device__ __forceinline__ void emit_fast(const WarpCore& c, uint8_t* __restrict__ dst, const uint32_t* tile,
                                          uint32_t* obuf, const uint8_t* lut, const int* s_ad, const int* s_bd, const int* s_x0,
                                          const int* s_y0, int x0, int y0, int x1, int y1, int base, int tid) {
    const int L = tid & 31, ty = tid >> 5;
    uint32_t* wb = obuf + (tid >> 6) * (2 * TW) + ((tid >> 5) & 1) * TW;
    int ad[4], bd[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { ad[i] = s_ad[L + 32 * i]; bd[i] = s_bd[L + 32 * i]; }
    // full tile, aligned rows: the stores need no per-lane checks (tile-uniform)
    const bool whole = c.dst_aligned && x1 - x0 == TW - 1 && y1 - y0 == TH - 1;
    const uint32_t dstride32 = (uint32_t)c.dstride;
    uint8_t* const dtile = dst + (size_t)y0 * c.dstride + (size_t)x0 * 3;      // wave-uniform base, 32-bit lane offsets
#pragma unroll
    for (int r = 0; r < TH / TYN; r++) {
        const int yl = ty + TYN * r;
        const int X0 = s_x0[yl], Y0 = s_y0[yl];
        uint32_t e00[4], e01[4], e10[4], e11[4];
        LutX wx[4];
        LutY wy[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int SX = X0 + ad[i], SY = Y0 + bd[i];              // 1/1024 px
            // byte address of the upper-left tap: one multiply-add and one shift-add
            int rowb = __mul24(SY >> 10, FPITCH * 4) + 4 * base;
            asm("" : "+v"(rowb));
            int sxi = SX >> 10;
            asm("" : "+v"(sxi));
            const uint32_t* t = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(tile) + ((sxi << 2) + rowb));
            e00[i] = t[0]; e01[i] = t[1];
            e10[i] = t[FPITCH]; e11[i] = t[FPITCH + 1];
            wx[i] = *reinterpret_cast<const LutX*>(lut + (SX & 0x3E0));
            wy[i] = *reinterpret_cast<const LutY*>(lut + 8 + (SY & 0x3E0));
        }
#pragma unroll
        for (int i = 0; i < 4; i++) wb[L + 32 * i] = blend3_fast(e00[i], e01[i], e10[i], e11[i], wx[i], wy[i]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint4 q = *reinterpret_cast<const uint4*>(&wb[4 * L]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (whole) {
            U3 v;
            v.a = __builtin_amdgcn_perm(q.y, q.x, 0x04020100u);
            v.b = __builtin_amdgcn_perm(q.z, q.y, 0x05040201u);
            v.c = __builtin_amdgcn_perm(q.w, q.z, 0x06050402u);
            // the result is not read again by this launch: streaming stores (nt) leave the caches to the source rows
            // (measured: 96.1 -> 88.5 us per 32-frame launch; sc1 / sc0 sc1 stores and nt LOADS did not help)
            store_nt3(dtile + (__umul24((uint32_t)yl, dstride32) + 12u * L), v);
            continue;
        }
        const int y = y0 + yl, x = x0 + 4 * L;
        if (y <= y1 && x <= x1) {
            uint8_t* d = dst + (size_t)y * c.dstride + (size_t)x * 3;
            if (c.dst_aligned && x + 3 <= x1) {
                U3 v;
                v.a = __builtin_amdgcn_perm(q.y, q.x, 0x04020100u);
                v.b = __builtin_amdgcn_perm(q.z, q.y, 0x05040201u);
                v.c = __builtin_amdgcn_perm(q.w, q.z, 0x06050402u);
                store_nt3(d, v);
            } else {
                const uint32_t o[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (x + i > x1) break;
                    d[3 * i] = (uint8_t)o[i]; d[3 * i + 1] = (uint8_t)(o[i] >> 8); d[3 * i + 2] = (uint8_t)(o[i] >> 16);
                }
            }
        }
    }
}

// hal::warpAffine / WarpAffineInvoker coordinate terms in one form: round((p*v + q) * 1024)  (warp_tab.h)
__device__ __forceinline__ int coord_term(double p, double q, double v) { return wt_coord_term(p, q, v); }

// The inverse map of frame bz (wave-uniform: scalar loads or kernel arguments).
__device__ __forceinline__ void load_map(const WarpArgs& a, int bz, double* m) {
    if (a.Minv_dev) {
        const __attribute__((address_space(4))) double* mp =
            (const __attribute__((address_space(4))) double*)(a.Minv_dev + a.minv_stride * bz);
#pragma unroll
        for (int i = 0; i < 6; i++) m[i] = mp[i];
    } else {
#pragma unroll
        for (int i = 0; i < 6; i++) m[i] = a.Minv_val[6 * bz + i];
    }
}

// Coordinate tables of the frames of a launch, once per frame instead of once per tile (layout and arithmetic: warp_tab.h).
// The batch schedule has them built by the kernel that computes the inverse maps (k_ransac.hip, release workgroups); this
// launch serves the standalone operator, deferred warps and steps whose maps come out of a one-kernel tail (Kalman).
constexpr int TAB_COL = WT_COL;

__global__ __launch_bounds__(NT) void warp_tables_kernel(WarpArgs a) {
    const int bz = blockIdx.y;
    double m[6];
    load_map(a, bz, m);
    TabLayout L;
    L.row = a.tab_row; L.ad = a.tab_ad; L.stride = a.tab_stride;
    const int j = blockIdx.x * NT + threadIdx.x;
    if (j < wt_entries(a.c.dw, a.c.dh))
        wt_build_entry(a.tabs + (size_t)bz * a.tab_stride, L, m, a.c.dw, a.c.dh, a.use_list ? a.srcs[bz] : a.src + (size_t)bz * a.sframe,
                       a.use_list ? a.dsts[bz] : a.dst + (size_t)bz * a.dframe, j);
}

// The table blocks of up to NVT_MAX NV12 surfaces, both planes, from inverse maps given on the host, in ONE launch
// (blockIdx.y = surface, blockIdx.z = plane): the rotations of a batch of roll-corrected surfaces.
constexpr int NVT_MAX = 16;
struct NvTabArgs {
    int32_t* tabs;
    int block, chroma;           // ints between the blocks of consecutive surfaces / offset of the chroma table inside a block
    int w, h;
    size_t src_uv, dst_uv;
    const uint8_t* ys[NVT_MAX];
    uint8_t* yd[NVT_MAX];
    double my[NVT_MAX * 6], muv[NVT_MAX * 6];
};

__global__ __launch_bounds__(NT) void warp_tables_nv12_kernel(NvTabArgs a) {
    const int bz = blockIdx.y, plane = blockIdx.z;
    const int dw = plane ? a.w / 2 : a.w, dh = plane ? a.h / 2 : a.h;
    TabLayout L = tab_layout(dw, dh);
    L.stride = a.block;
    double m[6];
#pragma unroll
    for (int i = 0; i < 6; i++) m[i] = plane ? a.muv[6 * bz + i] : a.my[6 * bz + i];
    const int j = blockIdx.x * NT + threadIdx.x;
    if (j < wt_entries(dw, dh))
        wt_build_entry(a.tabs + (size_t)bz * a.block + (plane ? a.chroma : 0), L, m, dw, dh, plane ? a.ys[bz] + a.src_uv : a.ys[bz],
                       plane ? a.yd[bz] + a.dst_uv : a.yd[bz], j);
}

template <int CN, bool TABS>
__global__ __launch_bounds__(NT) void warp_affine_kernel(WarpArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t tile[LDS_PX];
    __shared__ __attribute__((aligned(16))) uint32_t obuf[OBUF];
    __shared__ int s_ad[TW], s_bd[TW], s_x0[TH], s_y0[TH];
    __shared__ __attribute__((aligned(16))) uint8_t lut[CN == 3 ? 32 * LUT_STRIDE : 16];
    const int bz = blockIdx.z;
    const uint8_t* __restrict__ src = a.use_list ? a.srcs[bz] : a.src + (size_t)bz * a.sframe;
    uint8_t* __restrict__ dst = a.use_list ? a.dsts[bz] : a.dst + (size_t)bz * a.dframe;
    const int tid = threadIdx.x, wave = tid >> 6;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int x1 = min(x0 + TW, a.c.dw) - 1, y1 = min(y0 + TH, a.c.dh) - 1;
    if (CN == 3 && tid >= NT - 32) {      // weight table of the fast path (see blend3_fast): the last 32 lanes of wave 3
        const uint32_t f = tid - (NT - 32);
        const uint32_t wlo = (32u - f) | (f << 8);
        *reinterpret_cast<uint4*>(lut + f * LUT_STRIDE) =
            make_uint4(wlo, wlo << 16, __float_as_uint((float)(32u - f) * 0x1p121f), __float_as_uint((float)f * 0x1p121f));
    }

    // ---- 1. coordinate terms of the tile (128 columns: adelta, bdelta; 16 rows: X0, Y0) -> LDS.  Columns and rows
    // past the image repeat the last one, so every lane of a partial tile stays inside the box.  The maps are
    // monotone in x and in y separately, so the source bounding box of the tile follows from the terms of its
    // first/last column and row.
    int ad0, ad1, bd0, bd1, Xa, Xb, Ya, Yb;
    int tv0 = 0, tv1 = 0;
    if (TABS) {
        // from the frame's tables: the corner terms with scalar loads, the tile's share with one vector load per
        // lane; they travel together with the staging loads of step 2 and reach LDS in front of the one barrier
        const size_t tb = (size_t)bz * a.tab_stride;
        const __attribute__((address_space(4))) int32_t* Ts = (const __attribute__((address_space(4))) int32_t*)(a.tabs + tb);
        const __attribute__((address_space(4))) int32_t* Cc = Ts + TAB_COL * blockIdx.x + 4;
        const __attribute__((address_space(4))) int32_t* Cr = Ts + a.tab_row + 4 * blockIdx.y;
        ad0 = Cc[0]; ad1 = Cc[1]; bd0 = Cc[2]; bd1 = Cc[3];
        Xa = Cr[0]; Xb = Cr[1]; Ya = Cr[2]; Yb = Cr[3];
        const int32_t* Tg = a.tabs + tb + a.tab_ad;
        if (wave < 2) {
            const int c = min(x0 + tid, x1);
            tv0 = Tg[c]; tv1 = Tg[a.c.dw + c];
        } else if (tid < 2 * 64 + TH) {
            const int r = min(y0 + (tid - 128), y1);
            tv0 = Tg[2 * a.c.dw + r]; tv1 = Tg[2 * a.c.dw + a.c.dh + r];
        }
    } else {
        // no tables (single launches of the image operators): waves 0 and 1 evaluate the columns, 16 lanes of
        // wave 2 the rows, in double; the corner terms come back through LDS
        double m[6];
        load_map(a, bz, m);
        if (wave < 2) {
            const double dv = (double)min(x0 + tid, x1);
            s_ad[tid] = coord_term(m[0], 0.0, dv);
            s_bd[tid] = coord_term(m[3], 0.0, dv);
        } else if (tid < 2 * 64 + TH) {
            const double dv = (double)min(y0 + (tid - 128), y1);
            s_x0[tid - 128] = coord_term(m[1], m[2], dv) + 16;
            s_y0[tid - 128] = coord_term(m[4], m[5], dv) + 16;
        }
        __syncthreads();
        const int cl = x1 - x0, rl = y1 - y0;
        ad0 = __builtin_amdgcn_readfirstlane(s_ad[0]); ad1 = __builtin_amdgcn_readfirstlane(s_ad[cl]);
        bd0 = __builtin_amdgcn_readfirstlane(s_bd[0]); bd1 = __builtin_amdgcn_readfirstlane(s_bd[cl]);
        Xa = __builtin_amdgcn_readfirstlane(s_x0[0]); Xb = __builtin_amdgcn_readfirstlane(s_x0[rl]);
        Ya = __builtin_amdgcn_readfirstlane(s_y0[0]); Yb = __builtin_amdgcn_readfirstlane(s_y0[rl]);
    }
    int bx0, bx1, by0, by1;
    bool saturated;
    {
        const int sx00 = (Xa + ad0) >> 10, sx01 = (Xa + ad1) >> 10, sx10 = (Xb + ad0) >> 10, sx11 = (Xb + ad1) >> 10;
        const int sy00 = (Ya + bd0) >> 10, sy01 = (Ya + bd1) >> 10, sy10 = (Yb + bd0) >> 10, sy11 = (Yb + bd1) >> 10;
        const int rx0 = min(min(sx00, sx01), min(sx10, sx11)), rx1 = max(max(sx00, sx01), max(sx10, sx11));
        const int ry0 = min(min(sy00, sy01), min(sy10, sy11)), ry1 = max(max(sy00, sy01), max(sy10, sy11));
        saturated = rx0 < -32768 || ry0 < -32768 || rx1 > 32767 || ry1 > 32767;   // saturate_cast<short> acts
        bx0 = max(rx0, -32768); bx1 = min(rx1, 32767) + 1;                        // (values past the other end
        by0 = max(ry0, -32768); by1 = min(ry1, 32767) + 1;                        //  only occur when saturated)
    }
    const int bx0a = bx0 & ~3;                       // 4-pixel (12-byte) aligned start
    const int bw = (bx1 - bx0a + 1 + 3) & ~3;        // staged width, multiple of 4
    const int bh = by1 - by0 + 1;
    const bool fast = CN == 3 && !saturated && bw <= FDATA && bh <= FROWS;
    const bool use_lds = fast || (!saturated && (long long)bw * bh <= LDS_PX);
    // (the last group of a row loads 4 bytes past the box: inside the frame unless the box ends with the frame)
    const bool interior = fast && a.c.src_aligned && bx0a >= 0 && bx0a + bw <= a.c.sw && by0 >= 0 && by0 + bh <= a.c.sh &&
                          (bx0a + bw + 2 <= a.c.sw || by0 + bh < a.c.sh);

    // ---- 2. staging
    if (interior) {
        // branch-free: 4 passes of 7 rows x 34 groups; rows past the box repeat its last row.  A lane loads 16 bytes
        // = its 4 pixels and the B of the next one (the staged dword of a pixel carries its right neighbour's B)
        const int ly = tid / SG, lx = tid - ly * SG;
        if (tid < SG * SR && 4 * lx < bw) {
            // wave-uniform base + 32-bit lane offset (the box spans < 2^24 bytes of rows)
            const uint8_t* box = src + (size_t)bx0a * 3 + (size_t)by0 * a.c.sstride;
            const uint32_t stride32 = (uint32_t)a.c.sstride, col = 12u * lx;
            U4 d[SPASS];
#pragma unroll
            for (int k = 0; k < SPASS; k++)
                if (SR * k < bh)       // tile-uniform: small rotations need 3 passes (21 rows)
                    d[k] = *reinterpret_cast<const U4*>(box + (__umul24((uint32_t)min(ly + SR * k, bh - 1), stride32) + col));
#pragma unroll
            for (int k = 0; k < SPASS; k++) {
                if (SR * k < bh && (SR * (k + 1) <= FROWS || ly + SR * k < FROWS))
                    *reinterpret_cast<uint4*>(&tile[(ly + SR * k) * FPITCH + 4 * lx]) =
                        make_uint4(__builtin_amdgcn_perm(d[k].a, d[k].a, 0x02010300u),      // bytes 0,3,1,2
                                   __builtin_amdgcn_perm(d[k].b, d[k].a, 0x05040603u),      // bytes 3,6,4,5
                                   __builtin_amdgcn_perm(d[k].c, d[k].b, 0x04030502u),      // bytes 6,9,7,8
                                   __builtin_amdgcn_perm(d[k].d, d[k].c, 0x03020401u));     // bytes 9,12,10,11
            }
        }
    } else if (use_lds) {
        const int pitch = fast ? FPITCH : bw;
        const int gpr = bw >> 2;
        const int total = gpr * bh;
        for (int g = tid; g < total; g += NT) {
            const int row = g / gpr, gx = g - row * gpr;
            uint4 px = stage_group<CN>(a.c, src, bx0a + 4 * gx, by0 + row);
            if (CN == 3 && fast) {     // the fast path's layout (tiles at the image border)
                const uint32_t nx = load_px_checked<CN>(src, a.c.sstride, a.c.sw, a.c.sh, bx0a + 4 * gx + 4, by0 + row, a.c.border);
                px = make_uint4(pair_b(px.x, px.y), pair_b(px.y, px.z), pair_b(px.z, px.w), pair_b(px.w, nx));
            }
            *reinterpret_cast<uint4*>(&tile[row * pitch + 4 * gx]) = px;
        }
    }
    if (TABS) {
        if (wave < 2) { s_ad[tid] = tv0; s_bd[tid] = tv1; }
        else if (tid < 2 * 64 + TH) { s_x0[tid - 128] = tv0; s_y0[tid - 128] = tv1; }
    }
    __syncthreads();

    // ---- 3. output (the choice is tile-uniform)
    if (CN == 3 && fast) emit_fast(a.c, dst, tile, obuf, lut, s_ad, s_bd, s_x0, s_y0, x0, y0, x1, y1, -(by0 * FPITCH + bx0a), tid);
    else if (use_lds) emit_rows<CN, true>(a.c, src, dst, tile, s_ad, s_bd, s_x0, s_y0, x0, y0, x1, y1, bx0a, by0, bw, tid);
    else emit_rows<CN, false>(a.c, src, dst, tile, s_ad, s_bd, s_x0, s_y0, x0, y0, x1, y1, bx0a, by0, bw, tid);
}

// ---- several small warps of different geometry in ONE launch --------------------------------------------------------------
// AutoZoomCrop's crop-and-scale of a batch of NV12 surfaces (k_azc.hip): 2 x 8 jobs of 640 x 360 / 320 x 180 pixels, every one
// with its own crop rectangle, as one launch instead of sixteen (a launch costs the host 6 - 7 us on this runtime).  The jobs
// travel as kernel arguments; blockIdx.z = job, a job's tiles beyond its own size return at once.  Terms and taps as the
// general kernel's direct path (no staging: a scale map spreads the taps of a tile over a box that no staging area holds).
struct WarpJobsArg { WarpJob j[WARP_JOBS_MAX]; };

__global__ __launch_bounds__(NT) void warp_jobs_kernel(WarpJobsArg a) {
    __shared__ int s_ad[TW], s_bd[TW], s_x0[TH], s_y0[TH];
    const WarpJob& j = a.j[blockIdx.z];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    if (x0 >= j.dw || y0 >= j.dh) return;                 // (workgroup-uniform)
    const int x1 = min(x0 + TW, j.dw) - 1, y1 = min(y0 + TH, j.dh) - 1;
    if (tid < TW) {
        const double dv = (double)min(x0 + tid, x1);
        s_ad[tid] = coord_term(j.m[0], 0.0, dv);
        s_bd[tid] = coord_term(j.m[3], 0.0, dv);
    } else if (tid < TW + TH) {
        const double dv = (double)min(y0 + (tid - TW), y1);
        s_x0[tid - TW] = coord_term(j.m[1], j.m[2], dv) + 16;
        s_y0[tid - TW] = coord_term(j.m[4], j.m[5], dv) + 16;
    }
    __syncthreads();
    WarpCore c;
    c.sstride = j.sstride; c.dstride = j.dstride;
    c.sw = j.sw; c.sh = j.sh; c.dw = j.dw; c.dh = j.dh;
    c.src_aligned = 0;
    const int galign = j.cn == 2 ? 8 : 4;
    c.dst_aligned = ((uintptr_t)j.dst % galign == 0) && (j.dstride % galign == 0);
    c.border = j.border;
    if (j.cn == 1) emit_rows<1, false>(c, j.src, j.dst, nullptr, s_ad, s_bd, s_x0, s_y0, x0, y0, x1, y1, 0, 0, 0, tid);
    else if (j.cn == 2) emit_rows<2, false>(c, j.src, j.dst, nullptr, s_ad, s_bd, s_x0, s_y0, x0, y0, x1, y1, 0, 0, 0, tid);
    else emit_rows<3, false>(c, j.src, j.dst, nullptr, s_ad, s_bd, s_x0, s_y0, x0, y0, x1, y1, 0, 0, 0, tid);
}

// ---- tile building blocks shared by the table kernels ------------------------------------------------------------
constexpr int TABN = 2 * TW + 2 * TH;     // ints of coordinate terms per tile: ad[128] bd[128] x0[16] y0[16]

// Launch constants of a table kernel.
typedef const __attribute__((address_space(1))) int32_t* gtab_t;     // the tables are global memory (not generic: flat loads)
struct TabK {
    WarpCore c;
    gtab_t tabs;
    int tab_stride, tab_row, tab_ad;
    int gx, gy;
};

struct TileIn {            // wave-uniform inputs of a tile
    const uint8_t* src;
    uint8_t* dst;
    int bz, x0, y0, x1, y1;
    int ad0, ad1, bd0, bd1, Xa, Xb, Ya, Yb;
};

struct TileBox { int bx0a, by0, bw, bh; bool fast, use_lds, interior, pre, slowcols; };

__device__ __forceinline__ TileBox tile_box(const WarpCore& c, const TileIn& t) {
    TileBox b;
    const int sx00 = (t.Xa + t.ad0) >> 10, sx01 = (t.Xa + t.ad1) >> 10, sx10 = (t.Xb + t.ad0) >> 10, sx11 = (t.Xb + t.ad1) >> 10;
    const int sy00 = (t.Ya + t.bd0) >> 10, sy01 = (t.Ya + t.bd1) >> 10, sy10 = (t.Yb + t.bd0) >> 10, sy11 = (t.Yb + t.bd1) >> 10;
    const int rx0 = min(min(sx00, sx01), min(sx10, sx11)), rx1 = max(max(sx00, sx01), max(sx10, sx11));
    const int ry0 = min(min(sy00, sy01), min(sy10, sy11)), ry1 = max(max(sy00, sy01), max(sy10, sy11));
    const bool saturated = rx0 < -32768 || ry0 < -32768 || rx1 > 32767 || ry1 > 32767;
    const int bx0 = max(rx0, -32768), bx1 = min(rx1, 32767) + 1;
    b.by0 = max(ry0, -32768);
    const int by1 = min(ry1, 32767) + 1;
    b.bx0a = bx0 & ~3;
    b.bw = (bx1 - b.bx0a + 1 + 3) & ~3;
    b.bh = by1 - b.by0 + 1;
    b.fast = !saturated && b.bw <= FDATA && b.bh <= FROWS;
    b.use_lds = b.fast || (!saturated && (long long)b.bw * b.bh <= LDS_PX);
    b.interior = b.fast && c.src_aligned && b.bx0a >= 0 && b.bx0a + b.bw <= c.sw && b.by0 >= 0 && b.by0 + b.bh <= c.sh &&
                 (b.bx0a + b.bw + 2 <= c.sw || b.by0 + b.bh < c.sh);
    // a box that leaves the image is staged with the same 16-byte loads (issue_tile); only its groups that straddle the
    // left / right image edge (when sw is no multiple of 4, or under BORDER_REPLICATE) are fetched pixel by pixel
    b.pre = b.fast && c.src_aligned && c.sw >= 8;
    b.slowcols = !b.interior && (b.bx0a < 0 || b.bx0a + b.bw + 2 > c.sw) && ((c.sw & 3) != 0 || c.border == VS_BORDER_REPLICATE);
    return b;
}

struct TilePre { U4 d[SPASS]; int tv0, tv1; };     // staging loads of a tile and its share of the tables, in flight

// This lane's share of the tile's coordinate terms: (ad, bd) of a column for the first 128 lanes, (X0, Y0) of a row for
// the next 16.  Columns and rows past the image repeat the last one.
__device__ __forceinline__ void load_terms(const TabK& k, const TileIn& t, int tid, int& tv0, int& tv1) {
    gtab_t Tg = k.tabs + (size_t)t.bz * k.tab_stride + k.tab_ad;
    if (tid < TW) {
        const int c = min(t.x0 + tid, t.x1);
        tv0 = Tg[c]; tv1 = Tg[k.c.dw + c];
    } else if (tid < TW + TH) {
        const int r = min(t.y0 + (tid - TW), t.y1);
        tv0 = Tg[2 * k.c.dw + r]; tv1 = Tg[2 * k.c.dw + k.c.dh + r];
    }
}

// (subx, suby): the origin of the staged box in 1/1024 px, taken off the row terms of a fast tile so that the blend's tap
// address is relative to the staged tile.
__device__ __forceinline__ void store_terms(int* tab, int tid, int tv0, int tv1, int subx = 0, int suby = 0) {
    if (tid < TW) { tab[tid] = tv0; tab[TW + tid] = tv1; }
    else if (tid < TW + TH) { tab[2 * TW + (tid - TW)] = tv0 - subx; tab[2 * TW + TH + (tid - TW)] = tv1 - suby; }
}

// Staging loads of a fast tile.  ONE load instruction per pass for every kind of tile (interior tiles and tiles that
// leave the image run the same straight code, no per-pixel checked loads at the image border): 16 bytes per lane,
// from an address that depends on the lane's group class at the image border
//   0  inside the row                 its own 16 bytes: four pixels and the B of the next one
//   1  the row's last four pixels     the 16 bytes that END with the row: the pixels are dwords b, c, d
//   4  the group left of the image    the row's first 16 bytes: the staged dword of pixel -1 carries the B of pixel 0
//   2  outside (BORDER_CONSTANT)      the row's first 16 bytes (not used)
//   3  straddles the edge             the row's first 16 bytes (not used): fetched pixel by pixel when stored
// and store_tile shifts the registers into place once they have landed.  Interior tiles are class 0 throughout.
__device__ __forceinline__ int edge_class(const WarpCore& c, int cg, bool row_ok) {
    if (!row_ok) return 2;
    if (cg >= 0 && cg + 6 <= c.sw) return 0;
    if (cg >= 0 && cg + 4 == c.sw) return 1;
    if (c.border != VS_BORDER_REPLICATE) {
        if (cg == -4) return 4;
        if (cg + 3 < 0 || cg >= c.sw) return 2;
    }
    return 3;
}

__device__ __forceinline__ void issue_tile(const TabK& k, const TileIn& t, const TileBox& b, int tid, TilePre& pf) {
    const int ly = tid / SG, lx = tid - ly * SG;
    if (tid < SG * SR && 4 * lx < b.bw) {
        const int cg = b.bx0a + 4 * lx;
        const uint32_t stride32 = (uint32_t)k.c.sstride;
        uint32_t off[SPASS];
        if (b.interior) {
            const uint32_t col = 3u * (uint32_t)cg;
#pragma unroll
            for (int q = 0; q < SPASS; q++) off[q] = __umul24((uint32_t)(b.by0 + min(ly + SR * q, b.bh - 1)), stride32) + col;
        } else {
#pragma unroll
            for (int q = 0; q < SPASS; q++) {
                const int r = b.by0 + min(ly + SR * q, b.bh - 1);
                const int rc = min(max(r, 0), k.c.sh - 1);                       // BORDER_REPLICATE: the nearest row
                const int cls = edge_class(k.c, cg, r == rc || k.c.border == VS_BORDER_REPLICATE);
                off[q] = __umul24((uint32_t)rc, stride32) + (cls == 0 ? 3u * (uint32_t)cg : (cls == 1 ? 3u * (uint32_t)cg - 4u : 0u));
            }
        }
#pragma unroll
        for (int q = 0; q < SPASS; q++)
            if (SR * q < b.bh) pf.d[q] = *reinterpret_cast<const U4*>(t.src + off[q]);
    }
}

__device__ __forceinline__ void store_tile(const TabK& k, const TileIn& t, const TileBox& b, int tid, const TilePre& pf, uint32_t* tile, int* tab) {
    const int ly = tid / SG, lx = tid - ly * SG;
    if (tid < SG * SR && 4 * lx < b.bw) {
        const int cg = b.bx0a + 4 * lx;
#pragma unroll
        for (int q = 0; q < SPASS; q++) {
            if (SR * q < b.bh && (SR * (q + 1) <= FROWS || ly + SR * q < FROWS)) {
                U4 d = pf.d[q];
                int cls = 0, r = 0;
                if (!b.interior) {
                    r = b.by0 + min(ly + SR * q, b.bh - 1);
                    const int rc = min(max(r, 0), k.c.sh - 1);
                    cls = edge_class(k.c, cg, r == rc || k.c.border == VS_BORDER_REPLICATE);
                    if (cls == 1) {                  // loaded 4 bytes early: the pixels are b, c, d; right of the last one is the border
                        d.a = d.b; d.b = d.c; d.c = d.d;
                        d.d = k.c.border == VS_BORDER_REPLICATE ? (d.c >> 8) & 0xFFu : 0u;
                    } else if (cls == 4) {           // pixels -4 .. -1: zeros, and the B of pixel 0
                        d.d = d.a; d.a = 0u; d.b = 0u; d.c = 0u;
                    } else if (cls != 0) { d.a = 0u; d.b = 0u; d.c = 0u; d.d = 0u; }
                }
                uint4 e = make_uint4(__builtin_amdgcn_perm(d.a, d.a, 0x02010300u), __builtin_amdgcn_perm(d.b, d.a, 0x05040603u),
                                     __builtin_amdgcn_perm(d.c, d.b, 0x04030502u), __builtin_amdgcn_perm(d.d, d.c, 0x03020401u));
                if (b.slowcols && cls == 3) {
                    uint32_t px[5];
#pragma unroll
                    for (int i = 0; i < 5; i++) px[i] = load_px_checked<3>(t.src, k.c.sstride, k.c.sw, k.c.sh, cg + i, r, k.c.border);
                    e = make_uint4(pair_b(px[0], px[1]), pair_b(px[1], px[2]), pair_b(px[2], px[3]), pair_b(px[3], px[4]));
                }
                *reinterpret_cast<uint4*>(&tile[(ly + SR * q) * FPITCH + 4 * lx]) = e;
            }
        }
    }
    store_terms(tab, tid, pf.tv0, pf.tv1, b.bx0a << 10, b.by0 << 10);
}

// XCD-aware tile order.  The hardware hands consecutive workgroups to the 8 XCDs in turn (linear id mod 8), and every XCD has
// an L2 of its own: with the plain (x, y, frame) order the eight neighbours of a tile row sit on eight different XCDs, and every
// 128-byte line that two tiles share (the box of a tile is 144 / 408 bytes wide at an arbitrary 4- / 12-byte offset, plus the halo
// rows above and below) is fetched once per XCD - read traffic of 2.28 x (planes) and 1.60 x (BGR) the algorithmic bytes at the
// L2's memory side (FETCH_SIZE, profiles/r03_a).  Here XCD c takes the c-th contiguous eighth of the launch's tiles in
// (frame, tile row, tile column) order instead: neighbours in x and y meet in one L2.  flags bit 8 selects it.
struct TileId { int x, y, z; };
// (gx, gy follow from the frame size, the frame count travels in flags bits 16.. - gridDim would be one more scalar load in
// front of the table loads - and the two divisions are multiplications by reciprocals the host provides: mgx = ceil(2^32 / gx),
// exact while seq * gx < 2^32, which the launcher checks before it sets the flag)
__device__ __forceinline__ TileId tile_id(uint32_t flags, uint32_t gx, uint32_t gy, uint32_t mgx, uint32_t mgy) {
    TileId t;
    if (!(flags & 0x100u)) { t.x = blockIdx.x; t.y = blockIdx.y; t.z = blockIdx.z; return t; }
    const uint32_t n = gx * gy * (flags >> 16);
    const uint32_t lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const uint32_t c = lin & 7u, k = lin >> 3, q = n >> 3, r = n & 7u;
    const uint32_t seq = c * q + (c < r ? c : r) + k;
    const uint32_t row = gx == 1 ? seq : __umulhi(seq, mgx);       // (ceil(2^32 / 1) does not fit 32 bits)
    t.x = (int)(seq - row * gx);
    t.z = (int)(gy == 1 ? row : __umulhi(row, mgy));
    t.y = (int)(row - (uint32_t)t.z * gy);
    return t;
}

// ---- one tile per workgroup, everything the prologue needs in 10 dwords of kernel arguments -------------------
// gfx950 preloads the leading kernel arguments (up to 14 dwords) into scalar registers when the wave is launched
// (k_warp.hip is built with -mllvm -amdgpu-kernarg-preload-count=16), so the two scalar loads of the tile's inputs
// (frame pointers + column terms, row terms) go out with the first instructions: ONE round trip in front of the staging
// loads.  (Measured with s_memtime stamps: an argument that is not preloaded, or an input that is fetched by a third
// load later on, each cost another ~1000 cycles of a workgroup's ~9000.)  Staging through issue_tile / store_tile: tiles that leave the image take the same
// 16-byte loads as interior ones.
__global__ __launch_bounds__(NT, 8) void warp_tab_kernel(gtab_t tabs, int tab_stride, int tab_row, int tab_ad, uint32_t sstride, uint32_t dstride,
                                                      uint32_t swh, uint32_t dwh, uint32_t flags, uint32_t mgx, uint32_t mgy) {
    __shared__ __attribute__((aligned(16))) uint32_t tile[LDS_PX];
    __shared__ __attribute__((aligned(16))) uint32_t obuf[OBUF];
    __shared__ int s_tab[TABN];
    __shared__ __attribute__((aligned(16))) uint8_t lut[32 * LUT_STRIDE];
    typedef const __attribute__((address_space(4))) int32_t* cptr;
    typedef __attribute__((address_space(1))) uint8_t* gptr;
    const int tid = threadIdx.x;
    TabK k;
    k.c.sstride = sstride; k.c.dstride = dstride;
    k.c.sw = swh & 0xFFFFu; k.c.sh = swh >> 16; k.c.dw = dwh & 0xFFFFu; k.c.dh = dwh >> 16;
    k.c.src_aligned = flags & 1u; k.c.dst_aligned = (flags >> 1) & 1u; k.c.border = (flags >> 2) & 7u;
    k.tabs = tabs; k.tab_stride = tab_stride; k.tab_row = tab_row; k.tab_ad = tab_ad;
    k.gx = (k.c.dw + TW - 1) / TW; k.gy = (k.c.dh + TH - 1) / TH;
    TileIn cur;
    const TileId tq = tile_id(flags, (uint32_t)(k.c.dw + TW - 1) / TW, (uint32_t)(k.c.dh + TH - 1) / TH, mgx, mgy);
    {
        cptr Ts = (cptr)(const int32_t*)(tabs + (size_t)tq.z * tab_stride);
        // one 32-byte and one 16-byte scalar load, issued together (as separate loads the compiler moves the frame
        // pointers down to their first use: a third round trip in front of the staging loads)
        typedef int32_t i32x8 __attribute__((ext_vector_type(8)));
        typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
        const i32x8 cc = *(const __attribute__((address_space(4))) i32x8*)(Ts + TAB_COL * tq.x);
        const i32x4 cr = *(const __attribute__((address_space(4))) i32x4*)(Ts + tab_row + 4 * tq.y);
        cur.src = (const uint8_t*)(gptr)((unsigned long long)(uint32_t)cc[0] | (unsigned long long)(uint32_t)cc[1] << 32);
        cur.dst = (uint8_t*)(gptr)((unsigned long long)(uint32_t)cc[2] | (unsigned long long)(uint32_t)cc[3] << 32);
        cur.ad0 = cc[4]; cur.ad1 = cc[5]; cur.bd0 = cc[6]; cur.bd1 = cc[7];
        cur.Xa = cr[0]; cur.Xb = cr[1]; cur.Ya = cr[2]; cur.Yb = cr[3];
        cur.bz = tq.z; cur.x0 = tq.x * TW; cur.y0 = tq.y * TH;
        cur.x1 = min(cur.x0 + TW, k.c.dw) - 1; cur.y1 = min(cur.y0 + TH, k.c.dh) - 1;
    }
    if (tid >= NT - 32) {      // weight table of the fast path (see blend3_fast)
        const uint32_t f = tid - (NT - 32);
        const uint32_t wlo = (32u - f) | (f << 8);
        *reinterpret_cast<uint4*>(lut + f * LUT_STRIDE) =
            make_uint4(wlo, wlo << 16, __float_as_uint((float)(32u - f) * 0x1p121f), __float_as_uint((float)f * 0x1p121f));
    }
    // this lane's share of the tile's coordinate terms does not depend on the box: requested while the scalar loads travel
    TilePre pf;
    pf.tv0 = 0; pf.tv1 = 0;
    load_terms(k, cur, tid, pf.tv0, pf.tv1);
    const TileBox cb = tile_box(k.c, cur);
    if (cb.pre) {
        issue_tile(k, cur, cb, tid, pf);
        store_tile(k, cur, cb, tid, pf, tile, s_tab);
    } else {
        const int tv0 = pf.tv0, tv1 = pf.tv1;
        if (cb.use_lds) {
            const int pitch = cb.fast ? FPITCH : cb.bw;
            const int gpr = cb.bw >> 2;
            const int total = gpr * cb.bh;
            for (int g = tid; g < total; g += NT) {
                const int row = g / gpr, gxx = g - row * gpr;
                uint4 px = stage_group<3>(k.c, cur.src, cb.bx0a + 4 * gxx, cb.by0 + row);
                if (cb.fast) {
                    const uint32_t nx = load_px_checked<3>(cur.src, k.c.sstride, k.c.sw, k.c.sh, cb.bx0a + 4 * gxx + 4, cb.by0 + row, k.c.border);
                    px = make_uint4(pair_b(px.x, px.y), pair_b(px.y, px.z), pair_b(px.z, px.w), pair_b(px.w, nx));
                }
                *reinterpret_cast<uint4*>(&tile[row * pitch + 4 * gxx]) = px;
            }
        }
        store_terms(s_tab, tid, tv0, tv1, cb.fast ? cb.bx0a << 10 : 0, cb.fast ? cb.by0 << 10 : 0);
    }
    __syncthreads();
    if (cb.fast) emit_fast(k.c, cur.dst, tile, obuf, lut, s_tab, s_tab + TW, s_tab + 2 * TW, s_tab + 2 * TW + TH, cur.x0, cur.y0, cur.x1, cur.y1, 0, tid);
    else if (cb.use_lds) emit_rows<3, true>(k.c, cur.src, cur.dst, tile, s_tab, s_tab + TW, s_tab + 2 * TW, s_tab + 2 * TW + TH, cur.x0, cur.y0, cur.x1,
                                            cur.y1, cb.bx0a, cb.by0, cb.bw, tid);
    else emit_rows<3, false>(k.c, cur.src, cur.dst, tile, s_tab, s_tab + TW, s_tab + 2 * TW, s_tab + 2 * TW + TH, cur.x0, cur.y0, cur.x1, cur.y1,
                             cb.bx0a, cb.by0, cb.bw, tid);
}

// ---- one- and two-channel planes of batched launches (NV12: Y and interleaved UV) ------------------------------------
// The BGR table kernel above owes its rate to what a workgroup carries through its latency chain (scalar prologue ->
// staging loads -> barrier -> blend): 6 KB of output.  A 128 x 16 tile of a Y plane is 2 KB, of a UV plane 4 KB, and the general
// kernel stages them as one dword per pixel with per-group checks: 0.22 of the HBM peak at 3840 x 2160.  This kernel gives a
// plane workgroup 8 KB of output - 128 x 64 pixels of one channel, 128 x 32 of two - and stages the source box as BYTES
// (row pitch 144 / 288, at most 73 / 41 rows: 10.5 / 11.8 KB, eight workgroups per CU):
//   staging   16-byte chunks, vector loads for chunks inside the frame, per-byte checked loads (zeros outside:
//             BORDER_CONSTANT) for the few that cross its edge;
//   taps      the two horizontal taps of a pixel are adjacent in the staged row: an 8-byte LDS read at the 4-byte
//             aligned address below them and one v_alignbyte put [p(x), p(x+1)] (or [U0 V0 U1 V1]) into the low bytes of a
//             dword, for the upper and the lower tap row;
//   blend     v_dot4 against ((32-f) | f << 8) (one channel) or ((32-f) | f << 16, and << 8 for V) = the weights as one
//             multiply-add of f; vertical lerp in fp32 on denormals as in the BGR kernel;
//   output    lane L of a 32-lane row takes the consecutive pixels 4L .. 4L+3 (with a byte-wide box neighbouring lanes then
//             read neighbouring LDS dwords) and stores its 4 / 8 bytes.
// It shares the per-frame coordinate tables with the other kernels (the tables know tiles of 16 rows: the corner terms of
// a taller tile come from the records of its first and last 16 rows).  A box that does not fit the staging area (large
// rotations or zooms) takes emit_rows' direct path, 16 rows at a time.
// A 16-byte chunk of a source row that straddles the row's left or right end (byte column xb .. xb + 15, row of `rowbytes`
// bytes at rowp): zeros outside.  All loads unconditional, from clamped addresses, so that they travel together (under
// conditions the compiler waits for each of sixteen byte loads in turn: 16 round trips for a tile at the frame's edge).
__device__ __forceinline__ uint4 load_chunk_straddling(const uint8_t* rowp, int xb, int rowbytes, bool dwords) {
    uint32_t w[4];
    if (dwords) {            // row start and row length multiples of 4: a dword is inside or outside as a whole
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int xq = xb + 4 * q;
            const bool in = xq >= 0 && xq + 4 <= rowbytes;
            const uint32_t v = *reinterpret_cast<const uint32_t*>(rowp + (in ? xq : 0));
            w[q] = in ? v : 0u;
        }
    } else {                 // four bytes at a time (a rolled loop: sixteen loads in flight would cost the callers their registers)
        uint32_t wq[4] = {0u, 0u, 0u, 0u};
#pragma unroll 1
        for (int k = 0; k < 4; k++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int xx = xb + 4 * q + k;
                const uint32_t v = rowp[min(max(xx, 0), rowbytes - 1)];
                wq[q] |= (xx >= 0 && xx < rowbytes ? v : 0u) << (8 * k);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) w[q] = wq[q];
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// The same chunk under BORDER_REPLICATE (remapBilinear's clip()): a byte outside the row is the byte of the same channel of the
// row's first / last pixel.  (Chunks of tiles at the picture's edge only.)
template <int CN>
__device__ __forceinline__ uint4 load_chunk_replicate(const uint8_t* rowp, int xb, int sw) {
    uint32_t wq[4] = {0u, 0u, 0u, 0u};
#pragma unroll 1
    for (int k = 0; k < 4; k++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int xx = xb + 4 * q + k;
            const int px = min(max(CN == 1 ? xx : xx >> 1, 0), sw - 1);
            wq[q] |= (uint32_t)rowp[CN == 1 ? px : 2 * px + (xx & 1)] << (8 * k);
        }
    }
    return make_uint4(wq[0], wq[1], wq[2], wq[3]);
}

template <int CN> struct PlaneCfg {
    static constexpr int THP = 64 / CN;                  // rows of a tile
    static constexpr int DB = 136 * CN + 8 * CN;         // staged bytes of a row (136 pixels + slack for the 8-byte tap read)
    // Row pitch of the staged box: a multiple of 128 bytes = of the 32 LDS banks.  The lanes of a tap read sit on consecutive dwords
    // of a row until the map's rotation moves them a source row down (or up), once per tile row for anything but a pure
    // translation; with a pitch of 144 bytes those lanes land 4 banks to the side, on banks their neighbours use, and every tap
    // read pays a second pass.  With this pitch a lane's bank does not depend on its row.  (scratch/blend_lab.sh: the tap reads
    // were 66 of the kernel's 209 us per 32 4K frames.)
    static constexpr int PB = (DB + 127) / 128 * 128;    // 256 / 384
    static constexpr int ROWS = THP + 9;                 // staged rows (rotations up to ~3.5 degrees)
    static constexpr int CPR = DB / 16;                  // 16-byte chunks per staged row
};

// Output of a staged plane tile: lane L of a 32-lane row takes the four CONSECUTIVE pixels 4L .. 4L+3 of rows ty, ty + 8, ...: with
// the box staged as bytes that puts neighbouring lanes on neighbouring LDS dwords, and the lane's four results are its 4 / 8
// output bytes.  tl: the staged box (row pitch PB), the terms relative to it: ad / bd of the lane's columns, s_x0 / s_y0 of the
// tile's rows.  Written for few instructions: the taps are dword reads (the address is a multiple of 4, not of 8, and an 8-byte LDS
// read off its alignment is some 20 x slower) shifted into place by v_alignbyte, which takes its byte count from the low two
// bits of the tap address as it is; the weights of both lerps come from the table; and for the common tile - whole, aligned -
// the rows are unrolled and the store address is a lane offset computed once plus a base that moves from row to row.
// What the measurements of round 3 say about it (same-box pairs, scratch/ab_lib.sh): with the row pitch
// of PlaneCfg the LDS reads no longer show (a build without them is no faster), the kernel's time does not move between 4
// and 8 waves per SIMD, and the staging loads plus the stores alone take 105 of the luma kernel's 133 us - a copy of the
// same bytes takes 95.
// One output pixel of a staged plane tile -> its value(s) in bytes 0 (Y) / 0, 1 (U, V) of the result.  Vertical lerp as in the BGR
// kernel (vlerp: three float operations of the 2-cycle class on the integer sums read as denormals).  (Measured against it on one
// box, scratch/ab_lib.sh: the integer form - t | b << 16 against (64 (32 - f), 64 f) in one v_dot2_u32_u16 preset to 2^15, result
// in byte 2: one instruction and 4 bytes of LDS weights less per pixel, bit-identical - 204 us against 198: its two
// instructions are of the 4-cycle class, scratch/valu_rate.hip.)
template <int CN>
__device__ __forceinline__ uint32_t plane_blend_px(const uint8_t* tl, const uint8_t* lut, int SX, int SY) {
    typedef PlaneCfg<CN> P;
    int addr;                                                     // byte of the upper-left tap: row * PB + column * CN
    if (P::PB == 256 && CN == 1) {          // (one shift-add; left to itself the compiler makes shift, mask, add of it)
        const int sy = SY >> 10, sx = SX >> 10;
        asm("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(addr) : "v"(sy), "v"(sx));
    } else {
        addr = __mul24(SY >> 10, P::PB) + (SX >> 10) * CN;
    }
    const uint32_t* tp = reinterpret_cast<const uint32_t*>(tl + (addr & ~3));
    const uint32_t top = __builtin_amdgcn_alignbyte(tp[1], tp[0], (uint32_t)addr), bot = __builtin_amdgcn_alignbyte(tp[P::PB / 4 + 1], tp[P::PB / 4], (uint32_t)addr);
    const LutY wy = *reinterpret_cast<const LutY*>(lut + 8 + (SY & 0x3E0));
    const uint32_t wx = *reinterpret_cast<const uint32_t*>(lut + (CN == 1 ? 0 : 16) + (SX & 0x3E0));   // (32 - fx) | fx << 8  /  (32 - fx) | fx << 16
    if (CN == 1) {
        float m = vlerp(__builtin_amdgcn_udot4(top, wx, 0u, false), __builtin_amdgcn_udot4(bot, wx, 0u, false), wy);
        asm("" : "+v"(m));            // (keeps the pixels' float operations apart: the packed f32 forms need moves and are no faster)
        return __float_as_uint(m);
    }
    const uint32_t wv = wx << 8;                                  // wx against bytes 0 and 2 (U0, U1), wv against bytes 1 and 3 (V0, V1)
    float mu = vlerp(__builtin_amdgcn_udot4(top, wx, 0u, false), __builtin_amdgcn_udot4(bot, wx, 0u, false), wy);
    float mv = vlerp(__builtin_amdgcn_udot4(top, wv, 0u, false), __builtin_amdgcn_udot4(bot, wv, 0u, false), wy);
    asm("" : "+v"(mu), "+v"(mv));
    return __builtin_amdgcn_perm(__float_as_uint(mv), __float_as_uint(mu), 0x0C0C0400u);     // (U, V, 0, 0)
}

template <int CN>
__device__ __forceinline__ void plane_store4(uint8_t* dq, const uint32_t (&res)[4]) {       // four pixels = 4 / 8 aligned bytes
    if (CN == 1) {
        const uint32_t lo = __builtin_amdgcn_perm(res[1], res[0], 0x0C0C0400u), hi = __builtin_amdgcn_perm(res[3], res[2], 0x0C0C0400u);   // low bytes
        __builtin_nontemporal_store(__builtin_amdgcn_perm(hi, lo, 0x05040100u), reinterpret_cast<uint32_t*>(dq));
    } else {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        __builtin_nontemporal_store(u32x2{res[0] | (res[1] << 16), res[2] | (res[3] << 16)}, reinterpret_cast<u32x2*>(dq));
    }
}

template <int CN>
__device__ __forceinline__ void plane_blend_rows(const WarpCore& c, const uint8_t* tl, const uint8_t* lut, const int2* s_row,
                                                 const int (&ad)[4], const int (&bd)[4], int L, int ty, uint8_t* dst, uint32_t dstride,
                                                 int x0, int y0, int x1, int y1) {
    typedef PlaneCfg<CN> P;
    constexpr int NR = P::THP / TYN;
    uint8_t* const dtile = dst + (size_t)y0 * dstride + (size_t)x0 * CN;      // wave-uniform base, 32-bit lane offsets
    if (c.dst_aligned && x1 - x0 == TW - 1 && y1 - y0 == P::THP - 1) {         // (tile-uniform)
        const uint32_t voff = __umul24((uint32_t)ty, dstride) + (uint32_t)(4 * CN) * L;
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int2 XY = s_row[ty + TYN * r];
            uint32_t res[4];
#pragma unroll
            for (int i = 0; i < 4; i++) res[i] = plane_blend_px<CN>(tl, lut, XY.x + ad[i], XY.y + bd[i]);
            plane_store4<CN>(dtile + (size_t)(TYN * r) * dstride + voff, res);
        }
        return;
    }
    const int x = x0 + 4 * L;
    for (int r = 0; r < NR; r++) {
        const int yl = ty + TYN * r;
        const int2 XY = s_row[ty + TYN * r];
        const int X0 = XY.x, Y0 = XY.y;
        uint32_t res[4];
#pragma unroll
        for (int i = 0; i < 4; i++) res[i] = plane_blend_px<CN>(tl, lut, X0 + ad[i], Y0 + bd[i]);
        const int y = y0 + yl;
        if (y > y1 || x > x1) continue;
        uint8_t* dp = dst + (size_t)y * c.dstride + (size_t)x * CN;
        if (c.dst_aligned && x + 3 <= x1) { plane_store4<CN>(dp, res); continue; }
        for (int i = 0; i < 4; i++) {
            if (x + i > x1) break;
            for (int k = 0; k < CN; k++) dp[i * CN + k] = (uint8_t)(res[i] >> (8 * k));     // (one channel: the low byte of the float's bits)
        }
    }
}

// A plane tile whose source box does not fit the staging area (large rotations, zooms, saturated coordinates, BORDER_REPLICATE): the
// general path without staging (emit_rows works on 16 rows, terms as arrays: they go where the box would be).  Not inlined: as
// part of the kernel body it costs the common path its eighth wave per SIMD or register spills.
template <int CN>
__device__ __attribute__((noinline)) void plane_direct_tile(WarpCore c, const uint8_t* src, uint8_t* dst, gtab_t Tg, int* s_tab, int x0, int y0, int x1, int y1,
                                                            int bx0a, int by0, int bw, int tid) {
    typedef PlaneCfg<CN> P;
    if (tid < TW) {           // ad[128] bd[128] x0[THP] y0[THP]
        const int cx = min(x0 + tid, x1);
        s_tab[tid] = Tg[cx]; s_tab[TW + tid] = Tg[c.dw + cx];
    } else if (tid < TW + P::THP) {
        const int r = min(y0 + (tid - TW), y1);
        s_tab[2 * TW + (tid - TW)] = Tg[2 * c.dw + r]; s_tab[2 * TW + P::THP + (tid - TW)] = Tg[2 * c.dw + c.dh + r];
    }
    __syncthreads();
    for (int j = 0; y0 + TH * j <= y1; j++)
        emit_rows<CN, false>(c, src, dst, nullptr, s_tab, s_tab + TW, s_tab + 2 * TW + TH * j, s_tab + 2 * TW + P::THP + TH * j, x0, y0 + TH * j, x1,
                             min(y0 + TH * j + TH - 1, y1), bx0a, by0, bw, tid);
}

// One tile of a plane: tile (tx, ty) of the frame whose plane table starts at Ts (column records) / Tg (the per-column and
// per-row terms).  tile / lut / s_row: the workgroup's LDS.
// BORDER: what the staged box holds where it leaves the picture - zeros (cv::warpAffine BORDER_CONSTANT: the stabilizer's warp) or the
// nearest picture pixel (BORDER_REPLICATE: the roll stage's rotation); a launch whose border is the other one takes the direct path.
template <int CN, int BORDER = VS_BORDER_BLACK>
__device__ __forceinline__ void plane_tile(const WarpCore& c, const __attribute__((address_space(4))) int32_t* Ts, gtab_t Tg, int tab_row, int tx, int tyl,
                                           uint8_t* tile, uint8_t* lut, int2* s_row, int tid) {
    typedef PlaneCfg<CN> P;
    typedef __attribute__((address_space(1))) uint8_t* gptr;
    const int L = tid & 31, ty = tid >> 5;
    int ad[4], bd[4];
    const int x0 = tx * TW, y0 = tyl * P::THP;
    const int x1 = min(x0 + TW, c.dw) - 1, y1 = min(y0 + P::THP, c.dh) - 1;
    const uint8_t* src;
    uint8_t* dst;
    int ad0, ad1, bd0, bd1, Xa, Xb, Ya, Yb;
    {
        typedef int32_t i32x8 __attribute__((ext_vector_type(8)));
        typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
        const i32x8 cc = *(const __attribute__((address_space(4))) i32x8*)(Ts + TAB_COL * tx);
        const i32x4 r0 = *(const __attribute__((address_space(4))) i32x4*)(Ts + tab_row + 4 * (y0 / TH));
        const i32x4 r1 = *(const __attribute__((address_space(4))) i32x4*)(Ts + tab_row + 4 * (y1 / TH));
        src = (const uint8_t*)(gptr)((unsigned long long)(uint32_t)cc[0] | (unsigned long long)(uint32_t)cc[1] << 32);
        dst = (uint8_t*)(gptr)((unsigned long long)(uint32_t)cc[2] | (unsigned long long)(uint32_t)cc[3] << 32);
        ad0 = cc[4]; ad1 = cc[5]; bd0 = cc[6]; bd1 = cc[7];
        Xa = r0[0]; Ya = r0[2]; Xb = r1[1]; Yb = r1[3];
    }
    if (tid >= NT - 32) {      // vertical weights (the table of the BGR kernel: only W0 / W1 are read here)
        const uint32_t f = tid - (NT - 32);
        const uint32_t wlo = (32u - f) | (f << 8);
        *reinterpret_cast<uint4*>(lut + f * LUT_STRIDE) =
            make_uint4(wlo, wlo << 16, __float_as_uint((float)(32u - f) * 0x1p121f), __float_as_uint((float)f * 0x1p121f));
        *reinterpret_cast<uint32_t*>(lut + f * LUT_STRIDE + 16) = (32u - f) | (f << 16);
    }
    // source box of the tile (the maps are monotone in x and in y separately)
    int bx0a, by0, bw, bh;
    bool fit;
    {
        const int sx00 = (Xa + ad0) >> 10, sx01 = (Xa + ad1) >> 10, sx10 = (Xb + ad0) >> 10, sx11 = (Xb + ad1) >> 10;
        const int sy00 = (Ya + bd0) >> 10, sy01 = (Ya + bd1) >> 10, sy10 = (Yb + bd0) >> 10, sy11 = (Yb + bd1) >> 10;
        const int rx0 = min(min(sx00, sx01), min(sx10, sx11)), rx1 = max(max(sx00, sx01), max(sx10, sx11));
        const int ry0 = min(min(sy00, sy01), min(sy10, sy11)), ry1 = max(max(sy00, sy01), max(sy10, sy11));
        const bool saturated = rx0 < -32768 || ry0 < -32768 || rx1 > 32767 || ry1 > 32767;
        const int bx0 = max(rx0, -32768), bx1 = min(rx1, 32767) + 1;
        by0 = max(ry0, -32768);
        const int by1 = min(ry1, 32767) + 1;
        bx0a = bx0 & ~3;
        bw = bx1 - bx0a + 1;
        bh = by1 - by0 + 1;
        fit = !saturated && bw <= 136 && bh <= P::ROWS && c.border == BORDER;
    }
    if (fit) {
        // ---- staging: chunks of 16 bytes, (row, chunk) = (i / CPR, i % CPR); all loads of a lane first, then its stores
        constexpr int NCH = (P::ROWS * P::CPR + NT - 1) / NT;
        const int total = bh * P::CPR;
        const long long rowbytes = (long long)c.sw * CN;
        uint4 d[NCH];
#pragma unroll
        for (int k = 0; k < NCH; k++) {
            const int i = tid + NT * k;
            d[k] = make_uint4(0u, 0u, 0u, 0u);
            if (i < total) {
                const int r = i / P::CPR, ch = i - r * P::CPR;
                const int y = by0 + r;
                const long long xb = (long long)bx0a * CN + 16 * ch;          // byte column of the chunk in the source row
                if (BORDER == VS_BORDER_REPLICATE) {
                    const uint8_t* row = src + (size_t)min(max(y, 0), c.sh - 1) * c.sstride;
                    if (xb >= 0 && xb + 16 <= rowbytes && c.src_aligned) d[k] = *reinterpret_cast<const uint4*>(row + xb);
                    else d[k] = load_chunk_replicate<CN>(row, (int)max(min(xb, (long long)rowbytes + 64), -64ll), c.sw);
                } else if ((unsigned)y < (unsigned)c.sh) {
                    const uint8_t* row = src + (size_t)y * c.sstride;
                    if (xb >= 0 && xb + 16 <= rowbytes && c.src_aligned) {
                        d[k] = *reinterpret_cast<const uint4*>(row + xb);       // 4-byte aligned: bx0a is a multiple of 4 pixels
                    } else if (xb + 16 > 0 && xb < rowbytes) {
                        d[k] = load_chunk_straddling(row, (int)xb, (int)rowbytes, c.src_aligned && (rowbytes & 3) == 0);
                    }
                }
            }
        }
        // the terms (behind the box's loads): (ad, bd) of the lane's four columns straight into its registers, (X0, Y0) of the tile's
        // rows, relative to the box, into a small array (broadcast reads).  (Kept in the free bytes of the staged rows instead - one
        // LDS array less - the lanes of a read sit 256 bytes apart on ONE bank: 217 us instead of 192.)
        const int x = x0 + 4 * L;
        if (x1 - x0 == TW - 1 && (c.dw & 3) == 0) {
            typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
            typedef const __attribute__((address_space(1))) i32x4* g4;
            const i32x4 a4 = *(g4)(Tg + x), b4 = *(g4)(Tg + c.dw + x);
            ad[0] = a4.x; ad[1] = a4.y; ad[2] = a4.z; ad[3] = a4.w;
            bd[0] = b4.x; bd[1] = b4.y; bd[2] = b4.z; bd[3] = b4.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int cx = min(x + i, x1);
                ad[i] = Tg[cx]; bd[i] = Tg[c.dw + cx];
            }
        }
        if (tid >= TW && tid < TW + P::THP) {
            const int r = min(y0 + (tid - TW), y1);
            s_row[tid - TW] = make_int2(Tg[2 * c.dw + r] - (bx0a << 10), Tg[2 * c.dw + c.dh + r] - (by0 << 10));
        }
#pragma unroll
        for (int k = 0; k < NCH; k++) {
            const int i = tid + NT * k;
            if (i < total) {
                const int r = i / P::CPR;
                *reinterpret_cast<uint4*>(tile + r * P::PB + 16 * (i - r * P::CPR)) = d[k];
            }
        }
    }
    if (!fit) {
        plane_direct_tile<CN>(c, src, dst, Tg, reinterpret_cast<int*>(tile), x0, y0, x1, y1, bx0a, by0, bw, tid);
        return;
    }
    __syncthreads();
    // ---- output
    plane_blend_rows<CN>(c, tile, lut, s_row, ad, bd, L, ty, dst, (uint32_t)c.dstride, x0, y0, x1, y1);
}

template <int CN>
__global__ __launch_bounds__(NT, 8) void warp_plane_kernel(gtab_t tabs, int tab_stride, int tab_row, int tab_ad, uint32_t sstride, uint32_t dstride,
                                                        uint32_t swh, uint32_t dwh, uint32_t flags, uint32_t mgx, uint32_t mgy) {
    typedef PlaneCfg<CN> P;
    __shared__ __attribute__((aligned(16))) uint8_t tile[P::ROWS * P::PB];
    __shared__ __attribute__((aligned(16))) uint8_t lut[32 * LUT_STRIDE];
    __shared__ __attribute__((aligned(16))) int2 s_row[P::THP];                 // (X0, Y0) of the tile's rows
    typedef const __attribute__((address_space(4))) int32_t* cptr;
    WarpCore c;
    c.sstride = sstride; c.dstride = dstride;
    c.sw = swh & 0xFFFFu; c.sh = swh >> 16; c.dw = dwh & 0xFFFFu; c.dh = dwh >> 16;
    c.src_aligned = flags & 1u; c.dst_aligned = (flags >> 1) & 1u; c.border = (flags >> 2) & 7u;
    const TileId tq = tile_id(flags, (uint32_t)(c.dw + TW - 1) / TW, (uint32_t)(c.dh + P::THP - 1) / P::THP, mgx, mgy);
    plane_tile<CN>(c, (cptr)(const int32_t*)(tabs + (size_t)tq.z * tab_stride), tabs + (size_t)tq.z * tab_stride + tab_ad, tab_row, tq.x, tq.y,
                   tile, lut, s_row, threadIdx.x);
}

// ---- NV12 surfaces: luma and chroma tiles of all frames in ONE launch ---------------------------------------------------------
// As two launches per 32 surfaces (warp_plane_kernel<1>, then <2>) the warp had two launch tails and two table walks, and the
// luma launch - bound by its blend's instruction count - never shared a CU with the chroma launch, which is bound by its memory
// phase.  Here the launch is a sequence of tiles (frame, plane, tile row, tile column): a frame's luma tiles, then its chroma
// tiles; workgroup `lin` takes tile seq(lin) of that sequence, the XCD-aware order as in tile_id (XCD c = lin mod 8 takes the
// c-th contiguous eighth).  A frame's tables lie in one block (warp_tab.h): luma table, then chroma table, `tab_stride` ints from
// frame to frame.  The chroma plane's geometry is the luma plane's halved (surfaces share one pitch); flags as in the plane
// kernels plus the frame count in bits 16...  mtpf = ceil(2^32 / tiles per frame), mgx1 / mgx2 likewise for the tile columns.
template <int BORDER>
__global__ __launch_bounds__(NT, 8) void warp_nv12_kernel(gtab_t tabs, int tab_stride, uint32_t sstride, uint32_t dstride, uint32_t swh, uint32_t dwh,
                                                       uint32_t flags, uint32_t mtpf, uint32_t mgx1, uint32_t mgx2) {
    typedef PlaneCfg<1> P1;
    typedef PlaneCfg<2> P2;
    constexpr int TILE_BYTES = P1::ROWS * P1::PB > P2::ROWS * P2::PB ? P1::ROWS * P1::PB : P2::ROWS * P2::PB;
    __shared__ __attribute__((aligned(16))) uint8_t tile[TILE_BYTES];
    __shared__ __attribute__((aligned(16))) uint8_t lut[32 * LUT_STRIDE];
    __shared__ __attribute__((aligned(16))) int2 s_row[P1::THP];
    typedef const __attribute__((address_space(4))) int32_t* cptr;
    WarpCore c;
    c.sstride = sstride; c.dstride = dstride;
    c.sw = swh & 0xFFFFu; c.sh = swh >> 16; c.dw = dwh & 0xFFFFu; c.dh = dwh >> 16;
    c.src_aligned = flags & 1u; c.dst_aligned = (flags >> 1) & 1u; c.border = (flags >> 2) & 7u;
    const uint32_t gx1 = (uint32_t)(c.dw + TW - 1) / TW, gy1 = (uint32_t)(c.dh + P1::THP - 1) / P1::THP;
    const uint32_t dw2 = (uint32_t)c.dw >> 1, dh2 = (uint32_t)c.dh >> 1;
    const uint32_t gx2 = (dw2 + TW - 1) / TW, gy2 = (dh2 + P2::THP - 1) / P2::THP;
    const uint32_t n1 = gx1 * gy1, tpf = n1 + gx2 * gy2, n = tpf * (flags >> 16);
    const uint32_t lin = blockIdx.x;
    uint32_t seq = lin;
    if (flags & 0x100u) {
        const uint32_t cx = lin & 7u, k = lin >> 3, q = n >> 3, r = n & 7u;
        seq = cx * q + (cx < r ? cx : r) + k;
    }
    const uint32_t f = __umulhi(seq, mtpf);
    uint32_t t = seq - f * tpf;
    gtab_t T = tabs + (size_t)f * tab_stride;
    if (t < n1) {
        const TabLayout L = tab_layout(c.dw, c.dh);
        const uint32_t row = gx1 == 1 ? t : __umulhi(t, mgx1);
        plane_tile<1, BORDER>(c, (cptr)(const int32_t*)T, T + L.ad, L.row, (int)(t - row * gx1), (int)row, tile, lut, s_row, threadIdx.x);
    } else {
        t -= n1;
        T += tab_layout(c.dw, c.dh).stride;
        c.sw >>= 1; c.sh >>= 1; c.dw = (int)dw2; c.dh = (int)dh2;
        c.src_aligned = (flags >> 5) & 1u; c.dst_aligned = (flags >> 6) & 1u;
        const TabLayout L = tab_layout(c.dw, c.dh);
        const uint32_t row = gx2 == 1 ? t : __umulhi(t, mgx2);
        plane_tile<2, BORDER>(c, (cptr)(const int32_t*)T, T + L.ad, L.row, (int)(t - row * gx2), (int)row, tile, lut, s_row, threadIdx.x);
    }
}

// Ints of table workspace per frame of dw x dh (see warp_tables_kernel).
inline int tab_stride_of(int dw, int dh) { return tab_layout(dw, dh).stride; }

// what: VS_WARP_ALL = tables (when d_tabs is given) and warp; VS_WARP_TABLES_ONLY / VS_WARP_ONLY = the two halves apart, so
// that a caller whose maps are ready long before it warps (the batch tail of the stabilizer) builds the tables then.
// tab_stride: ints from one frame's table to the next (0: the tables are packed; NV12 blocks hold two planes' tables per frame)
template <int CN>
void launch_one(WarpArgs& a, dim3 grid, int32_t* d_tabs, int what, hipStream_t st, int tab_stride) {
    a.tabs = d_tabs;
    const TabLayout tl = tab_layout(a.c.dw, a.c.dh);
    a.tab_stride = tab_stride > 0 ? tab_stride : tl.stride; a.tab_row = tl.row; a.tab_ad = tl.ad;
    if (d_tabs) {
        if (what != VS_WARP_ONLY)
            hipLaunchKernelGGL(warp_tables_kernel, dim3((a.c.dw + a.c.dh + grid.x + grid.y + NT - 1) / NT, grid.z), dim3(NT), 0, st, a);
        if (what == VS_WARP_TABLES_ONLY) return;
        // (the table kernels pack sizes into 16 bits and form row offsets with 24-bit multiplies: larger frames or pitches of
        // 16 MiB and more take the general kernel)
        const bool packs = a.c.sw < 65536 && a.c.sh < 65536 && a.c.dw < 65536 && a.c.dh < 65536 && a.c.sstride < (1ull << 24) && a.c.dstride < (1ull << 24);
        // XCD-aware tile order (tile_id): the reciprocals of the grid's x and y extents, and the flag only where they are exact
        auto order = [](const dim3& g, uint32_t* mgx, uint32_t* mgy) -> uint32_t {
            *mgx = (uint32_t)((0x100000000ull + g.x - 1) / g.x); *mgy = (uint32_t)((0x100000000ull + g.y - 1) / g.y);
            const unsigned long long n = (unsigned long long)g.x * g.y * g.z;
            return (g.x > 1 || g.y > 1) && n * std::max(g.x, g.y) < (1ull << 31) && g.z < 65536 ? (0x100u | (uint32_t)g.z << 16) : 0u;
        };
        uint32_t mgx = 0, mgy = 0;
        if (CN == 3 && packs) {
            // 12 dwords of arguments: all of them among the 14 the hardware preloads into scalar registers
            const uint32_t of = order(grid, &mgx, &mgy);
            hipLaunchKernelGGL(warp_tab_kernel, grid, dim3(NT), 0, st, (gtab_t)a.tabs, a.tab_stride, a.tab_row, a.tab_ad, (uint32_t)a.c.sstride,
                               (uint32_t)a.c.dstride, (uint32_t)a.c.sw | (uint32_t)a.c.sh << 16, (uint32_t)a.c.dw | (uint32_t)a.c.dh << 16,
                               (uint32_t)(a.c.src_aligned ? 1 : 0) | (uint32_t)(a.c.dst_aligned ? 2 : 0) | (uint32_t)a.c.border << 2 | of, mgx, mgy);
        } else if (CN != 3 && packs && a.c.border == VS_BORDER_BLACK) {
            const dim3 pg(grid.x, (a.c.dh + PlaneCfg<CN>::THP - 1) / PlaneCfg<CN>::THP, grid.z);
            const uint32_t of = order(pg, &mgx, &mgy);
            hipLaunchKernelGGL((warp_plane_kernel<(CN == 3 ? 1 : CN)>), pg, dim3(NT), 0, st, (gtab_t)a.tabs, a.tab_stride, a.tab_row, a.tab_ad, (uint32_t)a.c.sstride,
                               (uint32_t)a.c.dstride, (uint32_t)a.c.sw | (uint32_t)a.c.sh << 16, (uint32_t)a.c.dw | (uint32_t)a.c.dh << 16,
                               (uint32_t)(a.c.src_aligned ? 1 : 0) | (uint32_t)(a.c.dst_aligned ? 2 : 0) | (uint32_t)a.c.border << 2 | of, mgx, mgy);
        } else {
            hipLaunchKernelGGL((warp_affine_kernel<CN, true>), grid, dim3(NT), 0, st, a);
        }
    } else {
        hipLaunchKernelGGL((warp_affine_kernel<CN, false>), grid, dim3(NT), 0, st, a);
    }
}

void launch_cn(WarpArgs& a, dim3 grid, int cn, int32_t* d_tabs, hipStream_t st, int what = VS_WARP_ALL, int tab_stride = 0) {
    if (cn == 3) launch_one<3>(a, grid, d_tabs, what, st, tab_stride);
    else if (cn == 1) launch_one<1>(a, grid, d_tabs, what, st, tab_stride);
    else launch_one<2>(a, grid, d_tabs, what, st, tab_stride);
}

void fill_common(WarpArgs& a, const uint8_t* d_src, size_t sstride, size_t sframe, int sw, int sh, uint8_t* d_dst,
                 size_t dstride, size_t dframe, int dw, int dh, int cn) {
    a.src = d_src; a.dst = d_dst;
    a.c.sstride = sstride; a.sframe = sframe; a.c.dstride = dstride; a.dframe = dframe;
    a.c.sw = sw; a.c.sh = sh; a.c.dw = dw; a.c.dh = dh;
    a.c.border = VS_BORDER_BLACK;
    a.minv_stride = 6;
    a.use_list = 0;
    a.tabs = nullptr; a.tab_stride = 0; a.tab_row = 0; a.tab_ad = 0;
    for (int i = 0; i < MAXB; i++) { a.srcs[i] = nullptr; a.dsts[i] = nullptr; }
    const int galign = cn == 2 ? 8 : 4;
    a.c.src_aligned = ((uintptr_t)d_src % galign == 0) && (sstride % galign == 0) && (sframe % galign == 0);
    a.c.dst_aligned = ((uintptr_t)d_dst % galign == 0) && (dstride % galign == 0) && (dframe % galign == 0);
}

bool bad_args(const void* d_src, const void* d_dst, const void* M, size_t sstride, int sw, int sh, size_t dstride,
              int dw, int dh, int cn, int batch) {
    return !d_src || !d_dst || !M || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0 || batch <= 0 ||
           (cn != 1 && cn != 2 && cn != 3) || sstride < (size_t)sw * cn || dstride < (size_t)dw * cn ||
           dh > 65535 * TH;
}

// Table scratch of the standalone operator (vs_op_warp_affine): one grow-only device buffer per stream.  Launches on
// a stream run in order, so the tables of a call are consumed before the next call on that stream rewrites them.
// (hipMallocAsync memory is not used: with it, in-flight builds of round 2 read zeroed tables for the later frames of a launch
// and faulted on the null source pointers; the cause was not established - DESIGN.md section 8 has the evidence.)
struct OpScratch { int32_t* p = nullptr; size_t bytes = 0; };
std::mutex g_op_mutex;
std::map<std::pair<int, hipStream_t>, OpScratch> g_op_scratch;

int op_tabs(hipStream_t st, size_t bytes, int32_t** out) {
    int dev = 0;
    VS_HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> g(g_op_mutex);
    OpScratch& sc = g_op_scratch[std::make_pair(dev, st)];
    if (sc.bytes < bytes) {
        if (sc.p) {                       // queued launches may still read the old block
            VS_HIP_TRY(hipStreamSynchronize(st));
            (void)hipFree(sc.p);
            sc.p = nullptr; sc.bytes = 0;
        }
        VS_HIP_TRY(hipMalloc((void**)&sc.p, bytes));
        sc.bytes = bytes;
    }
    *out = sc.p;
    return VS_OK;
}

}  // namespace

// Ints of workspace the table form of a launch over `frames` frames of dw x dh needs (d_tabs of the launchers).
size_t warp_tabs_ints(int dw, int dh, int frames) { return (size_t)tab_stride_of(dw, dh) * (size_t)(frames > 0 ? frames : 1); }

int launch_warp_affine(const uint8_t* d_src, size_t sstride, size_t sframe, int sw, int sh,
                       uint8_t* d_dst, size_t dstride, size_t dframe, int dw, int dh, int cn,
                       const double* d_Minv, int batch, int32_t* d_tabs, hipStream_t st) {
    if (bad_args(d_src, d_dst, d_Minv, sstride, sw, sh, dstride, dw, dh, cn, batch) || batch > 65535) {
        set_last_error("warp_affine: invalid argument");
        return VS_ERR_INVALID_ARG;
    }
    WarpArgs a;
    fill_common(a, d_src, sstride, sframe, sw, sh, d_dst, dstride, dframe, dw, dh, cn);
    a.Minv_dev = d_Minv;
    dim3 grid((dw + TW - 1) / TW, (dh + TH - 1) / TH, batch);
    launch_cn(a, grid, cn, d_tabs, st);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

// Frames given one by one (deferred output of a stream: each result goes to its caller's buffer);
// all share one geometry.  d_Minv: inverse maps on the device, minv_stride doubles apart.
int launch_warp_affine_list(const uint8_t* const* srcs, uint8_t* const* dsts, int n, size_t sstride, int sw, int sh,
                            size_t dstride, int dw, int dh, int cn, const double* d_Minv, int minv_stride, int32_t* d_tabs,
                            hipStream_t st, int what, int tab_stride) {
    if ((what != VS_WARP_ALL && !d_tabs) || n < 1 || n > MAXB || !srcs || !dsts || bad_args(srcs[0], dsts[0], d_Minv, sstride, sw, sh, dstride, dw, dh, cn, n)) {
        set_last_error("warp_affine_list: invalid argument");
        return VS_ERR_INVALID_ARG;
    }
    WarpArgs a;
    fill_common(a, srcs[0], sstride, 0, sw, sh, dsts[0], dstride, 0, dw, dh, cn);
    const int galign = cn == 2 ? 8 : 4;
    for (int i = 0; i < MAXB; i++) {
        a.srcs[i] = srcs[i < n ? i : 0]; a.dsts[i] = dsts[i < n ? i : 0];
        if (!a.srcs[i] || !a.dsts[i]) { set_last_error("warp_affine_list: null frame"); return VS_ERR_INVALID_ARG; }
        if ((uintptr_t)a.srcs[i] % galign) a.c.src_aligned = 0;
        if ((uintptr_t)a.dsts[i] % galign) a.c.dst_aligned = 0;
    }
    a.use_list = 1;
    a.Minv_dev = d_Minv;
    a.minv_stride = minv_stride;
    dim3 grid((dw + TW - 1) / TW, (dh + TH - 1) / TH, n);
    launch_cn(a, grid, cn, d_tabs, st, what, tab_stride);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

// NV12 surfaces of one geometry and pitch, luma and chroma planes in ONE launch (warp_nv12_kernel).  ys / yd: the surfaces (luma
// plane first), the interleaved chroma plane src_uv / dst_uv bytes behind them.  d_tabs: the frames' table blocks (warp_tab.h:
// luma table, then chroma table; nv12_tab_ints(w, h) ints per frame), ALREADY BUILT (the release workgroups of the batch tail,
// or launch_warp_affine_list(.., VS_WARP_TABLES_ONLY, nv12_tab_ints) per plane).  Returns VS_ERR_UNSUPPORTED when the geometry
// is outside what the kernel packs (the caller then launches the planes one by one).
int launch_warp_nv12_list(const uint8_t* const* ys, uint8_t* const* yd, int n, size_t sstride, size_t dstride, int w, int h, size_t src_uv,
                          size_t dst_uv, const int32_t* d_tabs, hipStream_t st, int border) {
    if (!ys || !yd || !d_tabs || n < 1 || w < 2 || h < 2 || (w & 1) || (h & 1) || (border != VS_BORDER_BLACK && border != VS_BORDER_REPLICATE)) {
        set_last_error("warp_nv12_list: invalid argument");
        return VS_ERR_INVALID_ARG;
    }
    typedef PlaneCfg<1> P1;
    typedef PlaneCfg<2> P2;
    const unsigned long long gx1 = (w + TW - 1) / TW, gy1 = (h + P1::THP - 1) / P1::THP, gx2 = (w / 2 + TW - 1) / TW, gy2 = (h / 2 + P2::THP - 1) / P2::THP;
    const unsigned long long tpf = gx1 * gy1 + gx2 * gy2, total = tpf * (unsigned long long)n;
    if (w >= 65536 || h >= 65536 || sstride >= (1ull << 24) || dstride >= (1ull << 24) || n >= 65536 || total * std::max(tpf, std::max(gx1, gx2)) >= (1ull << 31))
        return VS_ERR_UNSUPPORTED;
    uint32_t al = 0xFu;          // bit 0 / 1: luma source / destination 4-byte aligned; bit 2 / 3: chroma 8-byte aligned
    if (sstride % 4) al &= ~1u;
    if (dstride % 4) al &= ~2u;
    if (sstride % 8 || src_uv % 8) al &= ~4u;
    if (dstride % 8 || dst_uv % 8) al &= ~8u;
    for (int i = 0; i < n; i++) {
        if (!ys[i] || !yd[i]) { set_last_error("warp_nv12_list: null frame"); return VS_ERR_INVALID_ARG; }
        if ((uintptr_t)ys[i] % 4) al &= ~1u;
        if ((uintptr_t)yd[i] % 4) al &= ~2u;
        if ((uintptr_t)ys[i] % 8) al &= ~4u;
        if ((uintptr_t)yd[i] % 8) al &= ~8u;
    }
    const uint32_t flags = (al & 1u) | (al & 2u) | (uint32_t)border << 2 | ((al >> 2) & 1u) << 5 | ((al >> 3) & 1u) << 6 | 0x100u | (uint32_t)n << 16;
    const uint32_t mtpf = (uint32_t)((0x100000000ull + tpf - 1) / tpf), mgx1 = (uint32_t)((0x100000000ull + gx1 - 1) / gx1),
                   mgx2 = (uint32_t)((0x100000000ull + gx2 - 1) / gx2);
    if (border == VS_BORDER_REPLICATE)
        hipLaunchKernelGGL(warp_nv12_kernel<VS_BORDER_REPLICATE>, dim3((unsigned)total), dim3(NT), 0, st, (gtab_t)d_tabs, nv12_tab_ints(w, h), (uint32_t)sstride,
                           (uint32_t)dstride, (uint32_t)w | (uint32_t)h << 16, (uint32_t)w | (uint32_t)h << 16, flags, mtpf, mgx1, mgx2);
    else
        hipLaunchKernelGGL(warp_nv12_kernel<VS_BORDER_BLACK>, dim3((unsigned)total), dim3(NT), 0, st, (gtab_t)d_tabs, nv12_tab_ints(w, h), (uint32_t)sstride,
                           (uint32_t)dstride, (uint32_t)w | (uint32_t)h << 16, (uint32_t)w | (uint32_t)h << 16, flags, mtpf, mgx1, mgx2);
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
    // coordinate tables for launches of several frames
    int32_t* d_tabs = nullptr;
    if (batch >= 4) VS_TRY(op_tabs(st, warp_tabs_ints(dw, dh, batch < MAXB ? batch : MAXB) * sizeof(int32_t), &d_tabs));
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
        launch_cn(a, grid, cn, d_tabs, st);
        VS_HIP_TRY(hipGetLastError());
    }
    return VS_OK;
}

// NV12 surfaces (luma plane, interleaved chroma plane h * pitch behind it), host matrices: launches of four and more frames build
// both planes' tables (chroma: the map with the halved translation) and warp them in ONE grid; fewer frames go plane by plane.
int launch_warp_nv12_hostM(const uint8_t* d_src, size_t sstride, size_t sframe, uint8_t* d_dst, size_t dstride, size_t dframe, int w, int h,
                           const float* h_M, int batch, hipStream_t st) {
    if (!d_src || !d_dst || !h_M || w < 2 || h < 2 || (w & 1) || (h & 1) || batch <= 0 || sstride < (size_t)w || dstride < (size_t)w) {
        set_last_error("warp_affine_nv12: invalid argument (w and h must be even)");
        return VS_ERR_INVALID_ARG;
    }
    std::vector<float> Mc((size_t)batch * 6);
    for (int b = 0; b < batch; b++) {
        const float* m = h_M + 6 * b;
        float* c = &Mc[6 * (size_t)b];
        c[0] = m[0]; c[1] = m[1]; c[2] = m[2] * 0.5f;
        c[3] = m[3]; c[4] = m[4]; c[5] = m[5] * 0.5f;
    }
    const size_t suv = (size_t)h * sstride, duv = (size_t)h * dstride;
    const int block = nv12_tab_ints(w, h), sy = tab_layout(w, h).stride;
    for (int b0 = 0; b0 < batch; b0 += MAXB) {
        const int nb = std::min(MAXB, batch - b0);
        int one = VS_ERR_UNSUPPORTED;
        if (nb >= 4) {
            int32_t* d_tabs = nullptr;
            VS_TRY(op_tabs(st, (size_t)block * nb * sizeof(int32_t), &d_tabs));
            const uint8_t* ys[MAXB];
            uint8_t* yd[MAXB];
            for (int plane = 0; plane < 2; plane++) {
                WarpArgs a;
                fill_common(a, d_src + (size_t)b0 * sframe + (plane ? suv : 0), sstride, sframe, plane ? w / 2 : w, plane ? h / 2 : h,
                            d_dst + (size_t)b0 * dframe + (plane ? duv : 0), dstride, dframe, plane ? w / 2 : w, plane ? h / 2 : h, plane ? 2 : 1);
                a.Minv_dev = nullptr;
                const float* Ms = plane ? Mc.data() : h_M;
                for (int b = 0; b < MAXB; b++) {
                    if (b < nb) warp_invert(Ms + (size_t)(b0 + b) * 6, a.Minv_val + 6 * b);
                    else for (int i = 0; i < 6; i++) a.Minv_val[6 * b + i] = 0.;
                }
                dim3 grid((a.c.dw + TW - 1) / TW, (a.c.dh + TH - 1) / TH, nb);
                launch_cn(a, grid, plane ? 2 : 1, d_tabs + (plane ? sy : 0), st, VS_WARP_TABLES_ONLY, block);
            }
            for (int b = 0; b < nb; b++) { ys[b] = d_src + (size_t)(b0 + b) * sframe; yd[b] = d_dst + (size_t)(b0 + b) * dframe; }
            one = launch_warp_nv12_list(ys, yd, nb, sstride, dstride, w, h, suv, duv, d_tabs, st);
            if (one != VS_OK && one != VS_ERR_UNSUPPORTED) return one;
        }
        if (one == VS_ERR_UNSUPPORTED) {
            VS_TRY(launch_warp_affine_hostM(d_src + (size_t)b0 * sframe, sstride, sframe, w, h, d_dst + (size_t)b0 * dframe, dstride, dframe, w, h, 1,
                                            h_M + (size_t)b0 * 6, nb, st));
            VS_TRY(launch_warp_affine_hostM(d_src + (size_t)b0 * sframe + suv, sstride, sframe, w / 2, h / 2, d_dst + (size_t)b0 * dframe + duv, dstride,
                                            dframe, w / 2, h / 2, 2, Mc.data() + (size_t)b0 * 6, nb, st));
        }
    }
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

// Frames given one by one, inverse maps given on the host in double (6 per frame), selectable border: the rotations of a batch
// of roll-corrected frames (cv::warpAffine(..., BORDER_REPLICATE), RollCorrection.cpp:146-149) as ONE launch per plane.  Four
// frames and more take the coordinate tables (scratch of the stream, op_tabs).
int launch_warp_affine_list_inv(const uint8_t* const* srcs, uint8_t* const* dsts, int n, size_t sstride, int sw, int sh, size_t dstride, int dw, int dh,
                                int cn, const double* h_Minv, int border, hipStream_t st) {
    if (n < 1 || n > MAXB || !srcs || !dsts || bad_args(srcs[0], dsts[0], h_Minv, sstride, sw, sh, dstride, dw, dh, cn, n) ||
        (border != VS_BORDER_BLACK && border != VS_BORDER_REPLICATE)) {
        set_last_error("warp_affine_list_inv: invalid argument");
        return VS_ERR_INVALID_ARG;
    }
    WarpArgs a;
    fill_common(a, srcs[0], sstride, 0, sw, sh, dsts[0], dstride, 0, dw, dh, cn);
    const int galign = cn == 2 ? 8 : 4;
    for (int i = 0; i < MAXB; i++) {
        a.srcs[i] = srcs[i < n ? i : 0]; a.dsts[i] = dsts[i < n ? i : 0];
        if (!a.srcs[i] || !a.dsts[i]) { set_last_error("warp_affine_list_inv: null frame"); return VS_ERR_INVALID_ARG; }
        if ((uintptr_t)a.srcs[i] % galign) a.c.src_aligned = 0;
        if ((uintptr_t)a.dsts[i] % galign) a.c.dst_aligned = 0;
    }
    a.use_list = 1;
    a.Minv_dev = nullptr;
    a.c.border = border;
    for (int i = 0; i < MAXB * 6; i++) a.Minv_val[i] = i < 6 * n ? h_Minv[i] : 0.;
    int32_t* d_tabs = nullptr;
    if (n >= 4) VS_TRY(op_tabs(st, warp_tabs_ints(dw, dh, n) * sizeof(int32_t), &d_tabs));
    dim3 grid((dw + TW - 1) / TW, (dh + TH - 1) / TH, n);
    launch_cn(a, grid, cn, d_tabs, st);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

// NV12 surfaces given one by one, inverse maps of both planes given on the host (6 doubles per frame and plane), selectable
// border: four and more surfaces build their table blocks with one launch (more than 16: one per plane) and are warped in ONE
// grid (the rotations of a batch of roll-corrected surfaces); fewer go plane by plane.
int launch_warp_nv12_list_inv(const uint8_t* const* ys, uint8_t* const* yd, int n, size_t sstride, size_t dstride, int w, int h, size_t src_uv,
                              size_t dst_uv, const double* h_MinvY, const double* h_MinvUV, int border, hipStream_t st) {
    if (n < 1 || n > MAXB || !ys || !yd || !h_MinvY || !h_MinvUV) { set_last_error("warp_nv12_list_inv: invalid argument"); return VS_ERR_INVALID_ARG; }
    const uint8_t* us[MAXB];
    uint8_t* ud[MAXB];
    for (int i = 0; i < n; i++) {
        if (!ys[i] || !yd[i]) { set_last_error("warp_nv12_list_inv: null frame"); return VS_ERR_INVALID_ARG; }
        us[i] = ys[i] + src_uv; ud[i] = yd[i] + dst_uv;
    }
    int one = VS_ERR_UNSUPPORTED;
    if (n >= 4) {
        const int block = nv12_tab_ints(w, h), sy = tab_layout(w, h).stride;
        int32_t* d_tabs = nullptr;
        VS_TRY(op_tabs(st, (size_t)block * n * sizeof(int32_t), &d_tabs));
        if (n <= NVT_MAX) {       // both planes' tables of every surface with one launch
            NvTabArgs t;
            t.tabs = d_tabs; t.block = block; t.chroma = sy; t.w = w; t.h = h; t.src_uv = src_uv; t.dst_uv = dst_uv;
            for (int i = 0; i < NVT_MAX; i++) { t.ys[i] = ys[i < n ? i : 0]; t.yd[i] = yd[i < n ? i : 0]; }
            for (int i = 0; i < NVT_MAX * 6; i++) { t.my[i] = i < 6 * n ? h_MinvY[i] : 0.; t.muv[i] = i < 6 * n ? h_MinvUV[i] : 0.; }
            const int entries = w + h + (w + TW - 1) / TW + (h + TH - 1) / TH;
            hipLaunchKernelGGL(warp_tables_nv12_kernel, dim3((entries + NT - 1) / NT, n, 2), dim3(NT), 0, st, t);
        } else for (int plane = 0; plane < 2; plane++) {
            WarpArgs a;
            const int pw = plane ? w / 2 : w, ph = plane ? h / 2 : h;
            fill_common(a, plane ? us[0] : ys[0], sstride, 0, pw, ph, plane ? ud[0] : yd[0], dstride, 0, pw, ph, plane ? 2 : 1);
            for (int i = 0; i < MAXB; i++) { a.srcs[i] = plane ? us[i < n ? i : 0] : ys[i < n ? i : 0]; a.dsts[i] = plane ? ud[i < n ? i : 0] : yd[i < n ? i : 0]; }
            a.use_list = 1;
            a.Minv_dev = nullptr;
            a.c.border = border;
            const double* Mi = plane ? h_MinvUV : h_MinvY;
            for (int i = 0; i < MAXB * 6; i++) a.Minv_val[i] = i < 6 * n ? Mi[i] : 0.;
            dim3 grid((pw + TW - 1) / TW, (ph + TH - 1) / TH, n);
            launch_cn(a, grid, plane ? 2 : 1, d_tabs + (plane ? sy : 0), st, VS_WARP_TABLES_ONLY, block);
        }
        one = launch_warp_nv12_list(ys, yd, n, sstride, dstride, w, h, src_uv, dst_uv, d_tabs, st, border);
        if (one != VS_OK && one != VS_ERR_UNSUPPORTED) return one;
    }
    if (one == VS_ERR_UNSUPPORTED) {
        VS_TRY(launch_warp_affine_list_inv(ys, yd, n, sstride, w, h, dstride, w, h, 1, h_MinvY, border, st));
        VS_TRY(launch_warp_affine_list_inv(us, ud, n, sstride, w / 2, h / 2, dstride, w / 2, h / 2, 2, h_MinvUV, border, st));
    }
    VS_HIP_TRY(hipGetLastError());
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
    a.c.border = border;
    for (int i = 0; i < MAXB * 6; i++) a.Minv_val[i] = i < 6 ? h_Minv[i] : 0.;
    dim3 grid((dw + TW - 1) / TW, (dh + TH - 1) / TH, 1);
    launch_cn(a, grid, cn, nullptr, st);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

// n <= WARP_JOBS_MAX warps of any geometry (inverse maps in double on the host) as one launch.
int launch_warp_jobs(const WarpJob* jobs, int n, hipStream_t st) {
    if (!jobs || n < 1 || n > WARP_JOBS_MAX) { set_last_error("warp_jobs: invalid argument"); return VS_ERR_INVALID_ARG; }
    WarpJobsArg a;
    int gw = 1, gh = 1;
    for (int i = 0; i < WARP_JOBS_MAX; i++) {
        a.j[i] = jobs[i < n ? i : 0];
        if (i >= n) continue;
        const WarpJob& j = jobs[i];
        if (bad_args(j.src, j.dst, j.m, j.sstride, j.sw, j.sh, j.dstride, j.dw, j.dh, j.cn, 1) ||
            (j.border != VS_BORDER_BLACK && j.border != VS_BORDER_REPLICATE)) {
            set_last_error("warp_jobs: invalid argument");
            return VS_ERR_INVALID_ARG;
        }
        gw = std::max(gw, (j.dw + TW - 1) / TW);
        gh = std::max(gh, (j.dh + TH - 1) / TH);
    }
    hipLaunchKernelGGL(warp_jobs_kernel, dim3(gw, gh, n), dim3(NT), 0, st, a);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

}  // namespace vsd
