// Pyramidal Lucas-Kanade tracker for gfx950: device counterpart of
//   cv::calcOpticalFlowPyrLK(prevGray, currGray, prevPts, nextPts, status, err,
//                            Size(win,win), maxLevel, TermCriteria(COUNT+EPS, iters, eps))
// as called at /root/reference/src/Stabilizer.cpp:611-619 (15x15, maxLevel 2,
// 20 iterations, eps 0.03; Stabilizer_legacy.cpp:218-224 uses 21x21).
//
// One wavefront (64 lanes) owns one feature point for ALL pyramid levels
// (coarse to fine), so the whole tracker is a single launch: points are
// independent, levels of one point are sequential.  The win*win window is
// spread over the lanes (ceil(win^2/64) pixels per lane, kept in registers as
// int16 I / dIx / dIy; every sample of the iteration comes from a search region staged in LDS, which
// is staged again around the point if it walks out of it); the 2x2 normal-equation sums (A11,A12,A22 and per
// iteration b1,b2) are exact int64 wave reductions, so the float solve that
// follows is bit-identical to the oracle's.  Latency-bound gather kernel
// (SURVEY.md 8a L1): images are <= 0.5 MB and stay in L2.
#include "vs_common.h"

namespace vsd {
namespace {

constexpr int MAX_LEVELS = 8;

struct LKArgs {
    LKLevel levels[MAX_LEVELS];
    int max_level;
    const float* prev_pts;
    float* next_pts;
    uint8_t* status;
    float* err;
    int n;
    const int32_t* d_n;
    int win;
    int max_count;
    double eps2;
};

// ---- exact wave-wide integer sums without LDS traffic -----------------------
// Butterfly inside each 16-lane row with DPP (xor 1, xor 2, half-mirror, mirror),
// then the four row totals are read with v_readlane and added on the scalar unit.
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ int wave_sum_i32(int v) {
    v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);   // row_half_mirror
    v += dpp_mov<0x140>(v);   // row_mirror
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) +
           __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
// 64-bit variant: the partner's (lo,hi) pair is fetched with two DPP moves and added
// with a normal 64-bit add, four butterfly steps per 16-lane row, then the four row
// totals are read with v_readlane and added on the scalar unit.  Per-lane partial
// sums fit int32 (<= 16 px * 8160 * 4080); the wave total may not.
template <int CTRL>
__device__ __forceinline__ long long dpp_mov64(long long v) {
    const int lo = dpp_mov<CTRL>((int)(unsigned)(unsigned long long)v);
    const int hi = dpp_mov<CTRL>((int)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ long long wave_sum(int p) {
    long long v = p;
    v += dpp_mov64<0xB1>(v);
    v += dpp_mov64<0x4E>(v);
    v += dpp_mov64<0x141>(v);
    v += dpp_mov64<0x140>(v);
    const int lo = (int)(unsigned)(unsigned long long)v, hi = (int)((unsigned long long)v >> 32);
    long long t = 0;
#pragma unroll
    for (int r = 0; r < 64; r += 16) {
        const unsigned l = (unsigned)__builtin_amdgcn_readlane(lo, r);
        const unsigned h = (unsigned)__builtin_amdgcn_readlane(hi, r);
        t += (long long)(((unsigned long long)h << 32) | l);
    }
    return t;
}

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

struct Weights { int w00, w01, w10, w11; };

__device__ __forceinline__ Weights make_weights(float a, float b) {
    Weights w;
    w.w00 = f_round((1.f - a) * (1.f - b) * 16384.f);
    w.w01 = f_round(a * (1.f - b) * 16384.f);
    w.w10 = f_round((1.f - a) * b * 16384.f);
    w.w11 = 16384 - w.w00 - w.w01 - w.w10;
    return w;
}

__device__ __forceinline__ short2 load_deriv(const int16_t* __restrict__ d, int w, int h, int x, int y) {
    if ((unsigned)x >= (unsigned)w || (unsigned)y >= (unsigned)h) return make_short2(0, 0);
    return *reinterpret_cast<const short2*>(d + ((size_t)y * w + x) * 2);
}

// A pointer every lane of the wave holds the same value of, as a global-memory pointer in scalar registers.
typedef const __attribute__((address_space(1))) uint8_t* gmem_u8;
typedef const __attribute__((address_space(1))) int16_t* gmem_i16;
__device__ __forceinline__ gmem_u8 uniform_global(const uint8_t* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (gmem_u8)(((unsigned long long)hi << 32) | lo);
}

constexpr int LK_MARGIN = 6;                       // search region = window + 1 + 2*margin
constexpr int LK_WIN_MAX = 31;



// i / d and i % d for 0 <= i < 4096, 1 <= d <= 64 (exact: the float product is
// off by < 1e-3 from the true quotient and we add 0.5/d of slack)
__device__ __forceinline__ void divmod_small(int i, int d, float inv_d, int& q, int& r) {
    q = (int)(((float)i + 0.5f) * inv_d);
    r = i - q * d;
}

template <int NPX>
__device__ __forceinline__ void lk_track(const LKArgs& a, const int pt) {
    // Per level everything the wave will touch is staged into LDS with ONE round of
    // global loads: the template patch of the previous image, its derivative patch,
    // and the search region of the next image (pyramid padding already applied).
    // sized by the window class of the instantiation (NPX pixels per lane: windows up to 16 / 21 / 31): with the
    // arrays of the largest class a 21x21 tracker held 7 KB, 22 waves fitted a CU and the 6400 points of a
    // 32-frame batch took two rounds on 5632 slots; 3.6 KB leaves the wave limit (32 per CU) as the only one
    constexpr int WM = NPX <= 4 ? 16 : (NPX <= 7 ? 21 : LK_WIN_MAX);
    constexpr int PWM = WM + 1, RWM = WM + 1 + 2 * LK_MARGIN;
    // elements per lane of the lane-strided staging walks; the arrays are padded to whole rounds of 64 so that the stores of a
    // round need no bounds check (the padding is never read)
    constexpr int PN = (PWM * PWM + 63) / 64, RN = (RWM * RWM + 63) / 64;
    __shared__ uint8_t pI[PN * 64];
    __shared__ short2 pD[PN * 64];
    __shared__ uint8_t region[RN * 64];
    const int lane = threadIdx.x;
    int n = a.n;
    if (a.d_n) { int dn = *a.d_n; n = dn < n ? dn : n; }
    if (pt >= n) return;
    const int win = a.win, area = win * win;
    const int PW = win + 1;
    const int RW = win + 1 + 2 * LK_MARGIN;
    const float inv_pw = 1.0f / (float)PW, inv_rw = 1.0f / (float)RW, inv_win = 1.0f / (float)win;
    // per window pixel of this lane: offsets into the template patch (low half) and the search region (high half);
    // kept packed, and the (x, y) of a pixel is recomputed where the rare out-of-region path needs it: the register
    // budget decides how many points a SIMD tracks at once
    uint32_t off[NPX];
#pragma unroll
    for (int k = 0; k < NPX; k++) {
        const int p = lane + 64 * k;
        int yy = 0, xx = 0;
        if (p < area) divmod_small(p, win, inv_win, yy, xx);
        off[k] = (uint32_t)(yy * PW + xx) | ((uint32_t)(yy * RW + xx) << 16);
    }
#define LK_VALID(k) (lane + 64 * (k) < area)
#define LK_POFF(k) ((int)(off[k] & 0xFFFFu))
#define LK_ROFF(k) ((int)(off[k] >> 16))
    // lane-strided walks over the PW x PW patch and the RW x RW region: start and step
    int p_y0, p_x0, p_sy, p_sx, r_y0, r_x0, r_sy, r_sx;
    divmod_small(lane, PW, inv_pw, p_y0, p_x0);
    divmod_small(64, PW, inv_pw, p_sy, p_sx);
    divmod_small(lane, RW, inv_rw, r_y0, r_x0);
    divmod_small(64, RW, inv_rw, r_sy, r_sx);
    const float halfWin = (win - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    const float px0 = a.prev_pts[2 * pt], py0 = a.prev_pts[2 * pt + 1];
    float outx = 0.f, outy = 0.f;  // nextPts[pt]
    int status = 1;
    float err = 0.f;

    for (int level = a.max_level; level >= 0; level--) {
        const LKLevel L = a.levels[level];
        const float scale = (float)(1. / (double)(1 << level));
        float prevx = px0 * scale, prevy = py0 * scale;
        if (level == a.max_level) { outx = prevx; outy = prevy; }
        else { outx = outx * 2.f; outy = outy * 2.f; }
        prevx -= halfWin; prevy -= halfWin;
        const int ipx = f_floor(prevx), ipy = f_floor(prevy);
        if (ipx < -win || ipx >= L.w || ipy < -win || ipy >= L.h) {
            if (level == 0) { status = 0; err = 0.f; }
            continue;
        }
        float cx = outx - halfWin, cy = outy - halfWin;
        int rx0 = f_floor(cx) - LK_MARGIN, ry0 = f_floor(cy) - LK_MARGIN;
        // ---- stage (all loads of the level are in flight together).  Lane-strided element
        // walk with incremental (row, col); patches that lie inside the image (the common
        // case) skip the border arithmetic.
        __syncthreads();
        // Patches and regions that lie inside the image (the common case) are staged by straight-line code: every load of the
        // level goes out before the first LDS store waits for one (elements past the end of a walk repeat element 0 and land in
        // the arrays' padding).  Written as loops over a run-time count the compiler had turned each walk into load - wait -
        // store per element: 26 dependent round trips per level and point, two thirds of the tracker's time
        // (SQ_WAIT_ANY 68 % of its wave cycles, profiles/r03_c_configs1_pmc.txt).
        // (Batches of at most ten loads: the values and offsets of a batch are all that is live, 128 registers - four waves per
        // SIMD - hold without spills; three to four round trips per level instead of 26.)
        const bool p_inside = ipx >= 0 && ipy >= 0 && ipx + PW <= L.w && ipy + PW <= L.h;
        if (p_inside) {
            // (the wave tracks ONE point: bases in scalar registers, global - not flat - loads with a 32-bit lane offset)
            const gmem_u8 pbase = uniform_global(L.prev + ((size_t)ipy * L.stride + ipx));
            const gmem_i16 dbase = (gmem_i16)uniform_global(reinterpret_cast<const uint8_t*>(L.deriv + ((size_t)ipy * L.w + ipx) * 2));
            constexpr int PB_ = 5;             // elements per batch (two loads each)
            int y = p_y0, x = p_x0;
#pragma unroll
            for (int k0 = 0; k0 < PN; k0 += PB_) {
                uint8_t vI[PB_];
                uint32_t vD[PB_];
#pragma unroll
                for (int k = k0; k < k0 + PB_ && k < PN; k++) {
                    const bool ok = lane + 64 * k < PW * PW;
                    const uint32_t eo = ok ? (uint32_t)(y * (int)L.stride + x) : 0u, dof = ok ? (uint32_t)((y * L.w + x) * 2) : 0u;
                    vI[k - k0] = pbase[eo];
                    vD[k - k0] = *reinterpret_cast<const __attribute__((address_space(1))) uint32_t*>(dbase + dof);
                    x += p_sx; y += p_sy;
                    if (x >= PW) { x -= PW; y++; }
                }
#pragma unroll
                for (int k = k0; k < k0 + PB_ && k < PN; k++) { pI[lane + 64 * k] = vI[k - k0]; *reinterpret_cast<uint32_t*>(&pD[lane + 64 * k]) = vD[k - k0]; }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            int y = p_y0, x = p_x0;
            for (int i = lane; i < PW * PW; i += 64) {
                const int X = ipx + x, Y = ipy + y;
                pI[i] = L.prev[(size_t)reflect101(Y, L.h) * L.stride + reflect101(X, L.w)];
                pD[i] = load_deriv(L.deriv, L.w, L.h, X, Y);
                x += p_sx; y += p_sy;
                if (x >= PW) { x -= PW; y++; }
            }
        }
        // the search region of the next image around (rx0, ry0); staged again, re-centred, if the point walks out
        // of it (more than LK_MARGIN pixels at one level), so every sample of the iteration comes from LDS
        auto stage_region = [&]() {
            const bool inside = rx0 >= 0 && ry0 >= 0 && rx0 + RW <= L.w && ry0 + RW <= L.h;
            int y = r_y0, x = r_x0;
            asm volatile("" : "+v"(x), "+v"(y));       // (the offsets of the walk are recomputed at every call: hoisted out of the iteration loop they cost 40 registers)
            if (inside) {
                const gmem_u8 rbase = uniform_global(L.next + ((size_t)ry0 * L.stride + rx0));
                constexpr int RB_ = 10;
#pragma unroll
                for (int k0 = 0; k0 < RN; k0 += RB_) {
                    uint8_t vR[RB_];
#pragma unroll
                    for (int k = k0; k < k0 + RB_ && k < RN; k++) {
                        const bool ok = lane + 64 * k < RW * RW;
                        vR[k - k0] = rbase[ok ? (uint32_t)(y * (int)L.stride + x) : 0u];
                        x += r_sx; y += r_sy;
                        if (x >= RW) { x -= RW; y++; }
                    }
#pragma unroll
                    for (int k = k0; k < k0 + RB_ && k < RN; k++) region[lane + 64 * k] = vR[k - k0];
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {                           // a region that leaves the image: element by element, reflected
                for (int i = lane; i < RW * RW; i += 64) {
                    const int X = rx0 + x, Y = ry0 + y;
                    region[i] = L.next[(size_t)reflect101(Y, L.h) * L.stride + reflect101(X, L.w)];
                    x += r_sx; y += r_sy;
                    if (x >= RW) { x -= RW; y++; }
                }
            }
        };
        stage_region();
        __syncthreads();
        // ---- template window, its gradients and the 2x2 matrix
        Weights wt = make_weights(prevx - ipx, prevy - ipy);
        short Iw[NPX], Ix[NPX], Iy[NPX];
        int pA11 = 0, pA12 = 0, pA22 = 0;
#pragma unroll
        for (int k = 0; k < NPX; k++) {
            // invalid slots read offset 0 and are zeroed afterwards
            const int o = LK_POFF(k);
            int ival = descale(__mul24((int)pI[o], wt.w00) + __mul24((int)pI[o + 1], wt.w01) +
                               __mul24((int)pI[o + PW], wt.w10) + __mul24((int)pI[o + PW + 1], wt.w11), 9);
            const short2 d00 = pD[o], d01 = pD[o + 1], d10 = pD[o + PW], d11 = pD[o + PW + 1];
            int ixval = descale(__mul24((int)d00.x, wt.w00) + __mul24((int)d01.x, wt.w01) +
                                __mul24((int)d10.x, wt.w10) + __mul24((int)d11.x, wt.w11), 14);
            int iyval = descale(__mul24((int)d00.y, wt.w00) + __mul24((int)d01.y, wt.w01) +
                                __mul24((int)d10.y, wt.w10) + __mul24((int)d11.y, wt.w11), 14);
            if (!LK_VALID(k)) { ival = 0; ixval = 0; iyval = 0; }
            Iw[k] = (short)ival; Ix[k] = (short)ixval; Iy[k] = (short)iyval;
            pA11 += __mul24(ixval, ixval);
            pA12 += __mul24(ixval, iyval);
            pA22 += __mul24(iyval, iyval);
        }
        const long long sA11 = wave_sum(pA11), sA12 = wave_sum(pA12), sA22 = wave_sum(pA22);
        const float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE, A22 = (float)sA22 * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) /
                             (float)(2 * win * win);
        if (minEig < 1e-4f || D < FLT_EPSILON) {
            if (level == 0) status = 0;
            continue;
        }
        D = 1.f / D;
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < a.max_count; j++) {
            const int inx = f_floor(cx), iny = f_floor(cy);
            if (inx < -win || inx >= L.w || iny < -win || iny >= L.h) {
                if (level == 0) status = 0;
                break;
            }
            wt = make_weights(cx - inx, cy - iny);
            int lx = inx - rx0, ly = iny - ry0;
            if (!(lx >= 0 && ly >= 0 && lx + win + 1 <= RW && ly + win + 1 <= RW)) {    // wave-uniform
                __syncthreads();
                rx0 = inx - LK_MARGIN; ry0 = iny - LK_MARGIN;
                stage_region();
                __syncthreads();
                lx = LK_MARGIN; ly = LK_MARGIN;
            }
            int pb1 = 0, pb2 = 0;
            {
                // every slot is computed unconditionally (invalid slots carry Ix = Iy = 0 and a
                // safe offset), so the LDS reads of all pixels are in flight together
                const int rbase = ly * RW + lx;
#pragma unroll
                for (int k = 0; k < NPX; k++) {
                    const uint8_t* r0 = &region[rbase + LK_ROFF(k)];
                    // signed 24-bit multiplies: w11 = 2^14 - w00 - w01 - w10 can be -1
                    const int v = descale(__mul24((int)r0[0], wt.w00) + __mul24((int)r0[1], wt.w01) +
                                          __mul24((int)r0[RW], wt.w10) + __mul24((int)r0[RW + 1], wt.w11), 9);
                    const int diff = v - Iw[k];
                    pb1 += __mul24(diff, (int)Ix[k]);
                    pb2 += __mul24(diff, (int)Iy[k]);
                }
            }
            const long long sb1 = wave_sum(pb1), sb2 = wave_sum(pb2);
            const float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
            const float dx = (A12 * b2 - A22 * b1) * D;
            const float dy = (A12 * b1 - A11 * b2) * D;
            cx += dx; cy += dy;
            outx = cx + halfWin; outy = cy + halfWin;
            if ((double)dx * (double)dx + (double)dy * (double)dy <= a.eps2) break;
            if (j > 0 && (double)fabsf(dx + pdx) < 0.01 && (double)fabsf(dy + pdy) < 0.01) {
                outx -= dx * 0.5f; outy -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        if (status && level == 0) {
            const float npx = outx - halfWin, npy = outy - halfWin;
            const int inx = f_floor(npx), iny = f_floor(npy);
            if (inx < -win || inx >= L.w || iny < -win || iny >= L.h) {
                status = 0;
            } else {
                wt = make_weights(npx - inx, npy - iny);
                int lx = inx - rx0, ly = iny - ry0;
                if (!(lx >= 0 && ly >= 0 && lx + win + 1 <= RW && ly + win + 1 <= RW)) {
                    __syncthreads();
                    rx0 = inx - LK_MARGIN; ry0 = iny - LK_MARGIN;
                    stage_region();
                    __syncthreads();
                    lx = LK_MARGIN; ly = LK_MARGIN;
                }
                int es = 0;
                {
                    const int rbase = ly * RW + lx;
#pragma unroll
                    for (int k = 0; k < NPX; k++) {
                        const uint8_t* r0 = &region[rbase + LK_ROFF(k)];
                        const int v = descale(__mul24((int)r0[0], wt.w00) + __mul24((int)r0[1], wt.w01) +
                                              __mul24((int)r0[RW], wt.w10) + __mul24((int)r0[RW + 1], wt.w11), 9);
                        const int diff = v - Iw[k];
                        es += LK_VALID(k) ? (diff < 0 ? -diff : diff) : 0;
                    }
                }
                const long long est = wave_sum(es);
                err = (float)est * 1.f / (float)(32 * win * win);
            }
        }
    }
#undef LK_VALID
#undef LK_POFF
#undef LK_ROFF
    if (lane == 0) {
        a.next_pts[2 * pt] = outx;
        a.next_pts[2 * pt + 1] = outy;
        a.status[pt] = (uint8_t)status;
        a.err[pt] = err;
    }
}

template <int NPX>
__global__ __launch_bounds__(64) void lk_kernel(LKArgs a) { lk_track<NPX>(a, blockIdx.x); }

// Several frames per launch: blockIdx.y selects the frame's argument block in a device table.
template <int NPX>
__global__ __launch_bounds__(64) void lk_batch_kernel(const LKArgs* __restrict__ table) {
    lk_track<NPX>(table[blockIdx.y], blockIdx.x);
}

int fill_lk_args(LKArgs& a, const LKLevel* levels, int max_level, const float* d_prev_pts, int n, const int32_t* d_n,
                 float* d_next_pts, uint8_t* d_status, float* d_err, int win, int max_iters, double eps) {
    if (!levels || max_level < 0 || max_level >= MAX_LEVELS || n < 0 || win < 3 || win > 31 ||
        !d_prev_pts || !d_next_pts || !d_status || !d_err) {
        set_last_error("pyr_lk: invalid argument (3 <= win <= 31, max_level < 8)");
        return VS_ERR_INVALID_ARG;
    }
    for (int i = 0; i <= max_level; i++) a.levels[i] = levels[i];
    for (int i = max_level + 1; i < MAX_LEVELS; i++) a.levels[i] = levels[max_level];
    a.max_level = max_level;
    a.prev_pts = d_prev_pts; a.next_pts = d_next_pts; a.status = d_status; a.err = d_err;
    a.n = n; a.d_n = d_n; a.win = win;
    // SparsePyrLKOpticalFlowImpl::calc: clamp criteria, epsilon is squared
    a.max_count = max_iters < 0 ? 0 : (max_iters > 100 ? 100 : max_iters);
    double e = eps < 0 ? 0. : (eps > 10. ? 10. : eps);
    a.eps2 = e * e;
    return VS_OK;
}

}  // namespace

// ---- batched form: one argument block per frame in a device table (see stabilizer.cpp) ----
size_t lk_item_bytes() { return sizeof(LKArgs); }

int lk_fill_item(void* host_item, const LKLevel* levels, int max_level, const float* d_prev_pts, int n,
                 const int32_t* d_n, float* d_next_pts, uint8_t* d_status, float* d_err, int win, int max_iters,
                 double eps) {
    return fill_lk_args(*static_cast<LKArgs*>(host_item), levels, max_level, d_prev_pts, n, d_n, d_next_pts, d_status,
                        d_err, win, max_iters, eps);
}

// items frames, at most n_max points each, all with the same window
int launch_pyr_lk_batch(const void* d_table, int items, int n_max, int win, hipStream_t st) {
    if (!d_table || items < 1 || items > 65535 || n_max < 0 || win < 3 || win > 31) {
        set_last_error("pyr_lk_batch: invalid argument");
        return VS_ERR_INVALID_ARG;
    }
    if (n_max == 0) return VS_OK;
    const LKArgs* t = static_cast<const LKArgs*>(d_table);
    const int npx = (win * win + 63) / 64;
    dim3 grid(n_max, items), block(64);
    if (npx <= 4) hipLaunchKernelGGL(lk_batch_kernel<4>, grid, block, 0, st, t);
    else if (npx <= 7) hipLaunchKernelGGL(lk_batch_kernel<7>, grid, block, 0, st, t);
    else hipLaunchKernelGGL(lk_batch_kernel<16>, grid, block, 0, st, t);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

int launch_pyr_lk(const LKLevel* levels, int max_level, const float* d_prev_pts, int n,
                  const int32_t* d_n, float* d_next_pts, uint8_t* d_status, float* d_err, int win,
                  int max_iters, double eps, hipStream_t st) {
    LKArgs a;
    VS_TRY(fill_lk_args(a, levels, max_level, d_prev_pts, n, d_n, d_next_pts, d_status, d_err, win, max_iters, eps));
    if (n == 0) return VS_OK;
    const int npx = (win * win + 63) / 64;
    dim3 grid(n), block(64);
    if (npx <= 4) hipLaunchKernelGGL(lk_kernel<4>, grid, block, 0, st, a);
    else if (npx <= 7) hipLaunchKernelGGL(lk_kernel<7>, grid, block, 0, st, a);
    else hipLaunchKernelGGL(lk_kernel<16>, grid, block, 0, st, a);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

// vs_op_pyr_lk: builds both pyramids and the derivative images, then tracks.
int run_pyr_lk_op(const uint8_t* d_prev, const uint8_t* d_next, size_t stride, int w, int h,
                  const float* d_prev_pts, int n, float* d_next_pts, uint8_t* d_status,
                  float* d_err, int win, int max_level, int max_iters, double eps, hipStream_t st) {
    if (!d_prev || !d_next || w <= 0 || h <= 0 || max_level < 0 || max_level >= MAX_LEVELS) {
        set_last_error("pyr_lk: invalid argument");
        return VS_ERR_INVALID_ARG;
    }
    // buildOpticalFlowPyramid: stop when the next level would not exceed the window
    int lw[MAX_LEVELS], lh[MAX_LEVELS];
    int levels = 0;
    {
        int sw = w, sh = h;
        for (int level = 0; level <= max_level; level++) {
            lw[level] = sw; lh[level] = sh;
            levels = level;
            sw = (sw + 1) / 2; sh = (sh + 1) / 2;
            if (sw <= win || sh <= win) break;
        }
    }
    size_t img_bytes = 0, der_bytes = 0;
    for (int i = 1; i <= levels; i++) img_bytes += (size_t)lw[i] * lh[i];
    for (int i = 0; i <= levels; i++) der_bytes += (size_t)lw[i] * lh[i] * 4;
    uint8_t* scratch = nullptr;
    const size_t total = 2 * img_bytes + der_bytes + 64;
    VS_HIP_TRY(hipMalloc((void**)&scratch, total));
    LKLevel L[MAX_LEVELS];
    uint8_t* pp = scratch;
    uint8_t* pn = scratch + img_bytes;
    int16_t* pd = reinterpret_cast<int16_t*>(scratch + ((2 * img_bytes + 15) & ~(size_t)15));
    int rc = VS_OK;
    for (int i = 0; i <= levels && rc == VS_OK; i++) {
        L[i].w = lw[i]; L[i].h = lh[i];
        if (i == 0) { L[i].prev = d_prev; L[i].next = d_next; L[i].stride = stride; }
        else {
            rc = launch_pyr_down(L[i - 1].prev, L[i - 1].stride, lw[i - 1], lh[i - 1], pp, lw[i], st);
            if (rc == VS_OK) rc = launch_pyr_down(L[i - 1].next, L[i - 1].stride, lw[i - 1], lh[i - 1], pn, lw[i], st);
            L[i].prev = pp; L[i].next = pn; L[i].stride = lw[i];
            pp += (size_t)lw[i] * lh[i]; pn += (size_t)lw[i] * lh[i];
        }
        if (rc == VS_OK) rc = launch_scharr(L[i].prev, L[i].stride, lw[i], lh[i], pd, st);
        L[i].deriv = pd;
        pd += (size_t)lw[i] * lh[i] * 2;
    }
    if (rc == VS_OK) rc = launch_pyr_lk(L, levels, d_prev_pts, n, nullptr, d_next_pts, d_status, d_err, win, max_iters, eps, st);
    hipError_t e = hipStreamSynchronize(st);
    (void)hipFree(scratch);
    if (rc == VS_OK && e != hipSuccess) { set_last_error(hipGetErrorString(e)); rc = VS_ERR_HIP; }
    return rc;
}

}  // namespace vsd
