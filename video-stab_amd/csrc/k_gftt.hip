// Shi-Tomasi corner detector for gfx950: device counterpart of
//   cv::goodFeaturesToTrack(gray, corners, maxCorners, quality, minDistance,
//                           noArray(), blockSize)
// as called at /root/reference/src/Stabilizer.cpp:355-357 (first frame) and
// :740-744 (every second frame; 200, 0.02, 15.0, 3).  Three launches:
//
//  1. min_eigen_kernel  - fused Sobel -> covariance -> box -> lambda_min on a
//     64x16 tile staged through LDS (gray read once, the 2 MB eigenvalue map
//     written once) + chip-wide max via one atomic per workgroup.
//  2. nms_kernel        - threshold(TOZERO, max*quality), 3x3 dilate compare,
//     wave-aggregated append of (value,index) keys.
//  3. select_kernel     - ONE workgroup: bitonic sort of the keys in LDS
//     (value descending, ties -> higher address first, i.e. OpenCV's
//     greaterThanPtr) followed by the sequential-equivalent greedy
//     min-distance selection: one wave takes 64 candidates per step, tests
//     them against an LDS cell grid, and resolves the in-step order with
//     ballot / first-set-bit, so the accepted list is exactly the one the
//     serial loop produces.
//
// Float definitions (no FMA contraction, see oracle/vso_gftt.cpp header):
//   Dx = ((r0+r2)*f1 + r1*f0), Dy = t2 - t0, f1=(float)(1/(4*bs*255)), f0=2*f1
//   box sums in double (exact), lambda = (a+c) - sqrtf((a-c)*(a-c) + b*b).
#include <cstdio>
#include <cstdlib>

#include <cstring>

#include "vs_common.h"

namespace vsd {
namespace {

constexpr int TW = 64, TH = 16, NT = 256;
constexpr int MAX_BS = 7;

constexpr int SORT_CAP = 8192;
constexpr int CELLS_MAX = 2560;
constexpr int SLOTS = 4;
constexpr int ACC_MAX = 4096;
constexpr int SEL_NT = 256;
constexpr int NMS_ROWS = 8;

__device__ __forceinline__ uint32_t order_key(float v) {
    uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(uint32_t k) {
    uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(b);
}

// BS: block size known at compile time (loops unrolled, LDS arrays sized for it), 0: any block size up to MAX_BS
template <int BS>
__device__ __forceinline__ void min_eigen_tile(const uint8_t* __restrict__ gray, size_t stride, int w,
                                               int h, int bs_arg, float f1, float* __restrict__ eig,
                                               uint32_t* __restrict__ max_key) {
    constexpr int BSM = BS ? BS : MAX_BS;
    __shared__ uint8_t g[TH + BSM - 1 + 2][TW + BSM - 1 + 2 + 2];
    __shared__ float cxx[TH + BSM - 1][TW + BSM - 1 + 1], cxy[TH + BSM - 1][TW + BSM - 1 + 1], cyy[TH + BSM - 1][TW + BSM - 1 + 1];
    __shared__ uint32_t smax[NT / 64];
    const int bs = BS ? BS : bs_arg;
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int anchor = bs / 2, hi = bs - 1 - anchor;
    const int gx0 = x0 - anchor - 1, gy0 = y0 - anchor - 1;   // image coords of g[0][0]
    const int GW = TW + bs - 1 + 2, GH = TH + bs - 1 + 2;
    const int CW = TW + bs - 1, CH = TH + bs - 1;
    const float f0 = 2.f * f1;
    // A tile whose staged rim lies inside the image (all but the image's border tiles) is staged by straight-line code: the
    // (up to) NG byte loads of a lane go out together.  As a loop over a run-time count with REFLECT_101 per element the
    // compiler made load - wait - store of every element: six dependent round trips per tile.
    const bool interior = gx0 >= 0 && gy0 >= 0 && gx0 + GW <= w && gy0 + GH <= h;       // (workgroup-uniform)
    if (interior) {
        constexpr int NG = ((TW + BSM - 1 + 2) * (TH + BSM - 1 + 2) + NT - 1) / NT;
        const uint8_t* gb = gray + ((size_t)gy0 * stride + gx0);
        uint8_t v[NG];
        int ly[NG], lx[NG];
#pragma unroll
        for (int k = 0; k < NG; k++) {
            const int i = tid + NT * k;
            ly[k] = i / GW; lx[k] = i - ly[k] * GW;
            v[k] = i < GW * GH ? gb[(size_t)ly[k] * stride + lx[k]] : (uint8_t)0;
        }
#pragma unroll
        for (int k = 0; k < NG; k++)
            if (tid + NT * k < GW * GH) g[ly[k]][lx[k]] = v[k];
    } else {
        for (int i = tid; i < GW * GH; i += NT) {
            const int ly = i / GW, lx = i - ly * GW;
            g[ly][lx] = gray[(size_t)reflect101(gy0 + ly, h) * stride + reflect101(gx0 + lx, w)];
        }
    }
    __syncthreads();
    // covariance at every position the tile's box windows touch; positions
    // outside the image take the value at the REFLECT_101 position
    for (int i = tid; i < CW * CH; i += NT) {
        const int cy = i / CW, cx = i - cy * CW;
        const int px = x0 - anchor + cx, py = y0 - anchor + cy;
        float vxx = 0.f, vxy = 0.f, vyy = 0.f;
        if (interior || (px >= -anchor && px <= w - 1 + hi && py >= -anchor && py <= h - 1 + hi)) {
            // (interior tiles: every position is its own reflection)
            const int qx = (interior ? px : reflect101(px, w)) - gx0, qy = (interior ? py : reflect101(py, h)) - gy0;   // LDS coords
            const int a00 = g[qy - 1][qx - 1], a01 = g[qy - 1][qx], a02 = g[qy - 1][qx + 1];
            const int a10 = g[qy][qx - 1], a11 = g[qy][qx], a12 = g[qy][qx + 1];
            const int a20 = g[qy + 1][qx - 1], a21 = g[qy + 1][qx], a22 = g[qy + 1][qx + 1];
            const float r0 = (float)(a02 - a00), r1 = (float)(a12 - a10), r2 = (float)(a22 - a20);
            const float dx = (r0 + r2) * f1 + r1 * f0;
            const float t0 = (float)a01 * f0 + (float)(a00 + a02) * f1;
            const float t2 = (float)a21 * f0 + (float)(a20 + a22) * f1;
            const float dy = t2 - t0;
            (void)a11;
            vxx = dx * dx; vxy = dx * dy; vyy = dy * dy;
        }
        cxx[cy][cx] = vxx; cxy[cy][cx] = vxy; cyy[cy][cx] = vyy;
    }
    __syncthreads();
    uint32_t kmax = 0;
    for (int i = tid; i < TW * TH; i += NT) {
        const int ty = i / TW, tx = i - ty * TW;
        const int x = x0 + tx, y = y0 + ty;
        if (x < w && y < h) {
            double s0 = 0, s1 = 0, s2 = 0;
            for (int j = 0; j < bs; j++) {           // constant trip counts when BS != 0: unrolled
                double r0 = 0, r1 = 0, r2 = 0;
                for (int k = 0; k < bs; k++) {
                    r0 += (double)cxx[ty + j][tx + k];
                    r1 += (double)cxy[ty + j][tx + k];
                    r2 += (double)cyy[ty + j][tx + k];
                }
                s0 += r0; s1 += r1; s2 += r2;
            }
            const float a = (float)s0 * 0.5f, b = (float)s1, c = (float)s2 * 0.5f;
            const float d = (a - c) * (a - c) + b * b;
            const float e = (a + c) - sqrtf(d);
            eig[(size_t)y * w + x] = e;
            const uint32_t k = order_key(e);
            kmax = k > kmax ? k : kmax;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t o = __shfl_xor(kmax, off, 64);
        kmax = o > kmax ? o : kmax;
    }
    if ((tid & 63) == 0) smax[tid >> 6] = kmax;
    __syncthreads();
    if (tid == 0) {
        uint32_t m = smax[0];
        for (int i = 1; i < NT / 64; i++) m = smax[i] > m ? smax[i] : m;
        atomicMax(max_key, m);
    }
}

template <int BS>
__global__ __launch_bounds__(NT) void min_eigen_kernel(const uint8_t* __restrict__ gray, size_t stride, int w,
                                                       int h, int bs, float f1, float* __restrict__ eig,
                                                       uint32_t* __restrict__ max_key) {
    min_eigen_tile<BS>(gray, stride, w, h, bs, f1, eig, max_key);
}

__device__ __forceinline__ void nms_row(const float* __restrict__ eig, int w, int h, double quality,
                                        const uint32_t* __restrict__ max_key,
                                        unsigned long long* __restrict__ cand, int cap,
                                        int32_t* __restrict__ counters) {
    // NMS_ROWS rows per workgroup: with one row each the launch is bound by the workgroup dispatch rate
    // (2000 workgroups of a few dozen instructions per image)
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const float maxv = key_to_float(*max_key);
    const float thr = (float)((double)maxv * quality);
    const int lane = threadIdx.x & 63;
    // the lane's values of all rows of the strip first (loads without conditions, from clamped positions): one round trip in front
    // of the loop instead of one per row - most positions lie under the threshold and need nothing else
    float centre[NMS_ROWS];
    {
        const int xc = min(max(x, 0), w - 1);
#pragma unroll
        for (int k = 0; k < NMS_ROWS; k++) centre[k] = eig[(size_t)min((int)blockIdx.y * NMS_ROWS + k, h - 1) * w + xc];
    }
#pragma unroll
    for (int k = 0; k < NMS_ROWS; k++) {
    const int y = blockIdx.y * NMS_ROWS + k;
    if (y >= h) break;
    bool is_cand = false;
    float v = 0.f;
    if (x >= 1 && x < w - 1 && y >= 1 && y < h - 1) {
        const float* r = eig + (size_t)y * w + x;
        v = centre[k] > thr ? centre[k] : 0.f;   // THRESH_TOZERO
        if (v != 0.f) {
            float m = v;
#pragma unroll
            for (int j = -1; j <= 1; j++)
#pragma unroll
                for (int i = -1; i <= 1; i++) {
                    const float n = r[j * w + i];
                    const float tn = n > thr ? n : 0.f;
                    m = tn > m ? tn : m;
                }
            is_cand = v == m;
        }
    }
    // one atomic per wave: the list is sorted by (value, index) afterwards, so its order is free
    const unsigned long long mask = __ballot(is_cand);
    if (!mask) continue;
    int base = 0;
    if (lane == __ffsll((long long)mask) - 1) base = atomicAdd(&counters[0], __popcll(mask));
    base = __builtin_amdgcn_readlane(base, __ffsll((long long)mask) - 1);
    if (!is_cand) continue;
    const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
    if (pos < cap) cand[pos] = ((unsigned long long)order_key(v) << 32) | (uint32_t)(y * w + x);
    else counters[2] = 1;
    }
}

__global__ __launch_bounds__(NT) void nms_kernel(const float* __restrict__ eig, int w, int h, double quality,
                                                 const uint32_t* __restrict__ max_key,
                                                 unsigned long long* __restrict__ cand, int cap,
                                                 int32_t* __restrict__ counters) {
    nms_row(eig, w, h, quality, max_key, cand, cap, counters);
}

struct SelArgs {
    const unsigned long long* cand;
    int32_t* counters;
    int cap, w, h, max_corners;
    int min_dist2_i;       // ceil(minDistance^2): integer d2 < minDistance^2  <=>  d2 < ceil(minDistance^2)
    int use_dist;          // minDistance >= 1
    int cell, gw, gh;
    float* out_pts;
    int32_t* out_count;
};

// In-place descending bitonic sort of keys[0 .. 256*E) by the 256 threads of the workgroup (thread t owns the keys
// t + 256*r).  The network is the textbook one, so the result is the same as with one compare-exchange per pair.
template <int E>
__device__ __forceinline__ void bitonic_sort_desc(unsigned long long* keys, int tid) {
    unsigned long long v[E];
#pragma unroll
    for (int r = 0; r < E; r++) v[r] = keys[tid + SEL_NT * r];
    const int n = SEL_NT * E;
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j < 64) {
#pragma unroll
                for (int r = 0; r < E; r++) {
                    const int i = tid + SEL_NT * r;
                    const unsigned int lo = (unsigned int)__shfl_xor((int)(unsigned int)v[r], j, 64);
                    const unsigned int hi = (unsigned int)__shfl_xor((int)(unsigned int)(v[r] >> 32), j, 64);
                    const unsigned long long o = ((unsigned long long)hi << 32) | lo;
                    // the lower index of a pair keeps the larger key in a descending run, the upper one the smaller
                    const bool keep_max = ((i & k) == 0) == ((i & j) == 0);
                    v[r] = keep_max ? (v[r] > o ? v[r] : o) : (v[r] < o ? v[r] : o);
                }
            } else {
                __syncthreads();                 // reads of the previous exchange through LDS are done
#pragma unroll
                for (int r = 0; r < E; r++) keys[tid + SEL_NT * r] = v[r];
                __syncthreads();
#pragma unroll
                for (int r = 0; r < E; r++) {
                    const int i = tid + SEL_NT * r;
                    const unsigned long long o = keys[i ^ j];
                    const bool keep_max = ((i & k) == 0) == ((i & j) == 0);
                    v[r] = keep_max ? (v[r] > o ? v[r] : o) : (v[r] < o ? v[r] : o);
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < E; r++) keys[tid + SEL_NT * r] = v[r];
    __syncthreads();
}

__device__ __forceinline__ void select_corners(const SelArgs& a) {
    __shared__ unsigned long long keys[SORT_CAP];
    __shared__ uint32_t cell_cnt[CELLS_MAX];
    __shared__ uint32_t cell_pts[CELLS_MAX * SLOTS];
    __shared__ uint32_t acc_xy[ACC_MAX];
    __shared__ int s_m, s_nacc, s_done, s_cnt, s_overflow;
    __shared__ unsigned long long s_upper;
    const int tid = threadIdx.x;
    int ncand = a.counters[0];
    if (ncand > a.cap) ncand = a.cap;
    const int ncells = a.gw * a.gh;
    const bool grid_ok = a.use_dist && ncells <= CELLS_MAX;
    if (grid_ok) for (int i = tid; i < ncells; i += SEL_NT) cell_cnt[i] = 0;
    if (tid == 0) { s_nacc = 0; s_done = 0; s_upper = ~0ull; s_overflow = 0; }
    __syncthreads();

    // The greedy pass below walks the candidates from the strongest on and stops at max_corners accepted ones: how deep it gets
    // depends on the picture, and sorting ALL candidates first made the launch twice as long on pictures with twice the local
    // maxima (58 -> 198 us per batch, round 3).  The list is therefore consumed in chunks - the strongest 1400 (or a few more)
    // first, then 4096 at a time - each chunk = every key in [lo, upper), chosen by a radix search on the key
    // bytes (one pass over the candidates per byte, usually three), sorted, and walked; the walk ends where the serial loop
    // would (list and order are unchanged: a chunk boundary is a position in the sorted order, nothing else).
    int want = 1400;             // (up to 2100 candidates are one chunk - sorted with eight keys per thread, as before round 4: the bench
                                 //  clips hold 750 - 1300 per detection; first chunks of 448 / 768 cost the headline clip a second chunk:
                                 //  69.8 / 64.2 us per launch against 63.0, gpurun_out/r04_t)
    while (true) {
        // ---- choose the next chunk: at least `want` (at most SORT_CAP) of the largest keys below s_upper
        const unsigned long long upper = s_upper;
        __syncthreads();
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        int remaining;
        if (upper == ~0ull) {
            remaining = ncand;               // first chunk: no counting pass needed
        } else {
            int local = 0;
            for (int i = tid; i < ncand; i += SEL_NT) local += a.cand[i] < upper ? 1 : 0;
            if (local) atomicAdd(&s_cnt, local);
            __syncthreads();
            remaining = s_cnt;
        }
        if (remaining == 0) break;
        unsigned long long lo = 0;   // keys in [lo, upper) form the chunk
        if (remaining > want + want / 2) {
            // radix search, most significant byte first: `prefix` = the bytes fixed so far, `above` = keys of the range that
            // are larger than every key with that prefix.  In each round the keys that share the prefix are counted by their
            // next byte; b = the largest byte value with above + count(byte >= b) >= want.  If that many fit the sort, the
            // chunk ends at the lower edge of bin b; otherwise the search descends into bin b (keys are unique: it ends).
            uint32_t* hist = reinterpret_cast<uint32_t*>(keys);        // (the chunk's keys are not in LDS yet)
            unsigned long long prefix = 0;
            int above = 0;
            for (int shift = 56; shift >= 0; shift -= 8) {
                __syncthreads();
                hist[tid] = 0;               // SEL_NT == 256 bins
                __syncthreads();
                for (int i = tid; i < ncand; i += SEL_NT) {
                    const unsigned long long k = a.cand[i];
                    if (k < upper && (shift == 56 || (k >> (shift + 8)) == (prefix >> (shift + 8)))) atomicAdd(&hist[(uint32_t)(k >> shift) & 255u], 1u);
                }
                __syncthreads();
                if (tid == 0) {
                    int acc = above, bsel = 0;
                    for (int v = 255; v >= 0; v--) {
                        if (acc + (int)hist[v] >= want) { bsel = v; break; }
                        acc += (int)hist[v];
                    }
                    s_m = bsel;              // the byte
                    s_cnt = acc;             // keys above bin bsel
                }
                __syncthreads();
                const int bsel = s_m, over = s_cnt, inbin = (int)hist[bsel];
                prefix |= (unsigned long long)bsel << shift;
                if (over + inbin <= SORT_CAP) { lo = prefix; break; }
                above = over;
                lo = prefix;                 // (after the last byte the bin holds one key: the chunk has exactly `want` keys)
            }
        }
        __syncthreads();
        if (tid == 0) s_m = 0;
        __syncthreads();
        for (int i = tid; i < ncand; i += SEL_NT) {
            const unsigned long long k = a.cand[i];
            if (k >= lo && k < upper) {
                const int p = atomicAdd(&s_m, 1);
                if (p < SORT_CAP) keys[p] = k;
            }
        }
        __syncthreads();
        const int m = s_m < SORT_CAP ? s_m : SORT_CAP;
        int np2 = SEL_NT;                    // at least one key per thread
        while (np2 < m) np2 <<= 1;
        for (int i = m + tid; i < np2; i += SEL_NT) keys[i] = 0ull;
        __syncthreads();
        // ---- bitonic sort, descending: every thread keeps its np2/256 keys in registers; exchanges at a distance below
        // 64 stay inside the wave (shuffles, no barrier), the others go through LDS
        switch (np2 / SEL_NT) {
            case 1: bitonic_sort_desc<1>(keys, tid); break;
            case 2: bitonic_sort_desc<2>(keys, tid); break;
            case 4: bitonic_sort_desc<4>(keys, tid); break;
            case 8: bitonic_sort_desc<8>(keys, tid); break;
            case 16: bitonic_sort_desc<16>(keys, tid); break;
            default: bitonic_sort_desc<32>(keys, tid); break;
        }
        // ---- decode the sorted keys in place: key -> (x | y<<16) | (xc | yc<<16) << 32
        // (integer divisions by w and by the cell size happen here, in parallel)
        const unsigned long long last_key = m > 0 ? keys[m - 1] : 0ull;
        __syncthreads();
        for (int i = tid; i < m; i += SEL_NT) {
            const int idx = (int)(uint32_t)keys[i];
            const int y = idx / a.w, x = idx - y * a.w;
            const int xc = x / a.cell, yc = y / a.cell;
            keys[i] = (unsigned long long)((uint32_t)x | ((uint32_t)y << 16)) |
                      ((unsigned long long)((uint32_t)xc | ((uint32_t)yc << 16)) << 32);
        }
        __syncthreads();
        // ---- greedy selection by wave 0 (sequential-equivalent)
        if (tid < 64) {
            const int lane = tid;
            int nacc = s_nacc;
            volatile uint32_t* vcnt = cell_cnt;
            volatile uint32_t* vpts = cell_pts;
            volatile uint32_t* vacc = acc_xy;
            const int md2 = a.min_dist2_i;
            bool full = false;
            for (int base = 0; base < m && !full; base += 64) {
                const int i = base + lane;
                const bool has = i < m;
                const unsigned long long e = has ? keys[i] : 0ull;
                const uint32_t xy = (uint32_t)e;
                const int x = (int)(xy & 0xFFFFu), y = (int)(xy >> 16);
                const int xc = (int)((uint32_t)(e >> 32) & 0xFFFFu), yc = (int)((uint32_t)(e >> 48));
                bool ok = has;
                if (a.use_dist && ok) {
                    if (grid_ok) {
                        // counts of the 3x3 neighbourhood first (independent LDS reads), then
                        // the few occupied cells
                        int cidx[9];
                        uint32_t cnt[9];
#pragma unroll
                        for (int q = 0; q < 9; q++) {
                            const int cx = xc + (q % 3) - 1, cy = yc + (q / 3) - 1;
                            const bool in = cx >= 0 && cx < a.gw && cy >= 0 && cy < a.gh;
                            cidx[q] = in ? cy * a.gw + cx : -1;
                        }
#pragma unroll
                        for (int q = 0; q < 9; q++) cnt[q] = cidx[q] >= 0 ? vcnt[cidx[q]] : 0u;
#pragma unroll
                        for (int q = 0; q < 9; q++) {
                            const int c = cnt[q] < (uint32_t)SLOTS ? (int)cnt[q] : SLOTS;
                            for (int sl = 0; sl < c; sl++) {
                                const uint32_t p = vpts[cidx[q] * SLOTS + sl];
                                const int dx = x - (int)(p & 0xFFFFu), dy = y - (int)(p >> 16);
                                if (dx * dx + dy * dy < md2) ok = false;
                            }
                        }
                    } else {
                        for (int sl = 0; sl < nacc && ok; sl++) {
                            const uint32_t p = vacc[sl];
                            const int dx = x - (int)(p & 0xFFFFu), dy = y - (int)(p >> 16);
                            if (dx * dx + dy * dy < md2) ok = false;
                        }
                    }
                }
                // in-step order: visit the surviving lanes from the strongest on; a visited lane is
                // accepted and knocks out the later lanes it conflicts with (registers only);
                // grid / output updates of the accepted lanes happen together after the loop
                unsigned long long mask = __ballot(ok);
                unsigned long long accepted = 0ull;
                const int nacc0 = nacc;
                while (mask) {
                    const int j = __ffsll((long long)mask) - 1;     // wave-uniform
                    accepted |= 1ull << j;
                    nacc++;
                    if (nacc == a.max_corners) { full = true; break; }
                    if (a.use_dist) {
                        // a pair closer than minDistance always sits in neighbouring cells
                        // (|dx| < minDistance <= cell + 0.5), so the distance test alone decides
                        const uint32_t pj = (uint32_t)__builtin_amdgcn_readlane((int)xy, j);
                        const int dx = x - (int)(pj & 0xFFFFu), dy = y - (int)(pj >> 16);
                        if (dx * dx + dy * dy < md2) ok = false;
                    }
                    if (lane == j) ok = false;
                    mask = __ballot(ok);
                }
                if ((accepted >> lane) & 1ull) {
                    const int pos = nacc0 + __popcll(accepted & ((1ull << lane) - 1ull));
                    a.out_pts[2 * pos] = (float)x;
                    a.out_pts[2 * pos + 1] = (float)y;
                    const uint32_t packed = (uint32_t)x | ((uint32_t)y << 16);
                    if (pos < ACC_MAX) acc_xy[pos] = packed;
                    if (grid_ok) {
                        const int c = yc * a.gw + xc;
                        const uint32_t slot = atomicAdd(&cell_cnt[c], 1u);
                        if (slot < (uint32_t)SLOTS) cell_pts[c * SLOTS + slot] = packed;
                        else s_overflow = 1;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
            }
            if (lane == 0) {
                s_nacc = nacc;
                if (full) s_done = 1;
                if (m > 0) s_upper = last_key;   // smallest key of this chunk
                else s_done = 1;
            }
        }
        __syncthreads();
        if (s_done) break;
        want = SORT_CAP / 2;
    }
    __syncthreads();
    if (tid == 0) {
        *a.out_count = s_nacc;
        a.counters[3] = s_nacc;
        if (s_overflow) a.counters[2] = 2;
    }
}

__global__ __launch_bounds__(SEL_NT) void select_kernel(SelArgs a) { select_corners(a); }

// ---- several images per launch (blockIdx.z / blockIdx.x selects the image's block in a device table)
struct GfttItem {
    const uint8_t* gray;
    size_t stride;
    int w, h, bs;
    float f1;
    double quality;
    float* eig;
    int32_t* counters;
    SelArgs sel;
};

__global__ __launch_bounds__(64) void gftt_zero_batch_kernel(const GfttItem* __restrict__ table) {
    if (threadIdx.x < 16) table[blockIdx.x].counters[threadIdx.x] = 0;
}
template <int BS>
__global__ __launch_bounds__(NT) void min_eigen_batch_kernel(const GfttItem* __restrict__ table) {
    const GfttItem& it = table[blockIdx.z];
    min_eigen_tile<BS>(it.gray, it.stride, it.w, it.h, it.bs, it.f1, it.eig, (uint32_t*)&it.counters[1]);
}
__global__ __launch_bounds__(NT) void nms_batch_kernel(const GfttItem* __restrict__ table) {
    const GfttItem& it = table[blockIdx.z];
    nms_row(it.eig, it.w, it.h, it.quality, (const uint32_t*)&it.counters[1], (unsigned long long*)it.sel.cand, it.sel.cap,
            it.counters);
}
__global__ __launch_bounds__(SEL_NT) void select_batch_kernel(const GfttItem* __restrict__ table) {
    select_corners(table[blockIdx.x].sel);
}

bool gftt_bad_args(const uint8_t* d_gray, const float* d_pts, const int32_t* d_count, int w, int h, int max_corners,
                   int block_size, const GfttWork& wk) {
    return !d_gray || !d_pts || !d_count || w < 3 || h < 3 || w > 65535 || h > 65535 || max_corners <= 0 ||
           max_corners > ACC_MAX || block_size < 1 || block_size > MAX_BS || !wk.eig || !wk.cand || !wk.counters ||
           wk.cap <= 0;
}

void fill_sel_args(SelArgs& a, int w, int h, int max_corners, double min_distance, const GfttWork& wk, float* d_pts,
                   int32_t* d_count) {
    a.cand = (const unsigned long long*)wk.cand;
    a.counters = wk.counters;
    a.cap = wk.cap; a.w = w; a.h = h; a.max_corners = max_corners;
    a.use_dist = min_distance >= 1 ? 1 : 0;
    a.min_dist2_i = (int)std::ceil(std::min(min_distance * min_distance, 2.0e9));
    a.cell = a.use_dist ? (int)lrint(min_distance) : 1;
    a.gw = (w + a.cell - 1) / a.cell;
    a.gh = (h + a.cell - 1) / a.cell;
    a.out_pts = d_pts; a.out_count = d_count;
}

}  // namespace

size_t gftt_item_bytes() { return sizeof(GfttItem); }

int gftt_fill_item(void* host_item, const uint8_t* d_gray, size_t stride, int w, int h, int max_corners, double quality,
                   double min_distance, int block_size, const GfttWork& wk, float* d_pts, int32_t* d_count) {
    if (!host_item || gftt_bad_args(d_gray, d_pts, d_count, w, h, max_corners, block_size, wk)) {
        set_last_error("gftt: invalid argument (1 <= blockSize <= 7, 0 < maxCorners <= 4096)");
        return VS_ERR_INVALID_ARG;
    }
    GfttItem& it = *static_cast<GfttItem*>(host_item);
    memset(&it, 0, sizeof it);
    it.gray = d_gray; it.stride = stride; it.w = w; it.h = h; it.bs = block_size;
    double scale = (double)(1 << 2) * block_size * 255.0;
    it.f1 = (float)(1.0 / scale);
    it.quality = quality;
    it.eig = wk.eig; it.counters = wk.counters;
    fill_sel_args(it.sel, w, h, max_corners, min_distance, wk, d_pts, d_count);
    return VS_OK;
}

// items images of one size (w x h), one launch per stage
// block_size: the block size every item of the table was filled with
// what: 0 = the whole detection; or its pieces, in this order: 1 = the reset of the counters (it does not read the images:
// the caller may issue it before the stream waits for them); 2 = min-eigenvalue map and non-maximum suppression (the wide
// launches; 4 and 5 = the two apart); 3 = the selection (one workgroup per image: it hardly occupies the GPU)
int launch_gftt_batch(const void* d_table, int items, int w, int h, int block_size, hipStream_t st, int what) {
    if (!d_table || items < 1 || items > 65535 || w < 3 || h < 3 || block_size < 1 || block_size > MAX_BS || what < 0 || what > 5) { set_last_error("gftt_batch: invalid argument"); return VS_ERR_INVALID_ARG; }
    const GfttItem* t = static_cast<const GfttItem*>(d_table);
    if (what == 0 || what == 1) hipLaunchKernelGGL(gftt_zero_batch_kernel, dim3(items), dim3(64), 0, st, t);
    if (what == 0 || what == 2 || what == 4) {
        const dim3 eg((w + TW - 1) / TW, (h + TH - 1) / TH, items);
        if (block_size == 3) hipLaunchKernelGGL(min_eigen_batch_kernel<3>, eg, dim3(NT), 0, st, t);
        else hipLaunchKernelGGL(min_eigen_batch_kernel<0>, eg, dim3(NT), 0, st, t);
    }
    if (what == 0 || what == 2 || what == 5)
        hipLaunchKernelGGL(nms_batch_kernel, dim3((w + NT - 1) / NT, (h + NMS_ROWS - 1) / NMS_ROWS, items), dim3(NT), 0, st, t);
    if (what == 0 || what == 3) hipLaunchKernelGGL(select_batch_kernel, dim3(items), dim3(SEL_NT), 0, st, t);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

size_t gftt_work_bytes(int w, int h, int cap) {
    return (size_t)w * h * 4 + (size_t)cap * 8 + 64 + 256;
}

void gftt_work_carve(void* base, int w, int h, int cap, GfttWork* out) {
    uint8_t* p = (uint8_t*)base;
    out->eig = (float*)p;
    p += ((size_t)w * h * 4 + 63) & ~(size_t)63;
    out->cand = (uint64_t*)p;
    p += (size_t)cap * 8;
    out->counters = (int32_t*)p;
    out->cap = cap;
}

int launch_gftt(const uint8_t* d_gray, size_t stride, int w, int h, int max_corners, double quality,
                double min_distance, int block_size, const GfttWork& wk, float* d_pts,
                int32_t* d_count, hipStream_t st) {
    if (gftt_bad_args(d_gray, d_pts, d_count, w, h, max_corners, block_size, wk)) {
        set_last_error("gftt: invalid argument (1 <= blockSize <= 7, 0 < maxCorners <= 4096)");
        return VS_ERR_INVALID_ARG;
    }
    VS_HIP_TRY(hipMemsetAsync(wk.counters, 0, 64, st));
    double scale = (double)(1 << 2) * block_size * 255.0;
    scale = 1.0 / scale;
    const float f1 = (float)scale;
    dim3 g1((w + TW - 1) / TW, (h + TH - 1) / TH);
    if (block_size == 3)
        hipLaunchKernelGGL(min_eigen_kernel<3>, g1, dim3(NT), 0, st, d_gray, stride, w, h, block_size, f1, wk.eig, (uint32_t*)&wk.counters[1]);
    else
        hipLaunchKernelGGL(min_eigen_kernel<0>, g1, dim3(NT), 0, st, d_gray, stride, w, h, block_size, f1, wk.eig, (uint32_t*)&wk.counters[1]);
    dim3 g2((w + NT - 1) / NT, (h + NMS_ROWS - 1) / NMS_ROWS);
    hipLaunchKernelGGL(nms_kernel, g2, dim3(NT), 0, st, wk.eig, w, h, quality, (const uint32_t*)&wk.counters[1],
                       (unsigned long long*)wk.cand, wk.cap, wk.counters);
    SelArgs a;
    fill_sel_args(a, w, h, max_corners, min_distance, wk, d_pts, d_count);
    hipLaunchKernelGGL(select_kernel, dim3(1), dim3(SEL_NT), 0, st, a);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

int run_gftt_op(const uint8_t* d_gray, size_t stride, int w, int h, int max_corners, double quality,
                double min_distance, int block_size, float* d_pts, int32_t* d_count, float* d_eig,
                hipStream_t st) {
    if (w <= 0 || h <= 0) { set_last_error("gftt: invalid size"); return VS_ERR_INVALID_ARG; }
    const int cap = w * h / 4 + 64;   // a 3x3 local maximum excludes its 8 neighbours
    void* scratch = nullptr;
    VS_HIP_TRY(hipMalloc(&scratch, gftt_work_bytes(w, h, cap)));
    GfttWork wk;
    gftt_work_carve(scratch, w, h, cap, &wk);
    int rc = launch_gftt(d_gray, stride, w, h, max_corners, quality, min_distance, block_size, wk, d_pts, d_count, st);
    if (rc == VS_OK && d_eig) {
        hipError_t e = hipMemcpyAsync(d_eig, wk.eig, (size_t)w * h * 4, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); rc = VS_ERR_HIP; }
    }
    hipError_t e = hipStreamSynchronize(st);
    (void)hipFree(scratch);
    if (rc == VS_OK && e != hipSuccess) { set_last_error(hipGetErrorString(e)); rc = VS_ERR_HIP; }
    return rc;
}

}  // namespace vsd
