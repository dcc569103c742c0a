// Roll correction for gfx950: device counterpart of vs::RollCorrection::autoCorrectRoll
// (/root/reference/src/RollCorrection.cpp:16-155).  The reference runs this stage on
// cv::cuda (resize, cvtColor, CannyEdgeDetector, HoughLinesDetector, remap) and has no
// CPU branch; this build follows the CPU OpenCV definitions of the same operators
// (see oracle/vso_roll.cpp), so every intermediate is bit-comparable with the oracle:
//
//   resize x0.25 + BGR2GRAY (fused, k_gray.hip)
//   sobel_kernel      3x3 Sobel, BORDER_REPLICATE -> dx,dy (int16) and L1 magnitude
//   canny_nms_kernel  non-maximum suppression with the tan(22.5) fixed-point sectors,
//                     double threshold -> bit planes E ("edge") and C ("maybe"), 64 pixels per word
//   canny_hyst_band_kernel  growth of the edge set through the "maybe" pixels on the bit planes: a band
//                     of 62 rows x up to 1024 pixels is iterated to its own fixed point in registers
//                     (wave = word, lane = row); passes over the image repeat until a pass changes
//                     nothing (chains that cross band borders)
//   edge_list_bits_kernel  edge pixels as a packed list straight from the bit plane, one atomic per wave
//                     (edge_list_kernel: the same from a byte edge map, for vs_op_hough_lines)
//   hough_accum_lds_kernel  one workgroup per angle, that angle's accumulator row in LDS; float rho as
//                     cv::HoughLines
//   hough_peaks_kernel  local maxima above the threshold -> (votes, index) keys
//   hough_select_kernel ONE workgroup: bitonic sort (votes desc, index asc) and the
//                     sequential angle filter / mean of RollCorrection.cpp:109-125
//   warp_affine_kernel<3> with BORDER_REPLICATE (k_warp.hip) for the rotation.
// The smoothed angle (EMA, clamp, decay) is host state of the vs_roll object; a frame needs ONE wait of the host:
// the 24-byte result comes back together with the flag that says whether four hysteresis passes were enough
// (if not, the growth is finished and the line search redone; the reference synchronises several times per frame).
#include <cmath>
#include <cstring>
#include <new>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <type_traits>
#include <vector>

#include "vs_common.h"

namespace vsd {

int launch_warp_affine_inv(const uint8_t* d_src, size_t sstride, int sw, int sh, uint8_t* d_dst, size_t dstride,
                           int dw, int dh, int cn, const double* h_Minv, int border, hipStream_t st);
int launch_warp_affine_list_inv(const uint8_t* const* srcs, uint8_t* const* dsts, int n, size_t sstride, int sw, int sh, size_t dstride, int dw, int dh,
                                int cn, const double* h_Minv, int border, hipStream_t st);
int launch_warp_nv12_list_inv(const uint8_t* const* ys, uint8_t* const* yd, int n, size_t sstride, size_t dstride, int w, int h, size_t src_uv,
                              size_t dst_uv, const double* h_MinvY, const double* h_MinvUV, int border, hipStream_t st);

namespace {

constexpr int NT = 256;
// Several frames per launch (the asynchronous NV12 path analyses eight frames with one launch per stage): blockIdx.z selects the
// frame, whose buffers lie `fb` bytes behind those of frame 0 (one work area per frame, laid out alike).  fb = 0: one frame.
template <typename T> __device__ __forceinline__ T* frame_ptr(T* p, size_t fb) {
    return reinterpret_cast<T*>(reinterpret_cast<uintptr_t>(p) + (size_t)blockIdx.z * fb);
}

constexpr int HOUGH_CAP = 8192;      // peaks kept for the sort (cv::HoughLines has no cap)

__global__ __launch_bounds__(NT) void sobel_kernel(const uint8_t* __restrict__ g, size_t stride, int w, int h,
                                                   short2* __restrict__ dxy, int* __restrict__ mag, int mw, size_t fb) {
    g = frame_ptr(g, fb); dxy = frame_ptr(dxy, fb); mag = frame_ptr(mag, fb);
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h) return;
    const int xm = x > 0 ? x - 1 : 0, xp = x < w - 1 ? x + 1 : w - 1;
    const uint8_t* r0 = g + (size_t)(y > 0 ? y - 1 : 0) * stride;
    const uint8_t* r1 = g + (size_t)y * stride;
    const uint8_t* r2 = g + (size_t)(y < h - 1 ? y + 1 : h - 1) * stride;
    const int a = r0[xm], b = r0[x], c = r0[xp], d = r1[xm], f = r1[xp], g0 = r2[xm], h0 = r2[x], i = r2[xp];
    const int dx = (c + 2 * f + i) - (a + 2 * d + g0);
    const int dy = (g0 + 2 * h0 + i) - (a + 2 * b + c);
    dxy[(size_t)y * w + x] = make_short2((short)dx, (short)dy);
    mag[(size_t)(y + 1) * mw + x + 1] = abs(dx) + abs(dy);
}

// mag is framed with one pixel of zeros, row pitch mw = w + 2.  The result goes out as two bit planes: word k of
// row y holds pixels 64k .. 64k+63 (bit b = pixel 64k + b); E = "edge" (above the high threshold), C = "maybe"
// (local maximum between the thresholds).  One wave covers 64 consecutive pixels, so a ballot is the word.
typedef unsigned long long u64;

__global__ __launch_bounds__(NT) void canny_nms_kernel(const short2* __restrict__ dxy, const int* __restrict__ mag,
                                                       int w, int h, int mw, int low, int high, u64* __restrict__ E,
                                                       u64* __restrict__ C, int wpr, size_t fb) {
    dxy = frame_ptr(dxy, fb); mag = frame_ptr(mag, fb); E = frame_ptr(E, fb); C = frame_ptr(C, fb);
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    int out = 1;
    if (x < w) {
        const int p = (y + 1) * mw + x + 1;
        const int m = mag[p];
        if (m > low) {
            const short2 d = dxy[(size_t)y * w + x];
            const int xs = d.x, ys = d.y;
            const int ax = abs(xs), ay = abs(ys) << 15;
            const int tg22x = ax * 13573;
            bool is_max;
            if (ay < tg22x) {
                is_max = m > mag[p - 1] && m >= mag[p + 1];
            } else {
                const int tg67x = tg22x + (ax << 16);
                if (ay > tg67x) {
                    is_max = m > mag[p - mw] && m >= mag[p + mw];
                } else {
                    const int s = (xs ^ ys) < 0 ? 1 : -1;
                    is_max = m > mag[p - mw - s] && m > mag[p + mw + s];
                }
            }
            if (is_max) {
                out = m > high ? 2 : 0;
            }
        }
    }
    const u64 be = __ballot(out == 2), bc = __ballot(out == 0);
    if ((threadIdx.x & 63) == 0 && x < w) {
        E[(size_t)y * wpr + (x >> 6)] = be;
        C[(size_t)y * wpr + (x >> 6)] = bc;
    }
}

constexpr int HB_ROWS = 62;      // rows a hysteresis band owns (lanes 1..62; lanes 0 and 63 hold the rows next to it)
constexpr int HB_WORDS = 16;     // words (of 64 pixels) one workgroup spans: one wave per word

// all candidate bits of `c` that a run of candidates links to a bit of `g` along the row (Kogge-Stone fill)
__device__ __forceinline__ u64 fill_row(u64 g, u64 c) {
    u64 a = g, p = c;
    a |= p & (a << 1);  p &= p << 1;
    a |= p & (a << 2);  p &= p << 2;
    a |= p & (a << 4);  p &= p << 4;
    a |= p & (a << 8);  p &= p << 8;
    a |= p & (a << 16); p &= p << 16;
    a |= p & (a << 32);
    u64 b = g; p = c;
    b |= p & (b >> 1);  p &= p >> 1;
    b |= p & (b >> 2);  p &= p >> 2;
    b |= p & (b >> 4);  p &= p >> 4;
    b |= p & (b >> 8);  p &= p >> 8;
    b |= p & (b >> 16); p &= p >> 16;
    b |= p & (b >> 32);
    return a | b;
}

// the same along the columns (lane = row): all candidate bits a vertical run of candidates links to a bit of `g`
__device__ __forceinline__ u64 fill_col(u64 g, u64 c, int lane) {
    u64 a = g, p = c;
#pragma unroll
    for (int n = 1; n < 64; n *= 2) {
        const u64 ta = __shfl_up(a, n), tp = __shfl_up(p, n);
        if (lane >= n) { a |= p & ta; p &= tp; } else { p = 0; }
    }
    u64 b = g; p = c;
#pragma unroll
    for (int n = 1; n < 64; n *= 2) {
        const u64 tb = __shfl_down(b, n), tp = __shfl_down(p, n);
        if (lane + n < 64) { b |= p & tb; p &= tp; } else { p = 0; }
    }
    return a | b;
}

// One pass of the growth of the edge set through the "maybe" pixels.  A workgroup owns a band of 62 rows by up to
// 16 words: wave = word, lane = row, so a row of 64 pixels is one register and the 8-neighbourhood is shifts plus
// a lane shuffle.  The band is iterated to its own fixed point (the waves trade their border columns through LDS);
// the rows and words around it are read as the neighbours left them.  Growth is monotone and its fixed point
// unique, so concurrent bands can only help each other; passes repeat until one changes nothing.
__global__ __launch_bounds__(64 * HB_WORDS) void canny_hyst_band_kernel(u64* __restrict__ E, const u64* __restrict__ C,
                                                                        int wpr, int h, int* __restrict__ changed,
                                                                        const int* __restrict__ before, size_t fb) {
    E = frame_ptr(E, fb); C = frame_ptr(C, fb); changed = frame_ptr(changed, fb);
    if (before) before = frame_ptr(before, fb);
    if (before && *before == 0) return;      // the pass before this one changed nothing: neither will this one
    __shared__ uint8_t s_lo[HB_WORDS][64], s_hi[HB_WORDS][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    const int k = blockIdx.x * nwv + wv, y = blockIdx.y * HB_ROWS - 1 + lane;
    const bool in_img = k < wpr && y >= 0 && y < h;
    const bool own = in_img && lane >= 1 && lane <= HB_ROWS;
    const size_t at = in_img ? (size_t)y * wpr + k : 0;
    u64 e = in_img ? E[at] : 0;
    const u64 c = own ? C[at] : 0;
    const u64 e0 = e;
    // columns next to the workgroup's span: fixed for this pass
    u64 gl = 0, gr = 0;
    if (in_img && wv == 0 && k > 0) gl = E[at - 1] >> 63;
    if (in_img && wv == nwv - 1 && k + 1 < wpr) gr = (E[at + 1] & 1) << 63;
    // a wave whose own rows and whose neighbours' columns stood still in the last round has nothing to do in
    // this one: the tail of a pass, when one chain is still growing, costs one wave and not sixteen
    bool active = true;
    u64 pl = ~0ull, pr = ~0ull;
    for (;;) {
        s_lo[wv][lane] = (uint8_t)(e & 1);
        s_hi[wv][lane] = (uint8_t)(e >> 63);
        __syncthreads();
        const u64 l = wv > 0 ? (u64)s_hi[wv - 1][lane] : gl;
        const u64 r = wv < nwv - 1 ? (u64)s_lo[wv + 1][lane] << 63 : gr;
        const bool moved = __ballot(l != pl || r != pr) != 0;
        pl = l; pr = r;
        bool grew = false;
        if (active || moved) {
            const u64 hd = e | (e << 1) | (e >> 1) | l | r;       // the row, spread by one pixel to both sides
            u64 up = __shfl_up(hd, 1), dn = __shfl_down(hd, 1);
            if (lane == 0) up = 0;
            if (lane == 63) dn = 0;
            const u64 ne = fill_col(fill_row(e | (c & (hd | up | dn)), c), c, lane);
            grew = ne != e;
            e = ne;
            active = __ballot(grew) != 0;
        }
        if (!__syncthreads_or(grew)) break;
    }
    const bool mine = own && e != e0;
    if (mine) E[at] = e;
    if (__syncthreads_or(mine) && threadIdx.x == 0) *changed = 1;
}

__global__ __launch_bounds__(NT) void canny_out_kernel(const u64* __restrict__ E, int wpr, int w, int h,
                                                       uint8_t* __restrict__ edges, size_t estride) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h) return;
    edges[(size_t)y * estride + x] = (E[(size_t)y * wpr + (x >> 6)] >> (x & 63)) & 1 ? 255 : 0;
}

// Edge pixels as packed (y << 16 | x), in no particular order (the votes do not depend on it).  A thread looks at
// eight pixels, a wave reserves its share of the list with one atomic.
constexpr int EL_PX = 8;

__global__ __launch_bounds__(NT) void edge_list_kernel(const uint8_t* __restrict__ edges, size_t stride, int w, int h,
                                                       int* __restrict__ list, int* __restrict__ counters) {
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * EL_PX, y = blockIdx.y;
    const uint8_t* row = edges + (size_t)y * stride;
    unsigned on = 0;
    if (x0 + EL_PX <= w && (((uintptr_t)(row + x0)) & 7) == 0) {
        const uint2 v = *(const uint2*)(row + x0);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            on |= ((v.x >> (8 * k)) & 255u ? 1u : 0u) << k;
            on |= ((v.y >> (8 * k)) & 255u ? 1u : 0u) << (4 + k);
        }
    } else {
        for (int k = 0; k < EL_PX; k++)
            if (x0 + k < w && row[x0 + k] != 0) on |= 1u << k;
    }
    const int mine = __popc(on), lane = threadIdx.x & 63;
    if (__ballot(mine != 0) == 0) return;
    int incl = mine;                             // inclusive scan over the wave
#pragma unroll
    for (int d = 1; d < 64; d *= 2) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    int base = 0;
    if (lane == 63) base = atomicAdd(&counters[1], incl);
    base = __shfl(base, 63) + incl - mine;
    while (on) {
        const int k = __builtin_ctz(on);
        on &= on - 1;
        list[base++] = (y << 16) | (x0 + k);
    }
}

// The same list straight from the Canny bit plane E (the roll stage never needs the edge map as bytes): a lane
// takes one word of 64 pixels.
__global__ __launch_bounds__(NT) void edge_list_bits_kernel(const u64* __restrict__ E, int wpr, int h,
                                                            int* __restrict__ list, int* __restrict__ counters, size_t fb) {
    E = frame_ptr(E, fb); list = frame_ptr(list, fb); counters = frame_ptr(counters, fb);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    u64 m = idx < wpr * h ? E[idx] : 0ull;
    const int mine = __popcll(m);
    if (__ballot(mine != 0) == 0) return;
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d *= 2) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    int base = 0;
    if (lane == 63) base = atomicAdd(&counters[1], incl);
    base = __shfl(base, 63) + incl - mine;
    const int y = idx / wpr, x0 = (idx - y * wpr) * 64;
    while (m) {
        const int b = __builtin_ctzll(m);
        m &= m - 1;
        list[base++] = (y << 16) | (x0 + b);
    }
}

// Votes of every (edge pixel, angle) pair.  One workgroup per angle keeps that angle's row of the accumulator in
// LDS, runs over the edge list and stores the row: no atomics on HBM and no clearing of the accumulator between
// frames (rows are stored whole; the frame of zeros around them is never written).
__global__ __launch_bounds__(1024) void hough_accum_lds_kernel(const int* __restrict__ list, const int* __restrict__ counters,
                                                               const float* __restrict__ tabSin,
                                                               const float* __restrict__ tabCos, int numrho,
                                                               int* __restrict__ accum, size_t fb) {
    extern __shared__ int s_row[];
    list = frame_ptr(list, fb); counters = frame_ptr(counters, fb); accum = frame_ptr(accum, fb);
    const int n = blockIdx.x, n_edges = counters[1];
    for (int r = threadIdx.x; r < numrho; r += blockDim.x) s_row[r] = 0;
    __syncthreads();
    const float cs = tabCos[n], sn = tabSin[n];
    const int shift = (numrho - 1) / 2;
    for (int e = threadIdx.x; e < n_edges; e += blockDim.x) {
        const int idx = list[e];
        const int i = idx >> 16, j = idx & 0xFFFF;
        const int r = f_round((float)j * cs + (float)i * sn) + shift;
        atomicAdd(&s_row[r], 1);
    }
    __syncthreads();
    int* out = accum + (size_t)(n + 1) * (numrho + 2) + 1;
    for (int r = threadIdx.x; r < numrho; r += blockDim.x) out[r] = s_row[r];
}

// The same with atomics on a zeroed accumulator in HBM, for accumulator rows that do not fit LDS.
__global__ __launch_bounds__(NT) void hough_accum_kernel(const int* __restrict__ list, const int* __restrict__ counters,
                                                         const float* __restrict__ tabSin,
                                                         const float* __restrict__ tabCos, int numangle, int numrho,
                                                         int* __restrict__ accum, size_t fb) {
    list = frame_ptr(list, fb); counters = frame_ptr(counters, fb); accum = frame_ptr(accum, fb);
    const int n_edges = counters[1];
    const long long total = (long long)n_edges * numangle;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(t / numangle), n = (int)(t - (long long)e * numangle);
        const int idx = list[e];
        const int i = idx >> 16, j = idx & 0xFFFF;
        int r = f_round((float)j * tabCos[n] + (float)i * tabSin[n]);
        r += (numrho - 1) / 2;
        atomicAdd(&accum[(size_t)(n + 1) * (numrho + 2) + r + 1], 1);
    }
}

__global__ __launch_bounds__(NT) void hough_peaks_kernel(const int* __restrict__ accum, int numangle, int numrho,
                                                         int threshold, unsigned long long* __restrict__ keys,
                                                         int* __restrict__ counters, size_t fb) {
    accum = frame_ptr(accum, fb); keys = frame_ptr(keys, fb); counters = frame_ptr(counters, fb);
    const int r = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
    if (r >= numrho || n >= numangle) return;
    const int base = (n + 1) * (numrho + 2) + r + 1;
    const int v = accum[base];
    if (v > threshold && v > accum[base - 1] && v >= accum[base + 1] && v > accum[base - numrho - 2] &&
        v >= accum[base + numrho + 2]) {
        const int pos = atomicAdd(&counters[2], 1);
        // sort key: votes descending, then index ascending (hough_cmp_gt)
        if (pos < HOUGH_CAP) keys[pos] = ((unsigned long long)(unsigned)v << 32) | (unsigned)(0x7FFFFFFF - base);
        else counters[3] = 1;
    }
}

struct RollResult {
    double sum_deg;      // sum of the accepted angles (degrees), in cv::HoughLines order
    int count;           // accepted lines
    int n_lines;         // lines found
    int overflow;
    int pad;
};

__global__ __launch_bounds__(1024) void hough_select_kernel(unsigned long long* __restrict__ keys,
                                                            const int* __restrict__ counters, int numrho, float rho,
                                                            float theta, double amin, double amax,
                                                            float* __restrict__ lines_out, int max_out,
                                                            RollResult* __restrict__ res, size_t fb) {
    __shared__ unsigned long long sk[HOUGH_CAP];
    keys = frame_ptr(keys, fb); counters = frame_ptr(counters, fb); lines_out = frame_ptr(lines_out, fb); res = frame_ptr(res, fb);
    const int tid = threadIdx.x;
    int n = counters[2];
    n = n < HOUGH_CAP ? n : HOUGH_CAP;
    int np2 = 2;
    while (np2 < n) np2 <<= 1;
    for (int i = tid; i < np2; i += 1024) sk[i] = i < n ? keys[i] : 0ull;
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < np2; i += 1024) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = sk[i], b = sk[ixj];
                    const bool desc = (i & k) == 0;
                    if (desc ? (a < b) : (a > b)) { sk[i] = b; sk[ixj] = a; }
                }
            }
            __syncthreads();
        }
    // lines in cv::HoughLines order; (rho, theta) as it computes them
    const double scale = 1. / (numrho + 2);
    for (int i = tid; i < n && i < max_out; i += 1024) {
        const int idx = 0x7FFFFFFF - (int)(unsigned)(sk[i] & 0xFFFFFFFFu);
        const int a = (int)floor(idx * scale) - 1;
        const int r = idx - (a + 1) * (numrho + 2) - 1;
        lines_out[2 * i] = ((float)r - (float)(numrho - 1) * 0.5f) * rho;
        lines_out[2 * i + 1] = 0.f + (float)a * theta;
    }
    if (tid == 0) {
        // RollCorrection.cpp:106-119: angle filter and sum, sequential in double
        double sum = 0.0;
        int count = 0;
        for (int i = 0; i < n; i++) {
            const int idx = 0x7FFFFFFF - (int)(unsigned)(sk[i] & 0xFFFFFFFFu);
            const int a = (int)floor(idx * scale) - 1;
            const float th = 0.f + (float)a * theta;
            const double deg = ((double)th * 180.0 / 3.1415926535897932384626433832795) - 90.0;
            if (deg >= amin && deg <= amax) { sum += deg; ++count; }
        }
        res->sum_deg = sum; res->count = count; res->n_lines = n; res->overflow = counters[3];
    }
}

struct HoughGeom {
    int numangle, numrho;
};

HoughGeom hough_geom(int w, int h, float rho, float theta) {
    HoughGeom g;
    const double PI = 3.1415926535897932384626433832795;
    int numangle = (int)std::floor((PI - 0.0) / theta) + 1;
    if (numangle > 1 && std::fabs(PI - (numangle - 1) * (double)theta) < (double)theta / 2) --numangle;
    const int max_rho = w + h, min_rho = -max_rho;
    g.numangle = numangle;
    g.numrho = (int)lrint(((max_rho - min_rho) + 1) / rho);
    return g;
}

}  // namespace

// Device scratch of one roll-correction / Canny+Hough evaluation on a w x h gray image.
struct RollWork {
    int w = 0, h = 0, mw = 0;
    float rho = 0, theta = 0;
    HoughGeom geom{};
    uint8_t* base = nullptr;
    uint8_t* gray = nullptr;
    uint8_t* edges = nullptr;
    short2* dxy = nullptr;
    int* mag = nullptr;
    unsigned long long* E = nullptr;     // Canny bit planes, wpr words per row
    unsigned long long* C = nullptr;
    int wpr = 0;
    int* queue = nullptr;
    int* list = nullptr;
    int* accum = nullptr;
    float* tabSin = nullptr;
    float* tabCos = nullptr;
    unsigned long long* keys = nullptr;
    float* lines = nullptr;
    int* counters = nullptr;
    int* hflags = nullptr;               // hysteresis pass flags (16 words after the counters)
    RollResult* res = nullptr;
    size_t accum_bytes = 0;
    int frames = 1;                      // work areas in this allocation, laid out alike, fb bytes apart (the pointers: frame 0)
    size_t fb = 0;
};

// Frame f of a work area for several frames, as a work area of its own (it does not own the memory).
static RollWork frame_view(const RollWork& k, int f) {
    RollWork v = k;
    const size_t o = (size_t)f * k.fb;
    auto adv = [&](auto*& p) { p = reinterpret_cast<std::remove_reference_t<decltype(p)>>(reinterpret_cast<uint8_t*>(p) + o); };
    adv(v.gray); adv(v.edges); adv(v.dxy); adv(v.mag); adv(v.E); adv(v.C); adv(v.queue); adv(v.list); adv(v.accum);
    adv(v.tabSin); adv(v.tabCos); adv(v.keys); adv(v.lines); adv(v.counters); adv(v.hflags); adv(v.res);
    v.base = nullptr; v.frames = 1; v.fb = 0;
    return v;
}

static void roll_work_free(RollWork& k) {
    if (k.base) (void)hipFree(k.base);
    k = RollWork();
}

static int roll_work_alloc(RollWork& k, int w, int h, float rho, float theta, hipStream_t st, int frames = 1) {
    if (k.base && k.w == w && k.h == h && k.rho == rho && k.theta == theta && k.frames == frames) return VS_OK;
    roll_work_free(k);
    if (!(rho > 0) || !(theta > 0)) { set_last_error("hough: rho and theta must be positive"); return VS_ERR_INVALID_ARG; }
    // the accumulator has (pi / theta) x (2 (w + h) / rho) cells: bounded before the int casts of hough_geom can overflow
    if (!(3.1415926535897932 / (double)theta <= 4096.0) || !((2.0 * (w + h) + 1.0) / (double)rho <= 65536.0)) {
        set_last_error("hough: theta / rho too fine (more than 4096 angles or 65536 distance bins)");
        return VS_ERR_INVALID_ARG;
    }
    k.w = w; k.h = h; k.mw = w + 2; k.wpr = (w + 63) / 64; k.rho = rho; k.theta = theta;
    k.geom = hough_geom(w, h, rho, theta);
    const size_t npx = (size_t)w * h, nfr = (size_t)(w + 2) * (h + 2);
    k.accum_bytes = (size_t)(k.geom.numangle + 2) * (k.geom.numrho + 2) * 4;
    size_t off = 0;
    auto take = [&](size_t b) { size_t o = off; off += (b + 255) & ~(size_t)255; return o; };
    const size_t o_gray = take(npx), o_edges = take(npx), o_dxy = take(npx * 4), o_mag = take(nfr * 4);
    const size_t o_E = take((size_t)k.wpr * h * 8), o_C = take((size_t)k.wpr * h * 8);
    const size_t o_queue = take(npx * 4 + 64), o_list = take(npx * 4 + 64), o_accum = take(k.accum_bytes);
    const size_t o_sin = take((size_t)k.geom.numangle * 4), o_cos = take((size_t)k.geom.numangle * 4);
    const size_t o_keys = take((size_t)HOUGH_CAP * 8), o_lines = take((size_t)HOUGH_CAP * 8), o_cnt = take(128), o_res = take(64);
    VS_HIP_TRY(hipMalloc((void**)&k.base, off * frames));
    VS_HIP_TRY(hipMemsetAsync(k.base, 0, off * frames, st));
    k.frames = frames; k.fb = frames > 1 ? off : 0;
    uint8_t* b = k.base;
    k.gray = b + o_gray; k.edges = b + o_edges; k.dxy = (short2*)(b + o_dxy); k.mag = (int*)(b + o_mag);
    k.E = (unsigned long long*)(b + o_E); k.C = (unsigned long long*)(b + o_C); k.queue = (int*)(b + o_queue); k.list = (int*)(b + o_list); k.accum = (int*)(b + o_accum);
    k.tabSin = (float*)(b + o_sin); k.tabCos = (float*)(b + o_cos); k.keys = (unsigned long long*)(b + o_keys);
    k.lines = (float*)(b + o_lines); k.counters = (int*)(b + o_cnt); k.hflags = k.counters + 16; k.res = (RollResult*)(b + o_res);
    // the frame of mag stays 0; its interior and the bit planes are rewritten every frame
    // createTrigTable: float angle accumulation, sin/cos in double (host libm, as the oracle)
    std::vector<float> ts(k.geom.numangle), tc(k.geom.numangle);
    const float irho = 1 / rho;
    float ang = 0.f;
    for (int n = 0; n < k.geom.numangle; ang += theta, n++) {
        ts[n] = (float)(std::sin((double)ang) * irho);
        tc[n] = (float)(std::cos((double)ang) * irho);
    }
    for (int f = 0; f < frames; f++) {       // (every frame's area carries its own copy of the tables: one layout)
        VS_HIP_TRY(hipMemcpyAsync((uint8_t*)k.tabSin + (size_t)f * k.fb, ts.data(), ts.size() * 4, hipMemcpyHostToDevice, st));
        VS_HIP_TRY(hipMemcpyAsync((uint8_t*)k.tabCos + (size_t)f * k.fb, tc.data(), tc.size() * 4, hipMemcpyHostToDevice, st));
    }
    VS_HIP_TRY(hipStreamSynchronize(st));
    return VS_OK;
}

// One group of hysteresis passes: one flag word per pass (k.hflags), the first pass of a group runs always, a later
// one returns at once when the pass before it changed nothing.
static int hyst_group(RollWork& k, int group, hipStream_t st) {
    const int nwv = k.wpr < HB_WORDS ? k.wpr : HB_WORDS;
    dim3 hg((k.wpr + nwv - 1) / nwv, (k.h + HB_ROWS - 1) / HB_ROWS, k.frames);
    if (k.frames > 1) VS_HIP_TRY(hipMemset2DAsync(k.hflags, k.fb, 0, 64, k.frames, st));
    else VS_HIP_TRY(hipMemsetAsync(k.hflags, 0, 64, st));
    for (int p = 0; p < group; p++)
        hipLaunchKernelGGL(canny_hyst_band_kernel, hg, dim3(64 * nwv), 0, st, k.E, k.C, k.wpr, k.h, k.hflags + p,
                           p ? k.hflags + p - 1 : (int*)nullptr, k.fb);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

// Passes in groups of 8, 16, 16, ... after a first group of 4 whose last pass still changed something; the flags are
// read back per group.
static int hyst_finish(RollWork& k, hipStream_t st) {
    int32_t flags[16];
    for (int group = 8;; group = 16) {
        VS_TRY(hyst_group(k, group, st));
        VS_HIP_TRY(hipMemcpyAsync(flags, k.hflags, 64, hipMemcpyDeviceToHost, st));
        VS_HIP_TRY(hipStreamSynchronize(st));
        if (!flags[group - 1]) return VS_OK;
    }
}

// cv::Canny(gray, edges, low, high, 3, false) on device buffers.  With `unchecked` the edge map is written after
// the first group of four hysteresis passes without asking whether the growth had ended: the caller reads
// k.hflags[3] together with its own results and, if it is set, calls hyst_finish and canny_emit and redoes what it
// built on the edge map (the roll stage does; four passes are enough for nearly every frame, and this saves a round
// trip to the host per frame).
static int canny_emit(RollWork& k, uint8_t* d_edges, size_t estride, hipStream_t st) {
    dim3 grid((k.w + NT - 1) / NT, k.h);
    hipLaunchKernelGGL(canny_out_kernel, grid, dim3(NT), 0, st, k.E, k.wpr, k.w, k.h, d_edges, estride);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

static int run_canny(RollWork& k, const uint8_t* d_gray, size_t stride, double low_t, double high_t, uint8_t* d_edges,
                     size_t estride, hipStream_t st, bool unchecked = false, int first_group = 4) {
    if (low_t > high_t) std::swap(low_t, high_t);
    const int low = (int)std::floor(low_t), high = (int)std::floor(high_t);
    const int w = k.w, h = k.h;
    // (several frames: d_gray is frame 0's analysis image inside the work area, the others lie k.fb apart like everything else)
    dim3 grid((w + NT - 1) / NT, h, k.frames);
    hipLaunchKernelGGL(sobel_kernel, grid, dim3(NT), 0, st, d_gray, stride, w, h, k.dxy, k.mag, k.mw, k.fb);
    hipLaunchKernelGGL(canny_nms_kernel, grid, dim3(NT), 0, st, k.dxy, k.mag, w, h, k.mw, low, high, k.E, k.C, k.wpr, k.fb);
    VS_HIP_TRY(hipGetLastError());
    VS_TRY(hyst_group(k, first_group, st));
    if (!unchecked) {
        int32_t flag = 0;
        VS_HIP_TRY(hipMemcpyAsync(&flag, k.hflags + first_group - 1, 4, hipMemcpyDeviceToHost, st));
        VS_HIP_TRY(hipStreamSynchronize(st));
        if (flag) VS_TRY(hyst_finish(k, st));
    }
    return d_edges ? canny_emit(k, d_edges, estride, st) : VS_OK;
}

// The Canny stages of a batch of frames for the roll stage's asynchronous path, the edge set left in k.E: counters and hysteresis
// words of every frame cleared by ONE memset in front (run_hough is then told so), `passes` passes of the growth;
// k.hflags[passes - 1] != 0 afterwards: still growing.
// (All passes in one launch, the bands of a frame meeting at a barrier in global memory between passes - 72 workgroups at
// 3840 x 2160 / 4, resident together - was built and measured in round 4: eleven launches less per batch, but every pass then pays
// the barrier's round trip through memory and the growth runs to its end instead of twelve passes: 29.9 k frames/s for the roll
// stage alone against 36.3 k, the chain 13.0 k against 15.1 - 15.6 k on the same box, gpurun_out/r04_y.  Not kept.)
static int run_canny_batch(RollWork& k, const uint8_t* d_gray, size_t stride, double low_t, double high_t, hipStream_t st, int passes) {
    if (low_t > high_t) std::swap(low_t, high_t);
    const int low = (int)std::floor(low_t), high = (int)std::floor(high_t);
    const int w = k.w, h = k.h;
    VS_HIP_TRY(hipMemset2DAsync(k.counters, k.fb, 0, 128, k.frames, st));        // counters[16], hflags[16]
    dim3 grid((w + NT - 1) / NT, h, k.frames);
    hipLaunchKernelGGL(sobel_kernel, grid, dim3(NT), 0, st, d_gray, stride, w, h, k.dxy, k.mag, k.mw, k.fb);
    hipLaunchKernelGGL(canny_nms_kernel, grid, dim3(NT), 0, st, k.dxy, k.mag, w, h, k.mw, low, high, k.E, k.C, k.wpr, k.fb);
    const int nwv = k.wpr < HB_WORDS ? k.wpr : HB_WORDS;
    dim3 hg((k.wpr + nwv - 1) / nwv, (k.h + HB_ROWS - 1) / HB_ROWS, k.frames);
    for (int p = 0; p < passes; p++)
        hipLaunchKernelGGL(canny_hyst_band_kernel, hg, dim3(64 * nwv), 0, st, k.E, k.C, k.wpr, k.h, k.hflags + p,
                           p ? k.hflags + p - 1 : (int*)nullptr, k.fb);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

// cv::HoughLines(edges, lines, rho, theta, threshold) + the angle statistics of the roll stage
static int run_hough(RollWork& k, const uint8_t* d_edges, size_t estride, int threshold, double amin, double amax,
                     hipStream_t st, bool counters_cleared = false) {
    const int w = k.w, h = k.h;
    if (w > 65535 || h > 32767) { set_last_error("hough: image too large"); return VS_ERR_INVALID_ARG; }
    if (counters_cleared) {}
    else if (k.frames > 1) VS_HIP_TRY(hipMemset2DAsync(k.counters + 1, k.fb, 0, 60, k.frames, st));
    else VS_HIP_TRY(hipMemsetAsync(k.counters + 1, 0, 60, st));
    if (d_edges) {
        dim3 lgrid((w + NT * EL_PX - 1) / (NT * EL_PX), h);
        hipLaunchKernelGGL(edge_list_kernel, lgrid, dim3(NT), 0, st, d_edges, estride, w, h, k.list, k.counters);
    } else {        // the edge set of the last run_canny on this work area, as it stands in k.E
        hipLaunchKernelGGL(edge_list_bits_kernel, dim3((k.wpr * h + NT - 1) / NT, 1, k.frames), dim3(NT), 0, st, k.E, k.wpr, h, k.list,
                           k.counters, k.fb);
    }
    const size_t row_bytes = (size_t)k.geom.numrho * 4;
    if (row_bytes <= 60 * 1024) {
        hipLaunchKernelGGL(hough_accum_lds_kernel, dim3(k.geom.numangle, 1, k.frames), dim3(1024), row_bytes, st, k.list, k.counters,
                           k.tabSin, k.tabCos, k.geom.numrho, k.accum, k.fb);
    } else {
        if (k.frames > 1) VS_HIP_TRY(hipMemset2DAsync(k.accum, k.fb, 0, k.accum_bytes, k.frames, st));
        else VS_HIP_TRY(hipMemsetAsync(k.accum, 0, k.accum_bytes, st));
        hipLaunchKernelGGL(hough_accum_kernel, dim3(2048, 1, k.frames), dim3(NT), 0, st, k.list, k.counters, k.tabSin, k.tabCos,
                           k.geom.numangle, k.geom.numrho, k.accum, k.fb);
    }
    dim3 g2((k.geom.numrho + NT - 1) / NT, k.geom.numangle, k.frames);
    hipLaunchKernelGGL(hough_peaks_kernel, g2, dim3(NT), 0, st, k.accum, k.geom.numangle, k.geom.numrho, threshold, k.keys, k.counters, k.fb);
    hipLaunchKernelGGL(hough_select_kernel, dim3(1, 1, k.frames), dim3(1024), 0, st, k.keys, k.counters, k.geom.numrho, k.rho, k.theta,
                       amin, amax, k.lines, HOUGH_CAP, k.res, k.fb);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

}  // namespace vsd

using namespace vsd;

struct vs_roll {
    vs_roll_params_c p;
    int device = 0;
    hipStream_t st = nullptr;
    std::string err;
    RollWork wk;
    uint8_t* d_in = nullptr;
    uint8_t* d_out = nullptr;
    size_t io_bytes = 0;
    bool first = true;
    double smoothed = 0.0;       // sSmoothedAngle (RollCorrection.cpp:14)
    double last_detected = 0.0;
    int last_lines = 0, last_used = 0;
    // asynchronous NV12 path (vs_roll_correct_nv12_dev): the line search of a frame does not depend on the frames before it -
    // only the smoothed angle does, a three-flop recurrence on the host.  A frame's search is fifteen small launches: queued one
    // by one from the caller's thread they cost ~100 us of the runtime's launch path per frame (9 k frames/s; the same as a
    // captured graph, which the runtime replays node by node), and from eight threads, a frame each, still 80 us (12.5 k).  So the
    // searches of RB consecutive frames go through ONE launch per stage (blockIdx.z = frame, a work area per frame), queued by
    // one of nwk worker threads on its own stream; when the batch's results (24 bytes per frame) have arrived the worker - in
    // frame order - advances the angle (EMA, clamp, decay) and queues each frame's rotation (both planes) on `st`.  The caller's
    // thread only hands the frames over.
    static constexpr int NWK = 8, RB = 8, QMAX = 128;       // NWK: the most workers
    int nwk = 5;                         // worker threads in use (VS_ROLL_WORKERS, 1 .. NWK): alone three are as fast as eight (36 - 38 k
                                         // frames/s); beside a stabilizer on the same GPU a batch's launches wait behind its workgroups, and five
                                         // batches in flight keep the stage at 3.4 ms per 128 surfaces where three need 5.5 (gpurun_out/r04_ai)
    struct Job { const uint8_t* src; uint8_t* dst; int w, h; size_t pitch, uv, opitch, ouv; long seq; };
    struct Slot {
        RollWork wk;                     // RB frames
        hipStream_t st = nullptr;
        hipEvent_t ev = nullptr;
        int32_t* h_res = nullptr;        // page-locked, per frame 320 bytes: counters[16], hysteresis flags[16], pad, RollResult at byte 256
        ImgPair* h_pairs = nullptr;      // page-locked: (surface, analysis image) of the batch's frames
        ImgPair* d_pairs = nullptr;
        std::deque<std::vector<Job>> batches;
    } slot[NWK];
    std::vector<Job> pending;            // handed over, not yet a batch
    long nbatches = 0;
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    bool quit = false;
    int worker_rc = VS_OK;
    std::string worker_err;
    long nv_in = 0, nv_done = 0;          // frames handed over / closed (angle advanced, rotation queued)
    long slow_frames = 0;                 // frames whose edge set was still growing after the first group of passes
};

namespace {
thread_local RollWork g_op_work;   // scratch of the stand-alone vs_op_canny / vs_op_hough_lines
}

extern "C" {

void vs_roll_params_default(vs_roll_params_c* p) {   // RollCorrection.h:16-38
    if (!p) return;
    memset(p, 0, sizeof *p);
    p->struct_size = (int32_t)sizeof *p;
    p->scale_factor = 0.25;
    p->canny_threshold_low = 50.0;
    p->canny_threshold_high = 150.0;
    p->canny_aperture = 3;
    p->hough_rho = 1.0f;
    p->hough_theta = (float)(3.1415926535897932384626433832795 / 180.0f);
    p->hough_threshold = 100;
    p->angle_filter_min = -10.0;
    p->angle_filter_max = 10.0;
    p->angle_smoothing_alpha = 0.1;
    p->angle_decay = 0.995;
    p->max_angle_change_deg = 0.5;
}

int vs_op_canny(const void* d_gray, size_t stride, int w, int h, double low, double high, void* d_edges,
                size_t edges_stride, void* stream) {
    VS_TRY(ensure_device());
    if (!d_gray || !d_edges || w <= 0 || h <= 0 || h > 65535) { set_last_error("canny: invalid argument"); return VS_ERR_INVALID_ARG; }
    hipStream_t st = (hipStream_t)stream;
    VS_TRY(roll_work_alloc(g_op_work, w, h, 1.f, (float)(3.1415926535897932384626433832795 / 180.0f), st));
    return run_canny(g_op_work, (const uint8_t*)d_gray, stride, low, high, (uint8_t*)d_edges, edges_stride, st);
}

int vs_op_hough_lines(const void* d_edges, size_t stride, int w, int h, float rho, float theta, int threshold,
                      float* d_lines, int max_lines, int32_t* d_count, void* stream) {
    VS_TRY(ensure_device());
    if (!d_edges || !d_lines || !d_count || w <= 0 || h <= 0 || h > 65535 || max_lines <= 0) {
        set_last_error("hough_lines: invalid argument");
        return VS_ERR_INVALID_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    VS_TRY(roll_work_alloc(g_op_work, w, h, rho, theta, st));
    VS_TRY(run_hough(g_op_work, (const uint8_t*)d_edges, stride, threshold, -1e30, 1e30, st));
    VS_HIP_TRY(hipStreamSynchronize(st));
    RollResult r;
    VS_HIP_TRY(hipMemcpy(&r, g_op_work.res, sizeof r, hipMemcpyDeviceToHost));
    const int n = r.n_lines < max_lines ? r.n_lines : max_lines;
    if (n > 0) VS_HIP_TRY(hipMemcpy(d_lines, g_op_work.lines, (size_t)n * 8, hipMemcpyDeviceToDevice));
    VS_HIP_TRY(hipMemcpy(d_count, &n, 4, hipMemcpyHostToDevice));
    return VS_OK;
}

// cv::warpAffine: the forward matrix (double) is inverted in double
static void invert_forward(const double* M, double* Mi) {
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    const double A11 = M[4] * D, A22 = M[0] * D;
    Mi[0] = A11; Mi[1] = M[1] * -D; Mi[3] = M[3] * -D; Mi[4] = A22;
    Mi[2] = -Mi[0] * M[2] - Mi[1] * M[5];
    Mi[5] = -Mi[3] * M[2] - Mi[4] * M[5];
}

int vs_op_warp_affine_ex(const void* d_src, size_t src_stride, int sw, int sh, void* d_dst, size_t dst_stride, int dw,
                         int dh, int cn, const double* M, int border, void* stream) {
    VS_TRY(ensure_device());
    if (!M) return VS_ERR_INVALID_ARG;
    double Mi[6];
    invert_forward(M, Mi);
    return launch_warp_affine_inv((const uint8_t*)d_src, src_stride, sw, sh, (uint8_t*)d_dst, dst_stride, dw, dh, cn, Mi,
                                  border, (hipStream_t)stream);
}

int vs_roll_create(const vs_roll_params_c* params, int device, vs_roll** out) {
    if (!out) return VS_ERR_INVALID_ARG;
    *out = nullptr;
    if (!params || params->struct_size != (int32_t)sizeof(vs_roll_params_c)) { set_last_error("roll params: struct_size mismatch"); return VS_ERR_INVALID_ARG; }
    if (params->canny_aperture != 3) { set_last_error("roll: only cannyAperture 3 is supported"); return VS_ERR_UNSUPPORTED; }
    VS_TRY(ensure_device());
    VS_HIP_TRY(hipSetDevice(device));
    vs_roll* r = new (std::nothrow) vs_roll();
    if (!r) return VS_ERR_HIP;
    r->p = *params;
    r->device = device;
    hipError_t e = hipStreamCreateWithFlags(&r->st, hipStreamNonBlocking);
    if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); delete r; return VS_ERR_HIP; }
    *out = r;
    return VS_OK;
}

void vs_roll_destroy(vs_roll* r) {
    if (!r) return;
    (void)hipSetDevice(r->device);
    if (!r->workers.empty()) {
        { std::lock_guard<std::mutex> g(r->mu); r->quit = true; }
        r->cv_job.notify_all();
        for (auto& t : r->workers) t.join();
    }
    for (auto& q : r->slot) {
        if (q.st) { (void)hipStreamSynchronize(q.st); (void)hipStreamDestroy(q.st); }
        if (q.ev) (void)hipEventDestroy(q.ev);
        if (q.h_res) (void)hipHostFree(q.h_res);
        if (q.h_pairs) (void)hipHostFree(q.h_pairs);
        if (q.d_pairs) (void)hipFree(q.d_pairs);
        roll_work_free(q.wk);
    }
    if (r->st) (void)hipStreamSynchronize(r->st);
    roll_work_free(r->wk);
    if (r->d_in) (void)hipFree(r->d_in);
    if (r->d_out) (void)hipFree(r->d_out);
    if (r->st) (void)hipStreamDestroy(r->st);
    delete r;
}

const char* vs_roll_last_error(const vs_roll* r) { return r ? r->err.c_str() : ""; }

// The reference passes Parameters on every call while the smoothed angle persists (RollCorrection.cpp:13-19).
int vs_roll_set_params(vs_roll* r, const vs_roll_params_c* params) {
    if (!r || !params || params->struct_size != (int32_t)sizeof(vs_roll_params_c)) return VS_ERR_INVALID_ARG;
    if (params->canny_aperture != 3) { r->err = "roll: only cannyAperture 3 is supported"; set_last_error(r->err); return VS_ERR_UNSUPPORTED; }
    if (r->nv_in != r->nv_done) { const int rc = vs_roll_sync(r); if (rc != VS_OK) return rc; }      // (frames in flight keep the parameters they were given)
    r->p = *params;
    return VS_OK;
}

int vs_roll_get_state(const vs_roll* r, double* smoothed_deg, double* detected_deg, int* n_lines, int* n_used) {
    if (!r) return VS_ERR_INVALID_ARG;
    if (smoothed_deg) *smoothed_deg = r->smoothed;
    if (detected_deg) *detected_deg = r->last_detected;
    if (n_lines) *n_lines = r->last_lines;
    if (n_used) *n_used = r->last_used;
    return VS_OK;
}

#define R_HIP(r, expr)                                                             \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) { (r)->err = std::string(#expr) + ": " + hipGetErrorString(_e); set_last_error((r)->err); return VS_ERR_HIP; } \
    } while (0)
#define R_TRY(r, expr)                                                             \
    do { int _s = (expr); if (_s != VS_OK) { (r)->err = get_last_error(); return _s; } } while (0)

// The angle recurrence of RollCorrection.cpp:76-77, :106-135 on a frame's line statistics.
static void roll_update(vs_roll* r, const RollResult& res) {
    const vs_roll_params_c& p = r->p;
    r->last_lines = res.n_lines; r->last_used = res.count; r->last_detected = 0.0;
    if (res.n_lines == 0 || res.count == 0) {
        r->smoothed *= p.angle_decay;                                                             // :76-77,:122-123
    } else {
        const double detected = res.sum_deg / res.count;                                          // :125-135
        r->last_detected = detected;
        double na = p.angle_smoothing_alpha * detected + (1.0 - p.angle_smoothing_alpha) * r->smoothed;
        double diff = na - r->smoothed;
        if (std::fabs(diff) > p.max_angle_change_deg && p.max_angle_change_deg > 0.0) {
            diff = (diff > 0) ? p.max_angle_change_deg : -p.max_angle_change_deg;
            na = r->smoothed + diff;
        }
        r->smoothed = na;
    }
}

// autoCorrectRoll on device buffers (BGR8 in, BGR8 out, same size)
int vs_roll_correct_dev(vs_roll* r, const void* d_data, int w, int h, size_t stride, void* d_out, size_t out_stride) {
    if (!r || !d_data || !d_out || w <= 0 || h <= 0 || stride < (size_t)w * 3 || out_stride < (size_t)w * 3) return VS_ERR_INVALID_ARG;
    R_HIP(r, hipSetDevice(r->device));
    if (r->nv_in != r->nv_done) R_TRY(r, vs_roll_sync(r));         // (asynchronous NV12 frames first: one angle, one order)
    const vs_roll_params_c& p = r->p;
    if (r->first) { r->first = false; r->smoothed = 0.0; }                                       // :24-27
    int sw = (int)(w * p.scale_factor), sh = (int)(h * p.scale_factor);                         // :35-38
    if (!(sw > 0 && sh > 0)) { sw = w; sh = h; }                                                 // :40-45
    R_TRY(r, roll_work_alloc(r->wk, sw, sh, p.hough_rho, p.hough_theta, r->st));
    RollWork& k = r->wk;
    // resize + BGR2GRAY (:41,:51), Canny (:54-61), HoughLines (:66-73), angle statistics (:106-119)
    R_TRY(r, launch_resize_gray((const uint8_t*)d_data, stride, w, h, VS_FMT_BGR8, k.gray, sw, sw, sh, r->st));
    R_TRY(r, run_canny(k, k.gray, sw, p.canny_threshold_low, p.canny_threshold_high, nullptr, 0, r->st, /*unchecked=*/true));
    R_TRY(r, run_hough(k, nullptr, 0, p.hough_threshold, p.angle_filter_min, p.angle_filter_max, r->st));
    RollResult res;
    int32_t growing = 0;
    R_HIP(r, hipMemcpyAsync(&res, k.res, sizeof res, hipMemcpyDeviceToHost, r->st));
    R_HIP(r, hipMemcpyAsync(&growing, k.hflags + 3, 4, hipMemcpyDeviceToHost, r->st));
    R_HIP(r, hipStreamSynchronize(r->st));
    if (growing) {          // the edge set was still growing after four passes: finish it and redo the line search
        R_TRY(r, hyst_finish(k, r->st));
        R_TRY(r, run_hough(k, nullptr, 0, p.hough_threshold, p.angle_filter_min, p.angle_filter_max, r->st));
        R_HIP(r, hipMemcpyAsync(&res, k.res, sizeof res, hipMemcpyDeviceToHost, r->st));
        R_HIP(r, hipStreamSynchronize(r->st));
    }
    roll_update(r, res);
    // cv::getRotationMatrix2D(center, angle, 1.0) (:141-144)
    const float cx = w / 2.0f, cy = h / 2.0f;
    const double a = r->smoothed * 3.1415926535897932384626433832795 / 180;
    const double alpha = std::cos(a), beta = std::sin(a);
    const double M[6] = {alpha, beta, (1 - alpha) * cx - beta * cy, -beta, alpha, beta * cx + (1 - alpha) * cy};
    return vs_op_warp_affine_ex(d_data, stride, w, h, d_out, out_stride, w, h, 3, M, VS_BORDER_REPLICATE, r->st);   // :146-149
}

// One batch of the asynchronous NV12 path: the line searches of its frames as one launch per stage, then - in frame order - the
// angle recurrence and the rotations.  Returns with r->mu held by `lk` when the ordered part was reached.
static int roll_worker_batch(vs_roll* r, vs_roll::Slot& q, const std::vector<vs_roll::Job>& jobs, std::unique_lock<std::mutex>& lk) {
    const vs_roll_params_c& p = r->p;
    const int n = (int)jobs.size();
    const vs_roll::Job& j0 = jobs[0];
    if (!q.st) {
        R_HIP(r, hipStreamCreateWithFlags(&q.st, hipStreamNonBlocking));
        R_HIP(r, hipEventCreateWithFlags(&q.ev, hipEventDisableTiming | hipEventBlockingSync));   // (a worker waiting for its batch spends no core on it)
        R_HIP(r, hipHostMalloc((void**)&q.h_res, 320 * vs_roll::RB, hipHostMallocDefault));
        R_HIP(r, hipHostMalloc((void**)&q.h_pairs, sizeof(ImgPair) * vs_roll::RB, hipHostMallocDefault));
        R_HIP(r, hipMalloc((void**)&q.d_pairs, sizeof(ImgPair) * vs_roll::RB));
    }
    int sw = (int)(j0.w * p.scale_factor), sh = (int)(j0.h * p.scale_factor);                   // :35-38
    if (!(sw > 0 && sh > 0)) { sw = j0.w; sh = j0.h; }                                           // :40-45
    VS_TRY(roll_work_alloc(q.wk, sw, sh, p.hough_rho, p.hough_theta, q.st, vs_roll::RB));
    RollWork k = q.wk;               // the first n work areas
    k.base = nullptr; k.frames = n;
    for (int f = 0; f < n; f++) { q.h_pairs[f].src = jobs[f].src; q.h_pairs[f].dst = k.gray + (size_t)f * k.fb; }
    VS_HIP_TRY(hipMemcpyAsync(q.d_pairs, q.h_pairs, sizeof(ImgPair) * n, hipMemcpyHostToDevice, q.st));
    VS_TRY(launch_resize_gray_batch(q.d_pairs, n, j0.pitch, j0.w, j0.h, VS_FMT_GRAY8, sw, sw, sh, 0, q.st));                     // :41
    // (twelve hysteresis passes per batch - a pass whose predecessor changed nothing for its frame returns at once: with four, 40 % of
    // the bench clip's frames had to finish their growth one by one behind the batch, 30 us per frame)
    constexpr int PASSES = 12;
    VS_TRY(run_canny_batch(k, k.gray, sw, p.canny_threshold_low, p.canny_threshold_high, q.st, PASSES));                            // :54-61
    VS_TRY(run_hough(k, nullptr, 0, p.hough_threshold, p.angle_filter_min, p.angle_filter_max, q.st, /*counters_cleared=*/true));  // :66-73, :106-119
    // per frame: counters (64 B), hysteresis flags (64 B) and, 256 bytes on, the line statistics
    VS_HIP_TRY(hipMemcpy2DAsync(q.h_res, 320, k.counters, k.fb, 256 + sizeof(RollResult), n, hipMemcpyDeviceToHost, q.st));
    VS_HIP_TRY(hipEventRecord(q.ev, q.st));
    VS_HIP_TRY(hipEventSynchronize(q.ev));
    std::vector<RollResult> res((size_t)n);
    std::vector<char> slow((size_t)n, 0);
    for (int f = 0; f < n; f++) {
        const uint8_t* h = reinterpret_cast<const uint8_t*>(q.h_res) + (size_t)320 * f;
        memcpy(&res[f], h + 256, sizeof(RollResult));
        if (reinterpret_cast<const int32_t*>(h)[16 + PASSES - 1]) {      // the edge set was still growing after the last pass: finish it, search again
            slow[f] = 1;
            RollWork v = frame_view(q.wk, f);
            VS_TRY(hyst_finish(v, q.st));
            VS_TRY(run_hough(v, nullptr, 0, p.hough_threshold, p.angle_filter_min, p.angle_filter_max, q.st));
            VS_HIP_TRY(hipMemcpyAsync(&res[f], v.res, sizeof(RollResult), hipMemcpyDeviceToHost, q.st));
            VS_HIP_TRY(hipStreamSynchronize(q.st));
        }
    }
    // ---- in frame order: the angle recurrence and the rotations
    lk.lock();
    r->cv_done.wait(lk, [&] { return r->nv_done == j0.seq; });
    double Iy[vs_roll::RB * 6], Iu[vs_roll::RB * 6];
    const uint8_t *ys[vs_roll::RB], *us[vs_roll::RB];
    uint8_t *yd[vs_roll::RB], *ud[vs_roll::RB];
    bool one_launch = true;
    for (int f = 0; f < n; f++) {
        const vs_roll::Job& j = jobs[f];
        if (slow[f]) r->slow_frames++;
        roll_update(r, res[f]);
        // cv::getRotationMatrix2D(center, angle, 1.0) (:141-144); the interleaved chroma plane is the half-size picture: the same
        // rotation with the translation halved
        const float cx = j.w / 2.0f, cy = j.h / 2.0f;
        const double a = r->smoothed * 3.1415926535897932384626433832795 / 180;
        const double alpha = std::cos(a), beta = std::sin(a);
        const double M[6] = {alpha, beta, (1 - alpha) * cx - beta * cy, -beta, alpha, beta * cx + (1 - alpha) * cy};
        const double Mc[6] = {M[0], M[1], M[2] * 0.5, M[3], M[4], M[5] * 0.5};
        invert_forward(M, Iy + 6 * f);
        invert_forward(Mc, Iu + 6 * f);
        ys[f] = j.src; us[f] = j.src + j.uv; yd[f] = j.dst; ud[f] = j.dst + j.ouv;
        one_launch &= j.opitch == j0.opitch && j.uv == j0.uv && j.ouv == j0.ouv;
    }
    // the rotations of the batch (:146-149): luma and chroma tiles of all its surfaces in ONE grid (warp_nv12_kernel with the
    // BORDER_REPLICATE staging) when the results share one layout (the surfaces do), else frame by frame
    if (one_launch) {
        VS_TRY(launch_warp_nv12_list_inv(ys, yd, n, j0.pitch, j0.opitch, j0.w, j0.h, j0.uv, j0.ouv, Iy, Iu, VS_BORDER_REPLICATE, r->st));
    } else {
        for (int f = 0; f < n; f++) {
            const vs_roll::Job& j = jobs[f];
            VS_TRY(launch_warp_affine_inv(ys[f], j.pitch, j.w, j.h, yd[f], j.opitch, j.w, j.h, 1, Iy + 6 * f, VS_BORDER_REPLICATE, r->st));
            VS_TRY(launch_warp_affine_inv(us[f], j.pitch, j.w / 2, j.h / 2, ud[f], j.opitch, j.w / 2, j.h / 2, 2, Iu + 6 * f, VS_BORDER_REPLICATE, r->st));
        }
    }
    return VS_OK;
}

static void roll_worker(vs_roll* r, int wi) {
    (void)hipSetDevice(r->device);
    vs_roll::Slot& q = r->slot[wi];
    for (;;) {
        std::vector<vs_roll::Job> jobs;
        {
            std::unique_lock<std::mutex> lk(r->mu);
            r->cv_job.wait(lk, [&] { return r->quit || !q.batches.empty(); });
            if (q.batches.empty()) return;
            jobs.swap(q.batches.front());
            q.batches.pop_front();
        }
        std::unique_lock<std::mutex> lk(r->mu, std::defer_lock);
        const int rc = roll_worker_batch(r, q, jobs, lk);
        if (!lk.owns_lock()) {           // (a batch that failed before its ordered part still has to let the next one pass)
            lk.lock();
            r->cv_done.wait(lk, [&] { return r->nv_done == jobs[0].seq; });
        }
        if (rc != VS_OK && r->worker_rc == VS_OK) { r->worker_rc = rc; r->worker_err = get_last_error(); }
        r->nv_done += (long)jobs.size();
        lk.unlock();
        r->cv_done.notify_all();
    }
}

// (r->mu held) what has been handed over becomes a batch of the next worker
static void roll_flush_pending(vs_roll* r) {
    if (r->pending.empty()) return;
    r->slot[r->nbatches % r->nwk].batches.emplace_back();
    r->slot[r->nbatches % r->nwk].batches.back().swap(r->pending);
    r->nbatches++;
}

// autoCorrectRoll for an NV12 surface in HBM (luma plane at d_surface, interleaved chroma plane uv_offset bytes behind it; the
// same for the result), ASYNCHRONOUS: the line search runs on the luma plane (resize x scale_factor -> Canny -> HoughLines; a
// gray picture needs no cvtColor), the rotation is applied to both planes.  The call hands the frame over and returns; eight
// consecutive frames of one geometry form a batch that a worker thread analyses with one launch per stage.  Results are complete
// after vs_roll_sync (which also closes an incomplete batch).  The surface and the result buffer of a call must stay untouched
// until then.
int vs_roll_correct_nv12_dev(vs_roll* r, const void* d_surface, int w, int h, size_t pitch, size_t uv_offset, void* d_out, size_t out_pitch,
                             size_t out_uv_offset) {
    if (!r || !d_surface || !d_out || w < 2 || h < 2 || (w & 1) || (h & 1) || pitch < (size_t)w || out_pitch < (size_t)w) return VS_ERR_INVALID_ARG;
    if (uv_offset == 0) uv_offset = (size_t)h * pitch;
    if (out_uv_offset == 0) out_uv_offset = (size_t)h * out_pitch;
    R_HIP(r, hipSetDevice(r->device));
    if (r->workers.empty()) {
        try {
            if (const char* e = std::getenv("VS_ROLL_WORKERS")) r->nwk = std::max(1, std::min(std::atoi(e), (int)vs_roll::NWK));
            for (int i = 0; i < r->nwk; i++) r->workers.emplace_back(roll_worker, r, i);
        } catch (...) {
            r->err = "roll: cannot start worker threads"; set_last_error(r->err);
            return VS_ERR_HIP;
        }
    }
    {
        std::unique_lock<std::mutex> lk(r->mu);
        if (r->first) { r->first = false; r->smoothed = 0.0; }                                   // :24-27
        r->cv_done.wait(lk, [&] { return r->nv_in - r->nv_done < vs_roll::QMAX; });
        if (!r->pending.empty() && (r->pending[0].w != w || r->pending[0].h != h || r->pending[0].pitch != pitch)) roll_flush_pending(r);
        r->pending.push_back(vs_roll::Job{(const uint8_t*)d_surface, (uint8_t*)d_out, w, h, pitch, uv_offset, out_pitch, out_uv_offset, r->nv_in});
        r->nv_in++;
        if ((int)r->pending.size() >= vs_roll::RB) roll_flush_pending(r);
    }
    r->cv_job.notify_all();
    return VS_OK;
}

// n surfaces of one layout in call order (what n calls of vs_roll_correct_nv12_dev do, without n trips through a binding)
int vs_roll_correct_nv12_dev_n(vs_roll* r, const void* const* d_surfaces, void* const* d_outs, int n, int w, int h, size_t pitch, size_t uv_offset,
                               size_t out_pitch, size_t out_uv_offset) {
    if (!r || !d_surfaces || !d_outs || n < 0) return VS_ERR_INVALID_ARG;
    for (int i = 0; i < n; i++) {
        const int rc = vs_roll_correct_nv12_dev(r, d_surfaces[i], w, h, pitch, uv_offset, d_outs[i], out_pitch, out_uv_offset);
        if (rc != VS_OK) return rc;
    }
    return VS_OK;
}

int vs_roll_sync(vs_roll* r) {
    if (!r) return VS_ERR_INVALID_ARG;
    R_HIP(r, hipSetDevice(r->device));
    if (!r->workers.empty()) {          // asynchronous NV12 frames: until the last one is closed
        std::unique_lock<std::mutex> lk(r->mu);
        roll_flush_pending(r);
        r->cv_job.notify_all();
        r->cv_done.wait(lk, [&] { return r->nv_done == r->nv_in; });
        if (r->worker_rc != VS_OK) {
            const int rc = r->worker_rc;
            r->err = r->worker_err; set_last_error(r->err);
            r->worker_rc = VS_OK;
            return rc;
        }
    }
    R_HIP(r, hipStreamSynchronize(r->st));
    return VS_OK;
}

int vs_roll_correct(vs_roll* r, const uint8_t* data, int w, int h, size_t stride, uint8_t* out, size_t out_stride) {
    if (!r || !data || !out || w <= 0 || h <= 0) return VS_ERR_INVALID_ARG;
    R_HIP(r, hipSetDevice(r->device));
    const size_t row = (size_t)w * 3, bytes = row * h;
    if (r->io_bytes < bytes) {
        if (r->d_in) (void)hipFree(r->d_in);
        if (r->d_out) (void)hipFree(r->d_out);
        r->d_in = r->d_out = nullptr;
        R_HIP(r, hipMalloc((void**)&r->d_in, bytes));
        R_HIP(r, hipMalloc((void**)&r->d_out, bytes));
        r->io_bytes = bytes;
    }
    R_HIP(r, hipMemcpy2DAsync(r->d_in, row, data, stride, row, h, hipMemcpyHostToDevice, r->st));
    int rc = vs_roll_correct_dev(r, r->d_in, w, h, row, r->d_out, row);
    if (rc != VS_OK) return rc;
    R_HIP(r, hipMemcpy2DAsync(out, out_stride, r->d_out, row, row, h, hipMemcpyDeviceToHost, r->st));
    R_HIP(r, hipStreamSynchronize(r->st));
    return VS_OK;
}

}  // extern "C"
