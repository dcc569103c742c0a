// RANSAC partial-affine (similarity) estimator for gfx950: device counterpart of
//   cv::estimateAffinePartial2D(from, to, noArray(), cv::RANSAC, 5.0, 500)
// as called at /root/reference/src/Stabilizer.cpp:647-649.
//
// OpenCV's loop is sequential (adaptive iteration count), but everything a
// hypothesis needs is known up front: cv::RNG is seeded with the constant
// (uint64)-1, so the index pair of hypothesis k depends only on k and on the
// number of correspondences M.  The host generates that pair table once
// (RansacTables), the score kernel evaluates ALL hypotheses in parallel (one
// wavefront per hypothesis, inlier votes by ballot+popcount), and a
// single-wave kernel replays OpenCV's sequential bookkeeping (strict
// improvement, RANSACUpdateNumIters early stop) over the vote counts, so the
// kept hypothesis, the inlier mask and the iteration count are exactly those
// of the serial algorithm.  The refinement solves the 4x4 normal equations of
// the linear residual in closed form, double precision,
// inliers summed in index order.
//
// MFMA note: the "2xN normal-equation GEMM" here is 7 running sums over <= a
// few hundred points - it is not reshaped into a matrix-core GEMM (HBM/latency
// bound path).
#include <map>
#include <mutex>
#include <vector>

#include <cstring>

#include "traj_device.h"
#include "traj_emit_device.h"
#include "vs_common.h"
#include "warp_tab.h"

namespace vsd {

struct RansacTables {
    int max_m = 0, iters = 0;
    uint32_t* d_pairs = nullptr;   // [max_m+1][iters]  (i0 | i1<<16)
    uint32_t* d_update = nullptr;  // triangular [(m*(m+1))/2 + good] : mstar | kround<<16
};

namespace {

// ---- host: tables -------------------------------------------------------------
struct CvRng {
    uint64_t state;
    explicit CvRng(uint64_t s) : state(s ? s : 0xffffffffULL) {}
    uint32_t next() {
        state = (uint64_t)(uint32_t)state * 4164903690U + (uint32_t)(state >> 32);
        return (uint32_t)state;
    }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (uint32_t)(b - a) + a); }
};

// RANSACUpdateNumIters(0.99, ep, 2, maxIters) is `maxIters` while
// maxIters <= mstar and `kround` above it; both derived with the exact
// double expressions of ptsetreg.cpp.
static void update_entry(int m, int good, int iters, uint16_t* mstar, uint16_t* kround) {
    double p = 0.99;
    double ep = (double)(m - good) / m;
    ep = ep < 0. ? 0. : (ep > 1. ? 1. : ep);
    double num = 1. - p;
    if (num < DBL_MIN) num = DBL_MIN;
    double denom = 1. - std::pow(1. - ep, 2);
    if (denom < DBL_MIN) { *mstar = 0xFFFF; *kround = 0; return; }   // returns 0 for every maxIters
    num = std::log(num);
    denom = std::log(denom);
    if (denom >= 0) { *mstar = (uint16_t)iters; *kround = (uint16_t)iters; return; }  // always maxIters
    // largest m in [0, iters] with -num >= m * (-denom)  (monotone in m)
    int lo = 0, hi = iters;   // condition true at 0 (-num > 0)
    while (lo < hi) {
        int mid = (lo + hi + 1) / 2;
        if (-num >= mid * (-denom)) lo = mid; else hi = mid - 1;
    }
    long r = lrint(num / denom);
    if (r < 0) r = 0;
    if (r > iters) r = iters;   // only reached when maxIters > mstar >= r-ish; clamp is a no-op in range
    *mstar = (uint16_t)lo;
    *kround = (uint16_t)r;
}

std::mutex g_tab_mutex;
std::map<std::pair<int, int>, RansacTables> g_tables;

}  // namespace

// Returns device tables valid for M <= max_m and `iters` hypotheses (cached per process/device).
int get_ransac_tables(int max_m, int iters, const RansacTables** out) {
    if (max_m < 2) max_m = 2;
    if (max_m > 4096 || iters < 1 || iters > 4096) {
        set_last_error("ransac: max points 4096, max iterations 4096");
        return VS_ERR_INVALID_ARG;
    }
    int dev = 0;
    VS_HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_tab_mutex);
    // round max_m up so that a few sizes serve every instance
    int cap = 256;
    while (cap < max_m) cap *= 2;
    auto key = std::make_pair(dev * 8192 + cap, iters);
    auto it = g_tables.find(key);
    if (it == g_tables.end()) {
        RansacTables t;
        t.max_m = cap; t.iters = iters;
        std::vector<uint32_t> pairs((size_t)(cap + 1) * iters, 0);
        for (int m = 3; m <= cap; m++) {
            CvRng rng((uint64_t)-1);
            for (int k = 0; k < iters; k++) {
                int i0 = rng.uniform(0, m);
                int i1;
                for (i1 = rng.uniform(0, m); i1 == i0; i1 = rng.uniform(0, m)) {}
                pairs[(size_t)m * iters + k] = (uint32_t)i0 | ((uint32_t)i1 << 16);
            }
        }
        std::vector<uint32_t> upd((size_t)(cap + 1) * (cap + 2) / 2 + 1, 0);
        for (int m = 1; m <= cap; m++)
            for (int g = 0; g <= m; g++) {
                uint16_t ms, kr;
                update_entry(m, g, iters, &ms, &kr);
                upd[(size_t)m * (m + 1) / 2 + g] = (uint32_t)ms | ((uint32_t)kr << 16);
            }
        VS_HIP_TRY(hipMalloc((void**)&t.d_pairs, pairs.size() * 4));
        VS_HIP_TRY(hipMalloc((void**)&t.d_update, upd.size() * 4));
        VS_HIP_TRY(hipMemcpy(t.d_pairs, pairs.data(), pairs.size() * 4, hipMemcpyHostToDevice));
        VS_HIP_TRY(hipMemcpy(t.d_update, upd.data(), upd.size() * 4, hipMemcpyHostToDevice));
        it = g_tables.emplace(key, t).first;
    }
    *out = &it->second;
    return VS_OK;
}

namespace {

struct Model { double m[6]; };

// AffinePartial2DEstimatorCallback::runKernel (2-point closed form)
__device__ __forceinline__ Model kernel2(float fx1, float fy1, float fx2, float fy2, float tx1, float ty1,
                                         float tx2, float ty2) {
    const double x1 = fx1, y1 = fy1, x2 = fx2, y2 = fy2;
    const double X1 = tx1, Y1 = ty1, X2 = tx2, Y2 = ty2;
    const double d = 1. / ((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2));
    const double S0 = d * ((X1 - X2) * (x1 - x2) + (Y1 - Y2) * (y1 - y2));
    const double S1 = d * ((Y1 - Y2) * (x1 - x2) - (X1 - X2) * (y1 - y2));
    const double S2 = d * ((Y1 - Y2) * (x1 * y2 - x2 * y1) - (X1 * y2 - X2 * y1) * (y1 - y2) -
                           (X1 * x2 - X2 * x1) * (x1 - x2));
    const double S3 = d * (-(X1 - X2) * (x1 * y2 - x2 * y1) - (Y1 * x2 - Y2 * x1) * (x1 - x2) -
                           (Y1 * y2 - Y2 * y1) * (y1 - y2));
    Model r;
    r.m[0] = S0; r.m[1] = -S1; r.m[2] = S2; r.m[3] = S1; r.m[4] = S0; r.m[5] = S3;
    return r;
}

// Affine2DEstimatorCallback::computeError + findInliers for one point
__device__ __forceinline__ bool is_inlier(const float F[6], float fx, float fy, float tx, float ty, float t) {
    const float a = F[0] * fx + F[1] * fy + F[2] - tx;
    const float b = F[3] * fx + F[4] * fy + F[5] - ty;
    return a * a + b * b <= t;
}

constexpr int MAX_PTS = 4096;

// What the ordered tail of a batch needs of a frame, in one place: written by the frame's selection, read by the tail with ONE
// round of loads (from the argument block it was three dependent ones - block, then the pointers in it, then what they point
// to - and a round trip costs the one-workgroup tail 10 us and more while the detector's kernels stream beside it).
struct TailIn {
    double model[6];
    int32_t info[4];
    int32_t nprev, have_prev_gray, pad[2];
};

struct RansacArgs {
    // correspondences.  With `status` != nullptr they are the raw LK output
    // (prev/next + status) and every workgroup compacts them itself
    // (Stabilizer.cpp:629-641); workgroup 0 also stores the compacted lists.
    const float* from;
    const float* to;
    const uint8_t* status;
    int n;                    // host count (capacity); device count overrides when d_n != nullptr
    const int32_t* d_n;
    float* vp;                // compacted lists (written when status != nullptr)
    float* vc;
    int32_t* d_m;             // compacted count
    int min_points;           // 2 for the plain operator, 4 when gated like Stabilizer.cpp:645
    float t;                  // (float)(thr*thr)
    int iters;
    const uint32_t* pairs;    // tables
    const uint32_t* update;
    int table_max_m;
    int32_t* counts;          // [iters] scratch
    double* model;            // out: 6 doubles (NaN when no model)
    uint8_t* inliers;         // out: n bytes
    int32_t* info;            // out: {ok, best_iter, iters_run, n_inliers}
    // fused trajectory append (pipeline only)
    TrajState* traj;
    TrajParams tp;
    vs_debug_frame* dbg;
    int have_prev_gray;
    int last_of_stream;       // group launches (several streams in one table): this is its stream's last frame of the batch
    TailIn* tail_in;          // batch mode: where the selection leaves its results for the tail (or nullptr)
};

__device__ __forceinline__ int device_count(const RansacArgs& a) {
    int n = a.n;
    if (a.d_n) { const int dn = *a.d_n; n = dn < n ? dn : n; }
    return n;
}

// Order-preserving compaction of the tracked pairs into LDS; returns M.
__device__ __forceinline__ int compact_to_lds(const RansacArgs& a, int n, float2* sf, float2* st, int lane) {
    // Four rounds of 64 pairs at a time: their status bytes and coordinates are loaded without conditions (from clamped
    // positions) before the first ballot - one round trip per 256 pairs where status load -> ballot -> coordinate loads per
    // round made two dependent ones per 64.
    int m = 0;
    for (int base = 0; base < n; base += 256) {
        uint8_t sv[4];
        float2 fv[4], tv[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int i = min(base + 64 * r + lane, n - 1);
            sv[r] = a.status ? a.status[i] : (uint8_t)1;
            fv[r] = *reinterpret_cast<const float2*>(a.from + 2 * i);
            tv[r] = *reinterpret_cast<const float2*>(a.to + 2 * i);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const bool keep = base + 64 * r + lane < n && sv[r] != 0;
            const unsigned long long mask = __ballot(keep);
            const int pos = m + __popcll(mask & ((1ull << lane) - 1ull));
            if (keep) { sf[pos] = fv[r]; st[pos] = tv[r]; }
            m += __popcll(mask);
        }
    }
    return m;
}

// Hypothesis scoring, SC_HYP hypotheses per wave: four lanes share a hypothesis and take every fourth point each, their counts
// meet in two DPP adds.  A wave compacts the correspondences once for 16 hypotheses, and a batch of 32 frames x 500 hypotheses
// is 1024 waves - one per SIMD: the launch is as long as one wave's chain (compaction, the closed-form model, m / 4 inlier
// tests).  (The votes are integers: who counts them does not change them.  Round 2 measured one wave per hypothesis, 16 000
// waves per batch, against it: 22 -> 13.7 us per batch; scratch/README.md.)
constexpr int SC_HYP = 16;

__device__ __forceinline__ void ransac_score16(const RansacArgs& a, const int kgroup) {
    extern __shared__ float2 s_pts[];          // [2][a.n]: compacted from / to
    float2* sf = s_pts;
    float2* st = s_pts + a.n;
    const int lane = threadIdx.x;
    const int n = device_count(a);
    const int m = compact_to_lds(a, n, sf, st, lane);
    __syncthreads();
    if (kgroup == 0 && a.status) {
        for (int i = lane; i < m; i += 64) {
            a.vp[2 * i] = sf[i].x; a.vp[2 * i + 1] = sf[i].y;
            a.vc[2 * i] = st[i].x; a.vc[2 * i + 1] = st[i].y;
        }
        if (lane == 0) *a.d_m = m;
    }
    if (m < a.min_points || m <= 2 || m > a.table_max_m) return;
    const int k = kgroup * SC_HYP + (lane >> 2), q = lane & 3;
    const bool valid = k < a.iters;
    const uint32_t pr = a.pairs[(size_t)m * a.iters + (valid ? k : 0)];
    const int i0 = pr & 0xFFFFu, i1 = pr >> 16;
    const Model M = kernel2(sf[i0].x, sf[i0].y, sf[i1].x, sf[i1].y, st[i0].x, st[i0].y, st[i1].x, st[i1].y);
    float F[6];
#pragma unroll
    for (int i = 0; i < 6; i++) F[i] = (float)M.m[i];
    int good = 0;
    for (int i = q; i < m; i += 4) good += is_inlier(F, sf[i].x, sf[i].y, st[i].x, st[i].y, a.t) ? 1 : 0;
    good += __builtin_amdgcn_update_dpp(0, good, 0xB1, 0xF, 0xF, true);      // quad_perm [1,0,3,2]
    good += __builtin_amdgcn_update_dpp(0, good, 0x4E, 0xF, 0xF, true);      // quad_perm [2,3,0,1]
    if (valid && q == 0) a.counts[k] = good;
}

__global__ __launch_bounds__(64) void ransac_score16_kernel(RansacArgs a) { ransac_score16(a, blockIdx.x); }

__device__ __forceinline__ double wave_butterfly_sum(double v) {
    // fixed combine order (mirrored by oracle/vso_ransac.cpp): s[l] += s[l ^ off], off = 32..1
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off, 64);
    return v;
}

// One wave: replays RANSACPointSetRegistrator::run over the vote counts, rebuilds
// the inlier mask of the kept hypothesis, refines, and (pipeline) appends the
// measured transform to the trajectory.
// MULTI: several waves of one workgroup each select for their own frame (batch tail); the trajectory
// append is then left to the caller, and only the wave flagged write_dbg reports its counts.
template <bool MULTI>
__device__ __forceinline__ void ransac_select(const RansacArgs& a, const bool write_dbg = true) {
    const int lane = threadIdx.x & 63;
    const float* from = a.status ? a.vp : a.from;
    const float* to = a.status ? a.vc : a.to;
    int n = a.status ? *a.d_m : device_count(a);
    const int nprev = a.status ? device_count(a) : n;
    // failure defaults
    if (lane < 6) a.model[lane] = __longlong_as_double(0x7FF8000000000000LL);
    for (int i = lane; i < a.n; i += 64) a.inliers[i] = 0;
    if (lane < 4) a.info[lane] = lane == 1 ? -1 : 0;
    int r_info[4] = {0, -1, 0, 0};                    // what a.info ends up holding (wave-uniform), for the tail's record
    if (a.dbg && write_dbg && lane == 0) { a.dbg->n_prev = nprev; a.dbg->n_valid = n; }
    if (MULTI) { __threadfence_block(); __builtin_amdgcn_wave_barrier(); }
    else __syncthreads();
    bool ok = false;
    Model best;
    if (n >= a.min_points && n >= 2 && n <= a.table_max_m) {
        if (n == 2) {
            // count == modelPoints: the model of the two points, all inliers, no refinement
            best = kernel2(from[0], from[1], from[2], from[3], to[0], to[1], to[2], to[3]);
            if (lane < 6) a.model[lane] = best.m[lane];
            if (lane < 2) a.inliers[lane] = 1;
            if (lane == 0) { a.info[0] = 1; a.info[1] = 0; a.info[2] = 0; a.info[3] = 2; }
            r_info[0] = 1; r_info[1] = 0; r_info[2] = 0; r_info[3] = 2;
            ok = true;
        } else {
            int best_iter = -1, niters = a.iters > 1 ? a.iters : 1, max_good = 0;
            // RANSACPointSetRegistrator::run bookkeeping: 64 iterations per step; inside a step
            // the strict improvements are visited in order with ballot / first-set-bit.
            for (int base = 0; base < niters; base += 64) {
                const int iter = base + lane;
                const int good = iter < a.iters ? a.counts[iter] : 0;
                while (true) {
                    const int floor_good = max_good > 1 ? max_good : 1;
                    const unsigned long long mask = __ballot(iter < niters && good > floor_good);
                    if (!mask) break;
                    const int j = __ffsll((long long)mask) - 1;
                    max_good = __builtin_amdgcn_readlane(good, j);
                    best_iter = base + j;
                    const uint32_t u = a.update[(size_t)n * (n + 1) / 2 + max_good];
                    const int mstar = u & 0xFFFFu, kround = u >> 16;
                    if (mstar == 0xFFFF) niters = 0;              // denom < DBL_MIN -> 0
                    else if (niters > mstar) niters = kround;     // else unchanged (returns maxIters)
                    // lanes <= j are settled: later candidates must beat the new maximum
                    if (lane <= j) { /* good of these lanes can no longer exceed max_good strictly before j */ }
                }
            }
            if (lane == 0) { a.info[1] = best_iter; a.info[2] = niters; }
            r_info[1] = best_iter; r_info[2] = niters;
            if (max_good > 0 && best_iter >= 0) {
                const uint32_t pr = a.pairs[(size_t)n * a.iters + best_iter];
                const int i0 = pr & 0xFFFFu, i1 = pr >> 16;
                best = kernel2(from[2 * i0], from[2 * i0 + 1], from[2 * i1], from[2 * i1 + 1],
                               to[2 * i0], to[2 * i0 + 1], to[2 * i1], to[2 * i1 + 1]);
                float F[6];
#pragma unroll
                for (int i = 0; i < 6; i++) F[i] = (float)best.m[i];
                // refinement on the inliers: least squares of the linear residual.  Summation
                // order: lane l accumulates points i = l, l+64, ... in increasing i, then the 64
                // partial sums are combined by the fixed butterfly above.
                double Sx = 0, Sy = 0, SX = 0, SY = 0, Sxx = 0, SxX = 0, SxY = 0;
                int cnt = 0;
                for (int i = lane; i < n; i += 64) {
                    const float fx = from[2 * i], fy = from[2 * i + 1], tx = to[2 * i], ty = to[2 * i + 1];
                    const bool in = is_inlier(F, fx, fy, tx, ty, a.t);
                    a.inliers[i] = in ? 1 : 0;
                    if (in) {
                        const double x = fx, y = fy, X = tx, Y = ty;
                        Sx += x; Sy += y; SX += X; SY += Y;
                        Sxx += x * x + y * y;
                        SxX += x * X + y * Y;
                        SxY += x * Y - y * X;
                        cnt++;
                    }
                }
                Sx = wave_butterfly_sum(Sx); Sy = wave_butterfly_sum(Sy);
                SX = wave_butterfly_sum(SX); SY = wave_butterfly_sum(SY);
                Sxx = wave_butterfly_sum(Sxx); SxX = wave_butterfly_sum(SxX); SxY = wave_butterfly_sum(SxY);
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
                if (cnt > 0) {
                    const double N = (double)cnt;
                    const double den = N * Sxx - Sx * Sx - Sy * Sy;
                    if (fabs(den) > 0) {
                        const double ra = (N * SxX - Sx * SX - Sy * SY) / den;
                        const double rb = (N * SxY - Sx * SY + Sy * SX) / den;
                        const double tx = (SX - ra * Sx + rb * Sy) / N;
                        const double ty = (SY - rb * Sx - ra * Sy) / N;
                        best.m[0] = ra; best.m[1] = -rb; best.m[2] = tx; best.m[3] = rb; best.m[4] = ra; best.m[5] = ty;
                    }
                }
                if (lane < 6) a.model[lane] = best.m[lane];
                if (lane == 0) { a.info[0] = 1; a.info[3] = max_good; }
                r_info[0] = 1; r_info[3] = max_good;
                ok = true;
            }
        }
    }
    if (MULTI && a.tail_in) {
        TailIn* ti = a.tail_in;
        if (lane < 6) ti->model[lane] = ok ? best.m[lane] : __longlong_as_double(0x7FF8000000000000LL);
        if (lane == 6) { ti->info[0] = r_info[0]; ti->info[1] = r_info[1]; ti->info[2] = r_info[2]; ti->info[3] = r_info[3]; }
        if (lane == 7) { ti->nprev = nprev; ti->have_prev_gray = a.have_prev_gray; }
    }
    (void)ok;
    if (!MULTI && a.traj) {
        // the results above were written by this same wave; make them visible to lane 0's reads
        __threadfence_block();
        __syncthreads();
        if (lane == 0) traj_append_device(a.traj, a.tp, a.model, a.info, nprev, a.dbg, a.have_prev_gray);
    }
}

__global__ __launch_bounds__(64) void ransac_select_kernel(RansacArgs a) { ransac_select<false>(a); }

// Several frames per launch: blockIdx.y selects the frame's argument block in a device table, blockIdx.x the hypothesis.
__global__ __launch_bounds__(64) void ransac_score16_batch_kernel(const RansacArgs* __restrict__ table) {
    ransac_score16(table[blockIdx.y], blockIdx.x);
}

// The selections of a batch, one wave (workgroup) per frame, launched right behind the scoring: they do not depend on
// each other, so they do not belong into the one-workgroup tail (where they took two rounds of sixteen waves on one CU).
// (Tried first: the scoring workgroup that finishes a frame last selects for it - a counter per frame and a device-scope
// fence per workgroup.  16 000 L2 write-backs per launch: the scoring launch went from 20 to 320 us and dragged the
// kernels beside it along.)
__global__ __launch_bounds__(64) void ransac_select_batch_kernel(const RansacArgs* __restrict__ table, int last_item) {
    ransac_select<true>(table[blockIdx.x], (int)blockIdx.x == last_item || table[blockIdx.x].last_of_stream != 0);
}

// Ordered tail of a batch, ONE launch: for every frame in push order, hypothesis selection + trajectory append,
// then - when that push produces an output - the map of the frame leaving the queue, which must see exactly
// the transforms appended so far (Stabilizer.cpp:380-389).
struct TailItem {
    int out_due, out_idx;
    double* Minv_out;
    int ncnt, seg;               // ncnt: written by the tail - transforms appended after this push (read by the release kernel);
                                 // seg: index of the frame's stream segment
    WarpTabJob tab[2];           // coordinate tables of the due frame's warp (frame plane; chroma plane of an NV12 surface), built
                                 // by the release workgroup as soon as the inverse maps exist (tabs == nullptr: not wanted)
};
// A step's frames belong to one or more streams (a standalone instance is a group of one): the items of stream s are
// table[first .. first + n), workgroup s of the tail takes them, and the stream's frame matrix goes to its own M_out.
// apart: the tail workgroup ends with the appends (phases 1 and 2a) and leaves the releases - smoothing around the frame each
// due push lets go, its matrix and inverse map - to ransac_release_batch_kernel, one workgroup per push: on the one CU of
// a tail workgroup sixteen waves share four SIMDs and a batch of 64 releases took 60 of the tail's 130 us.  (Not for the Kalman
// smoother, whose filter state advances from release to release: apart = 0, the releases stay inside the tail.)
struct TailSeg { int first, n; float* M_out; TrajState* traj; vs_debug_frame* dbg; int apart, pad; };

__global__ __launch_bounds__(1024) void ransac_tail_batch_kernel(const RansacArgs* __restrict__ table, TailItem* __restrict__ tail,
                                                                 const TailSeg* __restrict__ segs, const TailIn* __restrict__ tin) {
    // (Launched as sixteen waves this kernel took 85 - 110 us beside the detector's and the pyramid's launches of the next batch,
    // alone 10 - 25, and s_setprio(3) for its waves changed nothing: the time went into waiting for a compute unit with room for
    // the whole workgroup.  As four waves: 32 us beside them - launch_ransac_tail_group.)
    const TailSeg sg = segs[blockIdx.x];
    table += sg.first; tail += sg.first; tin += sg.first;
    const int n = sg.n;
    float* const M_out = sg.M_out;
    TrajState* const g_state = sg.traj;
    vs_debug_frame* const g_dbg = sg.dbg;
    if (n <= 0) return;
    // The ordered part below is a chain of small dependent steps executed by one lane; run from global
    // memory every step would pay an HBM round trip (and every barrier would wait for the stores of the
    // step before).  The stream's trajectory state, the parameters, the per-frame inputs and all results
    // therefore live in LDS for the duration of the kernel and are written out once at the end.
    __shared__ TrajState l_state;
    __shared__ TrajParams l_tp;
    __shared__ vs_debug_frame l_dbg;
    constexpr int TB = 64;           // frames per tail (= BATCH_MAX of stabilizer.cpp)
    __shared__ double l_model[TB][6];
    __shared__ double l_minv[TB][12];
    __shared__ float l_M[12];
    __shared__ int32_t l_info[TB][4];
    __shared__ int l_nprev[TB], l_hpg[TB], l_due[TB], l_oidx[TB];
    __shared__ float l_t3[TB][4];
    __shared__ int l_ncnt[TB];                       // transforms appended after push i
    __shared__ float l_trf[TB][3];                   // measured (dx, dy, da) of frame i, decomposed by the frame's own wave
    __shared__ vs_debug_frame l_dbg_scratch[16];     // per wave: record of a release that is not the batch's last
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // (state, record, parameters and the frames' inputs all lie at addresses the kernel was given: one round of loads)
    for (int i = tid; i < (int)(sizeof(TrajState) / 4); i += blockDim.x)
        reinterpret_cast<uint32_t*>(&l_state)[i] = reinterpret_cast<const uint32_t*>(g_state)[i];
    for (int i = tid; i < (int)(sizeof(TrajParams) / 4); i += blockDim.x)
        reinterpret_cast<uint32_t*>(&l_tp)[i] = reinterpret_cast<const uint32_t*>(&table[0].tp)[i];
    if (tid < 12) l_M[tid] = M_out[tid];
    // phase 1: per-frame inputs of the ordered part: sixteen lanes serve a frame, so that the dependent global loads of all
    // 64 frames of a batch are one round (a wave per frame took four rounds on the 16 waves of this workgroup)
    const int nwaves = blockDim.x >> 6;
    // (the selections themselves ran behind the scoring launch, one workgroup per frame: ransac_select_batch_kernel)
    for (int f = tid >> 4; f < n; f += blockDim.x >> 4) {
        const TailIn& a = tin[f];
        const int l16 = tid & 15;
        if (l16 < 6) l_model[f][l16] = a.model[l16];
        if (l16 >= 8 && l16 < 12) l_info[f][l16 - 8] = a.info[l16 - 8];
        if (l16 == 15) {
            l_nprev[f] = a.nprev; l_hpg[f] = a.have_prev_gray;
            l_due[f] = tail[f].out_due; l_oidx[f] = tail[f].out_idx;
            // :644-662 for this frame (identity when the estimation failed or was skipped): the arctangent leaves the
            // ordered chain below
            float t3[3] = {0.f, 0.f, 0.f};
            if (a.nprev > 0 && a.have_prev_gray) {
                float T[6] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f};
                if (a.info[0]) for (int i = 0; i < 6; i++) T[i] = (float)a.model[i];
                t3[0] = T[2]; t3[1] = T[5]; t3[2] = vslibm::atan2f_ref(T[3], T[0]);
            }
            l_trf[f][0] = t3[0]; l_trf[f][1] = t3[1]; l_trf[f][2] = t3[2];
        }
    }
    // (the selections ran in the launch before this one: the last one has reported n_prev / n_valid into the global record)
    for (int i = tid; i < (int)(sizeof(vs_debug_frame) / 4); i += blockDim.x)
        reinterpret_cast<uint32_t*>(&l_dbg)[i] = reinterpret_cast<const uint32_t*>(g_dbg)[i];
    __syncthreads();
    // phase 2a, ordered, one lane: append the measured transforms of all frames (Stabilizer.cpp:660-693), remembering how
    // long the trajectory was after each push
    if (tid == 0) {
        const bool plain = !l_tp.drone && !l_tp.adaptive;   // no filter state between the measurement and the path
        int i = 0;
        if (plain) {
            // path accumulation alone (:681-688), the running path kept in registers; the batch's last frame takes the
            // full routine below, which also writes the debug record
            int m = l_state.n;
            float lp[3] = {l_state.last_path[0], l_state.last_path[1], l_state.last_path[2]};
            for (; i < n - 1; i++) {
                const int slot = m & (TRAJ_RING - 1);
                for (int c = 0; c < 3; c++) {
                    const float t = l_trf[i][c];
                    lp[c] = m == 0 ? t : lp[c] + t;
                    l_state.transforms[slot][c] = t;
                    l_state.path[slot][c] = lp[c];
                }
                m++;
                l_ncnt[i] = m;
            }
            l_state.n = m;
            for (int c = 0; c < 3; c++) l_state.last_path[c] = lp[c];
        }
        for (; i < n; i++) {
            traj_append_device(&l_state, l_tp, l_model[i], l_info[i], l_nprev[i], &l_dbg, l_hpg[i]);
            l_ncnt[i] = l_state.n;
        }
    }
    int last_due = -1;
    for (int i = 0; i < n; i++) if (l_due[i]) last_due = i;
    __syncthreads();
    if (sg.apart) {
        for (int i = tid; i < (int)(sizeof(TrajState) / 4); i += blockDim.x)
            reinterpret_cast<uint32_t*>(g_state)[i] = reinterpret_cast<const uint32_t*>(&l_state)[i];
        for (int i = tid; i < (int)(sizeof(vs_debug_frame) / 4); i += blockDim.x)
            reinterpret_cast<uint32_t*>(g_dbg)[i] = reinterpret_cast<const uint32_t*>(&l_dbg)[i];
        if (tid < n) tail[tid].ncnt = l_ncnt[tid];
        return;
    }
    // phase 2b: smooth around the frame that push i releases -> its transform (dx, dy, da).  The release of push i must
    // see exactly the transforms appended up to push i (:380-389): it is given that length, and the later entries of the
    // rings lie beyond everything it reads (ring of 256, at most 64 pushes - one batch - ahead).  Box and Gaussian smoothing keep no
    // state of their own, so the releases of a batch run on different waves; the Kalman recursion (:1416-1458) advances a
    // filter state from release to release and stays on wave 0, in order.  The debug record keeps the last release.
    if (l_tp.method == VS_SMOOTH_KALMAN) {
        if (wave == 0) {
            for (int i = 0; i < n; i++) {
                if (l_due[i]) traj_emit_lds_wave0(&l_state, l_tp, l_oidx[i], l_ncnt[i], &l_dbg, l_t3[i]);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    } else {
        for (int i = wave; i < n; i += nwaves)
            if (l_due[i]) traj_emit_lds_wave0(&l_state, l_tp, l_oidx[i], l_ncnt[i], i == last_due ? &l_dbg : &l_dbg_scratch[wave], l_t3[i]);
    }
    __syncthreads();
    // phase 3: the matrices and inverse maps of all due outputs, one lane each (cosf / sinf / the double
    // inversions leave the ordered chain); the last due output also leaves its matrix in the debug record
    {
        if (tid < n && l_due[tid]) {
            float Mt[12];
            traj_matrix_lane(l_t3[tid], Mt, l_minv[tid], nullptr);
            if (tid == last_due) {
                for (int i = 0; i < 12; i++) l_M[i] = Mt[i];
                for (int i = 0; i < 6; i++) l_dbg.warp_matrix[i] = Mt[i];
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < (int)(sizeof(TrajState) / 4); i += blockDim.x)
        reinterpret_cast<uint32_t*>(g_state)[i] = reinterpret_cast<const uint32_t*>(&l_state)[i];
    for (int i = tid; i < (int)(sizeof(vs_debug_frame) / 4); i += blockDim.x)
        reinterpret_cast<uint32_t*>(g_dbg)[i] = reinterpret_cast<const uint32_t*>(&l_dbg)[i];
    if (tid < 12) M_out[tid] = l_M[tid];
    for (int f = wave; f < n; f += nwaves)
        if (lane < 12 && l_due[f]) tail[f].Minv_out[lane] = l_minv[f][lane];
}

// The releases of a step whose appends ransac_tail_batch_kernel has made (segments with `apart`): workgroup f = push f.  Each sees
// the trajectory as long as it was after its push (TailItem::ncnt); the last due one of a stream leaves its record and matrix
// behind, like the last release of the one-kernel tail.
constexpr int RELEASE_NT = 256;
__global__ __launch_bounds__(RELEASE_NT) void ransac_release_batch_kernel(const RansacArgs* __restrict__ table, const TailItem* __restrict__ tail,
                                                                          const TailSeg* __restrict__ segs) {
    __shared__ vs_debug_frame l_dbg_unused;
    __shared__ float l_M[12];
    __shared__ double l_minv[12];
    int f = blockIdx.x;
    const TailItem t = tail[f];
    if (!t.out_due) return;
    const TailSeg sg = segs[t.seg];          // the frame's own stream: its items, its matrix
    if (!sg.apart) return;
    table += sg.first; tail += sg.first; f -= sg.first;
    int last_due = -1;
    for (int i = 0; i < sg.n; i++) if (tail[i].out_due) last_due = i;
    const bool last = f == last_due;
    // wave 0: smoothing around the released frame, its matrix and inverse maps (12 doubles: frame plane, chroma plane)
    traj_emit_device(table[0].traj, table[0].tp, t.out_idx, last ? sg.M_out : l_M, l_minv, last ? table[0].dbg : &l_dbg_unused, nullptr, t.ncnt);
    __syncthreads();
    if (threadIdx.x < 12) t.Minv_out[threadIdx.x] = l_minv[threadIdx.x];
    // all waves: the coordinate tables of the frame's warp from the maps lane 0 has left in LDS (cv::warpAffine's adelta /
    // bdelta / X0 / Y0, warp_tab.h).  As a launch of their own behind the tail they cost 4 - 5 us of kernel and an event gap of
    // 6.5 us per 32 frames on the critical stream (round 3: 4 points of the warp stage's HBM fraction).
    wt_build_plane(t.tab[0], l_minv, threadIdx.x, RELEASE_NT);
    wt_build_plane(t.tab[1], l_minv + 6, threadIdx.x, RELEASE_NT);
}

}  // namespace

static void fill_ransac_args(RansacArgs& a, const float* d_from, const float* d_to, const uint8_t* d_status, int n,
                             const int32_t* d_n, float* d_vp, float* d_vc, int32_t* d_m, int min_points, double thr,
                             int iters, const RansacTables* tab, int32_t* d_counts, double* d_model, uint8_t* d_inliers,
                             int32_t* d_info, TrajState* traj, const TrajParams* tp, vs_debug_frame* dbg,
                             int have_prev_gray) {
    memset(&a, 0, sizeof a);
    a.from = d_from; a.to = d_to; a.status = d_status; a.n = n; a.d_n = d_n;
    a.vp = d_vp; a.vc = d_vc; a.d_m = d_m; a.min_points = min_points;
    a.t = (float)(thr * thr);
    a.iters = iters; a.pairs = tab->d_pairs; a.update = tab->d_update; a.table_max_m = tab->max_m;
    a.counts = d_counts; a.model = d_model; a.inliers = d_inliers; a.info = d_info;
    a.traj = traj; a.dbg = dbg; a.have_prev_gray = have_prev_gray;
    if (tp) a.tp = *tp;
}

static bool bad_ransac_args(const float* d_from, const float* d_to, const uint8_t* d_status, int n, float* d_vp, float* d_vc,
                            int32_t* d_m, int iters, const RansacTables* tab, int32_t* d_counts, double* d_model,
                            uint8_t* d_inliers, int32_t* d_info, TrajState* traj, const TrajParams* tp, vs_debug_frame* dbg) {
    return !d_from || !d_to || n < 0 || n > MAX_PTS || !tab || !d_counts || !d_model || !d_inliers || !d_info ||
           iters != tab->iters || n > tab->max_m || (d_status && (!d_vp || !d_vc || !d_m)) || (traj && (!tp || !dbg));
}

// ---- batched form (see stabilizer.cpp): the hypotheses of several frames are scored by one launch from a
// device table of argument blocks; the selection (and the fused trajectory append, which is ordered) is then
// launched per frame from the host copy of its block.
size_t ransac_item_bytes() { return sizeof(RansacArgs); }

int ransac_fill_item(void* host_item, const float* d_from, const float* d_to, const uint8_t* d_status, int n,
                     const int32_t* d_n, float* d_vp, float* d_vc, int32_t* d_m, int min_points, double thr, int iters,
                     const RansacTables* tab, int32_t* d_counts, double* d_model, uint8_t* d_inliers, int32_t* d_info,
                     TrajState* traj, const TrajParams* tp, vs_debug_frame* dbg, int have_prev_gray) {
    if (!host_item || bad_ransac_args(d_from, d_to, d_status, n, d_vp, d_vc, d_m, iters, tab, d_counts, d_model, d_inliers,
                                      d_info, traj, tp, dbg)) {
        set_last_error("ransac: invalid argument");
        return VS_ERR_INVALID_ARG;
    }
    fill_ransac_args(*static_cast<RansacArgs*>(host_item), d_from, d_to, d_status, n, d_n, d_vp, d_vc, d_m, min_points, thr,
                     iters, tab, d_counts, d_model, d_inliers, d_info, traj, tp, dbg, have_prev_gray);
    return VS_OK;
}

int launch_ransac_score_batch(const void* d_table, int items, int iters, int n_max, hipStream_t st) {
    if (!d_table || items < 1 || items > 65535 || iters < 1 || n_max < 0 || n_max > MAX_PTS) {
        set_last_error("ransac_score_batch: invalid argument");
        return VS_ERR_INVALID_ARG;
    }
    hipLaunchKernelGGL(ransac_score16_batch_kernel, dim3((iters + SC_HYP - 1) / SC_HYP, items), dim3(64), (size_t)(n_max > 0 ? n_max : 1) * 16, st,
                       static_cast<const RansacArgs*>(d_table));
    hipLaunchKernelGGL(ransac_select_batch_kernel, dim3(items), dim3(64), 0, st, static_cast<const RansacArgs*>(d_table), items - 1);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

size_t tail_item_bytes() { return sizeof(TailItem); }

// tabs: nullptr, or the two table jobs of the due frame (frame plane, chroma plane)
void tail_fill_item(void* host_item, int out_due, int out_idx, double* d_Minv_out, const WarpTabJob* tabs) {
    TailItem& t = *static_cast<TailItem*>(host_item);
    t.out_due = out_due; t.out_idx = out_idx; t.Minv_out = d_Minv_out; t.ncnt = 0; t.seg = 0;
    for (int i = 0; i < 2; i++) {
        if (tabs) t.tab[i] = tabs[i];
        else { t.tab[i].tabs = nullptr; t.tab[i].src = nullptr; t.tab[i].dst = nullptr; t.tab[i].dw = 0; t.tab[i].dh = 0; }
    }
}

// ---- group launches (vs_batch: the frames of several streams in one table) ----
size_t tail_seg_bytes() { return sizeof(TailSeg); }
void tail_fill_seg(void* host_seg, int first, int n, float* d_M_out, TrajState* traj, vs_debug_frame* dbg, int smoothing_method) {
    TailSeg& g = *static_cast<TailSeg*>(host_seg);
    g.first = first; g.n = n; g.M_out = d_M_out; g.traj = traj; g.dbg = dbg;
    g.apart = smoothing_method != VS_SMOOTH_KALMAN ? 1 : 0; g.pad = 0;
}
size_t tail_in_bytes() { return sizeof(TailIn); }
void ransac_item_set_tail_in(void* host_item, void* d_tail_in) { static_cast<RansacArgs*>(host_item)->tail_in = static_cast<TailIn*>(d_tail_in); }
void tail_item_set_seg(void* host_item, int seg) { static_cast<TailItem*>(host_item)->seg = seg; }
void ransac_item_set_last(void* host_item, int last) { static_cast<RansacArgs*>(host_item)->last_of_stream = last; }

// The ordered tails of `nsegs` streams in one launch (workgroup = stream), then the releases of all their pushes
// (any_apart: some segment leaves its releases to the second launch).
int launch_ransac_tail_group(const void* d_table, const void* d_tail, const void* d_segs, const void* d_tail_in, int nsegs, int max_n, int items,
                             int any_apart, hipStream_t st) {
    if (!d_table || !d_tail || !d_segs || !d_tail_in || nsegs < 1 || max_n < 1 || max_n > 64 || items < 1) {
        set_last_error("ransac_tail_group: invalid argument");
        return VS_ERR_INVALID_ARG;
    }
    // Four waves, one per SIMD: the workgroup then fits a compute unit that the detector's and the pyramid's launches of the next batch
    // keep nearly full (with sixteen waves - a quarter of a unit's registers and 22 KB of LDS at once - it waited for one to drain:
    // 60 - 110 us on `main` beside them against 10 - 25 alone); its ordered part is one lane's work either way.
    const int threads = 64 * (max_n > 4 ? 4 : max_n);
    const RansacArgs* tb = static_cast<const RansacArgs*>(d_table);
    TailItem* tl = static_cast<TailItem*>(const_cast<void*>(d_tail));
    const TailSeg* sg = static_cast<const TailSeg*>(d_segs);
    const TailIn* ti = static_cast<const TailIn*>(d_tail_in);
    hipLaunchKernelGGL(ransac_tail_batch_kernel, dim3(nsegs), dim3(threads), 0, st, tb, tl, sg, ti);
    if (any_apart) hipLaunchKernelGGL(ransac_release_batch_kernel, dim3(items), dim3(RELEASE_NT), 0, st, tb, tl, sg);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

// counts: device scratch of `iters` int32.  d_n (optional) = device count.
// status != nullptr: (d_from,d_to,status) are the raw LK arrays; vp/vc/d_m receive
// the compacted pairs.  traj != nullptr: the measured transform is appended too.
int launch_ransac(const float* d_from, const float* d_to, const uint8_t* d_status, int n, const int32_t* d_n,
                  float* d_vp, float* d_vc, int32_t* d_m, int min_points, double thr, int iters,
                  const RansacTables* tab, int32_t* d_counts, double* d_model, uint8_t* d_inliers,
                  int32_t* d_info, TrajState* traj, const TrajParams* tp, vs_debug_frame* dbg,
                  int have_prev_gray, hipStream_t st) {
    if (!d_from || !d_to || n < 0 || n > MAX_PTS || !tab || !d_counts || !d_model || !d_inliers || !d_info ||
        iters != tab->iters || n > tab->max_m || (d_status && (!d_vp || !d_vc || !d_m)) || (traj && (!tp || !dbg))) {
        set_last_error("ransac: invalid argument");
        return VS_ERR_INVALID_ARG;
    }
    RansacArgs a;
    fill_ransac_args(a, d_from, d_to, d_status, n, d_n, d_vp, d_vc, d_m, min_points, thr, iters, tab, d_counts, d_model,
                     d_inliers, d_info, traj, tp, dbg, have_prev_gray);
    if (n > 2 || d_status)
        hipLaunchKernelGGL(ransac_score16_kernel, dim3((iters + SC_HYP - 1) / SC_HYP), dim3(64), (size_t)(n > 0 ? n : 1) * 16, st, a);
    hipLaunchKernelGGL(ransac_select_kernel, dim3(1), dim3(64), 0, st, a);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

int run_estimate_affine_partial2d(const float* d_from, const float* d_to, int n, double thr,
                                  int max_iters, double* d_model, uint8_t* d_inliers,
                                  int32_t* d_info, hipStream_t st) {
    if (n < 0 || max_iters < 1) { set_last_error("ransac: invalid argument"); return VS_ERR_INVALID_ARG; }
    const RansacTables* tab = nullptr;
    VS_TRY(get_ransac_tables(n, max_iters, &tab));
    int32_t* counts = nullptr;
    VS_HIP_TRY(hipMalloc((void**)&counts, (size_t)max_iters * 4));
    int rc = launch_ransac(d_from, d_to, nullptr, n, nullptr, nullptr, nullptr, nullptr, 2, thr, max_iters, tab,
                           counts, d_model, d_inliers, d_info, nullptr, nullptr, nullptr, 0, st);
    hipError_t e = hipStreamSynchronize(st);
    (void)hipFree(counts);
    if (rc == VS_OK && e != hipSuccess) { set_last_error(hipGetErrorString(e)); rc = VS_ERR_HIP; }
    return rc;
}

}  // namespace vsd
