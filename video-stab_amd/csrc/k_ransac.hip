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

#include "vs_common.h"

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

struct RansacArgs {
    const float* from;
    const float* to;
    int n;                    // host count (capacity); device count overrides when d_n != nullptr
    const int32_t* d_n;
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
};

__device__ __forceinline__ int device_count(const RansacArgs& a) {
    int n = a.n;
    if (a.d_n) { const int dn = *a.d_n; n = dn < n ? dn : n; }
    return n;
}

__global__ __launch_bounds__(64) void ransac_score_kernel(RansacArgs a) {
    const int k = blockIdx.x, lane = threadIdx.x;
    const int n = device_count(a);
    if (n < a.min_points || n <= 2 || n > a.table_max_m) return;
    const uint32_t pr = a.pairs[(size_t)n * a.iters + k];
    const int i0 = pr & 0xFFFFu, i1 = pr >> 16;
    const Model M = kernel2(a.from[2 * i0], a.from[2 * i0 + 1], a.from[2 * i1], a.from[2 * i1 + 1],
                            a.to[2 * i0], a.to[2 * i0 + 1], a.to[2 * i1], a.to[2 * i1 + 1]);
    float F[6];
#pragma unroll
    for (int i = 0; i < 6; i++) F[i] = (float)M.m[i];
    int good = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        bool in = false;
        if (i < n) in = is_inlier(F, a.from[2 * i], a.from[2 * i + 1], a.to[2 * i], a.to[2 * i + 1], a.t);
        good += __popcll(__ballot(in));
    }
    if (lane == 0) a.counts[k] = good;
}

__global__ __launch_bounds__(64) void ransac_select_kernel(RansacArgs a) {
    __shared__ int s_counts[4096];
    const int lane = threadIdx.x;
    const int n = device_count(a);
    // failure defaults
    if (lane < 6) a.model[lane] = __longlong_as_double(0x7FF8000000000000LL);
    for (int i = lane; i < a.n; i += 64) a.inliers[i] = 0;
    if (lane < 4) a.info[lane] = lane == 1 ? -1 : 0;
    if (n < a.min_points || n < 2 || n > a.table_max_m) return;

    Model best;
    int best_iter = -1, niters = a.iters > 1 ? a.iters : 1, max_good = 0;
    if (n == 2) {
        // count == modelPoints: the model of the two points, all inliers, no refinement
        best = kernel2(a.from[0], a.from[1], a.from[2], a.from[3], a.to[0], a.to[1], a.to[2], a.to[3]);
        if (lane < 6) a.model[lane] = best.m[lane];
        if (lane < 2) a.inliers[lane] = 1;
        if (lane == 0) { a.info[0] = 1; a.info[1] = 0; a.info[2] = 0; a.info[3] = 2; }
        return;
    }
    for (int i = lane; i < a.iters; i += 64) s_counts[i] = a.counts[i];
    __syncthreads();
    // RANSACPointSetRegistrator::run bookkeeping, replayed over the vote counts
    for (int iter = 0; iter < niters; iter++) {
        const int good = s_counts[iter];
        if (good > (max_good > 1 ? max_good : 1)) {
            max_good = good;
            best_iter = iter;
            const uint32_t u = a.update[(size_t)n * (n + 1) / 2 + good];
            const int mstar = u & 0xFFFFu, kround = u >> 16;
            if (mstar == 0xFFFF) niters = 0;              // denom < DBL_MIN -> 0
            else if (niters > mstar) niters = kround;     // else: unchanged (returns maxIters)
        }
    }
    if (lane == 0) { a.info[1] = best_iter; a.info[2] = niters; }
    if (max_good <= 0 || best_iter < 0) return;
    const uint32_t pr = a.pairs[(size_t)n * a.iters + best_iter];
    const int i0 = pr & 0xFFFFu, i1 = pr >> 16;
    best = kernel2(a.from[2 * i0], a.from[2 * i0 + 1], a.from[2 * i1], a.from[2 * i1 + 1],
                   a.to[2 * i0], a.to[2 * i0 + 1], a.to[2 * i1], a.to[2 * i1 + 1]);
    float F[6];
#pragma unroll
    for (int i = 0; i < 6; i++) F[i] = (float)best.m[i];
    for (int i = lane; i < n; i += 64)
        a.inliers[i] = is_inlier(F, a.from[2 * i], a.from[2 * i + 1], a.to[2 * i], a.to[2 * i + 1], a.t) ? 1 : 0;
    __syncthreads();
    // refinement on the inliers: least squares of the linear residual, summed in index order
    if (lane == 0) {
        double Sx = 0, Sy = 0, SX = 0, SY = 0, Sxx = 0, SxX = 0, SxY = 0;
        int m = 0;
        for (int i = 0; i < n; i++) {
            if (!a.inliers[i]) continue;
            const double x = a.from[2 * i], y = a.from[2 * i + 1], X = a.to[2 * i], Y = a.to[2 * i + 1];
            Sx += x; Sy += y; SX += X; SY += Y;
            Sxx += x * x + y * y;
            SxX += x * X + y * Y;
            SxY += x * Y - y * X;
            m++;
        }
        if (m > 0) {
            const double N = (double)m;
            const double den = N * Sxx - Sx * Sx - Sy * Sy;
            if (fabs(den) > 0) {
                const double ra = (N * SxX - Sx * SX - Sy * SY) / den;
                const double rb = (N * SxY - Sx * SY + Sy * SX) / den;
                const double tx = (SX - ra * Sx + rb * Sy) / N;
                const double ty = (SY - rb * Sx - ra * Sy) / N;
                best.m[0] = ra; best.m[1] = -rb; best.m[2] = tx; best.m[3] = rb; best.m[4] = ra; best.m[5] = ty;
            }
        }
        for (int i = 0; i < 6; i++) a.model[i] = best.m[i];
        a.info[0] = 1;
        a.info[3] = max_good;
    }
}

}  // namespace

// counts: device scratch of `iters` int32.  d_n (optional) = device count.
int launch_ransac(const float* d_from, const float* d_to, int n, const int32_t* d_n, int min_points,
                  double thr, int iters, const RansacTables* tab, int32_t* d_counts, double* d_model,
                  uint8_t* d_inliers, int32_t* d_info, hipStream_t st) {
    if (!d_from || !d_to || n < 0 || !tab || !d_counts || !d_model || !d_inliers || !d_info ||
        iters != tab->iters || n > tab->max_m) {
        set_last_error("ransac: invalid argument");
        return VS_ERR_INVALID_ARG;
    }
    RansacArgs a;
    a.from = d_from; a.to = d_to; a.n = n; a.d_n = d_n; a.min_points = min_points;
    a.t = (float)(thr * thr);
    a.iters = iters; a.pairs = tab->d_pairs; a.update = tab->d_update; a.table_max_m = tab->max_m;
    a.counts = d_counts; a.model = d_model; a.inliers = d_inliers; a.info = d_info;
    if (n > 2) hipLaunchKernelGGL(ransac_score_kernel, dim3(iters), dim3(64), 0, st, a);
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
    int rc = launch_ransac(d_from, d_to, n, nullptr, 2, thr, max_iters, tab, counts, d_model, d_inliers, d_info, st);
    hipError_t e = hipStreamSynchronize(st);
    (void)hipFree(counts);
    if (rc == VS_OK && e != hipSuccess) { set_last_error(hipGetErrorString(e)); rc = VS_ERR_HIP; }
    return rc;
}

}  // namespace vsd
