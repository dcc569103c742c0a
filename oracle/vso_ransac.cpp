// CPU ORACLE (test infrastructure) - cv::estimateAffinePartial2D restated
// (reference call site src/Stabilizer.cpp:647-649: RANSAC, thr 5.0, 500 iters,
// default confidence 0.99, refineIters 10).  Follows OpenCV 4.11
// calib3d/src/ptsetreg.cpp: RANSACPointSetRegistrator::run with
// AffinePartial2DEstimatorCallback (2-point closed form), cv::RNG seeded with
// (uint64)-1, RANSACUpdateNumIters, then refinement on the inliers.
//
// Definition choice: the refinement's residual is linear in
// (a,b,tx,ty), so cv::LMSolver converges to the linear least-squares solution;
// the oracle solves the 4x4 normal equations in closed form (double; summation
// order documented at refine()).  Expected gap to an LM run: ~1e-12 relative.
#include "vso_internal.h"

#include <algorithm>

namespace vso {

struct RNG {
    uint64_t state;
    explicit RNG(uint64_t s) : state(s ? s : 0xffffffffULL) {}
    uint32_t next() {
        state = (uint64_t)(uint32_t)state * 4164903690U + (uint32_t)(state >> 32);
        return (uint32_t)state;
    }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (uint32_t)(b - a) + a); }
};

static int update_num_iters(double p, double ep, int model_points, int max_iters) {
    p = std::max(p, 0.); p = std::min(p, 1.);
    ep = std::max(ep, 0.); ep = std::min(ep, 1.);
    double num = std::max(1. - p, DBL_MIN);
    double denom = 1. - std::pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = std::log(num);
    denom = std::log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : cv_round(num / denom);
}

static void kernel2(const float* f0, const float* f1, const float* t0, const float* t1, double M[6]) {
    double x1 = f0[0], y1 = f0[1], x2 = f1[0], y2 = f1[1];
    double X1 = t0[0], Y1 = t0[1], X2 = t1[0], Y2 = t1[1];
    double d = 1. / ((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2));
    double S0 = d * ((X1 - X2) * (x1 - x2) + (Y1 - Y2) * (y1 - y2));
    double S1 = d * ((Y1 - Y2) * (x1 - x2) - (X1 - X2) * (y1 - y2));
    double S2 = d * ((Y1 - Y2) * (x1 * y2 - x2 * y1) - (X1 * y2 - X2 * y1) * (y1 - y2) -
                     (X1 * x2 - X2 * x1) * (x1 - x2));
    double S3 = d * (-(X1 - X2) * (x1 * y2 - x2 * y1) - (Y1 * x2 - Y2 * x1) * (x1 - x2) -
                     (Y1 * y2 - Y2 * y1) * (y1 - y2));
    M[0] = M[4] = S0; M[1] = -S1; M[2] = S2; M[3] = S1; M[5] = S3;
}

static int find_inliers(const float* from, const float* to, int n, const double* F, float t,
                        uint8_t* mask) {
    float F0 = (float)F[0], F1 = (float)F[1], F2 = (float)F[2];
    float F3 = (float)F[3], F4 = (float)F[4], F5 = (float)F[5];
    int nz = 0;
    for (int i = 0; i < n; i++) {
        float fx = from[2 * i], fy = from[2 * i + 1];
        float a = F0 * fx + F1 * fy + F2 - to[2 * i];
        float b = F3 * fx + F4 * fy + F5 - to[2 * i + 1];
        float e = a * a + b * b;
        int f = e <= t;
        mask[i] = (uint8_t)f;
        nz += f;
    }
    return nz;
}

// least squares for X = a x - b y + tx, Y = b x + a y + ty over the inliers.
// Summation order (any order is a valid restatement of the LS solution; this
// one is what a 64-lane wavefront does): partial sum l takes the inliers with
// index i = l, l+64, l+128, ... in increasing i; the 64 partials are then
// combined by a butterfly, s[l] += s[l ^ off] for off = 32,16,8,4,2,1.
static void refine(const float* from, const float* to, int n, const uint8_t* mask, double M[6]) {
    double P[7][64];
    for (auto& q : P) for (double& v : q) v = 0;
    int m = 0;
    for (int i = 0; i < n; i++) {
        if (!mask[i]) continue;
        const int l = i & 63;
        double x = from[2 * i], y = from[2 * i + 1], X = to[2 * i], Y = to[2 * i + 1];
        P[0][l] += x; P[1][l] += y; P[2][l] += X; P[3][l] += Y;
        P[4][l] += x * x + y * y;
        P[5][l] += x * X + y * Y;
        P[6][l] += x * Y - y * X;
        m++;
    }
    if (m == 0) return;
    double S[7];
    for (int q = 0; q < 7; q++) {
        double cur[64], nxt[64];
        for (int l = 0; l < 64; l++) cur[l] = P[q][l];
        for (int off = 32; off >= 1; off >>= 1) {
            for (int l = 0; l < 64; l++) nxt[l] = cur[l] + cur[l ^ off];
            for (int l = 0; l < 64; l++) cur[l] = nxt[l];
        }
        S[q] = cur[0];
    }
    const double Sx = S[0], Sy = S[1], SX = S[2], SY = S[3], Sxx = S[4], SxX = S[5], SxY = S[6];
    double N = (double)m;
    double den = N * Sxx - Sx * Sx - Sy * Sy;
    if (!(std::abs(den) > 0)) return;  // degenerate (all inliers coincide): keep the RANSAC model
    double a = (N * SxX - Sx * SX - Sy * SY) / den;
    double b = (N * SxY - Sx * SY + Sy * SX) / den;
    double tx = (SX - a * Sx + b * Sy) / N;
    double ty = (SY - b * Sx - a * Sy) / N;
    M[0] = M[4] = a; M[1] = -b; M[2] = tx; M[3] = b; M[5] = ty;
}

int estimate_affine_partial2d(const float* from, const float* to, int n, double thr,
                              int max_iters, double* model, uint8_t* inliers, int32_t* info) {
    const int model_points = 2;
    const double confidence = 0.99;
    int32_t dummy[4];
    if (!info) info = dummy;
    info[0] = 0; info[1] = -1; info[2] = 0; info[3] = 0;
    for (int i = 0; i < 6; i++) model[i] = NAN;
    std::vector<uint8_t> tmpmask(n > 0 ? n : 1);
    if (inliers) std::fill(inliers, inliers + n, 0);
    if (n < model_points) return 0;
    int niters = std::max(max_iters, 1);
    RNG rng((uint64_t)-1);
    double best[6] = {0, 0, 0, 0, 0, 0};
    std::vector<uint8_t> bestmask(n, 0);
    int max_good = 0;
    if (n == model_points) {
        kernel2(from, from + 2, to, to + 2, best);
        std::fill(bestmask.begin(), bestmask.end(), 1);
        max_good = n;
        info[1] = 0; info[2] = 0;
    } else {
        float t = (float)(thr * thr);
        int iter;
        for (iter = 0; iter < niters; iter++) {
            // getSubset(): two distinct indices; checkSubset() is vacuous for 2 points
            int i0 = rng.uniform(0, n);
            int i1;
            for (i1 = rng.uniform(0, n); i1 == i0; i1 = rng.uniform(0, n)) {}
            double M[6];
            kernel2(from + 2 * i0, from + 2 * i1, to + 2 * i0, to + 2 * i1, M);
            int good = find_inliers(from, to, n, M, t, tmpmask.data());
            if (good > std::max(max_good, model_points - 1)) {
                std::swap(tmpmask, bestmask);
                std::copy(M, M + 6, best);
                max_good = good;
                info[1] = iter;
                niters = update_num_iters(confidence, (double)(n - good) / n, model_points, niters);
            }
        }
        info[2] = niters;
    }
    if (max_good <= 0) return 0;
    if (n > 2) refine(from, to, n, bestmask.data(), best);
    std::copy(best, best + 6, model);
    if (inliers) std::copy(bestmask.begin(), bestmask.end(), inliers);
    info[0] = 1;
    info[3] = max_good;
    return 1;
}

}  // namespace vso

extern "C" {
void vso_rng_stream(uint64_t seed, uint32_t* out, int n) {
    vso::RNG r(seed);
    for (int i = 0; i < n; i++) out[i] = r.next();
}
int vso_estimate_affine_partial2d(const float* from, const float* to, int n, double thr,
                                  int max_iters, double* model, uint8_t* inliers, int32_t* info) {
    return vso::estimate_affine_partial2d(from, to, n, thr, max_iters, model, inliers, info);
}
}
