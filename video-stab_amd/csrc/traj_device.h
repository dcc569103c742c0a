// Device-side body of the transform append step (shared by the stand-alone
// kernel in k_traj.hip and the fused RANSAC-select kernel in k_ransac.hip).
// Restates /root/reference/src/Stabilizer.cpp:644-693 and the drone filters
// (:2468-2553, :2605-2682).
#ifndef VS_TRAJ_DEVICE_H
#define VS_TRAJ_DEVICE_H

#include "traj_state.h"
#include "vs_common.h"
#include "vs_libm.h"

namespace vsd {

__device__ __forceinline__ float hf_mag(const float t[3]) {
    return sqrtf(t[0] * t[0] + t[1] * t[1] + t[2] * t[2] * 100.0f);
}

// One lane.  `info` = {ok, best_iter, iters_run, n_inliers}, `model` = refined 2x3 (double).
__device__ __forceinline__ void traj_append_device(TrajState* s, const TrajParams& p, const double* model,
                                          const int32_t* info, int nprev, vs_debug_frame* dbg,
                                          int have_prev_gray) {
    float tr[3] = {0.f, 0.f, 0.f};
    dbg->ransac_best_iter = -1; dbg->ransac_iters_run = 0; dbg->n_inliers = 0;
    for (int i = 0; i < 6; i++) dbg->model[i] = __longlong_as_double(0x7FF8000000000000LL);
    if (nprev > 0 && have_prev_gray) {              // Stabilizer.cpp:596
        float T[6] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f};  // :644
        if (info[0]) {                               // :650-652
            for (int i = 0; i < 6; i++) { T[i] = (float)model[i]; dbg->model[i] = model[i]; }
        }
        dbg->ransac_best_iter = info[1]; dbg->ransac_iters_run = info[2]; dbg->n_inliers = info[3];
        tr[0] = T[2]; tr[1] = T[5]; tr[2] = vslibm::atan2f_ref(T[3], T[0]);   // :660-662
        if (p.drone) {
            // applyDeadZoneFreeze :2605-2655 (updateMotionAccumulator :2667-2682)
            const float magnitude = hf_mag(tr);
            const float decayed = s->hfAccum * p.hf_decay;
            s->hfAccum = fmaxf(decayed, magnitude);
            s->hfAccum = fminf(s->hfAccum, p.hf_dead_zone * 5.0f);
            s->hfAccum = fmaxf(0.0f, fminf(s->hfAccum, 100.0f));
            const float cur = magnitude;
            bool frozen = false;
            if (!s->hfInDeadZone && cur < p.hf_dead_zone) { s->hfInDeadZone = 1; s->hfFreezeCounter = p.hf_freeze_duration; }
            if (s->hfInDeadZone) {
                s->hfFreezeCounter--;
                const bool durationExpired = s->hfFreezeCounter <= 0;
                const bool significantMotion = cur > p.hf_dead_zone * 1.5f;
                const bool accumulatedMotion = s->hfAccum > p.hf_dead_zone * 1.2f;
                if (durationExpired || significantMotion || accumulatedMotion) {
                    s->hfInDeadZone = 0; s->hfFreezeCounter = 0; s->hfAccum = 0.0f;
                } else frozen = true;
            }
            if (frozen) { tr[0] = tr[1] = tr[2] = 0.0f; }
            // applyMicroShakeSuppression :2468-2503
            if (s->hfHistN >= 5) {
                float xs[10], ys[10];
                const int hn = s->hfHistN;
                for (int i = 0; i < hn; i++) { xs[i] = s->hfHist[i][0]; ys[i] = s->hfHist[i][1]; }
                for (int i = 1; i < hn; i++) {   // insertion sort (values only; any stable sort gives the same order statistics)
                    float vx = xs[i]; int j = i - 1;
                    while (j >= 0 && xs[j] > vx) { xs[j + 1] = xs[j]; j--; }
                    xs[j + 1] = vx;
                    float vy = ys[i]; j = i - 1;
                    while (j >= 0 && ys[j] > vy) { ys[j + 1] = ys[j]; j--; }
                    ys[j + 1] = vy;
                }
                const int mid = hn / 2;
                s->hfMedian[0] = hn % 2 == 0 ? (xs[mid - 1] + xs[mid]) / 2.0f : xs[mid];
                s->hfMedian[1] = hn % 2 == 0 ? (ys[mid - 1] + ys[mid]) / 2.0f : ys[mid];
            }
            const float d0 = tr[0] - s->hfMedian[0], d1 = tr[1] - s->hfMedian[1];
            const float dm = sqrtf(d0 * d0 + d1 * d1);
            if (dm < p.hf_shake_px) {
                tr[0] = s->hfMedian[0] + d0 * 0.01f; tr[1] = s->hfMedian[1] + d1 * 0.01f;
            } else if (dm < p.hf_shake_px * 2.0f) {
                tr[0] = s->hfMedian[0] + d0 * 0.05f; tr[1] = s->hfMedian[1] + d1 * 0.05f;
            }
            // applyRotationLowPass :2505-2520
            if (p.horizon_lock) {
                s->hfRotLP = (1.0f - p.hf_rot_lp_alpha) * s->hfRotLP + p.hf_rot_lp_alpha * tr[2];
                tr[2] = s->hfRotLP;
            }
            // updateTranslationHistory :2522-2529 (deque of the last 10)
            if (s->hfHistN < 10) { s->hfHist[s->hfHistN][0] = tr[0]; s->hfHist[s->hfHistN][1] = tr[1]; s->hfHistN++; }
            else {
                for (int i = 0; i < 9; i++) { s->hfHist[i][0] = s->hfHist[i + 1][0]; s->hfHist[i][1] = s->hfHist[i + 1][1]; }
                s->hfHist[9][0] = tr[0]; s->hfHist[9][1] = tr[1];
            }
        }
    }
    const int n = s->n;
    const int slot = n & (TRAJ_RING - 1);
    float pth[3];
    for (int c = 0; c < 3; c++) {
        s->transforms[slot][c] = tr[c];
        pth[c] = n == 0 ? tr[c] : s->last_path[c] + tr[c];   // :681-687
        s->path[slot][c] = pth[c];
        s->last_path[c] = pth[c];
        dbg->transform[c] = tr[c];
    }
    s->n = n + 1;
    // updateAdaptiveParameters :1562-1574 -> adaptSmoothingRadius :1461-1492
    if (p.adaptive && n + 1 >= 3) {
        const float magnitude = sqrtf(tr[0] * tr[0] + tr[1] * tr[1]);
        float motionScale = fmaxf(0.0f, fminf(1.0f, magnitude / 50.0f));
        motionScale = 1.0f - motionScale;
        const int newRadius = p.min_radius + (int)(motionScale * (float)(p.max_radius - p.min_radius));
        if (newRadius != s->smoothing_radius) s->smoothing_radius = newRadius;
    }
}

}  // namespace vsd
#endif
