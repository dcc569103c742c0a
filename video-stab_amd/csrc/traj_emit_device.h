// The part of applyNextSmoothTransform() that produces the 2x3 matrix of one output frame
// (/root/reference/src/Stabilizer.cpp:783-908) as a device function, so that it can run as its own
// kernel (k_traj.hip) or inside the ordered tail of a batch (k_ransac.hip).
//
// Every float SUM of the reference is a sequential accumulation, and its order is part of the result.
// A single lane walking an LDS array pays ~100 cycles per element (dependent load + add), so the sums
// are evaluated wave-cooperatively instead: lane j loads element j (all loads in flight together), and
// the accumulation then runs over v_readlane(j) in the reference's order - the same additions, in
// the same order, on values that are already in registers.  The loops are unrolled to their maximum
// length with constant lane numbers; lanes past the end hold +0 (or the mean, for a variance), and
// adding +0 to a running float sum that started at +0 never changes it.
//
// traj_emit_device is called by ALL threads of a workgroup (any multiple of 64); wave 0 does the work.
#ifndef VS_TRAJ_EMIT_DEVICE_H
#define VS_TRAJ_EMIT_DEVICE_H

#include <cstddef>
#include <cstdint>
#include "traj_state.h"
#include "vs_common.h"
#include "vs_libm.h"

namespace vsd {

__device__ __forceinline__ float lane_value(float v, int j) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
}

// sum of lanes 0..MAXN-1 in lane order (lanes >= n must hold +0)
template <int MAXN>
__device__ __forceinline__ float seq_sum(float v) {
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXN; i++) acc += lane_value(v, i);
    return acc;
}
// sum of (v[i] - mean)^2 in lane order (lanes >= n must hold `mean`)
template <int MAXN>
__device__ __forceinline__ float seq_sqdev(float v, float mean) {
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXN; i++) { const float d = lane_value(v, i) - mean; acc += d * d; }
    return acc;
}

// Stabilizer.cpp:1750-1780 on values held one per lane (lanes 0..n-1 of v, +0 beyond), n <= MAXN
template <int MAXN>
__device__ __forceinline__ float wave_variance_of(float v, int n) {
    if (n == 0) return 0.0f;
    float mean = seq_sum<MAXN>(v);
    mean /= n;
    float var = seq_sqdev<MAXN>((int)(threadIdx.x & 63) < n ? v : mean, mean);
    var /= n;
    return var;
}
template <int MAXN>
__device__ __forceinline__ float wave_consistency_of(float v, int n) {
    if (n < 2) return 0.0f;
    const float var = wave_variance_of<MAXN>(v, n);
    float mean = seq_sum<MAXN>(v);
    mean /= n;
    if (mean == 0.0f) return 0.0f;
    const float c = 1.0f / (1.0f + (var / (mean * mean)));
    return fmaxf(0.0f, fminf(1.0f, c));
}

// One lane: T = [cos -sin dx; sin cos dy] (:902-908) from the smoothed transform t3 = {dx, dy, da, valid},
// the chroma variant, and the inverse maps the warp kernels consume.  valid == 0: identity (:774-780).
__device__ __forceinline__ void traj_matrix_lane(const float* t3, float* __restrict__ M_out, double* __restrict__ Minv_out,
                                                 vs_debug_frame* dbg) {
    if (t3[3] == 0.f) {
        M_out[0] = 1.f; M_out[1] = 0.f; M_out[2] = 0.f; M_out[3] = 0.f; M_out[4] = 1.f; M_out[5] = 0.f;
        for (int i = 0; i < 6; i++) M_out[6 + i] = M_out[i];
    } else {
        const float dx = t3[0], dy = t3[1], da = t3[2];
        const float cs = vslibm::cosf_ref(da), sn = vslibm::sinf_ref(da);     // glibc's cosf / sinf, bit for bit (vs_libm.h)
        M_out[0] = cs; M_out[1] = -sn; M_out[2] = dx;
        M_out[3] = sn; M_out[4] = cs; M_out[5] = dy;
        // chroma plane of an NV12 surface: same rotation, translation halved
        M_out[6] = cs; M_out[7] = -sn; M_out[8] = dx * 0.5f;
        M_out[9] = sn; M_out[10] = cs; M_out[11] = dy * 0.5f;
    }
    warp_invert(M_out, Minv_out);
    warp_invert(M_out + 6, Minv_out + 6);
    if (dbg) for (int i = 0; i < 6; i++) dbg->warp_matrix[i] = M_out[i];
}

// Wave 0 (64 lanes, all active): smoothing around frame idx and motion-intent gain -> the smoothed transform.
// mg / dr: magnitude and direction of transform istart + lane (lanes 0..14).  Uniform values are computed by
// every lane; lane 0 stores.  DEFER: only t3 = {dx, dy, da, valid} is produced (the caller builds the matrices,
// for all outputs of a batch in parallel); otherwise lane 0 also builds them.
template <bool DEFER>
__device__ __forceinline__ void traj_emit_wave0(TrajState* s, const TrajParams& p, int idx, float* __restrict__ M_out,
                                                double* __restrict__ Minv_out, vs_debug_frame* dbg,
                                                const float (*l_path)[3], const float (*l_tr)[3], const float mg_lane,
                                                const float dr_lane, const int n, const int istart, float* t3) {
    const int lane = threadIdx.x & 63;
    auto path_at = [&](int i, int c) -> float { return l_path[i & (TRAJ_RING - 1)][c]; };
    auto tr_at = [&](int i, int c) -> float { return l_tr[i & (TRAJ_RING - 1)][c]; };
    if (idx >= n) {   // Stabilizer.cpp:774-780: no transform for this frame -> frame returned as is
        if (lane == 0) {
            dbg->out_index = idx; dbg->box_radius = 0; dbg->intent = 0;
            for (int c = 0; c < 3; c++) dbg->smoothed[c] = 0.f;
            float id[4] = {0.f, 0.f, 0.f, 0.f};
            if (DEFER) { for (int c = 0; c < 4; c++) t3[c] = id[c]; }
            else traj_matrix_lane(id, M_out, Minv_out, dbg);
        }
        return;
    }
    int box_radius = 0;
    float sm[3];
    if (p.method == VS_SMOOTH_GAUSSIAN) {          // :1364-1413, ksize <= GAUSS_MAX (63): one tap per lane
        const int ks = p.gauss_ksize, center = ks / 2;
        float tap[3] = {0.f, 0.f, 0.f}, kw = 0.f;
        if (lane < ks) {
            // padded[idx + lane]: reflect padding as the reference builds it (clamped when n <= center)
            const int q = idx + lane;
            int src;
            if (q < center) src = center - q;
            else if (q < center + n) src = q - center;
            else src = n - 1 - (q - center - n);
            src = src < 0 ? 0 : (src > n - 1 ? n - 1 : src);
            for (int c = 0; c < 3; c++) tap[c] = path_at(src, c);
            kw = p.gauss_kernel[lane];
        }
        for (int c = 0; c < 3; c++) {
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < GAUSS_MAX; j++) acc += lane_value(tap[c], j) * lane_value(kw, j);   // taps >= ks: 0 * 0
            sm[c] = acc;
        }
    } else if (p.method == VS_SMOOTH_KALMAN) {     // :1416-1458, advanced incrementally (a recursion: lane 0)
        float r3[3] = {0.f, 0.f, 0.f};
        if (lane == 0) {
            const float q = 0.01f, r = 0.1f;
            for (int c = 0; c < 3; c++) {
                float* k = s->kal[c];   // x0,x1,P00,P01,P10,P11
                int done = s->kal_n[c];
                if (done == 0) {
                    k[0] = path_at(0, c); k[1] = 0.f; k[2] = k[3] = k[4] = k[5] = 0.f;
                    s->kal_last[c] = k[0];
                    done = 1;
                }
                while (done <= idx) {
                    const float xp0 = k[0] + k[1], xp1 = k[1];
                    const float t00 = k[2] + k[4], t01 = k[3] + k[5], t10 = k[4], t11 = k[5];
                    const float Q00 = (t00 + t01) + q, Q01 = t01, Q10 = t10 + t11, Q11 = t11 + q;
                    const float S = Q00 + r;
                    const float K0 = Q00 / S, K1 = Q01 / S;
                    const float innov = path_at(done, c) - xp0;
                    k[0] = xp0 + K0 * innov; k[1] = xp1 + K1 * innov;
                    k[2] = Q00 - K0 * Q00; k[3] = Q01 - K0 * Q01;
                    k[4] = Q10 - K1 * Q00; k[5] = Q11 - K1 * Q01;
                    s->kal_last[c] = k[0];
                    done++;
                }
                s->kal_n[c] = done;
                r3[c] = s->kal_last[c];
            }
        }
        for (int c = 0; c < 3; c++) sm[c] = lane_value(r3[c], 0);
    } else {                                         // :807-823 box with adaptive radius
        // calculateAdaptiveRadius :1637-1673: statistics of the last <= 20 path samples, one per lane
        int ar = s->smoothing_radius;
        if (n >= 10) {
            const int start = n - 20 > 0 ? n - 20 : 0;
            const int count = n - start;
            float v[3] = {0.f, 0.f, 0.f};
            if (lane < count)
                for (int c = 0; c < 3; c++) v[c] = path_at(start + lane, c);
            float mean[3], var[3];
            for (int c = 0; c < 3; c++) mean[c] = seq_sum<20>(v[c]);
            for (int c = 0; c < 3; c++) mean[c] /= count;
            for (int c = 0; c < 3; c++) var[c] = seq_sqdev<20>(lane < count ? v[c] : mean[c], mean[c]);
            for (int c = 0; c < 3; c++) var[c] /= count;
            const float total = sqrtf(var[0] + var[1] + var[2] * 1000);
            ar = (int)fmaxf(5.0f, fminf(25.0f, total * 2.0f));
        }
        // boxFilterConvolve :1139-1172: window of at most 2*50+1 samples, 64 per pass
        const int r = p.drone ? max(10, min(ar, 50)) : max(2, min(ar, 8));
        box_radius = r;
        if (n <= r) {
            for (int c = 0; c < 3; c++) sm[c] = path_at(idx, c);
        } else {
            const int start = idx - r > 0 ? idx - r : 0;
            const int end = idx + r < n - 1 ? idx + r : n - 1;
            const int count = end - start + 1;
            float sum[3] = {0.f, 0.f, 0.f};
            if (count <= 17) {            // radius <= 8: the usual case
                float v[3] = {0.f, 0.f, 0.f};
                if (lane < count)
                    for (int c = 0; c < 3; c++) v[c] = path_at(start + lane, c);
                for (int c = 0; c < 3; c++) sum[c] = seq_sum<17>(v[c]);
            } else {                      // drone mode: up to 101 samples, 64 per pass
                for (int base = start; base <= end; base += 64) {
                    const int m = end - base + 1 < 64 ? end - base + 1 : 64;
                    float v[3] = {0.f, 0.f, 0.f};
                    if (lane < m)
                        for (int c = 0; c < 3; c++) v[c] = path_at(base + lane, c);
                    for (int c = 0; c < 3; c++)
                        for (int j = 0; j < m; j++) sum[c] += lane_value(v[c], j);
                }
            }
            for (int c = 0; c < 3; c++) sm[c] = sum[c] / count;
        }
    }
    float raw[3], diff[3];
    for (int c = 0; c < 3; c++) {
        raw[c] = tr_at(idx, c);
        diff[c] = sm[c] - path_at(idx, c);      // :850-851
    }
    int intent = 0;
    if (idx > 0) {                                   // :854-888, analyzeMotionIntent :1676-1719
        const float magnitude = sqrtf(raw[0] * raw[0] + raw[1] * raw[1]);
        const float angularVel = (float)((double)(fabsf(raw[2]) * 180.0f) / 3.14159265358979323846 * (double)30.0f);
        if (n >= 15) {
            // samples istart .. min(idx, n) - 1, one per lane (l_mag / l_dir were filled by lanes 0..14)
            const int last = idx < n ? idx : n;
            const int cnt = last - istart > 0 ? last - istart : 0;
            const float mg = lane < cnt ? mg_lane : 0.f, dr = lane < cnt ? dr_lane : 0.f;
            if (cnt > 0) {
                const float dv = wave_variance_of<15>(dr, cnt);
                const float mc = wave_consistency_of<15>(mg, cnt);
                if (dv < 0.5f && mc > 0.7f && magnitude > 5.0f) intent = 1;
                else if (magnitude < 3.0f && mc < 0.3f && angularVel > 10.0f) intent = 2;
                else if (magnitude > 3.0f && magnitude < 15.0f && dv > 0.5f) intent = 3;
            }
        }
        // calculateAdaptiveStabilizationStrength :1722-1747 is consumed only for NORMAL (0.7)
        const float g = intent == 1 ? 0.5f : intent == 2 ? 1.0f : intent == 3 ? 0.8f : 0.7f;
        for (int c = 0; c < 3; c++) diff[c] *= g;
    }
    if (lane != 0) return;
    dbg->out_index = idx; dbg->box_radius = box_radius; dbg->intent = intent;
    for (int c = 0; c < 3; c++) dbg->smoothed[c] = sm[c];
    const float dx = raw[0] + diff[0], dy = raw[1] + diff[1];
    float da = raw[2] + diff[2];
    if (p.horizon_lock) da = 0.0f;                   // :897-899
    const float r4[4] = {dx, dy, da, 1.f};
    if (DEFER) { for (int c = 0; c < 4; c++) t3[c] = r4[c]; }
    else {
        traj_matrix_lane(r4, M_out, Minv_out, dbg);
        if (t3) for (int c = 0; c < 3; c++) t3[c] = r4[c];       // the correction itself, for the virtual canvas
    }
}

// magnitude / direction of transform istart + lane, for the intent analysis (lanes 0..14 of wave 0)
__device__ __forceinline__ void traj_intent_samples(const float (*l_tr)[3], int idx, int n, int istart, float& mg, float& dr) {
    mg = 0.f; dr = 0.f;
    const int lane = threadIdx.x & 63;
    const int i = istart + lane;
    if (lane < 15 && i < idx && i < n) {
        const float t0 = l_tr[i & (TRAJ_RING - 1)][0], t1 = l_tr[i & (TRAJ_RING - 1)][1];
        mg = sqrtf(t0 * t0 + t1 * t1);
        dr = vslibm::atan2f_ref(t1, t0);
    }
}

// The same for a trajectory state that already lives in LDS (batch tail): wave 0 only, no workgroup barrier.
// n: transforms appended when this output is released (the rings may already hold later ones).  Any full wave.
__device__ __forceinline__ void traj_emit_lds_wave0(TrajState* s, const TrajParams& p, int idx, int n, vs_debug_frame* dbg, float* t3) {
    const int istart = idx - 15 > 0 ? idx - 15 : 0;
    float mg, dr;
    traj_intent_samples(s->transforms, idx, n, istart, mg, dr);
    traj_emit_wave0<true>(s, p, idx, nullptr, nullptr, dbg, s->path, s->transforms, mg, dr, n, istart, t3);
}

// n_seen >= 0: the number of transforms this release is to see (the rings may already hold later ones: batch mode)
__device__ __forceinline__ void traj_emit_device(TrajState* s, const TrajParams& p, int idx, float* __restrict__ M_out,
                                                 double* __restrict__ Minv_out, vs_debug_frame* dbg, float* t_out = nullptr,
                                                 int n_seen = -1) {
    // The history rings are mirrored into LDS by all lanes and the per-sample transcendental work of
    // the intent analysis is spread over lanes.
    __shared__ __attribute__((aligned(16))) float l_path[TRAJ_RING][3], l_tr[TRAJ_RING][3];
    const int n = n_seen >= 0 ? n_seen : s->n;
    static_assert((TRAJ_RING * 3) % (4 * 64) == 0 && offsetof(TrajState, transforms) % 16 == 0 && offsetof(TrajState, path) % 16 == 0, "ring copy");
    if (((uintptr_t)s & 15) == 0) {
        // the first wave (the release kernel, the per-frame emit): three 16-byte loads per ring and lane, all in flight before the
        // first LDS store - as a loop over blockDim the compiler made load - wait - store of each of its 12 rounds (24 round trips)
        constexpr int NV = TRAJ_RING * 3 / (4 * 64);
        const float4* gp = reinterpret_cast<const float4*>(&s->path[0][0]);
        const float4* gt = reinterpret_cast<const float4*>(&s->transforms[0][0]);
        if (threadIdx.x < 64) {
            float4 a[NV], b[NV];
#pragma unroll
            for (int k = 0; k < NV; k++) { a[k] = gp[threadIdx.x + 64 * k]; b[k] = gt[threadIdx.x + 64 * k]; }
#pragma unroll
            for (int k = 0; k < NV; k++) {
                reinterpret_cast<float4*>(&l_path[0][0])[threadIdx.x + 64 * k] = a[k];
                reinterpret_cast<float4*>(&l_tr[0][0])[threadIdx.x + 64 * k] = b[k];
            }
        }
    } else {
        for (int i = threadIdx.x; i < TRAJ_RING * 3; i += blockDim.x) {
            (&l_path[0][0])[i] = (&s->path[0][0])[i];
            (&l_tr[0][0])[i] = (&s->transforms[0][0])[i];
        }
    }
    __syncthreads();
    const int istart = idx - 15 > 0 ? idx - 15 : 0;
    if (threadIdx.x < 64) {
        float mg, dr;
        traj_intent_samples(l_tr, idx, n, istart, mg, dr);
        traj_emit_wave0<false>(s, p, idx, M_out, Minv_out, dbg, l_path, l_tr, mg, dr, n, istart, t_out);
    }
}

}  // namespace vsd
#endif
