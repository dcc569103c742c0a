// Small per-frame device kernels that keep stabilize() free of host round trips.
// (The status compaction, /root/reference/src/Stabilizer.cpp:629-641, is fused into the RANSAC scoring
// kernel and the transform append, :644-693 + drone filters :2468-2682, into the RANSAC selection:
// k_ransac.hip, traj_device.h.)
//
//  traj_emit_kernel - the part of applyNextSmoothTransform() that produces the
//      2x3 matrix (:783-908): box / gaussian / kalman smoothing evaluated only
//      around the frame that leaves the queue (the reference re-smooths the
//      whole history per frame), motion-intent gain, T = [cos -sin dx; sin cos dy].
//
// History lives in device ring buffers (last RING entries of transforms_ and
// path_); the Kalman filter is advanced incrementally (it is a forward
// recursion, so resuming from the stored state reproduces the reference's
// from-scratch run bit for bit).
#include "vs_common.h"
#include "traj_state.h"
#include "traj_device.h"
#include "traj_emit_device.h"

namespace vsd {
namespace {

__global__ __launch_bounds__(64) void traj_emit_kernel(TrajState* s, TrajParams p, int idx, float* __restrict__ M_out,
                                                       double* __restrict__ Minv_out, vs_debug_frame* dbg, float* t_out) {
    traj_emit_device(s, p, idx, M_out, Minv_out, dbg, t_out);
}

// test hook (VS_STAB_DEBUG_DELAY_US): one wave that does nothing for about `ticks` of the 100 MHz clock
__global__ void spin_kernel(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

__global__ void traj_reset_kernel(TrajState* s, int smoothing_radius) {
    if (threadIdx.x != 0) return;
    s->n = 0;
    for (int c = 0; c < 3; c++) { s->last_path[c] = 0.f; s->kal_n[c] = 0; s->kal_last[c] = 0.f; }
    s->smoothing_radius = smoothing_radius;
    s->hfMedian[0] = s->hfMedian[1] = 0.f; s->hfRotLP = 0.f;
    s->hfInDeadZone = 0; s->hfFreezeCounter = 0; s->hfAccum = 0.f; s->hfHistN = 0;
}

// Border pre-pad: cv::copyMakeBorder (Stabilizer.cpp:981-990)
__device__ __forceinline__ int border_index(int p, int len, int border) {
    if ((unsigned)p < (unsigned)len) return p;
    switch (border) {
        case VS_BORDER_REPLICATE: return p < 0 ? 0 : len - 1;
        case VS_BORDER_REFLECT:
        case VS_BORDER_REFLECT_101: {
            const int delta = border == VS_BORDER_REFLECT_101;
            if (len == 1) return 0;
            do {
                if (p < 0) p = -p - 1 + delta;
                else p = len - 1 - (p - len) - delta;
            } while ((unsigned)p >= (unsigned)len);
            return p;
        }
        case VS_BORDER_WRAP:
            if (p < 0) p -= ((p - len + 1) / len) * len;
            if (p >= len) p %= len;
            return p;
        default: return -1;
    }
}

__global__ __launch_bounds__(256) void make_border_kernel(const uint8_t* __restrict__ src, size_t sstride, int w, int h,
                                                          int cn, uint8_t* __restrict__ dst, size_t dstride, int b,
                                                          int border) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int ow = w + 2 * b;
    if (x >= ow) return;
    const int sy = border_index(y - b, h, border), sx = border_index(x - b, w, border);
    for (int k = 0; k < cn; k++)
        dst[(size_t)y * dstride + (size_t)x * cn + k] = (sx >= 0 && sy >= 0) ? src[(size_t)sy * sstride + (size_t)sx * cn + k] : 0;
}

// borderType "fade" (Stabilizer.cpp:914-978): frame = addWeighted(history, alpha, frame, 1 - alpha) over the whole padded
// frame (the reference's border mask is all 255), cv::addWeighted on 8-bit data = fma(a, alpha, fma(b, beta, 0)) in float,
// rounded half to even and saturated.  One thread = 4 bytes.
__global__ __launch_bounds__(256) void fade_blend_kernel(const uint8_t* __restrict__ hist, uint8_t* __restrict__ frame, size_t n4, float alpha,
                                                        float beta) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const uint32_t hv = reinterpret_cast<const uint32_t*>(hist)[i], fv = reinterpret_cast<const uint32_t*>(frame)[i];
    uint32_t o = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float a = (float)((hv >> (8 * k)) & 255u), b = (float)((fv >> (8 * k)) & 255u);
        int v = __float2int_rn(fmaf(a, alpha, fmaf(b, beta, 0.0f)));
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        o |= (uint32_t)v << (8 * k);
    }
    reinterpret_cast<uint32_t*>(frame)[i] = o;
}

// history = (uchar)((1 - 0.1f) * history + 0.1f * stabilized) for every sample (:1086-1100; two float products, one sum,
// truncation - the library is built without contraction).  `stab` has row pitch sstride, the history is packed.
__global__ __launch_bounds__(256) void fade_update_kernel(uint8_t* __restrict__ hist, const uint8_t* __restrict__ stab, size_t sstride, int row_bytes,
                                                          int rows) {
    const int x = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int y = blockIdx.y;
    if (x >= row_bytes || y >= rows) return;
    uint8_t* h = hist + (size_t)y * row_bytes + x;
    const uint8_t* s = stab + (size_t)y * sstride + x;
    const float updateRate = 0.1f;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (x + k < row_bytes) h[k] = (uint8_t)((1.0f - updateRate) * (float)h[k] + updateRate * (float)s[k]);
}

}  // namespace

int launch_fade_blend(const uint8_t* d_hist, uint8_t* d_frame, size_t bytes, float alpha, float beta, hipStream_t st) {
    const size_t n4 = bytes / 4;        // (padded frames of the stabilizer: bytes is a multiple of 4 after rounding the buffer)
    hipLaunchKernelGGL(fade_blend_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, d_hist, d_frame, n4, alpha, beta);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

int launch_fade_update(uint8_t* d_hist, const uint8_t* d_stab, size_t sstride, int row_bytes, int rows, hipStream_t st) {
    hipLaunchKernelGGL(fade_update_kernel, dim3((row_bytes / 4 + 1 + 255) / 256, rows), dim3(256), 0, st, d_hist, d_stab, sstride, row_bytes, rows);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

int launch_traj_emit(TrajState* s, const TrajParams& p, int idx, float* M_out, double* Minv_out, vs_debug_frame* dbg,
                     hipStream_t st, float* t_out) {
    hipLaunchKernelGGL(traj_emit_kernel, dim3(1), dim3(64), 0, st, s, p, idx, M_out, Minv_out, dbg, t_out);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}
int launch_spin(int microseconds, hipStream_t st) {
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, (unsigned long long)microseconds * 100ull);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

int launch_traj_reset(TrajState* s, int smoothing_radius, hipStream_t st) {
    hipLaunchKernelGGL(traj_reset_kernel, dim3(1), dim3(64), 0, st, s, smoothing_radius);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}
int launch_make_border(const uint8_t* src, size_t sstride, int w, int h, int cn, uint8_t* dst, size_t dstride,
                       int b, int border, hipStream_t st) {
    dim3 grid((w + 2 * b + 255) / 256, h + 2 * b);
    hipLaunchKernelGGL(make_border_kernel, grid, dim3(256), 0, st, src, sstride, w, h, cn, dst, dstride, b, border);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

}  // namespace vsd
