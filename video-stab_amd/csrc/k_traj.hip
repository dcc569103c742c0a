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
#include <algorithm>
#include <string>

#include "vs_common.h"
#include "vs_libm.h"
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

// Checksum of a libm restatement (vs_libm.h) over a range of arguments, for the parity test against the host's libm
// (tests/test_libm.py): fn 0 cosf, 1 sinf, 2 atanf - argument i is the float with bit pattern (uint32)(start + i) -, 3 atan2f -
// argument pair i from libm_pair().  The sum of mix(i, result bits) over the range, NaN results as one pattern.
__host__ __device__ inline uint64_t libm_mix(uint64_t i, uint32_t bits) {
    uint64_t z = (i * 0x9E3779B97F4A7C15ull) ^ (uint64_t)bits;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ inline void libm_pair(uint64_t i, float* y, float* x) {
    uint64_t v = libm_mix(i, 0x5EEDu);
    if (i & 1) { *y = vslibm::u2f((uint32_t)v); *x = vslibm::u2f((uint32_t)(v >> 32)); return; }      // any two bit patterns
    // shaped like the stabilizer's arguments: (sin, cos) of a small rotation times a scale near 1, or pixel translations
    float xx = 0.9f + 0.2f * ((float)(v & 0xFFFFFFu) / 16777216.0f);
    float yy = ((float)((v >> 24) & 0xFFFFFFu) / 16777216.0f - 0.5f) * (((v >> 48) & 1) ? 0.5f : 0.02f);
    if ((v >> 49) & 1) { xx = (xx - 1.0f) * 200.0f; yy *= 100.0f; }
    *y = yy; *x = xx;
}
__global__ __launch_bounds__(256) void libm_checksum_kernel(int fn, uint64_t start, uint64_t count, unsigned long long* out) {
    uint64_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (uint64_t)gridDim.x * 256) {
        const uint64_t idx = start + i;
        float r;
        if (fn == 3) { float y, x; libm_pair(idx, &y, &x); r = vslibm::atan2f_ref(y, x); }
        else {
            const float x = vslibm::u2f((uint32_t)idx);
            r = fn == 0 ? vslibm::cosf_ref(x) : (fn == 1 ? vslibm::sinf_ref(x) : vslibm::atanf_ref(x));
        }
        acc += libm_mix(idx, r != r ? 0x7FC00000u : vslibm::f2u(r));
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, (unsigned long long)acc);
}

// Bandwidth yardstick (vs_dev_copy_rate): a plain copy, 16 bytes per lane, four in flight per lane, streaming stores.
typedef uint32_t copy_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy_rate_kernel(const copy_u32x4* __restrict__ src, copy_u32x4* __restrict__ dst, size_t n16) {
    const size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x);
    const size_t stride = (size_t)gridDim.x * 256;
    copy_u32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) if (i0 + k * stride < n16) v[k] = src[i0 + k * stride];
#pragma unroll
    for (int k = 0; k < 4; k++) if (i0 + k * stride < n16) __builtin_nontemporal_store(v[k], &dst[i0 + k * stride]);
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

// vs_dev_copy_rate: `iters` back-to-back copies of `bytes` bytes between HIP events; source and destination walk through
// rings of buffers that together exceed 512 MB, so that no copy finds its source in the 256 MB Infinity Cache.
int run_copy_rate(size_t bytes, int iters, double* gbps) {
    const size_t n16 = bytes / 16, b = n16 * 16;
    const int ring = (int)std::max<size_t>(2, ((size_t)640 << 20) / b + 1);
    uint8_t *src = nullptr, *dst = nullptr;
    VS_HIP_TRY(hipMalloc((void**)&src, b * ring));
    if (hipMalloc((void**)&dst, b * ring) != hipSuccess) { (void)hipFree(src); set_last_error("vs_dev_copy_rate: out of device memory"); return VS_ERR_HIP; }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMemset(src, 0x5a, b * ring);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    const unsigned grid = (unsigned)((n16 + 4 * 256 - 1) / (4 * 256));
    float ms = 0.f;
    if (e == hipSuccess) {
        for (int i = 0; i < 3; i++)
            hipLaunchKernelGGL(copy_rate_kernel, dim3(grid), dim3(256), 0, nullptr, (const copy_u32x4*)(src + b * (i % ring)), (copy_u32x4*)(dst + b * (i % ring)), n16);
        e = hipEventRecord(e0, nullptr);
        for (int i = 0; i < iters; i++)
            hipLaunchKernelGGL(copy_rate_kernel, dim3(grid), dim3(256), 0, nullptr, (const copy_u32x4*)(src + b * ((i + 3) % ring)), (copy_u32x4*)(dst + b * ((i + 3) % ring)), n16);
        if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(src); (void)hipFree(dst);
    if (e != hipSuccess) { set_last_error(std::string("vs_dev_copy_rate: ") + hipGetErrorString(e)); return VS_ERR_HIP; }
    *gbps = ms > 0.f ? 2.0 * (double)b * iters / (ms * 1e-3) / 1e9 : 0.0;
    return VS_OK;
}

int run_libm_checksum(int fn, uint64_t start, uint64_t count, uint64_t* result) {
    unsigned long long* d = nullptr;
    VS_HIP_TRY(hipMalloc((void**)&d, sizeof *d));
    hipError_t e = hipMemset(d, 0, sizeof *d);
    if (e == hipSuccess) {
        const unsigned grid = (unsigned)std::min<uint64_t>((count + 255) / 256, 8192);
        hipLaunchKernelGGL(libm_checksum_kernel, dim3(grid ? grid : 1), dim3(256), 0, nullptr, fn, start, count, d);
        e = hipGetLastError();
    }
    unsigned long long h = 0;
    if (e == hipSuccess) e = hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) { set_last_error(std::string("vs_op_libm_checksum: ") + hipGetErrorString(e)); return VS_ERR_HIP; }
    *result = h;
    return VS_OK;
}

}  // namespace vsd
