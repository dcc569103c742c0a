// Auto zoom/crop for gfx950: counterpart of vs::AutoZoomCrop::autoZoomCrop
// (/root/reference/src/AutoZoomCrop.cpp:102-283).
//
// Split of the work (the reference has the same split, AutoZoomCrop.cpp:141-147: the mask is
// downloaded for cv::findContours on the CPU):
//   device  threshold_bits_kernel: BGR2GRAY + threshold(>1) as a bit plane (3 B/px read, 1 bit/px written)
//           close5_bits_kernel   : MORPH_CLOSE(5x5 ellipse) on the bit plane, 64 pixels per register
//   host    crop_from_mask      : border following on the bit mask (pointer chasing along one
//                                 contour: serial by nature), filled interior as row spans, the
//                                 shrink loop of :189-205 on the spans
//   device  warp_affine_kernel  : crop + scale to 640x360 (cv::warpAffine semantics, k_warp.hip)
// Only the mask (one bit per pixel) crosses PCIe between the two device stages; the frame stays in HBM.
#include <algorithm>
#include <climits>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "azc_contour.h"
#include <chrono>
#include <condition_variable>
#include <deque>
#include <thread>

#include "vs_common.h"

namespace vsd {

int launch_warp_affine_inv(const uint8_t* d_src, size_t sstride, int sw, int sh, uint8_t* d_dst, size_t dstride,
                           int dw, int dh, int cn, const double* h_Minv, int border, hipStream_t st);

namespace {

typedef unsigned long long u64;

struct __attribute__((aligned(4))) U3 { uint32_t a, b, c; };

// gray = (B*3735 + G*19235 + R*9798 + 2^14) >> 15 (BGR2GRAY); threshold(gray, 1, 255, BINARY) keeps gray >= 2
__device__ __forceinline__ unsigned content_bgr(uint32_t b, uint32_t g, uint32_t r) {
    return b * 3735u + g * 19235u + r * 9798u + (1u << 14) >= (2u << 15) ? 1u : 0u;
}

// Pass 1: BGR2GRAY + threshold as a bit plane T (64 pixels per word, wpr words per row, bits past the width 0).
// A lane takes 4 pixels (12 bytes, three dword loads when the rows allow), a workgroup 1024 pixels of one row;
// the 4-bit results go through LDS and 16 lanes pack them into the 16 words.
// (several pictures per launch: blockIdx.z selects the picture - its source from `srcs`, a kernel argument, its bit plane tfb
// words behind T)
constexpr int SRC_LIST_MAX = 8;
struct SrcList { const uint8_t* p[SRC_LIST_MAX]; };
template <int CN>
__global__ __launch_bounds__(256) void threshold_bits_kernel(const uint8_t* __restrict__ src, size_t stride, int w,
                                                             int aligned, u64* __restrict__ T, int wpr, SrcList srcs, size_t tfb) {
    __shared__ __attribute__((aligned(16))) uint8_t nib[256];
    if (!src) { src = srcs.p[blockIdx.z]; T += (size_t)blockIdx.z * tfb; }
    const int tid = threadIdx.x, x = (blockIdx.x * 256 + tid) * 4, y = blockIdx.y;
    const uint8_t* row = src + (size_t)y * stride;
    unsigned n = 0;
    if (x + 3 < w && aligned) {
        if (CN == 3) {
            const U3 d = *reinterpret_cast<const U3*>(row + (size_t)x * 3);
            n = content_bgr(d.a & 255u, (d.a >> 8) & 255u, (d.a >> 16) & 255u) |
                content_bgr(d.a >> 24, d.b & 255u, (d.b >> 8) & 255u) << 1 |
                content_bgr((d.b >> 16) & 255u, d.b >> 24, d.c & 255u) << 2 |
                content_bgr((d.c >> 8) & 255u, (d.c >> 16) & 255u, d.c >> 24) << 3;
        } else {
            const uint32_t d = *reinterpret_cast<const uint32_t*>(row + x);
            n = ((d & 255u) > 1u) | (((d >> 8) & 255u) > 1u) << 1 | (((d >> 16) & 255u) > 1u) << 2 | ((d >> 24) > 1u) << 3;
        }
    } else {
        for (int i = 0; i < 4; i++) {
            if (x + i >= w) break;
            const uint8_t* p = row + (size_t)(x + i) * CN;
            n |= (CN == 3 ? content_bgr(p[0], p[1], p[2]) : (p[0] > 1 ? 1u : 0u)) << i;
        }
    }
    nib[tid] = (uint8_t)n;
    __syncthreads();
    const int word = blockIdx.x * 16 + tid;
    if (tid < 16 && word < wpr) {
        const uint4 q = reinterpret_cast<const uint4*>(nib)[tid];
        auto squeeze = [](uint32_t d) {          // four nibbles, one per byte -> 16 bits
            return (d & 0xFu) | ((d >> 4) & 0xF0u) | ((d >> 8) & 0xF00u) | ((d >> 12) & 0xF000u);
        };
        const u64 lo = squeeze(q.x) | (squeeze(q.y) << 16), hi = squeeze(q.z) | (squeeze(q.w) << 16);
        T[(size_t)y * wpr + word] = lo | (hi << 32);
    }
}

// Pass 1 for one-channel pictures whose width is a multiple of 16 and whose rows are 16-byte aligned (the luma planes of the
// asynchronous NV12 path): a lane takes 16 pixels with ONE 16-byte load, a wave 1024 pixels of a row, a workgroup four rows; the
// lane's 16 bits meet their three neighbours' through two lane exchanges - no LDS, no barrier, a quarter of the workgroups.
// (Measured against the general kernel above on eight 3840 x 2160 luma planes per launch: see DESIGN.md section 5.)
__device__ __forceinline__ uint32_t above1_nibble(uint32_t d) {         // bit k = byte k of d > 1
    const uint32_t t = (d | ((d & 0x7F7F7F7Fu) + 0x7E7E7E7Eu)) & 0x80808080u;      // bit 7 of a byte: byte >= 2 (no carries between bytes)
    return (((t >> 7) * 0x00204081u) >> 21) & 0xFu;                                // bits 0, 8, 16, 24 -> 21, 22, 23, 24
}
__global__ __launch_bounds__(256) void threshold_bits_gray16_kernel(size_t stride, int w, int h, u64* __restrict__ T, int wpr, SrcList srcs,
                                                                    size_t tfb) {
    const uint8_t* __restrict__ src = srcs.p[blockIdx.z];
    T += (size_t)blockIdx.z * tfb;
    const int lane = threadIdx.x & 63, y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int x = blockIdx.x * 1024 + lane * 16, word = blockIdx.x * 16 + (lane >> 2);
    if (y >= h) return;                                                  // (wave-uniform)
    u64 bits = 0;
    if (x < w) {                                                         // (w is a multiple of 16: the lane's pixels are all inside or all outside)
        const uint4 d = *reinterpret_cast<const uint4*>(src + (size_t)y * stride + x);
        bits = (u64)(above1_nibble(d.x) | above1_nibble(d.y) << 4 | above1_nibble(d.z) << 8 | above1_nibble(d.w) << 12) << (16 * (lane & 3));
    }
    bits |= __shfl_xor(bits, 1);
    bits |= __shfl_xor(bits, 2);
    if ((lane & 3) == 0 && word < wpr) T[(size_t)y * wpr + word] = bits;
}

constexpr int MC_ROWS = 56;      // rows a wave of the morphology pass owns (lanes 4..59; halo 4 above and below)

// OR / AND of a row with itself moved by -2..2 pixels; l, r: the words left and right of it
__device__ __forceinline__ u64 spread5(u64 l, u64 t, u64 r) {
    return t | (t << 1) | (l >> 63) | (t << 2) | (l >> 62) | (t >> 1) | (r << 63) | (t >> 2) | (r << 62);
}
__device__ __forceinline__ u64 shrink5(u64 l, u64 t, u64 r) {
    return t & ((t << 1) | (l >> 63)) & ((t << 2) | (l >> 62)) & ((t >> 1) | (r << 63)) & ((t >> 2) | (r << 62));
}
__device__ __forceinline__ u64 row_from(u64 v, int d, int lane) {     // value of lane + d, zero past the wave
    const u64 t = __shfl(v, lane + d);
    return (unsigned)(lane + d) < 64u ? t : 0;
}

// Pass 2: MORPH_CLOSE with the 5x5 ellipse (rows -1..1 five wide, rows -2 and +2 the centre only) on the bit
// plane: wave = one word column, lane = row, a row of 64 pixels one register.  Dilation sees zeros outside the
// image, erosion ones (cv::morphologyEx border value).  The result goes out as a zero-framed BitFrame (host
// side below) or, for vs_op_content_mask, as a plain bit plane that expand_bits_kernel turns into bytes.
__global__ __launch_bounds__(64) void close5_bits_kernel(const u64* __restrict__ T, int wpr, int w, int h,
                                                         u64* __restrict__ out, int opitch, int oframe, size_t tfb, size_t ofb) {
    T += (size_t)blockIdx.z * tfb; out += (size_t)blockIdx.z * ofb;      // (words between the pictures of a launch)
    const int lane = threadIdx.x, k = blockIdx.x, y = blockIdx.y * MC_ROWS - 4 + lane;
    const bool row_in = y >= 0 && y < h;
    auto word_in = [&](int kk) { return kk >= 0 && kk < wpr; };
    auto ones_outside = [&](int kk) -> u64 {        // bits of word kk that lie outside the image
        if (!row_in || !word_in(kk)) return ~0ull;
        const int left = w - 64 * kk;                // pixels of this word inside the image
        return left >= 64 ? 0ull : ~0ull << left;
    };
    u64 t[5];
#pragma unroll
    for (int j = 0; j < 5; j++) t[j] = (row_in && word_in(k - 2 + j)) ? T[(size_t)y * wpr + k - 2 + j] : 0ull;
    u64 d[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {                    // dilated words k-1, k, k+1
        const u64 hrow = spread5(t[j], t[j + 1], t[j + 2]);
        d[j] = row_from(hrow, -1, lane) | hrow | row_from(hrow, 1, lane) | row_from(t[j + 1], -2, lane) |
               row_from(t[j + 1], 2, lane);
        d[j] |= ones_outside(k - 1 + j);
    }
    // rows past the wave count as outside for the lanes that look at them; those lanes' results are not stored
    const u64 erow = shrink5(d[0], d[1], d[2]);
    auto and_from = [&](u64 v, int dd) { const u64 tv = __shfl(v, lane + dd); return (unsigned)(lane + dd) < 64u ? tv : ~0ull; };
    u64 f = and_from(erow, -1) & erow & and_from(erow, 1) & and_from(d[1], -2) & and_from(d[1], 2);
    f &= ~ones_outside(k);
    if (row_in && lane >= 4 && lane < 4 + MC_ROWS) out[(size_t)(y + oframe) * opitch + k + oframe] = f;
}

__global__ __launch_bounds__(256) void expand_bits_kernel(const u64* __restrict__ F, int wpr, int w, int h,
                                                          uint8_t* __restrict__ mask, size_t mstride) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x < w && y < h) mask[(size_t)y * mstride + x] = (F[(size_t)y * wpr + (x >> 6)] >> (x & 63)) & 1 ? 255 : 0;
}

// d_T: scratch of ceil(w/64)*h words.  The closed mask goes to d_out with row pitch opitch words, shifted by
// oframe rows and words (1 for a BitFrame whose frame is already zero, 0 for a plain bit plane).
// srcs (optional, d_src == nullptr): `frames` <= SRC_LIST_MAX source pointers on the HOST, one geometry and pitch; the pictures'
// bit planes / results lie tfb / ofb words apart.
int launch_content_bits(const uint8_t* d_src, size_t stride, int w, int h, int cn, u64* d_T, u64* d_out, int opitch,
                        int oframe, hipStream_t st, const uint8_t* const* srcs = nullptr, int frames = 1, size_t tfb = 0, size_t ofb = 0,
                        int srcs_aligned = 0) {
    if ((!d_src && !srcs) || (!d_src && frames > SRC_LIST_MAX) || !d_T || !d_out || w <= 0 || h <= 0 || (cn != 1 && cn != 3) || frames < 1) { set_last_error("content_mask: invalid argument"); return VS_ERR_INVALID_ARG; }
    if (h > 65535) { set_last_error("content_mask: image too tall"); return VS_ERR_INVALID_ARG; }
    const int wpr = (w + 63) / 64;
    // (a table of sources: srcs_aligned = every one of them is 4-byte aligned)
    const int aligned = !d_src ? (srcs_aligned && (stride & 3) == 0) : ((((uintptr_t)d_src | stride) & 3) == 0);
    SrcList sl;
    for (int i = 0; i < SRC_LIST_MAX; i++) sl.p[i] = !d_src ? srcs[i < frames ? i : 0] : nullptr;
    dim3 g1((w + 1023) / 1024, h, frames);
    bool wide = !d_src && cn == 1 && (w & 15) == 0 && (stride & 15) == 0;
    for (int i = 0; wide && i < frames; i++) wide = ((uintptr_t)srcs[i] & 15) == 0;
    if (wide) hipLaunchKernelGGL(threshold_bits_gray16_kernel, dim3((w + 1023) / 1024, (h + 3) / 4, frames), dim3(256), 0, st, stride, w, h, d_T, wpr, sl, tfb);
    else if (cn == 3) hipLaunchKernelGGL(threshold_bits_kernel<3>, g1, dim3(256), 0, st, d_src, stride, w, aligned, d_T, wpr, sl, tfb);
    else hipLaunchKernelGGL(threshold_bits_kernel<1>, g1, dim3(256), 0, st, d_src, stride, w, aligned, d_T, wpr, sl, tfb);
    dim3 g2(wpr, (h + MC_ROWS - 1) / MC_ROWS, frames);
    hipLaunchKernelGGL(close5_bits_kernel, g2, dim3(64), 0, st, d_T, wpr, w, h, d_out, opitch, oframe, tfb, ofb);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

}  // namespace
}  // namespace vsd

using namespace vsd;

struct vs_azc {
    int device = 0;
    hipStream_t st = nullptr;
    std::string err;
    uint8_t* d_mask = nullptr;
    uint8_t* h_mask = nullptr;        // pinned; both hold the mask as a BitFrame
    u64* d_tbits = nullptr;           // thresholded bit plane, before the closing
    uint8_t* d_in = nullptr;
    uint8_t* d_out = nullptr;
    int mask_w = 0, mask_h = 0;       // the size d_mask / h_mask (a BitFrame) are laid out for
    size_t io_bytes = 0;
    CropScratch scratch;
    int32_t info[8] = {0};
    // ---- asynchronous NV12 path (vs_azc_apply_nv12_dev) ----
    // Eight consecutive frames form a batch: their mask kernels are ONE launch each (blockIdx.z = frame) and their bit masks come
    // to page-locked memory with one copy (a launch costs the host 6 - 7 us on this runtime: per frame they were half of the
    // chain's host time).  The contour logic - host work in the reference as well (AutoZoomCrop.cpp:141-147 downloads the mask for
    // cv::findContours) - runs on worker threads, a frame each (frames do not depend on each other; 0.3 ms per 4K frame and
    // thread); the worker that finishes a batch's last contour queues the crop-and-scale of the batch's surfaces, both planes, as
    // one launch.  NBS batches in flight.
    static constexpr int ZB = 8, NBS = 4, NRES = 1024;      // (ZB <= SRC_LIST_MAX, 2 ZB <= WARP_JOBS_MAX)
    int nw = 12;                         // worker threads (VS_AZC_WORKERS, 1 .. 16): 31 k frames/s alone with eight, 41 k with twelve
    struct Frame { const uint8_t* src; uint8_t* dst; int w, h; size_t pitch, uv, opitch, ouv; long ticket; };
    struct BatchSlot {
        uint8_t* d_masks = nullptr;      // ZB BitFrames
        uint8_t* h_masks = nullptr;      // page-locked
        u64* d_tbits = nullptr;          // ZB thresholded bit planes, before the closing
        hipStream_t st = nullptr;        // the batch's mask kernels and mask copy: a stream per slot, so that the next batch's kernels
                                         // do not queue behind this batch's 8 MB on the way to the host
        hipEvent_t ev = nullptr;
        int mw = 0, mh = 0;
        int n = 0, left = 0;             // frames of the batch / n until the batch's crop-and-scale has been queued, then 0 (slot free)
        int todo = 0, nwj = 0;           // host parts not yet through / crop-and-scale jobs collected (wj)
        WarpJob wj[2 * ZB];
        int arrived = 0;                 // the masks: 0 on their way, 2 there, 3 the wait for them failed
        std::chrono::steady_clock::time_point t_issue, t_arrive;
        Frame fr[ZB];
    } bslot[NBS];
    struct Result { long ticket = -1; int out_w = 0, out_h = 0, rc = 0; int32_t info[8] = {0}; } results[NRES];
    std::vector<Frame> pending;          // handed over, not yet a batch
    long nbatches = 0;
    hipStream_t st_out = nullptr;
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::deque<std::pair<int, int>> jobs;    // (batch slot, frame of the batch): the masks are there
    std::deque<int> on_the_way;              // batch slots whose masks are on their way, oldest first
    bool waiting = false;                    // a worker waits for the oldest of them
    bool quit = false;
    long issued = 0, completed = 0;
    int async_rc = 0;                    // first failure of a worker's launch (reported by vs_azc_sync)
    // where the workers' time goes (vs_azc_worker_times): frames, and seconds waiting for a job / for a batch's masks / in the
    // contour logic / queueing launches and publishing
    double wt[5] = {0, 0, 0, 0, 0};
    double bt[4] = {0, 0, 0, 0};         // batches; seconds from a batch's launches to its masks' arrival / from there to its crop launch /
                                         // the caller waited for a free batch slot
};

static_assert(vs_azc::ZB <= SRC_LIST_MAX && 2 * vs_azc::ZB <= WARP_JOBS_MAX, "a batch's sources and warp jobs travel as kernel arguments");

#define A_HIP(a, expr)                                                             \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) { (a)->err = std::string(#expr) + ": " + hipGetErrorString(_e); set_last_error((a)->err); return VS_ERR_HIP; } \
    } while (0)
#define A_TRY(a, expr)                                                             \
    do { int _s = (expr); if (_s != VS_OK) { (a)->err = get_last_error(); return _s; } } while (0)

extern "C" {

// The mask as 0 / 255 bytes (the form cv::threshold / cv::morphologyEx hand on); scratch for the two bit planes is
// kept between calls.
int vs_op_content_mask(const void* d_src, size_t stride, int w, int h, int cn, void* d_mask, size_t mask_stride,
                       void* stream) {
    VS_TRY(ensure_device());
    if (!d_mask || w <= 0 || h <= 0 || mask_stride < (size_t)w) { set_last_error("content_mask: invalid argument"); return VS_ERR_INVALID_ARG; }
    static std::mutex mu;
    static u64* planes = nullptr;
    static size_t plane_words = 0;
    std::lock_guard<std::mutex> lock(mu);
    const int wpr = (w + 63) / 64;
    const size_t words = (size_t)wpr * h;
    hipStream_t st = (hipStream_t)stream;
    if (plane_words < words) {
        if (planes) { VS_HIP_TRY(hipDeviceSynchronize()); (void)hipFree(planes); planes = nullptr; plane_words = 0; }
        VS_HIP_TRY(hipMalloc((void**)&planes, 2 * words * 8));
        plane_words = words;
    }
    VS_TRY(launch_content_bits((const uint8_t*)d_src, stride, w, h, cn, planes, planes + plane_words, wpr, 0, st));
    dim3 grid((w + 255) / 256, h);
    hipLaunchKernelGGL(expand_bits_kernel, grid, dim3(256), 0, st, planes + plane_words, wpr, w, h, (uint8_t*)d_mask, mask_stride);
    VS_HIP_TRY(hipGetLastError());
    return VS_OK;
}

int vs_azc_create(int device, vs_azc** out) {
    if (!out) return VS_ERR_INVALID_ARG;
    *out = nullptr;
    VS_TRY(ensure_device());
    VS_HIP_TRY(hipSetDevice(device));
    vs_azc* a = new (std::nothrow) vs_azc();
    if (!a) return VS_ERR_HIP;
    a->device = device;
    hipError_t e = hipStreamCreateWithFlags(&a->st, hipStreamNonBlocking);
    if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); delete a; return VS_ERR_HIP; }
    *out = a;
    return VS_OK;
}

void vs_azc_destroy(vs_azc* a) {
    if (!a) return;
    (void)hipSetDevice(a->device);
    if (!a->workers.empty()) {
        { std::lock_guard<std::mutex> g(a->mu); a->quit = true; }
        a->cv_job.notify_all();
        for (auto& t : a->workers) t.join();
    }
    if (a->st_out) { (void)hipStreamSynchronize(a->st_out); (void)hipStreamDestroy(a->st_out); }
    for (auto& q : a->bslot) {
        if (q.st) { (void)hipStreamSynchronize(q.st); (void)hipStreamDestroy(q.st); }
        if (q.d_masks) (void)hipFree(q.d_masks);
        if (q.h_masks) (void)hipHostFree(q.h_masks);
        if (q.d_tbits) (void)hipFree(q.d_tbits);
        if (q.ev) (void)hipEventDestroy(q.ev);
    }
    if (a->st) (void)hipStreamSynchronize(a->st);
    if (a->d_mask) (void)hipFree(a->d_mask);
    if (a->h_mask) (void)hipHostFree(a->h_mask);
    if (a->d_tbits) (void)hipFree(a->d_tbits);
    if (a->d_in) (void)hipFree(a->d_in);
    if (a->d_out) (void)hipFree(a->d_out);
    if (a->st) (void)hipStreamDestroy(a->st);
    delete a;
}

const char* vs_azc_last_error(const vs_azc* a) { return a ? a->err.c_str() : ""; }

int vs_azc_get_info(const vs_azc* a, int32_t* info8) {
    if (!a || !info8) return VS_ERR_INVALID_ARG;
    memcpy(info8, a->info, sizeof a->info);
    return VS_OK;
}

static int azc_flush_pending(vs_azc* a, std::unique_lock<std::mutex>& lk);

int vs_azc_sync(vs_azc* a) {
    if (!a) return VS_ERR_INVALID_ARG;
    A_HIP(a, hipSetDevice(a->device));
    if (!a->workers.empty()) {          // asynchronous NV12 frames: their host parts first, then what the workers queued
        std::unique_lock<std::mutex> lk(a->mu);
        const int frc = azc_flush_pending(a, lk);
        if (frc != VS_OK) return frc;
        a->cv_done.wait(lk, [&] { return a->completed == a->issued; });
        const int arc = a->async_rc;
        a->async_rc = VS_OK;
        lk.unlock();
        A_HIP(a, hipStreamSynchronize(a->st_out));
        if (arc != VS_OK) { a->err = "auto zoom/crop: a worker's launch failed"; set_last_error(a->err); return arc; }
    }
    A_HIP(a, hipStreamSynchronize(a->st));
    return VS_OK;
}

// Mask on the device, contour logic on the host: fills a->info (:111-228).
static int azc_plan(vs_azc* a, const void* d_data, int w, int h, size_t stride, int cn) {
    if (w > 65535 || h > 32767) { a->err = "auto zoom/crop: image too large"; set_last_error(a->err); return VS_ERR_INVALID_ARG; }
    const size_t mb = BitFrame::words_for(w, h) * 8;
    if (a->mask_w != w || a->mask_h != h) {
        if (a->d_mask) (void)hipFree(a->d_mask);
        if (a->h_mask) (void)hipHostFree(a->h_mask);
        if (a->d_tbits) (void)hipFree(a->d_tbits);
        a->d_mask = a->h_mask = nullptr; a->d_tbits = nullptr; a->mask_w = a->mask_h = 0;
        A_HIP(a, hipMalloc((void**)&a->d_mask, mb));
        A_HIP(a, hipMalloc((void**)&a->d_tbits, (size_t)((w + 63) / 64) * h * 8));
        A_HIP(a, hipHostMalloc((void**)&a->h_mask, mb, hipHostMallocDefault));
        A_HIP(a, hipMemsetAsync(a->d_mask, 0, mb, a->st));          // the frame; the kernel rewrites the inside
        a->mask_w = w; a->mask_h = h;
    }
    BitFrame bf;
    bf.w = w; bf.h = h; bf.pitch = BitFrame::pitch_for(w); bf.F = (const uint64_t*)a->h_mask;
    A_TRY(a, launch_content_bits((const uint8_t*)d_data, stride, w, h, cn, a->d_tbits, (u64*)a->d_mask, bf.pitch, 1, a->st));   // :111-139
    A_HIP(a, hipMemcpyAsync(a->h_mask, a->d_mask, mb, hipMemcpyDeviceToHost, a->st));                 // :142-143
    A_HIP(a, hipStreamSynchronize(a->st));
    crop_from_mask(bf, a->scratch, a->info, nullptr);                                                 // :146-228
    return VS_OK;
}

// Crop + scale (:246-270) or the unchanged frame (:149-152, :238-249), left in flight on a->st.
static int azc_emit(vs_azc* a, const void* d_data, int w, int h, size_t stride, int cn, void* d_out, size_t out_stride) {
    if (!a->info[7]) {
        if (out_stride < (size_t)w * cn) return VS_ERR_INVALID_ARG;
        A_HIP(a, hipMemcpy2DAsync(d_out, out_stride, d_data, stride, (size_t)w * cn, h, hipMemcpyDeviceToDevice, a->st));
        return VS_OK;
    }
    if (out_stride < (size_t)640 * cn) return VS_ERR_INVALID_ARG;
    const int cx = a->info[2], cy = a->info[3], cw = a->info[4], ch = a->info[5];
    // M = [sx 0 0; 0 sy 0] held as CV_32F (:261-262); cv::warpAffine inverts it in double
    const float Mf[6] = {(float)(640.0 / cw), 0.f, 0.f, 0.f, (float)(360.0 / ch), 0.f};
    double Mi[6];
    warp_invert(Mf, Mi);
    const uint8_t* roi = (const uint8_t*)d_data + (size_t)cy * stride + (size_t)cx * cn;
    A_TRY(a, launch_warp_affine_inv(roi, stride, cw, ch, (uint8_t*)d_out, out_stride, 640, 360, cn, Mi, VS_BORDER_BLACK, a->st));
    return VS_OK;
}

// Frame in HBM -> 640x360 crop in HBM (or an unchanged copy on the fallback paths).  out_stride must fit
// either outcome (>= max(w, 640) * cn).  The result is left in flight on the object's stream (vs_azc_sync).
int vs_azc_apply_dev(vs_azc* a, const void* d_data, int w, int h, size_t stride, int cn, void* d_out, size_t out_stride,
                     int* out_w, int* out_h) {
    if (!a || !d_data || !d_out || !out_w || !out_h || w <= 0 || h <= 0 || (cn != 1 && cn != 3) || stride < (size_t)w * cn ||
        out_stride < (size_t)std::max(w, 640) * cn)
        return VS_ERR_INVALID_ARG;
    A_HIP(a, hipSetDevice(a->device));
    int rc = azc_plan(a, d_data, w, h, stride, cn);
    if (rc != VS_OK) return rc;
    rc = azc_emit(a, d_data, w, h, stride, cn, d_out, out_stride);
    if (rc != VS_OK) return rc;
    *out_w = a->info[7] ? 640 : w;
    *out_h = a->info[7] ? 360 : h;
    return VS_OK;
}

// cv::Mat autoZoomCrop(const cv::Mat&, double) on host buffers.  `out` must hold max(w*h, 640*360)*cn bytes and
// receives packed rows of *out_w * cn bytes.
int vs_azc_apply(vs_azc* a, const uint8_t* data, int w, int h, size_t stride, int cn, uint8_t* out, int* out_w, int* out_h) {
    if (!a || !data || !out || !out_w || !out_h || w <= 0 || h <= 0 || (cn != 1 && cn != 3) || stride < (size_t)w * cn)
        return VS_ERR_INVALID_ARG;
    A_HIP(a, hipSetDevice(a->device));
    const size_t row = (size_t)w * cn, bytes = std::max(row * h, (size_t)640 * 360 * cn);
    if (a->io_bytes < bytes) {
        if (a->d_in) (void)hipFree(a->d_in);
        if (a->d_out) (void)hipFree(a->d_out);
        a->d_in = a->d_out = nullptr; a->io_bytes = 0;
        A_HIP(a, hipMalloc((void**)&a->d_in, bytes));
        A_HIP(a, hipMalloc((void**)&a->d_out, bytes));
        a->io_bytes = bytes;
    }
    A_HIP(a, hipMemcpy2DAsync(a->d_in, row, data, stride, row, h, hipMemcpyHostToDevice, a->st));
    int rc = azc_plan(a, a->d_in, w, h, row, cn);
    if (rc != VS_OK) return rc;
    const int ow = a->info[7] ? 640 : w, oh = a->info[7] ? 360 : h;
    rc = azc_emit(a, a->d_in, w, h, row, cn, a->d_out, (size_t)ow * cn);
    if (rc != VS_OK) return rc;
    A_HIP(a, hipMemcpyAsync(out, a->d_out, (size_t)ow * cn * oh, hipMemcpyDeviceToHost, a->st));
    A_HIP(a, hipStreamSynchronize(a->st));
    *out_w = ow; *out_h = oh;
    return VS_OK;
}

// ---- NV12, asynchronous ----------------------------------------------------------------------------------------------------
// One frame's host part, on a worker thread: the contour logic on the frame's mask once its batch's masks have arrived, then its
// crop-and-scale job (or, at once, the copy of the fall-back paths) for both planes on st_out; the worker that finishes a batch's
// last frame queues the batch's jobs as one launch.  The content mask is taken from the luma plane (gray > 1, :121-127 on a picture
// that is gray already); the crop rectangle applies to the luma plane as it is and, halved, to the half-size chroma plane;
// each plane is scaled to its share of 640 x 360 by the reference's scale matrix (:261-270).
static void azc_worker(vs_azc* a) {
    (void)hipSetDevice(a->device);
    CropScratch scratch;
    typedef std::chrono::steady_clock clk;
    auto secs = [](clk::time_point p, clk::time_point q) { return std::chrono::duration<double>(q - p).count(); };
    for (;;) {
        std::pair<int, int> job;
        const clk::time_point t_idle = clk::now();
        clk::time_point t_job = t_idle;
        {
            // A frame becomes a job when its batch's masks have arrived.  The batches on their way are waited for in order by ONE
            // worker at a time - whichever has nothing else to do (a blocking wait: no core spent on it) - so that the others
            // stay free for the frames of the batches that are there.  (With a wait per frame the workers stood in front of the
            // copy engine 250 - 330 us per frame; that time is idle time now - beside a stabilizer the stage is bound by the latency
            // of its device part, scratch/README.md.)
            std::unique_lock<std::mutex> lk(a->mu);
            for (;;) {
                if (!a->jobs.empty()) break;
                if (!a->on_the_way.empty() && !a->waiting) {
                    const int slot = a->on_the_way.front();
                    a->on_the_way.pop_front();
                    a->waiting = true;
                    lk.unlock();
                    const clk::time_point t0 = clk::now();
                    const hipError_t e = hipEventSynchronize(a->bslot[slot].ev);
                    const clk::time_point t1 = clk::now();
                    lk.lock();
                    a->waiting = false;
                    a->wt[2] += secs(t0, t1);
                    t_job = t_job + (t1 - t0);                      // (not idle time)
                    a->bslot[slot].arrived = e == hipSuccess ? 2 : 3;
                    a->bslot[slot].t_arrive = t1;
                    for (int i = 0; i < a->bslot[slot].n; i++) a->jobs.emplace_back(slot, i);
                    a->cv_job.notify_all();
                    continue;
                }
                if (a->quit) return;
                a->cv_job.wait(lk);
            }
            job = a->jobs.front();
            a->jobs.pop_front();
        }
        vs_azc::BatchSlot& b = a->bslot[job.first];
        const vs_azc::Frame q = b.fr[job.second];
        vs_azc::Result res;
        res.ticket = q.ticket;
        int rc = b.arrived == 3 ? VS_ERR_HIP : VS_OK;
        WarpJob wj[2];
        bool scaled = false;
        const clk::time_point t_masks = clk::now();
        clk::time_point t_contour = t_masks;
        t_job = t_job > t_masks ? t_masks : t_job;
        if (rc == VS_OK) {
            BitFrame bf;
            bf.w = q.w; bf.h = q.h; bf.pitch = BitFrame::pitch_for(q.w);
            bf.F = (const uint64_t*)(b.h_masks + (size_t)job.second * BitFrame::words_for(q.w, q.h) * 8);
            crop_from_mask(bf, scratch, res.info, nullptr);                                                // :146-228
            t_contour = clk::now();
            if (!res.info[7]) {                                                                            // :149-152, :238-249
                res.out_w = q.w; res.out_h = q.h;
                if (hipMemcpy2DAsync(q.dst, q.opitch, q.src, q.pitch, (size_t)q.w, q.h, hipMemcpyDeviceToDevice, a->st_out) != hipSuccess ||
                    hipMemcpy2DAsync(q.dst + q.ouv, q.opitch, q.src + q.uv, q.pitch, (size_t)q.w, q.h / 2, hipMemcpyDeviceToDevice, a->st_out) != hipSuccess)
                    rc = VS_ERR_HIP;
            } else {
                res.out_w = 640; res.out_h = 360;
                const int cx = res.info[2], cy = res.info[3], cw = res.info[4], ch = res.info[5];
                const int ux = cx / 2, uy = cy / 2, uw = std::max(1, cw / 2), uh = std::max(1, ch / 2);
                const float My[6] = {(float)(640.0 / cw), 0.f, 0.f, 0.f, (float)(360.0 / ch), 0.f};
                const float Mu[6] = {(float)(320.0 / uw), 0.f, 0.f, 0.f, (float)(180.0 / uh), 0.f};
                warp_invert(My, wj[0].m);
                warp_invert(Mu, wj[1].m);
                wj[0].src = q.src + (size_t)cy * q.pitch + cx; wj[0].dst = q.dst;
                wj[0].sw = cw; wj[0].sh = ch; wj[0].dw = 640; wj[0].dh = 360; wj[0].cn = 1;
                wj[1].src = q.src + q.uv + (size_t)uy * q.pitch + (size_t)ux * 2; wj[1].dst = q.dst + q.ouv;
                wj[1].sw = uw; wj[1].sh = uh; wj[1].dw = 320; wj[1].dh = 180; wj[1].cn = 2;
                for (WarpJob& w : wj) { w.sstride = (uint32_t)q.pitch; w.dstride = (uint32_t)q.opitch; w.border = VS_BORDER_BLACK; }
                scaled = true;
            }
        }
        res.rc = rc;
        // The crop-and-scale of the batch's frames is ONE launch (launch_warp_jobs), queued by the worker that finishes the batch's
        // last contour; a frame counts as complete (vs_azc_sync) when that launch has been queued.
        bool last;
        int njobs = 0;
        WarpJob all[2 * vs_azc::ZB];
        {
            std::lock_guard<std::mutex> g(a->mu);
            a->results[res.ticket % vs_azc::NRES] = res;
            if (scaled) { b.wj[b.nwj++] = wj[0]; b.wj[b.nwj++] = wj[1]; }
            last = --b.todo == 0;
            if (last) { njobs = b.nwj; memcpy(all, b.wj, sizeof(WarpJob) * njobs); }
        }
        a->cv_done.notify_all();          // (vs_azc_result waits for the host part only)
        int lrc = last && njobs ? launch_warp_jobs(all, njobs, a->st_out) : VS_OK;
        {
            std::lock_guard<std::mutex> g(a->mu);
            const clk::time_point t_end = clk::now();
            a->wt[0] += 1; a->wt[1] += secs(t_idle, t_masks) - secs(t_idle, t_job); a->wt[3] += secs(t_masks, t_contour);
            a->wt[4] += secs(t_contour, t_end);
            if (!last) continue;
            if (lrc != VS_OK) {
                a->async_rc = lrc;
                for (int i = 0; i < b.n; i++) a->results[b.fr[i].ticket % vs_azc::NRES].rc = lrc;
            }
            a->completed += b.n;
            b.left = 0;
            a->bt[0] += 1; a->bt[1] += secs(b.t_issue, b.t_arrive); a->bt[2] += secs(b.t_arrive, t_end);
        }
        a->cv_done.notify_all();
    }
}

// (a->mu held through lk) what has been handed over becomes a batch: mask kernels, one copy, a job per frame
static int azc_flush_pending(vs_azc* a, std::unique_lock<std::mutex>& lk) {
    if (a->pending.empty()) return VS_OK;
    vs_azc::BatchSlot& b = a->bslot[a->nbatches % vs_azc::NBS];
    {
        const auto t0 = std::chrono::steady_clock::now();
        a->cv_done.wait(lk, [&] { return b.left == 0; });
        a->bt[3] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    const int n = (int)a->pending.size();
    const int w = a->pending[0].w, h = a->pending[0].h;
    const size_t mb = BitFrame::words_for(w, h) * 8, tw = (size_t)((w + 63) / 64) * h;
    if (!b.ev) {
        A_HIP(a, hipEventCreateWithFlags(&b.ev, hipEventDisableTiming | hipEventBlockingSync));
        A_HIP(a, hipStreamCreateWithFlags(&b.st, hipStreamNonBlocking));
    }
    if (b.mw != w || b.mh != h) {
        if (b.d_masks) (void)hipFree(b.d_masks);
        if (b.h_masks) (void)hipHostFree(b.h_masks);
        if (b.d_tbits) (void)hipFree(b.d_tbits);
        b.d_masks = b.h_masks = nullptr; b.d_tbits = nullptr; b.mw = b.mh = 0;
        A_HIP(a, hipMalloc((void**)&b.d_masks, mb * vs_azc::ZB));
        A_HIP(a, hipMalloc((void**)&b.d_tbits, tw * 8 * vs_azc::ZB));
        A_HIP(a, hipHostMalloc((void**)&b.h_masks, mb * vs_azc::ZB, hipHostMallocDefault));
        A_HIP(a, hipMemsetAsync(b.d_masks, 0, mb * vs_azc::ZB, b.st));          // the frames of the BitFrames; the kernel rewrites the insides
        b.mw = w; b.mh = h;
    }
    int aligned = 1;
    const uint8_t* srcs[vs_azc::ZB];
    for (int i = 0; i < n; i++) {
        b.fr[i] = a->pending[i];
        srcs[i] = a->pending[i].src;
        if ((uintptr_t)a->pending[i].src & 3) aligned = 0;
    }
    A_TRY(a, launch_content_bits(nullptr, a->pending[0].pitch, w, h, 1, b.d_tbits, (u64*)b.d_masks, BitFrame::pitch_for(w), 1, b.st, srcs, n, tw,
                                 mb / 8, aligned));                                                                              // :111-139 on the luma planes
    A_HIP(a, hipMemcpyAsync(b.h_masks, b.d_masks, mb * n, hipMemcpyDeviceToHost, b.st));                                          // :142-143
    A_HIP(a, hipEventRecord(b.ev, b.st));
    b.t_issue = std::chrono::steady_clock::now();
    b.n = n; b.left = n; b.todo = n; b.nwj = 0; b.arrived = 0;
    a->on_the_way.push_back((int)(a->nbatches % vs_azc::NBS));
    a->nbatches++;
    a->pending.clear();
    a->cv_job.notify_all();
    return VS_OK;
}

// autoZoomCrop for an NV12 surface in HBM, ASYNCHRONOUS.  d_out receives the result - 640 x 360 (luma rows of out_pitch bytes,
// the 320 x 180 interleaved chroma plane out_uv_offset bytes behind) or, on the reference's fall-back paths, the unchanged
// w x h surface - so out_pitch >= max(w, 640) and out_uv_offset >= max(h, 360) * out_pitch.  The call returns at once with a
// ticket; eight consecutive frames of one geometry form a batch (vs_azc_sync and vs_azc_result close an incomplete one).
// vs_azc_result(ticket) tells what came out (it waits for that frame's host part), the pixels are complete after vs_azc_sync.
// Surface and result buffer must stay untouched until then; results of the last 1024 tickets are kept.
int vs_azc_apply_nv12_dev(vs_azc* a, const void* d_surface, int w, int h, size_t pitch, size_t uv_offset, void* d_out, size_t out_pitch,
                          size_t out_uv_offset, int64_t* ticket) {
    if (!a || !d_surface || !d_out || w < 2 || h < 2 || (w & 1) || (h & 1) || pitch < (size_t)w || out_pitch < (size_t)std::max(w, 640) ||
        out_uv_offset < (size_t)std::max(h, 360) * out_pitch)
        return VS_ERR_INVALID_ARG;
    if (w > 65535 || h > 32767) { a->err = "auto zoom/crop: image too large"; set_last_error(a->err); return VS_ERR_INVALID_ARG; }
    if (uv_offset == 0) uv_offset = (size_t)h * pitch;
    A_HIP(a, hipSetDevice(a->device));
    if (a->workers.empty()) {
        A_HIP(a, hipStreamCreateWithFlags(&a->st_out, hipStreamNonBlocking));
        try {
            if (const char* e = std::getenv("VS_AZC_WORKERS")) a->nw = std::max(1, std::min(std::atoi(e), 16));
            for (int i = 0; i < a->nw; i++) a->workers.emplace_back(azc_worker, a);
        } catch (...) {
            a->err = "auto zoom/crop: cannot start worker threads"; set_last_error(a->err);
            return VS_ERR_HIP;
        }
    }
    std::unique_lock<std::mutex> lk(a->mu);
    if (!a->pending.empty() && (a->pending[0].w != w || a->pending[0].h != h || a->pending[0].pitch != pitch)) {
        const int rc = azc_flush_pending(a, lk);
        if (rc != VS_OK) return rc;
    }
    a->pending.push_back(vs_azc::Frame{(const uint8_t*)d_surface, (uint8_t*)d_out, w, h, pitch, uv_offset, out_pitch, out_uv_offset, a->issued});
    if (ticket) *ticket = a->issued;
    a->issued++;
    if ((int)a->pending.size() >= vs_azc::ZB) return azc_flush_pending(a, lk);
    return VS_OK;
}

// n surfaces of one layout in call order; tickets (optional) receives n tickets
int vs_azc_apply_nv12_dev_n(vs_azc* a, const void* const* d_surfaces, void* const* d_outs, int n, int w, int h, size_t pitch, size_t uv_offset,
                            size_t out_pitch, size_t out_uv_offset, int64_t* tickets) {
    if (!a || !d_surfaces || !d_outs || n < 0) return VS_ERR_INVALID_ARG;
    for (int i = 0; i < n; i++) {
        const int rc = vs_azc_apply_nv12_dev(a, d_surfaces[i], w, h, pitch, uv_offset, d_outs[i], out_pitch, out_uv_offset, tickets ? tickets + i : nullptr);
        if (rc != VS_OK) return rc;
    }
    return VS_OK;
}

// Diagnostics of the asynchronous path: frames through the workers so far and the seconds the workers spent, summed over the
// threads, without a frame / waiting for the masks on their way / in the contour logic / queueing launches and publishing; then
// batches and, summed over them, the seconds from launches to masks, from masks to the crop launch, and the caller's waits for a slot.
int vs_azc_worker_times(vs_azc* a, double* out5) {      // (nine values: include/vs_stab.h)
    if (!a || !out5) return VS_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(a->mu);
    for (int i = 0; i < 5; i++) out5[i] = a->wt[i];
    for (int i = 0; i < 4; i++) out5[5 + i] = a->bt[i];
    return VS_OK;
}

// What ticket's frame became: waits for its host part (not for the pixels: vs_azc_sync).
int vs_azc_result(vs_azc* a, int64_t ticket, int* out_w, int* out_h, int32_t* info8) {
    if (!a || ticket < 0) return VS_ERR_INVALID_ARG;
    std::unique_lock<std::mutex> lk(a->mu);
    if (ticket >= a->issued || ticket + vs_azc::NRES <= a->issued) { a->err = "auto zoom/crop: no such ticket (results of the last 1024 frames are kept)"; set_last_error(a->err); return VS_ERR_INVALID_ARG; }
    if (!a->pending.empty() && ticket >= a->pending[0].ticket) {       // (still waiting for its batch to fill)
        const int frc = azc_flush_pending(a, lk);
        if (frc != VS_OK) return frc;
    }
    a->cv_done.wait(lk, [&] { return a->results[ticket % vs_azc::NRES].ticket == ticket; });
    const vs_azc::Result& r = a->results[ticket % vs_azc::NRES];
    if (out_w) *out_w = r.out_w;
    if (out_h) *out_h = r.out_h;
    if (info8) memcpy(info8, r.info, sizeof r.info);
    if (r.rc != VS_OK) { a->err = "auto zoom/crop: a worker's launch failed"; set_last_error(a->err); }
    return r.rc;
}

}  // extern "C"
