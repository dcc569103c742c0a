// Coordinate tables of the batched warp kernels (k_warp.hip), shared with the kernel that computes the inverse maps
// (k_ransac.hip: the release workgroup of a frame builds the frame's tables as soon as its map exists).
//
// cv::warpAffine's WarpAffineInvoker evaluates, in double with one rounding each,
//   adelta[x] = round(M0 * x * 1024), bdelta[x] = round(M3 * x * 1024)           per column
//   X0[y] = round((M1 * y + M2) * 1024) + 16, Y0[y] = round((M4 * y + M5) * 1024) + 16   per row (16 = round_delta, AB_BITS 10)
// (/root/reference/src/Stabilizer.cpp:1056-1060 calls it).  ~6000 terms per 1080p frame, built once per frame instead of once
// per 128 x 16 tile; the warp kernels then hold no double-precision arithmetic at all.
//
// Layout of one plane's table (ints), TabLayout:
//   [0, 8 gx)                  per tile column (128 pixels): source and destination pointer of the plane, adelta and bdelta of
//                              its first and last column                            (src, dst, ad0, ad1, bd0, bd1)
//   [row, + 4 gy)              per tile row (16 rows): X0, Y0 of its first and last row       (Xa, Xb, Ya, Yb)
//   [ad, + dw) adelta(x); [+dw) bdelta(x); [+dh) X0(y); [+dh) Y0(y)
// so that everything a tile needs before its staging loads comes with ONE 32-byte and ONE 16-byte scalar load.
// An NV12 surface has two planes: the table of its interleaved chroma plane (dw/2 x dh/2, two channels, the map with the
// halved translation) follows the luma table in the same block, nv12_tab_ints() apart from frame to frame.
#ifndef VS_WARP_TAB_H
#define VS_WARP_TAB_H

#include <stdint.h>

namespace vsd {

constexpr int WT_TW = 128;     // output tile width  (pixels) of the warp kernels
constexpr int WT_TH = 16;      // rows per row record (the plane kernels' taller tiles take the records of their first and last 16 rows)
constexpr int WT_COL = 8;      // ints per tile column record

struct TabLayout { int row, ad, stride; };

#ifdef __HIPCC__
__host__ __device__
#endif
inline TabLayout tab_layout(int dw, int dh) {
    const int gx = (dw + WT_TW - 1) / WT_TW, gy = (dh + WT_TH - 1) / WT_TH;
    TabLayout t;
    t.row = WT_COL * gx;
    t.ad = t.row + 4 * gy;
    t.stride = (t.ad + 2 * dw + 2 * dh + 3) & ~3;
    return t;
}

// Ints per frame of an NV12 surface of w x h luma pixels: luma table, then chroma table.
inline int nv12_tab_ints(int w, int h) { return tab_layout(w, h).stride + tab_layout(w / 2, h / 2).stride; }

// One plane of one due frame: where its table goes and what the table's pointer records name.  tabs == nullptr: no table.
struct WarpTabJob {
    int32_t* tabs;
    const uint8_t* src;
    uint8_t* dst;
    int32_t dw, dh;
};

#ifdef __HIPCC__
// hal::warpAffine / WarpAffineInvoker coordinate terms in one form: round((p*v + q) * 1024)
// (columns: adelta = round(M0*x*1024) with q = 0; rows: round((M1*y + M2)*1024) + round_delta).
__device__ __forceinline__ int wt_coord_term(double p, double q, double v) { return __double2int_rn((p * v + q) * 1024); }

// Entry j of a plane's table (j in [0, wt_entries(dw, dh))), from the plane's inverse map m[6]: any thread may take any entry.
__device__ __forceinline__ int wt_entries(int dw, int dh) { return dw + dh + (dw + WT_TW - 1) / WT_TW + (dh + WT_TH - 1) / WT_TH; }

__device__ __forceinline__ void wt_build_entry(int32_t* __restrict__ T, const TabLayout& L, const double* m, int dw, int dh,
                                               const uint8_t* src, uint8_t* dst, int j) {
    const int gx = (dw + WT_TW - 1) / WT_TW, gy = (dh + WT_TH - 1) / WT_TH;
    if (j < dw) {
        const double dv = (double)j;
        T[L.ad + j] = wt_coord_term(m[0], 0.0, dv);
        T[L.ad + dw + j] = wt_coord_term(m[3], 0.0, dv);
        return;
    }
    j -= dw;
    if (j < dh) {
        const double dv = (double)j;
        T[L.ad + 2 * dw + j] = wt_coord_term(m[1], m[2], dv) + 16;
        T[L.ad + 2 * dw + dh + j] = wt_coord_term(m[4], m[5], dv) + 16;
        return;
    }
    j -= dh;
    if (j < gx) {
        const int x1 = j * WT_TW + WT_TW < dw ? j * WT_TW + WT_TW : dw;
        const double v0 = (double)(j * WT_TW), v1 = (double)(x1 - 1);
        int32_t* C = T + WT_COL * j;
        *reinterpret_cast<const uint8_t**>(C) = src;
        *reinterpret_cast<uint8_t**>(C + 2) = dst;
        C[4] = wt_coord_term(m[0], 0.0, v0); C[5] = wt_coord_term(m[0], 0.0, v1);
        C[6] = wt_coord_term(m[3], 0.0, v0); C[7] = wt_coord_term(m[3], 0.0, v1);
        return;
    }
    j -= gx;
    if (j < gy) {
        const int y1 = j * WT_TH + WT_TH < dh ? j * WT_TH + WT_TH : dh;
        const double v0 = (double)(j * WT_TH), v1 = (double)(y1 - 1);
        int32_t* R = T + L.row + 4 * j;
        R[0] = wt_coord_term(m[1], m[2], v0) + 16; R[1] = wt_coord_term(m[1], m[2], v1) + 16;
        R[2] = wt_coord_term(m[4], m[5], v0) + 16; R[3] = wt_coord_term(m[4], m[5], v1) + 16;
    }
}

// The whole table of a plane by the `nthreads` threads of a workgroup (thread `tid`).
__device__ __forceinline__ void wt_build_plane(const WarpTabJob& job, const double* m, int tid, int nthreads) {
    if (!job.tabs) return;
    const TabLayout L = tab_layout(job.dw, job.dh);
    const int n = wt_entries(job.dw, job.dh);
    for (int j = tid; j < n; j += nthreads) wt_build_entry(job.tabs, L, m, job.dw, job.dh, job.src, job.dst, j);
}
#endif

}  // namespace vsd
#endif
