// vs_stab_*: one video stream = the device-resident mirror of vs::Stabilizer
// (/root/reference/include/video/Stabilizer.h:70-198, src/Stabilizer.cpp:50-1172).
//
// The host side only keeps what is decidable without looking at pixels: the
// frame queue bookkeeping (which ring slot holds which frame index, when the
// warm-up ends, on which frames features are re-detected).  Everything that
// depends on image content stays on the GPU between kernels: keypoints and
// their count, LK status, the RANSAC model, the trajectory history and the
// warp matrix.  A steady-state stabilize() is a fixed set of asynchronous
// launches with no host synchronisation, spread over three HIP streams so that
// independent work of consecutive frames overlaps:
//
//   pre   : copy-in -> resize+gray -> pyramid + Scharr      (frame k+1 ...)
//   det   : goodFeaturesToTrack on the new gray image        (frame k, every 2nd)
//   main  : LK -> RANSAC+append -> trajectory emit -> warp   (... while frame k tracks)
//
// Cross-stream order is expressed with events only where data demands it
// (pyramid ready, keypoints ready, buffer no longer read).  Pyramids are
// triple-buffered and the frame ring has spare slots so that `pre` of the next
// frame never waits for `main` of the current one.
//
// Batch mode (vs_stab_set_batch, device entry points): the same decisions are taken per
// push, but the device work of `batch` consecutive frames is issued together - one
// launch per stage over all frames (argument tables in device memory), one ordered
// tail kernel (selection + trajectory append + smoothing, state in LDS), one warp
// launch for the frames of a batch.  There is ONE batch schedule, group_run(): a vs_batch
// group runs it over the frames of all its streams, and a standalone instance in batch mode
// owns a private group of one.  DESIGN.md section 5 has the schedule.
#include <algorithm>
#include <array>
#include <atomic>
#include <cstring>
#include <cstdlib>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <vector>

#include "canvas.h"
#include "host_helper.h"
#include "traj_state.h"
#include "vs_common.h"
#include "warp_tab.h"

namespace vsd {

struct RansacTables;
int get_ransac_tables(int max_m, int iters, const RansacTables** out);
int launch_ransac(const float* d_from, const float* d_to, const uint8_t* d_status, int n, const int32_t* d_n,
                  float* d_vp, float* d_vc, int32_t* d_m, int min_points, double thr, int iters,
                  const RansacTables* tab, int32_t* d_counts, double* d_model, uint8_t* d_inliers,
                  int32_t* d_info, TrajState* traj, const TrajParams* tp, vs_debug_frame* dbg,
                  int have_prev_gray, hipStream_t st);
size_t lk_item_bytes();
int lk_fill_item(void* host_item, const LKLevel* levels, int max_level, const float* d_prev_pts, int n,
                 const int32_t* d_n, float* d_next_pts, uint8_t* d_status, float* d_err, int win, int max_iters,
                 double eps);
int launch_pyr_lk_batch(const void* d_table, int items, int n_max, int win, hipStream_t st);
size_t ransac_item_bytes();
int ransac_fill_item(void* host_item, const float* d_from, const float* d_to, const uint8_t* d_status, int n,
                     const int32_t* d_n, float* d_vp, float* d_vc, int32_t* d_m, int min_points, double thr, int iters,
                     const RansacTables* tab, int32_t* d_counts, double* d_model, uint8_t* d_inliers, int32_t* d_info,
                     TrajState* traj, const TrajParams* tp, vs_debug_frame* dbg, int have_prev_gray);
int launch_ransac_score_batch(const void* d_table, int items, int iters, int n_max, hipStream_t st);
size_t tail_item_bytes();
void tail_fill_item(void* host_item, int out_due, int out_idx, double* d_Minv_out, const WarpTabJob* tabs);
size_t tail_in_bytes();
void ransac_item_set_tail_in(void* host_item, void* d_tail_in);
size_t tail_seg_bytes();
void tail_fill_seg(void* host_seg, int first, int n, float* d_M_out, TrajState* traj, vs_debug_frame* dbg, int smoothing_method);
void tail_item_set_seg(void* host_item, int seg);
void ransac_item_set_last(void* host_item, int last);
int launch_ransac_tail_group(const void* d_table, const void* d_tail, const void* d_segs, const void* d_tail_in, int nsegs, int max_n, int items,
                             int any_apart, hipStream_t st);
size_t gftt_item_bytes();
int gftt_fill_item(void* host_item, const uint8_t* d_gray, size_t stride, int w, int h, int max_corners, double quality,
                   double min_distance, int block_size, const GfttWork& wk, float* d_pts, int32_t* d_count);
int launch_gftt_batch(const void* d_table, int items, int w, int h, int block_size, hipStream_t st, int what = 0);
int launch_traj_emit(TrajState* s, const TrajParams& p, int idx, float* M_out, double* Minv_out, vs_debug_frame* dbg,
                     hipStream_t st, float* t_out = nullptr);
int launch_traj_reset(TrajState* s, int smoothing_radius, hipStream_t st);
int launch_spin(int microseconds, hipStream_t st);
int launch_fade_blend(const uint8_t* d_hist, uint8_t* d_frame, size_t bytes, float alpha, float beta, hipStream_t st);
int launch_fade_update(uint8_t* d_hist, const uint8_t* d_stab, size_t sstride, int row_bytes, int rows, hipStream_t st);
int launch_make_border(const uint8_t* src, size_t sstride, int w, int h, int cn, uint8_t* dst, size_t dstride,
                       int b, int border, hipStream_t st);
int launch_resize_linear(const uint8_t* d_src, size_t sstride, int sw, int sh, int cn, uint8_t* d_dst,
                         size_t dstride, int dw, int dh, hipStream_t st);

constexpr int FRAME_RING = 128;     // <= 35 queued frames (clamp(smoothingRadius,5,35)) + slack so that a
                                    // slot is reused several frames after the warp that released it; in batch mode
                                    // also the batch being collected and the one whose warps are still to come
constexpr int FRAME_RING_MAX = 192; // the same for batches of more than 32 frames (s->ring_frames)
constexpr int MAX_PYR = 8;
constexpr int NPYR = 3;             // pyramid buffers: frame k writes k%3 while LK(k-1) still reads (k-1)%3,(k-2)%3
constexpr int WARP_BATCH_MAX = 32;   // = the warp kernel's frames per launch (k_warp.hip MAXB)
constexpr int BATCH_MAX = 64;        // frames analysed per launch in batch mode (vs_stab_set_batch)
constexpr int EVR = 4;              // per-frame event ring

struct Pyramid {
    uint8_t* img[MAX_PYR] = {};
    int16_t* der[MAX_PYR] = {};
};

}  // namespace vsd

using namespace vsd;

struct vs_batch;
int group_drain(vs_batch* g);
int group_run(vs_batch* g);
bool group_holds_warps(const vs_batch* g);
vs_batch* group_new_own(vs_stab* s);
void group_delete(vs_batch* g);

struct vs_stab {
    vs_params_c p;
    int device = 0;
    // batch mode: the schedule that runs this stream's batches (one launch per stage over the frames of all its streams) - the
    // vs_batch the stream was created in, or the private group of one a standalone instance owns (`own`, made by allocate())
    vs_batch* group = nullptr;
    vs_batch* own = nullptr;
    bool member = false;            // stream of a vs_batch_create group: driven through vs_batch_* only
    bool group_call = false;        // ... which set this around the vs_stab_* calls they make on a member
    hipStream_t st = nullptr;       // main
    hipStream_t st_pre = nullptr;
    hipStream_t st_det = nullptr;
    hipStream_t st_warp = nullptr;  // deferred (batched) warps, high priority
    bool shared_streams = false;    // the four streams belong to the per-device pool
    std::string err;
    // geometry, fixed by the first frame
    bool allocated = false;
    int w = 0, h = 0, fmt = VS_FMT_BGR8, cn = 3;
    size_t row_bytes = 0, frame_bytes = 0;
    size_t src_pitch = 0;               // row pitch of the frames the pipeline reads: row_bytes (queue ring) or the caller's (zero-copy)
    size_t in_uv_off = 0, out_uv_off = 0;   // NV12 surfaces of the device entry points: UV plane offset, 0 = h * pitch
    int rows_total = 0;
    int aw = 960, ah = 540;
    int levels = 0;                 // max pyramid level actually used
    int lw[MAX_PYR], lh[MAX_PYR];
    // frame queue (Stabilizer.h:311-312)
    uint8_t* d_ring = nullptr;
    std::deque<int> q_slot, q_idx;      // ring slot (-1: the caller's own buffer, zero-copy mode) and frame index
    std::deque<const uint8_t*> q_ptr;   // where the queued frame lives
    bool zero_copy = false;             // vs_stab_set_zero_copy
    std::deque<int> free_slots;     // FIFO: the slot released longest ago is reused first
    bool first = true;
    int next_index = 0;             // index of the frame being pushed (nextFrameIndex_)
    int detect_counter = 0;
    int n_transforms = 0;           // transforms_.size(), mirrored on the host
    int last_out_w = 0, last_out_h = 0;
    int orig_w = 0, orig_h = 0;
    int host_radius = 30;
    int dbg_delay_us = 0;           // VS_STAB_DEBUG_DELAY_US: a spin kernel between tracking and RANSAC (ordering tests)
    // analysis images
    uint8_t* d_first_gray = nullptr;     // 480x270 (Stabilizer.cpp:277)
    std::vector<Pyramid> pyr;           // ring: NPYR buffers, 2*batch+2 in batch mode
    int npyr = NPYR;
    bool prev_small = false;
    bool have_prev_gray = false;
    // keypoints (ping-pong: LK reads pts[pp], a re-detection writes pts[pp^1])
    int ncap = 0;
    // keypoint buffers: [0],[1] ping-pong per frame; batch mode cycles through all of them
    std::vector<float*> d_pts;
    std::vector<int32_t*> d_npts;
    std::vector<int> pts_cap;
    int pp = 0;
    int last_lk_pp = 0;
    float *d_next = nullptr, *d_err = nullptr, *d_vp = nullptr, *d_vc = nullptr;
    uint8_t *d_status = nullptr, *d_inliers = nullptr;
    int32_t *d_m = nullptr, *d_info = nullptr, *d_counts = nullptr;
    double* d_model = nullptr;
    void* d_gftt_scratch = nullptr;
    GfttWork gw;
    const RansacTables* tab = nullptr;
    TrajState* d_traj = nullptr;
    TrajParams tp;
    float* d_M = nullptr;               // [0..5] frame matrix, [6..11] chroma matrix
    double* d_Minv = nullptr;           // their inverse maps (what the warp kernels consume)
    vs_debug_frame* d_dbg = nullptr;
    int last_detect_pp = -1;            // buffer that holds the points detected on the last push
    bool last_detected = false;
    int last_gray_buf = 0;
    // scratch for border / host I/O
    uint8_t* d_tmp = nullptr;
    size_t tmp_bytes = 0;
    uint8_t* d_fade = nullptr;          // borderType "fade": borderHistory_ (padded frame, packed rows)
    bool fade_valid = false;
    Canvas* canvas = nullptr;           // enableVirtualCanvas: temporal buffer and canvas geometry (outlive clean(), like the fade history)
    float* d_ct = nullptr;              // the correction (dx, dy, da) of the output being produced, for the canvas
    int fade_count = 0, fade_w = 0, fade_h = 0;     // fadeFrameCount_; geometry of the history
    uint8_t* d_padB = nullptr;          // batch mode with a border: one padded (or to-be-cropped) frame per frame of a batch
    size_t pad_frame_bytes = 0;
    uint8_t* d_out = nullptr;
    size_t out_bytes = 0;
    // host pipeline (vs_stab_set_host_pipeline): the result of a call stays in d_hold[] and travels to the host during the
    // NEXT call, next to that call's upload and ahead of its analysis
    bool host_pipe = false, hold_valid = false;
    uint8_t* d_hold[2] = {nullptr, nullptr};
    int hold_cur = 0, hold_w = 0, hold_h = 0;
    hipEvent_t ev_hold = nullptr;
    std::unique_ptr<HostHelper> helper;     // issues the download when the caller's output buffer is pageable (push_host_pipelined)
    uint8_t* d_all = nullptr;           // one allocation for the small buffers
    vs_counters counters;
    // cross-stream dependencies
    hipEvent_t ev_gray[NPYR] = {}, ev_pre[NPYR] = {};
    hipEvent_t ev_lk[EVR] = {}, ev_det[EVR] = {};
    bool det_valid[EVR] = {false, false, false, false};
    hipEvent_t ev_first = nullptr;
    hipEvent_t ev_slot[FRAME_RING_MAX] = {};
    bool slot_valid[FRAME_RING_MAX] = {};
    int ring_frames = FRAME_RING;
    hipEvent_t pts_event[2] = {nullptr, nullptr};   // recorded by the detection that filled pts[i]
    bool pts_pending[2] = {false, false};
    // deferred output (vs_stab_set_warp_batch): warps of consecutive outputs wait for each other and
    // go out as ONE launch over up to WARP_BATCH_MAX frames, each into its caller's buffer
    int warp_batch = 1;
    struct PendWarp { const uint8_t* src; uint8_t* dst; int slot; };
    std::vector<PendWarp> pend;
    size_t pend_stride = 0;
    double* d_MinvB[2] = {nullptr, nullptr};   // inverse maps of the pending frames, 12 doubles each; two sets
    int32_t* d_tabs_def = nullptr;              // coordinate tables of a deferred warp launch
    int pend_set = 0;
    hipEvent_t ev_emit = nullptr, ev_warp[2] = {nullptr, nullptr};
    bool warp_valid[2] = {false, false};
    // batch mode (vs_stab_set_batch): the latency-bound analysis stages (GFTT, LK, RANSAC scoring) of `batch`
    // consecutive frames run as ONE launch each; the ordered tail (selection + trajectory append, emit) stays
    // per frame; the warps go out through the deferred list above.
    int batch = 1;
    bool batch_active = false;
    struct BFrame {
        int f, c, pv;
        const uint8_t* frame; bool prev_small;
        bool detect; int det_buf;
        int lk_buf, lk_cap;
        bool out_due; int out_slot, out_idx; const uint8_t* out_frame; uint8_t* d_out; size_t out_stride;
        int have_prev_gray;
    };
    std::vector<BFrame> bq;
    int kp_cur = 0, kp_next = 1;
    std::vector<GfttWork> gws;                       // one GFTT scratch per detection of a batch
    struct ItemBufs { float *next, *err, *vp, *vc; uint8_t *status, *inliers; int32_t *m, *info, *counts; double* model; };
    std::vector<ItemBufs> items;
    // what the debug getters read (last analysed frame)
    const float* dbg_prev_pts = nullptr; const float* dbg_next = nullptr;
    const uint8_t *dbg_status = nullptr, *dbg_inliers = nullptr;
    const float* dbg_det_pts = nullptr; const int32_t* dbg_det_n = nullptr;
    const int32_t* dbg_gftt_counters = nullptr;
    // stage profiling (HIP events on the stream the stage runs on)
    int prof_mode = 0;
    struct Pending { hipEvent_t a, b; int stage; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> ev_pool;
};

namespace {

// Records an event pair around one stage when profiling is on.
struct StageScope {
    vs_stab* s;
    int stage;
    hipStream_t st;
    bool on = false;
    hipEvent_t a = nullptr, b = nullptr;
    static hipEvent_t get(vs_stab* s) {
        if (!s->ev_pool.empty()) { hipEvent_t e = s->ev_pool.back(); s->ev_pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
    StageScope(vs_stab* s_, int stage_, hipStream_t st_) : s(s_), stage(stage_), st(st_) {
        if (s->prof_mode == 2 || (s->prof_mode == 1 && stage == VS_STAGE_WARP) ||
            (s->prof_mode == 3 && (stage == VS_STAGE_WARP || stage == VS_STAGE_WARP_TABLES))) {
            a = get(s); b = get(s);
            if (a && b && hipEventRecord(a, st) == hipSuccess) on = true;
        }
    }
    ~StageScope() {
        if (on && hipEventRecord(b, st) == hipSuccess) s->pending.push_back({a, b, stage});
    }
};

int fail(vs_stab* s, int code, const std::string& msg) {
    s->err = msg;
    set_last_error(msg);
    return code;
}

#define S_HIP(s, expr)                                                                   \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess)                                                            \
            return fail((s), VS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

#define S_TRY(s, expr)                                  \
    do {                                                \
        int _r = (expr);                                \
        if (_r != VS_OK) { (s)->err = get_last_error(); return _r; } \
    } while (0)

int effective_radius(int r) { return std::max(5, std::min(r, 35)); }

// enableVirtualCanvas acts where the reference reaches it: not behind the crop-and-zoom returns (Stabilizer.cpp:1108-1127)
bool canvas_on(const vs_stab* s) { return s->p.enable_virtual_canvas && !s->p.crop_n_zoom; }

void out_size(const vs_stab* s, int w, int h, int* ow, int* oh) {
    const int b = s->p.border_size;
    if (canvas_on(s)) { *ow = w; *oh = h; return; }     // the canvas window has the size of the unpadded frame (:2121-2126)
    if (b > 0 && !s->p.crop_n_zoom) { *ow = w + 2 * b; *oh = h + 2 * b; return; }
    *ow = w; *oh = h;   // crop+zoom resizes back to origSize_ == frame size
}

int flush_warps(vs_stab* s, bool on_main = false);

// Batch mode: everything queued so far is analysed and its warps are issued (nothing stays deferred).
int drain_batch(vs_stab* s) {
    if (!s->group) return VS_OK;
    const int rc = group_drain(s->group);
    if (rc != VS_OK) s->err = get_last_error();
    return rc;
}

int sync_all(vs_stab* s) {
    S_HIP(s, hipSetDevice(s->device));
    S_TRY(s, drain_batch(s));
    S_TRY(s, flush_warps(s));
    if (s->st_warp) S_HIP(s, hipStreamSynchronize(s->st_warp));
    if (s->st_pre) S_HIP(s, hipStreamSynchronize(s->st_pre));
    if (s->st_det) S_HIP(s, hipStreamSynchronize(s->st_det));
    if (s->st) S_HIP(s, hipStreamSynchronize(s->st));
    return VS_OK;
}

void free_all(vs_stab* s) {
    if (s->d_ring) (void)hipFree(s->d_ring);
    if (s->d_all) (void)hipFree(s->d_all);
    if (s->d_gftt_scratch) (void)hipFree(s->d_gftt_scratch);
    if (s->d_tmp) (void)hipFree(s->d_tmp);
    if (s->d_padB) (void)hipFree(s->d_padB);
    s->d_padB = nullptr;
    if (s->own) {                       // the private schedule goes with the buffers it was sized for
        group_delete(s->own);
        s->own = nullptr; s->group = nullptr;
    }
    // (d_fade, the fade history, outlives vs_stab_clean like borderHistory_ outlives Stabilizer::clean())
    if (s->d_out) (void)hipFree(s->d_out);
    for (auto& h : s->d_hold) { if (h) (void)hipFree(h); h = nullptr; }
    s->hold_valid = false;
    s->d_ring = s->d_all = s->d_tmp = s->d_out = nullptr;
    s->d_gftt_scratch = nullptr;
    s->allocated = false;
}

void analysis_size(const vs_stab* s, int w, int h, int* aw, int* ah) {
    *aw = 960; *ah = 540;                                  // Stabilizer.cpp:410
    if (s->p.drone_high_freq_mode) {                       // :2447-2466
        int maxWidth = std::min(s->p.hf_analysis_max_width, w);
        float aspect = (float)h / (float)w;
        int height = (int)(maxWidth * aspect);
        *aw = (maxWidth / 2) * 2;
        *ah = (height / 2) * 2;
    }
}

int allocate_buffers(vs_stab* s, int w, int h, int fmt);

// A failure half way leaves nothing behind: the next push starts from scratch instead of overwriting live pointers.
int allocate(vs_stab* s, int w, int h, int fmt) {
    const int rc = allocate_buffers(s, w, h, fmt);
    if (rc != VS_OK) {
        if (s->st) (void)hipStreamSynchronize(s->st);
        free_all(s);
    }
    return rc;
}

int allocate_buffers(vs_stab* s, int w, int h, int fmt) {
    s->w = w; s->h = h; s->fmt = fmt;
    s->cn = fmt == VS_FMT_BGR8 ? 3 : 1;
    s->rows_total = fmt == VS_FMT_NV12 ? h * 3 / 2 : h;
    s->row_bytes = (size_t)w * s->cn;
    s->frame_bytes = s->row_bytes * s->rows_total;
    s->src_pitch = s->row_bytes;
    analysis_size(s, w, h, &s->aw, &s->ah);
    if (s->aw < 3 || s->ah < 3) return fail(s, VS_ERR_INVALID_ARG, "analysis size too small");
    // (the reference's cvtColor(BGR2GRAY) of the canvas, Stabilizer.cpp:2225, throws on anything but three channels)
    if (canvas_on(s) && fmt != VS_FMT_BGR8) return fail(s, VS_ERR_UNSUPPORTED, "enableVirtualCanvas needs a BGR8 stream");
    // buildOpticalFlowPyramid: levels that fit the window
    {
        int sw = s->aw, sh = s->ah;
        for (int level = 0; level <= s->p.lk_max_level; level++) {
            s->lw[level] = sw; s->lh[level] = sh;
            s->levels = level;
            sw = (sw + 1) / 2; sh = (sh + 1) / 2;
            if (sw <= s->p.lk_win_size || sh <= s->p.lk_win_size) break;
        }
    }
    // (adaptive smoothing stays with the per-frame pipeline: whether a push produces a frame then depends on the data - the
    // radius moves the warm-up threshold, Stabilizer.cpp:383,1482-1486 - and a push answers that at once)
    // (so does borderType "fade": each output is blended with a history the output before it has just updated)
    const bool fade = s->p.border_type == VS_BORDER_FADE && s->p.border_size > 0 && !s->p.crop_n_zoom;
    // (and the virtual canvas: the window offset and the temporal fill are host decisions on each output's correction)
    s->batch_active = s->batch > 1 && !s->p.adaptive_smoothing && !fade && !canvas_on(s);
    const int B = s->batch_active ? s->batch : 1;
    s->npyr = s->batch_active ? 2 * B + 2 : NPYR;
    // keypoint buffers: one per detection, recycled after two batches' worth of detections
    const int nkp = s->batch_active ? B + 4 : 2, ngw = s->batch_active ? B / 2 + 1 : 1;
    s->pyr.assign(s->npyr, Pyramid());
    s->d_pts.assign(nkp, nullptr); s->d_npts.assign(nkp, nullptr); s->pts_cap.assign(nkp, 0);
    s->items.assign(B, vs_stab::ItemBufs());
    s->bq.clear(); s->kp_cur = 0; s->kp_next = 1;
    if (s->batch_active && !s->group) {            // a standalone instance: a group of one runs its batches
        s->own = group_new_own(s);
        if (!s->own) return fail(s, VS_ERR_HIP, get_last_error());
        s->group = s->own;
    }
    // (the frame queue ring - 128 frames, 3.2 GB at 4K BGR8 - is allocated by the first push that copies a frame in: a
    // stream that only ever hands over device frames in zero-copy mode never needs it)
    s->free_slots.clear();
    s->ring_frames = (s->batch_active && B > 32) ? FRAME_RING_MAX : FRAME_RING;
    for (int i = 0; i < s->ring_frames; i++) { s->free_slots.push_back(i); s->slot_valid[i] = false; }
    s->ncap = std::max(s->p.max_corners, 1);
    const int ncap = s->ncap;
    // carve the small buffers out of one allocation (256-byte aligned pieces)
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    size_t o_first = take((size_t)480 * 270);
    std::vector<std::array<size_t, MAX_PYR>> o_img(s->npyr), o_der(s->npyr);
    for (int k = 0; k < s->npyr; k++)
        for (int l = 0; l <= s->levels; l++) {
            o_img[k][l] = take((size_t)s->lw[l] * s->lh[l]);
            o_der[k][l] = take((size_t)s->lw[l] * s->lh[l] * 4);
        }
    std::vector<size_t> o_pts(nkp), o_npts(nkp);
    for (int k = 0; k < nkp; k++) { o_pts[k] = take((size_t)ncap * 8); o_npts[k] = take(16); }
    struct ItemOff { size_t next, err, vp, vc, status, inl, m, info, counts, model; };
    std::vector<ItemOff> o_it(B);
    for (int k = 0; k < B; k++) {
        o_it[k].next = take((size_t)ncap * 8); o_it[k].err = take((size_t)ncap * 4);
        o_it[k].vp = take((size_t)ncap * 8); o_it[k].vc = take((size_t)ncap * 8);
        o_it[k].status = take(ncap); o_it[k].inl = take(ncap);
        o_it[k].m = take(16); o_it[k].info = take(16); o_it[k].counts = take((size_t)s->p.ransac_max_iters * 4);
        o_it[k].model = take(48);
    }
    size_t o_traj = take(sizeof(TrajState)), o_M = take(96), o_Minv = take(96), o_dbg = take(sizeof(vs_debug_frame));
    size_t o_MinvB[2] = {take((size_t)BATCH_MAX * 96), take((size_t)BATCH_MAX * 96)};
    int tow, toh;
    out_size(s, w, h, &tow, &toh);
    const size_t o_tabs = take(warp_tabs_ints(std::max(w, tow), std::max(h, toh), WARP_BATCH_MAX) * sizeof(int32_t));
    S_HIP(s, hipMalloc((void**)&s->d_all, off));
    S_HIP(s, hipMemsetAsync(s->d_all, 0, off, s->st));
    uint8_t* b = s->d_all;
    s->d_first_gray = b + o_first;
    for (int k = 0; k < s->npyr; k++)
        for (int l = 0; l <= s->levels; l++) {
            s->pyr[k].img[l] = b + o_img[k][l];
            s->pyr[k].der[l] = (int16_t*)(b + o_der[k][l]);
        }
    for (int k = 0; k < nkp; k++) { s->d_pts[k] = (float*)(b + o_pts[k]); s->d_npts[k] = (int32_t*)(b + o_npts[k]); }
    for (int k = 0; k < B; k++) {
        vs_stab::ItemBufs& it = s->items[k];
        it.next = (float*)(b + o_it[k].next); it.err = (float*)(b + o_it[k].err);
        it.vp = (float*)(b + o_it[k].vp); it.vc = (float*)(b + o_it[k].vc);
        it.status = b + o_it[k].status; it.inliers = b + o_it[k].inl;
        it.m = (int32_t*)(b + o_it[k].m); it.info = (int32_t*)(b + o_it[k].info); it.counts = (int32_t*)(b + o_it[k].counts);
        it.model = (double*)(b + o_it[k].model);
    }
    // the per-frame path works on item 0
    s->d_next = s->items[0].next; s->d_err = s->items[0].err; s->d_vp = s->items[0].vp; s->d_vc = s->items[0].vc;
    s->d_status = s->items[0].status; s->d_inliers = s->items[0].inliers;
    s->d_m = s->items[0].m; s->d_info = s->items[0].info; s->d_counts = s->items[0].counts; s->d_model = s->items[0].model;
    s->d_traj = (TrajState*)(b + o_traj);
    s->d_M = (float*)(b + o_M); s->d_Minv = (double*)(b + o_Minv); s->d_dbg = (vs_debug_frame*)(b + o_dbg);
    s->d_MinvB[0] = (double*)(b + o_MinvB[0]); s->d_MinvB[1] = (double*)(b + o_MinvB[1]);
    s->d_tabs_def = (int32_t*)(b + o_tabs);
    s->pend.clear(); s->pend_set = 0; s->warp_valid[0] = s->warp_valid[1] = false;
    // GFTT scratch sized for the larger of the two detection images
    const int gmaxw = std::max(s->aw, 480), gmaxh = std::max(s->ah, 270);
    const int cap = gmaxw * gmaxh / 4 + 64;
    const size_t gwb = (gftt_work_bytes(gmaxw, gmaxh, cap) + 255) & ~(size_t)255;
    S_HIP(s, hipMalloc(&s->d_gftt_scratch, gwb * ngw));
    s->gws.assign(ngw, GfttWork());
    for (int k = 0; k < ngw; k++) gftt_work_carve((uint8_t*)s->d_gftt_scratch + gwb * k, gmaxw, gmaxh, cap, &s->gws[k]);
    s->gw = s->gws[0];
    s->dbg_gftt_counters = s->gw.counters;
    S_TRY(s, get_ransac_tables(ncap, s->p.ransac_max_iters, &s->tab));
    int ow, oh;
    out_size(s, w, h, &ow, &oh);
    s->out_bytes = (size_t)ow * s->cn * (fmt == VS_FMT_NV12 ? oh * 3 / 2 : oh);
    S_HIP(s, hipMalloc((void**)&s->d_out, s->out_bytes));
    s->tmp_bytes = std::max(s->out_bytes, s->frame_bytes);
    S_HIP(s, hipMalloc((void**)&s->d_tmp, s->tmp_bytes + 16));
    if (s->batch_active && s->p.border_size > 0) {
        s->pad_frame_bytes = (s->tmp_bytes + 255) & ~(size_t)255;
        S_HIP(s, hipMalloc((void**)&s->d_padB, s->pad_frame_bytes * B));
    }
    S_TRY(s, launch_traj_reset(s->d_traj, s->p.smoothing_radius, s->st));
    // the zero-fill and the reset ran on `main`; nothing may touch the buffers before that
    S_HIP(s, hipStreamSynchronize(s->st));
    for (int i = 0; i < EVR; i++) s->det_valid[i] = false;
    s->pts_pending[0] = s->pts_pending[1] = false;
    s->allocated = true;
    return VS_OK;
}

void fill_traj_params(vs_stab* s) {
    TrajParams& t = s->tp;
    memset(&t, 0, sizeof t);
    const vs_params_c& p = s->p;
    t.method = p.smoothing_method;
    t.horizon_lock = p.horizon_lock;
    t.drone = p.drone_high_freq_mode;
    t.adaptive = p.adaptive_smoothing;
    t.min_radius = p.min_smoothing_radius;
    t.max_radius = p.max_smoothing_radius;
    t.hf_shake_px = p.hf_shake_px;
    t.hf_rot_lp_alpha = p.hf_rot_lp_alpha;
    t.hf_dead_zone = p.hf_dead_zone_threshold;
    t.hf_decay = p.hf_motion_accumulator_decay;
    t.hf_freeze_duration = p.hf_freeze_duration;
    // gaussianFilterConvolve kernel (Stabilizer.cpp:1368-1386), built with the host libm
    float sigma = (float)p.gaussian_sigma;
    int ks = std::max(3, (int)std::ceil(6 * sigma));
    if (ks % 2 == 0) ks++;
    if (ks > GAUSS_MAX) ks = GAUSS_MAX;
    t.gauss_ksize = ks;
    float sum = 0.0f;
    int center = ks / 2;
    for (int i = 0; i < ks; i++) {
        float x = (float)(i - center);
        t.gauss_kernel[i] = std::exp(-(x * x) / (2 * sigma * sigma));
        sum += t.gauss_kernel[i];
    }
    for (int i = 0; i < ks; i++) t.gauss_kernel[i] /= sum;
}

int build_pyramid(vs_stab* s, int k, hipStream_t st) {
    Pyramid& P = s->pyr[k];
    for (int l = 1; l <= s->levels; l++)
        S_TRY(s, launch_pyr_down(P.img[l - 1], s->lw[l - 1], s->lw[l - 1], s->lh[l - 1], P.img[l], s->lw[l], st));
    for (int l = 0; l <= s->levels; l++)
        S_TRY(s, launch_scharr(P.img[l], s->lw[l], s->lw[l], s->lh[l], P.der[l], st));
    return VS_OK;
}

// NV12: where the interleaved UV plane of a queued frame / of an output surface starts
inline size_t src_uv(const vs_stab* s) { return (s->zero_copy && s->in_uv_off) ? s->in_uv_off : (size_t)s->h * s->src_pitch; }
inline size_t dst_uv(const vs_stab* s, const uint8_t* d_out, size_t out_stride) {
    return (d_out != s->d_out && s->out_uv_off) ? s->out_uv_off : (size_t)s->h * out_stride;   // s->d_out: staging of the host entry points
}

// `pre` stream, part 1: the frame enters the queue ring (waits until the slot's last reader is done)
int enqueue_copy_in(vs_stab* s, int slot, const void* src, size_t stride, hipMemcpyKind kind) {
    if (s->slot_valid[slot]) S_HIP(s, hipStreamWaitEvent(s->st_pre, s->ev_slot[slot], 0));
    StageScope t(s, VS_STAGE_COPY_IN, s->st_pre);
    uint8_t* dst = s->d_ring + (size_t)slot * s->frame_bytes;
    if (s->fmt == VS_FMT_NV12 && kind == hipMemcpyDeviceToDevice && s->in_uv_off) {      // decoder surface: planes apart
        S_HIP(s, hipMemcpy2DAsync(dst, s->row_bytes, src, stride, s->row_bytes, s->h, kind, s->st_pre));
        S_HIP(s, hipMemcpy2DAsync(dst + (size_t)s->h * s->row_bytes, s->row_bytes, (const uint8_t*)src + s->in_uv_off, stride,
                                  s->row_bytes, s->h / 2, kind, s->st_pre));
        return VS_OK;
    }
    S_HIP(s, hipMemcpy2DAsync(dst, s->row_bytes, src, stride, s->row_bytes, s->rows_total, kind, s->st_pre));
    return VS_OK;
}

// generateTransform (Stabilizer.cpp:402-761) for frame index f >= 1 held in `d_frame`
int generate_transform(vs_stab* s, const uint8_t* d_frame, int f) {
    const vs_params_c& p = s->p;
    const int c = f % NPYR, pv = (f - 1) % NPYR;
    // ---- pre: gray + pyramid into buffer c.  Its previous readers were LK(f-2) (as "prev")
    // and, if frame f-3 re-detected, the detector.
    if (f - 2 >= 1) S_HIP(s, hipStreamWaitEvent(s->st_pre, s->ev_lk[(f - 2) % EVR], 0));
    if (f - 3 >= 1 && s->det_valid[(f - 3) % EVR]) S_HIP(s, hipStreamWaitEvent(s->st_pre, s->ev_det[(f - 3) % EVR], 0));
    {
        StageScope t(s, VS_STAGE_GRAY, s->st_pre);
        S_TRY(s, launch_resize_gray(d_frame, s->src_pitch, s->w, s->h, s->fmt, s->pyr[c].img[0], s->aw, s->aw, s->ah, s->st_pre));  // :448-450
    }
    S_HIP(s, hipEventRecord(s->ev_gray[c], s->st_pre));
    {
        StageScope t(s, VS_STAGE_PYRAMID, s->st_pre);
        S_TRY(s, build_pyramid(s, c, s->st_pre));
        if (s->prev_small) {   // :598-603 (once: 480x270 -> analysis size)
            S_TRY(s, launch_resize_gray(s->d_first_gray, 480, 480, 270, VS_FMT_GRAY8, s->pyr[pv].img[0], s->aw, s->aw, s->ah, s->st_pre));
            S_TRY(s, build_pyramid(s, pv, s->st_pre));
            s->prev_small = false;
        }
    }
    S_HIP(s, hipEventRecord(s->ev_pre[c], s->st_pre));

    // ---- det: every second call re-detects on the new gray image (:696-746).  Needs only
    // img[0]; its output buffer pts[pp^1] was last read by LK(f-2), which `pre` waited for.
    const int pp = s->pp;
    s->last_detected = false;
    s->det_valid[f % EVR] = false;
    int next_pp = pp;
    if ((++s->detect_counter % 2) == 0) {
        const int q = pp ^ 1;
        const int mc = std::min(p.max_corners, 200);
        S_HIP(s, hipStreamWaitEvent(s->st_det, s->ev_gray[c], 0));
        {
            StageScope t(s, VS_STAGE_GFTT, s->st_det);
            S_TRY(s, launch_gftt(s->pyr[c].img[0], s->aw, s->aw, s->ah, mc, 0.02, 15.0, 3, s->gw, s->d_pts[q],
                                 s->d_npts[q], s->st_det));
        }
        S_HIP(s, hipEventRecord(s->ev_det[f % EVR], s->st_det));
        s->det_valid[f % EVR] = true;
        s->pts_event[q] = s->ev_det[f % EVR];
        s->pts_pending[q] = true;
        s->pts_cap[q] = mc;
        next_pp = q;
        s->last_detected = true;
        s->last_detect_pp = q;
        s->dbg_det_pts = s->d_pts[q]; s->dbg_det_n = s->d_npts[q];
        s->counters.detections++;
    }

    // ---- main: LK needs this frame's pyramid and the keypoints of the last detection
    S_HIP(s, hipStreamWaitEvent(s->st, s->ev_pre[c], 0));
    if (s->pts_pending[pp]) {
        S_HIP(s, hipStreamWaitEvent(s->st, s->pts_event[pp], 0));
        s->pts_pending[pp] = false;
    }
    LKLevel L[MAX_PYR];
    for (int l = 0; l <= s->levels; l++) {
        L[l].prev = s->pyr[pv].img[l]; L[l].next = s->pyr[c].img[l]; L[l].deriv = s->pyr[pv].der[l];
        L[l].w = s->lw[l]; L[l].h = s->lh[l]; L[l].stride = s->lw[l];
    }
    const int cap = s->pts_cap[pp];
    {
        StageScope t(s, VS_STAGE_LK, s->st);
        S_TRY(s, launch_pyr_lk(L, s->levels, s->d_pts[pp], cap, s->d_npts[pp], s->d_next, s->d_status, s->d_err,
                               p.lk_win_size, p.lk_max_iters, p.lk_epsilon, s->st));   // :611-619
    }
    s->last_lk_pp = pp;
    if (s->dbg_delay_us > 0) S_TRY(s, launch_spin(s->dbg_delay_us, s->st));        // test hook: widen the window between LK and RANSAC
    {
        // status compaction (:629-641) + estimateAffinePartial2D (:644-659) + transform append (:660-693)
        StageScope t(s, VS_STAGE_RANSAC, s->st);
        S_TRY(s, launch_ransac(s->d_pts[pp], s->d_next, s->d_status, std::max(cap, 0), s->d_npts[pp], s->d_vp, s->d_vc,
                               s->d_m, 4, p.ransac_threshold, p.ransac_max_iters, s->tab, s->d_counts, s->d_model,
                               s->d_inliers, s->d_info, s->d_traj, &s->tp, s->d_dbg, s->have_prev_gray ? 1 : 0, s->st));
    }
    // `pre` (pyramid buffers) and `det` (keypoint buffer pts[pp^1] two frames on) wait for this event: the tracker AND the
    // scoring / selection kernels have read pts[pp] and its count by then (recorded right behind the tracker, a re-detection
    // two frames later could overwrite the buffer under the RANSAC kernels)
    S_HIP(s, hipEventRecord(s->ev_lk[f % EVR], s->st));
    s->dbg_prev_pts = s->d_pts[pp]; s->dbg_next = s->d_next; s->dbg_status = s->d_status; s->dbg_inliers = s->d_inliers;
    s->n_transforms++;
    s->pp = next_pp;
    s->last_gray_buf = c;
    s->have_prev_gray = true;                                                         // :757-759
    return VS_OK;
}

// One launch for all pending warps; releases their ring slots.  Per-frame pipeline: on the high-priority warp
// stream, so that the analysis of the next frames is not held up.  Batch mode (on_main): in line on `main`,
// between the tail of this batch and the tracking of the next - the tracking kernel keeps ~90 KB of LDS per
// CU busy for its whole (latency-bound) run, which would leave room for 3 warp workgroups per CU instead of 8.
int flush_warps(vs_stab* s, bool on_main) {
    if (s->pend.empty()) return VS_OK;
    hipStream_t ws = on_main ? s->st : s->st_warp;
    const int n = (int)s->pend.size(), set = s->pend_set;
    const uint8_t* srcs[WARP_BATCH_MAX];
    uint8_t* dsts[WARP_BATCH_MAX];
    for (int i = 0; i < n; i++) { srcs[i] = s->pend[i].src; dsts[i] = s->pend[i].dst; }
    if (!on_main) {
        S_HIP(s, hipEventRecord(s->ev_emit, s->st));             // the maps of this batch are written on `main`
        S_HIP(s, hipStreamWaitEvent(ws, s->ev_emit, 0));
    }
    int rc;
    {
        StageScope t(s, VS_STAGE_WARP, ws);
        rc = launch_warp_affine_list(srcs, dsts, n, s->src_pitch, s->w, s->h, s->pend_stride, s->w, s->h, s->cn,
                                     s->d_MinvB[set], 12, n >= 4 ? s->d_tabs_def : nullptr, ws);
    }
    if (hipEventRecord(s->ev_warp[set], ws) == hipSuccess) s->warp_valid[set] = true;
    for (int i = 0; i < n; i++) {
        const int slot = s->pend[i].slot;
        if (slot < 0) continue;          // zero-copy: the frame is the caller's
        if (hipEventRecord(s->ev_slot[slot], ws) == hipSuccess) s->slot_valid[slot] = true;
        s->free_slots.push_back(slot);
    }
    s->pend.clear();
    s->pend_set = set ^ 1;
    if (rc != VS_OK) { s->err = get_last_error(); return rc; }
    return VS_OK;
}

// Deferred output: only the map of output `idx` is computed now (on `main`, in trajectory order); its warp
// joins the next batched launch.
int defer_output(vs_stab* s, int idx, const uint8_t* frame, uint8_t* d_out, size_t out_stride, int slot) {
    hipStream_t st = s->st;
    if (!s->pend.empty() && s->pend_stride != out_stride) S_TRY(s, flush_warps(s));
    const int set = s->pend_set, j = (int)s->pend.size();
    if (j == 0 && s->warp_valid[set]) {      // the previous user of this set of maps must have read them
        S_HIP(s, hipStreamWaitEvent(st, s->ev_warp[set], 0));
        s->warp_valid[set] = false;
    }
    {
        StageScope t(s, VS_STAGE_TRAJ, st);
        S_TRY(s, launch_traj_emit(s->d_traj, s->tp, idx, s->d_M, s->d_MinvB[set] + 12 * j, s->d_dbg, st));
    }
    s->pend.push_back({frame, d_out, slot});
    s->pend_stride = out_stride;
    if ((int)s->pend.size() >= s->warp_batch) S_TRY(s, flush_warps(s));
    return VS_OK;
}

// applyNextSmoothTransform (Stabilizer.cpp:763-1137) into d_out (device), on `main`
int apply_next(vs_stab* s, uint8_t* d_out, size_t out_stride, bool may_defer) {
    const vs_params_c& p = s->p;
    const int slot = s->q_slot.front(), idx = s->q_idx.front();
    const uint8_t* frame = s->q_ptr.front();
    s->q_slot.pop_front(); s->q_idx.pop_front(); s->q_ptr.pop_front();
    hipStream_t st = s->st;
    int ow, oh;
    out_size(s, s->w, s->h, &ow, &oh);
    s->last_out_w = ow; s->last_out_h = oh;
    const bool plain = idx < s->n_transforms && s->fmt != VS_FMT_NV12 && p.border_size <= 0 && !canvas_on(s);
    if (may_defer && plain && s->warp_batch > 1) {
        S_TRY(s, defer_output(s, idx, frame, d_out, out_stride, slot));
        s->counters.frames_out++;
        return VS_OK;
    }
    {
        StageScope t(s, VS_STAGE_TRAJ, st);
        if (canvas_on(s) && !s->d_ct) S_HIP(s, hipMalloc((void**)&s->d_ct, 4 * sizeof(float)));
        S_TRY(s, launch_traj_emit(s->d_traj, s->tp, idx, s->d_M, s->d_Minv, s->d_dbg, st, canvas_on(s) ? s->d_ct : nullptr));
    }
    int rc = VS_OK;
    if (idx >= s->n_transforms) {
        // Stabilizer.cpp:774-780: no transform exists for this frame (last frame of a
        // flush): the queued frame is returned as is, at its own size (no border pad).
        if (ow != s->w || oh != s->h)
            S_HIP(s, hipMemset2DAsync(d_out, out_stride, 0, (size_t)ow * s->cn, oh, st));
        S_HIP(s, hipMemcpy2DAsync(d_out, out_stride, frame, s->src_pitch, s->row_bytes, s->h, hipMemcpyDeviceToDevice, st));
        if (s->fmt == VS_FMT_NV12)
            S_HIP(s, hipMemcpy2DAsync(d_out + dst_uv(s, d_out, out_stride), out_stride, frame + src_uv(s), s->src_pitch, s->row_bytes,
                                      s->h / 2, hipMemcpyDeviceToDevice, st));
        s->last_out_w = s->w; s->last_out_h = s->h;
    } else if (canvas_on(s)) {                                                        // :1130-1134
        // the canvas replaces the warped frame, so the warp (and a fade history behind it) cannot be observed and is not run
        if (!s->canvas && !(s->canvas = canvas_new())) return fail(s, VS_ERR_HIP, "out of host memory");
        StageScope t(s, VS_STAGE_WARP, st);
        rc = canvas_apply(s->canvas, p, frame, s->src_pitch, s->w, s->h, s->d_ct, s->d_traj, d_out, out_stride, st);
    } else if (s->fmt == VS_FMT_NV12) {
        StageScope t(s, VS_STAGE_WARP, st);
        rc = launch_warp_affine(frame, s->src_pitch, 0, s->w, s->h, d_out, out_stride, 0, s->w, s->h, 1, s->d_Minv, 1, nullptr, st);
        if (rc == VS_OK)
            rc = launch_warp_affine(frame + src_uv(s), s->src_pitch, 0, s->w / 2, s->h / 2,
                                    d_out + dst_uv(s, d_out, out_stride), out_stride, 0, s->w / 2, s->h / 2, 2,
                                    s->d_Minv + 6, 1, nullptr, st);
    } else if (p.border_size > 0 && !p.crop_n_zoom && p.border_type == VS_BORDER_FADE) {     // :914-978, :1069-1106
        const int b = p.border_size, bw = s->w + 2 * b, bh = s->h + 2 * b;
        const size_t prow = (size_t)bw * s->cn, nb = prow * bh;
        rc = launch_make_border(frame, s->src_pitch, s->w, s->h, s->cn, s->d_tmp, prow, b, VS_BORDER_BLACK, st);
        if (rc == VS_OK && (!s->d_fade || s->fade_w != bw || s->fade_h != bh)) {     // :917-926 the first padded frame is the history
            if (s->d_fade) { (void)hipStreamSynchronize(st); (void)hipFree(s->d_fade); s->d_fade = nullptr; }
            S_HIP(s, hipMalloc((void**)&s->d_fade, (nb + 3) & ~(size_t)3));
            s->fade_w = bw; s->fade_h = bh; s->fade_valid = false;
        }
        if (rc == VS_OK && !s->fade_valid) {
            S_HIP(s, hipMemcpyAsync(s->d_fade, s->d_tmp, nb, hipMemcpyDeviceToDevice, st));
            s->fade_valid = true; s->fade_count = 0;
        }
        float alpha = p.fade_alpha;                                                       // :953-961
        if (s->fade_count < p.fade_duration) {
            alpha = alpha * (static_cast<float>(s->fade_count) / p.fade_duration);
            s->fade_count++;
        }
        if (rc == VS_OK) rc = launch_fade_blend(s->d_fade, s->d_tmp, (nb + 3) & ~(size_t)3, alpha, 1.0f - alpha, st);
        {
            StageScope t(s, VS_STAGE_WARP, st);
            if (rc == VS_OK) rc = launch_warp_affine(s->d_tmp, prow, 0, bw, bh, d_out, out_stride, 0, bw, bh, s->cn, s->d_Minv, 1, nullptr, st);
        }
        if (rc == VS_OK) rc = launch_fade_update(s->d_fade, d_out, out_stride, (int)prow, bh, st);
    } else if (p.border_size > 0 && !p.crop_n_zoom) {                                 // :981-990
        const int b = p.border_size, bw = s->w + 2 * b, bh = s->h + 2 * b;
        rc = launch_make_border(frame, s->src_pitch, s->w, s->h, s->cn, s->d_tmp, (size_t)bw * s->cn, b, p.border_type, st);
        StageScope t(s, VS_STAGE_WARP, st);
        if (rc == VS_OK)
            rc = launch_warp_affine(s->d_tmp, (size_t)bw * s->cn, 0, bw, bh, d_out, out_stride, 0, bw, bh, s->cn, s->d_Minv, 1, nullptr, st);
    } else if (p.crop_n_zoom && p.border_size > 0 && s->w - 2 * p.border_size > 0 && s->h - 2 * p.border_size > 0) {  // :1108-1124
        const int b = p.border_size;
        StageScope t(s, VS_STAGE_WARP, st);
        rc = launch_warp_affine(frame, s->src_pitch, 0, s->w, s->h, s->d_tmp, s->row_bytes, 0, s->w, s->h, s->cn, s->d_Minv, 1, nullptr, st);
        if (rc == VS_OK)
            rc = launch_resize_linear(s->d_tmp + ((size_t)b * s->w + b) * s->cn, s->row_bytes, s->w - 2 * b, s->h - 2 * b,
                                      s->cn, d_out, out_stride, s->orig_w, s->orig_h, st);
    } else {                                                                          // :1056-1060
        StageScope t(s, VS_STAGE_WARP, st);
        rc = launch_warp_affine(frame, s->src_pitch, 0, s->w, s->h, d_out, out_stride, 0, s->w, s->h, s->cn, s->d_Minv, 1, nullptr, st);
    }
    // the slot may be overwritten once this warp has read it
    if (slot >= 0) {
        if (hipEventRecord(s->ev_slot[slot], st) == hipSuccess) s->slot_valid[slot] = true;
        s->free_slots.push_back(slot);
    }
    if (rc != VS_OK) { s->err = get_last_error(); return rc; }
    s->counters.frames_out++;
    return VS_OK;
}

// ---- batch mode --------------------------------------------------------------------------------------
// Frame f (>= 1) enters: its gray image and pyramid are built at once on `pre` (ring slot f % npyr); the
// analysis is postponed until `batch` frames wait.  All host-side decisions of generateTransform that do not
// depend on data (detection cadence :696, keypoint buffer hand-over, warm-up :383-387) are taken here, in
// push order, so the device work is the same as in the per-frame path.
int batch_enqueue(vs_stab* s, const uint8_t* frame, int slot, int f, uint8_t* d_out, size_t out_stride, int* produced) {
    const vs_params_c& p = s->p;
    const int N = s->npyr, c = f % N, pv = (f - 1) % N;
    // one output pitch per batched warp launch: a change of pitch closes the batch that is being collected
    for (const vs_stab::BFrame& q : s->bq)
        if (q.out_due && q.out_stride != out_stride) { S_TRY(s, drain_batch(s)); break; }
    vs_stab::BFrame b;
    memset(&b, 0, sizeof b);
    b.f = f; b.c = c; b.pv = pv;
    b.frame = frame; b.prev_small = s->prev_small;
    s->prev_small = false;
    b.lk_buf = s->kp_cur; b.lk_cap = s->pts_cap[s->kp_cur];
    b.have_prev_gray = s->have_prev_gray ? 1 : 0;
    b.detect = (++s->detect_counter % 2) == 0;
    if (b.detect) {
        const int q = s->kp_next;
        s->kp_next = (s->kp_next + 1) % (int)s->d_pts.size();
        b.det_buf = q;
        s->pts_cap[q] = std::min(p.max_corners, 200);
        s->kp_cur = q;
        s->counters.detections++;
    }
    s->n_transforms++;
    s->have_prev_gray = true;
    s->last_gray_buf = c;
    s->q_slot.push_back(slot); s->q_idx.push_back(f); s->q_ptr.push_back(frame);     // :376-377
    const int R = effective_radius(s->host_radius);                                  // :383
    if ((int)s->q_idx.size() >= R) {                                                 // :384-389
        b.out_due = true;
        b.out_slot = s->q_slot.front(); b.out_idx = s->q_idx.front(); b.out_frame = s->q_ptr.front();
        s->q_slot.pop_front(); s->q_idx.pop_front(); s->q_ptr.pop_front();
        b.d_out = d_out; b.out_stride = out_stride;
        out_size(s, s->w, s->h, &s->last_out_w, &s->last_out_h);
        s->counters.frames_out++;
        *produced = 1;
    }
    s->bq.push_back(b);
    // a full batch runs (a vs_batch runs the batches of its streams together: vs_batch_push_dev decides)
    if (s->group == s->own && (int)s->bq.size() >= s->batch) {
        const int rc = group_run(s->group);
        if (rc != VS_OK) { s->err = get_last_error(); return rc; }
    }
    return VS_OK;
}

int check_params(const vs_params_c* p, std::string* why) {
    if (!p || p->struct_size != (int32_t)sizeof(vs_params_c)) { *why = "params: struct_size mismatch"; return VS_ERR_INVALID_ARG; }
    if (p->enable_virtual_canvas && !p->crop_n_zoom) {
        // temporalBufferSize < 0 never trims in the reference (size_t compare, Stabilizer.cpp:2159); a blend weight outside
        // [0, 1] leaves the range of the uchar cast of :2392-2396
        if (p->temporal_buffer_size < 0 || p->temporal_buffer_size > 256) { *why = "temporalBufferSize must be in [0,256]"; return VS_ERR_INVALID_ARG; }
        if (!(p->canvas_blend_weight >= 0.0f && p->canvas_blend_weight <= 1.0f)) { *why = "canvasBlendWeight must be in [0,1]"; return VS_ERR_UNSUPPORTED; }
        if (!(p->canvas_scale_factor > 0.0f && p->canvas_scale_factor <= 16.0f) || !(p->min_canvas_scale > 0.0f) || !(p->max_canvas_scale <= 16.0f)) {
            *why = "canvas scale factors must be in (0,16]"; return VS_ERR_INVALID_ARG;
        }
    }
    if (p->max_corners < 1 || p->max_corners > 4096) { *why = "maxCorners must be in [1,4096]"; return VS_ERR_INVALID_ARG; }
    if (p->block_size < 1 || p->block_size > 7) { *why = "blockSize must be in [1,7]"; return VS_ERR_INVALID_ARG; }
    if (p->lk_win_size < 3 || p->lk_win_size > 31 || p->lk_max_level < 0 || p->lk_max_level > 7) { *why = "LK window/levels out of range"; return VS_ERR_INVALID_ARG; }
    if (p->ransac_max_iters < 1 || p->ransac_max_iters > 4096) { *why = "ransac_max_iters out of range"; return VS_ERR_INVALID_ARG; }
    if (p->border_size < 0 || p->border_type < 0 || p->border_type > VS_BORDER_FADE) { *why = "border parameters out of range"; return VS_ERR_INVALID_ARG; }
    if (p->smoothing_method < 0 || p->smoothing_method > VS_SMOOTH_KALMAN) { *why = "smoothing_method out of range"; return VS_ERR_INVALID_ARG; }
    if (p->smoothing_method == VS_SMOOTH_GAUSSIAN) {
        // gaussianFilterConvolve's kernel has max(3, ceil(6 sigma)) taps (Stabilizer.cpp:1368-1370); the device evaluates one
        // tap per lane of a wave, at most GAUSS_MAX: a wider kernel would be cut short without a word
        const float sigma = (float)p->gaussian_sigma;
        if (!(sigma > 0.0f)) { *why = "gaussianSigma must be positive"; return VS_ERR_INVALID_ARG; }
        int ks = std::max(3, (int)std::ceil(6 * sigma));
        if (ks % 2 == 0) ks++;
        if (!(sigma <= 1e6f) || ks > GAUSS_MAX) { *why = "gaussianSigma above 10.5 (a smoothing kernel of more than 63 taps) is outside the accelerated path"; return VS_ERR_UNSUPPORTED; }
    }
    return VS_OK;
}

// Shared body of stabilize(): the frame is already on its way into ring slot `slot` (on `pre`).
int push_common(vs_stab* s, int slot, const uint8_t* zc_frame, uint8_t* d_out, size_t out_stride, int* produced, bool may_defer) {
    const vs_params_c& p = s->p;
    const uint8_t* frame = slot >= 0 ? s->d_ring + (size_t)slot * s->frame_bytes : zc_frame;
    *produced = 0;
    s->counters.frames_in++;
    if (p.crop_n_zoom && s->orig_w == 0) { s->orig_w = s->w; s->orig_h = s->h; }   // :267-269
    if (s->first) {                                                                  // :271-368
        S_TRY(s, launch_resize_gray(frame, s->src_pitch, s->w, s->h, s->fmt, s->d_first_gray, 480, 480, 270, s->st_pre));  // :304-305
        S_HIP(s, hipEventRecord(s->ev_first, s->st_pre));
        S_HIP(s, hipStreamWaitEvent(s->st_det, s->ev_first, 0));
        S_TRY(s, launch_gftt(s->d_first_gray, 480, 480, 270, p.max_corners, p.quality_level, p.min_distance,
                             p.block_size, s->gw, s->d_pts[0], s->d_npts[0], s->st_det));   // :354-358
        S_HIP(s, hipEventRecord(s->ev_det[0], s->st_det));
        s->pts_event[0] = s->ev_det[0];
        s->pts_pending[0] = true;
        s->pp = 0; s->pts_cap[0] = p.max_corners;
        s->last_detected = true; s->last_detect_pp = 0;
        s->dbg_det_pts = s->d_pts[0]; s->dbg_det_n = s->d_npts[0];
        s->counters.detections++;
        s->prev_small = true; s->have_prev_gray = true;
        s->q_slot.push_back(slot); s->q_idx.push_back(0); s->q_ptr.push_back(frame);
        s->first = false; s->next_index = 1;
        return VS_OK;
    }
    if (s->batch_active) {
        S_TRY(s, batch_enqueue(s, frame, slot, s->next_index, d_out, out_stride, produced));
        s->next_index++;
        if (!may_defer) { S_TRY(s, drain_batch(s)); S_TRY(s, flush_warps(s)); }
        return VS_OK;
    }
    s->q_slot.push_back(slot); s->q_idx.push_back(s->next_index); s->q_ptr.push_back(frame);   // :376-377
    S_TRY(s, generate_transform(s, frame, s->next_index));                          // :380
    if (p.adaptive_smoothing) {
        // params_.smoothingRadius is data dependent in this mode (:1482-1486) and
        // moves the warm-up threshold (:383): read it back (synchronises).
        int r = 0;
        S_HIP(s, hipMemcpyAsync(&r, &s->d_traj->smoothing_radius, sizeof r, hipMemcpyDeviceToHost, s->st));
        S_HIP(s, hipStreamSynchronize(s->st));
        s->host_radius = r;
    }
    const int R = effective_radius(s->host_radius);                                 // :383
    if ((int)s->q_idx.size() < R) { s->next_index++; return VS_OK; }                // :384-387
    S_TRY(s, apply_next(s, d_out, out_stride, may_defer));                          // :389
    s->next_index++;
    *produced = 1;
    return VS_OK;
}

int take_slot(vs_stab* s, int* slot) {
    if (!s->d_ring) S_HIP(s, hipMalloc((void**)&s->d_ring, s->frame_bytes * s->ring_frames));
    if (s->free_slots.empty()) return fail(s, VS_ERR_CAPACITY, "frame ring exhausted");
    *slot = s->free_slots.front();
    s->free_slots.pop_front();
    return VS_OK;
}

int prepare(vs_stab* s, int w, int h, int fmt, size_t stride) {
    if (w <= 0 || h <= 0 || (fmt != VS_FMT_BGR8 && fmt != VS_FMT_NV12 && fmt != VS_FMT_GRAY8))
        return fail(s, VS_ERR_INVALID_ARG, "push: bad geometry/format");
    if (fmt == VS_FMT_NV12 && ((w & 1) || (h & 1))) return fail(s, VS_ERR_INVALID_ARG, "NV12 needs even w,h");
    if (fmt != VS_FMT_BGR8 && s->p.border_size > 0) return fail(s, VS_ERR_UNSUPPORTED, "border/crop modes need BGR8 frames");
    const int cn = fmt == VS_FMT_BGR8 ? 3 : 1;
    if (stride < (size_t)w * cn) return fail(s, VS_ERR_INVALID_ARG, "push: stride < row bytes");
    S_HIP(s, hipSetDevice(s->device));
    if (!s->allocated) return allocate(s, w, h, fmt);
    if (w != s->w || h != s->h || fmt != s->fmt) return fail(s, VS_ERR_SIZE_CHANGED, "frame geometry changed; call vs_stab_clean()");
    return VS_OK;
}

void destroy_events(vs_stab* s) {
    auto kill = [](hipEvent_t& e) { if (e) { (void)hipEventDestroy(e); e = nullptr; } };
    for (auto& e : s->ev_gray) kill(e);
    for (auto& e : s->ev_pre) kill(e);
    for (auto& e : s->ev_lk) kill(e);
    for (auto& e : s->ev_det) kill(e);
    for (auto& e : s->ev_slot) kill(e);
    kill(s->ev_first); kill(s->ev_hold);
    kill(s->ev_emit); kill(s->ev_warp[0]); kill(s->ev_warp[1]);
}

int create_events(vs_stab* s) {
    auto mk = [&](hipEvent_t& e) { return hipEventCreateWithFlags(&e, hipEventDisableTiming); };
    for (auto& e : s->ev_gray) S_HIP(s, mk(e));
    for (auto& e : s->ev_pre) S_HIP(s, mk(e));
    for (auto& e : s->ev_lk) S_HIP(s, mk(e));
    for (auto& e : s->ev_det) S_HIP(s, mk(e));
    for (auto& e : s->ev_slot) S_HIP(s, mk(e));
    S_HIP(s, mk(s->ev_first)); S_HIP(s, mk(s->ev_hold));
    S_HIP(s, mk(s->ev_emit)); S_HIP(s, mk(s->ev_warp[0])); S_HIP(s, mk(s->ev_warp[1]));
    return VS_OK;
}

}  // namespace

// All instances of a process on one device share ONE set of four HIP streams.  Every instance's launches are already wide in batch mode, and the runtime maps HIP streams onto a
// handful of hardware queues: with four private streams per instance, 2 instances ran at 0.9x and 8 instances
// at 0.15x of ONE instance's total throughput; on shared streams the instances simply take turns.
namespace {
struct StreamPool {
    hipStream_t st = nullptr, pre = nullptr, det = nullptr, warp = nullptr;
    // (the batched warps of every schedule on the device run on `pre`, in the order they were issued: instances sharing the pool are
    //  driven from ONE host thread - INTEGRATION.md)
    int refs = 0;
};
std::mutex g_pool_mutex;
std::map<int, StreamPool> g_pools;

hipError_t make_streams(hipStream_t* st, hipStream_t* pre, hipStream_t* det, hipStream_t* warp) {
    hipError_t e = hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(pre, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(det, hipStreamNonBlocking);
    if (e == hipSuccess) {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        e = hipStreamCreateWithPriority(warp, hipStreamNonBlocking, greatest);
    }
    return e;
}

hipError_t acquire_streams(vs_stab* s) {
    std::lock_guard<std::mutex> g(g_pool_mutex);
    StreamPool& p = g_pools[s->device];
    if (p.refs == 0) {
        hipError_t e = make_streams(&p.st, &p.pre, &p.det, &p.warp);
        if (e != hipSuccess) return e;
    }
    p.refs++;
    s->st = p.st; s->st_pre = p.pre; s->st_det = p.det; s->st_warp = p.warp;
    s->shared_streams = true;
    return hipSuccess;
}

void release_streams(vs_stab* s) {
    if (!s->shared_streams) return;
    std::lock_guard<std::mutex> g(g_pool_mutex);
    StreamPool& p = g_pools[s->device];
    if (--p.refs == 0) {
        (void)hipStreamDestroy(p.st); (void)hipStreamDestroy(p.pre); (void)hipStreamDestroy(p.det); (void)hipStreamDestroy(p.warp);
        p.st = p.pre = p.det = p.warp = nullptr; p.refs = 0;
    }
}
}  // namespace

extern "C" {

int vs_stab_create(const vs_params_c* params, int device, vs_stab** out) {
    if (!out) return VS_ERR_INVALID_ARG;
    *out = nullptr;
    std::string why;
    int rc = check_params(params, &why);
    if (rc != VS_OK) { set_last_error(why); return rc; }
    VS_TRY(ensure_device());
    int ndev = 0;
    VS_HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) { set_last_error("vs_stab_create: bad device index"); return VS_ERR_INVALID_ARG; }
    VS_HIP_TRY(hipSetDevice(device));
    vs_stab* s = new (std::nothrow) vs_stab();
    if (!s) return VS_ERR_HIP;
    s->p = *params;
    s->device = device;
    s->host_radius = params->smoothing_radius;
    if (const char* e = lab_env("VS_STAB_DEBUG_DELAY_US")) s->dbg_delay_us = std::max(0, std::min(std::atoi(e), 20000));
    memset(&s->counters, 0, sizeof s->counters);
    fill_traj_params(s);
    hipError_t e = acquire_streams(s);
    if (e != hipSuccess || create_events(s) != VS_OK) {
        set_last_error(e != hipSuccess ? hipGetErrorString(e) : s->err);
        vs_stab_destroy(s);
        return VS_ERR_HIP;
    }
    *out = s;
    return VS_OK;
}

void vs_stab_destroy(vs_stab* s) {
    if (!s || s->member) return;        // (a stream of a vs_batch goes with its group: vs_batch_destroy)
    (void)hipSetDevice(s->device);
    // the destructor may race in-flight work (vsg.cpp:1374): drain all three streams first
    if (s->st_pre) (void)hipStreamSynchronize(s->st_pre);
    if (s->st_det) (void)hipStreamSynchronize(s->st_det);
    if (s->st) (void)hipStreamSynchronize(s->st);
    if (s->st_warp) (void)hipStreamSynchronize(s->st_warp);
    free_all(s);
    if (s->d_fade) (void)hipFree(s->d_fade);
    if (s->d_ct) (void)hipFree(s->d_ct);
    canvas_delete(s->canvas);
    for (auto& pe : s->pending) { (void)hipEventDestroy(pe.a); (void)hipEventDestroy(pe.b); }
    for (auto e : s->ev_pool) (void)hipEventDestroy(e);
    destroy_events(s);
    if (s->st) release_streams(s);
    delete s;
}

int vs_stab_clean(vs_stab* s) {   // Stabilizer.cpp:221-256
    if (!s) return VS_ERR_INVALID_ARG;
    if (s->member && !s->group_call) return fail(s, VS_ERR_INVALID_ARG, "this stream belongs to a vs_batch: drive it through vs_batch_* (its getters remain available)");
    S_TRY(s, sync_all(s));
    free_all(s);
    s->q_slot.clear(); s->q_idx.clear(); s->q_ptr.clear();
    s->hold_valid = false; s->hold_cur = 0;
    s->first = true; s->next_index = 0; s->w = s->h = 0; s->orig_w = s->orig_h = 0; s->n_transforms = 0;
    s->have_prev_gray = false; s->prev_small = false; s->pp = 0;
    s->host_radius = s->p.smoothing_radius;
    return VS_OK;
}

int vs_stab_out_size(const vs_stab* s, int w, int h, int* out_w, int* out_h) {
    if (!s || !out_w || !out_h) return VS_ERR_INVALID_ARG;
    out_size(s, w, h, out_w, out_h);
    return VS_OK;
}

int vs_stab_push_dev(vs_stab* s, const void* d_data, int w, int h, size_t stride, int fmt, void* d_out,
                     size_t out_stride, int* produced) {
    if (!s || !produced) return VS_ERR_INVALID_ARG;
    if (s->member && !s->group_call) return fail(s, VS_ERR_INVALID_ARG, "this stream belongs to a vs_batch: drive it through vs_batch_* (its getters remain available)");
    *produced = 0;
    if (!d_data) return VS_OK;   // empty frame -> empty result (Stabilizer.cpp:263-265)
    int rc = prepare(s, w, h, fmt, stride);
    if (rc != VS_OK) return rc;
    if (s->zero_copy) {
        // the frame is read where it is: it must stay valid and unchanged until its own result has been produced
        // one pitch for all frames in flight (the batched launches take it once): it may change when nothing is queued
        if (stride != s->src_pitch) {
            if (!s->q_slot.empty() || !s->bq.empty() || !s->pend.empty() || group_holds_warps(s->group))
                return fail(s, VS_ERR_INVALID_ARG, "zero-copy mode: the row pitch may only change while no frame is queued");
            s->src_pitch = stride;
        }
        return push_common(s, -1, (const uint8_t*)d_data, (uint8_t*)d_out, out_stride, produced, true);
    }
    int slot;
    S_TRY(s, take_slot(s, &slot));
    S_TRY(s, enqueue_copy_in(s, slot, d_data, stride, hipMemcpyDeviceToDevice));
    return push_common(s, slot, nullptr, (uint8_t*)d_out, out_stride, produced, true);
}

// n consecutive pushes of one geometry: result j of them (in the order they become due) goes to d_outs[j]; *produced = how many
// became due.  What n calls of vs_stab_push_dev do, without n trips through a binding.
int vs_stab_push_dev_n(vs_stab* s, const void* const* d_frames, int n, int w, int h, size_t stride, int fmt, void* const* d_outs, size_t out_stride,
                       int* produced) {
    if (!s || !d_frames || !d_outs || !produced || n < 0) return VS_ERR_INVALID_ARG;
    int k = 0;
    for (int i = 0; i < n; i++) {
        int now = 0;
        const int rc = vs_stab_push_dev(s, d_frames[i], w, h, stride, fmt, d_outs[k], out_stride, &now);
        if (rc != VS_OK) { *produced = k; return rc; }
        k += now;
    }
    *produced = k;
    return VS_OK;
}

static int flush_dev_impl(vs_stab* s, void* d_out, size_t out_stride, int* produced, bool may_defer_flush) {
    if (!s || !produced) return VS_ERR_INVALID_ARG;
    *produced = 0;
    if (!s->allocated || s->q_slot.empty()) return VS_OK;
    S_HIP(s, hipSetDevice(s->device));
    S_TRY(s, drain_batch(s));
    S_TRY(s, apply_next(s, (uint8_t*)d_out, out_stride, may_defer_flush));
    *produced = 1;
    return VS_OK;
}

int vs_stab_flush_dev(vs_stab* s, void* d_out, size_t out_stride, int* produced) {   // Stabilizer.cpp:394-400
    if (!s || !produced) return VS_ERR_INVALID_ARG;
    if (s->member && !s->group_call) return fail(s, VS_ERR_INVALID_ARG, "this stream belongs to a vs_batch: drive it through vs_batch_* (its getters remain available)");
    return flush_dev_impl(s, d_out, out_stride, produced, true);
}

// Host pipeline: a call returns the frame that the call before it computed.  Its download runs on the copy stream while
// this call's frame is uploaded (PCIe is full duplex) and the call returns as soon as both transfers are done: the
// analysis and the warp of this call's frame go on behind the caller's back and are picked up by the next call.
static int push_host_pipelined(vs_stab* s, const uint8_t* data, int w, int h, size_t stride, int fmt, uint8_t* out,
                               size_t out_stride, int* produced) {
    int ow, oh;
    out_size(s, w, h, &ow, &oh);
    const size_t orow = (size_t)ow * s->cn;
    const int orows = fmt == VS_FMT_NV12 ? oh * 3 / 2 : oh;
    const bool have_prev = s->hold_valid;
    // (as in the synchronous call: the buffer is only looked at when a frame will be delivered into it; a held frame is
    // always a full-size one - pass-through frames only come out of vs_stab_flush's synchronous part)
    if (have_prev && (!out || out_stride < orow)) return fail(s, VS_ERR_INVALID_ARG, "push: output buffer/stride too small");
    for (auto& hld : s->d_hold)
        if (!hld) S_HIP(s, hipMalloc((void**)&hld, s->out_bytes));
    // A copy to or from PAGEABLE memory (the frames of a cv::Mat) keeps its caller inside hipMemcpy for the whole transfer -
    // about 0.2 ms per direction at 1080p, through the runtime's bounce buffers -, so download and upload issued from this
    // thread run one after the other: 2 520 frames/s against 4 730 with page-locked frames.  With a pageable output buffer the
    // download is therefore issued by the instance's helper thread, beside this thread's upload (VS_STAB_HOST_HELPER=0: from
    // this thread, as before).  (Staging both directions through page-locked buffers of our own with a pool of copy threads
    // was measured first and lost to the runtime's path: scratch/README.md.)
    static const bool use_helper = [] { const char* e = lab_env("VS_STAB_HOST_HELPER"); return !(e && e[0] == '0'); }();
    struct Join {           // the helper's job refers to the caller's buffer: no way out of this call without waiting for it
        HostHelper* h = nullptr;
        ~Join() { if (h) (void)h->wait(); }
        int wait() { HostHelper* t = h; h = nullptr; return t ? t->wait() : 0; }
    } join;
    const int held_w = s->hold_w, held_h = s->hold_h;
    if (have_prev) {       // the held result: on its way while this call's frame comes in
        S_HIP(s, hipStreamWaitEvent(s->st_warp, s->ev_hold, 0));
        const uint8_t* d_src = s->d_hold[s->hold_cur ^ 1];

        bool helped = false;
        if (use_helper && !host_ptr_page_locked(out)) {
            try {           // (no exception leaves the C ABI: without a helper thread the download goes out from this one)
                if (!s->helper) s->helper.reset(new HostHelper);
                const int dev = s->device;
                hipStream_t stw = s->st_warp;
                s->helper->start([=]() -> int {
                    hipError_t e = hipSetDevice(dev);
                    if (e == hipSuccess) e = hipMemcpy2DAsync(out, out_stride, d_src, orow, orow, orows, hipMemcpyDeviceToHost, stw);
                    if (e == hipSuccess) e = hipStreamSynchronize(stw);
                    return (int)e;
                });
                join.h = s->helper.get();
                helped = true;
            } catch (...) {
                s->helper.reset();
            }
        }
        if (!helped) {
            S_HIP(s, hipMemcpy2DAsync(out, out_stride, d_src, orow, orow, orows, hipMemcpyDeviceToHost, s->st_warp));
        }
    }
    int slot;
    S_TRY(s, take_slot(s, &slot));
    S_TRY(s, enqueue_copy_in(s, slot, data, stride, hipMemcpyHostToDevice));
    int now = 0;
    int rc = push_common(s, slot, nullptr, s->d_hold[s->hold_cur], orow, &now, false);
    S_HIP(s, hipStreamSynchronize(s->st_pre));        // the caller's frame has been consumed
    if (join.h) S_HIP(s, (hipError_t)join.wait());
    else if (have_prev) S_HIP(s, hipStreamSynchronize(s->st_warp));
    if (rc != VS_OK) return rc;
    s->hold_valid = now != 0;
    if (now) {
        S_HIP(s, hipEventRecord(s->ev_hold, s->st));
        s->hold_cur ^= 1;
        s->hold_w = s->last_out_w; s->hold_h = s->last_out_h;
    }
    *produced = have_prev ? 1 : 0;
    if (have_prev) { s->last_out_w = held_w; s->last_out_h = held_h; }     // vs_stab_last_out_dims: the frame that was handed out
    return VS_OK;
}

int vs_stab_push(vs_stab* s, const uint8_t* data, int w, int h, size_t stride, int fmt, uint8_t* out,
                 size_t out_stride, int* produced) {
    if (!s || !produced) return VS_ERR_INVALID_ARG;
    if (s->member && !s->group_call) return fail(s, VS_ERR_INVALID_ARG, "this stream belongs to a vs_batch: drive it through vs_batch_* (its getters remain available)");
    *produced = 0;
    if (!data) return VS_OK;
    int rc = prepare(s, w, h, fmt, stride);
    if (rc != VS_OK) return rc;
    if (s->zero_copy && (s->src_pitch != s->row_bytes || (s->fmt == VS_FMT_NV12 && s->in_uv_off)))
        return fail(s, VS_ERR_INVALID_ARG, "push: host frames cannot join a queue of pitched zero-copy surfaces");
    if (s->host_pipe && !s->batch_active) return push_host_pipelined(s, data, w, h, stride, fmt, out, out_stride, produced);
    int ow, oh;
    out_size(s, w, h, &ow, &oh);
    const size_t orow = (size_t)ow * s->cn;
    // (the output buffer is checked before the frame is consumed: a bad call loses nothing)
    const int R = effective_radius(s->host_radius);
    if ((!out || out_stride < orow) && !s->first && (int)s->q_idx.size() + 1 >= R)
        return fail(s, VS_ERR_INVALID_ARG, "push: output buffer/stride too small");
    int slot;
    S_TRY(s, take_slot(s, &slot));
    rc = enqueue_copy_in(s, slot, data, stride, hipMemcpyHostToDevice);
    if (rc == VS_OK) rc = drain_batch(s);
    if (rc == VS_OK) rc = flush_warps(s);
    if (rc == VS_OK) rc = push_common(s, slot, nullptr, s->d_out, orow, produced, false);
    if (rc != VS_OK) {
        (void)hipStreamSynchronize(s->st_pre);       // the upload from the caller's buffer may still be in flight
        return rc;
    }
    if (*produced) {
        if (!out || out_stride < orow) {
            (void)hipStreamSynchronize(s->st_pre);
            return fail(s, VS_ERR_INVALID_ARG, "push: output buffer/stride too small");
        }
        const int orows = fmt == VS_FMT_NV12 ? oh * 3 / 2 : oh;
        S_HIP(s, hipStreamSynchronize(s->st_warp));   // (per-frame pipeline: the warp ran on the warp stream)
        if (s->batch_active) S_HIP(s, hipStreamSynchronize(s->st_pre));     // batch mode: the warps run on `pre` (group_launch_ready)
        S_HIP(s, hipMemcpy2DAsync(out, out_stride, s->d_out, orow, orow, orows, hipMemcpyDeviceToHost, s->st));
    }
    // the caller's frame must be consumed and its result delivered before returning
    S_HIP(s, hipStreamSynchronize(s->st_pre));
    S_HIP(s, hipStreamSynchronize(s->st));
    return VS_OK;
}

int vs_stab_flush(vs_stab* s, uint8_t* out, size_t out_stride, int* produced) {
    if (!s || !produced) return VS_ERR_INVALID_ARG;
    if (s->member && !s->group_call) return fail(s, VS_ERR_INVALID_ARG, "this stream belongs to a vs_batch: drive it through vs_batch_* (its getters remain available)");
    *produced = 0;
    if (!s->allocated) return VS_OK;
    if (s->hold_valid) {      // host pipeline: the frame the last push computed
        const size_t orow = (size_t)s->hold_w * s->cn;
        if (!out || out_stride < orow) return fail(s, VS_ERR_INVALID_ARG, "flush: output buffer/stride too small");
        const int orows = s->fmt == VS_FMT_NV12 ? s->hold_h * 3 / 2 : s->hold_h;
        S_HIP(s, hipStreamWaitEvent(s->st_warp, s->ev_hold, 0));
        S_HIP(s, hipMemcpy2DAsync(out, out_stride, s->d_hold[s->hold_cur ^ 1], orow, orow, orows, hipMemcpyDeviceToHost, s->st_warp));
        S_HIP(s, hipStreamSynchronize(s->st_warp));
        s->hold_valid = false;
        s->last_out_w = s->hold_w; s->last_out_h = s->hold_h;
        *produced = 1;
        return VS_OK;
    }
    if (s->q_slot.empty()) return VS_OK;
    int ow, oh;
    out_size(s, s->w, s->h, &ow, &oh);
    const size_t orow = (size_t)ow * s->cn;
    if (!out || out_stride < orow) return fail(s, VS_ERR_INVALID_ARG, "flush: output buffer/stride too small");
    S_TRY(s, drain_batch(s));
    S_TRY(s, flush_warps(s));
    int rc = flush_dev_impl(s, s->d_out, orow, produced, false);
    if (rc != VS_OK) return rc;
    const int orows = s->fmt == VS_FMT_NV12 ? oh * 3 / 2 : oh;
    S_HIP(s, hipStreamSynchronize(s->st_warp));
    if (s->batch_active) S_HIP(s, hipStreamSynchronize(s->st_pre));         // batch mode: the warps run on `pre`
    S_HIP(s, hipMemcpy2DAsync(out, out_stride, s->d_out, orow, orow, orows, hipMemcpyDeviceToHost, s->st));
    S_HIP(s, hipStreamSynchronize(s->st));
    return VS_OK;
}

// Host pipeline for vs_stab_push / vs_stab_flush (see push_host_pipelined): one more call of latency, the transfers of
// consecutive calls overlap each other and the device work.  To be chosen while no frame is queued.
int vs_stab_set_host_pipeline(vs_stab* s, int enable) {
    if (!s) return VS_ERR_INVALID_ARG;
    if (s->member && !s->group_call) return fail(s, VS_ERR_INVALID_ARG, "this stream belongs to a vs_batch: drive it through vs_batch_* (its getters remain available)");
    if (!s->q_slot.empty() || s->hold_valid) return fail(s, VS_ERR_INVALID_ARG, "vs_stab_set_host_pipeline: the frame queue must be empty");
    s->host_pipe = enable != 0;
    return VS_OK;
}

// Deferred output for the device entry points: up to `frames` consecutive results are warped by one
// launch.  A result is complete after vs_stab_sync(); every push must then be given its own d_out.
int vs_stab_set_warp_batch(vs_stab* s, int frames) {
    if (!s || frames < 1 || frames > WARP_BATCH_MAX) return VS_ERR_INVALID_ARG;
    if (s->member && !s->group_call) return fail(s, VS_ERR_INVALID_ARG, "this stream belongs to a vs_batch: drive it through vs_batch_* (its getters remain available)");
    if (s->allocated) { S_HIP(s, hipSetDevice(s->device)); S_TRY(s, flush_warps(s)); }
    s->warp_batch = frames;
    return VS_OK;
}

// Batch mode for the device entry points: the analysis of `frames` consecutive pushes (feature detection,
// tracking, hypothesis scoring) runs as one launch per stage, and their warps as one launch (implies
// vs_stab_set_warp_batch(frames)).  To be chosen before the first frame (or after vs_stab_clean).
int vs_stab_set_batch(vs_stab* s, int frames) {
    if (!s || frames < 1 || frames > BATCH_MAX) return VS_ERR_INVALID_ARG;
    if (s->member && !s->group_call) return fail(s, VS_ERR_INVALID_ARG, "this stream belongs to a vs_batch: drive it through vs_batch_* (its getters remain available)");
    if (s->allocated) return fail(s, VS_ERR_INVALID_ARG, "vs_stab_set_batch: call before the first frame or after vs_stab_clean");
    s->batch = frames;
    if (frames > 1) s->warp_batch = std::min(frames, WARP_BATCH_MAX);
    return VS_OK;
}

// Zero-copy input for vs_stab_push_dev: the frame is not copied into the instance's queue but read where the
// caller put it (a decoder surface pool, a resident clip).  It must stay valid and unchanged until the result
// of the SAME push count has been produced, i.e. for clamp(smoothingRadius,5,35) further pushes plus twice
// the batch depth and a vs_stab_sync, or until vs_stab_flush_dev has drained the queue.  Tightly packed frames.
int vs_stab_set_zero_copy(vs_stab* s, int enable) {
    if (!s) return VS_ERR_INVALID_ARG;
    if (s->member && !s->group_call) return fail(s, VS_ERR_INVALID_ARG, "this stream belongs to a vs_batch: drive it through vs_batch_* (its getters remain available)");
    if (!s->q_slot.empty()) return fail(s, VS_ERR_INVALID_ARG, "vs_stab_set_zero_copy: the frame queue must be empty");
    s->zero_copy = enable != 0;
    s->src_pitch = s->row_bytes;
    return VS_OK;
}

// NV12 surfaces of the device entry points as hardware decoders export them (rocDecode, VA-API): Y and interleaved
// UV plane with a common pitch, the UV plane `uv_offset` bytes behind the Y pointer.  0 = contiguous (h * pitch).
int vs_stab_set_nv12_layout(vs_stab* s, size_t in_uv_offset, size_t out_uv_offset) {
    if (!s) return VS_ERR_INVALID_ARG;
    if (s->member && !s->group_call) return fail(s, VS_ERR_INVALID_ARG, "this stream belongs to a vs_batch: drive it through vs_batch_* (its getters remain available)");
    if (!s->q_slot.empty()) return fail(s, VS_ERR_INVALID_ARG, "vs_stab_set_nv12_layout: the frame queue must be empty");
    s->in_uv_off = in_uv_offset;
    s->out_uv_off = out_uv_offset;
    return VS_OK;
}

int vs_stab_sync(vs_stab* s) {
    if (!s) return VS_ERR_INVALID_ARG;
    return sync_all(s);
}

int vs_stab_get_counters(vs_stab* s, vs_counters* out) {
    if (!s || !out) return VS_ERR_INVALID_ARG;
    *out = s->counters;
    if (s->allocated) {
        S_TRY(s, sync_all(s));
        vs_debug_frame d;
        int32_t c[4] = {0, 0, 0, 0};
        S_HIP(s, hipMemcpy(&d, s->d_dbg, sizeof d, hipMemcpyDeviceToHost));
        S_HIP(s, hipMemcpy(c, s->dbg_gftt_counters, sizeof c, hipMemcpyDeviceToHost));
        out->last_features = d.n_prev;
        out->last_tracked = d.n_valid;
        out->last_inliers = d.n_inliers;
        out->last_candidates = c[0];
        out->gftt_overflow = c[2];
    }
    return VS_OK;
}

int vs_stab_canvas_info(const vs_stab* s, int32_t info[8]) {
    if (!s || !info) return VS_ERR_INVALID_ARG;
    canvas_info(s->canvas, info);
    return VS_OK;
}

int vs_stab_get_debug(vs_stab* s, vs_debug_frame* out) {
    if (!s || !out) return VS_ERR_INVALID_ARG;
    memset(out, 0, sizeof *out);
    out->out_index = -1;
    if (!s->allocated) return VS_OK;
    S_TRY(s, sync_all(s));
    S_HIP(s, hipMemcpy(out, s->d_dbg, sizeof *out, hipMemcpyDeviceToHost));
    out->detected = s->last_detected ? 1 : 0;
    out->n_detected = 0;
    if (s->last_detected) {
        int32_t n = 0;
        S_HIP(s, hipMemcpy(&n, s->dbg_det_n, sizeof n, hipMemcpyDeviceToHost));
        out->n_detected = n;
    }
    if (s->counters.frames_in <= 1) { out->n_prev = 0; out->n_valid = 0; out->out_index = -1; }
    return VS_OK;
}

int vs_stab_get_debug_arrays(vs_stab* s, float* prev_pts, float* curr_pts, uint8_t* status, uint8_t* inliers,
                             float* detected_pts, uint8_t* gray, int* aw, int* ah) {
    if (!s) return VS_ERR_INVALID_ARG;
    vs_debug_frame d;
    int rc = vs_stab_get_debug(s, &d);
    if (rc != VS_OK) return rc;
    if (!s->allocated) return VS_OK;
    const bool first_only = s->counters.frames_in <= 1;
    if (d.n_prev > 0 && !first_only) {
        if (prev_pts) S_HIP(s, hipMemcpy(prev_pts, s->dbg_prev_pts, (size_t)d.n_prev * 8, hipMemcpyDeviceToHost));
        if (curr_pts) S_HIP(s, hipMemcpy(curr_pts, s->dbg_next, (size_t)d.n_prev * 8, hipMemcpyDeviceToHost));
        if (status) S_HIP(s, hipMemcpy(status, s->dbg_status, (size_t)d.n_prev, hipMemcpyDeviceToHost));
    }
    if (d.n_valid > 0 && inliers && !first_only) S_HIP(s, hipMemcpy(inliers, s->dbg_inliers, (size_t)d.n_valid, hipMemcpyDeviceToHost));
    if (d.n_detected > 0 && detected_pts)
        S_HIP(s, hipMemcpy(detected_pts, s->dbg_det_pts, (size_t)d.n_detected * 8, hipMemcpyDeviceToHost));
    if (first_only) {
        if (gray) S_HIP(s, hipMemcpy(gray, s->d_first_gray, (size_t)480 * 270, hipMemcpyDeviceToHost));
        if (aw) *aw = 480;
        if (ah) *ah = 270;
    } else {
        if (gray) S_HIP(s, hipMemcpy(gray, s->pyr[s->last_gray_buf].img[0], (size_t)s->aw * s->ah, hipMemcpyDeviceToHost));
        if (aw) *aw = s->aw;
        if (ah) *ah = s->ah;
    }
    return VS_OK;
}

int vs_stab_last_out_dims(const vs_stab* s, int* w, int* h) {
    if (!s || !w || !h) return VS_ERR_INVALID_ARG;
    *w = s->last_out_w; *h = s->last_out_h;
    return VS_OK;
}

const char* vs_stab_last_error(const vs_stab* s) { return s ? s->err.c_str() : ""; }
void* vs_stab_stream(vs_stab* s) { return s ? (void*)s->st : nullptr; }

int vs_stab_set_profiling(vs_stab* s, int mode) {
    if (!s || mode < 0 || mode > 3) return VS_ERR_INVALID_ARG;
    s->prof_mode = mode;
    return VS_OK;
}

int vs_stab_get_stage_times(vs_stab* s, double* total_ms, int64_t* launches) {
    if (!s || !total_ms || !launches) return VS_ERR_INVALID_ARG;
    for (int i = 0; i < VS_STAGE_COUNT; i++) { total_ms[i] = 0; launches[i] = 0; }
    if (!s->st) return VS_OK;
    S_TRY(s, sync_all(s));
    for (auto& pe : s->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pe.a, pe.b) == hipSuccess && pe.stage >= 0 && pe.stage < VS_STAGE_COUNT) {
            total_ms[pe.stage] += ms;
            launches[pe.stage]++;
        }
        s->ev_pool.push_back(pe.a);
        s->ev_pool.push_back(pe.b);
    }
    s->pending.clear();
    return VS_OK;
}

}  // extern "C"

// ---- the batch schedule -------------------------------------------------------------------------------------------------
// ONE schedule runs every batch: a vs_batch steps the frames that all its streams have queued (BASELINE configs[4]: 64 streams =
// 8 per GPU), and a standalone instance in batch mode owns a private group of one (vs_stab::own).  N instances that each ran
// their own batches took turns on the device's streams: every stage launched once per instance, the chains of one queued behind
// the waits of another (8 instances: 101.7 k frames/s in total against 109.1 k for one, round 2).  A group goes through ONE
// step for its members: the frames that all of them have queued since the last step go into one argument table per stage - the
// tables hold one block per frame anyway, and a block names its frame's buffers, so a launch does not care whose frame it is -,
// the ordered tails run as one launch with a workgroup per stream, and all due warps leave in launches of 32 frames.  The members
// stay ordinary vs_stab instances (queues, pyramid rings, keypoint buffers, trajectory state, counters and debug records of
// their own); host and device tables, the events of the schedule, the inverse maps and coordinate tables of the pending warps
// belong to the group.
//
// One step (group_run), three streams:
//   pre :  gray(n, two parts: re-detecting frames first) -> table uploads -> pyramid level(n) x levels
//   det :  zero -> min_eigen -> nms -> select                 (starts behind the first gray part)
//   main:  [warps of the PREVIOUS step] -> LK(n x 200 waves) -> RANSAC score -> select(n waves) -> ordered tail (1 WG per stream)
//          -> releases (1 WG per push) -> coordinate tables of this step's warps
// `main` waits for `pre` and for the wide launches of `det` of the same step, then issues the warps of the step before: at that
// point the rest of pre/det of this step is done (it overlapped the tracking and tail of the step before) and its tracking has
// not started, and `pre` of the next step waits for these warps - the HBM-bound warp has the GPU to itself (on a stream of its
// own it overlapped the tracker, which holds ~90 KB of LDS per CU: 3 warp workgroups per CU instead of 8).
struct vs_batch {
    int device = 0, S = 0, B = 0, cap = 0;
    bool own = false;                       // the private schedule of one standalone instance
    std::vector<vs_stab*> m;
    std::string err;
    hipStream_t st = nullptr, st_pre = nullptr, st_det = nullptr, st_up = nullptr;      // st_up: the table uploads (the pool's warp stream: idle in batch mode)
    bool allocated = false;
    // Host images of the argument tables of a step, in page-locked memory so that their uploads are asynchronous (from pageable
    // memory hipMemcpyAsync holds the host until the stream gets to the copy, and the host then no longer runs ahead of the
    // GPU): four sets, step k writes set k % 4 once the tail of step k-4 has run.  On the device the tracker / scoring / tail
    // tables exist twice (k & 1): step k+1's are uploaded on `pre` while step k's are still read on `main`.
    uint8_t* h_tables = nullptr;
    size_t h_set_bytes = 0, ho_pairs = 0, ho_lk = 0, ho_rs = 0, ho_tail = 0, ho_gf = 0, ho_seg = 0;
    uint8_t* d_all = nullptr;
    uint8_t *d_lk[2] = {nullptr, nullptr}, *d_rs[2] = {nullptr, nullptr}, *d_tail[2] = {nullptr, nullptr}, *d_seg[2] = {nullptr, nullptr}, *d_gf = nullptr;
    uint8_t* d_tin[2] = {nullptr, nullptr};         // per frame of a step: what the selection leaves for the tail
    ImgPair* d_pairs[2] = {nullptr, nullptr};     // (per table set: a step's pair table goes up with its other tables, one copy)
    size_t up_bytes = 0;                 // bytes of a table set that go to the device: pairs, tracker, scoring, tail, segments
    int pre_rel_step = -1;               // the latest step whose tail `pre` has waited for (through the event of its warps' maps)
    double* d_MinvB[2] = {nullptr, nullptr};        // inverse maps of the due frames of a step, 12 doubles each; two sets
    int32_t* d_tabs[2] = {nullptr, nullptr};        // coordinate tables of those frames (warp_tab.h), tab_ints per frame; two sets
    int tab_ints = 0;                               // one plane's table, or an NV12 surface's block of two
    hipEvent_t ev_bpre = nullptr, ev_bgray = nullptr, ev_bnms = nullptr, ev_bdet[4] = {}, ev_blk[4] = {}, ev_warp[2] = {}, ev_rel[2] = {}, ev_up[2] = {}, ev_go = nullptr;
    bool bdet_valid[4] = {false, false, false, false}, warp_valid[2] = {false, false}, rel_valid[2] = {false, false};
    int last_det_batch = -1, last_warp_set = -1, batch_id = 0, pend_set = 0;
    struct Ready {
        bool valid = false, tabs_built = false;
        int n = 0, set = 0, step = -1;         // step: the group_run that analysed these frames
        size_t stride = 0;
        std::vector<const uint8_t*> srcs;
        std::vector<uint8_t*> dsts;
        std::vector<int> slots, pad_idx;       // pad_idx: which of its owner's scratch frames (border pad / crop-and-zoom)
        std::vector<vs_stab*> owner;
    } ready, next;          // ready: the step whose tails are queued (its warps go out with the next step); next: the step being built
};

namespace {

int gfail(vs_batch* g, int code, const std::string& msg) {
    g->err = msg;
    set_last_error(msg);
    return code;
}
#define G_HIP(g, expr)                                                                        \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) return gfail((g), VS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)
#define G_TRY(g, expr)                                  \
    do {                                                \
        int _r = (expr);                                \
        if (_r != VS_OK) { (g)->err = get_last_error(); return _r; } \
    } while (0)

void group_free(vs_batch* g) {
    if (g->h_tables) (void)hipHostFree(g->h_tables);
    if (g->d_all) (void)hipFree(g->d_all);
    g->h_tables = nullptr; g->d_all = nullptr;
    g->allocated = false;
}

bool group_make_events(vs_batch* g) {
    auto mk = [&](hipEvent_t& e) { return hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess; };
    bool ok = mk(g->ev_bpre) && mk(g->ev_bgray) && mk(g->ev_bnms) && mk(g->ev_warp[0]) && mk(g->ev_warp[1]) && mk(g->ev_rel[0]) && mk(g->ev_rel[1]) && mk(g->ev_up[0]) && mk(g->ev_up[1]) && mk(g->ev_go);
    for (auto& e : g->ev_bdet) ok = ok && mk(e);
    for (auto& e : g->ev_blk) ok = ok && mk(e);
    return ok;
}

// What the members must agree on: everything that shapes a launch (frame and analysis geometry, pitch, input mode, pyramid
// depth, tracking window, hypothesis count, border mode).  Smoothing radius and method, horizon lock, the drone filters'
// settings, corner count and thresholds are per stream: they live in each frame's argument block or in the stream's state.
bool same_launch_shape(const vs_stab* a, const vs_stab* b) {
    const vs_params_c &p = a->p, &q = b->p;
    return a->w == b->w && a->h == b->h && a->fmt == b->fmt && a->src_pitch == b->src_pitch && a->zero_copy == b->zero_copy &&
           a->in_uv_off == b->in_uv_off && a->out_uv_off == b->out_uv_off && a->aw == b->aw && a->ah == b->ah && a->levels == b->levels &&
           p.lk_win_size == q.lk_win_size && p.ransac_max_iters == q.ransac_max_iters && p.border_size == q.border_size &&
           p.crop_n_zoom == q.crop_n_zoom && (p.border_size <= 0 || p.border_type == q.border_type);
}

// Tables and workspaces for cap = S * B frames per step, once the members know their geometry.
int group_allocate(vs_batch* g) {
    const vs_stab* s0 = g->m[0];
    const int cap = g->cap, ngf = g->S * (g->B / 2 + 1);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_gf = take(gftt_item_bytes() * ngf);
    // a table set on the device = its image on the host (pairs, tracker items, scoring items, tail items, segments: ONE upload per step)
    size_t ho = 0;
    auto htake = [&](size_t bytes) { size_t o = ho; ho += (bytes + 255) & ~(size_t)255; return o; };
    g->ho_pairs = htake(sizeof(ImgPair) * cap * (2 + 2 * MAX_PYR));
    g->ho_lk = htake(lk_item_bytes() * cap); g->ho_rs = htake(ransac_item_bytes() * cap);
    g->ho_tail = htake(tail_item_bytes() * cap); g->ho_seg = htake(tail_seg_bytes() * g->S);
    g->up_bytes = ho;
    g->ho_gf = htake(gftt_item_bytes() * ngf);
    g->h_set_bytes = ho;
    const size_t o_set[2] = {take(g->up_bytes), take(g->up_bytes)};
    const size_t o_ti[2] = {take(tail_in_bytes() * cap), take(tail_in_bytes() * cap)};
    const size_t o_minv[2] = {take((size_t)cap * 96), take((size_t)cap * 96)};
    int tow, toh;
    out_size(s0, s0->w, s0->h, &tow, &toh);
    g->tab_ints = s0->fmt == VS_FMT_NV12 ? nv12_tab_ints(s0->w, s0->h) : (int)warp_tabs_ints(std::max(s0->w, tow), std::max(s0->h, toh), 1);
    size_t o_tabs[2];
    for (auto& o : o_tabs) o = take((size_t)g->tab_ints * cap * sizeof(int32_t));
    G_HIP(g, hipMalloc((void**)&g->d_all, off));
    G_HIP(g, hipMemsetAsync(g->d_all, 0, off, g->st));
    uint8_t* b = g->d_all;
    g->d_gf = b + o_gf;
    for (int i = 0; i < 2; i++) {
        uint8_t* ds = b + o_set[i];
        g->d_pairs[i] = (ImgPair*)(ds + g->ho_pairs);
        g->d_lk[i] = ds + g->ho_lk; g->d_rs[i] = ds + g->ho_rs; g->d_tail[i] = ds + g->ho_tail; g->d_seg[i] = ds + g->ho_seg; g->d_tin[i] = b + o_ti[i];
        g->d_MinvB[i] = (double*)(b + o_minv[i]);
    }
    for (int i = 0; i < 2; i++) g->d_tabs[i] = (int32_t*)(b + o_tabs[i]);
    G_HIP(g, hipHostMalloc((void**)&g->h_tables, 4 * ho));
    memset(g->h_tables, 0, 4 * ho);
    G_HIP(g, hipStreamSynchronize(g->st));
    for (vs_batch::Ready* r : {&g->ready, &g->next}) {
        r->srcs.assign(cap, nullptr); r->dsts.assign(cap, nullptr); r->slots.assign(cap, -1); r->pad_idx.assign(cap, 0); r->owner.assign(cap, nullptr);
    }
    g->allocated = true;
    return VS_OK;
}

// Where the frames of a step's warps come from and go to (border pad: the padded scratch frame is the source; crop-and-zoom: the
// scratch frame is the destination, resized into the result afterwards).
struct WarpEnds { const uint8_t* src; uint8_t* dst; };
WarpEnds warp_ends(const vs_stab* o, const uint8_t* frame, uint8_t* d_out, int pad_idx, bool pad, bool crop) {
    if (!pad && !crop) return {frame, d_out};
    uint8_t* scratch = o->d_padB + (size_t)pad_idx * o->pad_frame_bytes;
    return pad ? WarpEnds{scratch, d_out} : WarpEnds{frame, scratch};
}

// The warps of the step whose tails were queued last, 32 frames per launch (`what` = VS_WARP_ONLY / VS_WARP_ALL), or only their
// coordinate tables (VS_WARP_TABLES_ONLY: steps whose release workgroups have not built them - a Kalman stream's releases stay
// inside its tail workgroup).  The frames' tables lie tab_ints apart in d_tabs[set].
int group_ready_launches(vs_batch* g, int what, hipStream_t st) {
    vs_batch::Ready& R = g->ready;
    const vs_stab* s0 = g->m[0];
    const vs_params_c& p = s0->p;
    const int bsz = p.border_size;
    const bool pad = bsz > 0 && !p.crop_n_zoom;                                                       // Stabilizer.cpp:981-990
    const bool crop = bsz > 0 && p.crop_n_zoom && s0->w - 2 * bsz > 0 && s0->h - 2 * bsz > 0;          // :1108-1124
    const int pw = pad ? s0->w + 2 * bsz : s0->w, ph = pad ? s0->h + 2 * bsz : s0->h;
    const size_t prow = (size_t)pw * s0->cn;
    int rc = VS_OK;
    for (int i0 = 0; i0 < R.n && rc == VS_OK; i0 += WARP_BATCH_MAX) {      // the warp kernels take WARP_BATCH_MAX frames per launch
        const int m = std::min(WARP_BATCH_MAX, R.n - i0);
        const bool tabs = m >= 4;
        if (what == VS_WARP_TABLES_ONLY && !tabs) continue;
        const int w = tabs ? what : VS_WARP_ALL;
        int32_t* T = tabs ? g->d_tabs[R.set] + (size_t)i0 * g->tab_ints : nullptr;
        const uint8_t* srcs[WARP_BATCH_MAX];
        uint8_t* dsts[WARP_BATCH_MAX];
        for (int i = 0; i < m; i++) {
            const WarpEnds e = warp_ends(R.owner[i0 + i], R.srcs[i0 + i], R.dsts[i0 + i], R.pad_idx[i0 + i], pad, crop);
            srcs[i] = e.src; dsts[i] = e.dst;
            // pad: the frames get their border first and the padded frames are warped into the (larger) results
            if (pad && what != VS_WARP_TABLES_ONLY && rc == VS_OK)
                rc = launch_make_border(R.srcs[i0 + i], s0->src_pitch, s0->w, s0->h, s0->cn, const_cast<uint8_t*>(e.src), prow, bsz, p.border_type, st);
        }
        if (rc != VS_OK) break;
        if (s0->fmt == VS_FMT_NV12) {
            // interleaved chroma plane: half size, two channels, the map with the halved translation (Minv + 6)
            const size_t suv = src_uv(s0);
            const uint8_t* us[WARP_BATCH_MAX];
            uint8_t* ud[WARP_BATCH_MAX];
            bool same_duv = true;
            for (int i = 0; i < m; i++) {
                us[i] = srcs[i] + suv;
                ud[i] = dsts[i] + dst_uv(s0, dsts[i], R.stride);
                same_duv &= dst_uv(s0, dsts[i], R.stride) == dst_uv(s0, dsts[0], R.stride);
            }
            const int sy = tab_layout(s0->w, s0->h).stride;
            if (tabs && w != VS_WARP_ONLY) {       // the tables by a launch per plane
                rc = launch_warp_affine_list(srcs, dsts, m, s0->src_pitch, s0->w, s0->h, R.stride, s0->w, s0->h, 1, g->d_MinvB[R.set] + 12 * i0, 12, T, st,
                                             VS_WARP_TABLES_ONLY, g->tab_ints);
                if (rc == VS_OK)
                    rc = launch_warp_affine_list(us, ud, m, s0->src_pitch, s0->w / 2, s0->h / 2, R.stride, s0->w / 2, s0->h / 2, 2,
                                                 g->d_MinvB[R.set] + 12 * i0 + 6, 12, T + sy, st, VS_WARP_TABLES_ONLY, g->tab_ints);
            }
            if (w == VS_WARP_TABLES_ONLY || rc != VS_OK) continue;
            // luma and chroma tiles of the launch's frames in ONE grid
            int one = tabs && same_duv ? launch_warp_nv12_list(srcs, dsts, m, s0->src_pitch, R.stride, s0->w, s0->h, suv, dst_uv(s0, dsts[0], R.stride), T, st)
                                       : VS_ERR_UNSUPPORTED;
            if (one == VS_ERR_UNSUPPORTED) {       // (geometry outside what that kernel packs, or fewer than four frames: plane by plane)
                rc = launch_warp_affine_list(srcs, dsts, m, s0->src_pitch, s0->w, s0->h, R.stride, s0->w, s0->h, 1, g->d_MinvB[R.set] + 12 * i0, 12, T, st,
                                             tabs ? VS_WARP_ONLY : VS_WARP_ALL, g->tab_ints);
                if (rc == VS_OK)
                    rc = launch_warp_affine_list(us, ud, m, s0->src_pitch, s0->w / 2, s0->h / 2, R.stride, s0->w / 2, s0->h / 2, 2,
                                                 g->d_MinvB[R.set] + 12 * i0 + 6, 12, tabs ? T + sy : nullptr, st, tabs ? VS_WARP_ONLY : VS_WARP_ALL, g->tab_ints);
            } else rc = one;
            continue;
        }
        rc = launch_warp_affine_list(srcs, dsts, m, pad ? prow : s0->src_pitch, pw, ph, crop ? prow : R.stride, pw, ph, s0->cn,
                                     g->d_MinvB[R.set] + 12 * i0, 12, T, st, w, g->tab_ints);
        // crop-and-zoom: the inner part of the warped scratch frames is resized to the results
        for (int i = 0; crop && what != VS_WARP_TABLES_ONLY && i < m && rc == VS_OK; i++)
            rc = launch_resize_linear(dsts[i] + ((size_t)bsz * s0->w + bsz) * s0->cn, prow, s0->w - 2 * bsz, s0->h - 2 * bsz, s0->cn,
                                      R.dsts[i0 + i], R.stride, R.owner[i0 + i]->orig_w, R.owner[i0 + i]->orig_h, st);
    }
    if (rc != VS_OK) g->err = get_last_error();
    return rc;
}

// The warps of the step in g->ready.  They run on `pre`, behind the gray / pyramid work of the step that issues them and in front of
// the next step's: the cycle warps -> gray -> pyramid -> warps that sets the step time is then the order of ONE stream (as launches
// on `main` with events in both directions - the pyramid's to `main`, the warps' back to `pre` - every period paid two event hand-overs,
// 2 x 25 us of 545).  Their maps and tables come from the tail on `main`: ev_rel.
int group_launch_ready(vs_batch* g, hipEvent_t det_done = nullptr) {
    vs_batch::Ready& R = g->ready;
    if (!R.valid) return VS_OK;
    hipStream_t st = g->st_pre;
    vs_stab* s0 = g->m[0];
    int rc;
    // what the warps wait for - the step's maps and tables (ev_rel), the wide launches of the detector (det_done) - is gathered on the
    // upload stream into ONE event: one packet in front of the warps on `pre` instead of two
    if (g->rel_valid[R.set] || det_done) {
        hipError_t e = hipSuccess;
        if (det_done) e = hipStreamWaitEvent(g->st_up, det_done, 0);
        if (e == hipSuccess && g->rel_valid[R.set]) e = hipStreamWaitEvent(g->st_up, g->ev_rel[R.set], 0);
        if (e == hipSuccess) e = hipEventRecord(g->ev_go, g->st_up);
        if (e == hipSuccess) e = hipStreamWaitEvent(st, g->ev_go, 0);
        if (e != hipSuccess) return gfail(g, VS_ERR_HIP, "hipStreamWaitEvent failed");
        if (g->rel_valid[R.set]) g->pre_rel_step = std::max(g->pre_rel_step, R.step);
        g->rel_valid[R.set] = false;
    }
    {
        StageScope t(s0, VS_STAGE_WARP, st);       // (stage times of a group are booked on its first member)
        rc = group_ready_launches(g, R.tabs_built ? VS_WARP_ONLY : VS_WARP_ALL, st);
    }
    if (hipEventRecord(g->ev_warp[R.set], st) == hipSuccess) { g->warp_valid[R.set] = true; g->last_warp_set = R.set; }
    for (int i = 0; i < R.n; i++) {
        const int slot = R.slots[i];
        vs_stab* o = R.owner[i];
        if (slot < 0 || !o) continue;              // zero-copy: the frame is the caller's
        if (hipEventRecord(o->ev_slot[slot], st) == hipSuccess) o->slot_valid[slot] = true;
        o->free_slots.push_back(slot);
    }
    R.valid = false;
    return rc;
}

}  // namespace

bool group_holds_warps(const vs_batch* g) { return g && (g->ready.valid || g->next.valid); }

// One step: everything the members have queued.
int group_run(vs_batch* g) {
    std::vector<vs_stab*> act;
    int n = 0, max_n = 0;
    for (vs_stab* s : g->m)
        if (!s->bq.empty()) { act.push_back(s); n += (int)s->bq.size(); max_n = std::max(max_n, (int)s->bq.size()); }
    if (n == 0) return VS_OK;
    G_HIP(g, hipSetDevice(g->device));
    const vs_stab* s0 = g->m[0];
    for (vs_stab* s : act)
        if (!s->allocated || !s->batch_active || !s0->allocated || !same_launch_shape(s, s0))
            return gfail(g, VS_ERR_INVALID_ARG, "vs_batch: the streams of a group share one frame geometry, pitch, input mode and launch shape "
                                                "(analysis size, pyramid depth, tracking window, hypothesis count, border mode)");
    if (n > g->cap || max_n > BATCH_MAX) return gfail(g, VS_ERR_CAPACITY, "vs_batch: more frames queued than a step holds");
    if (!g->allocated) G_TRY(g, group_allocate(g));
    const int k = g->batch_id++;
    // host images of this step's tables: the set step k-4 used (its tail, the last reader of anything uploaded from it, has run
    // by now unless the host is four steps ahead of the GPU - then it waits here)
    if (k >= 4) G_HIP(g, hipEventSynchronize(g->ev_blk[k % 4]));
    uint8_t* hset = g->h_tables + (size_t)(k % 4) * g->h_set_bytes;
    ImgPair* h_pairs = reinterpret_cast<ImgPair*>(hset + g->ho_pairs);
    uint8_t *h_lk = hset + g->ho_lk, *h_rs = hset + g->ho_rs, *h_tail = hset + g->ho_tail, *h_gf = hset + g->ho_gf, *h_seg = hset + g->ho_seg;
    const int dset = k & 1;
    // ---- pre: gray images and pyramids of all frames of the step, one launch per stage and level
    if (k >= 2 && g->pre_rel_step < k - 2) {
        // ring reuse: these pyramid slots were read by the analysis two steps ago (npyr = 2 * batch + 2).  (When that step had
        // outputs this stream has waited for its tail already, in front of its warps: nothing to wait for.)
        G_HIP(g, hipStreamWaitEvent(g->st_pre, g->ev_blk[(k - 2) % 4], 0));
        if (g->bdet_valid[(k - 2) % 4]) G_HIP(g, hipStreamWaitEvent(g->st_pre, g->ev_bdet[(k - 2) % 4], 0));
    }
    // (The HBM-bound warps stay alone on the GPU although the host runs steps ahead: they are launches on this stream, behind the
    // pyramid of the step that issues them and in front of the next step's gray kernels - group_launch_ready.  Round 3 had them on
    // `main` and an event from there that this stream waited for; without that guard 99.0 - 102.4 k frames/s against 112.4 - 114.2 k,
    // warps 182 us instead of 86.)
    // ---- argument tables of the tracker, the scoring and the tail (segment = stream): filled here (they do not depend on this
    // step's images), uploaded in the middle of `pre`; which outputs become due and where their maps go is known on the host
    const int set = g->pend_set;
    vs_batch::Ready& R = g->next;           // (g->ready still holds the warps of the step before: they go out further down)
    int idx = 0, n_max = 0, npend = 0, nseg = 0, any_apart = 0, all_apart = 1, ndue = 0;
    size_t pend_stride = 0;
    for (vs_stab* s : act) {
        for (const vs_stab::BFrame& b : s->bq) ndue += b.out_due ? 1 : 0;
        if (s->p.smoothing_method == VS_SMOOTH_KALMAN) all_apart = 0;
    }
    // the coordinate tables of a due frame's warp are built by the workgroup that releases the frame (every stream but a Kalman
    // one: those maps come out of the tail workgroup, and the step's tables are a launch of their own behind it)
    const vs_params_c& p0 = s0->p;
    const bool pad = p0.border_size > 0 && !p0.crop_n_zoom;
    const bool crop = p0.border_size > 0 && p0.crop_n_zoom && s0->w - 2 * p0.border_size > 0 && s0->h - 2 * p0.border_size > 0;
    for (vs_stab* s : act) {
        const vs_params_c& p = s->p;
        const int ns = (int)s->bq.size(), first = idx;
        int npad = 0;
        for (int i = 0; i < ns; i++, idx++) {
            const vs_stab::BFrame& b = s->bq[i];
            const vs_stab::ItemBufs& it = s->items[i];
            LKLevel L[MAX_PYR];
            for (int l = 0; l <= s->levels; l++) {
                L[l].prev = s->pyr[b.pv].img[l]; L[l].next = s->pyr[b.c].img[l]; L[l].deriv = s->pyr[b.pv].der[l];
                L[l].w = s->lw[l]; L[l].h = s->lh[l]; L[l].stride = s->lw[l];
            }
            const int cap = std::max(b.lk_cap, 0);
            n_max = std::max(n_max, cap);
            G_TRY(g, lk_fill_item(h_lk + lk_item_bytes() * idx, L, s->levels, s->d_pts[b.lk_buf], cap, s->d_npts[b.lk_buf], it.next, it.status, it.err,
                                  p.lk_win_size, p.lk_max_iters, p.lk_epsilon));                                    // :611-619
            G_TRY(g, ransac_fill_item(h_rs + ransac_item_bytes() * idx, s->d_pts[b.lk_buf], it.next, it.status, cap, s->d_npts[b.lk_buf], it.vp, it.vc,
                                      it.m, 4, p.ransac_threshold, p.ransac_max_iters, s->tab, it.counts, it.model, it.inliers, it.info, s->d_traj,
                                      &s->tp, s->d_dbg, b.have_prev_gray));
            ransac_item_set_last(h_rs + ransac_item_bytes() * idx, i == ns - 1 ? 1 : 0);
            ransac_item_set_tail_in(h_rs + ransac_item_bytes() * idx, g->d_tin[dset] + tail_in_bytes() * idx);
            double* minv = nullptr;
            WarpTabJob jobs[2] = {{nullptr, nullptr, nullptr, 0, 0}, {nullptr, nullptr, nullptr, 0, 0}};
            if (b.out_due) {
                if (npend > 0 && pend_stride != b.out_stride) return gfail(g, VS_ERR_INVALID_ARG, "batch mode: one output pitch per step");
                minv = g->d_MinvB[set] + 12 * npend;
                R.srcs[npend] = b.out_frame; R.dsts[npend] = b.d_out; R.slots[npend] = b.out_slot; R.owner[npend] = s; R.pad_idx[npend] = npad;
                // (launches of fewer than four frames - the rest of a step's due frames beyond a multiple of 32 - run without tables)
                if (all_apart && std::min(WARP_BATCH_MAX, ndue - npend / WARP_BATCH_MAX * WARP_BATCH_MAX) >= 4) {
                    const WarpEnds e = warp_ends(s, b.out_frame, b.d_out, npad, pad, crop);
                    int32_t* T = g->d_tabs[set] + (size_t)npend * g->tab_ints;
                    jobs[0] = WarpTabJob{T, e.src, e.dst, pad ? s->w + 2 * p0.border_size : s->w, pad ? s->h + 2 * p0.border_size : s->h};
                    if (s->fmt == VS_FMT_NV12)
                        jobs[1] = WarpTabJob{T + tab_layout(s->w, s->h).stride, e.src + src_uv(s), e.dst + dst_uv(s, e.dst, b.out_stride), s->w / 2, s->h / 2};
                }
                npad++;
                pend_stride = b.out_stride;
                npend++;
            }
            tail_fill_item(h_tail + tail_item_bytes() * idx, b.out_due ? 1 : 0, b.out_idx, minv, jobs);
            tail_item_set_seg(h_tail + tail_item_bytes() * idx, nseg);
        }
        tail_fill_seg(h_seg + tail_seg_bytes() * nseg, first, ns, s->d_M, s->d_traj, s->d_dbg, p.smoothing_method);
        any_apart |= p.smoothing_method != VS_SMOOTH_KALMAN;
        nseg++;
        if (s->bq[0].prev_small) {   // Stabilizer.cpp:598-603 (once per stream: 480x270 -> analysis size)
            StageScope t(g->m[0], VS_STAGE_PYRAMID, g->st_pre);
            G_TRY(g, launch_resize_gray(s->d_first_gray, 480, 480, 270, VS_FMT_GRAY8, s->pyr[s->bq[0].pv].img[0], s->aw, s->aw, s->ah, g->st_pre));
            G_TRY(g, build_pyramid(s, s->bq[0].pv, g->st_pre));
        }
    }
    {
        // pair tables: [0] frame -> img[0]; [1..levels] img[l-1] -> img[l]; [levels+1 ..] img[l] -> der[l]
        const int L = s0->levels;
        int aligned = 1, n_detect = 0;
        // level-0 pairs: the frames that re-detect first, so that the detector can start after a first, smaller launch
        for (vs_stab* s : act) for (const vs_stab::BFrame& b : s->bq) n_detect += b.detect ? 1 : 0;
        int n_first = 0, n_rest = 0, i = 0;
        for (vs_stab* s : act)
            for (const vs_stab::BFrame& b : s->bq) {
                const Pyramid& P = s->pyr[b.c];
                const int slot = b.detect ? n_first++ : n_detect + n_rest++;
                h_pairs[slot] = ImgPair{b.frame, P.img[0]};
                if ((uintptr_t)b.frame % 8) aligned = 0;
                for (int l = 1; l <= L; l++) h_pairs[(size_t)l * n + i] = ImgPair{P.img[l - 1], P.img[l]};
                for (int l = 0; l <= L; l++) h_pairs[(size_t)(L + 1 + l) * n + i] = ImgPair{P.img[l], P.der[l]};
                i++;
            }
        // ONE upload per step: the pair tables of the gray / pyramid launches and, behind them in the set, the tables of the tracker,
        // the scoring and the tail (all filled above).  (As five copies - the four small ones in the middle of `pre`, which had slack
        // while the warps ran on `main` - they stood 46 us on what is the step's longest chain since the warps run on this stream.)
        // The copy runs on a stream of its own: the host is a step or two ahead of the GPU, so the tables are there long before `pre`
        // gets to this step (as a copy on `pre` it stood between the warps and the gray kernels: 26 us of hand-over to the copy engine
        // and back on the step's longest chain).  The set was last used by step k - 2: its tail must have run.
        ImgPair* const d_pairs = g->d_pairs[dset];
        if (k >= 2) G_HIP(g, hipStreamWaitEvent(g->st_up, g->ev_blk[(k - 2) % 4], 0));
        G_HIP(g, hipMemcpyAsync(g->d_pairs[dset], hset, g->up_bytes, hipMemcpyHostToDevice, g->st_up));
        G_HIP(g, hipEventRecord(g->ev_up[dset], g->st_up));
        G_HIP(g, hipStreamWaitEvent(g->st_pre, g->ev_up[dset], 0));
        {
            StageScope t(g->m[0], VS_STAGE_GRAY, g->st_pre);
            // NV12: the Y plane is the gray image (no reference path; same policy as the per-frame pipeline)
            const int gfmt = s0->fmt == VS_FMT_NV12 ? VS_FMT_GRAY8 : s0->fmt;
            const int n_a = (n_detect > 0 && n_detect < n) ? n_detect : n;
            G_TRY(g, launch_resize_gray_batch(d_pairs, n_a, s0->src_pitch, s0->w, s0->h, gfmt, s0->aw, s0->aw, s0->ah, aligned, g->st_pre));  // :448-450
            G_HIP(g, hipEventRecord(g->ev_bgray, g->st_pre));      // the detector needs the analysis images of its frames only
            if (n_a < n)
                G_TRY(g, launch_resize_gray_batch(d_pairs + n_a, n - n_a, s0->src_pitch, s0->w, s0->h, gfmt, s0->aw, s0->aw, s0->ah, aligned, g->st_pre));
        }
        StageScope t(g->m[0], VS_STAGE_PYRAMID, g->st_pre);
        // One launch per level (pyr_level_kernel): derivatives of level l and the image of level l+1 from one staged read of
        // level l.  (83.0 k -> 92.2 k frames/s at 1080p against the two stencils as separate launches, round 2.)
        for (int l = 0; l <= L; l++)
            G_TRY(g, launch_pyr_level_batch(d_pairs + (size_t)(L + 1 + l) * n, l < L ? d_pairs + (size_t)(l + 1) * n : nullptr, n, s0->lw[l], s0->lw[l],
                                            s0->lh[l], l < L ? s0->lw[l + 1] : 0, g->st_pre));
    }
    // (`main` waits for the event behind this step's warps when there are any: it covers the pyramid, which lies in front of them)
    if (!g->ready.valid) G_HIP(g, hipEventRecord(g->ev_bpre, g->st_pre));
    // ---- det: every frame of the step that re-detects, one launch per GFTT stage
    int ndet = 0;
    for (vs_stab* s : act) {
        int local = 0;
        for (const vs_stab::BFrame& b : s->bq) {
            if (!b.detect) continue;
            G_TRY(g, gftt_fill_item(h_gf + gftt_item_bytes() * ndet, s->pyr[b.c].img[0], s->aw, s->aw, s->ah, s->pts_cap[b.det_buf], 0.02, 15.0, 3,
                                    s->gws[local], s->d_pts[b.det_buf], s->d_npts[b.det_buf]));                      // :740-744
            s->dbg_det_pts = s->d_pts[b.det_buf]; s->dbg_det_n = s->d_npts[b.det_buf];
            s->dbg_gftt_counters = s->gws[local].counters;
            local++; ndet++;
        }
        s->last_detected = s->bq.back().detect;
    }
    hipStream_t sd = g->st_det;
    if (ndet > 0) {
        // starts as soon as the analysis images of its frames exist, next to the pyramid levels of this step and the tracking of
        // the previous one (the table and the reset of the counters first: they are through by the time the images are)
        G_HIP(g, hipMemcpyAsync(g->d_gf, h_gf, gftt_item_bytes() * ndet, hipMemcpyHostToDevice, sd));
        // keypoint buffers are recycled after B + 4 detections (two steps): the tracking of the step before the previous one
        // must have read them (the GFTT scratch is only touched on this stream)
        if (k >= 2) G_HIP(g, hipStreamWaitEvent(sd, g->ev_blk[(k - 2) % 4], 0));
        G_TRY(g, launch_gftt_batch(g->d_gf, ndet, s0->aw, s0->ah, 3, sd, 1));
        G_HIP(g, hipStreamWaitEvent(sd, g->ev_bgray, 0));
        {
            StageScope t(g->m[0], VS_STAGE_GFTT, sd);
            G_TRY(g, launch_gftt_batch(g->d_gf, ndet, s0->aw, s0->ah, 3, sd, 4));
            G_TRY(g, launch_gftt_batch(g->d_gf, ndet, s0->aw, s0->ah, 3, sd, 5));
            // the wide launches of the detection are through: the warps of the step before may go (below); the selection -
            // one workgroup per image - runs beside them
            G_HIP(g, hipEventRecord(g->ev_bnms, sd));
            G_TRY(g, launch_gftt_batch(g->d_gf, ndet, s0->aw, s0->ah, 3, sd, 3));
        }
        G_HIP(g, hipEventRecord(g->ev_bdet[k % 4], sd));
        g->last_det_batch = k;
    }
    g->bdet_valid[k % 4] = ndet > 0;
    // ---- main: tracking and hypothesis scoring of all frames, one launch each
    hipStream_t st = g->st;
    const bool wait_det = g->last_det_batch >= 0 && g->last_det_batch >= k - 1;
    const bool early = wait_det && g->last_det_batch == k;
    // The warps of the PREVIOUS step go out here, on `pre` behind this step's pyramid, once the wide launches of this step's
    // detection are through: nothing but the corner selection (a workgroup per image) runs beside them.
    const bool warps_go = g->ready.valid;
    G_TRY(g, group_launch_ready(g, early ? g->ev_bnms : (wait_det ? g->ev_bdet[g->last_det_batch % 4] : (hipEvent_t) nullptr)));
    // ---- main: waits for this step's gray / pyramid work, its corners and - the tracker takes every vector register of every SIMD,
    // beside it the warps would crawl - the warps just issued
    if (warps_go && g->last_warp_set >= 0) G_HIP(g, hipStreamWaitEvent(st, g->ev_warp[g->last_warp_set], 0));
    else G_HIP(g, hipStreamWaitEvent(st, g->ev_bpre, 0));
    for (vs_stab* s : act)
        if (s->pts_pending[0]) { G_HIP(g, hipStreamWaitEvent(st, s->pts_event[0], 0)); s->pts_pending[0] = false; }
    if (wait_det) G_HIP(g, hipStreamWaitEvent(st, g->ev_bdet[g->last_det_batch % 4], 0));       // the tracker needs the selected corners
    {
        StageScope t(g->m[0], VS_STAGE_LK, st);
        G_TRY(g, launch_pyr_lk_batch(g->d_lk[dset], n, n_max, s0->p.lk_win_size, st));
    }
    {
        StageScope t(g->m[0], VS_STAGE_RANSAC, st);
        G_TRY(g, launch_ransac_score_batch(g->d_rs[dset], n, s0->p.ransac_max_iters, n_max, st));
    }
    for (vs_stab* s : act)
        if (s->dbg_delay_us > 0) { G_TRY(g, launch_spin(s->dbg_delay_us, st)); break; }
    // ---- ordered tails, ONE launch (a workgroup per stream): per frame in push order the trajectory append (:644-693), then
    // the map of the output that has become due (applyNextSmoothTransform sees exactly the transforms appended so far)
    if (npend > 0 && g->warp_valid[set]) {         // the previous user of this set of maps must have read them
        G_HIP(g, hipStreamWaitEvent(st, g->ev_warp[set], 0));
        g->warp_valid[set] = false;
    }
    {
        StageScope t(g->m[0], VS_STAGE_TRAJ, st);
        G_TRY(g, launch_ransac_tail_group(g->d_rs[dset], g->d_tail[dset], g->d_seg[dset], g->d_tin[dset], nseg, max_n, n, any_apart, st));
    }
    // the keypoint and pyramid buffers of this step may be recycled (two steps on) once the tail, which still reads the points
    // and their counts, has run
    G_HIP(g, hipEventRecord(g->ev_blk[k % 4], st));
    // the warps of this step wait for the next one (or a drain); their maps exist once the tail has run: the coordinate tables
    // are built right behind it
    R.n = npend; R.set = set; R.stride = pend_stride; R.valid = npend > 0; R.tabs_built = all_apart != 0; R.step = k;
    std::swap(g->ready, g->next);            // (the previous step's warps have been issued: g->ready was free)
    if (g->ready.valid) {
        g->pend_set = set ^ 1;
        if (!g->ready.tabs_built) {          // (a Kalman stream in the step: the tables as a launch behind the tail)
            StageScope t(g->m[0], VS_STAGE_WARP_TABLES, st);
            G_TRY(g, group_ready_launches(g, VS_WARP_TABLES_ONLY, st));
            g->ready.tabs_built = true;
        }
        G_HIP(g, hipEventRecord(g->ev_rel[set], st));          // maps and tables of this step's warps exist
        g->rel_valid[set] = true;
    }
    for (vs_stab* s : act) {
        const vs_stab::BFrame& lb = s->bq.back();
        const int nl = (int)s->bq.size();
        s->dbg_prev_pts = s->d_pts[lb.lk_buf]; s->dbg_next = s->items[nl - 1].next;
        s->dbg_status = s->items[nl - 1].status; s->dbg_inliers = s->items[nl - 1].inliers;
        s->bq.clear();
    }
    return VS_OK;
}

// Everything the members have queued is analysed and its warps are issued.  (The warps of a step normally go out with the NEXT
// step, between its detection and its tracking; group_run issues the pending ones itself, so the step before the drained one is
// covered too.)
int group_drain(vs_batch* g) {
    G_HIP(g, hipSetDevice(g->device));
    G_TRY(g, group_run(g));                 // (issues the warps of the step before on its way; nothing queued: nothing done)
    return group_launch_ready(g);
}

void group_delete(vs_batch* g) {
    if (!g) return;
    group_free(g);
    auto kill = [](hipEvent_t& e) { if (e) { (void)hipEventDestroy(e); e = nullptr; } };
    kill(g->ev_bpre); kill(g->ev_bgray); kill(g->ev_bnms); kill(g->ev_warp[0]); kill(g->ev_warp[1]); kill(g->ev_rel[0]); kill(g->ev_rel[1]); kill(g->ev_up[0]); kill(g->ev_up[1]); kill(g->ev_go);
    for (auto& e : g->ev_bdet) kill(e);
    for (auto& e : g->ev_blk) kill(e);
    delete g;
}

// The private schedule of a standalone instance in batch mode: a group of one, steps of `batch` frames.
vs_batch* group_new_own(vs_stab* s) {
    vs_batch* g = new (std::nothrow) vs_batch();
    if (!g) { set_last_error("out of host memory"); return nullptr; }
    g->device = s->device; g->S = 1; g->B = s->batch; g->cap = s->batch; g->own = true;
    g->m.push_back(s);
    g->st = s->st; g->st_pre = s->st_pre; g->st_det = s->st_det; g->st_up = s->st_warp;
    if (!group_make_events(g)) { set_last_error("hipEventCreate failed"); group_delete(g); return nullptr; }
    return g;
}

namespace {
// vs_stab_* calls that vs_batch_* makes on a member
struct MemberCall {
    vs_stab* s;
    explicit MemberCall(vs_stab* s_) : s(s_) { s->group_call = true; }
    ~MemberCall() { s->group_call = false; }
};
}  // namespace

extern "C" {

// params: ONE block for all streams (per_stream = 0) or n_streams blocks.  Per-stream blocks may differ in everything that does
// not shape a launch (same_launch_shape): smoothing radius and method, horizon lock, the drone filters' settings, corner count.
static int batch_create(int device, int n_streams, const vs_params_c* params, int per_stream, int frames_per_step, vs_batch** out) {
    if (!out) return VS_ERR_INVALID_ARG;
    *out = nullptr;
    if (!params || n_streams < 1 || n_streams > 256 || frames_per_step < 1 || frames_per_step > BATCH_MAX) {
        set_last_error("vs_batch_create: 1..256 streams, 1..64 frames per stream and step");
        return VS_ERR_INVALID_ARG;
    }
    for (int i = 0; i < (per_stream ? n_streams : 1); i++) {
        const vs_params_c& p = params[i];
        if (p.struct_size != (int32_t)sizeof(vs_params_c)) { set_last_error("params: struct_size mismatch"); return VS_ERR_INVALID_ARG; }
        // modes whose outputs depend on each other or on a host decision per output run in the per-frame pipeline only
        if (p.adaptive_smoothing || (p.border_size > 0 && !p.crop_n_zoom && p.border_type == VS_BORDER_FADE) || (p.enable_virtual_canvas && !p.crop_n_zoom)) {
            set_last_error("vs_batch_create: adaptive smoothing, the fade border and the virtual canvas are per-stream modes (use vs_stab_*)");
            return VS_ERR_UNSUPPORTED;
        }
    }
    vs_batch* g = new (std::nothrow) vs_batch();
    if (!g) return VS_ERR_HIP;
    g->device = device; g->S = n_streams; g->B = frames_per_step; g->cap = n_streams * frames_per_step;
    for (int i = 0; i < n_streams; i++) {
        vs_stab* s = nullptr;
        int rc = vs_stab_create(&params[per_stream ? i : 0], device, &s);
        if (rc == VS_OK) rc = vs_stab_set_batch(s, frames_per_step);
        if (rc != VS_OK) { if (s) vs_stab_destroy(s); vs_batch_destroy(g); return rc; }
        s->group = g; s->member = true;
        g->m.push_back(s);
    }
    g->st = g->m[0]->st; g->st_pre = g->m[0]->st_pre; g->st_det = g->m[0]->st_det; g->st_up = g->m[0]->st_warp;
    if (!group_make_events(g)) { set_last_error("vs_batch_create: hipEventCreate failed"); vs_batch_destroy(g); return VS_ERR_HIP; }
    *out = g;
    return VS_OK;
}

int vs_batch_create(int device, int n_streams, const vs_params_c* params, int frames_per_step, vs_batch** out) {
    return batch_create(device, n_streams, params, 0, frames_per_step, out);
}

int vs_batch_create_params(int device, int n_streams, const vs_params_c* params_per_stream, int frames_per_step, vs_batch** out) {
    return batch_create(device, n_streams, params_per_stream, 1, frames_per_step, out);
}

void vs_batch_destroy(vs_batch* g) {
    if (!g) return;
    (void)hipSetDevice(g->device);
    if (g->st_pre) (void)hipStreamSynchronize(g->st_pre);
    if (g->st_det) (void)hipStreamSynchronize(g->st_det);
    if (g->st) (void)hipStreamSynchronize(g->st);
    for (vs_stab* s : g->m) { s->group = nullptr; s->member = false; s->bq.clear(); vs_stab_destroy(s); }
    group_delete(g);
}

int vs_batch_streams(const vs_batch* g) { return g ? g->S : 0; }
vs_stab* vs_batch_stream(vs_batch* g, int i) { return (g && i >= 0 && i < g->S) ? g->m[(size_t)i] : nullptr; }
const char* vs_batch_last_error(const vs_batch* g) { return g ? g->err.c_str() : ""; }

int vs_batch_set_zero_copy(vs_batch* g, int enable) {
    if (!g) return VS_ERR_INVALID_ARG;
    for (vs_stab* s : g->m) { MemberCall mc(s); const int rc = vs_stab_set_zero_copy(s, enable); if (rc != VS_OK) { g->err = s->err; return rc; } }
    return VS_OK;
}

int vs_batch_set_nv12_layout(vs_batch* g, size_t in_uv_offset, size_t out_uv_offset) {
    if (!g) return VS_ERR_INVALID_ARG;
    for (vs_stab* s : g->m) { MemberCall mc(s); const int rc = vs_stab_set_nv12_layout(s, in_uv_offset, out_uv_offset); if (rc != VS_OK) { g->err = s->err; return rc; } }
    return VS_OK;
}

int vs_batch_push_dev(vs_batch* g, const void* const* d_frames, int w, int h, size_t stride, int fmt, void* const* d_outs, size_t out_stride,
                      int* produced) {
    if (!g || !d_frames || !d_outs || !produced) return VS_ERR_INVALID_ARG;
    for (int i = 0; i < g->S; i++) produced[i] = 0;
    bool full = false;
    for (int i = 0; i < g->S; i++) {
        if (!d_frames[i]) continue;                       // no frame for this stream in this call
        vs_stab* s = g->m[(size_t)i];
        MemberCall mc(s);
        const int rc = vs_stab_push_dev(s, d_frames[i], w, h, stride, fmt, d_outs[i], out_stride, &produced[i]);
        if (rc != VS_OK) { g->err = s->err; return rc; }
        full |= (int)s->bq.size() >= g->B;
    }
    if (full) return group_run(g);
    return VS_OK;
}

int vs_batch_flush_dev(vs_batch* g, void* const* d_outs, size_t out_stride, int* produced) {
    if (!g || !d_outs || !produced) return VS_ERR_INVALID_ARG;
    int rc = group_drain(g);
    if (rc != VS_OK) return rc;
    for (int i = 0; i < g->S; i++) {
        produced[i] = 0;
        MemberCall mc(g->m[(size_t)i]);
        rc = vs_stab_flush_dev(g->m[(size_t)i], d_outs[i], out_stride, &produced[i]);
        if (rc != VS_OK) { g->err = g->m[(size_t)i]->err; return rc; }
    }
    return VS_OK;
}

int vs_batch_sync(vs_batch* g) {
    if (!g) return VS_ERR_INVALID_ARG;
    int rc = group_drain(g);
    if (rc != VS_OK) return rc;
    for (vs_stab* s : g->m) { rc = vs_stab_sync(s); if (rc != VS_OK) { g->err = s->err; return rc; } }
    return VS_OK;
}

}  // extern "C"
