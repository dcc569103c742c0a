// CPU ORACLE (test infrastructure) - vs::Stabilizer restated.
// Control flow, constants and quirks follow /root/reference/src/Stabilizer.cpp
// CPU branch (useGpu_ == false); line numbers cited inline.  Deviations:
//   * per-instance re-detection counter instead of the function-static one
//     (:696; identical for a single instance - SURVEY Q3);
//   * queued frames are always copied (:376 aliases the caller's buffer);
//   * NV12 input (no reference path): analysis runs on the luma plane.
//   * enableVirtualCanvas (vso_canvas.cpp) replaces the warped frame (:1130-1134): the warp and the fade history behind
//     it cannot be observed then and are not run; BGR8 streams only (cv::cvtColor(BGR2GRAY) of :2225 needs 3 channels).
#include "vso_internal.h"

#include <algorithm>
#include <cstring>
#include <deque>

namespace vso {
int g_threads = 1;
}

using namespace vso;

struct vso_stab {
    vs_params_c p;
    std::deque<std::vector<uint8_t>> frameQueue;
    std::deque<int> idxQueue;
    std::vector<float> transforms, path;  // n*3
    Gray prevGray;
    std::vector<float> prevKeypoints;
    bool firstFrame = true;
    int nextFrameIndex = 0;
    int frameW = 0, frameH = 0, fmt = VS_FMT_BGR8;
    int origW = 0, origH = 0;
    std::vector<uint8_t> borderHistory;   // Stabilizer.h:356-358 (borderHistory_, fadeFrameCount_)
    int fadeFrameCount = 0;
    CanvasState* canvas = nullptr;        // Stabilizer.h:408-416
    int detectCounter = 0;
    // drone (Stabilizer.h:418-429)
    std::deque<std::pair<float, float>> hfHistory;
    float hfMedian[2] = {0, 0};
    float hfRotLP = 0;
    bool hfInDeadZone = false;
    int hfFreezeCounter = 0;
    float hfAccum = 0;
    // debug
    vs_debug_frame dbg;
    std::vector<float> dbgPrev, dbgCurr, dbgDetected;
    std::vector<uint8_t> dbgStatus, dbgInliers;
    Gray dbgGray;
};

static size_t frame_bytes(int fmt, int h, size_t stride) {
    return fmt == VS_FMT_NV12 ? stride * h * 3 / 2 : stride * h;
}
static int fmt_cn(int fmt) { return fmt == VS_FMT_BGR8 ? 3 : 1; }

// resize + BGR2GRAY (Stabilizer.cpp:304-305, 448-450)
static void analysis_gray(const uint8_t* data, int w, int h, size_t stride, int fmt, int aw, int ah, Gray& g) {
    g.create(aw, ah);
    if (fmt == VS_FMT_BGR8) {
        std::vector<uint8_t> small((size_t)aw * ah * 3);
        resize_linear_u8(data, w, h, stride, 3, small.data(), aw, ah, (size_t)aw * 3);
        bgr2gray(small.data(), aw, ah, (size_t)aw * 3, g.d.data(), aw);
    } else {
        resize_linear_u8(data, w, h, stride, 1, g.d.data(), aw, ah, aw);
    }
}

// ---- drone helpers (Stabilizer.cpp:2447-2682) --------------------------------
static float hf_mag(const float t[3]) { return std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2] * 100.0f); }

static void dead_zone_freeze(vso_stab* s, float t[3]) {  // :2605-2655
    const vs_params_c& p = s->p;
    float magnitude = hf_mag(t);  // updateMotionAccumulator :2667-2682
    float decayed = s->hfAccum * p.hf_motion_accumulator_decay;
    s->hfAccum = std::max(decayed, magnitude);
    s->hfAccum = std::min(s->hfAccum, p.hf_dead_zone_threshold * 5.0f);
    s->hfAccum = std::max(0.0f, std::min(s->hfAccum, 100.0f));
    float cur = hf_mag(t);
    if (!s->hfInDeadZone && cur < p.hf_dead_zone_threshold) {
        s->hfInDeadZone = true;
        s->hfFreezeCounter = p.hf_freeze_duration;
    }
    if (s->hfInDeadZone) {
        s->hfFreezeCounter--;
        bool durationExpired = s->hfFreezeCounter <= 0;
        bool significantMotion = cur > p.hf_dead_zone_threshold * 1.5f;
        bool accumulatedMotion = s->hfAccum > p.hf_dead_zone_threshold * 1.2f;
        if (durationExpired || significantMotion || accumulatedMotion) {
            s->hfInDeadZone = false;
            s->hfFreezeCounter = 0;
            s->hfAccum = 0.0f;
            return;
        }
        t[0] = t[1] = t[2] = 0.0f;
    }
}

static void median_translation(vso_stab* s) {  // :2531-2553
    if (s->hfHistory.empty()) { s->hfMedian[0] = s->hfMedian[1] = 0; return; }
    std::vector<float> xs, ys;
    for (auto& t : s->hfHistory) { xs.push_back(t.first); ys.push_back(t.second); }
    std::sort(xs.begin(), xs.end());
    std::sort(ys.begin(), ys.end());
    size_t mid = xs.size() / 2;
    s->hfMedian[0] = xs.size() % 2 == 0 ? (xs[mid - 1] + xs[mid]) / 2.0f : xs[mid];
    s->hfMedian[1] = ys.size() % 2 == 0 ? (ys[mid - 1] + ys[mid]) / 2.0f : ys[mid];
}

static void micro_shake(vso_stab* s, float t[3]) {  // :2468-2503
    if (s->hfHistory.size() >= 5) median_translation(s);
    float d0 = t[0] - s->hfMedian[0], d1 = t[1] - s->hfMedian[1];
    float magnitude = std::sqrt(d0 * d0 + d1 * d1);
    if (magnitude < s->p.hf_shake_px) {
        t[0] = s->hfMedian[0] + d0 * 0.01f;
        t[1] = s->hfMedian[1] + d1 * 0.01f;
    } else if (magnitude < s->p.hf_shake_px * 2.0f) {
        t[0] = s->hfMedian[0] + d0 * 0.05f;
        t[1] = s->hfMedian[1] + d1 * 0.05f;
    }
}

static void rotation_low_pass(vso_stab* s, float t[3]) {  // :2505-2520
    if (s->p.horizon_lock) {
        s->hfRotLP = (1.0f - s->p.hf_rot_lp_alpha) * s->hfRotLP + s->p.hf_rot_lp_alpha * t[2];
        t[2] = s->hfRotLP;
    }
}

// Stabilizer.cpp:1461-1492 adaptSmoothingRadius (via updateAdaptiveParameters :1562-1574)
static void update_adaptive(vso_stab* s) {
    size_t n = s->transforms.size() / 3;
    if (n < 3 || !s->p.adaptive_smoothing) return;
    const float* m = &s->transforms[(n - 1) * 3];
    float magnitude = std::sqrt(m[0] * m[0] + m[1] * m[1]);
    float motionScale = std::max(0.0f, std::min(1.0f, magnitude / 50.0f));
    motionScale = 1.0f - motionScale;
    int newRadius = s->p.min_smoothing_radius +
                    (int)(motionScale * (s->p.max_smoothing_radius - s->p.min_smoothing_radius));
    if (newRadius != s->p.smoothing_radius) s->p.smoothing_radius = newRadius;
}

// Stabilizer.cpp:402-761
static void generate_transform(vso_stab* s, const uint8_t* data, int w, int h, size_t stride) {
    vs_params_c& p = s->p;
    int aw = 960, ah = 540;  // :410
    if (p.drone_high_freq_mode) {  // :2447-2466
        int maxWidth = std::min(p.hf_analysis_max_width, w);
        float aspect = (float)h / (float)w;
        int height = (int)(maxWidth * aspect);
        aw = (maxWidth / 2) * 2;
        ah = (height / 2) * 2;
    }
    Gray curr;
    analysis_gray(data, w, h, stride, s->fmt, aw, ah, curr);
    // conditional CLAHE (:453-455) never fires: shouldApplyConditionalCLAHE(-1) is false (:2561-2567)

    vs_debug_frame& d = s->dbg;
    d.n_prev = 0; d.n_valid = 0; d.ransac_best_iter = -1; d.ransac_iters_run = 0; d.n_inliers = 0;
    d.detected = 0; d.n_detected = 0;
    for (int i = 0; i < 6; i++) d.model[i] = NAN;
    s->dbgPrev.clear(); s->dbgCurr.clear(); s->dbgStatus.clear(); s->dbgInliers.clear();
    s->dbgDetected.clear();
    float tr[3] = {0.f, 0.f, 0.f};
    if (!s->prevKeypoints.empty() && !s->prevGray.empty()) {  // :596
        if (s->prevGray.w != aw || s->prevGray.h != ah) {      // :598-603
            Gray r; r.create(aw, ah);
            resize_linear_u8(s->prevGray.d.data(), s->prevGray.w, s->prevGray.h, s->prevGray.w, 1,
                             r.d.data(), aw, ah, aw);
            s->prevGray = r;
        }
        int n = (int)(s->prevKeypoints.size() / 2);
        std::vector<float> tmpCurr((size_t)n * 2), err(n);
        std::vector<uint8_t> status(n);
        pyr_lk(s->prevGray.d.data(), curr.d.data(), aw, ah, aw, s->prevKeypoints.data(), n,
               tmpCurr.data(), status.data(), err.data(), p.lk_win_size, p.lk_max_level,
               p.lk_max_iters, p.lk_epsilon, g_threads);  // :611-619
        std::vector<float> vp, vc;
        for (int i = 0; i < n; i++)
            if (status[i]) {
                vp.push_back(s->prevKeypoints[2 * i]); vp.push_back(s->prevKeypoints[2 * i + 1]);
                vc.push_back(tmpCurr[2 * i]); vc.push_back(tmpCurr[2 * i + 1]);
            }
        int m = (int)(vp.size() / 2);
        float T[6] = {1, 0, 0, 0, 1, 0};  // :644
        std::vector<uint8_t> inl(m > 0 ? m : 1, 0);
        if (m >= 4) {  // :645
            double H[6];
            int32_t info[4];
            if (estimate_affine_partial2d(vp.data(), vc.data(), m, p.ransac_threshold,
                                          p.ransac_max_iters, H, inl.data(), info)) {
                for (int i = 0; i < 6; i++) { T[i] = (float)H[i]; d.model[i] = H[i]; }  // :651
            }
            d.ransac_best_iter = info[1]; d.ransac_iters_run = info[2]; d.n_inliers = info[3];
        }
        tr[0] = T[2]; tr[1] = T[5]; tr[2] = libm_atan2f(T[3], T[0]);  // :660-662
        if (p.drone_high_freq_mode) {  // :666-671
            dead_zone_freeze(s, tr);
            micro_shake(s, tr);
            rotation_low_pass(s, tr);
            s->hfHistory.push_back({tr[0], tr[1]});
            if (s->hfHistory.size() > 10) s->hfHistory.pop_front();
        }
        d.n_prev = n; d.n_valid = m;
        s->dbgPrev = s->prevKeypoints; s->dbgCurr = tmpCurr; s->dbgStatus = status;
        inl.resize(m);
        s->dbgInliers = inl;
    }
    s->transforms.insert(s->transforms.end(), tr, tr + 3);  // :673-677
    d.transform[0] = tr[0]; d.transform[1] = tr[1]; d.transform[2] = tr[2];
    // path (:681-688)
    size_t n = s->path.size();
    if (n == 0) s->path.insert(s->path.end(), tr, tr + 3);
    else {
        float last[3] = {s->path[n - 3], s->path[n - 2], s->path[n - 1]};
        s->path.push_back(last[0] + tr[0]); s->path.push_back(last[1] + tr[1]); s->path.push_back(last[2] + tr[2]);
    }
    if (p.adaptive_smoothing) update_adaptive(s);  // :691-693
    if ((++s->detectCounter % 2) == 0) {           // :696-746
        std::vector<float> corners;
        int nc = 0;
        gftt(curr.d.data(), aw, ah, aw, std::min(p.max_corners, 200), 0.02, 15.0, 3, corners, &nc);
        s->prevKeypoints = corners;
        d.detected = 1; d.n_detected = (int)(corners.size() / 2);
        s->dbgDetected = corners;
    }
    s->prevGray = curr;  // :757-759
    s->dbgGray = curr;
}

static bool canvas_on(const vs_params_c& p, int fmt) { return p.enable_virtual_canvas && !p.crop_n_zoom && fmt == VS_FMT_BGR8; }

static void out_size(const vs_params_c& p, int w, int h, int origW, int origH, int* ow, int* oh, int fmt = VS_FMT_BGR8) {
    int b = p.border_size;
    if (canvas_on(p, fmt)) { *ow = w; *oh = h; return; }      // the canvas window has the size of the unpadded frame (:2121-2126)
    if (b > 0 && !p.crop_n_zoom) { *ow = w + 2 * b; *oh = h + 2 * b; return; }
    if (p.crop_n_zoom && b > 0 && w - 2 * b > 0 && h - 2 * b > 0 && origW > 0) { *ow = origW; *oh = origH; return; }
    *ow = w; *oh = h;
}

// Stabilizer.cpp:763-1137
static int apply_next(vso_stab* s, uint8_t* out, size_t out_stride) {
    if (s->frameQueue.empty()) return 0;
    vs_params_c& p = s->p;
    std::vector<uint8_t> frame = std::move(s->frameQueue.front());
    int oldestIdx = s->idxQueue.front();
    s->frameQueue.pop_front();
    s->idxQueue.pop_front();
    const int w = s->frameW, h = s->frameH;
    const int cn = fmt_cn(s->fmt);
    const size_t fstride = (size_t)w * cn;
    const int rows_total = s->fmt == VS_FMT_NV12 ? h * 3 / 2 : h;
    vs_debug_frame& d = s->dbg;
    d.out_index = oldestIdx; d.box_radius = 0; d.intent = 0;
    int n = (int)(s->transforms.size() / 3);
    if (oldestIdx >= n) {  // :774-780 raw frame
        for (int y = 0; y < rows_total; y++) memcpy(out + (size_t)y * out_stride, &frame[(size_t)y * fstride], fstride);
        for (int i = 0; i < 3; i++) d.smoothed[i] = 0;
        float I[6] = {1, 0, 0, 0, 1, 0};
        memcpy(d.warp_matrix, I, sizeof I);
        return 1;
    }
    std::vector<float> px(n), py(n), pa(n);  // :783-791
    for (int i = 0; i < n; i++) { px[i] = s->path[3 * i]; py[i] = s->path[3 * i + 1]; pa[i] = s->path[3 * i + 2]; }
    std::vector<float> sx, sy, sa;
    if (p.smoothing_method == VS_SMOOTH_GAUSSIAN) {  // :797-801
        float sig = (float)p.gaussian_sigma;
        sx = gaussian_filter(px, sig); sy = gaussian_filter(py, sig); sa = gaussian_filter(pa, sig);
    } else if (p.smoothing_method == VS_SMOOTH_KALMAN) {  // :802-806
        sx = kalman_filter(px); sy = kalman_filter(py); sa = kalman_filter(pa);
    } else {  // :807-823
        int ar = adaptive_radius(px, py, pa, p.smoothing_radius);
        bool drone = p.drone_high_freq_mode != 0;
        sx = box_filter(px, ar, drone); sy = box_filter(py, ar, drone); sa = box_filter(pa, ar, drone);
        d.box_radius = drone ? std::max(10, std::min(ar, 50)) : std::max(2, std::min(ar, 8));
    }
    float raw[3] = {s->transforms[3 * oldestIdx], s->transforms[3 * oldestIdx + 1], s->transforms[3 * oldestIdx + 2]};
    float diff[3] = {sx[oldestIdx] - px[oldestIdx], sy[oldestIdx] - py[oldestIdx], sa[oldestIdx] - pa[oldestIdx]};
    d.smoothed[0] = sx[oldestIdx]; d.smoothed[1] = sy[oldestIdx]; d.smoothed[2] = sa[oldestIdx];
    if (oldestIdx > 0) {  // :854-888
        int intent = motion_intent(s->transforms, raw, oldestIdx);
        float k = adaptive_strength(intent, raw);
        d.intent = intent;
        float g = intent == 1 ? 0.5f : intent == 2 ? 1.0f : intent == 3 ? 0.8f : k;
        for (float& v : diff) v *= g;
    }
    float dx = raw[0] + diff[0], dy = raw[1] + diff[1], da = raw[2] + diff[2];  // :890-894
    if (p.horizon_lock) da = 0.0f;  // :897-899
    float T[6] = {libm_cosf(da), -libm_sinf(da), dx, libm_sinf(da), libm_cosf(da), dy};  // :902-908
    memcpy(d.warp_matrix, T, sizeof T);

    if (s->fmt == VS_FMT_NV12) {
        vso_warp_affine_nv12(frame.data(), w, h, fstride, out, out_stride, T);
        return 1;
    }
    if (canvas_on(p, s->fmt)) {  // :1130-1134
        const float t[3] = {dx, dy, da};
        if (!s->canvas) s->canvas = canvas_new();
        canvas_apply(s->canvas, p, frame.data(), w, h, fstride, t, s->transforms, out, out_stride);
        return 1;
    }
    int b = p.border_size;
    if (b > 0 && !p.crop_n_zoom && p.border_type == VS_BORDER_FADE) {  // :914-978, :1069-1106
        // The "border mask" of the reference is drawn as two filled rectangles of which the second, over the whole image,
        // is 255 (:935-947, :1074-1084): the blend with the history and the history update act on EVERY pixel of the
        // padded frame, not on the border only.  Restated as written.
        int bw = w + 2 * b, bh = h + 2 * b;
        const size_t nb = (size_t)bw * bh * cn;
        std::vector<uint8_t> padded(nb);
        copy_make_border(frame.data(), w, h, fstride, cn, padded.data(), (size_t)bw * cn, b, VS_BORDER_BLACK);   // :927-932
        if (s->borderHistory.size() != nb) { s->borderHistory = padded; s->fadeFrameCount = 0; }                 // :917-926
        float alpha = p.fade_alpha;
        if (s->fadeFrameCount < p.fade_duration) {                                                               // :957-961
            alpha = alpha * (static_cast<float>(s->fadeFrameCount) / p.fade_duration);
            s->fadeFrameCount++;
        }
        // cv::addWeighted(borderHistory_, alpha, frameWithBorder, 1.0f - alpha, 0.0, ...) copied back under the all-255 mask
        vso_add_weighted_u8(s->borderHistory.data(), (double)alpha, padded.data(), (double)(1.0f - alpha), 0.0, padded.data(), nb);
        warp_affine(padded.data(), bw, bh, (size_t)bw * cn, cn, out, out_stride, T, g_threads);                  // :1056-1060
        const float updateRate = 0.1f;                                                                           // :1091-1098
        for (int y = 0; y < bh; y++) {
            uint8_t* hrow = &s->borderHistory[(size_t)y * bw * cn];
            const uint8_t* srow = out + (size_t)y * out_stride;
            for (int x = 0; x < bw * cn; x++)
                hrow[x] = static_cast<uint8_t>((1.0f - updateRate) * hrow[x] + updateRate * srow[x]);
        }
        return 1;
    }
    if (b > 0 && !p.crop_n_zoom) {  // :981-990
        int bw = w + 2 * b, bh = h + 2 * b;
        std::vector<uint8_t> padded((size_t)bw * bh * cn);
        copy_make_border(frame.data(), w, h, fstride, cn, padded.data(), (size_t)bw * cn, b, p.border_type);
        warp_affine(padded.data(), bw, bh, (size_t)bw * cn, cn, out, out_stride, T, g_threads);
        return 1;
    }
    if (p.crop_n_zoom && b > 0 && w - 2 * b > 0 && h - 2 * b > 0) {  // :1108-1124
        std::vector<uint8_t> st((size_t)w * h * cn);
        warp_affine(frame.data(), w, h, fstride, cn, st.data(), fstride, T, g_threads);
        int cw = w - 2 * b, ch = h - 2 * b;
        resize_linear_u8(&st[((size_t)b * w + b) * cn], cw, ch, fstride, cn, out, s->origW, s->origH, out_stride);
        return 1;
    }
    warp_affine(frame.data(), w, h, fstride, cn, out, out_stride, T, g_threads);  // :1056-1060
    return 1;
}

extern "C" {

vso_stab* vso_stab_create(const vs_params_c* p) {
    vso_stab* s = new vso_stab();
    s->p = *p;
    memset(&s->dbg, 0, sizeof s->dbg);
    s->dbg.out_index = -1;
    return s;
}
void vso_stab_destroy(vso_stab* s) {
    if (s && s->canvas) canvas_delete(s->canvas);
    delete s;
}
/* {canvas w, h, scale (float bits), regions, regions filled, temporal index of the last fill, window x, y} */
void vso_stab_canvas_info(const vso_stab* s, int32_t info[8]) {
    memset(info, 0, 8 * sizeof(int32_t));
    if (s->canvas) canvas_info(s->canvas, info);
}

void vso_stab_clean(vso_stab* s) {  // Stabilizer.cpp:221-256
    s->frameQueue.clear(); s->idxQueue.clear();
    s->transforms.clear(); s->path.clear();
    s->prevGray = Gray(); s->prevKeypoints.clear();
    s->firstFrame = true; s->nextFrameIndex = 0; s->frameW = s->frameH = 0; s->origW = s->origH = 0;
}

void vso_stab_out_size(const vso_stab* s, int w, int h, int* ow, int* oh) {
    int oW = s->origW > 0 ? s->origW : w, oH = s->origH > 0 ? s->origH : h;
    out_size(s->p, w, h, oW, oH, ow, oh, s->fmt);
}

int vso_stab_push(vso_stab* s, const uint8_t* data, int w, int h, size_t stride, int fmt,
                  uint8_t* out, size_t out_stride) {  // Stabilizer.cpp:258-392
    if (!data || w <= 0 || h <= 0) return 0;  // :263-265
    vs_params_c& p = s->p;
    if (p.crop_n_zoom && s->origW == 0) { s->origW = w; s->origH = h; }  // :267-269
    const int cn = fmt_cn(fmt);
    const int rows_total = fmt == VS_FMT_NV12 ? h * 3 / 2 : h;
    std::vector<uint8_t> copy((size_t)w * cn * rows_total);
    for (int y = 0; y < rows_total; y++) memcpy(&copy[(size_t)y * w * cn], data + (size_t)y * stride, (size_t)w * cn);
    (void)frame_bytes;
    s->dbg.out_index = -1;
    if (s->firstFrame) {  // :271-368
        s->frameW = w; s->frameH = h; s->fmt = fmt;
        analysis_gray(data, w, h, stride, fmt, 480, 270, s->prevGray);  // :277,:304-305
        std::vector<float> corners;
        int nc = 0;
        gftt(s->prevGray.d.data(), 480, 270, 480, p.max_corners, p.quality_level, p.min_distance,
             p.block_size, corners, &nc);  // :354-358
        s->prevKeypoints = corners;
        s->dbg.detected = 1; s->dbg.n_detected = (int)(corners.size() / 2);
        s->dbg.n_prev = 0; s->dbg.n_valid = 0;
        s->dbgDetected = corners; s->dbgGray = s->prevGray;
        s->dbgPrev.clear(); s->dbgCurr.clear(); s->dbgStatus.clear(); s->dbgInliers.clear();
        s->frameQueue.push_back(std::move(copy));
        s->idxQueue.push_back(0);
        s->firstFrame = false;
        s->nextFrameIndex = 1;
        return 0;
    }
    s->frameQueue.push_back(std::move(copy));  // :376-377
    s->idxQueue.push_back(s->nextFrameIndex);
    generate_transform(s, data, w, h, stride);  // :380
    int effectiveRadius = std::max(5, std::min(p.smoothing_radius, 35));  // :383
    if (s->idxQueue.size() < (size_t)effectiveRadius) { s->nextFrameIndex++; return 0; }
    int r = apply_next(s, out, out_stride);  // :389
    s->nextFrameIndex++;
    return r;
}

int vso_stab_flush(vso_stab* s, uint8_t* out, size_t out_stride) {  // :394-400
    if (s->frameQueue.empty()) return 0;
    s->dbg.out_index = -1;
    return apply_next(s, out, out_stride);
}

void vso_stab_get_debug(const vso_stab* s, vs_debug_frame* d) { *d = s->dbg; }

int vso_stab_get_debug_arrays(const vso_stab* s, float* prev_pts, float* curr_pts, uint8_t* status,
                              uint8_t* inliers, float* detected_pts, uint8_t* gray, int* aw, int* ah) {
    // The caller sizes its buffers by the counts of vso_stab_get_debug(): never more than that is written, whatever the
    // vectors hold (round 1 lost a GPU run to a stale dbgDetected copied over a 2-float numpy array: heap overflow in
    // the test process, SIGSEGV in the next library call).
    const vs_debug_frame& d = s->dbg;
    auto bounded = [](const auto& v, size_t cap, auto* out) { std::copy(v.begin(), v.begin() + std::min(v.size(), cap), out); };
    if (prev_pts) bounded(s->dbgPrev, (size_t)std::max(d.n_prev, 0) * 2, prev_pts);
    if (curr_pts) bounded(s->dbgCurr, (size_t)std::max(d.n_prev, 0) * 2, curr_pts);
    if (status) bounded(s->dbgStatus, (size_t)std::max(d.n_prev, 0), status);
    if (inliers) bounded(s->dbgInliers, (size_t)std::max(d.n_valid, 0), inliers);
    if (detected_pts) bounded(s->dbgDetected, (size_t)std::max(d.n_detected, 0) * 2, detected_pts);
    if (gray) std::copy(s->dbgGray.d.begin(), s->dbgGray.d.end(), gray);
    if (aw) *aw = s->dbgGray.w;
    if (ah) *ah = s->dbgGray.h;
    return 0;
}

void vso_set_threads(int n) { vso::g_threads = n < 1 ? 1 : n; }
}
