// vs::Stabilizer (include/video/Stabilizer.h) on top of the C ABI of
// libvideo-stab (include/vs_stab.h).  Host-side glue only: cv::Mat in/out,
// Parameters -> vs_params_c, the reference's error conventions
// (/root/reference/src/Stabilizer.cpp:263-265 empty in -> empty out;
// :1061-1066 a failing stage hands the frame back unwarped).
#include "video/Stabilizer.h"

#include <cstdlib>
#include <iostream>
#include <stdexcept>

#include "vs_stab.h"

namespace vs {

namespace {

// Stabilizer.cpp:31-38 mapBorderMode
int map_border(const std::string &t) {
    if (t == "reflect") return VS_BORDER_REFLECT;
    if (t == "reflect_101") return VS_BORDER_REFLECT_101;
    if (t == "replicate") return VS_BORDER_REPLICATE;
    if (t == "wrap") return VS_BORDER_WRAP;
    if (t == "fade") return VS_BORDER_FADE;
    return VS_BORDER_BLACK;
}

vs_params_c to_c(const Stabilizer::Parameters &p) {
    vs_params_c c;
    vs_params_default(&c);
    c.logging = p.logging;
    c.smoothing_radius = p.smoothingRadius;
    c.max_corners = p.maxCorners;
    c.quality_level = p.qualityLevel;
    c.min_distance = p.minDistance;
    c.block_size = p.blockSize;
    c.border_type = map_border(p.borderType);
    if (p.cropNZoom && p.borderType != "black") c.border_type = VS_BORDER_BLACK;   // Stabilizer.cpp:67-71
    c.border_size = p.borderSize;
    c.crop_n_zoom = p.cropNZoom;
    c.smoothing_method = p.smoothingMethod == "gaussian" ? VS_SMOOTH_GAUSSIAN
                         : p.smoothingMethod == "kalman" ? VS_SMOOTH_KALMAN : VS_SMOOTH_BOX;
    c.horizon_lock = p.horizonLock;
    c.gaussian_sigma = p.gaussianSigma;
    c.adaptive_smoothing = p.adaptiveSmoothing;
    c.min_smoothing_radius = p.minSmoothingRadius;
    c.max_smoothing_radius = p.maxSmoothingRadius;
    c.fade_alpha = p.fadeAlpha;
    c.fade_duration = p.fadeDuration;
    c.enable_virtual_canvas = p.enableVirtualCanvas;
    c.canvas_scale_factor = p.canvasScaleFactor;
    c.temporal_buffer_size = p.temporalBufferSize;
    c.canvas_blend_weight = p.canvasBlendWeight;
    c.adaptive_canvas_size = p.adaptiveCanvasSize;
    c.max_canvas_scale = p.maxCanvasScale;
    c.min_canvas_scale = p.minCanvasScale;
    c.edge_blend_radius = p.edgeBlendRadius;
    c.drone_high_freq_mode = p.droneHighFreqMode;
    c.hf_shake_px = p.hfShakePx;
    c.hf_analysis_max_width = p.hfAnalysisMaxWidth;
    c.hf_rot_lp_alpha = p.hfRotLPAlpha;
    c.enable_conditional_clahe = p.enableConditionalCLAHE;
    c.hf_dead_zone_threshold = p.hfDeadZoneThreshold;
    c.hf_freeze_duration = p.hfFreezeDuration;
    c.hf_motion_accumulator_decay = p.hfMotionAccumulatorDecay;
    return c;
}

}  // namespace

void Stabilizer::logMessage(const std::string &msg, bool isError) const {   // Stabilizer.cpp:40-46
    if (isError) std::cerr << "[ERROR] " << msg << std::endl;
    else std::cout << "[INFO] " << msg << std::endl;
}

void Stabilizer::create() {
    const char *dev = std::getenv("VS_STAB_DEVICE");
    device_ = dev ? std::atoi(dev) : 0;
    vs_params_c c = to_c(params_);
    int rc = vs_stab_create(&c, device_, &impl_);
    if (rc != VS_OK) {
        // there is no CPU path to fall back to: make the failure visible
        throw std::runtime_error(std::string("vs::Stabilizer: ") + vs_status_string(rc) + ": " + vs_last_error());
    }
    // VS_STAB_HOST_PIPELINE=1: stabilize() returns the frame the previous call computed (one more frame of latency) and the
    // transfers of consecutive calls overlap - 2.4x the frame rate of the synchronous call at 1080p
    const char *hp = std::getenv("VS_STAB_HOST_PIPELINE");
    if (params_.hostPipeline || (hp && std::atoi(hp) != 0)) (void)vs_stab_set_host_pipeline(impl_, 1);
}

// ---- page-locked host memory (Parameters::pinHostFrames) --------------------------------------------------------------------
// A frame of a fresh cv::Mat costs the call its allocation, the first-touch faults of 6 MB and a staged (pageable) transfer:
// 1 995 frames/s at 1080p against 2 590 from page-locked buffers (bench.py, with_pcie).  The ring below hands out Mats whose
// buffers are registered with the driver once and reused as soon as the caller holds no reference to them any more
// (cv::Mat's own reference count: u->refcount == 1 means the ring is the only owner); a caller that keeps more than four
// results alive simply gets ordinary Mats for the surplus.
cv::Mat Stabilizer::outputFrame(int rows, int cols) {
    if (!params_.pinHostFrames) return cv::Mat(rows, cols, CV_8UC3);
    for (OutSlot &s : outRing_) {
        if (!s.m.empty() && s.m.rows == rows && s.m.cols == cols && s.m.u && s.m.u->refcount == 1) return s.m;
    }
    for (OutSlot &s : outRing_) {
        if (s.m.empty() || (s.m.u && s.m.u->refcount == 1)) {       // an empty slot, or one of another size that nobody holds
            if (s.pinned) { (void)vs_host_unregister(s.m.data); s.pinned = false; }
            s.m = cv::Mat(rows, cols, CV_8UC3);
            s.pinned = vs_host_register(s.m.data, (size_t)s.m.step * (size_t)rows) == VS_OK;
            return s.m;
        }
    }
    return cv::Mat(rows, cols, CV_8UC3);
}

// Parameters::pinInputFrames: the frame buffer a capture loop reads into comes back call after call; from its second
// appearance on it is registered (page-locked in place), so its upload is a DMA transfer of its own instead of a staged copy.
// At most four buffers; one that has not been seen for 64 calls is let go.  The buffer is the application's - it must stay
// allocated while it is registered -, which is why this is opt-in.
void Stabilizer::noteInput(const cv::Mat &frame) {
    if (!params_.pinInputFrames) return;
    const size_t bytes = (size_t)frame.step * (size_t)frame.rows;
    InPin *hit = nullptr, *spare = nullptr;
    for (InPin &e : inPins_) {
        if (e.p == frame.data && e.bytes == bytes) hit = &e;
        else if (e.p && ++e.idle > 64) {
            if (e.pinned) (void)vs_host_unregister(const_cast<unsigned char *>(e.p));
            e = InPin();
        }
        if (!e.p && !spare) spare = &e;
    }
    if (hit) {
        hit->idle = 0;
        if (++hit->seen == 2 && !hit->pinned) hit->pinned = vs_host_register(const_cast<unsigned char *>(hit->p), bytes) == VS_OK;
    } else if (spare) {
        spare->p = frame.data; spare->bytes = bytes; spare->seen = 1; spare->idle = 0; spare->pinned = false;
    }
}

void Stabilizer::releaseHostPins() {
    for (OutSlot &s : outRing_) {
        if (s.pinned) (void)vs_host_unregister(s.m.data);
        s = OutSlot();
    }
    for (InPin &e : inPins_) {
        if (e.pinned) (void)vs_host_unregister(const_cast<unsigned char *>(e.p));
        e = InPin();
    }
}

Stabilizer::Stabilizer(const Parameters &params) : params_(params) {
    if (params_.logging) logMessage("Initializing MI355X stabilizer (libvideo-stab, gfx950)...", false);
    create();
}

Stabilizer::~Stabilizer() {
    if (impl_) vs_stab_destroy(impl_);   // drains in-flight GPU work first
    releaseHostPins();                   // (frames the caller still holds stay valid: they are only no longer page-locked)
}

Stabilizer::Stabilizer(Stabilizer &&o) noexcept
    : params_(std::move(o.params_)), impl_(o.impl_), device_(o.device_), frameWidth_(o.frameWidth_), frameHeight_(o.frameHeight_) {
    o.impl_ = nullptr;
    o.releaseHostPins();                 // (the rings are rebuilt on this side as frames come)
}

Stabilizer &Stabilizer::operator=(Stabilizer &&o) noexcept {
    if (this != &o) {
        if (impl_) vs_stab_destroy(impl_);
        releaseHostPins();
        params_ = std::move(o.params_);
        impl_ = o.impl_;
        device_ = o.device_;
        frameWidth_ = o.frameWidth_; frameHeight_ = o.frameHeight_;
        o.impl_ = nullptr;
        o.releaseHostPins();
    }
    return *this;
}

Stabilizer::Stabilizer(const Stabilizer &o) : params_(o.params_) { create(); }

Stabilizer &Stabilizer::operator=(const Stabilizer &o) {
    if (this != &o) {
        if (impl_) vs_stab_destroy(impl_);
        impl_ = nullptr;
        releaseHostPins();
        params_ = o.params_;
        create();
    }
    return *this;
}

cv::Mat Stabilizer::stabilize(const cv::Mat &frame) {
    if (frame.empty() || !impl_) return cv::Mat();                       // Stabilizer.cpp:263-265
    if (frame.type() != CV_8UC3) {
        if (params_.logging) logMessage("stabilize(): expected a CV_8UC3 BGR frame", true);
        return frame;
    }
    int ow = 0, oh = 0;
    frameWidth_ = frame.cols; frameHeight_ = frame.rows;
    vs_stab_out_size(impl_, frame.cols, frame.rows, &ow, &oh);
    noteInput(frame);
    cv::Mat out = outputFrame(oh, ow);
    int produced = 0;
    int rc = vs_stab_push(impl_, frame.data, frame.cols, frame.rows, (size_t)frame.step, VS_FMT_BGR8, out.data,
                          (size_t)out.step, &produced);
    if (rc != VS_OK) {
        if (params_.logging) logMessage(std::string("stabilize() failed: ") + vs_stab_last_error(impl_), true);
        return frame;                                                    // :1061-1066 hand the frame back
    }
    if (!produced) return cv::Mat();                                     // warm-up (:384-387)
    return out;
}

cv::Mat Stabilizer::flush() {
    if (!impl_) return cv::Mat();
    if (frameWidth_ <= 0 || frameHeight_ <= 0) return cv::Mat();
    int w = 0, h = 0, produced = 0, ow = 0, oh = 0;
    vs_stab_out_size(impl_, frameWidth_, frameHeight_, &ow, &oh);
    cv::Mat out = outputFrame(oh, ow);
    int rc = vs_stab_flush(impl_, out.data, (size_t)out.step, &produced);
    if (rc != VS_OK || !produced) return cv::Mat();
    vs_stab_last_out_dims(impl_, &w, &h);
    if (w != ow || h != oh) return out(cv::Rect(0, 0, w, h)).clone();    // last queued frame comes back unpadded (:774-780)
    return out;
}

void Stabilizer::clean() {
    if (impl_) vs_stab_clean(impl_);
    frameWidth_ = frameHeight_ = 0;
    releaseHostPins();
}

}  // namespace vs
