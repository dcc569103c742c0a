// Config layer: config.yaml of the reference's example mains as one
// reusable loader.  Every main of the reference repeats the same block of
//     cv::FileStorage fs(file, READ);  cv::FileNode n = fs["stabilizer"];  n["key"] >> params.field; ...
// (examples/vs.cpp:50-168, examples/vsg.cpp:1007-1112) and the same st_mtime poll for hot
// reload (vs.cpp:199-200,381-394).  Here:
//   vs::ConfigFile / vs::ConfigNode  read-only stand-ins for cv::FileStorage / cv::FileNode with the
//                                    same call forms (isOpened, operator[], empty, >>) and the same
//                                    conversions; no OpenCV needed
//   vs::read(node, Parameters&)      the key -> field tables for Mode, Enhancer, RollCorrection and
//                                    Stabilizer parameters
//   vs::AppConfig / vs::loadConfig   the whole file in one call
//   vs::ConfigWatcher                the st_mtime poll
// Header-only over the C ABI (vs_config_* in vs_stab.h, implemented in libvideo-stab).
#ifndef VS_VIDEO_CONFIG_H
#define VS_VIDEO_CONFIG_H

#include <cstdint>
#include <memory>
#include <string>

#include "../vs_stab.h"
#include "Enhancer.h"
#include "Mode.h"
#include "RollCorrection.h"
#include "Stabilizer.h"

namespace vs {

class ConfigNode {
public:
    ConfigNode() = default;
    ConfigNode(std::shared_ptr<vs_config> c, std::string path) : c_(std::move(c)), path_(std::move(path)) {}

    /// Child of a map; a node that does not exist is returned as an empty node, as cv::FileNode does.
    ConfigNode operator[](const std::string& key) const { return ConfigNode(c_, path_.empty() ? key : path_ + "." + key); }
    ConfigNode operator[](const char* key) const { return (*this)[std::string(key)]; }

    int kind() const { return c_ ? vs_config_kind(c_.get(), path_.c_str()) : VS_CFG_NONE; }
    bool empty() const { return kind() == VS_CFG_NONE; }
    bool isNone() const { return empty(); }
    bool isInt() const { return kind() == VS_CFG_INT; }
    bool isReal() const { return kind() == VS_CFG_REAL; }
    bool isString() const { return kind() == VS_CFG_STRING; }
    bool isMap() const { return kind() == VS_CFG_MAP; }
    bool isSeq() const { return kind() == VS_CFG_SEQ; }
    size_t size() const { return c_ ? (size_t)vs_config_size(c_.get(), path_.c_str()) : 0; }

    operator int() const { int32_t v = 0; if (c_) vs_config_get_int(c_.get(), path_.c_str(), &v); return v; }
    operator float() const { float v = 0; if (c_) vs_config_get_float(c_.get(), path_.c_str(), &v); return v; }
    operator double() const { double v = 0; if (c_) vs_config_get_double(c_.get(), path_.c_str(), &v); return v; }
    operator std::string() const { return string(); }
    std::string string() const {
        if (!c_) return std::string();
        std::string buf(256, '\0');
        // the C entry point refuses a buffer that is too small: grow until the value fits
        while (vs_config_get_string(c_.get(), path_.c_str(), &buf[0], buf.size()) != VS_OK && buf.size() < (1u << 24))
            buf.assign(buf.size() * 4, '\0');
        return std::string(buf.c_str());
    }
    /// Element of a sequence of numbers, e.g. node["roi"].at(2).
    double at(int index) const { double v = 0; if (c_) vs_config_seq_get_double(c_.get(), path_.c_str(), index, &v); return v; }

private:
    std::shared_ptr<vs_config> c_;
    std::string path_;
};

// `node >> value` with cv::FileNode's conversions: an absent node gives 0 / 0.0 / "" / false.
inline void operator>>(const ConfigNode& n, int& v) { v = (int)n; }
inline void operator>>(const ConfigNode& n, float& v) { v = (float)n; }
inline void operator>>(const ConfigNode& n, double& v) { v = (double)n; }
inline void operator>>(const ConfigNode& n, bool& v) { v = (int)n != 0; }
inline void operator>>(const ConfigNode& n, std::string& v) { v = n.string(); }

class ConfigFile {
public:
    ConfigFile() = default;
    explicit ConfigFile(const std::string& path) { open(path); }

    /// false (and isOpened() false) when the file is missing or malformed; error() says why.
    bool open(const std::string& path) {
        vs_config* raw = nullptr;
        c_.reset();
        if (vs_config_open(path.c_str(), &raw) != VS_OK) { error_ = vs_last_error(); return false; }
        c_.reset(raw, vs_config_close);
        error_.clear();
        return true;
    }
    bool parse(const std::string& text) {
        vs_config* raw = nullptr;
        c_.reset();
        if (vs_config_parse(text.data(), text.size(), &raw) != VS_OK) { error_ = vs_last_error(); return false; }
        c_.reset(raw, vs_config_close);
        error_.clear();
        return true;
    }
    bool isOpened() const { return (bool)c_; }
    void release() { c_.reset(); }
    const std::string& error() const { return error_; }
    ConfigNode root() const { return ConfigNode(c_, ""); }
    ConfigNode operator[](const std::string& key) const { return ConfigNode(c_, key); }
    ConfigNode operator[](const char* key) const { return ConfigNode(c_, key); }

private:
    std::shared_ptr<vs_config> c_;
    std::string error_;
};

namespace detail {
// keep = true: a key the file does not name leaves the field alone (the loader's default);
// keep = false: it is assigned 0 / "" like `node["k"] >> field` in the reference's mains.
template <class T>
inline void take(const ConfigNode& sec, const char* key, T& field, bool keep) {
    const ConfigNode n = sec[key];
    if (keep && n.empty()) return;
    n >> field;
}
template <class E>
inline void take_enum(const ConfigNode& sec, const char* key, E& field, bool keep) {
    const ConfigNode n = sec[key];
    if (n.empty()) return;                 // the mains preset the default before reading (vsg.cpp:1040,1063)
    (void)keep;
    field = static_cast<E>((int)n);
}
}  // namespace detail

/// "mode" section (vs.cpp:59-68).  Returns false and changes nothing when the section is absent.
inline bool read(const ConfigNode& sec, Mode::Parameters& p, bool keepMissing = true) {
    if (sec.empty()) return false;
    using detail::take;
    take(sec, "width", p.width, keepMissing);
    take(sec, "height", p.height, keepMissing);
    take(sec, "optimize_fps", p.optimizeFps, keepMissing);
    take(sec, "use_cuda", p.useCuda, keepMissing);
    take(sec, "enhancer_enabled", p.enhancerEnabled, keepMissing);
    take(sec, "roll_correction_enabled", p.rollCorrectionEnabled, keepMissing);
    take(sec, "stabilizer_enabled", p.stabilizationEnabled, keepMissing);
    take(sec, "tracker_enabled", p.trackerEnabled, keepMissing);
    return true;
}

/// "enhancer" section (vs.cpp:72-104).
inline bool read(const ConfigNode& sec, Enhancer::Parameters& p, bool keepMissing = true) {
    if (sec.empty()) return false;
    using detail::take;
    take(sec, "brightness", p.brightness, keepMissing);
    take(sec, "contrast", p.contrast, keepMissing);
    take(sec, "enable_white_balance", p.enableWhiteBalance, keepMissing);
    take(sec, "wb_strength", p.wbStrength, keepMissing);
    take(sec, "enable_vibrance", p.enableVibrance, keepMissing);
    take(sec, "vibrance_strength", p.vibranceStrength, keepMissing);
    take(sec, "enable_unsharp", p.enableUnsharp, keepMissing);
    take(sec, "sharpness", p.sharpness, keepMissing);
    take(sec, "blur_sigma", p.blurSigma, keepMissing);
    take(sec, "enable_denoise", p.enableDenoise, keepMissing);
    take(sec, "denoise_strength", p.denoiseStrength, keepMissing);
    take(sec, "gamma", p.gamma, keepMissing);
    take(sec, "enable_clahe", p.enableClahe, keepMissing);
    take(sec, "clahe_clip_limit", p.claheClipLimit, keepMissing);
    take(sec, "clahe_tile_grid_size", p.claheTileGridSize, keepMissing);
    take(sec, "use_cuda", p.useCuda, keepMissing);
    return true;
}

/// "roll_correction" section (vs.cpp:108-120).
inline bool read(const ConfigNode& sec, RollCorrection::Parameters& p, bool keepMissing = true) {
    if (sec.empty()) return false;
    using detail::take;
    take(sec, "scale_factor", p.scaleFactor, keepMissing);
    take(sec, "canny_threshold_low", p.cannyThresholdLow, keepMissing);
    take(sec, "canny_threshold_high", p.cannyThresholdHigh, keepMissing);
    take(sec, "canny_aperture", p.cannyAperture, keepMissing);
    take(sec, "hough_rho", p.houghRho, keepMissing);
    take(sec, "hough_theta", p.houghTheta, keepMissing);
    take(sec, "hough_threshold", p.houghThreshold, keepMissing);
    take(sec, "angle_smoothing_alpha", p.angleSmoothingAlpha, keepMissing);
    take(sec, "angle_decay", p.angleDecay, keepMissing);
    take(sec, "angle_filter_min", p.angleFilterMin, keepMissing);
    take(sec, "angle_filter_max", p.angleFilterMax, keepMissing);
    return true;
}

/// "stabilizer" section: the union of what the mains read (vs.cpp:124-131 reads six keys,
/// vsg.cpp:1007-1103 all of them).
inline bool read(const ConfigNode& sec, Stabilizer::Parameters& p, bool keepMissing = true) {
    if (sec.empty()) return false;
    using detail::take;
    using detail::take_enum;
    take(sec, "smoothing_radius", p.smoothingRadius, keepMissing);
    take(sec, "border_type", p.borderType, keepMissing);
    take(sec, "border_size", p.borderSize, keepMissing);
    take(sec, "crop_n_zoom", p.cropNZoom, keepMissing);
    take(sec, "logging", p.logging, keepMissing);
    take(sec, "use_cuda", p.useCuda, keepMissing);
    take(sec, "smoothing_method", p.smoothingMethod, keepMissing);
    take(sec, "gaussian_sigma", p.gaussianSigma, keepMissing);
    take(sec, "stage_one_radius", p.stageOneRadius, keepMissing);
    take(sec, "stage_two_radius", p.stageTwoRadius, keepMissing);
    take(sec, "use_temporal_filtering", p.useTemporalFiltering, keepMissing);
    take(sec, "temporal_window_size", p.temporalWindowSize, keepMissing);
    take(sec, "adaptive_smoothing", p.adaptiveSmoothing, keepMissing);
    take(sec, "min_smoothing_radius", p.minSmoothingRadius, keepMissing);
    take(sec, "max_smoothing_radius", p.maxSmoothingRadius, keepMissing);
    take(sec, "max_corners", p.maxCorners, keepMissing);
    take(sec, "quality_level", p.qualityLevel, keepMissing);
    take(sec, "min_distance", p.minDistance, keepMissing);
    take(sec, "block_size", p.blockSize, keepMissing);
    take(sec, "outlier_threshold", p.outlierThreshold, keepMissing);
    take(sec, "motion_prediction", p.motionPrediction, keepMissing);
    take(sec, "intentional_motion_threshold", p.intentionalMotionThreshold, keepMissing);
    take_enum(sec, "jitter_frequency", p.jitterFrequency, keepMissing);
    take(sec, "separate_translation_rotation", p.separateTranslationRotation, keepMissing);
    take(sec, "deep_stabilization", p.deepStabilization, keepMissing);
    take(sec, "model_path", p.modelPath, keepMissing);
    take(sec, "roll_compensation", p.rollCompensation, keepMissing);
    take(sec, "roll_compensation_factor", p.rollCompensationFactor, keepMissing);
    take(sec, "use_roi", p.useROI, keepMissing);
    if (p.useROI) {                              // vsg.cpp:1052-1059
        int x = 0, y = 0, w = 0, h = 0;
        sec["roi_x"] >> x; sec["roi_y"] >> y; sec["roi_width"] >> w; sec["roi_height"] >> h;
        p.roi = cv::Rect(x, y, w, h);
    }
    take(sec, "horizon_lock", p.horizonLock, keepMissing);
    take_enum(sec, "feature_detector_type", p.featureDetector, keepMissing);
    take(sec, "fast_threshold", p.fastThreshold, keepMissing);
    take(sec, "orb_features", p.orbFeatures, keepMissing);
    take(sec, "border_scale_factor", p.borderScaleFactor, keepMissing);
    take(sec, "motion_threshold_low", p.motionThresholdLow, keepMissing);
    take(sec, "motion_threshold_high", p.motionThresholdHigh, keepMissing);
    take(sec, "fadeDuration", p.fadeDuration, keepMissing);
    take(sec, "fadeAlpha", p.fadeAlpha, keepMissing);
    take(sec, "use_imu_data", p.useImuData, keepMissing);
    take(sec, "enable_virtual_canvas", p.enableVirtualCanvas, keepMissing);
    take(sec, "canvas_scale_factor", p.canvasScaleFactor, keepMissing);
    take(sec, "temporal_buffer_size", p.temporalBufferSize, keepMissing);
    take(sec, "canvas_blend_weight", p.canvasBlendWeight, keepMissing);
    take(sec, "adaptive_canvas_size", p.adaptiveCanvasSize, keepMissing);
    take(sec, "max_canvas_scale", p.maxCanvasScale, keepMissing);
    take(sec, "min_canvas_scale", p.minCanvasScale, keepMissing);
    take(sec, "preserve_edge_quality", p.preserveEdgeQuality, keepMissing);
    take(sec, "edge_blend_radius", p.edgeBlendRadius, keepMissing);
    take(sec, "drone_high_freq_mode", p.droneHighFreqMode, keepMissing);
    take(sec, "hf_shake_px", p.hfShakePx, keepMissing);
    take(sec, "hf_analysis_max_width", p.hfAnalysisMaxWidth, keepMissing);
    take(sec, "hf_rot_lp_alpha", p.hfRotLPAlpha, keepMissing);
    take(sec, "enable_conditional_clahe", p.enableConditionalCLAHE, keepMissing);
    take(sec, "hf_dead_zone_threshold", p.hfDeadZoneThreshold, keepMissing);
    take(sec, "hf_freeze_duration", p.hfFreezeDuration, keepMissing);
    take(sec, "hf_motion_accumulator_decay", p.hfMotionAccumulatorDecay, keepMissing);
    return true;
}

/// Everything the mains pull out of config.yaml for the stages of this library.
struct AppConfig {
    std::string videoSource;
    Mode::Parameters mode;
    Enhancer::Parameters enhancer;
    RollCorrection::Parameters roll;
    Stabilizer::Parameters stabilizer;
};

/// Reads `path` into `cfg`; sections the file lacks keep what `cfg` holds.  false when the file cannot be
/// read or parsed (`error`, if given, says why) — `cfg` is untouched then, so a bad edit of a watched file
/// does not take a running pipeline down.
inline bool loadConfig(const std::string& path, AppConfig& cfg, bool keepMissing = true, std::string* error = nullptr) {
    ConfigFile fs(path);
    if (!fs.isOpened()) { if (error) *error = fs.error(); return false; }
    if (!fs["video_source"].empty()) fs["video_source"] >> cfg.videoSource;
    read(fs["mode"], cfg.mode, keepMissing);
    read(fs["enhancer"], cfg.enhancer, keepMissing);
    read(fs["roll_correction"], cfg.roll, keepMissing);
    read(fs["stabilizer"], cfg.stabilizer, keepMissing);
    return true;
}

/// The st_mtime poll of the mains (vs.cpp:199-200,381-394): changed() is true once per modification.
class ConfigWatcher {
public:
    explicit ConfigWatcher(std::string path) : path_(std::move(path)) { vs_config_mtime(path_.c_str(), &seen_); }
    bool changed() {
        int64_t now = 0;
        if (vs_config_mtime(path_.c_str(), &now) != VS_OK || now == seen_) return false;
        seen_ = now;
        return true;
    }
    const std::string& path() const { return path_; }

private:
    std::string path_;
    int64_t seen_ = 0;
};

}  // namespace vs

#endif
