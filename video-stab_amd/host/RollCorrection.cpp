// vs::RollCorrection and vs::AutoZoomCrop (include/video/RollCorrection.h, AutoZoomCrop.h) on
// top of the C ABI (include/vs_stab.h).  Host glue only: cv::Mat in/out, Parameters ->
// vs_roll_params_c, and the reference's conventions for bad input
// (/root/reference/src/RollCorrection.cpp:21-23 empty in -> clone out;
//  /root/reference/src/AutoZoomCrop.cpp:105-107 likewise).
#include "video/RollCorrection.h"
#include "video/AutoZoomCrop.h"

#include <cstdlib>
#include <iostream>
#include <mutex>
#include <stdexcept>
#include <vector>

#include "vs_stab.h"

namespace vs {

namespace {

int env_device() {
    const char *e = std::getenv("VS_STAB_DEVICE");
    return e ? std::atoi(e) : 0;
}

std::mutex g_lock;
vs_roll *g_roll = nullptr;      // process-wide, like the reference's function statics
vs_azc *g_azc = nullptr;

vs_roll_params_c to_c(const RollCorrection::Parameters &p) {
    vs_roll_params_c c;
    vs_roll_params_default(&c);
    c.scale_factor = p.scaleFactor;
    c.canny_threshold_low = p.cannyThresholdLow;
    c.canny_threshold_high = p.cannyThresholdHigh;
    c.canny_aperture = p.cannyAperture;
    c.hough_rho = p.houghRho;
    c.hough_theta = p.houghTheta;
    c.hough_threshold = p.houghThreshold;
    c.angle_filter_min = p.angleFilterMin;
    c.angle_filter_max = p.angleFilterMax;
    c.angle_smoothing_alpha = p.angleSmoothingAlpha;
    c.angle_decay = p.angleDecay;
    c.max_angle_change_deg = p.maxAngleChangeDeg;
    return c;
}

}  // namespace

cv::Mat RollCorrection::autoCorrectRoll(const cv::Mat &input, const Parameters &params) {
    if (input.empty()) return input.clone();
    if (input.type() != CV_8UC3) throw std::runtime_error("vs::RollCorrection: CV_8UC3 (BGR) frames only");
    std::lock_guard<std::mutex> guard(g_lock);
    const vs_roll_params_c c = to_c(params);
    if (!g_roll) {
        const int rc = vs_roll_create(&c, env_device(), &g_roll);
        if (rc != VS_OK)
            throw std::runtime_error(std::string("vs::RollCorrection: ") + vs_status_string(rc) + ": " + vs_last_error() +
                                     " (this build runs on the GPU only; there is no CPU fallback)");
    } else {
        const int rc = vs_roll_set_params(g_roll, &c);
        if (rc != VS_OK) throw std::runtime_error(std::string("vs::RollCorrection: ") + vs_roll_last_error(g_roll));
    }
    cv::Mat out(input.rows, input.cols, CV_8UC3);
    const int rc = vs_roll_correct(g_roll, input.data, input.cols, input.rows, input.step, out.data, out.step);
    if (rc != VS_OK) {
        std::cerr << "vs::RollCorrection: " << vs_status_string(rc) << ": " << vs_roll_last_error(g_roll) << std::endl;
        return input.clone();
    }
    return out;
}

cv::Mat RollCorrection::autoCorrectRoll(const cv::Mat &input) { return autoCorrectRoll(input, Parameters()); }

cv::Mat AutoZoomCrop::autoZoomCrop(const cv::Mat &corrected, double /*marginPercent*/) {
    if (corrected.empty()) return corrected.clone();
    const int cn = corrected.channels();
    if (corrected.type() != CV_8UC3 && corrected.type() != CV_8UC1)
        throw std::runtime_error("vs::AutoZoomCrop: CV_8UC3 or CV_8UC1 frames only");
    std::lock_guard<std::mutex> guard(g_lock);
    if (!g_azc) {
        const int rc = vs_azc_create(env_device(), &g_azc);
        if (rc != VS_OK)
            throw std::runtime_error(std::string("vs::AutoZoomCrop: ") + vs_status_string(rc) + ": " + vs_last_error() +
                                     " (this build runs on the GPU only; there is no CPU fallback)");
    }
    const size_t px = std::max((size_t)corrected.cols * corrected.rows, (size_t)640 * 360);
    std::vector<unsigned char> buf(px * cn);
    int ow = 0, oh = 0;
    const int rc = vs_azc_apply(g_azc, corrected.data, corrected.cols, corrected.rows, corrected.step, cn, buf.data(), &ow, &oh);
    if (rc != VS_OK) {
        std::cerr << "vs::AutoZoomCrop: " << vs_status_string(rc) << ": " << vs_azc_last_error(g_azc) << std::endl;
        return corrected.clone();
    }
    cv::Mat out(oh, ow, corrected.type());
    for (int y = 0; y < oh; y++) std::memcpy(out.ptr(y), buf.data() + (size_t)y * ow * cn, (size_t)ow * cn);
    return out;
}

}  // namespace vs
