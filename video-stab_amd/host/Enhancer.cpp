// vs::Enhancer (include/video/Enhancer.h) on top of the C ABI (include/vs_stab.h).  Host glue only:
// cv::Mat in/out and Parameters -> vs_enh_params_c (/root/reference/src/Enhancer.cpp:138-239).
#include "video/Enhancer.h"

#include <cstdlib>
#include <mutex>
#include <stdexcept>
#include <string>

#include "vs_stab.h"

namespace vs {

namespace {

std::mutex g_enh_lock;
vs_enh *g_enh = nullptr;        // device tables + scratch, shared by all callers of the static function

vs_enh_params_c to_c(const Enhancer::Parameters &p) {
    vs_enh_params_c c;
    vs_enh_params_default(&c);
    c.brightness = p.brightness;
    c.contrast = p.contrast;
    c.enable_white_balance = p.enableWhiteBalance;
    c.wb_strength = p.wbStrength;
    c.enable_vibrance = p.enableVibrance;
    c.vibrance_strength = p.vibranceStrength;
    c.enable_unsharp = p.enableUnsharp;
    c.sharpness = p.sharpness;
    c.blur_sigma = p.blurSigma;
    c.enable_clahe = p.enableClahe;
    c.clahe_clip_limit = p.claheClipLimit;
    c.clahe_tile_grid_size = p.claheTileGridSize;
    c.enable_denoise = p.enableDenoise;
    c.denoise_strength = p.denoiseStrength;
    c.gamma = p.gamma;
    c.use_cuda = p.useCuda;
    return c;
}

}  // namespace

cv::Mat Enhancer::enhanceImage(const cv::Mat &input, const Parameters &params) {
    if (input.empty()) return cv::Mat();
    if (input.type() != CV_8UC3) throw std::runtime_error("vs::Enhancer: CV_8UC3 (BGR) frames only");
    std::lock_guard<std::mutex> guard(g_enh_lock);
    if (!g_enh) {
        const char *dev = std::getenv("VS_STAB_DEVICE");
        const int rc = vs_enh_create(dev ? std::atoi(dev) : 0, &g_enh);
        if (rc != VS_OK)
            throw std::runtime_error(std::string("vs::Enhancer: ") + vs_status_string(rc) + ": " + vs_last_error() +
                                     " (this build runs on the GPU only; there is no CPU fallback)");
    }
    const vs_enh_params_c c = to_c(params);
    cv::Mat out(input.rows, input.cols, CV_8UC3);
    const int rc = vs_enh_apply(g_enh, &c, input.data, input.cols, input.rows, input.step, out.data, out.step);
    if (rc != VS_OK)
        throw std::runtime_error(std::string("vs::Enhancer: ") + vs_status_string(rc) + ": " + vs_enh_last_error(g_enh));
    return out;
}

}  // namespace vs
