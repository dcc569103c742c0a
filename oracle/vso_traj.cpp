// CPU ORACLE (test infrastructure) - trajectory smoothing and motion-intent
// logic restated from /root/reference/src/Stabilizer.cpp (lines cited).
#include "vso_internal.h"

#include <algorithm>

namespace vso {

// Stabilizer.cpp:1139-1172 boxFilterConvolve.  `radius_param` is the value of
// params_.smoothingRadius at the time of the call (the caller overwrites it
// with calculateAdaptiveRadius(), :809-813).
std::vector<float> box_filter(const std::vector<float>& path, int radius_param, bool drone) {
    if (path.empty()) return {};
    int r = drone ? std::max(10, std::min(radius_param, 50)) : std::max(2, std::min(radius_param, 8));
    if (path.size() <= (size_t)r) return path;
    std::vector<float> result(path.size());
    for (size_t i = 0; i < path.size(); i++) {
        float sum = 0.0f;
        int count = 0;
        int start = std::max(0, (int)i - r);
        int end = std::min((int)path.size() - 1, (int)i + r);
        for (int j = start; j <= end; j++) { sum += path[j]; count++; }
        result[i] = sum / count;
    }
    return result;
}

// Stabilizer.cpp:1364-1413 gaussianFilterConvolve.  The reference reads out of
// bounds when path.size() <= kernel/2 (UB); the oracle clamps the
// index into the path for that regime.
std::vector<float> gaussian_filter(const std::vector<float>& path, float sigma) {
    if (path.empty()) return {};
    int kernelSize = std::max(3, (int)std::ceil(6 * sigma));
    if (kernelSize % 2 == 0) kernelSize++;
    std::vector<float> kernel(kernelSize);
    float sum = 0.0f;
    int center = kernelSize / 2;
    for (int i = 0; i < kernelSize; i++) {
        float x = (float)(i - center);
        kernel[i] = std::exp(-(x * x) / (2 * sigma * sigma));
        sum += kernel[i];
    }
    for (float& k : kernel) k /= sum;
    int n = (int)path.size();
    auto at = [&](int i) { return path[std::min(std::max(i, 0), n - 1)]; };
    std::vector<float> padded(path.size() + 2 * center);
    for (int i = 0; i < center; i++) padded[i] = at(center - i);
    for (int i = 0; i < n; i++) padded[center + i] = path[i];
    for (int i = 0; i < center; i++) padded[center + n + i] = at(n - 1 - i);
    std::vector<float> result(path.size());
    for (int i = 0; i < n; i++) {
        float s = 0.0f;
        for (int j = 0; j < kernelSize; j++) s += padded[i + j] * kernel[j];
        result[i] = s;
    }
    return result;
}

// Stabilizer.cpp:1416-1458 kalmanFilterSmooth: cv::KalmanFilter(2,1,0) float,
// A=[1 1;0 1], H=[1 0], Q=0.01 I, R=0.1, x0=(path0,0), P0=0; forward only.
// Restated as scalar float recursions in the natural operation order
// (cv::gemm's 2x2 float special cases round the same way; the 1x1 SVD solve
// is taken as a float division).
std::vector<float> kalman_filter(const std::vector<float>& path) {
    if (path.empty()) return {};
    const float q = 0.01f, r = 0.1f;
    float x0 = path[0], x1 = 0.f;
    float P00 = 0, P01 = 0, P10 = 0, P11 = 0;
    std::vector<float> result(path.size());
    result[0] = path[0];
    for (size_t i = 1; i < path.size(); i++) {
        // predict: x' = A x ; P' = A P A^T + Q
        float xp0 = x0 + x1, xp1 = x1;
        float t00 = P00 + P10, t01 = P01 + P11, t10 = P10, t11 = P11;  // A*P
        float Q00 = (t00 + t01) + q, Q01 = t01, Q10 = t10 + t11, Q11 = t11 + q;
        // correct
        // temp2 = H*P' = [Q00 Q01]; temp3 = temp2*H^T + R; temp4 = temp2/temp3; gain = temp4^T
        float S = Q00 + r;
        float K0 = Q00 / S, K1 = Q01 / S;
        float innov = path[i] - xp0;
        x0 = xp0 + K0 * innov;
        x1 = xp1 + K1 * innov;
        // P = P' - K * temp2
        P00 = Q00 - K0 * Q00; P01 = Q01 - K0 * Q01;
        P10 = Q10 - K1 * Q00; P11 = Q11 - K1 * Q01;
        result[i] = x0;
    }
    return result;
}

// Stabilizer.cpp:1637-1673 calculateAdaptiveRadius
int adaptive_radius(const std::vector<float>& px, const std::vector<float>& py,
                    const std::vector<float>& pa, int smoothing_radius) {
    if (px.size() < 10) return smoothing_radius;
    float varianceX = 0, varianceY = 0, varianceA = 0;
    float meanX = 0, meanY = 0, meanA = 0;
    size_t start = std::max(0, (int)px.size() - 20);
    size_t count = px.size() - start;
    for (size_t i = start; i < px.size(); i++) { meanX += px[i]; meanY += py[i]; meanA += pa[i]; }
    meanX /= count; meanY /= count; meanA /= count;
    for (size_t i = start; i < px.size(); i++) {
        varianceX += (px[i] - meanX) * (px[i] - meanX);
        varianceY += (py[i] - meanY) * (py[i] - meanY);
        varianceA += (pa[i] - meanA) * (pa[i] - meanA);
    }
    varianceX /= count; varianceY /= count; varianceA /= count;
    float totalVariance = std::sqrt(varianceX + varianceY + varianceA * 1000);
    return (int)std::max(5.0f, std::min(25.0f, totalVariance * 2.0f));
}

// Stabilizer.cpp:1750-1780
static float variance(const std::vector<float>& v) {
    if (v.empty()) return 0.0f;
    float mean = 0.0f;
    for (float x : v) mean += x;
    mean /= v.size();
    float var = 0.0f;
    for (float x : v) { float d = x - mean; var += d * d; }
    var /= v.size();
    return var;
}
static float consistency(const std::vector<float>& v) {
    if (v.size() < 2) return 0.0f;
    float var = variance(v);
    float mean = 0.0f;
    for (float x : v) mean += x;
    mean /= v.size();
    if (mean == 0.0f) return 0.0f;
    float c = 1.0f / (1.0f + (var / (mean * mean)));
    return std::max(0.0f, std::min(1.0f, c));
}

// Stabilizer.cpp:1676-1719 analyzeMotionIntent.  Returns 0 NORMAL,
// 1 DELIBERATE_PAN, 2 SHAKE_REMOVAL, 3 FOLLOW_ACTION (enum order Stabilizer.h:52-57)
int motion_intent(const std::vector<float>& tr, const float motion[3], int frameIndex) {
    int n = (int)(tr.size() / 3);
    float magnitude = std::sqrt(motion[0] * motion[0] + motion[1] * motion[1]);
    float angularVel = std::abs(motion[2]) * 180.0f / M_PI * 30.0f;
    if (n >= 15) {
        std::vector<float> mags, dirs;
        for (int i = std::max(0, frameIndex - 15); i < frameIndex; i++) {
            if (i < n) {
                float t0 = tr[3 * i], t1 = tr[3 * i + 1];
                mags.push_back(std::sqrt(t0 * t0 + t1 * t1));
                dirs.push_back(libm_atan2f(t1, t0));
            }
        }
        if (!mags.empty()) {
            float dv = variance(dirs);
            float mc = consistency(mags);
            if (dv < 0.5f && mc > 0.7f && magnitude > 5.0f) return 1;
            if (magnitude < 3.0f && mc < 0.3f && angularVel > 10.0f) return 2;
            if (magnitude > 3.0f && magnitude < 15.0f && dv > 0.5f) return 3;
        }
    }
    return 0;
}

// Stabilizer.cpp:1722-1747 (only the NORMAL branch value is consumed, :883-886)
float adaptive_strength(int intent, const float motion[3]) {
    float base = std::sqrt(motion[0] * motion[0] + motion[1] * motion[1]);
    float strength = 0.7f;
    switch (intent) {
        case 1: strength = 0.1f + (base / 50.0f) * 0.2f; break;
        case 2: strength = 0.9f - (base / 10.0f) * 0.2f; break;
        case 3: strength = 0.6f + (base / 20.0f) * 0.2f; break;
        default: strength = 0.7f; break;
    }
    return std::max(0.1f, std::min(1.0f, strength));
}

}  // namespace vso

extern "C" {
void vso_box_filter(const float* path, int n, int radius_param, int drone, float* out) {
    std::vector<float> p(path, path + n);
    auto r = vso::box_filter(p, radius_param, drone != 0);
    std::copy(r.begin(), r.end(), out);
}
void vso_gaussian_filter(const float* path, int n, float sigma, float* out) {
    std::vector<float> p(path, path + n);
    auto r = vso::gaussian_filter(p, sigma);
    std::copy(r.begin(), r.end(), out);
}
void vso_kalman_filter(const float* path, int n, float* out) {
    std::vector<float> p(path, path + n);
    auto r = vso::kalman_filter(p);
    std::copy(r.begin(), r.end(), out);
}
int vso_adaptive_radius(const float* px, const float* py, const float* pa, int n, int smoothing_radius) {
    std::vector<float> x(px, px + n), y(py, py + n), a(pa, pa + n);
    return vso::adaptive_radius(x, y, a, smoothing_radius);
}
int vso_motion_intent(const float* transforms, int n, int frame_index) {
    std::vector<float> t(transforms, transforms + (size_t)n * 3);
    float m[3] = {0, 0, 0};
    if (frame_index >= 0 && frame_index < n) { m[0] = t[3 * frame_index]; m[1] = t[3 * frame_index + 1]; m[2] = t[3 * frame_index + 2]; }
    return vso::motion_intent(t, m, frame_index);
}
}
