// Drives vs::Enhancer like the reference's examples/vs.cpp:547-550 with the settings of its
// examples/config.yaml:23-47.  Prints "<w> <h> <sum of all output bytes>".
#include <cstdio>
#include <cstdlib>
#include "video/Enhancer.h"

int main(int argc, char **argv) {
    const int w = argc > 1 ? std::atoi(argv[1]) : 320, h = argc > 2 ? std::atoi(argv[2]) : 200;
    cv::Mat frame(h, w, CV_8UC3);
    for (int y = 0; y < h; y++) {
        unsigned char *p = frame.ptr(y);
        for (int x = 0; x < w; x++) {
            p[3 * x] = (unsigned char)(x * 3 + y * 5);
            p[3 * x + 1] = (unsigned char)(x ^ y);
            p[3 * x + 2] = (unsigned char)(x * y);
        }
    }
    vs::Enhancer::Parameters params;
    params.brightness = 1.5f; params.contrast = 1.1f;
    params.enableUnsharp = true; params.sharpness = 2.0f; params.blurSigma = 1.0f;
    params.gamma = 1.2f;
    params.useCuda = true;
    if (!vs::Enhancer::enhanceImage(cv::Mat(), params).empty()) return 2;
    cv::Mat out = vs::Enhancer::enhanceImage(frame, params);
    if (out.empty() || out.cols != w || out.rows != h) return 3;
    unsigned long long sum = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < 3 * w; x++) sum += out.ptr(y)[x];
    std::printf("%d %d %llu\n", w, h, sum);
    return 0;
}
