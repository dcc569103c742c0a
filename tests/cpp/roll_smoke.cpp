// Drives vs::RollCorrection + vs::AutoZoomCrop like the reference's
// examples/roll-correction-file.cpp:52-70 (autoCorrectRoll on every frame, autoZoomCrop on
// its result).  Frames: a tilted horizon.  Prints "<frames> <out_w> <out_h> <checksum>".
#include <cstdio>
#include <cstdlib>
#include "video/AutoZoomCrop.h"
#include "video/RollCorrection.h"

static cv::Mat make_frame(int w, int h, int k) {
    cv::Mat f(h, w, CV_8UC3);
    for (int y = 0; y < h; y++) {
        unsigned char *p = f.ptr(y);
        for (int x = 0; x < w; x++) {
            const bool ground = (y - h / 2 - k) * 1024 > (x - w / 2) * 55;
            p[3 * x] = ground ? 40 : 230;
            p[3 * x + 1] = (unsigned char)((ground ? 90 : 200) + ((x / 16 + y / 16) & 1) * 8);
            p[3 * x + 2] = ground ? 60 : 170;
        }
    }
    return f;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 6;
    unsigned long long sum = 0;
    int ow = 0, oh = 0;
    vs::RollCorrection::Parameters params;
    for (int k = 0; k < n; k++) {
        cv::Mat frame = make_frame(640, 360, k);
        cv::Mat corrected = (k & 1) ? vs::RollCorrection::autoCorrectRoll(frame, params)
                                    : vs::RollCorrection::autoCorrectRoll(frame);
        if (corrected.empty() || corrected.cols != 640 || corrected.rows != 360) return 2;
        cv::Mat finalFrame = vs::AutoZoomCrop::autoZoomCrop(corrected, 0.05);
        if (finalFrame.empty()) return 3;
        ow = finalFrame.cols; oh = finalFrame.rows;
        for (int y = 0; y < oh; y += 8) sum += finalFrame.ptr(y)[3 * (y % ow)];
    }
    std::printf("%d %d %d %llu\n", n, ow, oh, sum);
    return (ow == 640 && oh == 360) ? 0 : 1;
}
