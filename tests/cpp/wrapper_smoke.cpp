// Drives vs::Stabilizer exactly like the reference's examples/file-capture.cpp:22-64
// (construct from Parameters, stabilize() per frame, empty Mat during warm-up),
// plus the reassignment idiom of examples/vs.cpp:403.  Frames are a synthetic
// moving pattern; prints one "F <rows> <cols> <hash>" line per delivered frame (stabilize() and flush() results in
// order), then "<inputs> <outputs> <flushed> <checksum>".  argv[2] selects the Parameters: reflect (default) | fade |
// canvas | canvas12 (scale 1.2, fills from the temporal buffer) | keep (reflect; every result is kept alive and hashed again at
// the end: a recycled output buffer would show) | unpinned (reflect with Parameters::pinHostFrames = false) | piped (reflect with
// Parameters::hostPipeline).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "video/Stabilizer.h"

static unsigned long long hash_of(const cv::Mat &m) {
    unsigned long long h = 1469598103934665603ull;               // FNV-1a over every byte
    for (int y = 0; y < m.rows; y++) {
        const unsigned char *p = m.ptr(y);
        for (int x = 0; x < m.cols * 3; x++) { h ^= p[x]; h *= 1099511628211ull; }
    }
    return h;
}
static std::vector<cv::Mat> g_kept;
static std::vector<unsigned long long> g_hashes;
static bool g_keep = false;
static void report(const cv::Mat &m) {
    const unsigned long long h = hash_of(m);
    std::printf("F %d %d %llu\n", m.rows, m.cols, h);
    if (g_keep) { g_kept.push_back(m); g_hashes.push_back(h); }
}

static cv::Mat make_frame(int w, int h, int k) {
    cv::Mat f(h, w, CV_8UC3);
    for (int y = 0; y < h; y++) {
        unsigned char *p = f.ptr(y);
        for (int x = 0; x < w; x++) {
            int u = x + 2 * k, v = y + (k % 3);
            unsigned char c = (unsigned char)((((u / 24) + (v / 24)) & 1) ? 200 : 50);
            p[3 * x] = c; p[3 * x + 1] = (unsigned char)(c / 2 + ((u * 7 + v * 3) & 31)); p[3 * x + 2] = (unsigned char)(255 - c);
        }
    }
    return f;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 40;
    vs::Stabilizer::Parameters stabParams;          // file-capture.cpp:22-29
    stabParams.smoothingRadius = 20;
    stabParams.borderType = "reflect";
    stabParams.useCuda = true;
    stabParams.logging = false;
    const char *mode = argc > 2 ? argv[2] : "reflect";
    if (!std::strcmp(mode, "fade")) {
        stabParams.borderType = "fade"; stabParams.borderSize = 12; stabParams.fadeAlpha = 0.3f; stabParams.fadeDuration = 4;
    } else if (!std::strcmp(mode, "canvas")) {
        stabParams.enableVirtualCanvas = true;
    } else if (!std::strcmp(mode, "canvas12")) {
        stabParams.enableVirtualCanvas = true; stabParams.adaptiveCanvasSize = false; stabParams.canvasScaleFactor = 1.2f;
        stabParams.temporalBufferSize = 5;
    } else if (!std::strcmp(mode, "keep")) {
        g_keep = true;
    } else if (!std::strcmp(mode, "unpinned")) {
        stabParams.pinHostFrames = false;
    } else if (!std::strcmp(mode, "piped")) {
        stabParams.hostPipeline = true;
    }
    vs::Stabilizer stab(stabParams);
    stab = vs::Stabilizer(stabParams);               // vs.cpp:403
    int outputs = 0, flushed = 0;
    unsigned long long sum = 0;
    for (int k = 0; k < n; k++) {
        cv::Mat frame = make_frame(320, 240, k);
        cv::Mat stabilized = stab.stabilize(frame); // file-capture.cpp:64
        if (!stabilized.empty()) {
            outputs++;
            report(stabilized);
            for (int y = 0; y < stabilized.rows; y += 16) sum += stabilized.ptr(y)[3 * (y % stabilized.cols)];
        }
    }
    for (;;) {
        cv::Mat f = stab.flush();
        if (f.empty()) break;
        flushed++;
        report(f);
    }
    for (size_t i = 0; i < g_kept.size(); i++)
        if (hash_of(g_kept[i]) != g_hashes[i]) { std::printf("kept frame %zu changed after it was handed out\n", i); return 2; }
    std::printf("%d %d %d %llu\n", n, outputs, flushed, sum);
    return (outputs + flushed == n) ? 0 : 1;
}
