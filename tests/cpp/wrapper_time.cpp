// wrapper_time [device]: frames/s of vs::Stabilizer::stabilize(cv::Mat) at 1920x1080 measured THROUGH THE C++ CLASS, the way an
// application drives it (examples/file-capture.cpp:58-64: one capture Mat that every read fills, one stabilize() per frame):
//   default           Parameters as constructed (synchronous call; results from the page-locked frame ring)
//   pin_input         Parameters::pinInputFrames = true as well (the capture buffers are registered)
//   host_pipeline     Parameters::hostPipeline = true
//   unpinned          Parameters::pinHostFrames = false (plain cv::Mat allocations: round 2's behaviour)
//   keeps_results     default, but the application keeps every result alive for 8 frames (the ring cannot recycle)
// wrapper_time <device> checker: the same two calls (default, host_pipeline) on a 24-pixel checkerboard - tens of thousands of corner
// candidates per analysis image: the detector's worst case (the selection consumes its candidate list in chunks since round 4).
// Prints one JSON line.  The capture side is three Mats holding consecutive synthetic frames, taken in turn (a decoder's
// buffer pool); only the stabilize() calls are timed.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <string>
#include <vector>
#include "video/Stabilizer.h"

// A picture with the statistics of the bench clips (vsamd/synth.py: square-wave gratings, a few hundred rectangles, weak noise):
// a few thousand corner candidates, not the tens of thousands of a fine checkerboard.
static cv::Mat make_frame(int w, int h, int k) {
    cv::Mat f(h, w, CV_8UC3);
    for (int y = 0; y < h; y++) {
        unsigned char *p = f.ptr(y);
        for (int x = 0; x < w; x++) {
            const int u = x + 2 * k, v = y + (k % 3);
            int c = 96 + 22 * (((3 * u + 2 * v) / 310) & 1) - 18 * (((5 * v - u + 40000) / 470) & 1) + 14 * (((u + 7 * v) / 640) & 1);
            c += (int)((((unsigned)u * 73856093u) ^ ((unsigned)v * 19349663u)) >> 7 & 7) - 3;
            p[3 * x] = (unsigned char)c; p[3 * x + 1] = (unsigned char)(c + 10); p[3 * x + 2] = (unsigned char)(c - 10);
        }
    }
    unsigned s = 12345u;
    for (int r = 0; r < 300; r++) {                  // filled rectangles, moving with the camera
        s = s * 1664525u + 1013904223u; const int x0 = (int)((s >> 8) % (unsigned)(w + 200)) - 100 - 2 * k;
        s = s * 1664525u + 1013904223u; const int y0 = (int)((s >> 8) % (unsigned)(h + 100)) - 50 - (k % 3);
        s = s * 1664525u + 1013904223u; const int rw = 8 + (int)((s >> 8) % 64u), rh = 8 + (int)((s >> 16) % 64u);
        s = s * 1664525u + 1013904223u; const unsigned col = s >> 8;
        for (int y = y0 < 0 ? 0 : y0; y < y0 + rh && y < h; y++) {
            unsigned char *p = f.ptr(y);
            for (int x = x0 < 0 ? 0 : x0; x < x0 + rw && x < w; x++) { p[3 * x] = (unsigned char)col; p[3 * x + 1] = (unsigned char)(col >> 8); p[3 * x + 2] = (unsigned char)(col >> 16); }
        }
    }
    return f;
}

// A moving checkerboard of 24-pixel squares with a little texture: every square corner is a corner candidate.
static cv::Mat make_checker(int w, int h, int k) {
    cv::Mat f(h, w, CV_8UC3);
    for (int y = 0; y < h; y++) {
        unsigned char *p = f.ptr(y);
        for (int x = 0; x < w; x++) {
            const int u = x + 2 * k, v = y + (k % 3);
            int c = (((u / 24) + (v / 24)) & 1) ? 200 : 60;
            c += (int)((((unsigned)u * 73856093u) ^ ((unsigned)v * 19349663u)) >> 7 & 7) - 3;
            p[3 * x] = (unsigned char)c; p[3 * x + 1] = (unsigned char)(c + 10); p[3 * x + 2] = (unsigned char)(c - 10);
        }
    }
    return f;
}

static double run(const vs::Stabilizer::Parameters &p, std::vector<cv::Mat> &cap, int warm, int timed, int keep) {
    vs::Stabilizer stab(p);
    std::deque<cv::Mat> kept;
    double secs = 0;
    int outs = 0;
    for (int k = 0; k < warm + timed; k++) {
        const int i = k % 4 == 3 ? 1 : k % 4;            // 0 1 2 1 0 1 2 1 ...: consecutive frames differ by one camera step
        const auto t0 = std::chrono::steady_clock::now();
        cv::Mat out = stab.stabilize(cap[(size_t)i]);
        const auto t1 = std::chrono::steady_clock::now();
        if (k >= warm) { secs += std::chrono::duration<double>(t1 - t0).count(); outs += out.empty() ? 0 : 1; }
        if (keep > 0 && !out.empty()) { kept.push_back(out); if ((int)kept.size() > keep) kept.pop_front(); }
    }
    return outs / secs;
}

int main(int argc, char **argv) {
    if (argc > 1) setenv("VS_STAB_DEVICE", argv[1], 1);
    const int W = 1920, H = 1080;
    std::vector<cv::Mat> cap;
    const bool checker = argc > 2 && std::string(argv[2]) == "checker";
    for (int k = 0; k < 3; k++) cap.push_back(checker ? make_checker(W, H, k) : make_frame(W, H, k));
    vs::Stabilizer::Parameters p;                    // the reference's defaults (smoothingRadius 30 ...)
    p.logging = false;
    const int warm = 80, timed = 400;
    if (checker) {
        const double sync = run(p, cap, warm, timed, 0);
        vs::Stabilizer::Parameters q = p;
        q.hostPipeline = true;
        const double piped = run(q, cap, warm, timed, 0);
        std::printf("{\"what\": \"the same calls on a 24-pixel checkerboard (tens of thousands of corner candidates per analysis image)\", "
                    "\"default_synchronous\": %.1f, \"host_pipeline\": %.1f, \"unit\": \"frames/s\"}\n", sync, piped);
        return 0;
    }
    const double sync = run(p, cap, warm, timed, 0);
    vs::Stabilizer::Parameters q = p;
    q.hostPipeline = true;
    const double piped = run(q, cap, warm, timed, 0);
    vs::Stabilizer::Parameters r = p;
    r.pinHostFrames = false;
    const double unpinned = run(r, cap, warm, timed, 0);
    const double keeps = run(p, cap, warm, timed, 8);
    vs::Stabilizer::Parameters pi = p;
    pi.pinInputFrames = true;
    const double sync_pi = run(pi, cap, warm, timed, 0);
    vs::Stabilizer::Parameters qi = q;
    qi.pinInputFrames = true;
    const double piped_pi = run(qi, cap, warm, timed, 0);
    std::printf("{\"what\": \"vs::Stabilizer::stabilize(cv::Mat) at %dx%d through the C++ class, %d timed calls, default Parameters\", "
                "\"default_synchronous\": %.1f, \"host_pipeline\": %.1f, \"unpinned_synchronous\": %.1f, \"default_caller_keeps_8_results\": %.1f, "
                "\"pin_input_synchronous\": %.1f, \"pin_input_host_pipeline\": %.1f, \"unit\": \"frames/s\"}\n", W, H, timed, sync, piped, unpinned, keeps,
                sync_pi, piped_pi);
    return 0;
}
