// CPU ORACLE (test infrastructure) - pins the restatement to a real OpenCV, and optionally to the reference itself.
//
// Built only where OpenCV is installed (`make -C oracle opencv-pin [REF=/path/to/video-stab]`): this image has none, so
// the file has never been compiled here - it is the C++ twin of tests/test_opencv_pin.py for machines with the OpenCV
// development package.  Without -DVS_WITH_OPENCV_ORACLE it is an empty program.
//
//   part A  every primitive of vso.h against the cv:: function it restates, on seeded inputs (byte-exact for the 8-bit
//           ones, tolerance printed for the float ones);
//   part B  (-DVS_PIN_REFERENCE, REF given) the reference's own Stabilizer.cpp, compiled from where it lies against the
//           installed OpenCV, fed the same clip as vso_stab_push: transforms, window/fill decisions and output frames.
//           This is the end-to-end pin docs/opencv_semantics.md calls "E2E".
#ifdef VS_WITH_OPENCV_ORACLE
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include <opencv2/opencv.hpp>

#include "vso.h"
#ifdef VS_PIN_REFERENCE
#include "video/Stabilizer.h"
#endif

static int g_fail = 0, g_run = 0;
static void report(const char* what, bool ok, const char* note = "") {
    g_run++;
    if (!ok) g_fail++;
    std::printf("%-46s %s %s\n", what, ok ? "same" : "DIFFERS", note);
}
static bool same(const cv::Mat& a, const cv::Mat& b) {
    return a.size() == b.size() && a.type() == b.type() && cv::norm(a, b, cv::NORM_INF) == 0;
}

// a textured world, frame k of a panning, shaking camera
static cv::Mat frame_of(int k, int w, int h) {
    static cv::Mat world;
    if (world.empty()) {
        world.create(1200, 1800, CV_8UC3);
        std::mt19937 rng(1234);
        cv::randu(world, 30, 220);
        cv::GaussianBlur(world, world, cv::Size(0, 0), 3.0);
        for (int i = 0; i < 400; i++) {
            const cv::Rect r(rng() % 1700, rng() % 1100, 10 + rng() % 80, 10 + rng() % 80);
            cv::rectangle(world, r, cv::Scalar(rng() % 256, rng() % 256, rng() % 256), (rng() % 3) ? -1 : 2);
        }
    }
    std::mt19937 rng(99 + k);
    const double x = 200 + 2.0 * k + (int)(rng() % 7) - 3, y = 150 + (int)(rng() % 7) - 3, a = ((int)(rng() % 9) - 4) * 0.001;
    const cv::Matx23d M(std::cos(a), -std::sin(a), x, std::sin(a), std::cos(a), y);
    cv::Mat out;
    cv::warpAffine(world, out, M, cv::Size(w, h), cv::INTER_LINEAR | cv::WARP_INVERSE_MAP);
    return out;
}

static void part_a() {
    const cv::Mat f = frame_of(0, 640, 360), f1 = frame_of(1, 640, 360);
    cv::Mat g, g1, t, o;
    cv::cvtColor(f, g, cv::COLOR_BGR2GRAY);
    cv::cvtColor(f1, g1, cv::COLOR_BGR2GRAY);
    {   // cvtColor, resize
        o.create(f.size(), CV_8UC1);
        vso_bgr2gray(f.data, f.cols, f.rows, f.step, o.data, o.step);
        report("cvtColor(BGR2GRAY)", same(o, g));
        for (cv::Size s : {cv::Size(960, 540), cv::Size(333, 211), cv::Size(320, 180)}) {
            cv::resize(f, t, s, 0, 0, cv::INTER_LINEAR);
            o.create(s, CV_8UC3);
            vso_resize_linear_u8(f.data, f.cols, f.rows, f.step, 3, o.data, s.width, s.height, o.step);
            report("resize(INTER_LINEAR) 8UC3", same(o, t));
        }
    }
    {   // pyrDown, Scharr (the LK pyramid)
        cv::pyrDown(g, t);
        o.create(t.size(), CV_8UC1);
        vso_pyr_down(g.data, g.cols, g.rows, g.step, o.data, o.step);
        report("pyrDown", same(o, t));
        cv::Mat dx, dy;
        cv::Scharr(g, dx, CV_16S, 1, 0);
        cv::Scharr(g, dy, CV_16S, 0, 1);
        std::vector<int16_t> d((size_t)g.total() * 2);
        vso_scharr(g.data, g.cols, g.rows, g.step, d.data());
        bool ok = true;
        for (int y = 0; y < g.rows && ok; y++)
            for (int x = 0; x < g.cols; x++)
                if (d[((size_t)y * g.cols + x) * 2] != dx.at<short>(y, x) || d[((size_t)y * g.cols + x) * 2 + 1] != dy.at<short>(y, x)) { ok = false; break; }
        report("Scharr dx, dy (16S)", ok);
    }
    std::vector<cv::Point2f> corners;
    {   // goodFeaturesToTrack
        cv::goodFeaturesToTrack(g, corners, 200, 0.01, 30.0, cv::noArray(), 3);
        std::vector<float> p(400);
        int nc = 0;
        const int n = vso_gftt(g.data, g.cols, g.rows, g.step, 200, 0.01, 30.0, 3, p.data(), &nc);
        bool ok = n == (int)corners.size();
        for (int i = 0; ok && i < n; i++) ok = p[2 * i] == corners[i].x && p[2 * i + 1] == corners[i].y;
        report("goodFeaturesToTrack", ok);
    }
    {   // calcOpticalFlowPyrLK
        std::vector<cv::Point2f> nxt;
        std::vector<uchar> st;
        std::vector<float> err;
        cv::calcOpticalFlowPyrLK(g, g1, corners, nxt, st, err, cv::Size(15, 15), 2,
                                 cv::TermCriteria(cv::TermCriteria::COUNT + cv::TermCriteria::EPS, 20, 0.03));
        const int n = (int)corners.size();
        std::vector<float> out(2 * n), e(n);
        std::vector<uint8_t> s(n);
        vso_pyr_lk(g.data, g1.data, g.cols, g.rows, g.step, &corners[0].x, n, out.data(), s.data(), e.data(), 15, 2, 20, 0.03);
        bool ok = true;
        float worst = 0;
        for (int i = 0; i < n; i++) {
            ok = ok && s[i] == st[i];
            if (st[i]) worst = std::max(worst, std::max(std::abs(out[2 * i] - nxt[i].x), std::abs(out[2 * i + 1] - nxt[i].y)));
        }
        char note[64];
        std::snprintf(note, sizeof note, "(max |d| %.2g px)", worst);
        report("calcOpticalFlowPyrLK status, points <= 1e-2", ok && worst <= 1e-2f, note);
        // estimateAffinePartial2D on the tracked pairs
        std::vector<cv::Point2f> a, b;
        for (int i = 0; i < n; i++) if (st[i]) { a.push_back(corners[i]); b.push_back(nxt[i]); }
        std::vector<uchar> inl;
        const cv::Mat M = cv::estimateAffinePartial2D(a, b, inl, cv::RANSAC, 5.0, 500);
        double model[6];
        std::vector<uint8_t> mine(a.size());
        int32_t info[4];
        const int got = vso_estimate_affine_partial2d(&a[0].x, &b[0].x, (int)a.size(), 5.0, 500, model, mine.data(), info);
        ok = (got != 0) == !M.empty();
        if (ok && !M.empty()) {
            ok = std::memcmp(mine.data(), inl.data(), inl.size()) == 0;
            for (int i = 0; ok && i < 6; i++) ok = model[i] == M.at<double>(i / 3, i % 3);
        }
        report("estimateAffinePartial2D (RANSAC 5.0, 500)", ok);
    }
    {   // warpAffine, copyMakeBorder
        const cv::Matx23f M(std::cos(0.004f), -std::sin(0.004f), 3.25f, std::sin(0.004f), std::cos(0.004f), -5.5f);
        cv::warpAffine(f, t, M, f.size(), cv::INTER_LINEAR, cv::BORDER_CONSTANT);
        o.create(f.size(), CV_8UC3);
        vso_warp_affine(f.data, f.cols, f.rows, f.step, 3, o.data, o.step, M.val);
        report("warpAffine(INTER_LINEAR, CONSTANT) 8UC3", same(o, t));
        const int modes[5] = {cv::BORDER_CONSTANT, cv::BORDER_REFLECT, cv::BORDER_REFLECT_101, cv::BORDER_REPLICATE, cv::BORDER_WRAP};
        for (int k = 0; k < 5; k++) {
            cv::copyMakeBorder(f, t, 24, 24, 24, 24, modes[k], cv::Scalar(0, 0, 0));
            o.create(t.size(), CV_8UC3);
            vso_copy_make_border(f.data, f.cols, f.rows, f.step, 3, o.data, o.step, 24, k);
            report("copyMakeBorder", same(o, t));
            if (modes[k] == cv::BORDER_REPLICATE || modes[k] == cv::BORDER_REFLECT) {
                const cv::Matx23d Md(M.val[0], M.val[1], M.val[2], M.val[3], M.val[4], M.val[5]);
                cv::warpAffine(f, t, Md, f.size(), cv::INTER_LINEAR, modes[k]);
                o.create(f.size(), CV_8UC3);
                vso_warp_affine_d(f.data, f.cols, f.rows, f.step, 3, o.data, o.step, Md.val, k);
                report("warpAffine, REPLICATE / REFLECT border", same(o, t));
            }
        }
    }
    {   // Canny, HoughLines, findContours
        cv::Mat e;
        cv::Canny(g, e, 50, 150);
        o.create(g.size(), CV_8UC1);
        vso_canny(g.data, g.cols, g.rows, g.step, 50, 150, o.data);
        report("Canny(50, 150)", same(o, e));
        std::vector<cv::Vec2f> lines;
        cv::HoughLines(e, lines, 1.0, CV_PI / 180.0, 80);
        std::vector<float> mine(2 * 65536);
        const int n = vso_hough_lines(e.data, e.cols, e.rows, e.step, 1.0f, (float)(CV_PI / 180.0), 80, mine.data(), 65536);
        bool ok = n == (int)lines.size();
        for (int i = 0; ok && i < n; i++) ok = mine[2 * i] == lines[i][0] && mine[2 * i + 1] == lines[i][1];
        report("HoughLines(1, pi/180, 80): lines and order", ok);
        cv::Mat m = g > 128;
        std::vector<std::vector<cv::Point>> cs;
        cv::findContours(m.clone(), cs, cv::RETR_EXTERNAL, cv::CHAIN_APPROX_SIMPLE);
        std::vector<int32_t> counts(m.total() / 2 + 4), xy(4 * m.total() + 32);
        const int nc = vso_find_contours(m.data, m.cols, m.rows, m.step, counts.data(), (int)counts.size(), xy.data(), (int)xy.size() / 2);
        ok = nc == (int)cs.size();
        size_t k = 0;
        for (int i = 0; ok && i < nc; i++) {
            ok = counts[i] == (int)cs[i].size();
            for (int j = 0; ok && j < counts[i]; j++, k++) ok = xy[2 * k] == cs[i][j].x && xy[2 * k + 1] == cs[i][j].y;
        }
        report("findContours(EXTERNAL, SIMPLE): order and points", ok);
    }
}

#ifdef VS_PIN_REFERENCE
static void run_config(const char* name, const vs::Stabilizer::Parameters& rp, const vs_params_c& op, int w, int h, int n) {
    vs::Stabilizer ref(rp);
    vso_stab* mine = vso_stab_create(&op);
    int ow = 0, oh = 0;
    bool ok = true;
    long long differing = 0, total = 0;
    for (int k = 0; k < n + 64 && ok; k++) {
        const bool flushing = k >= n;
        const cv::Mat f = frame_of(flushing ? 0 : k, w, h);
        const cv::Mat r = flushing ? ref.flush() : ref.stabilize(f);
        vso_stab_out_size(mine, w, h, &ow, &oh);
        cv::Mat o(std::max(oh, h), std::max(ow, w), CV_8UC3, cv::Scalar(0, 0, 0));
        const int got = flushing ? vso_stab_flush(mine, o.data, o.step) : vso_stab_push(mine, f.data, w, h, f.step, VS_FMT_BGR8, o.data, o.step);
        if ((got != 0) != !r.empty()) { ok = false; break; }
        if (r.empty()) { if (flushing) break; continue; }
        const cv::Mat oc = o(cv::Rect(0, 0, r.cols, r.rows));
        cv::Mat d;
        cv::absdiff(oc, r, d);
        differing += cv::countNonZero(d.reshape(1));
        total += (long long)d.total() * 3;
    }
    vso_stab_destroy(mine);
    char note[96];
    std::snprintf(note, sizeof note, "(%lld of %lld samples differ)", differing, total);
    report(name, ok && differing == 0, note);
}

static void part_b() {
    vs::Stabilizer::Parameters rp;
    vs_params_c op;
    vso_params_default(&op);
    rp.smoothingRadius = op.smoothing_radius = 10;
    rp.useCuda = false;
    run_config("E2E box, radius 10, 640x360", rp, op, 640, 360, 40);
    rp.borderType = "reflect"; rp.borderSize = op.border_size = 16; op.border_type = VS_BORDER_REFLECT;
    run_config("E2E reflect border 16", rp, op, 640, 360, 30);
    rp.borderType = "fade"; op.border_type = VS_BORDER_FADE;
    run_config("E2E fade border 16", rp, op, 640, 360, 30);
    rp.borderType = "black"; op.border_type = VS_BORDER_BLACK; rp.cropNZoom = true; op.crop_n_zoom = 1;
    run_config("E2E crop-and-zoom 16", rp, op, 640, 360, 30);
    rp.cropNZoom = false; op.crop_n_zoom = 0; rp.borderSize = op.border_size = 0;
    rp.smoothingMethod = "gaussian"; op.smoothing_method = VS_SMOOTH_GAUSSIAN;
    run_config("E2E gaussian", rp, op, 640, 360, 30);
    rp.smoothingMethod = "kalman"; op.smoothing_method = VS_SMOOTH_KALMAN;
    run_config("E2E kalman", rp, op, 640, 360, 30);
    rp.smoothingMethod = "box"; op.smoothing_method = VS_SMOOTH_BOX;
    rp.enableVirtualCanvas = true; op.enable_virtual_canvas = 1;
    run_config("E2E virtual canvas, defaults", rp, op, 640, 360, 30);
    rp.adaptiveCanvasSize = false; op.adaptive_canvas_size = 0; rp.canvasScaleFactor = op.canvas_scale_factor = 1.2f;
    run_config("E2E virtual canvas, scale 1.2 (fills)", rp, op, 640, 360, 30);
    rp.enableVirtualCanvas = false; op.enable_virtual_canvas = 0;
    rp.droneHighFreqMode = true; op.drone_high_freq_mode = 1;
    run_config("E2E drone high-frequency mode", rp, op, 640, 360, 40);
}
#endif

int main() {
    cv::setNumThreads(1);
    part_a();
#ifdef VS_PIN_REFERENCE
    part_b();
#else
    std::printf("(part B, the reference's Stabilizer.cpp end to end, needs REF=/path/to/video-stab)\n");
#endif
    std::printf("%d of %d checks differ\n", g_fail, g_run);
    return g_fail ? 1 : 0;
}
#else
int main() { return 0; }
#endif
