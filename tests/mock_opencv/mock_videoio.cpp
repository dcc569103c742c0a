// TEST-ONLY: the synthetic cv::VideoCapture / cv::cvtColor declared in tests/mock_opencv/opencv2/opencv.hpp.
#include <opencv2/opencv.hpp>

#include <cstdio>
#include <map>
#include <mutex>

namespace cv {

namespace {
std::mutex g_m;
std::map<std::string, int> g_reads;     // reads so far, per source string (survives release() / open())
}  // namespace

bool VideoCapture::open(const std::string &src, int) {
    opened_ = false;
    int w = 0, h = 0, n = 0, a = -1, b = -1;
    const int got = std::sscanf(src.c_str(), "mock:%dx%d:%d:fail=%d-%d", &w, &h, &n, &a, &b);
    if (got < 3 || w <= 0 || h <= 0) return false;
    w_ = w; h_ = h; frames_ = n; fail_a_ = got == 5 ? a : -1; fail_b_ = got == 5 ? b : -1;
    src_ = src;
    opened_ = true;
    return true;
}

bool VideoCapture::read(Mat &m) {
    if (!opened_) return false;
    int k;
    {
        std::lock_guard<std::mutex> lk(g_m);
        k = g_reads[src_]++;
    }
    if (k >= frames_ || (k >= fail_a_ && k <= fail_b_)) { m = Mat(); return false; }
    m.create(h_, w_, CV_8UC3);
    std::memset(m.data, k % 251, m.step * (size_t)m.rows);
    return true;
}

void cvtColor(const Mat &src, Mat &dst, int code) {
    if (code != COLOR_BGR2GRAY) { dst = src.clone(); return; }
    Mat g(src.rows, src.cols, CV_8UC1);
    for (int y = 0; y < src.rows; y++)
        for (int x = 0; x < src.cols; x++) {
            const unsigned char *p = src.ptr(y) + 3 * x;
            g.ptr(y)[x] = (unsigned char)((p[0] + p[1] + p[2]) / 3);
        }
    dst = g;
}

}  // namespace cv
