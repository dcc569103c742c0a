// TEST-ONLY stand-in for <opencv2/opencv.hpp>: just enough of cv::Mat / cv::Rect /
// cv::Size for tests/ to compile and run the vs::Stabilizer wrapper in an image
// that has no OpenCV.  It is NOT used to build or emulate the reference, and it
// is not part of the product (applications build against the real OpenCV).
#ifndef MOCK_OPENCV_HPP
#define MOCK_OPENCV_HPP
#include <cstddef>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#define CV_8U 0
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << 3))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_PI 3.1415926535897932384626433832795

namespace cv {
struct Size { int width = 0, height = 0; Size() = default; Size(int w, int h) : width(w), height(h) {} };
struct Rect { int x = 0, y = 0, width = 0, height = 0; Rect() = default; Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {} };
// (the data block and its reference count, with OpenCV's names: cv::Mat::u, cv::UMatData::refcount)
struct UMatData { int refcount = 0; unsigned char *origdata = nullptr; };
class Mat {
public:
    int rows = 0, cols = 0;
    unsigned char *data = nullptr;
    size_t step = 0;
    UMatData *u = nullptr;
    Mat() = default;
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(const Mat &m) : rows(m.rows), cols(m.cols), data(m.data), step(m.step), u(m.u), type_(m.type_) { if (u) __atomic_add_fetch(&u->refcount, 1, __ATOMIC_ACQ_REL); }
    Mat(Mat &&m) noexcept : rows(m.rows), cols(m.cols), data(m.data), step(m.step), u(m.u), type_(m.type_) { m.u = nullptr; m.data = nullptr; m.rows = m.cols = 0; }
    Mat &operator=(const Mat &m) {
        if (this != &m) {
            if (m.u) __atomic_add_fetch(&m.u->refcount, 1, __ATOMIC_ACQ_REL);
            release();
            rows = m.rows; cols = m.cols; data = m.data; step = m.step; u = m.u; type_ = m.type_;
        }
        return *this;
    }
    Mat &operator=(Mat &&m) noexcept {
        if (this != &m) {
            release();
            rows = m.rows; cols = m.cols; data = m.data; step = m.step; u = m.u; type_ = m.type_;
            m.u = nullptr; m.data = nullptr; m.rows = m.cols = 0;
        }
        return *this;
    }
    ~Mat() { release(); }
    void release() {
        if (u && __atomic_sub_fetch(&u->refcount, 1, __ATOMIC_ACQ_REL) == 0) { delete[] u->origdata; delete u; }
        u = nullptr; data = nullptr; rows = cols = 0;
    }
    void create(int r, int c, int type) {
        release();
        type_ = type; rows = r; cols = c;
        step = (size_t)c * channels();
        u = new UMatData;
        u->refcount = 1;
        u->origdata = new unsigned char[step * (size_t)r + 64];        // (not zeroed: like cv::Mat, the pages are untouched until written)
        data = u->origdata;
    }
    int type() const { return type_; }
    int channels() const { return (type_ >> 3) + 1; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    Size size() const { return Size(cols, rows); }
    Mat clone() const {
        Mat m;
        if (empty()) return m;
        m.create(rows, cols, type_);
        for (int y = 0; y < rows; y++) std::memcpy(m.data + (size_t)y * m.step, data + (size_t)y * step, (size_t)cols * channels());
        return m;
    }
    Mat operator()(const Rect &r) const {
        Mat m = *this;
        m.data = data + (size_t)r.y * step + (size_t)r.x * channels();
        m.rows = r.height; m.cols = r.width;
        return m;
    }
    unsigned char *ptr(int y) { return data + (size_t)y * step; }
    const unsigned char *ptr(int y) const { return data + (size_t)y * step; }
private:
    int type_ = CV_8UC3;
};

// ---- capture side (tests/test_host_shims.py): a synthetic cv::VideoCapture and the little of imgproc vs::CamCap calls.
// Source grammar: "mock:<w>x<h>:<frames>[:fail=<a>-<b>]" - frame k is filled with (k % 251); the reads numbered a..b
// (counted per source string, across re-opens) fail.  A camera index opens "mock:64x48:1000000".  Anything else does not open.
enum { CAP_ANY = 0, CAP_GSTREAMER = 1800, CAP_FFMPEG = 1900 };
enum { CAP_PROP_FRAME_WIDTH = 3, CAP_PROP_FRAME_HEIGHT = 4, CAP_PROP_FPS = 5 };
enum { COLOR_BGR2GRAY = 6, COLOR_BGR2HSV = 40, COLOR_BGR2YUV = 82 };
class VideoCapture {
public:
    bool open(int, int = CAP_ANY) { return open(std::string("mock:64x48:1000000")); }
    bool open(const std::string &src, int = CAP_ANY);
    bool isOpened() const { return opened_; }
    void release() { opened_ = false; }
    bool read(Mat &m);
    double get(int prop) const { return !opened_ ? 0.0 : prop == CAP_PROP_FRAME_WIDTH ? w_ : prop == CAP_PROP_FRAME_HEIGHT ? h_ : prop == CAP_PROP_FPS ? 25.0 : 0.0; }
private:
    bool opened_ = false;
    int w_ = 0, h_ = 0, frames_ = 0, fail_a_ = -1, fail_b_ = -1;
    std::string src_;
};
void cvtColor(const Mat &src, Mat &dst, int code);
}  // namespace cv
#endif
