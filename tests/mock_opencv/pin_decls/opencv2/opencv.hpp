// TEST-ONLY, DECLARATIONS ONLY.  The signatures of the OpenCV 4.x calls that the pin harness makes (oracle/opencv_pin/cv_pin.cpp),
// so that `g++ -fsyntax-only` can parse the harness in an image that has no OpenCV (tests/test_opencv_pin_parses.py).  Nothing
// here has a body that computes anything: it cannot build, link or emulate OpenCV or the reference, and it pins nothing.  Its one
// job is to keep the only route to a pinned oracle - running the harness where OpenCV 4.11 exists - from rotting unnoticed.
#ifndef PIN_DECLS_OPENCV_HPP
#define PIN_DECLS_OPENCV_HPP
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

typedef unsigned char uchar;

#define CV_8U 0
#define CV_16S 3
#define CV_64F 6
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << 3))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_PI 3.1415926535897932384626433832795

namespace cv {

template <typename T> struct Size_ {
    T width, height;
    Size_();
    Size_(T w, T h);
    bool operator==(const Size_&) const;
};
typedef Size_<int> Size;
template <typename T> struct Point_ { T x, y; };
typedef Point_<int> Point;
typedef Point_<float> Point2f;
template <typename T> struct Rect_ {
    T x, y, width, height;
    Rect_();
    Rect_(T x, T y, T w, T h);
};
typedef Rect_<int> Rect;
struct Scalar {
    double val[4];
    Scalar();
    Scalar(double v0, double v1 = 0, double v2 = 0, double v3 = 0);
};
template <typename T, int N> struct Vec {
    T val[N];
    const T& operator[](int i) const;
    T& operator[](int i);
};
typedef Vec<float, 2> Vec2f;
template <typename T, int M, int N> struct Matx {
    T val[M * N];
    Matx(T v0, T v1, T v2, T v3, T v4, T v5);
};
typedef Matx<double, 2, 3> Matx23d;
typedef Matx<float, 2, 3> Matx23f;
struct TermCriteria {
    enum { COUNT = 1, MAX_ITER = COUNT, EPS = 2 };
    TermCriteria(int type, int maxCount, double epsilon);
};

class MatExpr;
class Mat {
public:
    int rows, cols;
    uchar* data;
    struct Step { operator size_t() const; } step;
    Mat();
    Mat(int rows, int cols, int type);
    Mat(int rows, int cols, int type, const Scalar& s);
    Mat(const MatExpr& e);
    void create(int rows, int cols, int type);
    void create(Size size, int type);
    bool empty() const;
    int type() const;
    Size size() const;
    size_t total() const;
    Mat clone() const;
    Mat reshape(int cn, int rows = 0) const;
    Mat operator()(const Rect& roi) const;
    template <typename T> T& at(int y, int x);
    template <typename T> const T& at(int y, int x) const;
};
class MatExpr {};
MatExpr operator>(const Mat& a, double s);

// the proxy types OpenCV's functions take: anything the harness passes converts to them
class _InputArray {
public:
    _InputArray();
    _InputArray(const Mat& m);
    _InputArray(const MatExpr& e);
    template <typename T> _InputArray(const std::vector<T>& v);
    template <typename T, int M, int N> _InputArray(const Matx<T, M, N>& m);
};
class _OutputArray : public _InputArray {
public:
    _OutputArray();
    _OutputArray(Mat& m);
    template <typename T> _OutputArray(std::vector<T>& v);
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
typedef const _OutputArray& InputOutputArray;
typedef const _OutputArray& OutputArrayOfArrays;
InputOutputArray noArray();

enum { NORM_INF = 1 };
enum { INTER_LINEAR = 1, WARP_INVERSE_MAP = 16 };
enum { BORDER_CONSTANT = 0, BORDER_REPLICATE = 1, BORDER_REFLECT = 2, BORDER_WRAP = 3, BORDER_REFLECT_101 = 4, BORDER_DEFAULT = 4 };
enum { COLOR_BGR2GRAY = 6 };
enum { RETR_EXTERNAL = 0, CHAIN_APPROX_SIMPLE = 2 };
enum { RANSAC = 8 };

double norm(InputArray a, InputArray b, int normType = 4, InputArray mask = noArray());
void randu(InputOutputArray dst, InputArray low, InputArray high);
void randu(InputOutputArray dst, double low, double high);
void GaussianBlur(InputArray src, OutputArray dst, Size ksize, double sigmaX, double sigmaY = 0, int borderType = BORDER_DEFAULT);
void rectangle(InputOutputArray img, Rect rec, const Scalar& color, int thickness = 1, int lineType = 8, int shift = 0);
void warpAffine(InputArray src, OutputArray dst, InputArray M, Size dsize, int flags = INTER_LINEAR, int borderMode = BORDER_CONSTANT,
                const Scalar& borderValue = Scalar());
void cvtColor(InputArray src, OutputArray dst, int code, int dstCn = 0);
void resize(InputArray src, OutputArray dst, Size dsize, double fx = 0, double fy = 0, int interpolation = INTER_LINEAR);
void pyrDown(InputArray src, OutputArray dst, const Size& dstsize = Size(), int borderType = BORDER_DEFAULT);
void Scharr(InputArray src, OutputArray dst, int ddepth, int dx, int dy, double scale = 1, double delta = 0, int borderType = BORDER_DEFAULT);
void goodFeaturesToTrack(InputArray image, OutputArray corners, int maxCorners, double qualityLevel, double minDistance,
                         InputArray mask = noArray(), int blockSize = 3, bool useHarrisDetector = false, double k = 0.04);
void calcOpticalFlowPyrLK(InputArray prevImg, InputArray nextImg, InputArray prevPts, InputOutputArray nextPts, OutputArray status,
                          OutputArray err, Size winSize = Size(21, 21), int maxLevel = 3,
                          TermCriteria criteria = TermCriteria(TermCriteria::COUNT + TermCriteria::EPS, 30, 0.01), int flags = 0,
                          double minEigThreshold = 1e-4);
Mat estimateAffinePartial2D(InputArray from, InputArray to, OutputArray inliers = noArray(), int method = RANSAC,
                            double ransacReprojThreshold = 3, size_t maxIters = 2000, double confidence = 0.99, size_t refineIters = 10);
void copyMakeBorder(InputArray src, OutputArray dst, int top, int bottom, int left, int right, int borderType, const Scalar& value = Scalar());
void Canny(InputArray image, OutputArray edges, double threshold1, double threshold2, int apertureSize = 3, bool L2gradient = false);
void HoughLines(InputArray image, OutputArray lines, double rho, double theta, int threshold, double srn = 0, double stn = 0,
                double min_theta = 0, double max_theta = CV_PI);
void findContours(InputArray image, OutputArrayOfArrays contours, int mode, int method, Point offset = Point());
void absdiff(InputArray src1, InputArray src2, OutputArray dst);
int countNonZero(InputArray src);
void setNumThreads(int nthreads);

}  // namespace cv
#endif
