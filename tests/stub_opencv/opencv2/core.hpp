// NOT OpenCV.  A stand-in with just enough of cv::Mat / Point / Size / TermCriteria for the compiler to check
// include/ccalib_cv.hpp (tests/test_cv_adaptor_compiles.py).  It pins nothing about OpenCV's behaviour and is never
// used to compute anything that is compared with the reference.
#pragma once
#include <algorithm>
#include <cstddef>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#define CV_64F 6
#define CV_32F 5
namespace cv {
namespace Error { enum { StsError = -2, StsBadArg = -5, StsNotImplemented = -213 }; }
struct Exception : std::runtime_error { int code; Exception(int c, const std::string& m) : std::runtime_error(m), code(c) {} };
#define CV_Error(code, msg) throw cv::Exception(code, msg)
#define CV_Assert(expr) do { if (!(expr)) throw cv::Exception(cv::Error::StsError, #expr); } while (0)
template <class T> struct Point_ { T x, y; };
template <class T> struct Point3_ { T x, y, z; };
typedef Point_<float> Point2f;
typedef Point3_<float> Point3f;
struct Size { int width, height; Size(int w = 0, int h = 0) : width(w), height(h) {} };
struct TermCriteria {
    enum { COUNT = 1, MAX_ITER = 1, EPS = 2 };
    int type, maxCount; double epsilon;
    TermCriteria(int t = 0, int n = 0, double e = 0) : type(t), maxCount(n), epsilon(e) {}
};
// doubles only, shared storage, row views; enough for the adaptor
class Mat {
public:
    int rows = 0, cols = 0;
    Mat() {}
    Mat(int r, int c, int) { create(r, c, CV_64F); }
    void create(int r, int c, int) { rows = r, cols = c, ofs_ = 0, buf_ = std::make_shared<std::vector<double>>((size_t)r * c, 0.); }
    static Mat zeros(int r, int c, int t) { return Mat(r, c, t); }
    static Mat eye(int r, int c, int t) { Mat m(r, c, t); for (int i = 0; i < std::min(r, c); i++) m.ptr<double>()[(size_t)i * c + i] = 1; return m; }
    bool empty() const { return !buf_ || rows * cols == 0; }
    size_t total() const { return (size_t)rows * cols; }
    void release() { buf_.reset(); rows = cols = 0; }
    template <class T> T* ptr() { return reinterpret_cast<T*>(buf_->data() + ofs_); }
    template <class T> const T* ptr() const { return reinterpret_cast<const T*>(buf_->data() + ofs_); }
    Mat reshape(int, int r) const { Mat m = *this; m.rows = r; m.cols = r ? (int)total() / r : 0; return m; }
    void convertTo(Mat& dst, int) const { dst = clone(); }
    Mat clone() const { Mat m(rows, cols, CV_64F); if (!empty()) std::copy(ptr<double>(), ptr<double>() + total(), m.ptr<double>()); return m; }
    void copyTo(Mat dst) const { std::copy(ptr<double>(), ptr<double>() + total(), dst.ptr<double>()); }
    Mat rowRange(int a, int b) const { Mat m = *this; m.ofs_ = ofs_ + (size_t)a * cols; m.rows = b - a; return m; }
    Mat row(int i) const { return rowRange(i, i + 1); }
    Mat t() const { Mat m = clone(); std::swap(m.rows, m.cols); if (rows > 1 && cols > 1) for (int i = 0; i < rows; i++) for (int j = 0; j < cols; j++) m.ptr<double>()[(size_t)j * rows + i] = ptr<double>()[(size_t)i * cols + j]; return m; }
private:
    std::shared_ptr<std::vector<double>> buf_;
    size_t ofs_ = 0;
};
}  // namespace cv
