// SYNTAX-CHECK MOCK - NOT OpenCV, not a stand-in to build or run anything against.
//
// Declarations only (no definitions, never linked) of the dozen OpenCV 4.x symbols that the FRT_HAVE_OPENCV branch of include/frt/*.h
// touches, written from OpenCV's public API documentation (core/mat.hpp, core/types.hpp), so that tests/test_cpp_shells.py can run
// `g++ -fsyntax-only` over that branch: the build image has no OpenCV, and before this file a typo in the branch shipped unseen
// (round-4 review item 8).  Types whose SHAPE matters for the shells are modelled faithfully: cv::Mat::step is a MatStep object
// convertible to size_t (not a size_t), Mat's external-data constructor takes (rows, cols, type, void*, size_t step = AUTO_STEP),
// Scalar is a 4-double value type, Point has int x / y.  Nothing here can pin numerics: it proves the branch parses and type-checks.
#ifndef FRT_TEST_OPENCV_DECL_MOCK_CORE_HPP
#define FRT_TEST_OPENCV_DECL_MOCK_CORE_HPP
#include <cstddef>
#include <string>

#define CV_8U 0
#define CV_32F 5
#define CV_CN_SHIFT 3
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << CV_CN_SHIFT))
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#ifndef MIN
#define MIN(a, b) ((a) > (b) ? (b) : (a))
#endif
#ifndef MAX
#define MAX(a, b) ((a) < (b) ? (b) : (a))
#endif

namespace cv {
typedef unsigned char uchar;
typedef std::string String;

struct MatStep {
    MatStep();
    MatStep(size_t s);
    operator size_t() const;
    MatStep &operator=(size_t s);
    size_t *p;
    size_t buf[2];
};

template <typename T>
class Point_ {
  public:
    Point_();
    Point_(T x_, T y_);
    T x, y;
};
typedef Point_<int> Point;

template <typename T>
class Scalar_ {
  public:
    Scalar_();
    Scalar_(T v0, T v1, T v2 = 0, T v3 = 0);
    T val[4];
};
typedef Scalar_<double> Scalar;

class Mat {
  public:
    enum { AUTO_STEP = 0 };
    Mat();
    Mat(int rows, int cols, int type);
    Mat(int rows, int cols, int type, void *data, size_t step = AUTO_STEP);
    Mat(const Mat &m);
    ~Mat();
    Mat &operator=(const Mat &m);
    void create(int rows, int cols, int type);
    Mat clone() const;
    void release();
    int type() const;
    int channels() const;
    size_t elemSize() const;
    bool empty() const;
    bool isContinuous() const;
    template <typename T> T *ptr(int i0 = 0);
    template <typename T> const T *ptr(int i0 = 0) const;
    int flags, dims, rows, cols;
    uchar *data;
    MatStep step;
};
}  // namespace cv
#endif
