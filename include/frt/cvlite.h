// Minimal cv::Mat-compatible view used ONLY when OpenCV is not installed (this build image has none), so that the drop-in
// shells in this directory compile and can be tested.  With OpenCV present (FRT_HAVE_OPENCV, detected through
// __has_include) the real <opencv2/core.hpp> is used instead and this file is inert.  It implements no image processing:
// every pixel operation of the hot path runs in libfrt's HIP kernels.
#ifndef FRT_CVLITE_H
#define FRT_CVLITE_H

#if defined(__has_include)
#if __has_include(<opencv2/core.hpp>)
#define FRT_HAVE_OPENCV 1
#endif
#endif

#if defined(FRT_EXPECT_OPENCV_BRANCH) && !defined(FRT_HAVE_OPENCV)
#error "FRT_EXPECT_OPENCV_BRANCH: <opencv2/core.hpp> was not found on the include path"
#endif
#ifdef FRT_HAVE_OPENCV
#include <opencv2/core.hpp>
#include <opencv2/imgproc.hpp>
#else
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << 3))
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
class Mat {
  public:
    int rows, cols;
    uchar *data;
    size_t step;
    Mat() : rows(0), cols(0), data(nullptr), step(0), type_(CV_8UC3) {}
    Mat(int r, int c, int type) : rows(0), cols(0), data(nullptr), step(0), type_(type) { create(r, c, type); }
    Mat(int r, int c, int type, void *ext, size_t step_bytes = 0) : rows(r), cols(c), data((uchar *)ext), step(step_bytes), type_(type) {
        if (!step) step = (size_t)c * elemSize();
    }
    void create(int r, int c, int type) {
        type_ = type;
        rows = r;
        cols = c;
        step = (size_t)c * elemSize();
        store_ = std::make_shared<std::vector<uchar>>((size_t)r * step);
        data = store_->data();
    }
    int type() const { return type_; }
    int channels() const { return (type_ >> 3) + 1; }
    size_t elemSize() const { return (size_t)channels() * ((type_ & 7) == CV_32F ? 4 : 1); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    bool isContinuous() const { return step == (size_t)cols * elemSize(); }
    template <typename T> T *ptr(int r = 0) { return reinterpret_cast<T *>(data + (size_t)r * step); }
    template <typename T> const T *ptr(int r = 0) const { return reinterpret_cast<const T *>(data + (size_t)r * step); }
    Mat clone() const {
        Mat m(rows, cols, type_);
        for (int r = 0; r < rows; ++r) std::memcpy(m.data + (size_t)r * m.step, data + (size_t)r * step, (size_t)cols * elemSize());
        return m;
    }
    void release() {
        store_.reset();
        data = nullptr;
        rows = cols = 0;
    }

  private:
    int type_;
    std::shared_ptr<std::vector<uchar>> store_;
};
}  // namespace cv
#endif  // FRT_HAVE_OPENCV
#endif  // FRT_CVLITE_H
