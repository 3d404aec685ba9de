// cv_compat.h — the handful of OpenCV value types the three front-end classes expose in their signatures.
//
// When OpenCV is installed (the normal ORB_SLAM3 build) this header simply includes it and the shims compile
// against the real cv::Mat / cv::KeyPoint.  In this repository's containers there is no OpenCV, so a minimal
// stand-in with the same member names and the same binary layout (cv::KeyPoint = 28 bytes) is provided; it is
// just enough for the shim sources and their tests, not an OpenCV replacement.
#pragma once
#if defined(__has_include)
#if __has_include(<opencv2/core/core.hpp>) && !defined(RGBL_FORCE_CV_COMPAT)
#define RGBL_HAVE_OPENCV 1
#endif
#endif

#ifdef RGBL_HAVE_OPENCV
#include <opencv2/core/core.hpp>
#else
#include <stdint.h>
#include <string.h>

#include <memory>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_8UC1 0
#define CV_32FC1 5

namespace cv {
typedef unsigned char uchar;
struct Point2f { float x = 0, y = 0; Point2f() {} Point2f(float x_, float y_) : x(x_), y(y_) {} };
struct Size { int width = 0, height = 0; Size() {} Size(int w, int h) : width(w), height(h) {} };
struct KeyPoint {
  Point2f pt;
  float size = 0, angle = -1, response = 0;
  int octave = 0, class_id = -1;
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint layout");

// Reference-counted dense 2-D array, row-major, like a continuous cv::Mat.
class Mat {
 public:
  int rows = 0, cols = 0;
  size_t step = 0;
  uchar* data = nullptr;
  Mat() {}
  Mat(int r, int c, int type) { create(r, c, type); }
  Mat(int r, int c, int type, void* ext, size_t step_ = 0) : rows(r), cols(c), data((uchar*)ext), type_(type) {
    step = step_ ? step_ : (size_t)c * elemSize();
  }
  void create(int r, int c, int type) {
    if (r == rows && c == cols && type == type_ && data && own_) return;
    rows = r; cols = c; type_ = type;
    step = (size_t)c * elemSize();
    own_ = std::shared_ptr<uchar>(new uchar[step * (size_t)r + 16](), std::default_delete<uchar[]>());
    data = own_.get();
  }
  static Mat zeros(int r, int c, int type) { return Mat(r, c, type); }
  void release() { own_.reset(); data = nullptr; rows = cols = 0; }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  int type() const { return type_; }
  size_t elemSize() const { return type_ == CV_32F ? 4 : 1; }
  bool isContinuous() const { return step == (size_t)cols * elemSize(); }
  template <class T> T* ptr(int r = 0) { return (T*)(data + step * (size_t)r); }
  template <class T> const T* ptr(int r = 0) const { return (const T*)(data + step * (size_t)r); }
  template <class T> T& at(int r, int c) { return ptr<T>(r)[c]; }
  template <class T> const T& at(int r, int c) const { return ptr<T>(r)[c]; }
  template <class T> T& at(int i) { return rows == 1 ? ptr<T>(0)[i] : ptr<T>(i)[0]; }  // element of a row or column vector
  template <class T> const T& at(int i) const { return rows == 1 ? ptr<T>(0)[i] : ptr<T>(i)[0]; }
  size_t total() const { return (size_t)rows * cols; }
  Mat row(int r) const { Mat m(1, cols, type_, data + step * (size_t)r, step); m.own_ = own_; return m; }
  Mat clone() const {
    Mat m(rows, cols, type_);
    for (int r = 0; r < rows; ++r) memcpy(m.data + m.step * r, data + step * r, (size_t)cols * elemSize());
    return m;
  }
 private:
  int type_ = 0;
  std::shared_ptr<uchar> own_;
};
typedef const Mat& InputArray;
typedef Mat& OutputArray;
}  // namespace cv
#endif
