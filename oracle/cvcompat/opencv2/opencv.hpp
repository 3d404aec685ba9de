// Minimal OpenCV stand-in used ONLY to compile the reference's own src/ORBextractor.cc, unmodified and where it
// lies under /root/reference, into oracle/_ref/ (TEST INFRASTRUCTURE; see oracle/Makefile target `ref`).
//
// It provides the container / geometry types that file uses (Mat with ROI semantics, Point_, Size, Rect, KeyPoint,
// InputArray / OutputArray) and forwards the five OpenCV algorithms it calls — cv::FAST, cv::resize,
// cv::copyMakeBorder, cv::GaussianBlur, cv::fastAtan2 — to this repo's CPU oracle (oracle/orb_oracle.cpp).
// Consequently oracle/_ref pins everything in the extractor that is NOT OpenCV-internal (detection-cell loop,
// two-threshold rule, quad-tree with the real std::list / std::sort, IC_Angle, steering + the pattern table,
// level scaling, lapping-area packing) against the reference's actual source; the OpenCV-internal arithmetic
// stays "restated from upstream" (SURVEY.md Appendix A).
#pragma once
#include <assert.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>  // the real headers pull these in; ORBextractor.cc relies on it (std::sort, std::cout)
#include <iostream>
#include <memory>
#include <vector>

#include "../../oracle.h"

// The stand-ins below stand for OpenCV's separately compiled library: their arithmetic must not follow the flags of the
// translation unit that includes them (oracle/Makefile builds the reference's files a second time with the reference's own
// -O3 -march=native, where GCC contracts a * b + c into FMAs).  Popped at the end of depth_api.hpp.
#pragma GCC push_options
#pragma GCC optimize("fp-contract=off")

#define CV_PI 3.1415926535897932384626433832795
#define CV_8U 0
#define CV_8UC1 0
#define CV_16U 2
#define CV_32F 5

typedef unsigned char uchar;

inline int cvRound(double v) { return (int)lrint(v); }
inline int cvRound(float v) { return (int)lrintf(v); }
inline int cvRound(int v) { return v; }
inline int cvFloor(double v) { int i = (int)v; return i - (v < i); }
inline int cvCeil(double v) { int i = (int)v; return i + (v > i); }

namespace cv {
using ::uchar;
using ::cvRound;

template <class T> struct Point_ {
  T x, y;
  Point_() : x(0), y(0) {}
  Point_(T x_, T y_) : x(x_), y(y_) {}
  template <class U> Point_(const Point_<U>& o) : x((T)o.x), y((T)o.y) {}
  Point_& operator*=(float s) { x = (T)(x * s); y = (T)(y * s); return *this; }  // cv: saturate_cast<T>(x * s)
};
typedef Point_<int> Point2i;
typedef Point_<int> Point;
typedef Point_<float> Point2f;
struct Size { int width, height; Size() : width(0), height(0) {} Size(int w, int h) : width(w), height(h) {} };
struct Rect { int x, y, width, height; Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {} };

struct KeyPoint {
  Point2f pt;
  float size, angle, response;
  int octave, class_id;
  KeyPoint() : size(0), angle(-1), response(0), octave(0), class_id(-1) {}
  KeyPoint(float x, float y, float s, float a = -1, float r = 0, int o = 0, int c = -1)
      : pt(x, y), size(s), angle(a), response(r), octave(o), class_id(c) {}
};
struct KeyPointsFilter {  // only referenced by the dead ComputeKeyPointsOld
  static void retainBest(std::vector<KeyPoint>&, int) { assert(!"KeyPointsFilter::retainBest is not provided"); }
};

struct MatExpr;  // depth_api.hpp: eagerly evaluated expression result with cv::MatExpr's assignment semantics
class Mat {
 public:
  int rows, cols;
  size_t step;
  uchar* data;
  Mat() : rows(0), cols(0), step(0), data(nullptr), type_(0) {}
  Mat(int r, int c, int type) : Mat() { create(r, c, type); }
  Mat(Size s, int type) : Mat() { create(s.height, s.width, type); }
  Mat(Size s, int type, void* user) : rows(s.height), cols(s.width), step(0), data((uchar*)user), type_(type) { step = (size_t)cols * elemSize(); }
  inline Mat(const MatExpr& e);
  inline Mat& operator=(const MatExpr& e);  // writes INTO an existing same-shape Mat (so `m.row(0) = expr` works)
  inline MatExpr mul(const Mat& m, double scale = 1) const;
  inline void convertTo(Mat& dst, int rtype) const;
  static inline Mat ones(int r, int c, int type);
  static Mat zeros(Size s, int type) { return Mat(s, type); }
  Size size() const { return Size(cols, rows); }
  void create(int r, int c, int type) {
    if (data && r == rows && c == cols && type == type_) return;
    rows = r; cols = c; type_ = type;
    step = (size_t)c * elemSize();
    buf_ = std::shared_ptr<uchar>(new uchar[step * (size_t)r + 64](), std::default_delete<uchar[]>());
    data = buf_.get();
  }
  static Mat zeros(int r, int c, int type) { return Mat(r, c, type); }
  size_t elemSize() const { return type_ == CV_32F ? 4 : type_ == CV_16U ? 2 : 1; }
  size_t step1() const { return step / elemSize(); }
  int type() const { return type_; }
  bool empty() const { return data == nullptr || rows * cols == 0; }
  void release() { buf_.reset(); data = nullptr; rows = cols = 0; }
  Mat operator()(const Rect& r) const { Mat m = *this; m.data = data + (size_t)r.y * step + (size_t)r.x * elemSize(); m.rows = r.height; m.cols = r.width; return m; }
  Mat rowRange(int a, int b) const { return (*this)(Rect(0, a, cols, b - a)); }
  Mat colRange(int a, int b) const { return (*this)(Rect(a, 0, b - a, rows)); }
  Mat row(int r) const { return rowRange(r, r + 1); }
  template <class T> T& at(int r, int c) { return ((T*)(data + (size_t)r * step))[c]; }
  template <class T> const T& at(int r, int c) const { return ((const T*)(data + (size_t)r * step))[c]; }
  // element i of a vector stored as a row or as a column (cv::Mat::at(int i0))
  template <class T> T& at(int i) { return rows == 1 ? at<T>(0, i) : at<T>(i, 0); }
  template <class T> const T& at(int i) const { return rows == 1 ? at<T>(0, i) : at<T>(i, 0); }
  size_t total() const { return (size_t)rows * cols; }
  template <class T> T* ptr(int r = 0) { return (T*)(data + (size_t)r * step); }
  template <class T> const T* ptr(int r = 0) const { return (const T*)(data + (size_t)r * step); }
  uchar* ptr(int r = 0) { return data + (size_t)r * step; }
  const uchar* ptr(int r = 0) const { return data + (size_t)r * step; }
  Mat clone() const {
    Mat m(rows, cols, type_);
    for (int r = 0; r < rows; ++r) memcpy(m.data + (size_t)r * m.step, data + (size_t)r * step, (size_t)cols * elemSize());
    return m;
  }
  void copyTo(Mat dst) const {  // dst is a view with the right shape (desc.row(i).copyTo(descriptors.row(k)))
    for (int r = 0; r < rows; ++r) memcpy(dst.data + (size_t)r * dst.step, data + (size_t)r * step, (size_t)cols * elemSize());
  }
  // InputArray / OutputArray duck typing
  Mat getMat() const { return *this; }
 private:
  int type_;
  std::shared_ptr<uchar> buf_;
};
typedef const Mat& InputArray;
typedef Mat& OutputArray;

enum { BORDER_CONSTANT = 0, BORDER_REPLICATE = 1, BORDER_REFLECT = 2, BORDER_WRAP = 3, BORDER_REFLECT_101 = 4, BORDER_ISOLATED = 16 };
enum { INTER_NEAREST = 0, INTER_LINEAR = 1 };

inline float fastAtan2(float y, float x) { return orc_fast_atan2(y, x); }

inline void FAST(const Mat& img, std::vector<KeyPoint>& kps, int threshold, bool nonmax) {
  std::vector<orc_keypoint> tmp((size_t)img.rows * img.cols + 1);
  const int n = orc_fast(img.data, img.cols, img.rows, (int)img.step, threshold, nonmax ? 1 : 0, tmp.data(), (int)tmp.size());
  kps.clear();
  for (int i = 0; i < n; ++i) kps.push_back(KeyPoint(tmp[i].x, tmp[i].y, tmp[i].size, tmp[i].angle, tmp[i].response, tmp[i].octave, tmp[i].class_id));
}

inline void resize(const Mat& src, Mat& dst, Size dsize, double, double, int interpolation) {
  assert(interpolation == INTER_LINEAR && src.type() == CV_8UC1);
  dst.create(dsize.height, dsize.width, src.type());  // no-op for the ROI the reference passes in
  orc_resize_linear_u8(src.data, src.cols, src.rows, (int)src.step, dst.data, dst.cols, dst.rows, (int)dst.step);
}

inline int border101(int p, int len) { while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p; return p; }
inline void copyMakeBorder(const Mat& src_, Mat& dst, int top, int bottom, int left, int right, int borderType) {
  assert((borderType & ~BORDER_ISOLATED) == BORDER_REFLECT_101);
  const Mat src = src_.clone();  // src may be an ROI of dst (ORBextractor.cc:1185)
  dst.create(src.rows + top + bottom, src.cols + left + right, src.type());
  for (int y = 0; y < dst.rows; ++y) {
    const uchar* s = src.ptr(border101(y - top, src.rows));
    uchar* d = dst.ptr(y);
    for (int x = 0; x < dst.cols; ++x) d[x] = s[border101(x - left, src.cols)];
  }
}

inline void GaussianBlur(const Mat& src_, Mat& dst, Size ksize, double sx, double sy, int borderType) {
  assert(ksize.width == 7 && ksize.height == 7 && sx == 2 && sy == 2 && borderType == BORDER_REFLECT_101);
  const Mat src = src_.clone();
  dst.create(src.rows, src.cols, src.type());
  orc_gaussian_blur7_u8(src.data, src.cols, src.rows, (int)src.step, dst.data, (int)dst.step);
}

}  // namespace cv

#include "depth_api.hpp"  // what the reference's src/DepthModule.cc needs on top
