// OpenCV stand-in, part 2: what the reference's OWN src/DepthModule.cc needs (TEST INFRASTRUCTURE, included from
// opencv.hpp; see oracle/Makefile target `ref`).  It supplies cv::MatExpr-style arithmetic, threshold, dilate,
// filter2D, distanceTransform, copyMakeBorder(BORDER_CONSTANT), minMaxLoc, getStructuringElement and a YAML
// cv::FileStorage reader, each restating the OpenCV 4.x arithmetic listed in SURVEY.md A.7 / A.8.  What compiling
// DepthModule.cc against it pins to the reference source: the parameter parsing and its failure modes, the
// projection loop (order, guards, truncation, last-writer-wins), the three up-sampling call chains with their
// thresholds / scale constants / search-box geometry, and the keypoint depth / uRight rule.
#pragma once
#include <float.h>

#include <fstream>
#include <map>
#include <sstream>
#include <string>

namespace cv {

typedef std::string String;
struct Scalar { double v[4]; Scalar(double a = 0) : v{a, 0, 0, 0} {} };

enum { THRESH_BINARY = 0, THRESH_BINARY_INV = 1, THRESH_TRUNC = 2, THRESH_TOZERO = 3, THRESH_TOZERO_INV = 4 };
enum { MORPH_RECT = 0, MORPH_CROSS = 1, MORPH_ELLIPSE = 2 };
enum { DIST_L1 = 1, DIST_L2 = 2, DIST_C = 3 };
enum { DIST_MASK_3 = 3, DIST_MASK_5 = 5, DIST_MASK_PRECISE = 0 };
enum { BORDER_DEFAULT = BORDER_REFLECT_101 };

// An expression result.  cv::MatExpr is lazy; every expression DepthModule.cc forms is a single operation, so
// evaluating eagerly is equivalent.  What matters is the ASSIGNMENT rule: `Mat = MatExpr` computes into the
// existing buffer when shape and type already match (Mat::create is a no-op), which is how
// `P.row(0) = P.row(0).mul(1 / P.row(2))` (DepthModule.cc:118) writes through a temporary row header.
struct MatExpr {
  Mat m;
  explicit MatExpr(const Mat& r) : m(r) {}
  operator Mat() const { return m; }
};

inline Mat::Mat(const MatExpr& e) : Mat(e.m) {}
inline Mat& Mat::operator=(const MatExpr& e) {
  if (data && rows == e.m.rows && cols == e.m.cols && type() == e.m.type()) e.m.copyTo(*this);
  else *this = e.m;
  return *this;
}
inline Mat Mat::ones(int r, int c, int type) {
  Mat m(r, c, type);
  for (int y = 0; y < r; ++y)
    for (int x = 0; x < c; ++x) {
      if (type == CV_32F) m.at<float>(y, x) = 1.f;
      else if (type == CV_16U) m.at<unsigned short>(y, x) = 1;
      else m.at<uchar>(y, x) = 1;
    }
  return m;
}

// saturate_cast<uchar>(float): round half to even, clamp
inline void Mat::convertTo(Mat& dst, int rtype) const {
  assert(type() == CV_32F && rtype == CV_8U);
  Mat out(rows, cols, CV_8U);
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      const int q = cvRound(at<float>(y, x));
      out.at<uchar>(y, x) = (uchar)(q < 0 ? 0 : q > 255 ? 255 : q);
    }
  dst = out;
}

// a.mul(b): cv::multiply with scale 1 -> fp32 product
inline MatExpr Mat::mul(const Mat& b, double scale) const {
  assert(type() == CV_32F && b.type() == CV_32F && rows == b.rows && cols == b.cols && scale == 1);
  Mat r(rows, cols, CV_32F);
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) r.at<float>(y, x) = at<float>(y, x) * b.at<float>(y, x);
  return MatExpr(r);
}

// A * B: cv::gemm(A, B, 1, noArray(), 0, D).  Two code paths of gemmImpl (matmul.simd.hpp):
//  * "small" special case, flags == 0 && 2 <= len <= 4 && (len == D.cols || len == D.rows): float temporaries,
//    t = a0*b0 + a1*b1 + ... written as one expression.  The translation unit is dispatched for AVX2 / AVX-512 and
//    compiled with the compiler's default contraction, so on every FMA-capable host the sum is the fma chain
//    fma(a3,b3, fma(a2,b2, fma(a1,b1, a0*b0))) - that is what is restated here (DepthModule.cc:434 takes this path);
//  * generic GEMMSingleMul<float,double>: every dot product accumulated in double in k order, rounded once
//    (DepthModule.cc:115, 3x4 times 4xN).
inline MatExpr operator*(const Mat& A, const Mat& B) {
  assert(A.type() == CV_32F && B.type() == CV_32F && A.cols == B.rows);
  const int len = A.cols;
  Mat D(A.rows, B.cols, CV_32F);
  const bool small = 2 <= len && len <= 4 && (len == D.cols || len == D.rows);
  for (int i = 0; i < D.rows; ++i)
    for (int j = 0; j < D.cols; ++j) {
      if (small) {
        float t = A.at<float>(i, 0) * B.at<float>(0, j);
        for (int k = 1; k < len; ++k) t = fmaf(A.at<float>(i, k), B.at<float>(k, j), t);
        D.at<float>(i, j) = t;
      } else {
        double acc = 0.0;
        for (int k = 0; k < len; ++k) acc += (double)A.at<float>(i, k) * (double)B.at<float>(k, j);
        D.at<float>(i, j) = (float)acc;
      }
    }
  return MatExpr(D);
}

// s / M: cv::divide(s, M) -> fp32 `(float)s / m` (IEEE: x/0 = inf, OpenCV >= 4 for floating types)
inline MatExpr operator/(double s, const Mat& M) {
  assert(M.type() == CV_32F);
  Mat r(M.rows, M.cols, CV_32F);
  const float fs = (float)s;
  for (int y = 0; y < M.rows; ++y)
    for (int x = 0; x < M.cols; ++x) r.at<float>(y, x) = fs / M.at<float>(y, x);
  return MatExpr(r);
}
// M / s: MatExpr scaling by alpha = 1/s in double, stored with saturate_cast<float>
inline MatExpr operator/(const Mat& M, double s) {
  assert(M.type() == CV_32F);
  Mat r(M.rows, M.cols, CV_32F);
  const double alpha = 1. / s;
  for (int y = 0; y < M.rows; ++y)
    for (int x = 0; x < M.cols; ++x) r.at<float>(y, x) = (float)((double)M.at<float>(y, x) * alpha);
  return MatExpr(r);
}
// s - M: cv::subtract(Scalar(s), M): a scalar that is exactly representable in fp32 keeps the work type fp32
inline MatExpr operator-(double s, const Mat& M) {
  assert(M.type() == CV_32F && (double)(float)s == s);
  Mat r(M.rows, M.cols, CV_32F);
  const float fs = (float)s;
  for (int y = 0; y < M.rows; ++y)
    for (int x = 0; x < M.cols; ++x) r.at<float>(y, x) = fs - M.at<float>(y, x);
  return MatExpr(r);
}
inline MatExpr operator/(double s, const MatExpr& e) { return s / e.m; }
inline MatExpr operator/(const MatExpr& e, double s) { return e.m / s; }
inline MatExpr operator-(double s, const MatExpr& e) { return s - e.m; }

inline std::ostream& operator<<(std::ostream& os, const Mat& m) {
  os << "[";
  for (int y = 0; y < m.rows; ++y) {
    for (int x = 0; x < m.cols; ++x) {
      if (m.type() == CV_32F) os << m.at<float>(y, x);
      else os << (int)m.at<uchar>(y, x);
      if (x + 1 < m.cols) os << ", ";
    }
    os << (y + 1 < m.rows ? ";\n " : "");
  }
  return os << "]";
}

inline double threshold(const Mat& src_, Mat& dst, double thresh, double maxval, int type) {
  const Mat src = src_;  // dst may be src (header copy keeps the buffer alive; the operation is element-wise)
  dst.create(src.rows, src.cols, src.type());
  for (int y = 0; y < src.rows; ++y)
    for (int x = 0; x < src.cols; ++x) {
      if (src.type() == CV_32F) {
        const float v = src.at<float>(y, x), t = (float)thresh, mv = (float)maxval;
        float& d = dst.at<float>(y, x);
        switch (type) {
          case THRESH_BINARY: d = v > t ? mv : 0.f; break;
          case THRESH_BINARY_INV: d = v > t ? 0.f : mv; break;
          case THRESH_TOZERO_INV: d = v > t ? 0.f : v; break;
          default: assert(!"threshold type not provided");
        }
      } else {
        assert(src.type() == CV_8U);
        const int v = src.at<uchar>(y, x), t = cvFloor(thresh);
        const int mv = cvRound(maxval);
        uchar& d = dst.at<uchar>(y, x);
        switch (type) {
          case THRESH_BINARY: d = (uchar)(v > t ? mv : 0); break;
          case THRESH_BINARY_INV: d = (uchar)(v > t ? 0 : mv); break;
          default: assert(!"threshold type not provided");
        }
      }
    }
  return thresh;
}

// cv::dilate, fp32, arbitrary u8 mask, anchor (-1,-1) = centre, default border value (-DBL_MAX: never wins)
inline void dilate(const Mat& src_, Mat& dst, const Mat& kernel, Point anchor, int iterations) {
  assert(src_.type() == CV_32F && kernel.type() == CV_8U && iterations == 1);
  const Mat src = src_.clone();
  const int ax = anchor.x < 0 ? kernel.cols / 2 : anchor.x, ay = anchor.y < 0 ? kernel.rows / 2 : anchor.y;
  dst.create(src.rows, src.cols, CV_32F);
  for (int y = 0; y < src.rows; ++y)
    for (int x = 0; x < src.cols; ++x) {
      float m = -FLT_MAX;
      for (int ky = 0; ky < kernel.rows; ++ky)
        for (int kx = 0; kx < kernel.cols; ++kx) {
          if (!kernel.at<uchar>(ky, kx)) continue;
          const int yy = y + ky - ay, xx = x + kx - ax;
          if (yy < 0 || yy >= src.rows || xx < 0 || xx >= src.cols) continue;
          m = std::max(m, src.at<float>(yy, xx));
        }
      dst.at<float>(y, x) = m;
    }
}

// cv::filter2D (correlation), fp32 source, kernel converted to fp32, direct form: fp32 accumulation over the
// kernel taps in row-major order starting from delta, reflect-101 border
inline void filter2D(const Mat& src_, Mat& dst, int ddepth, const Mat& kernel, Point anchor, double delta, int borderType) {
  assert(src_.type() == CV_32F && ddepth == -1 && borderType == BORDER_REFLECT_101);
  const Mat src = src_.clone();
  const int ax = anchor.x < 0 ? kernel.cols / 2 : anchor.x, ay = anchor.y < 0 ? kernel.rows / 2 : anchor.y;
  dst.create(src.rows, src.cols, CV_32F);
  for (int y = 0; y < src.rows; ++y)
    for (int x = 0; x < src.cols; ++x) {
      float s = (float)delta;
      for (int ky = 0; ky < kernel.rows; ++ky)
        for (int kx = 0; kx < kernel.cols; ++kx) {
          const float kf = kernel.type() == CV_32F ? kernel.at<float>(ky, kx)
                         : kernel.type() == CV_16U ? (float)kernel.at<unsigned short>(ky, kx) : (float)kernel.at<uchar>(ky, kx);
          if (kf == 0.f) continue;
          s += kf * src.at<float>(border101(y + ky - ay, src.rows), border101(x + kx - ax, src.cols));
        }
      dst.at<float>(y, x) = s;
    }
}

inline void distanceTransform(const Mat& src_, Mat& dst, Mat& labels, int distanceType, int maskSize) {
  assert(src_.type() == CV_8U && distanceType == DIST_L2 && maskSize == DIST_MASK_5);
  const Mat src = src_.clone();
  Mat out(src.rows, src.cols, CV_32F);
  orc_distance_transform_l2_5x5(src.data, src.cols, src.rows, (int)src.step, (float*)out.data);
  dst = out;
  labels = Mat();  // never read by the reference
}

inline void copyMakeBorder(const Mat& src_, Mat& dst, int top, int bottom, int left, int right, int borderType, const Scalar& value) {
  assert(borderType == BORDER_CONSTANT && src_.type() == CV_32F);
  const Mat src = src_.clone();
  Mat out(src.rows + top + bottom, src.cols + left + right, CV_32F);
  for (int y = 0; y < out.rows; ++y)
    for (int x = 0; x < out.cols; ++x) {
      const int sy = y - top, sx = x - left;
      out.at<float>(y, x) = (sy >= 0 && sy < src.rows && sx >= 0 && sx < src.cols) ? src.at<float>(sy, sx) : (float)value.v[0];
    }
  dst = out;
}

inline void minMaxLoc(const Mat& m, double* minVal, double* maxVal) {
  assert(m.type() == CV_32F && m.rows > 0 && m.cols > 0);
  float lo = m.at<float>(0, 0), hi = lo;
  for (int y = 0; y < m.rows; ++y)
    for (int x = 0; x < m.cols; ++x) { lo = std::min(lo, m.at<float>(y, x)); hi = std::max(hi, m.at<float>(y, x)); }
  if (minVal) *minVal = lo;
  if (maxVal) *maxVal = hi;
}

inline Mat getStructuringElement(int shape, Size ksize) {
  Mat k(ksize.height, ksize.width, CV_8U);
  const int rc = orc_structuring_element(shape, ksize.width, ksize.height, k.data);
  assert(rc == 0);
  (void)rc;
  return k;
}

// ---- cv::FileStorage (READ) for the flat `key: value` YAML the reference's settings files use ----
class FileNode {
 public:
  enum { NONE = 0, INT = 1, REAL = 2, STRING = 3 };
  FileNode() : kind_(NONE), num_(0) {}
  FileNode(int kind, double num, const std::string& str) : kind_(kind), num_(num), str_(str) {}
  bool empty() const { return kind_ == NONE; }
  bool isInt() const { return kind_ == INT; }
  bool isReal() const { return kind_ == REAL; }
  bool isString() const { return kind_ == STRING; }
  double real() const { return kind_ == INT || kind_ == REAL ? num_ : kind_ == NONE ? 0.0 : DBL_MAX; }
  operator std::string() const { return kind_ == STRING ? str_ : std::string(); }
  operator int() const { return kind_ == INT || kind_ == REAL ? cvRound(num_) : kind_ == NONE ? 0 : 0x7fffffff; }
  operator float() const { return (float)real(); }
  operator double() const { return real(); }
  // sequences / maps (the yml vocabulary format of DBoW2's save / load): never present in the flat settings files read
  // here, provided so that the reference's vendored DBoW2 compiles; its text loader (loadFromTextFile) is what runs
  FileNode operator[](const char*) const { return FileNode(); }
  FileNode operator[](const std::string&) const { return FileNode(); }
  FileNode operator[](int) const { return FileNode(); }
  size_t size() const { return 0; }
 private:
  int kind_;
  double num_;
  std::string str_;
};

class FileStorage {
 public:
  enum { READ = 0, WRITE = 1 };
  template <class T> FileStorage& operator<<(const T&) { return *this; }  // WRITE mode is a sink
  FileStorage(const std::string& path, int mode) {
    if (mode != READ) return;
    std::ifstream f(path);
    opened_ = f.good();
    std::string line;
    while (std::getline(f, line)) {
      if (line.empty() || line[0] == '%' || line[0] == '#' || line[0] == '-') continue;
      const size_t colon = line.find(':');
      if (colon == std::string::npos) continue;
      std::string key = trim(line.substr(0, colon)), val = line.substr(colon + 1);
      bool quoted = false;
      size_t q = val.find_first_not_of(" \t");
      if (q != std::string::npos && (val[q] == '"' || val[q] == '\'')) {
        const size_t e = val.find(val[q], q + 1);
        val = val.substr(q + 1, e == std::string::npos ? std::string::npos : e - q - 1);
        quoted = true;
      } else {
        const size_t hash = val.find('#');
        if (hash != std::string::npos) val = val.substr(0, hash);
        val = trim(val);
      }
      if (key.empty() || (val.empty() && !quoted)) continue;
      nodes_[key] = classify(val, quoted);
    }
  }
  bool isOpened() const { return opened_; }
  FileNode operator[](const char* key) const { auto it = nodes_.find(key); return it == nodes_.end() ? FileNode() : it->second; }
  FileNode operator[](const std::string& key) const { return (*this)[key.c_str()]; }
  void release() {}
 private:
  static std::string trim(const std::string& s) {
    const size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
  }
  // OpenCV's YAML scalar rule: an optionally signed digit string is an INT node unless it continues with
  // '.', 'e' or 'E', in which case it is a REAL node; everything else is a string
  static FileNode classify(const std::string& v, bool quoted) {
    if (!quoted) {
      size_t i = (v[0] == '+' || v[0] == '-') ? 1 : 0;
      if (i < v.size() && (isdigit((unsigned char)v[i]) || (v[i] == '.' && i + 1 < v.size() && isdigit((unsigned char)v[i + 1])))) {
        size_t j = i;
        while (j < v.size() && isdigit((unsigned char)v[j])) ++j;
        char* end = nullptr;
        if (j < v.size() && (v[j] == '.' || v[j] == 'e' || v[j] == 'E')) {
          const double d = strtod(v.c_str(), &end);
          if (end && *end == 0) return FileNode(FileNode::REAL, d, v);
        } else {
          const long l = strtol(v.c_str(), &end, 0);
          if (end && *end == 0) return FileNode(FileNode::INT, (double)l, v);
        }
      }
    }
    return FileNode(FileNode::STRING, 0, v);
  }
  bool opened_ = false;
  std::map<std::string, FileNode> nodes_;
};

}  // namespace cv

#pragma GCC pop_options  // pushed in opencv.hpp
