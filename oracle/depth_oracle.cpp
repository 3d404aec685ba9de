// depth_oracle.cpp — CPU oracle for DepthModule (TEST INFRASTRUCTURE, see oracle.h).
//
// Restates /root/reference/src/DepthModule.cc:50-274 (+ include/DepthModule.h:138-161 diamond masks)
// and the OpenCV 4.x semantics it leans on: MatExpr GEMM with double accumulation,
// per-element fp32 reciprocal / product, threshold, dilate with an arbitrary mask, filter2D,
// distanceTransform(DIST_L2, 5x5, labels variant), copyMakeBorder, minMaxLoc.
// PARITY UNPINNED for the OpenCV-internal pieces; see oracle.h.
#include "oracle.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

namespace {

inline int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
  return p;
}

// DepthModule.cc:106-139
void project(const orc_depth_params& P, const float* cloud, int n, int ld, int w, int h,
             std::vector<float>& raw) {
  raw.assign((size_t)w * h, 0.f);
  const float* X = cloud;
  const float* Y = cloud + ld;
  const float* Z = cloud + 2 * (size_t)ld;
  const float* O = cloud + 3 * (size_t)ld;
  for (int i = 0; i < n; ++i) {
    float p[3];
    for (int r = 0; r < 3; ++r) {
      // OpenCV generic GEMM: each dot product accumulated in double in k order, rounded once
      double acc = 0.0;
      acc += (double)P.proj[4 * r + 0] * (double)X[i];
      acc += (double)P.proj[4 * r + 1] * (double)Y[i];
      acc += (double)P.proj[4 * r + 2] * (double)Z[i];
      acc += (double)P.proj[4 * r + 3] * (double)O[i];
      p[r] = (float)acc;
    }
    const float recip = 1.0f / p[2];
    const float u = p[0] * recip, v = p[1] * recip, d = p[2];
    if (u > 0 && v > 0 && u < (float)w && v < (float)h) {
      if (d > P.min_dist && d < P.max_dist) raw[(size_t)(int)v * w + (int)u] = d;  // last writer wins
    }
  }
}

// DepthModule.cc:230-274
void inverse_dilation(const orc_depth_params& P, const std::vector<float>& raw, int w, int h,
                      std::vector<float>& out) {
  const float S = P.max_dist * 1.0f;  // opt_max_dist * ScaleFactor (ScaleFactor is never parsed: 1.0)
  const float thr = S - 1;
  std::vector<float> inv((size_t)w * h);
  for (size_t i = 0; i < inv.size(); ++i) {
    float t = S - raw[i];
    inv[i] = t > thr ? 0.f : t;  // THRESH_TOZERO_INV
  }
  out.assign((size_t)w * h, 0.f);
  const int ax = P.kw / 2, ay = P.kh / 2;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      float m = -std::numeric_limits<float>::max();  // dilate border value: -DBL_MAX
      for (int ky = 0; ky < P.kh; ++ky)
        for (int kx = 0; kx < P.kw; ++kx) {
          if (!P.kernel[ky * P.kw + kx]) continue;
          const int yy = y + ky - ay, xx = x + kx - ax;
          if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
          m = std::max(m, inv[(size_t)yy * w + xx]);
        }
      float t = S - m;
      out[(size_t)y * w + x] = t > thr ? 0.f : t;
    }
}

// DepthModule.cc:200-228
void average_filtering(const orc_depth_params& P, const std::vector<float>& raw, int w, int h,
                       std::vector<float>& out) {
  const int k = P.avg_ksize, a = k / 2;
  const float coef = (float)(1.0 / (double)(k * k));
  out.assign((size_t)w * h, 0.f);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      float sum = 0.f, cnt = 0.f;
      for (int ky = 0; ky < k; ++ky)
        for (int kx = 0; kx < k; ++kx) {
          const float v = raw[(size_t)reflect101(y + ky - a, h) * w + reflect101(x + kx - a, w)];
          sum += coef * v;
          cnt += (v > 0.f) ? 1.f : 0.f;
        }
      out[(size_t)y * w + x] = sum * ((float)(k * k) / cnt);  // cnt==0 -> 0*inf = NaN, fails d>0
    }
}

// cv::distanceTransform(src, dst, labels, DIST_L2, 5): 5x5 chamfer, 16.16 fixed point (distransform.cpp)
void distance_transform_5x5(const std::vector<uint8_t>& zero_mask /*1 where src==0*/, int w, int h,
                            std::vector<float>& dist) {
  const int B = 2;
  const unsigned HV = (unsigned)lrint(1.0 * 65536), DG = (unsigned)lrint(1.4 * 65536),
                 LG = (unsigned)lrint(2.1969 * 65536);
  const unsigned INIT = (unsigned)(INT_MAX >> 2);
  const int tw = w + 2 * B, th = h + 2 * B;
  std::vector<unsigned> t((size_t)tw * th, INIT);
  auto T = [&](int y, int x) -> unsigned& { return t[(size_t)(y + B) * tw + (x + B)]; };
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      if (zero_mask[(size_t)y * w + x]) { T(y, x) = 0; continue; }
      unsigned m = INIT;
      m = std::min(m, T(y - 2, x - 1) + LG);
      m = std::min(m, T(y - 2, x + 1) + LG);
      m = std::min(m, T(y - 1, x - 2) + LG);
      m = std::min(m, T(y - 1, x - 1) + DG);
      m = std::min(m, T(y - 1, x) + HV);
      m = std::min(m, T(y - 1, x + 1) + DG);
      m = std::min(m, T(y - 1, x + 2) + LG);
      m = std::min(m, T(y, x - 1) + HV);
      T(y, x) = m;
    }
  dist.assign((size_t)w * h, 0.f);
  const float scale = 1.f / 65536;
  for (int y = h - 1; y >= 0; --y)
    for (int x = w - 1; x >= 0; --x) {
      unsigned m = T(y, x);
      if (m > HV) {
        m = std::min(m, T(y + 2, x + 1) + LG);
        m = std::min(m, T(y + 2, x - 1) + LG);
        m = std::min(m, T(y + 1, x + 2) + LG);
        m = std::min(m, T(y + 1, x + 1) + DG);
        m = std::min(m, T(y + 1, x) + HV);
        m = std::min(m, T(y + 1, x - 1) + DG);
        m = std::min(m, T(y + 1, x - 2) + LG);
        m = std::min(m, T(y, x + 1) + HV);
        T(y, x) = m;
      }
      dist[(size_t)y * w + x] = (float)m * scale;
    }
}

void gather(const orc_depth_params& P, const std::vector<float>& map, int w, const float* kp_xy,
            const float* kpun_x, int k, float* depth, float* uright) {
  // DepthModule.cc:82-104
  for (int i = 0; i < k; ++i) {
    depth[i] = -1.f;
    uright[i] = -1.f;
    const float u = kp_xy[2 * i], v = kp_xy[2 * i + 1];
    const float d = map[(size_t)(int)v * w + (int)u];
    if (d > 0) {
      depth[i] = d;
      uright[i] = kpun_x[i] - P.mbf / d;
    }
  }
}

// DepthModule.cc:145-198
void nearest_neighbor(const orc_depth_params& P, const std::vector<float>& raw, int w, int h,
                      const float* kp_xy, const float* kpun_x, int k, float* depth, float* uright) {
  const int R = (int)P.nn_radius;
  const int pw = w + 2 * R;
  std::vector<float> padded((size_t)pw * (h + 2 * R), 0.f);
  for (int y = 0; y < h; ++y)
    std::memcpy(&padded[(size_t)(y + R) * pw + R], &raw[(size_t)y * w], sizeof(float) * w);
  std::vector<uint8_t> zero_mask((size_t)w * h);
  for (size_t i = 0; i < zero_mask.size(); ++i) {
    // convertTo(CV_8U) saturates round(raw); threshold(...,0,1,BINARY_INV) -> 0 where a point landed
    long q = lrintf(raw[i]);
    zero_mask[i] = q > 0 ? 1 : 0;
  }
  std::vector<float> dist;
  distance_transform_5x5(zero_mask, w, h, dist);
  for (int i = 0; i < k; ++i) {
    depth[i] = -1.f;
    uright[i] = -1.f;
    const float u = kp_xy[2 * i], v = kp_xy[2 * i + 1];
    int sr = (int)dist[(size_t)(int)v * w + (int)u];
    float d = 0;
    if (sr >= 0 && (float)sr < P.nn_radius) {
      ++sr;
      // cv::Rect(float,float,int,int) -> int conversions of u+R-sr, v+R-sr
      const int bx = (int)(u + P.nn_radius - sr), by = (int)(v + P.nn_radius - sr);
      float mx = -std::numeric_limits<float>::max();
      for (int yy = by; yy < by + 2 * sr; ++yy)
        for (int xx = bx; xx < bx + 2 * sr; ++xx) mx = std::max(mx, padded[(size_t)yy * pw + xx]);
      d = mx;
    }
    if (d > 0) {
      depth[i] = d;
      uright[i] = kpun_x[i] - P.mbf / d;
    }
  }
}

}  // namespace

extern "C" {

int orc_depth(const orc_depth_params* P, const float* cloud, int n, int ld, int w, int h,
              const float* kp_xy, const float* kpun_x, int k, float* out_depth, float* out_uright,
              float* out_raw, float* out_processed) {
  std::vector<float> raw, processed;
  project(*P, cloud, n, ld, w, h, raw);
  if (out_raw) std::memcpy(out_raw, raw.data(), sizeof(float) * raw.size());
  switch (P->method) {
    case ORC_UPS_NEAREST:
      nearest_neighbor(*P, raw, w, h, kp_xy, kpun_x, k, out_depth, out_uright);
      break;  // ProcessedDepthMap is never written by this method (reference quirk)
    case ORC_UPS_AVERAGE:
      average_filtering(*P, raw, w, h, processed);
      gather(*P, processed, w, kp_xy, kpun_x, k, out_depth, out_uright);
      break;
    case ORC_UPS_INVDIL:
      inverse_dilation(*P, raw, w, h, processed);
      gather(*P, processed, w, kp_xy, kpun_x, k, out_depth, out_uright);
      break;
    default:
      return -1;  // None / IPBasic: the reference computes no keypoint depth
  }
  if (out_processed && !processed.empty())
    std::memcpy(out_processed, processed.data(), sizeof(float) * processed.size());
  return 0;
}

void orc_distance_transform_l2_5x5(const uint8_t* src, int w, int h, int stride, float* dst) {
  std::vector<uint8_t> zero_mask((size_t)w * h);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) zero_mask[(size_t)y * w + x] = src[(size_t)y * stride + x] == 0;
  std::vector<float> dist;
  distance_transform_5x5(zero_mask, w, h, dist);
  std::memcpy(dst, dist.data(), sizeof(float) * dist.size());
}

void orc_projection_matrix(const float K[12], const float Tr[16], float out[12]) {
  // DepthModule.cc:434 `CameraMatrix(3x4) * RotationMatrix(4x4)`: cv::gemm's small-matrix special case
  // (flags == 0, inner length 4 == D.cols; matmul.simd.hpp gemmImpl) keeps FLOAT temporaries,
  // t = a0*b0 + a1*b1 + a2*b2 + a3*b3 in one expression.  That translation unit is dispatched for AVX2 / AVX-512
  // and compiled with the compiler's default contraction, so on every FMA-capable x86-64 / aarch64 host the value is
  // the fma chain below (an SSE-only host would round each product separately; the generic double-accumulating
  // path is NOT taken for this shape).  Platform-dependent in the reference itself; see DESIGN.md section 4.
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) {
      float t = K[4 * r + 0] * Tr[c];
      for (int k = 1; k < 4; ++k) t = fmaf(K[4 * r + k], Tr[4 * k + c], t);
      out[4 * r + c] = t;
    }
}

int orc_structuring_element(int shape, int kw, int kh, uint8_t* out) {
  if (kw < 1 || kh < 1 || kw * kh > 81) return -1;
  if (shape == 3) {  // Diamond: |dx|+|dy| <= r, square, kw is used for both sides (DepthModule.h:138-161)
    if (kw != 3 && kw != 5 && kw != 7 && kw != 9) return -1;
    const int r = kw / 2;
    for (int y = 0; y < kw; ++y)
      for (int x = 0; x < kw; ++x) out[y * kw + x] = (std::abs(x - r) + std::abs(y - r) <= r) ? 1 : 0;
    return 0;
  }
  // cv::getStructuringElement, anchor = centre
  const int ax = kw / 2, ay = kh / 2;
  const int r = kh / 2, c = kw / 2;
  const double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
  for (int i = 0; i < kh; ++i) {
    int j1 = 0, j2 = 0;
    if (shape == 0 || (shape == 1 && i == ay)) j2 = kw;
    else if (shape == 1) { j1 = ax; j2 = j1 + 1; }
    else {
      const int dy = i - r;
      if (std::abs(dy) <= r) {
        const int dx = (int)lrint(c * std::sqrt((r * r - dy * dy) * inv_r2));
        j1 = std::max(c - dx, 0);
        j2 = std::min(c + dx + 1, kw);
      }
    }
    for (int j = 0; j < kw; ++j) out[i * kw + j] = (j >= j1 && j < j2) ? 1 : 0;
  }
  return 0;
}

}  // extern "C"

// LoadPointcloudBinaryMat (/root/reference/Examples/RGB-L/rgbl_kitti.cc:151-185): n records (x, y, z, reflectance) of a
// KITTI velodyne .bin file -> the 4 x n CV_32F matrix with rows x, y, z, 1 that CalculateDepthFromPcd receives.
extern "C" void orc_kitti_bin_to_cloud(const float* xyzi, int n, float* cloud4xn) {
  for (int i = 0; i < n; ++i) {
    cloud4xn[i] = xyzi[4 * (size_t)i];
    cloud4xn[(size_t)n + i] = xyzi[4 * (size_t)i + 1];
    cloud4xn[2 * (size_t)n + i] = xyzi[4 * (size_t)i + 2];
    cloud4xn[3 * (size_t)n + i] = 1.0f;
  }
}
