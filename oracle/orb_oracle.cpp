// orb_oracle.cpp — CPU oracle for the ORB extraction path (TEST INFRASTRUCTURE, see oracle.h).
//
// Restates, without OpenCV:
//   * ORB_SLAM3::ORBextractor            /root/reference/src/ORBextractor.cc:71-146, 409-896, 1077-1195
//   * cv::resize INTER_LINEAR u8         OpenCV 4.x modules/imgproc/src/resize.cpp          
//   * cv::GaussianBlur 7x7 sigma 2 u8    OpenCV 4.x smooth.dispatch.cpp fixed-point path    
//   * cv::FAST TYPE_9_16 + cornerScore   OpenCV 4.x modules/features2d/src/fast{,_score}.cpp
//   * cv::fastAtan2, cvRound             OpenCV 4.x core mathfuncs / fast_math               
// PARITY UNPINNED for the OpenCV-internal pieces (no OpenCV in this image); see oracle.h.
// Build: g++ -O2 -ffp-contract=off (no FMA contraction anywhere).
#include "oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <list>
#include <utility>
#include <vector>

#include "../orb_slam3_rgbl_amd/csrc/brief_pattern.h"

namespace {

// ---------------------------------------------------------------- A.0 rounding helpers
inline int cv_round(double v) { return (int)lrint(v); }  // round-half-even under the default FP mode
inline int cv_round(float v) { return (int)lrintf(v); }
inline int cv_floor(float v) {
  int i = (int)v;
  return i - (v < (float)i);
}
inline int cv_ceil(float v) {
  int i = (int)v;
  return i + (v > (float)i);
}
inline int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) {
    if (p < 0) p = -p;
    else p = 2 * (len - 1) - p;
  }
  return p;
}

struct Image {
  int w = 0, h = 0;
  std::vector<uint8_t> px;
  void alloc(int w_, int h_) { w = w_; h = h_; px.assign((size_t)w * h, 0); }
  const uint8_t* row(int y) const { return px.data() + (size_t)y * w; }
  uint8_t* row(int y) { return px.data() + (size_t)y * w; }
};

// ---------------------------------------------------------------- A.1 cv::resize (INTER_LINEAR, CV_8UC1)
void resize_linear(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh,
                   int dstride) {
  const int COEF = 2048;  // INTER_RESIZE_COEF_SCALE
  const double inv_sx = (double)dw / sw, inv_sy = (double)dh / sh;
  const double scale_x = 1.0 / inv_sx, scale_y = 1.0 / inv_sy;
  std::vector<int> xofs(dw), yofs(dh);
  std::vector<short> xa(2 * (size_t)dw), ya(2 * (size_t)dh);
  for (int dx = 0; dx < dw; ++dx) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cv_floor(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    xofs[dx] = sx;
    xa[2 * dx] = (short)cv_round((1.f - fx) * COEF);
    xa[2 * dx + 1] = (short)cv_round(fx * COEF);
  }
  for (int dy = 0; dy < dh; ++dy) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cv_floor(fy);
    fy -= sy;
    yofs[dy] = sy;
    ya[2 * dy] = (short)cv_round((1.f - fy) * COEF);
    ya[2 * dy + 1] = (short)cv_round(fy * COEF);
  }
  std::vector<int> r0(dw), r1(dw);
  auto hline = [&](int sy, std::vector<int>& out) {
    sy = std::min(std::max(sy, 0), sh - 1);
    const uint8_t* S = src + (size_t)sy * sstride;
    for (int dx = 0; dx < dw; ++dx) {
      int sx = xofs[dx];
      int s1 = (sx + 1 < sw) ? S[sx + 1] : 0;  // weight is 0 whenever sx is the last column
      out[dx] = S[sx] * xa[2 * dx] + s1 * xa[2 * dx + 1];
    }
  };
  for (int dy = 0; dy < dh; ++dy) {
    hline(yofs[dy], r0);
    hline(yofs[dy] + 1, r1);
    const int b0 = ya[2 * dy], b1 = ya[2 * dy + 1];
    uint8_t* D = dst + (size_t)dy * dstride;
    for (int dx = 0; dx < dw; ++dx)
      D[dx] = (uint8_t)((((b0 * (r0[dx] >> 4)) >> 16) + ((b1 * (r1[dx] >> 4)) >> 16) + 2) >> 2);
  }
}

// ---------------------------------------------------------------- A.2 GaussianBlur 7x7, sigma 2, u8
// 8.8 fixed-point kernel produced by OpenCV's error-diffusing quantiser; sums to 256.
const int kGauss7[7] = {18, 34, 48, 56, 48, 34, 18};

void gaussian_blur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
  std::vector<uint16_t> hbuf((size_t)w * h);
  for (int y = 0; y < h; ++y) {
    const uint8_t* S = src + (size_t)y * sstride;
    for (int x = 0; x < w; ++x) {
      unsigned acc = 0;
      for (int k = -3; k <= 3; ++k) acc += kGauss7[k + 3] * S[reflect101(x + k, w)];
      hbuf[(size_t)y * w + x] = (uint16_t)acc;  // <= 255*256, exact
    }
  }
  for (int y = 0; y < h; ++y) {
    uint8_t* D = dst + (size_t)y * dstride;
    for (int x = 0; x < w; ++x) {
      uint32_t acc = 0;
      for (int k = -3; k <= 3; ++k) acc += (uint32_t)kGauss7[k + 3] * hbuf[(size_t)reflect101(y + k, h) * w + x];
      D[x] = (uint8_t)((acc + 32768u) >> 16);
    }
  }
}

// ---------------------------------------------------------------- A.3 cv::FAST TYPE_9_16
const int kRingDx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
const int kRingDy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

int corner_score16(const uint8_t* p, int stride, int threshold) {
  int d[25];
  const int v = p[0];
  for (int k = 0; k < 25; ++k) d[k] = v - p[kRingDy[k & 15] * stride + kRingDx[k & 15]];
  int a0 = threshold;
  for (int k = 0; k < 16; k += 2) {
    int a = std::min(d[k + 1], d[k + 2]);
    a = std::min(a, d[k + 3]);
    if (a <= a0) continue;
    for (int j = 4; j <= 8; ++j) a = std::min(a, d[k + j]);
    a0 = std::max(a0, std::min(a, d[k]));
    a0 = std::max(a0, std::min(a, d[k + 9]));
  }
  int b0 = -a0;
  for (int k = 0; k < 16; k += 2) {
    int b = std::max(d[k + 1], d[k + 2]);
    b = std::max(b, d[k + 3]);
    b = std::max(b, d[k + 4]);
    b = std::max(b, d[k + 5]);
    if (b >= b0) continue;
    for (int j = 6; j <= 8; ++j) b = std::max(b, d[k + j]);
    b0 = std::min(b0, std::max(b, d[k]));
    b0 = std::min(b0, std::max(b, d[k + 9]));
  }
  return -b0 - 1;
}

// Segment test exactly as FAST_t<16>: 9 contiguous ring pixels (of the 25-long wrapped ring) all darker
// than v-t or all brighter than v+t.
bool is_corner16(const uint8_t* p, int stride, int t) {
  const int v = p[0];
  int run_dark = 0, run_bright = 0;
  for (int k = 0; k < 25; ++k) {
    const int x = p[kRingDy[k & 15] * stride + kRingDx[k & 15]];
    if (x < v - t) { if (++run_dark > 8) return true; } else run_dark = 0;
    if (x > v + t) { if (++run_bright > 8) return true; } else run_bright = 0;
  }
  return false;
}

void fast9_16(const uint8_t* img, int w, int h, int stride, int threshold, bool nonmax,
              std::vector<orc_keypoint>& out) {
  out.clear();
  threshold = std::min(std::max(threshold, 0), 255);
  if (w < 7 || h < 7) return;
  // rolling 3-row score buffers, as the upstream implementation (outside the scanned range = 0)
  std::vector<uint8_t> buf[3];
  std::vector<int> corners[3];
  for (auto& b : buf) b.assign(w, 0);
  for (int i = 3; i < h - 2; ++i) {
    std::vector<uint8_t>& curr = buf[(i - 3) % 3];
    std::vector<int>& cpos = corners[(i - 3) % 3];
    std::fill(curr.begin(), curr.end(), 0);
    cpos.clear();
    if (i < h - 3) {
      for (int j = 3; j < w - 3; ++j) {
        const uint8_t* p = img + (size_t)i * stride + j;
        if (is_corner16(p, stride, threshold)) {
          cpos.push_back(j);
          if (nonmax) curr[j] = (uint8_t)corner_score16(p, stride, threshold);
        }
      }
    }
    if (i == 3) continue;
    const std::vector<uint8_t>& prev = buf[(i - 4 + 3) % 3];
    const std::vector<uint8_t>& pprev = buf[(i - 5 + 3) % 3];
    for (int j : corners[(i - 4 + 3) % 3]) {
      const int s = prev[j];
      if (!nonmax || (s > prev[j + 1] && s > prev[j - 1] && s > pprev[j - 1] && s > pprev[j] &&
                      s > pprev[j + 1] && s > curr[j - 1] && s > curr[j] && s > curr[j + 1])) {
        orc_keypoint kp{(float)j, (float)(i - 1), 7.f, -1.f, (float)s, 0, -1};
        out.push_back(kp);
      }
    }
  }
}

// ---------------------------------------------------------------- A.5 cv::fastAtan2 (degrees)
float fast_atan2(float y, float x) {
  const float scale = (float)(180.0 / 3.14159265358979323846);
  const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale,
              p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
  const float ax = std::fabs(x), ay = std::fabs(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// ---------------------------------------------------------------- ORBextractor restatement
const int kPatch = 31, kHalfPatch = 15, kEdge = 19;

void compute_umax(int* umax /*16*/) {
  // ORBextractor.cc:451-468
  const int vmax = cv_floor(kHalfPatch * std::sqrt(2.f) / 2 + 1);
  const int vmin = cv_ceil(kHalfPatch * std::sqrt(2.f) / 2);
  const double hp2 = kHalfPatch * kHalfPatch;
  for (int v = 0; v <= vmax; ++v) umax[v] = cv_round(std::sqrt(hp2 - v * v));
  for (int v = kHalfPatch, v0 = 0; v >= vmin; --v) {
    while (umax[v0] == umax[v0 + 1]) ++v0;
    umax[v] = v0;
    ++v0;
  }
}

float ic_angle(const uint8_t* center, int stride, const int* umax) {
  // ORBextractor.cc:76-103 — integer moments over the circular patch, un-blurred level
  int m01 = 0, m10 = 0;
  for (int u = -kHalfPatch; u <= kHalfPatch; ++u) m10 += u * center[u];
  for (int v = 1; v <= kHalfPatch; ++v) {
    int vsum = 0;
    const int d = umax[v];
    for (int u = -d; u <= d; ++u) {
      const int below = center[u + v * stride], above = center[u - v * stride];
      vsum += below - above;
      m10 += u * (below + above);
    }
    m01 += v * vsum;
  }
  return fast_atan2((float)m01, (float)m10);
}

void brief_descriptor(const uint8_t* center, int stride, float angle_deg, uint8_t* desc) {
  // ORBextractor.cc:105-146
  const float factorPI = (float)(3.14159265358979323846 / 180.f);
  const float angle = angle_deg * factorPI;
  const float a = cosf(angle), b = sinf(angle);  // glibc float overloads, as the reference binds them
  const int8_t* pat = rgbl::kBriefPattern;
  for (int i = 0; i < 32; ++i) {
    int val = 0;
    for (int k = 0; k < 8; ++k, pat += 4) {
      const int x0 = pat[0], y0 = pat[1], x1 = pat[2], y1 = pat[3];
      const int t0 = center[cv_round(x0 * b + y0 * a) * stride + cv_round(x0 * a - y0 * b)];
      const int t1 = center[cv_round(x1 * b + y1 * a) * stride + cv_round(x1 * a - y1 * b)];
      val |= (t0 < t1) << k;
    }
    desc[i] = (uint8_t)val;
  }
}

// --- quad-tree distribution, ORBextractor.cc:480-779 -------------------------------------------
struct QNode {
  int x0, x1, y0, y1;  // UL=(x0,y0) UR=(x1,y0) BL=(x0,y1) BR=(x1,y1)
  std::vector<orc_keypoint> keys;
  std::list<QNode>::iterator self;
  bool leaf = false;  // bNoMore
};

void split_node(const QNode& n, QNode c[4]) {
  const int hx = (int)std::ceil(static_cast<float>(n.x1 - n.x0) / 2);
  const int hy = (int)std::ceil(static_cast<float>(n.y1 - n.y0) / 2);
  const int mx = n.x0 + hx, my = n.y0 + hy;
  c[0].x0 = n.x0; c[0].x1 = mx;   c[0].y0 = n.y0; c[0].y1 = my;
  c[1].x0 = mx;   c[1].x1 = n.x1; c[1].y0 = n.y0; c[1].y1 = my;
  c[2].x0 = n.x0; c[2].x1 = mx;   c[2].y0 = my;   c[2].y1 = n.y1;
  c[3].x0 = mx;   c[3].x1 = n.x1; c[3].y0 = my;   c[3].y1 = n.y1;
  for (int q = 0; q < 4; ++q) c[q].keys.reserve(n.keys.size());
  for (const orc_keypoint& kp : n.keys) {
    const bool left = kp.x < (float)mx, top = kp.y < (float)my;
    c[left ? (top ? 0 : 2) : (top ? 1 : 3)].keys.push_back(kp);
  }
  for (int q = 0; q < 4; ++q) c[q].leaf = (c[q].keys.size() == 1);
}

typedef std::pair<int, QNode*> SizedNode;
bool sized_node_less(const SizedNode& a, const SizedNode& b) {
  if (a.first != b.first) return a.first < b.first;
  return a.second->x0 < b.second->x0;
}

std::vector<orc_keypoint> distribute_octree(const std::vector<orc_keypoint>& cand, int min_x, int max_x,
                                            int min_y, int max_y, int N, int reserve_hint) {
  std::vector<orc_keypoint> result;
  const int n_ini = (int)std::round(static_cast<float>(max_x - min_x) / (max_y - min_y));
  if (n_ini < 1) return result;  // the reference would divide by zero here (portrait images)
  const float hX = static_cast<float>(max_x - min_x) / n_ini;

  std::list<QNode> nodes;
  std::vector<QNode*> roots(n_ini);
  for (int i = 0; i < n_ini; ++i) {
    QNode r;
    r.x0 = (int)(hX * static_cast<float>(i));
    r.x1 = (int)(hX * static_cast<float>(i + 1));
    r.y0 = 0;
    r.y1 = max_y - min_y;
    nodes.push_back(r);
    roots[i] = &nodes.back();
  }
  for (const orc_keypoint& kp : cand) roots[(size_t)(kp.x / hX)]->keys.push_back(kp);
  for (auto it = nodes.begin(); it != nodes.end();) {
    if (it->keys.size() == 1) { it->leaf = true; ++it; }
    else if (it->keys.empty()) it = nodes.erase(it);
    else ++it;
  }

  std::vector<SizedNode> expandable;
  // push the non-empty children of `parent` to the list front (order n1..n4), record the splittable ones
  auto push_children = [&](const QNode& parent, int* n_to_expand) {
    QNode c[4];
    split_node(parent, c);
    for (int q = 0; q < 4; ++q) {
      if (c[q].keys.empty()) continue;
      nodes.push_front(c[q]);
      if (nodes.front().keys.size() > 1) {
        if (n_to_expand) ++*n_to_expand;
        expandable.push_back(std::make_pair((int)nodes.front().keys.size(), &nodes.front()));
        nodes.front().self = nodes.begin();
      }
    }
  };

  bool done = false;
  while (!done) {
    const int prev_size = (int)nodes.size();
    int n_to_expand = 0;
    expandable.clear();
    for (auto it = nodes.begin(); it != nodes.end();) {
      if (it->leaf) { ++it; continue; }
      push_children(*it, &n_to_expand);
      it = nodes.erase(it);
    }
    if ((int)nodes.size() >= N || (int)nodes.size() == prev_size) {
      done = true;
    } else if ((int)nodes.size() + n_to_expand * 3 > N) {
      while (!done) {
        const int prev = (int)nodes.size();
        std::vector<SizedNode> todo = expandable;
        expandable.clear();
        std::sort(todo.begin(), todo.end(), sized_node_less);
        for (int j = (int)todo.size() - 1; j >= 0; --j) {
          push_children(*todo[j].second, nullptr);
          nodes.erase(todo[j].second->self);
          if ((int)nodes.size() >= N) break;
        }
        if ((int)nodes.size() >= N || (int)nodes.size() == prev) done = true;
      }
    }
  }

  result.reserve(reserve_hint);
  for (const QNode& n : nodes) {
    const orc_keypoint* best = &n.keys[0];
    for (size_t k = 1; k < n.keys.size(); ++k)
      if (n.keys[k].response > best->response) best = &n.keys[k];
    result.push_back(*best);
  }
  return result;
}

}  // namespace

// ================================================================= extractor object
struct orc_extractor {
  int nfeatures, nlevels, ini_th, min_th;
  float scale_factor;
  std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
  std::vector<int> per_level;
  int umax[16];
  // intermediates of the last call
  std::vector<Image> pyr, blurred;
  std::vector<std::vector<orc_keypoint>> cand, keys;
};

extern "C" {

orc_extractor* orc_extractor_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th) {
  orc_extractor* e = new orc_extractor;
  e->nfeatures = nfeatures; e->nlevels = nlevels; e->ini_th = ini_th; e->min_th = min_th;
  e->scale_factor = scale_factor;
  e->scale.resize(nlevels); e->sigma2.resize(nlevels);
  e->inv_scale.resize(nlevels); e->inv_sigma2.resize(nlevels);
  e->scale[0] = 1.0f; e->sigma2[0] = 1.0f;
  for (int i = 1; i < nlevels; ++i) {
    e->scale[i] = e->scale[i - 1] * scale_factor;
    e->sigma2[i] = e->scale[i] * e->scale[i];
  }
  for (int i = 0; i < nlevels; ++i) {
    e->inv_scale[i] = 1.0f / e->scale[i];
    e->inv_sigma2[i] = 1.0f / e->sigma2[i];
  }
  e->per_level.resize(nlevels);
  const float factor = 1.0f / scale_factor;
  float desired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
  int sum = 0;
  for (int l = 0; l < nlevels - 1; ++l) {
    e->per_level[l] = cv_round(desired);
    sum += e->per_level[l];
    desired *= factor;
  }
  e->per_level[nlevels - 1] = std::max(nfeatures - sum, 0);
  compute_umax(e->umax);
  return e;
}

void orc_extractor_destroy(orc_extractor* e) { delete e; }

void orc_extractor_tables(const orc_extractor* e, float* scale, float* inv_scale, float* sigma2,
                          float* inv_sigma2, int* per_level, int* umax16) {
  for (int i = 0; i < e->nlevels; ++i) {
    if (scale) scale[i] = e->scale[i];
    if (inv_scale) inv_scale[i] = e->inv_scale[i];
    if (sigma2) sigma2[i] = e->sigma2[i];
    if (inv_sigma2) inv_sigma2[i] = e->inv_sigma2[i];
    if (per_level) per_level[i] = e->per_level[i];
  }
  if (umax16) std::memcpy(umax16, e->umax, sizeof(e->umax));
}

int orc_extract(orc_extractor* e, const uint8_t* img, int w, int h, int stride, int lap0, int lap1,
                orc_keypoint* kps, uint8_t* desc, int cap, int* n_out) {
  if (n_out) *n_out = 0;
  if (!img || w <= 0 || h <= 0) return -1;
  const int L = e->nlevels;
  // ---- ComputePyramid (ORBextractor.cc:1170-1195); the 19-px frame is produced on demand only
  e->pyr.assign(L, Image());
  e->blurred.assign(L, Image());
  for (int l = 0; l < L; ++l) {
    const float s = e->inv_scale[l];
    const int lw = cv_round((float)w * s), lh = cv_round((float)h * s);
    e->pyr[l].alloc(lw, lh);
    if (l == 0) {
      for (int y = 0; y < h; ++y) std::memcpy(e->pyr[0].row(y), img + (size_t)y * stride, w);
    } else {
      resize_linear(e->pyr[l - 1].px.data(), e->pyr[l - 1].w, e->pyr[l - 1].h, e->pyr[l - 1].w,
                    e->pyr[l].px.data(), lw, lh, lw);
    }
  }
  // ---- ComputeKeyPointsOctTree (ORBextractor.cc:781-896)
  e->cand.assign(L, {});
  e->keys.assign(L, {});
  const float W = 35;
  std::vector<orc_keypoint> cell;
  for (int l = 0; l < L; ++l) {
    const Image& im = e->pyr[l];
    const int minBX = kEdge - 3, minBY = minBX;
    const int maxBX = im.w - kEdge + 3, maxBY = im.h - kEdge + 3;
    const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
    const int nCols = (int)(width / W), nRows = (int)(height / W);
    if (nCols < 1 || nRows < 1) continue;  // the reference would divide by zero; level too small
    const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
    std::vector<orc_keypoint>& cand = e->cand[l];
    for (int i = 0; i < nRows; ++i) {
      const float iniY = (float)(minBY + i * hCell);
      float maxY = iniY + hCell + 6;
      if (iniY >= maxBY - 3) continue;
      if (maxY > maxBY) maxY = (float)maxBY;
      for (int j = 0; j < nCols; ++j) {
        const float iniX = (float)(minBX + j * wCell);
        float maxX = iniX + wCell + 6;
        if (iniX >= maxBX - 6) continue;
        if (maxX > maxBX) maxX = (float)maxBX;
        const int x0 = (int)iniX, y0 = (int)iniY, cw = (int)maxX - x0, ch = (int)maxY - y0;
        const uint8_t* sub = im.row(y0) + x0;
        fast9_16(sub, cw, ch, im.w, e->ini_th, true, cell);
        if (cell.empty()) fast9_16(sub, cw, ch, im.w, e->min_th, true, cell);
        for (orc_keypoint kp : cell) {
          kp.x += j * wCell;
          kp.y += i * hCell;
          cand.push_back(kp);
        }
      }
    }
    std::vector<orc_keypoint>& keys = e->keys[l];
    keys = distribute_octree(cand, minBX, maxBX, minBY, maxBY, e->per_level[l], e->nfeatures);
    const int scaled_patch = (int)(kPatch * e->scale[l]);
    for (orc_keypoint& kp : keys) {
      kp.x += minBX;
      kp.y += minBY;
      kp.octave = l;
      kp.size = (float)scaled_patch;
    }
  }
  for (int l = 0; l < L; ++l)
    for (orc_keypoint& kp : e->keys[l])
      kp.angle = ic_angle(e->pyr[l].row(cv_round(kp.y)) + cv_round(kp.x), e->pyr[l].w, e->umax);

  // ---- descriptors + packing (ORBextractor.cc:1103-1167)
  int total = 0;
  for (int l = 0; l < L; ++l) total += (int)e->keys[l].size();
  if (n_out) *n_out = total;
  const bool fits = total <= cap;
  int mono = 0, stereo = total - 1;
  uint8_t d[32];
  for (int l = 0; l < L; ++l) {
    if (e->keys[l].empty()) continue;
    const Image& im = e->pyr[l];
    e->blurred[l].alloc(im.w, im.h);
    gaussian_blur7(im.px.data(), im.w, im.h, im.w, e->blurred[l].px.data(), im.w);
    const float scale = e->scale[l];
    for (const orc_keypoint& k0 : e->keys[l]) {
      brief_descriptor(e->blurred[l].row(cv_round(k0.y)) + cv_round(k0.x), im.w, k0.angle, d);
      orc_keypoint kp = k0;
      if (l != 0) { kp.x *= scale; kp.y *= scale; }
      int slot;
      if (kp.x >= (float)lap0 && kp.x <= (float)lap1) slot = stereo--;
      else slot = mono++;
      if (slot < cap && kps && desc) {
        kps[slot] = kp;
        std::memcpy(desc + (size_t)slot * 32, d, 32);
      }
    }
  }
  return fits ? mono : -2;
}

int orc_level_size(const orc_extractor* e, int level, int* w, int* h) {
  if (level < 0 || level >= (int)e->pyr.size()) return -1;
  *w = e->pyr[level].w; *h = e->pyr[level].h;
  return 0;
}
void orc_level_image(const orc_extractor* e, int level, uint8_t* dst, int ds) {
  const Image& im = e->pyr[level];
  for (int y = 0; y < im.h; ++y) std::memcpy(dst + (size_t)y * ds, im.row(y), im.w);
}
void orc_level_blurred(const orc_extractor* e, int level, uint8_t* dst, int ds) {
  const Image& im = e->blurred[level];
  for (int y = 0; y < im.h; ++y) std::memcpy(dst + (size_t)y * ds, im.row(y), im.w);
}
void orc_level_bordered(const orc_extractor* e, int level, uint8_t* dst, int ds) {
  const Image& im = e->pyr[level];
  for (int y = -kEdge; y < im.h + kEdge; ++y) {
    const uint8_t* S = im.row(reflect101(y, im.h));
    uint8_t* D = dst + (size_t)(y + kEdge) * ds;
    for (int x = -kEdge; x < im.w + kEdge; ++x) D[x + kEdge] = S[reflect101(x, im.w)];
  }
}
static int copy_out(const std::vector<orc_keypoint>& v, orc_keypoint* out, int cap) {
  const int n = (int)v.size();
  if (out) std::memcpy(out, v.data(), sizeof(orc_keypoint) * (size_t)std::min(n, cap));
  return n;
}
int orc_level_candidates(const orc_extractor* e, int level, orc_keypoint* out, int cap) {
  return copy_out(e->cand[level], out, cap);
}
int orc_level_keypoints(const orc_extractor* e, int level, orc_keypoint* out, int cap) {
  return copy_out(e->keys[level], out, cap);
}

void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int ss, uint8_t* dst, int dw, int dh, int ds) {
  resize_linear(src, sw, sh, ss, dst, dw, dh, ds);
}
void orc_gaussian_blur7_u8(const uint8_t* src, int w, int h, int ss, uint8_t* dst, int ds) {
  gaussian_blur7(src, w, h, ss, dst, ds);
}
int orc_fast(const uint8_t* img, int w, int h, int stride, int threshold, int nonmax, orc_keypoint* out,
             int cap) {
  std::vector<orc_keypoint> v;
  fast9_16(img, w, h, stride, threshold, nonmax != 0, v);
  return copy_out(v, out, cap);
}
int orc_fast_corner_score(const uint8_t* img, int stride, int x, int y, int threshold) {
  return corner_score16(img + (size_t)y * stride + x, stride, threshold);
}
float orc_fast_atan2(float y, float x) { return fast_atan2(y, x); }
int orc_cv_round_f(float v) { return cv_round(v); }
float orc_ic_angle(const uint8_t* img, int stride, int x, int y) {
  int umax[16];
  compute_umax(umax);
  return ic_angle(img + (size_t)y * stride + x, stride, umax);
}
void orc_brief(const uint8_t* blurred, int stride, int x, int y, float angle_deg, uint8_t* desc32) {
  brief_descriptor(blurred + (size_t)y * stride + x, stride, angle_deg, desc32);
}
int orc_distribute_octree(const orc_keypoint* cand, int n, int min_x, int max_x, int min_y, int max_y,
                          int n_features, orc_keypoint* out, int cap) {
  std::vector<orc_keypoint> c(cand, cand + n);
  return copy_out(distribute_octree(c, min_x, max_x, min_y, max_y, n_features, n_features), out, cap);
}

}  // extern "C"

// ================================================================= Frame::ComputeStereoMatches
// Restates /root/reference/src/Frame.cc:901-1071 literally (row table, Hamming search with strict '<', 11x11 SAD
// over 11 shifts on the pyramid level of the left keypoint, parabola fit, median-based outlier cut with std::sort).
// mvImagePyramid[level] is a view into a buffer with a 19-px reflect-101 frame, so reads slightly outside the
// level are defined; pyr_at() reproduces that.
namespace {
inline int pyr_at(const Image& im, int x, int y) { return im.row(reflect101(y, im.h))[reflect101(x, im.w)]; }
inline int hamming_words(const uint8_t* a, const uint8_t* b) {
  int d = 0;
  for (int i = 0; i < 32; ++i) d += __builtin_popcount((unsigned)(a[i] ^ b[i]));
  return d;
}
}  // namespace

extern "C" void orc_stereo_matches(const orc_extractor* L, const orc_extractor* R, const orc_keypoint* kpl,
                                   const uint8_t* dl, int N, const orc_keypoint* kpr, const uint8_t* dr, int Nr,
                                   float mb, float mbf, float* uright, float* depth) {
  for (int i = 0; i < N; ++i) { uright[i] = -1.0f; depth[i] = -1.0f; }
  const int TH_HIGH = 100, TH_LOW = 50;
  const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
  const int nRows = L->pyr[0].h;
  std::vector<std::vector<size_t>> rows(nRows);
  for (int iR = 0; iR < Nr; ++iR) {
    const float kpY = kpr[iR].y;
    const float r = 2.0f * L->scale[kpr[iR].octave];
    const int maxr = (int)std::ceil(kpY + r);
    const int minr = (int)std::floor(kpY - r);
    for (int yi = minr; yi <= maxr; ++yi)
      if (yi >= 0 && yi < nRows) rows[yi].push_back(iR);  // (the reference indexes unchecked)
  }
  const float minZ = mb, minD = 0, maxD = mbf / minZ;
  std::vector<std::pair<int, int>> dist_idx;
  for (int iL = 0; iL < N; ++iL) {
    const orc_keypoint& kpL = kpl[iL];
    const int levelL = kpL.octave;
    const float vL = kpL.y, uL = kpL.x;
    const std::vector<size_t>& cand = rows[(size_t)vL];
    if (cand.empty()) continue;
    const float minU = uL - maxD, maxU = uL - minD;
    if (maxU < 0) continue;
    int bestDist = TH_HIGH;
    size_t bestIdxR = 0;
    for (size_t iC = 0; iC < cand.size(); ++iC) {
      const size_t iR = cand[iC];
      const orc_keypoint& kR = kpr[iR];
      if (kR.octave < levelL - 1 || kR.octave > levelL + 1) continue;
      const float uR = kR.x;
      if (uR >= minU && uR <= maxU) {
        const int dist = hamming_words(dl + 32 * (size_t)iL, dr + 32 * iR);
        if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
      }
    }
    if (bestDist < thOrbDist) {
      const float uR0 = kpr[bestIdxR].x;
      const float scaleFactor = L->inv_scale[kpL.octave];
      const float scaleduL = std::round(kpL.x * scaleFactor);
      const float scaledvL = std::round(kpL.y * scaleFactor);
      const float scaleduR0 = std::round(uR0 * scaleFactor);
      const int w = 5, Lw = 5;
      const Image& IL = L->pyr[kpL.octave];
      const Image& IR = R->pyr[kpL.octave];
      int best = INT32_MAX, bestinc = 0;
      float vDists[2 * 5 + 1];
      const float iniu = scaleduR0 + Lw - w;
      const float endu = scaleduR0 + Lw + w + 1;
      if (iniu < 0 || endu >= IR.w) continue;
      const int yl0 = (int)(scaledvL - w), xl0 = (int)(scaleduL - w);
      for (int inc = -Lw; inc <= Lw; ++inc) {
        const int xr0 = (int)(scaleduR0 + inc - w);
        int sad = 0;
        for (int yy = 0; yy < 2 * w + 1; ++yy)
          for (int xx = 0; xx < 2 * w + 1; ++xx) sad += std::abs(pyr_at(IL, xl0 + xx, yl0 + yy) - pyr_at(IR, xr0 + xx, yl0 + yy));
        const float dist = (float)sad;  // cv::norm(IL, IR, NORM_L1)
        if (dist < best) { best = (int)dist; bestinc = inc; }
        vDists[Lw + inc] = dist;
      }
      if (bestinc == -Lw || bestinc == Lw) continue;
      const float dist1 = vDists[Lw + bestinc - 1], dist2 = vDists[Lw + bestinc], dist3 = vDists[Lw + bestinc + 1];
      const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
      if (deltaR < -1 || deltaR > 1) continue;
      float bestuR = L->scale[kpL.octave] * ((float)scaleduR0 + (float)bestinc + deltaR);
      float disparity = (uL - bestuR);
      if (disparity >= minD && disparity < maxD) {
        if (disparity <= 0) {
          disparity = 0.01;
          bestuR = uL - 0.01;
        }
        depth[iL] = mbf / disparity;
        uright[iL] = bestuR;
        dist_idx.push_back(std::pair<int, int>(best, iL));
      }
    }
  }
  if (dist_idx.empty()) return;  // the reference reads vDistIdx[0] of an empty vector here (undefined behaviour)
  std::sort(dist_idx.begin(), dist_idx.end());
  const float median = dist_idx[dist_idx.size() / 2].first;
  const float thDist = 1.5f * 1.4f * median;
  for (int i = (int)dist_idx.size() - 1; i >= 0; --i) {
    if (dist_idx[i].first < thDist) break;
    uright[dist_idx[i].second] = -1;
    depth[dist_idx[i].second] = -1;
  }
}

// ---- ingest --------------------------------------------------------------------------------
// cv::cvtColor(src, dst, COLOR_{RGB,BGR,RGBA,BGRA}2GRAY) for CV_8U, the call of Tracking::GrabImageRGBL
// (/root/reference/src/Tracking.cc:1567-1580).  OpenCV 4.x (color_rgb.simd.hpp, RGB2Gray<uchar>): 15-bit weights
// RY15 = 9798, GY15 = 19235, BY15 = 3735, dst = CV_DESCALE(b*BY + g*GY + r*RY, 15) = (sum + (1 << 14)) >> 15.
// (OpenCV 3.x used 14-bit weights 4899 / 9617 / 1868; the reference requires >= 4.4.)  PARITY UNPINNED: restated.
extern "C" void orc_cvt_gray(const uint8_t* src, int channels, int blue_first, int w, int h, int sstride, uint8_t* dst,
                             int dstride) {
  const int RY = 9798, GY = 19235, BY = 3735;
  const int w0 = blue_first ? BY : RY, w2 = blue_first ? RY : BY;
  for (int y = 0; y < h; ++y) {
    const uint8_t* s = src + (size_t)y * sstride;
    uint8_t* d = dst + (size_t)y * dstride;
    for (int x = 0; x < w; ++x, s += channels) d[x] = (uint8_t)((s[0] * w0 + s[1] * GY + s[2] * w2 + (1 << 14)) >> 15);
  }
}
