// match_oracle.cpp — CPU oracle for the ORBmatcher Hamming paths (TEST INFRASTRUCTURE, see oracle.h).
//
// Restates /root/reference/src/ORBmatcher.cc:35-37 (thresholds), :907-1146 (SearchForTriangulation),
// :2012-2053 (ComputeThreeMaxima), :2058-2074 (DescriptorDistance) and
// /root/reference/src/CameraModels/Pinhole.cpp:107-129 (epipolarConstrain, with F12 hoisted out of the
// candidate loop: it is constant per key-frame pair).  Mono/stereo pinhole key-frames only
// (mpCamera2 == nullptr, NLeft == -1), which is what KITTI RGB-L / Stereo produce.
#include "oracle.h"

#include <cmath>
#include <cstring>
#include <climits>
#include <vector>

namespace {
const int TH_LOW = 50;
const int HISTO_LENGTH = 30;
const int TH_HIGH = 100;  // ORBmatcher.cc:35

inline int hamming256(const uint8_t* a, const uint8_t* b) {
  // the reference's SWAR bit count over 8 little-endian 32-bit words
  int dist = 0;
  for (int i = 0; i < 8; ++i) {
    uint32_t wa, wb;
    std::memcpy(&wa, a + 4 * i, 4);
    std::memcpy(&wb, b + 4 * i, 4);
    uint32_t v = wa ^ wb;
    v = v - ((v >> 1) & 0x55555555u);
    v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
    dist += (int)((((v + (v >> 4)) & 0xF0F0F0Fu) * 0x1010101u) >> 24);
  }
  return dist;
}

void three_maxima(const std::vector<int>* histo, int L, int& i1, int& i2, int& i3) {
  int m1 = 0, m2 = 0, m3 = 0;
  for (int i = 0; i < L; ++i) {
    const int s = (int)histo[i].size();
    if (s > m1) { m3 = m2; m2 = m1; m1 = s; i3 = i2; i2 = i1; i1 = i; }
    else if (s > m2) { m3 = m2; m2 = s; i3 = i2; i2 = i; }
    else if (s > m3) { m3 = s; i3 = i; }
  }
  if (m2 < 0.1f * (float)m1) { i2 = -1; i3 = -1; }
  else if (m3 < 0.1f * (float)m1) { i3 = -1; }
}

inline void mat3_mul(const float* A, const float* B, float* C) {
  // Eigen fixed-size lazy product coefficient: sum of three products, tree order p0 + (p1 + p2)
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      const float p0 = A[3 * r] * B[c], p1 = A[3 * r + 1] * B[3 + c], p2 = A[3 * r + 2] * B[6 + c];
      C[3 * r + c] = p0 + (p1 + p2);
    }
}
inline float cof(const float* m, int i, int j) {
  const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return m[3 * i1 + j1] * m[3 * i2 + j2] - m[3 * i1 + j2] * m[3 * i2 + j1];
}
inline void mat3_inv(const float* m, float* r) {
  // Eigen compute_inverse<…,3>: cofactors of column 0 give the determinant
  const float c0 = cof(m, 0, 0), c1 = cof(m, 1, 0), c2 = cof(m, 2, 0);
  const float det = c0 * m[0] + (c1 * m[3] + c2 * m[6]);
  const float id = 1.0f / det;
  r[0] = c0 * id; r[1] = c1 * id; r[2] = c2 * id;
  r[3] = cof(m, 0, 1) * id; r[4] = cof(m, 1, 1) * id; r[5] = cof(m, 2, 1) * id;
  r[6] = cof(m, 0, 2) * id; r[7] = cof(m, 1, 2) * id; r[8] = cof(m, 2, 2) * id;
}
}  // namespace

extern "C" {

int orc_descriptor_distance(const uint8_t* a, const uint8_t* b) { return hamming256(a, b); }

void orc_hamming_bf(const uint8_t* a, int na, const uint8_t* b, int nb, int* best_idx, int* best_dist,
                    int* second_dist) {
  for (int i = 0; i < na; ++i) {
    int d1 = 256, d2 = 256, idx = -1;
    for (int j = 0; j < nb; ++j) {
      const int d = hamming256(a + 32 * (size_t)i, b + 32 * (size_t)j);
      if (d < d1) { d2 = d1; d1 = d; idx = j; }
      else if (d < d2) { d2 = d; }
    }
    best_idx[i] = idx;
    best_dist[i] = d1;
    if (second_dist) second_dist[i] = d2;
  }
}

void orc_stereo_fisheye_matches(const uint8_t* desc_left, int n_left, int mono_left, const uint8_t* desc_right, int n_right,
                                int mono_right, int* left_to_right, int* best_dist, int* second_dist) {
  // /root/reference/src/Frame.cc:1256-1296 up to the triangulation.  cv::BFMatcher::knnMatch(k = 2) with NORM_HAMMING and
  // no cross check: per query the two smallest distances over the train rows in index order; a row displaces an entry of
  // the pair only when its distance is strictly smaller (cv::batchDistance's K-nearest update), so the lower index wins ties.
  for (int i = 0; i < n_left; ++i) { left_to_right[i] = -1; best_dist[i] = 256; second_dist[i] = 256; }
  const int nt = n_right - mono_right;
  for (int i = mono_left; i < n_left; ++i) {
    int d0 = INT_MAX, d1 = INT_MAX, j0 = -1;
    for (int j = mono_right; j < n_right; ++j) {
      const int d = hamming256(desc_left + 32 * (size_t)i, desc_right + 32 * (size_t)j);
      if (d < d0) { d1 = d0; d0 = d; j0 = j; }
      else if (d < d1) d1 = d;
    }
    if (nt >= 1) best_dist[i] = d0;
    if (nt >= 2) second_dist[i] = d1;
    // Lowe's ratio: (*it).size() >= 2 && (*it)[0].distance < (*it)[1].distance * 0.7   (DMatch::distance is a float)
    if (nt >= 2 && (double)(float)d0 < (double)(float)d1 * 0.7) left_to_right[i] = j0;
  }
}

void orc_fundamental(const float K1[4], const float K2[4], const float R12[9], const float t12[3],
                     float F12[9]) {
  // Pinhole.cpp:109-112: F12 = K1^T^-1 * [t12]x * R12 * K2^-1, fp32, products left to right
  const float k1t[9] = {K1[0], 0.f, 0.f, 0.f, K1[1], 0.f, K1[2], K1[3], 1.f};
  const float k2[9] = {K2[0], 0.f, K2[2], 0.f, K2[1], K2[3], 0.f, 0.f, 1.f};
  const float tx[9] = {0.f, -t12[2], t12[1], t12[2], 0.f, -t12[0], -t12[1], t12[0], 0.f};
  float a[9], b[9], c[9], k2i[9];
  mat3_inv(k1t, a);
  mat3_mul(a, tx, b);
  mat3_mul(b, R12, c);
  mat3_inv(k2, k2i);
  mat3_mul(c, k2i, F12);
}

int orc_search_triangulation(const orc_tri_input* in, int* matches12) {
  int nmatches = 0;
  for (int i = 0; i < in->n1; ++i) matches12[i] = -1;
  std::vector<int> rot_hist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;
  const float* F = in->F12;

  int a = 0, b = 0;
  while (a < in->nnodes1 && b < in->nnodes2) {
    if (in->node_id1[a] < in->node_id2[b]) { ++a; continue; }  // lower_bound on a sorted map
    if (in->node_id1[a] > in->node_id2[b]) { ++b; continue; }
    for (int p = in->node_off1[a]; p < in->node_off1[a + 1]; ++p) {
      const int idx1 = in->node_feat1[p];
      if (in->has_mp1[idx1]) continue;
      const bool stereo1 = in->uright1[idx1] >= 0;
      if (in->only_stereo && !stereo1) continue;
      const float x1 = in->kp1_xy[2 * idx1], y1 = in->kp1_xy[2 * idx1 + 1];
      const uint8_t* d1 = in->desc1 + 32 * (size_t)idx1;
      int best_dist = TH_LOW, best_idx2 = -1;
      for (int q = in->node_off2[b]; q < in->node_off2[b + 1]; ++q) {
        const int idx2 = in->node_feat2[q];
        if (in->has_mp2[idx2]) continue;  // vbMatched2 is never set by the reference
        const bool stereo2 = in->uright2[idx2] >= 0;
        if (in->only_stereo && !stereo2) continue;
        const int dist = hamming256(d1, in->desc2 + 32 * (size_t)idx2);
        if (dist > TH_LOW || dist > best_dist) continue;
        const float x2 = in->kp2_xy[2 * idx2], y2 = in->kp2_xy[2 * idx2 + 1];
        const int oct2 = in->kp2_octave[idx2];
        if (!stereo1 && !stereo2) {
          const float ex = in->ep[0] - x2, ey = in->ep[1] - y2;
          if (ex * ex + ey * ey < 100 * in->scale_factors2[oct2]) continue;
        }
        bool ok = in->coarse != 0;
        if (!ok) {
          // Pinhole::epipolarConstrain, all fp32 except the final compare (3.84 is a double literal)
          const float la = x1 * F[0] + y1 * F[3] + F[6];
          const float lb = x1 * F[1] + y1 * F[4] + F[7];
          const float lc = x1 * F[2] + y1 * F[5] + F[8];
          const float num = la * x2 + lb * y2 + lc;
          const float den = la * la + lb * lb;
          if (den != 0) {
            const float dsqr = num * num / den;
            ok = (double)dsqr < 3.84 * (double)in->level_sigma2_2[oct2];
          }
        }
        if (ok) { best_idx2 = idx2; best_dist = dist; }
      }
      if (best_idx2 >= 0) {
        matches12[idx1] = best_idx2;
        ++nmatches;
        if (in->check_orientation) {
          float rot = in->kp1_angle[idx1] - in->kp2_angle[best_idx2];
          if (rot < 0.0) rot += 360.0f;
          int bin = (int)std::round(rot * factor);
          if (bin == HISTO_LENGTH) bin = 0;
          rot_hist[bin].push_back(idx1);
        }
      }
    }
    ++a;
    ++b;
  }

  if (in->check_orientation) {
    int i1 = -1, i2 = -1, i3 = -1;
    three_maxima(rot_hist, HISTO_LENGTH, i1, i2, i3);
    for (int i = 0; i < HISTO_LENGTH; ++i) {
      if (i == i1 || i == i2 || i == i3) continue;
      for (int idx1 : rot_hist[i]) { matches12[idx1] = -1; --nmatches; }
    }
  }
  return nmatches;
}

}  // extern "C"

// std::sort of (key, value) pairs with a comparator that looks at the key only — the literal libstdc++
// behaviour the quad-tree relies on for nodes with equal (size, UL.x).  Used to pin the product's
// restatement of the introsort (tests/test_introsort.py).
#include <algorithm>
#include <utility>
extern "C" void orc_std_sort_pairs(uint64_t* key, uint32_t* val, int n) {
  std::vector<std::pair<uint64_t, uint32_t>> v(n);
  for (int i = 0; i < n; ++i) v[i] = std::make_pair(key[i], val[i]);
  std::sort(v.begin(), v.end(),
            [](const std::pair<uint64_t, uint32_t>& a, const std::pair<uint64_t, uint32_t>& b) { return a.first < b.first; });
  for (int i = 0; i < n; ++i) { key[i] = v[i].first; val[i] = v[i].second; }
}

// ---- ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono), /root/reference/src/ORBmatcher.cc:1676-1887 ----
// (single camera, Nleft == -1) with Frame::AssignFeaturesToGrid / PosInGrid / GetFeaturesInArea (src/Frame.cc:475-506,
// 815-825, 747-813) and the Eigen / Sophus arithmetic spelled out: SE3f * p = q._transformVector(p) + t,
// inverse().translation() = q^-1 * (-t).  fp32, no contraction.
namespace {
struct V3 { float x, y, z; };
inline V3 cross(const V3& a, const V3& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline V3 quat_rotate(const float q[4], const V3& p) {  // q = (x, y, z, w); Eigen QuaternionBase::_transformVector
  const V3 u = {q[0], q[1], q[2]};
  V3 uv = cross(u, p);
  uv = {uv.x + uv.x, uv.y + uv.y, uv.z + uv.z};
  const V3 c = cross(u, uv);
  return {p.x + q[3] * uv.x + c.x, p.y + q[3] * uv.y + c.y, p.z + q[3] * uv.z + c.z};
}
inline V3 se3_apply(const float q[4], const float t[3], const V3& p) {
  const V3 r = quat_rotate(q, p);
  return {r.x + t[0], r.y + t[1], r.z + t[2]};
}
}  // namespace

extern "C" int orc_search_by_projection(const orc_projection_input* in, int* match2) {
  const int COLS = 64, ROWS = 48;  // FRAME_GRID_COLS / FRAME_GRID_ROWS (Frame.h:46-47)
  const float mnMinX = in->grid[0], mnMinY = in->grid[1], mnMaxX = in->grid[2], mnMaxY = in->grid[3];
  const float invW = in->grid[4], invH = in->grid[5];
  // AssignFeaturesToGrid
  std::vector<std::vector<int>> cells((size_t)COLS * ROWS);
  for (int i = 0; i < in->n2; ++i) {
    const int px = (int)roundf((in->kp2_xy[2 * i] - mnMinX) * invW), py = (int)roundf((in->kp2_xy[2 * i + 1] - mnMinY) * invH);
    if (px < 0 || px >= COLS || py < 0 || py >= ROWS) continue;
    cells[(size_t)px * ROWS + py].push_back(i);
  }
  // twc = Tcw.inverse().translation(); tlc = Tlw * twc
  const float qinv[4] = {-in->Tcw_q[0], -in->Tcw_q[1], -in->Tcw_q[2], in->Tcw_q[3]};
  const V3 twc = quat_rotate(qinv, V3{-in->Tcw_t[0], -in->Tcw_t[1], -in->Tcw_t[2]});
  const V3 tlc = se3_apply(in->Tlw_q, in->Tlw_t, twc);
  const bool bForward = tlc.z > in->mb && !in->mono;
  const bool bBackward = -tlc.z > in->mb && !in->mono;

  std::vector<int> holder(in->n2, -1);  // CurrentFrame.mvpMapPoints as "index of the LastFrame feature" (all NULL on entry)
  std::vector<int> rot_hist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;
  int nmatches = 0;
  for (int i = 0; i < in->n1; ++i) {
    if (!in->valid1[i]) continue;
    const V3 xw = {in->world_pos1[3 * i], in->world_pos1[3 * i + 1], in->world_pos1[3 * i + 2]};
    const V3 xc = se3_apply(in->Tcw_q, in->Tcw_t, xw);
    const float invzc = (float)(1.0 / (double)xc.z);
    if (invzc < 0) continue;
    const float u = in->K[0] * xc.x / xc.z + in->K[2], v = in->K[1] * xc.y / xc.z + in->K[3];  // Pinhole::project
    if (u < mnMinX || u > mnMaxX) continue;
    if (v < mnMinY || v > mnMaxY) continue;
    const int oct = in->octave1[i];
    const float radius = in->th * in->scale_factors[oct];
    int minLevel, maxLevel;
    if (bForward) { minLevel = oct; maxLevel = -1; }
    else if (bBackward) { minLevel = 0; maxLevel = oct; }
    else { minLevel = oct - 1; maxLevel = oct + 1; }
    // GetFeaturesInArea
    if (!(u == u) || !(v == v)) continue;  // NaN: the cell range is empty on every platform we know of
    const int nMinCellX = std::max(0, (int)floorf((u - mnMinX - radius) * invW));
    if (nMinCellX >= COLS) continue;
    const int nMaxCellX = std::min(COLS - 1, (int)ceilf((u - mnMinX + radius) * invW));
    if (nMaxCellX < 0) continue;
    const int nMinCellY = std::max(0, (int)floorf((v - mnMinY - radius) * invH));
    if (nMinCellY >= ROWS) continue;
    const int nMaxCellY = std::min(ROWS - 1, (int)ceilf((v - mnMinY + radius) * invH));
    if (nMaxCellY < 0) continue;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    int bestDist = 256, bestIdx2 = -1;
    const uint8_t* dMP = in->mp_desc1 + 32 * (size_t)i;
    for (int ix = nMinCellX; ix <= nMaxCellX; ++ix)
      for (int iy = nMinCellY; iy <= nMaxCellY; ++iy)
        for (int i2 : cells[(size_t)ix * ROWS + iy]) {
          if (bCheckLevels) {
            if (in->kp2_octave[i2] < minLevel) continue;
            if (maxLevel >= 0 && in->kp2_octave[i2] > maxLevel) continue;
          }
          const float distx = in->kp2_xy[2 * i2] - u, disty = in->kp2_xy[2 * i2 + 1] - v;
          if (!(fabsf(distx) < radius && fabsf(disty) < radius)) continue;
          // a feature holding a map point with observations is taken (ORBmatcher.cc:1745-1747)
          if (holder[i2] >= 0 && in->mp_observed1[holder[i2]]) continue;
          if (in->uright2[i2] > 0) {
            const float ur = u - in->mbf * invzc;
            const float er = fabsf(ur - in->uright2[i2]);
            if (er > radius) continue;
          }
          const int dist = hamming256(dMP, in->desc2 + 32 * (size_t)i2);
          if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
    if (bestDist <= TH_HIGH) {
      holder[bestIdx2] = i;
      ++nmatches;
      if (in->check_orientation) {
        float rot = in->angle1[i] - in->kp2_angle[bestIdx2];
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)roundf(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        rot_hist[bin].push_back(bestIdx2);
      }
    }
  }
  if (in->check_orientation) {
    int i1 = -1, i2 = -1, i3 = -1;
    three_maxima(rot_hist, HISTO_LENGTH, i1, i2, i3);
    for (int i = 0; i < HISTO_LENGTH; ++i) {
      if (i == i1 || i == i2 || i == i3) continue;
      for (int idx2 : rot_hist[i]) { holder[idx2] = -1; --nmatches; }  // a feature listed twice is counted twice
    }
  }
  for (int i = 0; i < in->n2; ++i) match2[i] = holder[i];
  return nmatches;
}

// ---- per-point search of Fuse(pKF, Scw, ...) (ORBmatcher.cc:1373-1436) and SearchBySim3 (:1505-1566, :1583-1649) ----
extern "C" void orc_project_search(const orc_project_search_input* in, int* best_idx, int* best_dist) {
  const int COLS = 64, ROWS = 48;
  const float mnMinX = in->grid[0], mnMinY = in->grid[1], mnMaxX = in->grid[2], mnMaxY = in->grid[3];
  const float invW = in->grid[4], invH = in->grid[5];
  const float fx = in->K[0], fy = in->K[1], cx = in->K[2], cy = in->K[3];
  std::vector<std::vector<int>> cells((size_t)COLS * ROWS);
  for (int i = 0; i < in->n2; ++i) {
    const int px = (int)roundf((in->kp2_xy[2 * i] - mnMinX) * invW), py = (int)roundf((in->kp2_xy[2 * i + 1] - mnMinY) * invH);
    if (px < 0 || px >= COLS || py < 0 || py >= ROWS) continue;
    cells[(size_t)px * ROWS + py].push_back(i);
  }
  for (int i = 0; i < in->n1; ++i) {
    best_idx[i] = -1;
    best_dist[i] = 256;
    if (!in->valid1[i]) continue;
    const float X = in->cam_pos1[3 * i], Y = in->cam_pos1[3 * i + 1], Z = in->cam_pos1[3 * i + 2];
    if (Z < 0.0f) continue;  // depth must be positive
    float u, v;
    if (in->proj_form == 1) {
      const float invz = 1.0 / Z;
      const float x = X * invz, y = Y * invz;
      u = fx * x + cx; v = fy * y + cy;
    } else {
      u = fx * X / Z + cx; v = fy * Y / Z + cy;
    }
    if (!(u >= mnMinX && u < mnMaxX && v >= mnMinY && v < mnMaxY)) continue;
    const int nPredictedLevel = in->level1[i];
    const float radius = in->th * in->scale_factors[nPredictedLevel];
    const int nMinCellX = std::max(0, (int)floorf((u - mnMinX - radius) * invW));
    if (nMinCellX >= COLS) continue;
    const int nMaxCellX = std::min(COLS - 1, (int)ceilf((u - mnMinX + radius) * invW));
    if (nMaxCellX < 0) continue;
    const int nMinCellY = std::max(0, (int)floorf((v - mnMinY - radius) * invH));
    if (nMinCellY >= ROWS) continue;
    const int nMaxCellY = std::min(ROWS - 1, (int)ceilf((v - mnMinY + radius) * invH));
    if (nMaxCellY < 0) continue;
    const uint8_t* dMP = in->mp_desc1 + 32 * (size_t)i;
    int bestDist = INT_MAX, bestIdx = -1;
    for (int ix = nMinCellX; ix <= nMaxCellX; ++ix)
      for (int iy = nMinCellY; iy <= nMaxCellY; ++iy)
        for (int idx : cells[(size_t)ix * ROWS + iy]) {
          if (!(fabsf(in->kp2_xy[2 * idx] - u) < radius && fabsf(in->kp2_xy[2 * idx + 1] - v) < radius)) continue;
          const int kpLevel = in->kp2_octave[idx];
          if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
          const int dist = hamming256(dMP, in->desc2 + 32 * (size_t)idx);
          if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
    if (bestIdx >= 0) best_dist[i] = bestDist;
    if (bestDist <= in->max_dist) best_idx[i] = bestIdx;
  }
}

// ---- SearchByProjection(pKF, Scw, vpPoints, [vpPointsKFs,] vpMatched, [vpMatchedKF,] th, ratioHamming), ORBmatcher.cc:427-646 ----
extern "C" int orc_search_by_projection_sim3(const orc_project_search_input* in, const uint8_t* matched2, int* match2) {
  const int COLS = 64, ROWS = 48;
  const float mnMinX = in->grid[0], mnMinY = in->grid[1], mnMaxX = in->grid[2], mnMaxY = in->grid[3];
  const float invW = in->grid[4], invH = in->grid[5];
  const float fx = in->K[0], fy = in->K[1], cx = in->K[2], cy = in->K[3];
  std::vector<std::vector<int>> cells((size_t)COLS * ROWS);
  for (int i = 0; i < in->n2; ++i) {
    const int px = (int)roundf((in->kp2_xy[2 * i] - mnMinX) * invW), py = (int)roundf((in->kp2_xy[2 * i + 1] - mnMinY) * invH);
    if (px < 0 || px >= COLS || py < 0 || py >= ROWS) continue;
    cells[(size_t)px * ROWS + py].push_back(i);
  }
  std::vector<int> vpMatched(in->n2, -1);  // -2: matched before the call
  for (int i = 0; i < in->n2; ++i)
    if (matched2 && matched2[i]) vpMatched[i] = -2;
  int nmatches = 0;
  for (int iMP = 0; iMP < in->n1; ++iMP) {
    if (!in->valid1[iMP]) continue;
    const float X = in->cam_pos1[3 * iMP], Y = in->cam_pos1[3 * iMP + 1], Z = in->cam_pos1[3 * iMP + 2];
    if (Z < 0.0) continue;
    float u, v;
    if (in->proj_form == 2) {
      const float invz = 1 / Z;
      const float x = X * invz, y = Y * invz;
      u = fx * x + cx; v = fy * y + cy;
    } else {
      u = fx * X / Z + cx; v = fy * Y / Z + cy;
    }
    if (!(u >= mnMinX && u < mnMaxX && v >= mnMinY && v < mnMaxY)) continue;
    const int nPredictedLevel = in->level1[iMP];
    const float radius = in->th * in->scale_factors[nPredictedLevel];
    const int nMinCellX = std::max(0, (int)floorf((u - mnMinX - radius) * invW));
    if (nMinCellX >= COLS) continue;
    const int nMaxCellX = std::min(COLS - 1, (int)ceilf((u - mnMinX + radius) * invW));
    if (nMaxCellX < 0) continue;
    const int nMinCellY = std::max(0, (int)floorf((v - mnMinY - radius) * invH));
    if (nMinCellY >= ROWS) continue;
    const int nMaxCellY = std::min(ROWS - 1, (int)ceilf((v - mnMinY + radius) * invH));
    if (nMaxCellY < 0) continue;
    const uint8_t* dMP = in->mp_desc1 + 32 * (size_t)iMP;
    int bestDist = 256, bestIdx = -1;
    for (int ix = nMinCellX; ix <= nMaxCellX; ++ix)
      for (int iy = nMinCellY; iy <= nMaxCellY; ++iy)
        for (int idx : cells[(size_t)ix * ROWS + iy]) {
          if (!(fabsf(in->kp2_xy[2 * idx] - u) < radius && fabsf(in->kp2_xy[2 * idx + 1] - v) < radius)) continue;
          if (vpMatched[idx] != -1) continue;
          const int kpLevel = in->kp2_octave[idx];
          if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
          const int dist = hamming256(dMP, in->desc2 + 32 * (size_t)idx);
          if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
    if (bestDist <= in->max_dist) {  // <= TH_LOW * ratioHamming
      vpMatched[bestIdx] = iMP;
      ++nmatches;
    }
  }
  for (int i = 0; i < in->n2; ++i) match2[i] = vpMatched[i] >= 0 ? vpMatched[i] : -1;
  return nmatches;
}

// ---- MapPoint::ComputeDistinctiveDescriptors, MapPoint.cc:329-403 ----
extern "C" void orc_distinctive_descriptors(const uint8_t* desc, const int32_t* off, int n_points, int32_t* best) {
  for (int p = 0; p < n_points; ++p) {
    const size_t N = (size_t)(off[p + 1] - off[p]);
    if (N == 0) { best[p] = -1; continue; }
    const uint8_t* D = desc + 32 * (size_t)off[p];
    std::vector<float> Distances(N * N);
    for (size_t i = 0; i < N; i++) {
      Distances[i * N + i] = 0;
      for (size_t j = i + 1; j < N; j++) {
        const int distij = hamming256(D + 32 * i, D + 32 * j);
        Distances[i * N + j] = distij;
        Distances[j * N + i] = distij;
      }
    }
    int BestMedian = INT_MAX, BestIdx = 0;
    for (size_t i = 0; i < N; i++) {
      std::vector<int> vDists(Distances.begin() + i * N, Distances.begin() + (i + 1) * N);
      std::sort(vDists.begin(), vDists.end());
      const int median = vDists[0.5 * (N - 1)];
      if (median < BestMedian) { BestMedian = median; BestIdx = (int)i; }
    }
    best[p] = BestIdx;
  }
}

// ---- Frame::UndistortKeyPoints, Frame.cc:837-870 = cv::undistortPoints(src, dst, K, D, noArray(), P = K) ----
// OpenCV 4.x cvUndistortPointsInternal: double arithmetic, k[0..4] = k1 k2 p1 p2 k3, k[5..13] = 0, tilt model off (identity),
// R = identity so RR = P * R = K, criteria = 5 iterations (no epsilon test).
extern "C" void orc_undistort_points(const float* xy, int n, const float K[4], const float* dist, int n_dist, float* out_xy) {
  double k[14] = {0};
  for (int i = 0; i < n_dist && i < 5; ++i) k[i] = (double)dist[i];
  const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
  const double ifx = 1. / fx, ify = 1. / fy;
  const double RR[3][3] = {{fx, 0, cx}, {0, fy, cy}, {0, 0, 1}};
  for (int i = 0; i < n; ++i) {
    double x = xy[2 * i], y = xy[2 * i + 1], x0, y0;
    const double u = x, v = y;
    x = (x - cx) * ifx;
    y = (y - cy) * ify;
    x0 = x; y0 = y;  // invMatTilt = identity: vecUntilt = (x, y, 1), invProj = 1
    for (int j = 0;; j++) {
      if (j >= 5) break;
      const double r2 = x * x + y * y;
      const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
      if (icdist < 0) { x = (u - cx) * ifx; y = (v - cy) * ify; break; }
      const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
      const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
      x = (x0 - deltaX) * icdist;
      y = (y0 - deltaY) * icdist;
    }
    const double xx = RR[0][0] * x + RR[0][1] * y + RR[0][2];
    const double yy = RR[1][0] * x + RR[1][1] * y + RR[1][2];
    const double ww = 1. / (RR[2][0] * x + RR[2][1] * y + RR[2][2]);
    out_xy[2 * i] = (float)(xx * ww);
    out_xy[2 * i + 1] = (float)(yy * ww);
  }
}

// ---- ORBmatcher::SearchByProjection(Frame&, KeyFrame*, const set<MapPoint*>&, th, ORBdist), ORBmatcher.cc:1889-2010 ----
namespace {
// the part of the loop body that needs the MapPoint object (:1908-1937 without the projection): -1 = skipped, else PredictScale
int kf_point_level(const orc_kf_projection_input* in, int i, const V3& Ow) {
  if (!in->has_mp1[i] || in->bad1[i] || in->found1[i]) return -1;
  const V3 PO = {in->world_pos1[3 * i] - Ow.x, in->world_pos1[3 * i + 1] - Ow.y, in->world_pos1[3 * i + 2] - Ow.z};
  // Eigen's norm(): sqrt of the unrolled 3-term sum x*x + (y*y + z*z)
  const float dist3D = sqrtf(PO.x * PO.x + (PO.y * PO.y + PO.z * PO.z));
  const float maxDistance = 1.2f * in->max_dist1[i], minDistance = 0.8f * in->min_dist1[i];
  if (dist3D < minDistance || dist3D > maxDistance) return -1;
  // MapPoint::PredictScale(dist3D, &CurrentFrame)
  const float ratio = in->max_dist1[i] / dist3D;
  int nScale = (int)ceil(logf(ratio) / in->log_scale_factor);  // ceil(float) promotes to double: exact
  if (nScale < 0) nScale = 0;
  else if (nScale >= in->n_levels) nScale = in->n_levels - 1;
  return nScale;
}
V3 camera_centre(const orc_kf_projection_input* in) {  // Tcw.inverse().translation()
  const float qinv[4] = {-in->Tcw_q[0], -in->Tcw_q[1], -in->Tcw_q[2], in->Tcw_q[3]};
  return quat_rotate(qinv, V3{-in->Tcw_t[0], -in->Tcw_t[1], -in->Tcw_t[2]});
}
}  // namespace

extern "C" void orc_kf_projection_prepass(const orc_kf_projection_input* in, uint8_t* valid1, int32_t* level1) {
  const V3 Ow = camera_centre(in);
  for (int i = 0; i < in->n1; ++i) {
    const int l = kf_point_level(in, i, Ow);
    valid1[i] = l >= 0;
    level1[i] = l >= 0 ? l : 0;
  }
}

extern "C" int orc_search_by_projection_kf(const orc_kf_projection_input* in, int* match2) {
  const int COLS = 64, ROWS = 48;
  const float mnMinX = in->grid[0], mnMinY = in->grid[1], mnMaxX = in->grid[2], mnMaxY = in->grid[3];
  const float invW = in->grid[4], invH = in->grid[5];
  std::vector<std::vector<int>> cells((size_t)COLS * ROWS);
  for (int i = 0; i < in->n2; ++i) {
    const int px = (int)roundf((in->kp2_xy[2 * i] - mnMinX) * invW), py = (int)roundf((in->kp2_xy[2 * i + 1] - mnMinY) * invH);
    if (px < 0 || px >= COLS || py < 0 || py >= ROWS) continue;
    cells[(size_t)px * ROWS + py].push_back(i);
  }
  const V3 Ow = camera_centre(in);
  // CurrentFrame.mvpMapPoints: -2 = a map point from before the call, >= 0 = index of the key-frame feature, -1 = NULL
  std::vector<int> holder(in->n2, -1);
  for (int i = 0; i < in->n2; ++i)
    if (in->occupied2 && in->occupied2[i]) holder[i] = -2;
  std::vector<int> rot_hist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;
  int nmatches = 0;
  for (int i = 0; i < in->n1; ++i) {
    if (!in->has_mp1[i] || in->bad1[i] || in->found1[i]) continue;
    const V3 xw = {in->world_pos1[3 * i], in->world_pos1[3 * i + 1], in->world_pos1[3 * i + 2]};
    const V3 xc = se3_apply(in->Tcw_q, in->Tcw_t, xw);
    const float u = in->K[0] * xc.x / xc.z + in->K[2], v = in->K[1] * xc.y / xc.z + in->K[3];  // Pinhole::project
    if (u < mnMinX || u > mnMaxX) continue;
    if (v < mnMinY || v > mnMaxY) continue;
    const int nPredictedLevel = kf_point_level(in, i, Ow);
    if (nPredictedLevel < 0) continue;
    const float radius = in->th * in->scale_factors[nPredictedLevel];
    const int minLevel = nPredictedLevel - 1, maxLevel = nPredictedLevel + 1;
    if (!(u == u) || !(v == v)) continue;
    const int nMinCellX = std::max(0, (int)floorf((u - mnMinX - radius) * invW));
    if (nMinCellX >= COLS) continue;
    const int nMaxCellX = std::min(COLS - 1, (int)ceilf((u - mnMinX + radius) * invW));
    if (nMaxCellX < 0) continue;
    const int nMinCellY = std::max(0, (int)floorf((v - mnMinY - radius) * invH));
    if (nMinCellY >= ROWS) continue;
    const int nMaxCellY = std::min(ROWS - 1, (int)ceilf((v - mnMinY + radius) * invH));
    if (nMaxCellY < 0) continue;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    int bestDist = 256, bestIdx2 = -1;
    const uint8_t* dMP = in->mp_desc1 + 32 * (size_t)i;
    for (int ix = nMinCellX; ix <= nMaxCellX; ++ix)
      for (int iy = nMinCellY; iy <= nMaxCellY; ++iy)
        for (int i2 : cells[(size_t)ix * ROWS + iy]) {
          if (bCheckLevels) {
            if (in->kp2_octave[i2] < minLevel) continue;
            if (maxLevel >= 0 && in->kp2_octave[i2] > maxLevel) continue;
          }
          const float distx = in->kp2_xy[2 * i2] - u, disty = in->kp2_xy[2 * i2 + 1] - v;
          if (!(fabsf(distx) < radius && fabsf(disty) < radius)) continue;
          if (holder[i2] != -1) continue;  // CurrentFrame.mvpMapPoints[i2] (:1949-1950)
          const int dist = hamming256(dMP, in->desc2 + 32 * (size_t)i2);
          if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
    if (bestDist <= in->orb_dist) {
      holder[bestIdx2] = i;
      ++nmatches;
      if (in->check_orientation) {
        float rot = in->angle1[i] - in->kp2_angle[bestIdx2];
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)roundf(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        rot_hist[bin].push_back(bestIdx2);
      }
    }
  }
  if (in->check_orientation) {
    int i1 = -1, i2 = -1, i3 = -1;
    three_maxima(rot_hist, HISTO_LENGTH, i1, i2, i3);
    for (int i = 0; i < HISTO_LENGTH; ++i) {
      if (i == i1 || i == i2 || i == i3) continue;
      for (int idx2 : rot_hist[i]) { holder[idx2] = -1; --nmatches; }
    }
  }
  for (int i = 0; i < in->n2; ++i) match2[i] = holder[i] >= 0 ? holder[i] : -1;
  return nmatches;
}

// ---- the per-point search of ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th, bRight = false), ORBmatcher.cc:1148-1338 ----
namespace {
// the tests of the loop body that need the MapPoint object and do not depend on the projection: -1 = skipped, else PredictScale
int fuse_point_level(const orc_fuse_input* in, int i) {
  if (!in->has_mp1[i] || in->bad1[i] || in->in_kf1[i]) return -1;
  const V3 PO = {in->world_pos1[3 * i] - in->Ow[0], in->world_pos1[3 * i + 1] - in->Ow[1], in->world_pos1[3 * i + 2] - in->Ow[2]};
  const float dist3D = sqrtf(PO.x * PO.x + (PO.y * PO.y + PO.z * PO.z));  // Eigen norm(): unrolled 3-term sum
  const float maxDistance = 1.2f * in->max_dist1[i], minDistance = 0.8f * in->min_dist1[i];
  if (dist3D < minDistance || dist3D > maxDistance) return -1;
  // viewing angle below 60 degrees: PO.dot(Pn) < 0.5 * dist3D (float dot, double comparison)
  const float dot = PO.x * in->normal1[3 * i] + (PO.y * in->normal1[3 * i + 1] + PO.z * in->normal1[3 * i + 2]);
  if ((double)dot < 0.5 * (double)dist3D) return -1;
  const float ratio = in->max_dist1[i] / dist3D;
  int nScale = (int)ceil(logf(ratio) / in->log_scale_factor);
  if (nScale < 0) nScale = 0;
  else if (nScale >= in->n_levels) nScale = in->n_levels - 1;
  return nScale;
}
}  // namespace

extern "C" void orc_fuse_prepass(const orc_fuse_input* in, uint8_t* valid1, int32_t* level1) {
  for (int i = 0; i < in->n1; ++i) {
    const int l = fuse_point_level(in, i);
    valid1[i] = l >= 0;
    level1[i] = l >= 0 ? l : 0;
  }
}

extern "C" int orc_fuse_search(const orc_fuse_input* in, int* best_idx) {
  const int COLS = 64, ROWS = 48;
  const float mnMinX = in->grid[0], mnMinY = in->grid[1], mnMaxX = in->grid[2], mnMaxY = in->grid[3];
  const float invW = in->grid[4], invH = in->grid[5];
  std::vector<std::vector<int>> cells((size_t)COLS * ROWS);
  for (int i = 0; i < in->n2; ++i) {
    const int px = (int)roundf((in->kp2_xy[2 * i] - mnMinX) * invW), py = (int)roundf((in->kp2_xy[2 * i + 1] - mnMinY) * invH);
    if (px < 0 || px >= COLS || py < 0 || py >= ROWS) continue;
    cells[(size_t)px * ROWS + py].push_back(i);
  }
  int nFused = 0;
  for (int i = 0; i < in->n1; ++i) {
    best_idx[i] = -1;
    if (!in->has_mp1[i] || in->bad1[i] || in->in_kf1[i]) continue;
    const V3 p3Dw = {in->world_pos1[3 * i], in->world_pos1[3 * i + 1], in->world_pos1[3 * i + 2]};
    const V3 p3Dc = se3_apply(in->Tcw_q, in->Tcw_t, p3Dw);
    if (p3Dc.z < 0.0f) continue;
    const float invz = 1 / p3Dc.z;
    const float u = in->K[0] * p3Dc.x / p3Dc.z + in->K[2], v = in->K[1] * p3Dc.y / p3Dc.z + in->K[3];
    if (!(u >= mnMinX && u < mnMaxX && v >= mnMinY && v < mnMaxY)) continue;  // KeyFrame::IsInImage
    const float ur = u - in->bf * invz;
    const int nPredictedLevel = fuse_point_level(in, i);
    if (nPredictedLevel < 0) continue;
    const float radius = in->th * in->scale_factors[nPredictedLevel];
    const int nMinCellX = std::max(0, (int)floorf((u - mnMinX - radius) * invW));
    if (nMinCellX >= COLS) continue;
    const int nMaxCellX = std::min(COLS - 1, (int)ceilf((u - mnMinX + radius) * invW));
    if (nMaxCellX < 0) continue;
    const int nMinCellY = std::max(0, (int)floorf((v - mnMinY - radius) * invH));
    if (nMinCellY >= ROWS) continue;
    const int nMaxCellY = std::min(ROWS - 1, (int)ceilf((v - mnMinY + radius) * invH));
    if (nMaxCellY < 0) continue;
    const uint8_t* dMP = in->mp_desc1 + 32 * (size_t)i;
    int bestDist = 256, bestIdx = -1;
    for (int ix = nMinCellX; ix <= nMaxCellX; ++ix)
      for (int iy = nMinCellY; iy <= nMaxCellY; ++iy)
        for (int idx : cells[(size_t)ix * ROWS + iy]) {
          const float kpx = in->kp2_xy[2 * idx], kpy = in->kp2_xy[2 * idx + 1];
          if (!(fabsf(kpx - u) < radius && fabsf(kpy - v) < radius)) continue;
          const int kpLevel = in->kp2_octave[idx];
          if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
          if (in->uright2[idx] >= 0) {
            const float kpr = in->uright2[idx];
            const float ex = u - kpx, ey = v - kpy, er = ur - kpr;
            const float e2 = ex * ex + ey * ey + er * er;
            if (e2 * in->inv_level_sigma2[kpLevel] > 7.8) continue;
          } else {
            const float ex = u - kpx, ey = v - kpy;
            const float e2 = ex * ex + ey * ey;
            if (e2 * in->inv_level_sigma2[kpLevel] > 5.99) continue;
          }
          const int dist = hamming256(dMP, in->desc2 + 32 * (size_t)idx);
          if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
    if (bestDist <= TH_LOW) { best_idx[i] = bestIdx; ++nFused; }
  }
  return nFused;
}

// ---- ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th, bFarPoints, thFarPoints), ORBmatcher.cc:43-213 ----
// (single camera) with RadiusByViewingCos (:215-221) and Frame::GetFeaturesInArea (src/Frame.cc:747-813)
extern "C" int orc_search_local_points(const orc_local_points_input* in, int* match2) {
  const int COLS = 64, ROWS = 48;
  const float mnMinX = in->grid[0], mnMinY = in->grid[1], invW = in->grid[4], invH = in->grid[5];
  std::vector<std::vector<int>> cells((size_t)COLS * ROWS);
  for (int i = 0; i < in->n2; ++i) {
    const int px = (int)roundf((in->kp2_xy[2 * i] - mnMinX) * invW), py = (int)roundf((in->kp2_xy[2 * i + 1] - mnMinY) * invH);
    if (px < 0 || px >= COLS || py < 0 || py >= ROWS) continue;
    cells[(size_t)px * ROWS + py].push_back(i);
  }
  std::vector<int> holder(in->n2, -1);  // map point assigned by THIS call
  const bool bFactor = in->th != 1.0;
  int nmatches = 0;
  for (int i = 0; i < in->n1; ++i) {
    if (!in->valid1[i]) continue;
    const int nPredictedLevel = in->level1[i];
    float r = in->view_cos1[i] > 0.998 ? 2.5f : 4.0f;  // RadiusByViewingCos
    if (bFactor) r *= in->th;
    const float x = in->proj1[3 * i], y = in->proj1[3 * i + 1], xr = in->proj1[3 * i + 2];
    const float radius = r * in->scale_factors[nPredictedLevel];
    const int minLevel = nPredictedLevel - 1, maxLevel = nPredictedLevel;
    if (!(x == x) || !(y == y)) continue;
    const int nMinCellX = std::max(0, (int)floorf((x - mnMinX - radius) * invW));
    if (nMinCellX >= COLS) continue;
    const int nMaxCellX = std::min(COLS - 1, (int)ceilf((x - mnMinX + radius) * invW));
    if (nMaxCellX < 0) continue;
    const int nMinCellY = std::max(0, (int)floorf((y - mnMinY - radius) * invH));
    if (nMinCellY >= ROWS) continue;
    const int nMaxCellY = std::min(ROWS - 1, (int)ceilf((y - mnMinY + radius) * invH));
    if (nMaxCellY < 0) continue;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    bool any = false;
    const uint8_t* d1 = in->mp_desc1 + 32 * (size_t)i;
    for (int ix = nMinCellX; ix <= nMaxCellX; ++ix)
      for (int iy = nMinCellY; iy <= nMaxCellY; ++iy)
        for (int idx : cells[(size_t)ix * ROWS + iy]) {
          if (bCheckLevels) {
            if (in->kp2_octave[idx] < minLevel) continue;
            if (maxLevel >= 0 && in->kp2_octave[idx] > maxLevel) continue;
          }
          const float distx = in->kp2_xy[2 * idx] - x, disty = in->kp2_xy[2 * idx + 1] - y;
          if (!(fabsf(distx) < radius && fabsf(disty) < radius)) continue;
          any = true;
          // F.mvpMapPoints[idx] holds a point with observations: on entry, or assigned earlier in this loop
          if (in->blocked2[idx] && holder[idx] < 0) continue;
          if (holder[idx] >= 0 && in->mp_observed1[holder[idx]]) continue;
          if (in->uright2[idx] > 0) {
            const float er = fabsf(xr - in->uright2[idx]);
            if (er > radius) continue;
          }
          const int dist = hamming256(d1, in->desc2 + 32 * (size_t)idx);
          if (dist < bestDist) {
            bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = in->kp2_octave[idx]; bestIdx = idx;
          } else if (dist < bestDist2) {
            bestLevel2 = in->kp2_octave[idx]; bestDist2 = dist;
          }
        }
    if (!any) continue;
    if (bestDist <= TH_HIGH) {
      if (bestLevel == bestLevel2 && (float)bestDist > in->nnratio * (float)bestDist2) continue;
      if (bestLevel != bestLevel2 || (float)bestDist <= in->nnratio * (float)bestDist2) {
        holder[bestIdx] = i;
        ++nmatches;
      }
    }
  }
  for (int i = 0; i < in->n2; ++i) match2[i] = holder[i];
  return nmatches;
}

// ---- DBoW2 vocabulary descent, Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1208-1255 ----
extern "C" void orc_bow_descend(const orc_vocabulary* v, const uint8_t* desc, int n, int levelsup, int32_t* word, double* weight,
                                int32_t* node) {
  const int nid_level = v->L - levelsup;
  for (int i = 0; i < n; ++i) {
    const uint8_t* f = desc + 32 * (size_t)i;
    int nid = 0;  // root when nid_level <= 0 (a leaf above nid_level leaves the reference's value uninitialised)
    int final_id = 0, current_level = 0;
    do {
      ++current_level;
      const int b = v->child_off[final_id], e = v->child_off[final_id + 1];
      final_id = v->child[b];
      int best_d = hamming256(f, v->desc + 32 * (size_t)final_id);
      for (int k = b + 1; k < e; ++k) {
        const int id = v->child[k];
        const int d = hamming256(f, v->desc + 32 * (size_t)id);
        if (d < best_d) { best_d = d; final_id = id; }
      }
      if (current_level == nid_level) nid = final_id;
    } while (v->child_off[final_id] != v->child_off[final_id + 1]);
    word[i] = v->word_id[final_id];
    weight[i] = v->weight[final_id];
    node[i] = nid;
  }
}

// ---- ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&), /root/reference/src/ORBmatcher.cc:223-425 (Nleft == -1) ----
extern "C" int orc_search_by_bow_kf(const orc_tri_input* in, float nnratio, int check_orientation, int* match12) {
  for (int i = 0; i < in->n1; ++i) match12[i] = -1;
  std::vector<bool> vbMatched2(in->n2, false);
  std::vector<int> rot_hist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;
  int nmatches = 0;
  int a = 0, b = 0;
  while (a < in->nnodes1 && b < in->nnodes2) {
    if (in->node_id1[a] < in->node_id2[b]) { ++a; continue; }
    if (in->node_id1[a] > in->node_id2[b]) { ++b; continue; }
    for (int p = in->node_off1[a]; p < in->node_off1[a + 1]; ++p) {
      const int idx1 = in->node_feat1[p];
      if (!in->has_mp1[idx1]) continue;  // !pMP1 || pMP1->isBad()
      const uint8_t* d1 = in->desc1 + 32 * (size_t)idx1;
      int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
      for (int q = in->node_off2[b]; q < in->node_off2[b + 1]; ++q) {
        const int idx2 = in->node_feat2[q];
        if (vbMatched2[idx2] || !in->has_mp2[idx2]) continue;
        const int dist = hamming256(d1, in->desc2 + 32 * (size_t)idx2);
        if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = idx2; }
        else if (dist < bestDist2) bestDist2 = dist;
      }
      if (bestDist1 < TH_LOW && (float)bestDist1 < nnratio * (float)bestDist2) {
        match12[idx1] = bestIdx2;
        vbMatched2[bestIdx2] = true;
        if (check_orientation) {
          float rot = in->kp1_angle[idx1] - in->kp2_angle[bestIdx2];
          if (rot < 0.0) rot += 360.0f;
          int bin = (int)roundf(rot * factor);
          if (bin == HISTO_LENGTH) bin = 0;
          rot_hist[bin].push_back(idx1);
        }
        ++nmatches;
      }
    }
    ++a;
    ++b;
  }
  if (check_orientation) {
    int i1 = -1, i2 = -1, i3 = -1;
    three_maxima(rot_hist, HISTO_LENGTH, i1, i2, i3);
    for (int i = 0; i < HISTO_LENGTH; ++i) {
      if (i == i1 || i == i2 || i == i3) continue;
      for (int idx : rot_hist[i]) { match12[idx] = -1; --nmatches; }
    }
  }
  return nmatches;
}

extern "C" int orc_search_by_bow(const orc_tri_input* in, float nnratio, int check_orientation, int* match2) {
  for (int i = 0; i < in->n2; ++i) match2[i] = -1;
  std::vector<int> rot_hist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;
  int nmatches = 0;
  int a = 0, b = 0;
  while (a < in->nnodes1 && b < in->nnodes2) {
    if (in->node_id1[a] < in->node_id2[b]) { ++a; continue; }  // lower_bound on a sorted map
    if (in->node_id1[a] > in->node_id2[b]) { ++b; continue; }
    for (int p = in->node_off1[a]; p < in->node_off1[a + 1]; ++p) {
      const int realIdxKF = in->node_feat1[p];
      if (!in->has_mp1[realIdxKF]) continue;  // no map point, or a bad one
      const uint8_t* dKF = in->desc1 + 32 * (size_t)realIdxKF;
      int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
      for (int q = in->node_off2[b]; q < in->node_off2[b + 1]; ++q) {
        const int realIdxF = in->node_feat2[q];
        if (match2[realIdxF] >= 0) continue;
        const int dist = hamming256(dKF, in->desc2 + 32 * (size_t)realIdxF);
        if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
        else if (dist < bestDist2) bestDist2 = dist;
      }
      if (bestDist1 <= TH_LOW && (float)bestDist1 < nnratio * (float)bestDist2) {
        match2[bestIdxF] = realIdxKF;
        if (check_orientation) {
          float rot = in->kp1_angle[realIdxKF] - in->kp2_angle[bestIdxF];
          if (rot < 0.0) rot += 360.0f;
          int bin = (int)roundf(rot * factor);
          if (bin == HISTO_LENGTH) bin = 0;
          rot_hist[bin].push_back(bestIdxF);
        }
        ++nmatches;
      }
    }
    ++a;
    ++b;
  }
  if (check_orientation) {
    int i1 = -1, i2 = -1, i3 = -1;
    three_maxima(rot_hist, HISTO_LENGTH, i1, i2, i3);
    for (int i = 0; i < HISTO_LENGTH; ++i) {
      if (i == i1 || i == i2 || i == i3) continue;
      for (int idx : rot_hist[i]) { match2[idx] = -1; --nmatches; }
    }
  }
  return nmatches;
}

// ---- ORBmatcher::SearchByBoW(pKF, F, vpMapPointMatches) for a two-camera frame (F.Nleft != -1), ORBmatcher.cc:298-326, 357-386 ----
// n_left = F.Nleft.  kp1_angle / kp2_angle: the angle of the key point the reference reads for that index (:335-343, :362-373).
extern "C" int orc_search_by_bow_rig(const orc_tri_input* in, int n_left, float nnratio, int check_orientation, int* match2) {
  for (int i = 0; i < in->n2; ++i) match2[i] = -1;
  std::vector<int> rot_hist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;
  int nmatches = 0;
  int a = 0, b = 0;
  while (a < in->nnodes1 && b < in->nnodes2) {
    if (in->node_id1[a] < in->node_id2[b]) { ++a; continue; }
    if (in->node_id1[a] > in->node_id2[b]) { ++b; continue; }
    for (int p = in->node_off1[a]; p < in->node_off1[a + 1]; ++p) {
      const int realIdxKF = in->node_feat1[p];
      if (!in->has_mp1[realIdxKF]) continue;
      const uint8_t* dKF = in->desc1 + 32 * (size_t)realIdxKF;
      int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
      int bestDist1R = 256, bestIdxFR = -1, bestDist2R = 256;
      for (int q = in->node_off2[b]; q < in->node_off2[b + 1]; ++q) {
        const int realIdxF = in->node_feat2[q];
        if (match2[realIdxF] >= 0) continue;
        const int dist = hamming256(dKF, in->desc2 + 32 * (size_t)realIdxF);
        if (realIdxF < n_left && dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
        else if (realIdxF < n_left && dist < bestDist2) bestDist2 = dist;
        if (realIdxF >= n_left && dist < bestDist1R) { bestDist2R = bestDist1R; bestDist1R = dist; bestIdxFR = realIdxF; }
        else if (realIdxF >= n_left && dist < bestDist2R) bestDist2R = dist;
      }
      if (bestDist1 <= TH_LOW) {
        const int taken[2] = {(float)bestDist1 < nnratio * (float)bestDist2 ? bestIdxF : -1, bestDist1R <= TH_LOW ? bestIdxFR : -1};
        for (int idx : taken) {
          if (idx < 0) continue;
          match2[idx] = realIdxKF;
          if (check_orientation) {
            float rot = in->kp1_angle[realIdxKF] - in->kp2_angle[idx];
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            rot_hist[bin].push_back(idx);
          }
          ++nmatches;
        }
      }
    }
    ++a;
    ++b;
  }
  (void)0;
  if (check_orientation) {
    int i1 = -1, i2 = -1, i3 = -1;
    three_maxima(rot_hist, HISTO_LENGTH, i1, i2, i3);
    for (int i = 0; i < HISTO_LENGTH; ++i) {
      if (i == i1 || i == i2 || i == i3) continue;
      for (int idx : rot_hist[i]) { match2[idx] = -1; --nmatches; }
    }
  }
  return nmatches;
}

// ---- ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize), ORBmatcher.cc:648-763 ----
// with Frame::GetFeaturesInArea(x, y, r, minLevel, maxLevel) (src/Frame.cc:747-813) over AssignFeaturesToGrid's cells
extern "C" int orc_search_for_initialization(const orc_initialization_input* in, float* prev_matched, int* matches12) {
  const int COLS = 64, ROWS = 48;
  const float mnMinX = in->grid[0], mnMinY = in->grid[1], invW = in->grid[4], invH = in->grid[5];
  std::vector<std::vector<int>> cells((size_t)COLS * ROWS);
  for (int i = 0; i < in->n2; ++i) {
    const int px = (int)roundf((in->kp2_xy[2 * i] - mnMinX) * invW), py = (int)roundf((in->kp2_xy[2 * i + 1] - mnMinY) * invH);
    if (px < 0 || px >= COLS || py < 0 || py >= ROWS) continue;
    cells[(size_t)px * ROWS + py].push_back(i);
  }
  int nmatches = 0;
  for (int i = 0; i < in->n1; ++i) matches12[i] = -1;
  std::vector<int> rot_hist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;
  std::vector<int> vMatchedDistance(in->n2, INT_MAX), vnMatches21(in->n2, -1);
  for (int i1 = 0; i1 < in->n1; ++i1) {
    const int level1 = in->kp1_octave[i1];
    if (level1 > 0) continue;
    const float x = prev_matched[2 * i1], y = prev_matched[2 * i1 + 1], r = (float)in->window_size;
    const int minLevel = level1, maxLevel = level1;
    if (!(x == x) || !(y == y)) continue;
    const int nMinCellX = std::max(0, (int)floorf((x - mnMinX - r) * invW));
    if (nMinCellX >= COLS) continue;
    const int nMaxCellX = std::min(COLS - 1, (int)ceilf((x - mnMinX + r) * invW));
    if (nMaxCellX < 0) continue;
    const int nMinCellY = std::max(0, (int)floorf((y - mnMinY - r) * invH));
    if (nMinCellY >= ROWS) continue;
    const int nMaxCellY = std::min(ROWS - 1, (int)ceilf((y - mnMinY + r) * invH));
    if (nMaxCellY < 0) continue;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    const uint8_t* d1 = in->desc1 + 32 * (size_t)i1;
    int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
    for (int ix = nMinCellX; ix <= nMaxCellX; ++ix)
      for (int iy = nMinCellY; iy <= nMaxCellY; ++iy)
        for (int i2 : cells[(size_t)ix * ROWS + iy]) {
          if (bCheckLevels) {
            if (in->kp2_octave[i2] < minLevel) continue;
            if (maxLevel >= 0 && in->kp2_octave[i2] > maxLevel) continue;
          }
          const float distx = in->kp2_xy[2 * i2] - x, disty = in->kp2_xy[2 * i2 + 1] - y;
          if (!(fabsf(distx) < r && fabsf(disty) < r)) continue;
          const int dist = hamming256(d1, in->desc2 + 32 * (size_t)i2);
          if (vMatchedDistance[i2] <= dist) continue;
          if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
          else if (dist < bestDist2) bestDist2 = dist;
        }
    if (bestDist <= TH_LOW && (float)bestDist < (float)bestDist2 * in->nnratio) {
      if (vnMatches21[bestIdx2] >= 0) { matches12[vnMatches21[bestIdx2]] = -1; --nmatches; }
      matches12[i1] = bestIdx2;
      vnMatches21[bestIdx2] = i1;
      vMatchedDistance[bestIdx2] = bestDist;
      ++nmatches;
      if (in->check_orientation) {
        float rot = in->kp1_angle[i1] - in->kp2_angle[bestIdx2];
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)roundf(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        rot_hist[bin].push_back(i1);
      }
    }
  }
  if (in->check_orientation) {
    int i1 = -1, i2 = -1, i3 = -1;
    three_maxima(rot_hist, HISTO_LENGTH, i1, i2, i3);
    for (int i = 0; i < HISTO_LENGTH; ++i) {
      if (i == i1 || i == i2 || i == i3) continue;
      for (int idx1 : rot_hist[i])
        if (matches12[idx1] >= 0) { matches12[idx1] = -1; --nmatches; }
    }
  }
  for (int i1 = 0; i1 < in->n1; ++i1)
    if (matches12[i1] >= 0) {
      prev_matched[2 * i1] = in->kp2_xy[2 * matches12[i1]];
      prev_matched[2 * i1 + 1] = in->kp2_xy[2 * matches12[i1] + 1];
    }
  return nmatches;
}
