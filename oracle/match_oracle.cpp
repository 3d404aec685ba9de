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
#include <vector>

namespace {
const int TH_LOW = 50;
const int HISTO_LENGTH = 30;

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
