// matcher.hip — 256-bit Hamming matching: brute force best/second-best and SearchForTriangulation.
//
// Replaces ORB_SLAM3::ORBmatcher::DescriptorDistance (/root/reference/src/ORBmatcher.cc:2058-2074),
// ORBmatcher::SearchForTriangulation (:907-1146) incl. ComputeThreeMaxima (:2012-2053) and the pinhole
// epipolar test (/root/reference/src/CameraModels/Pinhole.cpp:107-129).
//
// Kernels:
//   k_hamming_bf          one query descriptor per lane (4 x u64 in VGPRs); the train descriptors are read
//                         through wave-uniform addresses (scalar loads, broadcast to the 64 lanes), XOR +
//                         v_bcnt popcount, running best / second-best with the reference's strict '<'.
//   k_search_triangulation one workgroup per BoW node shared by both key-frames, one query feature per
//                         work-item, the node's candidate bucket is walked in index order (ties -> later
//                         candidate wins, as in the reference) with the eligibility masks and epipolar test.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "common.h"

namespace rgbl {

__device__ __forceinline__ int hamming256(const unsigned long long q[4], const unsigned long long* __restrict__ t) {
  return __popcll(q[0] ^ t[0]) + __popcll(q[1] ^ t[1]) + __popcll(q[2] ^ t[2]) + __popcll(q[3] ^ t[3]);
}

// A workgroup owns 64 query descriptors (one per lane, 4 x u64 in VGPRs).  Its four waves split the train set
// into four contiguous index ranges; inside a wave the train descriptor address is wave-uniform, so it is
// fetched with scalar loads and broadcast to the 64 lanes.  Best and second-best are tracked on packed words
// (distance << 16 | train index): min() then implements the reference's "strict '<', first minimum wins" and the
// second-best is the second smallest word, whose distance part is the second smallest distance counted with
// multiplicity.  The four partial results are merged through LDS with the same two operations.
// grid = (ceil(cap/64), n_pairs), block = 256.  Requires train counts < 65536.
__device__ __forceinline__ void bf_track(uint32_t& best, uint32_t& second, uint32_t cur) {
  const uint32_t hi = best > cur ? best : cur;
  best = best < cur ? best : cur;
  second = second < hi ? second : hi;
}

__global__ __launch_bounds__(256) void k_hamming_bf(const uint8_t* __restrict__ desc, const int32_t* __restrict__ n_rows,
                                                    int cap, const int32_t* __restrict__ pair_a,
                                                    const int32_t* __restrict__ pair_b, int32_t* __restrict__ best_idx,
                                                    int32_t* __restrict__ best_dist, int32_t* __restrict__ second_dist) {
  __shared__ uint32_t s_best[4][64], s_second[4][64];
  const int p = blockIdx.y;
  const int fa = pair_a ? pair_a[p] : 0, fb = pair_b ? pair_b[p] : 1;
  const int na = n_rows[fa], nb = n_rows[fb];
  if ((int)blockIdx.x * 64 >= na) return;
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(wave_id());  // make the train range provably wave-uniform
  const int i = blockIdx.x * 64 + lane;
  const bool valid = i < na;
  const unsigned long long* A = reinterpret_cast<const unsigned long long*>(desc + ((size_t)fa * cap + (valid ? i : 0)) * 32);
  const unsigned long long* B = reinterpret_cast<const unsigned long long*>(desc + (size_t)fb * cap * 32);
  const unsigned long long q[4] = {A[0], A[1], A[2], A[3]};
  const int chunk = (nb + 3) >> 2;
  const int j0 = wave * chunk, j1 = imin(j0 + chunk, nb);
  const uint32_t kNone = (256u << 16) | 0xffffu;
  uint32_t best = kNone, second = kNone;
  int j = j0;
  for (; j + 8 <= j1; j += 8) {
    const unsigned long long* t = B + 4 * (size_t)j;
    uint32_t e[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) e[u] = ((uint32_t)hamming256(q, t + 4 * u) << 16) | (uint32_t)(j + u);
#pragma unroll
    for (int u = 0; u < 8; ++u) bf_track(best, second, e[u]);
  }
  for (; j < j1; ++j) bf_track(best, second, ((uint32_t)hamming256(q, B + 4 * (size_t)j) << 16) | (uint32_t)j);
  s_best[wave][lane] = best;
  s_second[wave][lane] = second;
  __syncthreads();
  if (wave == 0 && valid) {
    for (int w = 1; w < 4; ++w) {
      const uint32_t ob = s_best[w][lane], os = s_second[w][lane];
      const uint32_t hi = best > ob ? best : ob;
      best = best < ob ? best : ob;
      second = second < os ? second : os;
      second = second < hi ? second : hi;
    }
    const size_t o = (size_t)p * cap + i;
    best_idx[o] = best == kNone ? -1 : (int32_t)(best & 0xffffu);
    best_dist[o] = (int32_t)(best >> 16);
    if (second_dist) second_dist[o] = (int32_t)(second >> 16);
  }
}

struct TriDev {
  const uint8_t *desc1, *desc2;
  const float *xy1, *xy2;
  const int32_t* oct2;
  const float *ur1, *ur2;
  const uint8_t *mp1, *mp2;
  const int32_t *off1, *feat1, *off2, *feat2;  // CSR buckets
  const int32_t *pair_n1, *pair_n2;            // matched node pairs (indices into off1/off2)
  float F[9], ep[2];
  const float *scale2, *sigma2;
  int only_stereo, coarse;
  int32_t* matches12;
};

// grid = number of node ids present in both feature vectors, block = 256.
__global__ __launch_bounds__(256) void k_search_triangulation(TriDev T) {
  const int np = blockIdx.x;
  const int a = T.pair_n1[np], b = T.pair_n2[np];
  const int b1 = T.off1[a], e1 = T.off1[a + 1], b2 = T.off2[b], e2 = T.off2[b + 1];
  for (int p = b1 + (int)threadIdx.x; p < e1; p += 256) {
    const int idx1 = T.feat1[p];
    if (T.mp1[idx1]) continue;
    const bool stereo1 = T.ur1[idx1] >= 0;
    if (T.only_stereo && !stereo1) continue;
    const float x1 = T.xy1[2 * idx1], y1 = T.xy1[2 * idx1 + 1];
    const unsigned long long* D1 = reinterpret_cast<const unsigned long long*>(T.desc1 + (size_t)idx1 * 32);
    const unsigned long long q[4] = {D1[0], D1[1], D1[2], D1[3]};
    // epipolar line of kp1 in image 2 (Pinhole.cpp:115-117), constant over the candidates
    const float la = x1 * T.F[0] + y1 * T.F[3] + T.F[6];
    const float lb = x1 * T.F[1] + y1 * T.F[4] + T.F[7];
    const float lc = x1 * T.F[2] + y1 * T.F[5] + T.F[8];
    const float den = la * la + lb * lb;
    int best_dist = 50 /* TH_LOW */, best_idx2 = -1;
    for (int qi = b2; qi < e2; ++qi) {
      const int idx2 = T.feat2[qi];
      if (T.mp2[idx2]) continue;
      const bool stereo2 = T.ur2[idx2] >= 0;
      if (T.only_stereo && !stereo2) continue;
      const int dist = hamming256(q, reinterpret_cast<const unsigned long long*>(T.desc2 + (size_t)idx2 * 32));
      if (dist > 50 || dist > best_dist) continue;
      const float x2 = T.xy2[2 * idx2], y2 = T.xy2[2 * idx2 + 1];
      const int oct2 = T.oct2[idx2];
      if (!stereo1 && !stereo2) {
        const float ex = T.ep[0] - x2, ey = T.ep[1] - y2;
        if (ex * ex + ey * ey < 100 * T.scale2[oct2]) continue;
      }
      bool ok = T.coarse != 0;
      if (!ok && den != 0) {
        const float num = la * x2 + lb * y2 + lc;
        const float dsqr = __fdiv_rn(num * num, den);
        ok = (double)dsqr < 3.84 * (double)T.sigma2[oct2];  // 3.84 is a double literal in the reference
      }
      if (ok) { best_idx2 = idx2; best_dist = dist; }
    }
    if (best_idx2 >= 0) T.matches12[idx1] = best_idx2;
  }
}

}  // namespace rgbl

using namespace rgbl;

struct rgbl_matcher {
  int device = 0;
  hipStream_t stream = nullptr, own_stream = nullptr;
  KernelTimer timer;
  uint8_t* d_buf = nullptr;  // grow-only staging arena for the host entry points
  size_t buf_size = 0;
};

namespace {
int ensure_arena(rgbl_matcher* m, size_t bytes) {
  if (bytes <= m->buf_size) return RGBL_OK;
  if (m->d_buf) { RGBL_HIP(hipStreamSynchronize(m->stream)); RGBL_HIP(hipFree(m->d_buf)); m->d_buf = nullptr; m->buf_size = 0; }
  const size_t sz = std::max(bytes, (size_t)1 << 20);
  RGBL_HIP(hipMalloc(&m->d_buf, sz));
  m->buf_size = sz;
  return RGBL_OK;
}
struct Arena {
  uint8_t* base; size_t off = 0;
  template <class T> T* take(size_t count) {
    off = (off + 255) / 256 * 256;
    T* p = reinterpret_cast<T*>(base + off);
    off += count * sizeof(T);
    return p;
  }
};
inline size_t pad256(size_t b) { return (b + 255) / 256 * 256 + 256; }
template <class E>
int upload(Arena& A, hipStream_t s, const E** dst, const E* src, size_t count) {
  E* p = A.take<E>(count);
  *dst = p;
  RGBL_HIP(hipMemcpyAsync(p, src, count * sizeof(E), hipMemcpyHostToDevice, s));
  return RGBL_OK;
}
}  // namespace

extern "C" {

int rgbl_matcher_create(int device, rgbl_matcher** out) {
  if (!out) { set_error("null argument"); return RGBL_ERR_INVALID; }
  *out = nullptr;
  if (rgbl_device_count() <= device || device < 0) {
    set_error("no usable HIP device %d (this library has no CPU fallback)", device);
    return RGBL_ERR_NO_DEVICE;
  }
  RGBL_HIP(hipSetDevice(device));
  rgbl_matcher* m = new rgbl_matcher;
  m->device = device;
  if (hipStreamCreate(&m->own_stream) != hipSuccess) { delete m; set_error("hipStreamCreate failed"); return RGBL_ERR_HIP; }
  m->stream = m->own_stream;
  *out = m;
  return RGBL_OK;
}

void rgbl_matcher_destroy(rgbl_matcher* m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  (void)hipStreamSynchronize(m->stream);
  m->timer.collect();
  if (m->d_buf) (void)hipFree(m->d_buf);
  if (m->own_stream) (void)hipStreamDestroy(m->own_stream);
  delete m;
}

int rgbl_matcher_sync(rgbl_matcher* m) {
  if (!m) { set_error("null handle"); return RGBL_ERR_INVALID; }
  RGBL_HIP(hipSetDevice(m->device));
  RGBL_HIP(hipStreamSynchronize(m->stream));
  m->timer.collect();
  return RGBL_OK;
}
int rgbl_matcher_set_stream(rgbl_matcher* m, void* hip_stream) {
  if (!m) { set_error("null handle"); return RGBL_ERR_INVALID; }
  RGBL_HIP(hipStreamSynchronize(m->stream));
  m->stream = hip_stream ? (hipStream_t)hip_stream : m->own_stream;
  return RGBL_OK;
}
void* rgbl_matcher_stream(rgbl_matcher* m) { return m ? (void*)m->stream : nullptr; }
int rgbl_matcher_profile(rgbl_matcher* m, int enable) {
  if (!m) { set_error("null handle"); return RGBL_ERR_INVALID; }
  RGBL_HIP(hipStreamSynchronize(m->stream));
  m->timer.reset();
  m->timer.enabled = enable != 0;
  return RGBL_OK;
}
int rgbl_matcher_profile_read(rgbl_matcher* m, const char** names, double* total_ms, long* launches, int cap) {
  if (!m) return 0;
  (void)hipStreamSynchronize(m->stream);
  m->timer.collect();
  const int n = (int)m->timer.names.size();
  for (int i = 0; i < n && i < cap; ++i) {
    if (names) names[i] = m->timer.names[i].c_str();
    if (total_ms) total_ms[i] = m->timer.total_ms[i];
    if (launches) launches[i] = m->timer.count[i];
  }
  return n;
}

int rgbl_descriptor_distance(const uint8_t* a, const uint8_t* b) {
  // host helper with the reference's word-wise semantics (8 x 32-bit little-endian words)
  int dist = 0;
  for (int i = 0; i < 8; ++i) {
    uint32_t wa, wb;
    memcpy(&wa, a + 4 * i, 4);
    memcpy(&wb, b + 4 * i, 4);
    dist += __builtin_popcount(wa ^ wb);
  }
  return dist;
}

int rgbl_hamming_bf_batch_device(rgbl_matcher* m, const uint8_t* d_desc, const int32_t* d_n, int cap, const int32_t* d_pair_a,
                                 const int32_t* d_pair_b, int n_pairs, int32_t* d_best_idx, int32_t* d_best_dist,
                                 int32_t* d_second_dist) {
  if (!m || !d_desc || !d_n || !d_pair_a || !d_pair_b || !d_best_idx || !d_best_dist || cap < 1 || cap > 65535 || n_pairs < 0) {
    set_error("invalid argument");
    return RGBL_ERR_INVALID;
  }
  if (n_pairs == 0) return RGBL_OK;
  RGBL_HIP(hipSetDevice(m->device));
  m->timer.begin("k_hamming_bf", m->stream);
  hipLaunchKernelGGL(k_hamming_bf, dim3((cap + 63) / 64, n_pairs), dim3(256), 0, m->stream, d_desc, d_n, cap, d_pair_a,
                     d_pair_b, d_best_idx, d_best_dist, d_second_dist);
  m->timer.end(m->stream);
  RGBL_HIP(hipGetLastError());
  return RGBL_OK;
}

int rgbl_hamming_bf(rgbl_matcher* m, const uint8_t* desc_a, int na, const uint8_t* desc_b, int nb, int32_t* best_idx,
                    int32_t* best_dist, int32_t* second_dist) {
  if (!m || na < 0 || nb < 0 || nb > 65535 || (na > 0 && (!desc_a || !best_idx || !best_dist)) || (nb > 0 && !desc_b)) {
    set_error("invalid argument");
    return RGBL_ERR_INVALID;
  }
  if (na == 0) return RGBL_OK;
  RGBL_HIP(hipSetDevice(m->device));
  const int cap = std::max(std::max(na, nb), 1);
  RGBL_TRY(ensure_arena(m, pad256((size_t)2 * cap * 32) + pad256(8) + 3 * pad256((size_t)na * 4)));
  Arena A{m->d_buf};
  uint8_t* d_desc = A.take<uint8_t>((size_t)2 * cap * 32);
  int32_t* d_n = A.take<int32_t>(2);
  int32_t* d_bi = A.take<int32_t>(na);
  int32_t* d_bd = A.take<int32_t>(na);
  int32_t* d_sd = A.take<int32_t>(na);
  hipStream_t s = m->stream;
  const int32_t counts[2] = {na, nb};
  RGBL_HIP(hipMemcpyAsync(d_desc, desc_a, (size_t)na * 32, hipMemcpyHostToDevice, s));
  if (nb > 0) RGBL_HIP(hipMemcpyAsync(d_desc + (size_t)cap * 32, desc_b, (size_t)nb * 32, hipMemcpyHostToDevice, s));
  RGBL_HIP(hipMemcpyAsync(d_n, counts, sizeof(counts), hipMemcpyHostToDevice, s));
  m->timer.begin("k_hamming_bf", s);
  // pair_a/pair_b == NULL selects the fixed pair (frame 0 -> frame 1)
  hipLaunchKernelGGL(k_hamming_bf, dim3((na + 63) / 64, 1), dim3(256), 0, s, d_desc, d_n, cap, (const int32_t*)nullptr,
                     (const int32_t*)nullptr, d_bi, d_bd, d_sd);
  m->timer.end(s);
  RGBL_HIP(hipGetLastError());
  RGBL_HIP(hipMemcpyAsync(best_idx, d_bi, sizeof(int32_t) * na, hipMemcpyDeviceToHost, s));
  RGBL_HIP(hipMemcpyAsync(best_dist, d_bd, sizeof(int32_t) * na, hipMemcpyDeviceToHost, s));
  if (second_dist) RGBL_HIP(hipMemcpyAsync(second_dist, d_sd, sizeof(int32_t) * na, hipMemcpyDeviceToHost, s));
  RGBL_HIP(hipStreamSynchronize(s));
  m->timer.collect();
  return RGBL_OK;
}

void rgbl_fundamental(const float K1[4], const float K2[4], const float R12[9], const float t12[3], float F12[9]) {
  // Pinhole.cpp:109-112: K1^T^-1 * hat(t12) * R12 * K2^-1 in fp32; Eigen evaluates 3x3 products coefficient-wise
  // as p0 + (p1 + p2) and inverts 3x3 matrices through cofactors (determinant from column 0).
  auto mul = [](const float* A, const float* B, float* C) {
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        const float p0 = A[3 * r] * B[c], p1 = A[3 * r + 1] * B[3 + c], p2 = A[3 * r + 2] * B[6 + c];
        C[3 * r + c] = p0 + (p1 + p2);
      }
  };
  auto cof = [](const float* m, int i, int j) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return m[3 * i1 + j1] * m[3 * i2 + j2] - m[3 * i1 + j2] * m[3 * i2 + j1];
  };
  auto inv = [&](const float* m, float* r) {
    const float c0 = cof(m, 0, 0), c1 = cof(m, 1, 0), c2 = cof(m, 2, 0);
    const float det = c0 * m[0] + (c1 * m[3] + c2 * m[6]);
    const float id = 1.0f / det;
    r[0] = c0 * id; r[1] = c1 * id; r[2] = c2 * id;
    r[3] = cof(m, 0, 1) * id; r[4] = cof(m, 1, 1) * id; r[5] = cof(m, 2, 1) * id;
    r[6] = cof(m, 0, 2) * id; r[7] = cof(m, 1, 2) * id; r[8] = cof(m, 2, 2) * id;
  };
  const float k1t[9] = {K1[0], 0.f, 0.f, 0.f, K1[1], 0.f, K1[2], K1[3], 1.f};
  const float k2[9] = {K2[0], 0.f, K2[2], 0.f, K2[1], K2[3], 0.f, 0.f, 1.f};
  const float tx[9] = {0.f, -t12[2], t12[1], t12[2], 0.f, -t12[0], -t12[1], t12[0], 0.f};
  float a[9], b[9], c[9], k2i[9];
  inv(k1t, a);
  mul(a, tx, b);
  mul(b, R12, c);
  inv(k2, k2i);
  mul(c, k2i, F12);
}

int rgbl_search_triangulation(rgbl_matcher* m, const rgbl_keyframe_view* k1, const rgbl_keyframe_view* k2,
                              const rgbl_triangulation_params* prm, int32_t* matches12, int* out_nmatches) {
  if (!m || !k1 || !k2 || !prm || !matches12 || !out_nmatches || k1->n < 0 || k2->n < 0 || prm->n_levels < 1) {
    set_error("invalid argument");
    return RGBL_ERR_INVALID;
  }
  *out_nmatches = 0;
  const int n1 = k1->n, n2 = k2->n;
  for (int i = 0; i < n1; ++i) matches12[i] = -1;
  // merge walk of the two sorted FeatureVectors (ORBmatcher.cc:963-968, 1103-1116)
  std::vector<int32_t> pa, pb;
  for (int a = 0, b = 0; a < k1->n_nodes && b < k2->n_nodes;) {
    if (k1->node_id[a] == k2->node_id[b]) { pa.push_back(a++); pb.push_back(b++); }
    else if (k1->node_id[a] < k2->node_id[b]) ++a;
    else ++b;
  }
  const int npairs = (int)pa.size();
  if (npairs > 0 && n1 > 0 && n2 > 0) {
    RGBL_HIP(hipSetDevice(m->device));
    const int nf1 = k1->node_off[k1->n_nodes], nf2 = k2->node_off[k2->n_nodes];
    size_t need = pad256((size_t)n1 * 32) + pad256((size_t)n2 * 32) + pad256((size_t)n1 * 8) + pad256((size_t)n2 * 8) +
                  pad256((size_t)n2 * 4) + pad256((size_t)n1 * 4) + pad256((size_t)n2 * 4) + pad256(n1) + pad256(n2) +
                  pad256((size_t)(k1->n_nodes + 1) * 4) + pad256((size_t)nf1 * 4) + pad256((size_t)(k2->n_nodes + 1) * 4) +
                  pad256((size_t)nf2 * 4) + 2 * pad256((size_t)npairs * 4) + 2 * pad256((size_t)prm->n_levels * 4) +
                  pad256((size_t)n1 * 4);
    RGBL_TRY(ensure_arena(m, need));
    Arena A{m->d_buf};
    hipStream_t s = m->stream;
    TriDev T;
    RGBL_TRY(upload(A, s, &T.desc1, k1->desc, (size_t)n1 * 32));
    RGBL_TRY(upload(A, s, &T.desc2, k2->desc, (size_t)n2 * 32));
    RGBL_TRY(upload(A, s, &T.xy1, k1->kp_xy, (size_t)n1 * 2));
    RGBL_TRY(upload(A, s, &T.xy2, k2->kp_xy, (size_t)n2 * 2));
    RGBL_TRY(upload(A, s, &T.oct2, k2->kp_octave, (size_t)n2));
    RGBL_TRY(upload(A, s, &T.ur1, k1->uright, (size_t)n1));
    RGBL_TRY(upload(A, s, &T.ur2, k2->uright, (size_t)n2));
    RGBL_TRY(upload(A, s, &T.mp1, k1->has_mappoint, (size_t)n1));
    RGBL_TRY(upload(A, s, &T.mp2, k2->has_mappoint, (size_t)n2));
    RGBL_TRY(upload(A, s, &T.off1, k1->node_off, (size_t)k1->n_nodes + 1));
    RGBL_TRY(upload(A, s, &T.feat1, k1->node_feat, (size_t)nf1));
    RGBL_TRY(upload(A, s, &T.off2, k2->node_off, (size_t)k2->n_nodes + 1));
    RGBL_TRY(upload(A, s, &T.feat2, k2->node_feat, (size_t)nf2));
    RGBL_TRY(upload(A, s, &T.pair_n1, pa.data(), (size_t)npairs));
    RGBL_TRY(upload(A, s, &T.pair_n2, pb.data(), (size_t)npairs));
    RGBL_TRY(upload(A, s, &T.scale2, prm->scale_factors2, (size_t)prm->n_levels));
    RGBL_TRY(upload(A, s, &T.sigma2, prm->level_sigma2_2, (size_t)prm->n_levels));
    T.matches12 = A.take<int32_t>(n1);
    RGBL_HIP(hipMemsetAsync(T.matches12, 0xff, sizeof(int32_t) * n1, s));
    memcpy(T.F, prm->F12, sizeof(T.F));
    T.ep[0] = prm->epipole[0];
    T.ep[1] = prm->epipole[1];
    T.only_stereo = prm->only_stereo;
    T.coarse = prm->coarse;
    m->timer.begin("k_search_triangulation", s);
    hipLaunchKernelGGL(k_search_triangulation, dim3(npairs), dim3(256), 0, s, T);
    m->timer.end(s);
    RGBL_HIP(hipGetLastError());
    RGBL_HIP(hipMemcpyAsync(matches12, T.matches12, sizeof(int32_t) * n1, hipMemcpyDeviceToHost, s));
    RGBL_HIP(hipStreamSynchronize(s));
    m->timer.collect();
  }
  int nmatches = 0;
  for (int i = 0; i < n1; ++i) nmatches += matches12[i] >= 0;
  if (prm->check_orientation) {
    // rotation-consistency histogram (ORBmatcher.cc:1083-1096, 1119-1136) + ComputeThreeMaxima (:2012-2053).
    // Bins are filled in the order the reference visits idx1: node by node, bucket order.
    std::vector<int> hist[30];
    const float factor = 1.0f / 30;
    for (int p = 0; p < npairs; ++p)
      for (int q = k1->node_off[pa[p]]; q < k1->node_off[pa[p] + 1]; ++q) {
        const int idx1 = k1->node_feat[q];
        if (matches12[idx1] < 0) continue;
        float rot = k1->kp_angle[idx1] - k2->kp_angle[matches12[idx1]];
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)roundf(rot * factor);
        if (bin == 30) bin = 0;
        if (bin >= 0 && bin < 30) hist[bin].push_back(idx1);
      }
    int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
    for (int i = 0; i < 30; ++i) {
      const int sz = (int)hist[i].size();
      if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; i3 = i2; i2 = i1; i1 = i; }
      else if (sz > max2) { max3 = max2; max2 = sz; i3 = i2; i2 = i; }
      else if (sz > max3) { max3 = sz; i3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { i2 = -1; i3 = -1; }
    else if (max3 < 0.1f * (float)max1) { i3 = -1; }
    for (int i = 0; i < 30; ++i) {
      if (i == i1 || i == i2 || i == i3) continue;
      for (int idx1 : hist[i]) { matches12[idx1] = -1; --nmatches; }
    }
  }
  *out_nmatches = nmatches;
  return RGBL_OK;
}

}  // extern "C"
