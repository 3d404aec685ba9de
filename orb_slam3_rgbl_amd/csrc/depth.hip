// depth.hip — LiDAR depth path: projection + ordered scatter, up-sampling, per-keypoint depth gather.
//
// Replaces ORB_SLAM3::DepthModule::CalculateDepthFromPcd and the functions it calls
// (/root/reference/src/DepthModule.cc:50-274, include/DepthModule.h:34-161).
//
// Kernels (B scans per launch, blockIdx.y = frame):
//   k_project_index   thread per point: u,v,d with the reference's arithmetic (fp64-accumulated 3x4 dot,
//                     fp32 reciprocal and product); atomicMax(point index) per pixel = "last point wins"
//   k_project_write   thread per point: the winner of a pixel writes its depth into RawDepthMap
//   k_inverse_dilate  64x16 tile + halo in LDS: S-x, TOZERO_INV, max over the structuring element, S-x, TOZERO_INV
//   k_average_filter  k x k box (reflect-101) of depths and of the hit count, ratio as the reference forms it
//   k_gather_depth    mvDepth / mvuRight from ProcessedDepthMap (truncating index)
//   k_nn_depth        NearestNeighborPixel: 5x5-chamfer distance to the nearest hit, max depth in the box
// HBM layout per frame: idx map (u32 h*w) | raw map (f32 h*w) | processed map (f32 h*w).
#include <float.h>
#include <math.h>
#include <string.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "common.h"

namespace rgbl {

struct ProjParams { float m[12]; float min_dist, max_dist; };
struct DilateMask { int kw, kh; uint8_t m[81]; };

// DepthModule.cc:115-119 for one point. Returns the pixel index or -1.
// kXyzi: the scan is still in the KITTI velodyne .bin layout (x, y, z, reflectance per point, one 16-byte load);
// the reference's loader drops the reflectance and sets the homogeneous coordinate to 1 (rgbl_kitti.cc:151-185).
struct CloudPoint { float x, y, z, o; };
template <bool kXyzi>
__device__ __forceinline__ CloudPoint load_point(const float* __restrict__ cloud, int ld, int i) {
  CloudPoint c;
  if (kXyzi) {
    const float4 q = reinterpret_cast<const float4*>(cloud)[i];
    c.x = q.x; c.y = q.y; c.z = q.z; c.o = 1.0f;
  } else {
    c.x = cloud[i]; c.y = cloud[(size_t)ld + i]; c.z = cloud[2 * (size_t)ld + i]; c.o = cloud[3 * (size_t)ld + i];
  }
  return c;
}

__device__ __forceinline__ int project_loaded(const ProjParams& P, const CloudPoint& c, int w, int h, float* depth) {
  const double x = (double)c.x, y = (double)c.y, z = (double)c.z, o = (double)c.o;
  float p[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    // OpenCV's generic GEMM accumulates every dot product in double, k = 0..3, and rounds once.  A product of two floats
    // is exact in double, so fma(m, v, acc) rounds exactly what acc + m * v rounds: same bits, half the fp64 instructions.
    double acc = fma((double)P.m[4 * r + 0], x, 0.0);  // 0.0 + product as the reference forms it (the sign of a zero sum included)
    acc = fma((double)P.m[4 * r + 1], y, acc);
    acc = fma((double)P.m[4 * r + 2], z, acc);
    acc = fma((double)P.m[4 * r + 3], o, acc);
    p[r] = (float)acc;
  }
  const float recip = __fdiv_rn(1.0f, p[2]);
  const float u = p[0] * recip, v = p[1] * recip, d = p[2];
  *depth = d;
  if (u > 0 && v > 0 && u < (float)w && v < (float)h && d > P.min_dist && d < P.max_dist) return (int)v * w + (int)u;
  return -1;
}

template <bool kXyzi>
__device__ __forceinline__ int project_point(const ProjParams& P, const float* __restrict__ cloud, int ld, int i, int w,
                                             int h, float* depth) {
  return project_loaded(P, load_point<kXyzi>(cloud, ld, i), w, h, depth);
}

// index map entry = generation << kIdxBits | point index + 1: the indexed path does not clear the map between calls
constexpr int kIdxBits = 20;
constexpr int kProjectPts = 4;  // points per work-item of k_project_index
constexpr uint32_t kIdxMask = (1u << kIdxBits) - 1u;

template <bool kXyzi>
__global__ __launch_bounds__(256) void k_project_index(ProjParams P, const float* __restrict__ cloud, size_t cloud_stride,
                                                       int n, int ld, int w, int h, uint32_t* __restrict__ idx_map,
                                                       size_t map_stride, float* __restrict__ pt_depth, size_t pt_stride,
                                                       uint32_t tag) {
  // kProjectPts points per work-item, all of them requested before the first is used
  const int f = xcd_frame();
  const float* C = cloud + (size_t)f * cloud_stride;
  const int i0 = xcd_item() * (256 * kProjectPts) + threadIdx.x;
  CloudPoint c[kProjectPts];
#pragma unroll
  for (int j = 0; j < kProjectPts; ++j) c[j] = load_point<kXyzi>(C, ld, imin(i0 + 256 * j, n - 1));
#pragma unroll
  for (int j = 0; j < kProjectPts; ++j) {
    const int i = i0 + 256 * j;
    if (i >= n) break;
    float d;
    const int pix = project_loaded(P, c[j], w, h, &d);
    // the last point index wins (DepthModule.cc:123-137 scatters in order); tag = this call's generation in the bits above
    // kIdxBits, larger than whatever earlier calls left in the map (0 when the map was cleared instead)
    if (pix >= 0) atomicMax(idx_map + (size_t)f * map_stride + pix, tag | (uint32_t)(i + 1));
    if (pt_depth) pt_depth[(size_t)f * pt_stride + i] = d;  // read back through the index map by k_inverse_dilate<., true>
  }
}

template <bool kXyzi>
__global__ __launch_bounds__(256) void k_project_write(ProjParams P, const float* __restrict__ cloud, size_t cloud_stride,
                                                       int n, int ld, int w, int h, const uint32_t* __restrict__ idx_map,
                                                       float* __restrict__ raw, size_t map_stride) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int f = blockIdx.y;
  float d;
  const int pix = project_point<kXyzi>(P, cloud + (size_t)f * cloud_stride, ld, i, w, h, &d);
  if (pix >= 0 && idx_map[(size_t)f * map_stride + pix] == (uint32_t)(i + 1)) raw[(size_t)f * map_stride + pix] = d;
}

// Inverse dilation (DepthModule.cc:265-273): P = S - raw, TOZERO_INV(S-1), dilate, S - P, TOZERO_INV(S-1).
// 64x16 output tile per workgroup; the inverted tile + halo is staged in LDS.
// kRadius > 0: the reference's Diamond structuring element |dx|+|dy| <= kRadius with compile-time taps (all
// LDS reads of a pixel are issued back to back); kRadius == 0: arbitrary mask, tap offsets held in LDS.
// grid = (ceil(w/64), ceil(h/32), B), block = 256.
// kIndexed: the raw depth map is never written; a pixel's raw depth is looked up as pt_depth[idx_map[pixel] - 1]
// (0 where no point fell), which saves the map's zero fill, k_project_write and one map-sized round trip through HBM.
template <int kRadius, bool kIndexed>
__global__ __launch_bounds__(256) void k_inverse_dilate(DilateMask K, float S, const float* __restrict__ raw,
                                                        const uint32_t* __restrict__ idx_map, const float* __restrict__ pt_depth,
                                                        size_t pt_stride, uint32_t tag, float* __restrict__ out, size_t map_stride,
                                                        int w, int h) {
  constexpr int kTileH = 32;  // 64 x 32 outputs per workgroup: (68 x 36) / 2048 = 1.20 tile elements read per output at radius 2 (1.33 with 16 rows)
  __shared__ float s_inv[(kTileH + 8) * 72];
  __shared__ int s_tap[81];
  __shared__ int s_ntap;
  // grid = xcd_grid(tiles of a frame, B) (common.h): an XCD's L2 serves whole frames, tile halos are re-read from it
  const int tid = threadIdx.x, f = xcd_frame();
  const int tiles_x = (w + 63) >> 6;
  const int tile_y = xcd_item() / tiles_x, tile_x = xcd_item() - tile_y * tiles_x;
  const int x0 = tile_x * 64, y0 = tile_y * kTileH;
  const int ax = kRadius > 0 ? kRadius : K.kw / 2, ay = kRadius > 0 ? kRadius : K.kh / 2;
  const int tw = kRadius > 0 ? 64 + 2 * kRadius : 64 + K.kw - 1, th = kRadius > 0 ? kTileH + 2 * kRadius : kTileH + K.kh - 1;
  const float thr = S - 1;
  const float* R = kIndexed ? nullptr : raw + (size_t)f * map_stride;
  const uint32_t* I = kIndexed ? idx_map + (size_t)f * map_stride : nullptr;
  const float* D = kIndexed ? pt_depth + (size_t)f * pt_stride : nullptr;
  if (kRadius == 0 && tid < 64) {
    // compact the mask (<= 81 entries) into a tap list with two ballots of wave 0
    const int n = K.kw * K.kh, i1 = tid + 64;
    const bool on0 = tid < n && K.m[tid] != 0, on1 = i1 < n && K.m[i1 < 81 ? i1 : 0] != 0;
    const unsigned long long m0 = __ballot(on0), m1 = __ballot(on1);
    const unsigned long long lt = lanemask_lt();
    if (on0) s_tap[__popcll(m0 & lt)] = (tid / K.kw) * 72 + (tid % K.kw);
    if (on1) s_tap[__popcll(m0) + __popcll(m1 & lt)] = (i1 / K.kw) * 72 + (i1 % K.kw);
    if (tid == 0) s_ntap = __popcll(m0) + __popcll(m1);
  }
  // raw depth of tile element i (0 = empty pixel); taps outside the image never win (cv::dilate's default border)
  auto inverted = [&](bool inside, float r) {
    const float t = S - r;
    return inside ? (t > thr ? 0.f : t) : -FLT_MAX;  // THRESH_TOZERO_INV
  };
  if (kRadius > 0) {
    // compile-time tile (64 + 2 r) x (32 + 2 r): a work-item fetches column (tid & 63) of every fourth row, the first
    // 2 r (32 + 2 r) work-items also one element of the 2 r columns to the right of those - no divisions by the tile width, one 32-bit index per element, every load of a
    // work-item requested before the first is used; the indexed variant then requests all its depth look-ups together
    constexpr int kTw = 64 + 2 * kRadius, kTh = kTileH + 2 * kRadius;
    static_assert(kTw <= 72 && kTh <= kTileH + 8, "tile inside s_inv");
    constexpr int kMain = (kTh + 3) / 4;                       // rows per work-item in the 64 main columns
    constexpr int kSideElems = 2 * kRadius * kTh;              // the columns 64 .. kTw - 1
    constexpr int kSide = (kSideElems + 255) / 256, kIter = kMain + kSide;
    const int r0 = tid >> 6, c0 = tid & 63;
    // Two instantiations of the staging: tiles that lie inside the image with their halo (three quarters of a KITTI frame's)
    // know every element to be inside - no bounds tests, one base index plus scalar row offsets.
    auto stage = [&](auto interior_c) {
      constexpr bool kInterior = decltype(interior_c)::value;
      bool inside[kIter];
      int at[kIter], lds[kIter];
      float raw_v[kIter];
      uint32_t id[kIter];
      const int at0 = (int)__umul24((uint32_t)(kInterior ? y0 - kRadius + r0 : 0), (uint32_t)w) + x0 - kRadius + c0;
#pragma unroll
      for (int j = 0; j < kIter; ++j) {
        constexpr int kSideW = kRadius > 0 ? 2 * kRadius : 1;  // (this block is not reached with kRadius == 0)
        const int si = tid + 256 * (j - kMain), sr = si / kSideW;   // element si of the side columns (rows of 2 r)
        const int r = j < kMain ? r0 + 4 * j : sr, c = j < kMain ? c0 : 64 + (si - sr * kSideW);
        const bool live = j < kMain ? (4 * j + 3 < kTh || r < kTh) : si < kSideElems;
        const int yy = y0 + r - kRadius, xx = x0 + c - kRadius;
        inside[j] = live && (kInterior || (yy >= 0 && yy < h && xx >= 0 && xx < w));
        // rows and widths far below 2^24: full-rate multiplies; interior main columns: base + (scalar) 4 j w
        at[j] = !inside[j] ? 0 : (kInterior && j < kMain) ? at0 + 4 * j * w : (int)__umul24((uint32_t)yy, (uint32_t)w) + xx;
        lds[j] = live ? r * 72 + c : -1;
        if (kIndexed) id[j] = inside[j] ? I[at[j]] : 0u;
        else raw_v[j] = inside[j] ? R[at[j]] : 0.f;
      }
      if (kIndexed) {
#pragma unroll
        for (int j = 0; j < kIter; ++j) {
          const uint32_t pt = id[j] & kIdxMask;
          const bool hit = (id[j] ^ tag) == pt && pt;  // tag | index: entries of earlier generations (another tag) are empty pixels
          raw_v[j] = hit ? D[pt - 1] : 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < kIter; ++j)
        if (lds[j] >= 0) s_inv[lds[j]] = inverted(inside[j], raw_v[j]);
    };
    if (x0 >= kRadius && y0 >= kRadius && x0 + 64 + kRadius <= w && y0 + kTileH + kRadius <= h) stage(std::true_type{});
    else stage(std::false_type{});
  } else {
    for (int i = tid; i < tw * th; i += 256) {
      const int r = i / tw, c = i - r * tw;
      const int yy = y0 + r - ay, xx = x0 + c - ax;
      const bool in = yy >= 0 && yy < h && xx >= 0 && xx < w;
      float rv = 0.f;
      if (in) {
        if (kIndexed) {
          const uint32_t e = I[(size_t)yy * w + xx];
          rv = (e & ~kIdxMask) == tag && (e & kIdxMask) ? D[(e & kIdxMask) - 1] : 0.f;
        } else {
          rv = R[(size_t)yy * w + xx];
        }
      }
      s_inv[r * 72 + c] = inverted(in, rv);
    }
  }
  __syncthreads();
  const int x = tid & 63;
  if (kRadius > 0) {
    // Diamond |dx| + |dy| <= r: a work-item owns 8 consecutive rows of one column.  Per tile row it reads the 2 r + 1 values
    // around its column once and nests their maxima, H[k] = max over |dx| <= k; an output is then the maximum of
    // H[r - |dy|] over the rows dy = -r .. r.  (13 LDS reads and 13 maxima per output at r = 2 before, 7.5 reads and 5
    // three-input maxima now; the maximum is exact whatever the order.)
    constexpr int kRows = 8, R = kRadius;
    const int yb = (tid >> 6) * kRows;            // first output row of this work-item inside the tile
    float H[kRows + 2 * R][R + 1];
#pragma unroll
    for (int r = 0; r < kRows + 2 * R; ++r) {
      const float* row = &s_inv[(yb + r) * 72 + x];   // tile columns x .. x + 2 r, the output's own column at x + r
      float v[2 * R + 1];
#pragma unroll
      for (int c = 0; c <= 2 * R; ++c) v[c] = row[c];
      H[r][0] = v[R];
#pragma unroll
      for (int k = 1; k <= R; ++k) H[r][k] = fmaxf(fmaxf(H[r][k - 1], v[R - k]), v[R + k]);
    }
    if (x0 + x < w) {
      float* O = out + (size_t)f * map_stride + (__umul24((uint32_t)(y0 + yb), (uint32_t)w) + (uint32_t)(x0 + x));
#pragma unroll
      for (int j = 0; j < kRows; ++j) {
        const int y = yb + j;
        if (y0 + y >= h) break;
        float m = H[j + R][R];
#pragma unroll
        for (int d = 1; d <= R; ++d) m = fmaxf(fmaxf(m, H[j + R - d][R - d]), H[j + R + d][R - d]);
        const float t = S - m;
        O[(uint32_t)(j * w)] = t > thr ? 0.f : t;   // j * w: scalar
      }
    }
  } else {
    const int nt = s_ntap;
    for (int k = 0; k < kTileH / 4; ++k) {
      const int y = (tid >> 6) + 4 * k;
      if (x0 + x >= w || y0 + y >= h) continue;
      float m = -FLT_MAX;
      const float* base = &s_inv[y * 72 + x];
      for (int t = 0; t < nt; ++t) m = fmaxf(m, base[s_tap[t]]);
      const float t = S - m;
      out[(size_t)f * map_stride + (size_t)(y0 + y) * w + x0 + x] = t > thr ? 0.f : t;
    }
  }
}

// grid = (ceil(w/64), ceil(h/16), B), block = 256.  ksize <= 9.
__global__ __launch_bounds__(256) void k_average_filter(int ksize, const float* __restrict__ raw, float* __restrict__ out,
                                                        size_t map_stride, int w, int h) {
  __shared__ float s_in[24 * 72];
  const int tid = threadIdx.x, f = blockIdx.z;
  const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 16;
  const int a = ksize / 2;
  const int tw = 64 + ksize - 1, th = 16 + ksize - 1;
  const float* R = raw + (size_t)f * map_stride;
  for (int i = tid; i < tw * th; i += 256) {
    const int r = i / tw, c = i - r * tw;
    s_in[r * 72 + c] = R[(size_t)reflect101(y0 + r - a, h) * w + reflect101(x0 + c - a, w)];
  }
  __syncthreads();
  const float coef = (float)(1.0 / (double)(ksize * ksize));
  const int x = tid & 63;
  for (int k = 0; k < 4; ++k) {
    const int y = (tid >> 6) + 4 * k;
    if (x0 + x >= w || y0 + y >= h) continue;
    float sum = 0.f, cnt = 0.f;
    for (int ky = 0; ky < ksize; ++ky)
      for (int kx = 0; kx < ksize; ++kx) {
        const float v = s_in[(y + ky) * 72 + x + kx];
        sum = sum + coef * v;
        cnt = cnt + (v > 0.f ? 1.f : 0.f);
      }
    out[(size_t)f * map_stride + (size_t)(y0 + y) * w + x0 + x] = sum * __fdiv_rn((float)(ksize * ksize), cnt);
  }
}

// DepthModule.cc:82-104.  Keypoints: x at kp[i*stride], y at kp[i*stride+1]; undistorted x at kpun[i*un_stride].
__global__ __launch_bounds__(256) void k_gather_depth(const float* __restrict__ map, size_t map_stride, int w,
                                                      const float* __restrict__ kp, int kp_stride, size_t kp_frame,
                                                      const float* __restrict__ kpun, int un_stride, size_t un_frame,
                                                      const int32_t* __restrict__ n_per_frame, int n_fixed, float mbf,
                                                      float* __restrict__ depth, float* __restrict__ uright, size_t out_frame) {
  const int i = xcd_item() * 256 + threadIdx.x, f = xcd_frame();  // grid = xcd_grid(keypoint groups, B)
  const int n = n_per_frame ? n_per_frame[f] : n_fixed;
  if (i >= n) return;
  const float u = kp[f * kp_frame + (size_t)i * kp_stride], v = kp[f * kp_frame + (size_t)i * kp_stride + 1];
  const float d = map[(size_t)f * map_stride + (size_t)(int)v * w + (int)u];
  float od = -1.f, our = -1.f;
  if (d > 0) {
    od = d;
    our = kpun[f * un_frame + (size_t)i * un_stride] - __fdiv_rn(mbf, d);
  }
  depth[f * out_frame + i] = od;
  uright[f * out_frame + i] = our;
}

// DepthModule.cc:203-251 evaluated at the keypoints only (rgbl_depth_set_sparse): the value k_inverse_dilate would write at the
// keypoint's pixel - same taps, same THRESH_TOZERO_INV arithmetic, the maximum is exact in any order - read through the index
// map, then DepthModule.cc:82-104 as k_gather_depth.  One work-item per keypoint, <= 81 taps (13 for the shipped Diamond 5).
__global__ __launch_bounds__(256) void k_gather_depth_sparse(DilateMask K, float S, const uint32_t* __restrict__ idx_map,
                                                             const float* __restrict__ pt_depth, size_t pt_stride, uint32_t tag,
                                                             size_t map_stride, int w, int h,
                                                             const float* __restrict__ kp, int kp_stride, size_t kp_frame,
                                                             const float* __restrict__ kpun, int un_stride, size_t un_frame,
                                                             const int32_t* __restrict__ n_per_frame, int n_fixed, float mbf,
                                                             float* __restrict__ depth, float* __restrict__ uright, size_t out_frame) {
  const int i = xcd_item() * 256 + threadIdx.x, f = xcd_frame();  // grid = xcd_grid(keypoint groups, B)
  const int n = n_per_frame ? n_per_frame[f] : n_fixed;
  if (i >= n) return;
  const float u = kp[f * kp_frame + (size_t)i * kp_stride], v = kp[f * kp_frame + (size_t)i * kp_stride + 1];
  const uint32_t* I = idx_map + (size_t)f * map_stride;
  const float* D = pt_depth + (size_t)f * pt_stride;
  const int px = (int)u - K.kw / 2, py = (int)v - K.kh / 2;
  const float thr = S - 1;
  float m = -FLT_MAX;  // taps outside the image never win (cv::dilate's default border)
  for (int ky = 0; ky < K.kh; ++ky) {
    const int yy = py + ky;
    if (yy < 0 || yy >= h) continue;
    for (int kx = 0; kx < K.kw; ++kx) {
      const int xx = px + kx;
      if (!K.m[ky * K.kw + kx] || xx < 0 || xx >= w) continue;
      const uint32_t e = I[(size_t)yy * w + xx], pt = e & kIdxMask;
      const float r = ((e ^ tag) == pt && pt) ? D[pt - 1] : 0.f;
      const float t = S - r;
      m = fmaxf(m, t > thr ? 0.f : t);
    }
  }
  const float t = S - m;
  const float d = t > thr ? 0.f : t;
  float od = -1.f, our = -1.f;
  if (d > 0) {
    od = d;
    our = kpun[f * un_frame + (size_t)i * un_stride] - __fdiv_rn(mbf, d);
  }
  depth[f * out_frame + i] = od;
  uright[f * out_frame + i] = our;
}

// DepthModule.cc:145-198.  The reference runs cv::distanceTransform(DIST_L2, 5x5) over the whole map and then
// looks at one value per keypoint; the 5x5 chamfer metric has a closed form per displacement, so the distance
// at a keypoint is the minimum of that form over the hits inside a small window (16.16 fixed point, exact).
__global__ __launch_bounds__(256) void k_nn_depth(const float* __restrict__ raw, size_t map_stride, int w, int h,
                                                  const float* __restrict__ kp, int kp_stride, size_t kp_frame,
                                                  const float* __restrict__ kpun, int un_stride, size_t un_frame,
                                                  const int32_t* __restrict__ n_per_frame, int n_fixed, float mbf,
                                                  float radius, float* __restrict__ depth, float* __restrict__ uright,
                                                  size_t out_frame) {
  const int i = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
  const int n = n_per_frame ? n_per_frame[f] : n_fixed;
  if (i >= n) return;
  const float u = kp[f * kp_frame + (size_t)i * kp_stride], v = kp[f * kp_frame + (size_t)i * kp_stride + 1];
  const float* R = raw + (size_t)f * map_stride;
  const int px = (int)u, py = (int)v;
  const unsigned HV = 65536u, DG = 91750u, LG = 143976u;  // cvRound({1, 1.4, 2.1969} * 2^16)
  const int win = (int)radius + 2;
  unsigned best = 0x1fffffffu;  // INT_MAX >> 2, the transform's "infinity"
  for (int dy = -win; dy <= win; ++dy) {
    const int yy = py + dy;
    if (yy < 0 || yy >= h) continue;
    for (int dx = -win; dx <= win; ++dx) {
      const int xx = px + dx;
      if (xx < 0 || xx >= w) continue;
      if (__float2int_rn(R[(size_t)yy * w + xx]) <= 0) continue;  // convertTo(CV_8U) + THRESH_BINARY_INV
      int a = dx < 0 ? -dx : dx, b = dy < 0 ? -dy : dy;
      if (a < b) { const int t = a; a = b; b = t; }
      const unsigned cost = (2 * b <= a) ? (unsigned)b * LG + (unsigned)(a - 2 * b) * HV
                                         : (unsigned)(a - b) * LG + (unsigned)(2 * b - a) * DG;
      best = cost < best ? cost : best;
    }
  }
  int sr = (int)((float)best * (1.f / 65536));
  float d = 0.f;
  if (sr >= 0 && (float)sr < radius) {
    ++sr;
    const int pad = (int)radius;
    const int bx = (int)(u + radius - (float)sr) - pad, by = (int)(v + radius - (float)sr) - pad;  // un-padded coordinates
    float mx = -FLT_MAX;
    for (int yy = by; yy < by + 2 * sr; ++yy)
      for (int xx = bx; xx < bx + 2 * sr; ++xx) {
        const float val = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? R[(size_t)yy * w + xx] : 0.f;
        mx = fmaxf(mx, val);
      }
    d = mx;
  }
  float od = -1.f, our = -1.f;
  if (d > 0) {
    od = d;
    our = kpun[f * un_frame + (size_t)i * un_stride] - __fdiv_rn(mbf, d);
  }
  depth[f * out_frame + i] = od;
  uright[f * out_frame + i] = our;
}

}  // namespace rgbl

using namespace rgbl;

struct rgbl_depth {
  rgbl_depth_cfg cfg;
  int device = 0;
  hipStream_t stream = nullptr, own_stream = nullptr;
  KernelTimer timer;
  ProjParams proj;
  DilateMask mask;
  int diamond_radius = 0;  // > 0 when the mask is the reference's Diamond of that radius (compile-time taps)
  size_t map_stride = 0;
  uint32_t* d_idx = nullptr;  // idx | raw contiguous so one memset clears both
  float* d_raw = nullptr;
  bool xcd_map = true;          // XCD-aware launch geometry (common.h: xcd_grid); RGBL_XCD_MAP=0 switches it off
  uint32_t idx_gen = 0;        // generation of the index maps' newest entries; 0 = cleared / untagged content
  uint32_t max_gen = (1u << (32 - 20)) - 1u;  // RGBL_DEPTH_MAX_GEN lowers it (tests of the wrap-around)
  float* d_ptdepth = nullptr;  // depth of every projected point (max_batch x max_points), for the raw-map-free path
  float* d_proc = nullptr;
  float* d_cloud = nullptr;
  float *d_kp = nullptr, *d_kpun = nullptr, *d_depth = nullptr, *d_uright = nullptr;  // one block: kp (2 K) | kpun (K) | depth (K) | uright (K)
  int last_k = 0;              // keypoints of the last host-pointer call: their mvuRight is still in d_uright (rgbl_device_frame_capture)
  float* h_cloud = nullptr;    // page-locked staging of one scan (rgbl_depth_prefetch), allocated on first use
  float* h_kio = nullptr;      // page-locked mirror of that block: the keypoint arrays of the host entry points travel in one request each way
  std::vector<void*> allocs;
  // rgbl_depth_prefetch: the maps of this cloud are queued (or done) on the handle's stream
  struct { bool active = false; const float* cloud = nullptr; int n = 0, ld = 0; bool xyzi = false; } prefetched;
  // rgbl_depth_set_sparse: the dense ProcessedDepthMap is written only when a caller asks for it; `maps` describes what the
  // last projection left behind (dense == false: index maps of generation `tag`, the gather evaluates the dilation per keypoint)
  bool sparse = false;
  struct { bool dense = true; bool indexed = false; uint32_t tag = 0; int batch = 0; } maps;
};

namespace {
template <class T>
int dalloc(rgbl_depth* e, T** p, size_t count) {
  RGBL_HIP(hipMalloc(p, std::max<size_t>(count, 1) * sizeof(T)));
  e->allocs.push_back((void*)*p);
  return RGBL_OK;
}

// DepthModule.cc:203-251 / 253-300: the dense ProcessedDepthMap of the maps the last projection wrote.
int enqueue_upsample(rgbl_depth* e, int batch, int w, int h) {
  hipStream_t s = e->stream;
  const size_t ms = e->map_stride, pt_stride = (size_t)e->cfg.max_points;
  const bool indexed = e->maps.indexed;
  const uint32_t tag = e->maps.tag;
  const dim3 tiles((w + 63) / 64, (h + 15) / 16, batch);
  const dim3 dilate_tiles = xcd_grid(e->xcd_map, ((w + 63) / 64) * ((h + 31) / 32), batch);  // k_inverse_dilate works on 64 x 32 tiles
  switch (e->cfg.method) {
    case RGBL_UPS_INVERSE_DILATION:
      e->timer.begin("k_inverse_dilate", s);
      // opt_max_dist * ParamUpsampling_InverseDilation_ScaleFactor; the scale factor is never parsed (1.0)
#define RGBL_DILATE(R, I) hipLaunchKernelGGL((k_inverse_dilate<R, I>), dilate_tiles, dim3(256), 0, s, e->mask, e->cfg.max_dist * 1.0f, e->d_raw, \
                                            e->d_idx, e->d_ptdepth, pt_stride, tag, e->d_proc, ms, w, h)
      if (indexed) {
        switch (e->diamond_radius) {
          case 1: RGBL_DILATE(1, true); break;
          case 2: RGBL_DILATE(2, true); break;
          case 3: RGBL_DILATE(3, true); break;
          case 4: RGBL_DILATE(4, true); break;
          default: RGBL_DILATE(0, true); break;
        }
      } else {
        switch (e->diamond_radius) {
          case 1: RGBL_DILATE(1, false); break;
          case 2: RGBL_DILATE(2, false); break;
          case 3: RGBL_DILATE(3, false); break;
          case 4: RGBL_DILATE(4, false); break;
          default: RGBL_DILATE(0, false); break;
        }
      }
#undef RGBL_DILATE
      e->timer.end(s);
      break;
    case RGBL_UPS_AVERAGE_FILTERING:
      e->timer.begin("k_average_filter", s);
      hipLaunchKernelGGL(k_average_filter, tiles, dim3(256), 0, s, e->cfg.avg_kernel_size, e->d_raw, e->d_proc, ms, w, h);
      e->timer.end(s);
      break;
    default:
      break;
  }
  e->maps.dense = true;
  RGBL_HIP(hipGetLastError());
  return RGBL_OK;
}

// Part 1 (independent of the keypoints): projection + ordered scatter + dense up-sampling.
int enqueue_maps(rgbl_depth* e, const float* d_cloud, int batch, int n, int ld, size_t cloud_stride, int w, int h,
                 float* d_processed_out, bool xyzi = false, bool need_raw = true, bool want_dense = true) {
  hipStream_t s = e->stream;
  const size_t ms = e->map_stride;
  // Whatever a prefetch left in the maps is overwritten from here on: EVERY entry point that projects - the host ones, the
  // batch-device ones - comes through this function, so a later rgbl_depth_compute with the prefetched cloud's pointer
  // projects again instead of gathering from another scan's maps.  depth_prefetch_host re-arms it behind its own call.
  e->prefetched.active = false;
  // the inverse dilation can read the points' depths through the index map: no raw map unless the caller wants it
  // (device batches may bring more points per scan than the host staging size the per-point buffer was sized for)
  const bool indexed = e->cfg.method == RGBL_UPS_INVERSE_DILATION && !need_raw && n <= e->cfg.max_points && (uint32_t)n < kIdxMask;
  float* pt_depth = indexed ? e->d_ptdepth : nullptr;
  const size_t pt_stride = (size_t)e->cfg.max_points;
  // idx maps (max_batch of them) are followed by the raw maps: one memset clears both when the batch is full
  uint32_t tag = 0;
  if (indexed) {
    // every call stamps its entries with a new generation, so the maps are cleared once per max_gen calls only
    if (e->idx_gen == 0 || e->idx_gen >= e->max_gen) {
      RGBL_HIP(hipMemsetAsync(e->d_idx, 0, (size_t)e->cfg.max_batch * ms * sizeof(uint32_t), s));
      e->idx_gen = 1;
    } else {
      ++e->idx_gen;
    }
    tag = e->idx_gen << kIdxBits;
  } else if (batch == e->cfg.max_batch) {
    e->idx_gen = 0;  // untagged entries: the next indexed call starts from cleared maps
    RGBL_HIP(hipMemsetAsync(e->d_idx, 0, (size_t)batch * ms * 2 * sizeof(uint32_t), s));
  } else {
    e->idx_gen = 0;
    RGBL_HIP(hipMemsetAsync(e->d_idx, 0, (size_t)batch * ms * sizeof(uint32_t), s));
    RGBL_HIP(hipMemsetAsync(e->d_raw, 0, (size_t)batch * ms * sizeof(float), s));
  }
  if (n > 0) {
    e->timer.begin("k_project_index", s);
    const dim3 pgrid((n + 255) / 256, batch), igrid = xcd_grid(e->xcd_map, (n + 256 * kProjectPts - 1) / (256 * kProjectPts), batch);
    if (xyzi) hipLaunchKernelGGL(k_project_index<true>, igrid, dim3(256), 0, s, e->proj, d_cloud, cloud_stride, n, ld, w, h, e->d_idx, ms, pt_depth, pt_stride, tag);
    else hipLaunchKernelGGL(k_project_index<false>, igrid, dim3(256), 0, s, e->proj, d_cloud, cloud_stride, n, ld, w, h, e->d_idx, ms, pt_depth, pt_stride, tag);
    e->timer.end(s);
    if (!indexed) {
      e->timer.begin("k_project_write", s);
      if (xyzi) hipLaunchKernelGGL(k_project_write<true>, pgrid, dim3(256), 0, s, e->proj, d_cloud, cloud_stride, n, ld, w, h, e->d_idx, e->d_raw, ms);
      else hipLaunchKernelGGL(k_project_write<false>, pgrid, dim3(256), 0, s, e->proj, d_cloud, cloud_stride, n, ld, w, h, e->d_idx, e->d_raw, ms);
      e->timer.end(s);
    }
  }
  e->maps.indexed = indexed; e->maps.tag = tag; e->maps.batch = batch;
  // sparse handles keep the index maps and leave the dilation to the gather, unless the dense map is asked for here
  e->maps.dense = !(e->sparse && indexed && !want_dense);
  if (e->maps.dense) RGBL_TRY(enqueue_upsample(e, batch, w, h));
  if (d_processed_out && e->cfg.method != RGBL_UPS_NEAREST_NEIGHBOR_PIXEL)
    RGBL_HIP(hipMemcpyAsync(d_processed_out, e->d_proc, (size_t)batch * ms * sizeof(float), hipMemcpyDeviceToDevice, s));
  RGBL_HIP(hipGetLastError());
  return RGBL_OK;
}

// Part 2: depth / virtual right coordinate of every keypoint. kp / kpun are strided float views.
int enqueue_keypoints(rgbl_depth* e, int batch, int w, int h, const float* kp, int kp_stride, size_t kp_frame,
                      const float* kpun, int un_stride, size_t un_frame, const int32_t* d_n, int n_fixed, int kmax,
                      float* d_depth, float* d_uright, size_t out_frame) {
  if (kmax <= 0) return RGBL_OK;
  hipStream_t s = e->stream;
  const size_t ms = e->map_stride;
  const dim3 kgrid((kmax + 255) / 256, batch);
  if (e->cfg.method == RGBL_UPS_NEAREST_NEIGHBOR_PIXEL) {
    e->timer.begin("k_nn_depth", s);
    hipLaunchKernelGGL(k_nn_depth, kgrid, dim3(256), 0, s, e->d_raw, ms, w, h, kp, kp_stride, kp_frame, kpun, un_stride,
                       un_frame, d_n, n_fixed, e->cfg.mbf, e->cfg.nn_search_radius, d_depth, d_uright, out_frame);
    e->timer.end(s);
  } else if (!e->maps.dense) {
    e->timer.begin("k_gather_depth_sparse", s);
    hipLaunchKernelGGL(k_gather_depth_sparse, xcd_grid(e->xcd_map, (kmax + 255) / 256, batch), dim3(256), 0, s, e->mask, e->cfg.max_dist * 1.0f,
                       e->d_idx, e->d_ptdepth, (size_t)e->cfg.max_points, e->maps.tag, ms, w, h, kp, kp_stride, kp_frame, kpun,
                       un_stride, un_frame, d_n, n_fixed, e->cfg.mbf, d_depth, d_uright, out_frame);
    e->timer.end(s);
  } else {
    e->timer.begin("k_gather_depth", s);
    hipLaunchKernelGGL(k_gather_depth, xcd_grid(e->xcd_map, (kmax + 255) / 256, batch), dim3(256), 0, s, e->d_proc, ms, w, kp, kp_stride, kp_frame, kpun, un_stride,
                       un_frame, d_n, n_fixed, e->cfg.mbf, d_depth, d_uright, out_frame);
    e->timer.end(s);
  }
  RGBL_HIP(hipGetLastError());
  return RGBL_OK;
}
}  // namespace

extern "C" {

void rgbl_projection_matrix(const float K[12], const float Tr[16], float out[12]) {
  // cv::Mat product CameraMatrix(3x4) * RotationMatrix(4x4), DepthModule.cc:434.  cv::gemm takes its small-matrix
  // special case for this shape (inner length 4 == output width): float temporaries, one expression
  // a0*b0 + a1*b1 + a2*b2 + a3*b3, contracted to an fma chain by the dispatched AVX2 / AVX-512 / NEON builds
  // every FMA-capable host runs.  (The generic double-accumulating GEMM is what the 3x4 * 4xN projection uses.)
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) {
      float t = K[4 * r + 0] * Tr[c];
      for (int k = 1; k < 4; ++k) t = fmaf(K[4 * r + k], Tr[4 * k + c], t);
      out[4 * r + c] = t;
    }
}

int rgbl_structuring_element(int shape, int kw, int kh, uint8_t* out) {
  if (!out || kw < 1 || kh < 1 || kw > 9 || kh > 9) { set_error("structuring element must be 1..9 wide/high"); return RGBL_ERR_INVALID; }
  if (shape == 3) {
    // DiamondKernelData_{3,5,7,9} (DepthModule.h:138-161): |dx| + |dy| <= r; only KernelSize_u is honoured
    if (kw != 3 && kw != 5 && kw != 7 && kw != 9) { set_error("invalid kernel size for diamond kernel"); return RGBL_ERR_INVALID; }
    const int r = kw / 2;
    for (int y = 0; y < kw; ++y)
      for (int x = 0; x < kw; ++x) out[y * kw + x] = (abs(x - r) + abs(y - r) <= r) ? 1 : 0;
    return RGBL_OK;
  }
  if (shape < 0 || shape > 2) { set_error("invalid kernel type"); return RGBL_ERR_INVALID; }
  // cv::getStructuringElement(shape, Size(kw, kh)), anchor at the centre
  const int ax = kw / 2, ay = kh / 2, r = kh / 2, c = kw / 2;
  const double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
  for (int i = 0; i < kh; ++i) {
    int j1 = 0, j2 = 0;
    if (shape == 0 || (shape == 1 && i == ay)) j2 = kw;
    else if (shape == 1) { j1 = ax; j2 = j1 + 1; }
    else {
      const int dy = i - r;
      if (abs(dy) <= r) {
        const int dx = (int)lrint(c * sqrt((r * r - dy * dy) * inv_r2));
        j1 = std::max(c - dx, 0);
        j2 = std::min(c + dx + 1, kw);
      }
    }
    for (int j = 0; j < kw; ++j) out[i * kw + j] = (j >= j1 && j < j2) ? 1 : 0;
  }
  return RGBL_OK;
}

int rgbl_depth_create(const rgbl_depth_cfg* cfg, int device, rgbl_depth** out) {
  if (!cfg || !out) { set_error("null argument"); return RGBL_ERR_INVALID; }
  *out = nullptr;
  if (cfg->method == RGBL_UPS_IPBASIC || cfg->method == RGBL_UPS_NONE ||
      (cfg->method != RGBL_UPS_NEAREST_NEIGHBOR_PIXEL && cfg->method != RGBL_UPS_AVERAGE_FILTERING &&
       cfg->method != RGBL_UPS_INVERSE_DILATION)) {
    // DepthModule.cc:562-583: IPBasic / unknown methods make the parse fail and the module a no-op
    set_error("up-sampling method %d is not implemented (the reference disables the module for it too)", cfg->method);
    return RGBL_ERR_INVALID;
  }
  if (cfg->width < 1 || cfg->height < 1 || cfg->max_points < 1 || cfg->max_keypoints < 1 || cfg->max_batch < 1 ||
      cfg->kernel_w < 1 || cfg->kernel_w > 9 || cfg->kernel_h < 1 || cfg->kernel_h > 9 ||
      (cfg->method == RGBL_UPS_AVERAGE_FILTERING && (cfg->avg_kernel_size < 1 || cfg->avg_kernel_size > 9)) ||
      (cfg->method == RGBL_UPS_NEAREST_NEIGHBOR_PIXEL && !(cfg->nn_search_radius >= 1 && cfg->nn_search_radius <= 64))) {
    set_error("invalid depth configuration");
    return RGBL_ERR_INVALID;
  }
  if (rgbl_device_count() <= device || device < 0) {
    set_error("no usable HIP device %d (this library has no CPU fallback)", device);
    return RGBL_ERR_NO_DEVICE;
  }
  RGBL_HIP(hipSetDevice(device));
  rgbl_depth* e = new rgbl_depth;
  e->cfg = *cfg;
  e->device = device;
  memcpy(e->proj.m, cfg->proj, sizeof(float) * 12);
  e->proj.min_dist = cfg->min_dist;
  e->proj.max_dist = cfg->max_dist;
  e->mask.kw = cfg->kernel_w;
  e->mask.kh = cfg->kernel_h;
  memcpy(e->mask.m, cfg->kernel, 81);
  if (cfg->kernel_w == cfg->kernel_h && (cfg->kernel_w & 1) && cfg->kernel_w >= 3) {
    const int r = cfg->kernel_w / 2;
    bool diamond = true;
    for (int y = 0; y < cfg->kernel_w; ++y)
      for (int x = 0; x < cfg->kernel_w; ++x)
        diamond = diamond && ((cfg->kernel[y * cfg->kernel_w + x] != 0) == (abs(x - r) + abs(y - r) <= r));
    if (diamond) e->diamond_radius = r;
  }
  e->map_stride = (size_t)cfg->width * cfg->height;
  const size_t B = (size_t)cfg->max_batch;
  int rc = dalloc(e, &e->d_idx, B * e->map_stride * 2);
  e->d_raw = rc == RGBL_OK ? reinterpret_cast<float*>(e->d_idx + B * e->map_stride) : nullptr;
  if (rc == RGBL_OK) rc = dalloc(e, &e->d_proc, B * e->map_stride);
  if (rc == RGBL_OK) rc = dalloc(e, &e->d_cloud, (size_t)4 * cfg->max_points);
  if (rc == RGBL_OK) rc = dalloc(e, &e->d_ptdepth, B * (size_t)cfg->max_points);
  if (rc == RGBL_OK) rc = dalloc(e, &e->d_kp, (size_t)5 * cfg->max_keypoints);
  if (rc == RGBL_OK) {
    e->d_kpun = e->d_kp + (size_t)2 * cfg->max_keypoints;
    e->d_depth = e->d_kpun + cfg->max_keypoints;
    e->d_uright = e->d_depth + cfg->max_keypoints;
    if (hipHostMalloc(reinterpret_cast<void**>(&e->h_kio), sizeof(float) * 5 * (size_t)std::max(cfg->max_keypoints, 1), hipHostMallocDefault) != hipSuccess) {
      e->h_kio = nullptr;
      (void)hipGetLastError();
    }
  }
  if (const char* v = getenv("RGBL_XCD_MAP")) e->xcd_map = v[0] != '0';
  if (const char* v = getenv("RGBL_DEPTH_MAX_GEN")) e->max_gen = (uint32_t)std::min(std::max(atoi(v), 1), (int)e->max_gen);
  if (rc == RGBL_OK && hipStreamCreate(&e->own_stream) != hipSuccess) { set_error("hipStreamCreate failed"); rc = RGBL_ERR_HIP; }
  if (rc != RGBL_OK) { rgbl_depth_destroy(e); return rc; }
  e->stream = e->own_stream;
  *out = e;
  return RGBL_OK;
}

void rgbl_depth_destroy(rgbl_depth* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  e->timer.collect();
  for (void* p : e->allocs) (void)hipFree(p);
  if (e->h_kio) (void)hipHostFree(e->h_kio);
  if (e->h_cloud) (void)hipHostFree(e->h_cloud);
  if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
  delete e;
}

static int upload_cloud(rgbl_depth* e, const float* cloud, int n, int ld, bool xyzi) {
  hipStream_t s = e->stream;
  if (xyzi) RGBL_HIP(hipMemcpyAsync(e->d_cloud, cloud, sizeof(float) * 4 * (size_t)n, hipMemcpyHostToDevice, s));
  else RGBL_HIP(hipMemcpy2DAsync(e->d_cloud, sizeof(float) * n, cloud, sizeof(float) * ld, sizeof(float) * n, 4, hipMemcpyHostToDevice, s));
  return RGBL_OK;
}

// The part of CalculateDepthFromPcd that does not need the keypoints - upload of the scan, projection, up-sampling - queued
// on the handle's stream and NOT waited for: issued between rgbl_extract_begin and rgbl_extract it runs next to the extraction
// (the host stages the 1.9 MB of a KITTI scan while the GPU extracts).  The rgbl_depth_compute* call that follows with the same
// cloud pointer, point count and layout only gathers the keypoints' depths.  The caller must not change the scan in between.
static int depth_prefetch_host(rgbl_depth* e, const float* cloud, int n, int ld, bool xyzi, int w, int h) {
  if (!e) { set_error("null handle"); return RGBL_ERR_INVALID; }
  if (w != e->cfg.width || h != e->cfg.height || n < 0 || n > e->cfg.max_points || (n > 0 && (!cloud || (!xyzi && ld < n)))) {
    set_error("depth arguments do not match the handle (%dx%d, %d points)", e->cfg.width, e->cfg.height, e->cfg.max_points);
    return RGBL_ERR_INVALID;
  }
  RGBL_HIP(hipSetDevice(e->device));
  e->prefetched.active = false;
  if (n > 0) {
    // The scan goes through a page-locked block of the handle: the host copies it there (while the GPU extracts), the DMA
    // from there is asynchronous under every HIP runtime - an asynchronous copy straight from pageable memory is staged by
    // the runtime, and the one PyTorch ships (ROCm 7.0) holds it back behind the work of other streams: no overlap at all.
    if (!e->h_cloud && hipHostMalloc(reinterpret_cast<void**>(&e->h_cloud), sizeof(float) * 4 * (size_t)e->cfg.max_points, hipHostMallocDefault) != hipSuccess) {
      e->h_cloud = nullptr;
      (void)hipGetLastError();
    }
    if (e->h_cloud) {
      RGBL_HIP(hipStreamSynchronize(e->stream));  // the block's previous transfer (long done in a frame loop)
      if (xyzi) memcpy(e->h_cloud, cloud, sizeof(float) * 4 * (size_t)n);
      else for (int r = 0; r < 4; ++r) memcpy(e->h_cloud + (size_t)r * n, cloud + (size_t)r * ld, sizeof(float) * n);
      RGBL_HIP(hipMemcpyAsync(e->d_cloud, e->h_cloud, sizeof(float) * 4 * (size_t)n, hipMemcpyHostToDevice, e->stream));
    } else {
      RGBL_TRY(upload_cloud(e, cloud, n, ld, xyzi));
    }
  }
  RGBL_TRY(enqueue_maps(e, e->d_cloud, 1, n, n, 0, w, h, nullptr, xyzi, false, false));
  e->prefetched.active = true; e->prefetched.cloud = cloud; e->prefetched.n = n; e->prefetched.ld = ld; e->prefetched.xyzi = xyzi;
  return RGBL_OK;
}

static int depth_compute_host(rgbl_depth* e, const float* cloud, int n, int ld, bool xyzi, int w, int h, const float* kp_xy,
                              const float* kpun_x, int k, float* out_depth, float* out_uright, float* out_raw,
                              float* out_processed) {
  if (!e) { set_error("null handle"); return RGBL_ERR_INVALID; }
  if (w != e->cfg.width || h != e->cfg.height || n < 0 || n > e->cfg.max_points || k < 0 || k > e->cfg.max_keypoints ||
      (n > 0 && (!cloud || (!xyzi && ld < n))) || (k > 0 && (!kp_xy || !kpun_x || !out_depth || !out_uright))) {
    set_error("depth arguments do not match the handle (%dx%d, %d points, %d keypoints)", e->cfg.width, e->cfg.height,
              e->cfg.max_points, e->cfg.max_keypoints);
    return RGBL_ERR_INVALID;
  }
  RGBL_HIP(hipSetDevice(e->device));
  hipStream_t s = e->stream;
  StreamDrain drain(s);  // error returns included: the uploads below read the caller's buffers
  // rgbl_depth_prefetch was given this very cloud: its upload, projection and up-sampling are queued (or done)
  const bool have_maps = e->prefetched.active && e->prefetched.cloud == cloud && e->prefetched.n == n && e->prefetched.ld == ld &&
                         e->prefetched.xyzi == xyzi && !out_raw;
  e->prefetched.active = false;
  if (n > 0 && !have_maps) RGBL_TRY(upload_cloud(e, cloud, n, ld, xyzi));
  const size_t K = (size_t)e->cfg.max_keypoints;
  if (k > 0 && e->h_kio) {
    memcpy(e->h_kio, kp_xy, sizeof(float) * 2 * k);
    memcpy(e->h_kio + 2 * K, kpun_x, sizeof(float) * k);
    RGBL_HIP(hipMemcpyAsync(e->d_kp, e->h_kio, sizeof(float) * (2 * K + k), hipMemcpyHostToDevice, s));
  } else if (k > 0) {
    RGBL_HIP(hipMemcpyAsync(e->d_kp, kp_xy, sizeof(float) * 2 * k, hipMemcpyHostToDevice, s));
    RGBL_HIP(hipMemcpyAsync(e->d_kpun, kpun_x, sizeof(float) * k, hipMemcpyHostToDevice, s));
  }
  if (!have_maps) RGBL_TRY(enqueue_maps(e, e->d_cloud, 1, n, n, 0, w, h, nullptr, xyzi, out_raw != nullptr, out_processed != nullptr));
  else if (out_processed && !e->maps.dense) RGBL_TRY(enqueue_upsample(e, 1, w, h));  // a sparse prefetch, and the map is wanted after all
  RGBL_TRY(enqueue_keypoints(e, 1, w, h, e->d_kp, 2, 0, e->d_kpun, 1, 0, nullptr, k, k, e->d_depth, e->d_uright, 0));
  e->last_k = k;
  if (k > 0 && e->h_kio) {
    RGBL_HIP(hipMemcpyAsync(e->h_kio + 3 * K, e->d_depth, sizeof(float) * (K + k), hipMemcpyDeviceToHost, s));
  } else if (k > 0) {
    RGBL_HIP(hipMemcpyAsync(out_depth, e->d_depth, sizeof(float) * k, hipMemcpyDeviceToHost, s));
    RGBL_HIP(hipMemcpyAsync(out_uright, e->d_uright, sizeof(float) * k, hipMemcpyDeviceToHost, s));
  }
  if (out_raw) RGBL_HIP(hipMemcpyAsync(out_raw, e->d_raw, sizeof(float) * e->map_stride, hipMemcpyDeviceToHost, s));
  if (out_processed && e->cfg.method != RGBL_UPS_NEAREST_NEIGHBOR_PIXEL)
    RGBL_HIP(hipMemcpyAsync(out_processed, e->d_proc, sizeof(float) * e->map_stride, hipMemcpyDeviceToHost, s));
  RGBL_HIP(hipStreamSynchronize(s));
  if (k > 0 && e->h_kio) {
    memcpy(out_depth, e->h_kio + 3 * K, sizeof(float) * k);
    memcpy(out_uright, e->h_kio + 4 * K, sizeof(float) * k);
  }
  e->timer.collect();
  return RGBL_OK;
}

int rgbl_depth_compute(rgbl_depth* e, const float* cloud, int n, int ld, int w, int h, const float* kp_xy,
                       const float* kpun_x, int k, float* out_depth, float* out_uright, float* out_raw,
                       float* out_processed) {
  return depth_compute_host(e, cloud, n, ld, false, w, h, kp_xy, kpun_x, k, out_depth, out_uright, out_raw, out_processed);
}

int rgbl_depth_prefetch(rgbl_depth* e, const float* cloud, int n, int ld, int w, int h) { return depth_prefetch_host(e, cloud, n, ld, false, w, h); }
int rgbl_depth_prefetch_xyzi(rgbl_depth* e, const float* xyzi, int n, int w, int h) { return depth_prefetch_host(e, xyzi, n, n, true, w, h); }
// A prefetched scan is recognised by its host pointer, point count and layout only.  A caller that frees (or rewrites) the
// buffer without having called rgbl_depth_compute* on it says so here: the allocator may hand the same address to the next scan.
int rgbl_depth_prefetch_cancel(rgbl_depth* e) {
  if (!e) { set_error("null handle"); return RGBL_ERR_INVALID; }
  if (e->prefetched.active) {
    e->prefetched.active = false;
    RGBL_HIP(hipSetDevice(e->device));
    RGBL_HIP(hipStreamSynchronize(e->stream));  // the staged copy of the scan has left the page-locked block
  }
  return RGBL_OK;
}

// the scan as it lies in a KITTI velodyne .bin file (LoadPointcloudBinaryMat, rgbl_kitti.cc:151-185)
int rgbl_depth_compute_xyzi(rgbl_depth* e, const float* xyzi, int n, int w, int h, const float* kp_xy, const float* kpun_x,
                            int k, float* out_depth, float* out_uright, float* out_raw, float* out_processed) {
  return depth_compute_host(e, xyzi, n, n, true, w, h, kp_xy, kpun_x, k, out_depth, out_uright, out_raw, out_processed);
}

int rgbl_depth_project_xyzi_batch_device(rgbl_depth* e, const float* d_xyzi, int batch, int n, size_t scan_stride, int w, int h,
                                         float* d_processed) {
  if (!e || !d_xyzi) { set_error("null argument"); return RGBL_ERR_INVALID; }
  if (w != e->cfg.width || h != e->cfg.height || batch < 1 || batch > e->cfg.max_batch || n < 0 ||
      (batch > 1 && (scan_stride < (size_t)4 * n || (scan_stride & 3))) || ((uintptr_t)d_xyzi & 15)) {
    set_error("xyzi batch: scans must be 16-byte aligned, scan_stride a multiple of 4 floats and >= 4 n");
    return RGBL_ERR_INVALID;
  }
  RGBL_HIP(hipSetDevice(e->device));
  return enqueue_maps(e, d_xyzi, batch, n, n, scan_stride, w, h, d_processed, true, false, d_processed != nullptr);
}

int rgbl_depth_project_batch_device(rgbl_depth* e, const float* d_cloud, int batch, int n, int ld, size_t cloud_stride, int w,
                                    int h, float* d_processed) {
  if (!e || !d_cloud) { set_error("null argument"); return RGBL_ERR_INVALID; }
  if (w != e->cfg.width || h != e->cfg.height || batch < 1 || batch > e->cfg.max_batch || n < 0 || ld < n ||
      (batch > 1 && cloud_stride < (size_t)3 * ld + n)) {
    set_error("depth batch arguments do not match the handle");
    return RGBL_ERR_INVALID;
  }
  RGBL_HIP(hipSetDevice(e->device));
  return enqueue_maps(e, d_cloud, batch, n, ld, cloud_stride, w, h, d_processed, false, false, d_processed != nullptr);
}

int rgbl_depth_gather_batch_device(rgbl_depth* e, int batch, int w, int h, const rgbl_keypoint* d_kp, const int32_t* d_n,
                                   int kp_cap, const float* d_kpun_x, float* d_depth, float* d_uright) {
  if (!e || !d_kp || !d_n || !d_depth || !d_uright) { set_error("null argument"); return RGBL_ERR_INVALID; }
  if (w != e->cfg.width || h != e->cfg.height || batch < 1 || batch > e->cfg.max_batch || kp_cap < 1) {
    set_error("depth batch arguments do not match the handle");
    return RGBL_ERR_INVALID;
  }
  RGBL_HIP(hipSetDevice(e->device));
  const float* kp = reinterpret_cast<const float*>(d_kp);
  const int kstride = (int)(sizeof(rgbl_keypoint) / sizeof(float));
  const float* kpun = d_kpun_x ? d_kpun_x : kp;
  return enqueue_keypoints(e, batch, w, h, kp, kstride, (size_t)kp_cap * kstride, kpun, d_kpun_x ? 1 : kstride,
                           d_kpun_x ? (size_t)kp_cap : (size_t)kp_cap * kstride, d_n, 0, kp_cap, d_depth, d_uright,
                           (size_t)kp_cap);
}

int rgbl_depth_batch_device(rgbl_depth* e, const float* d_cloud, int batch, int n, int ld, size_t cloud_stride, int w, int h,
                            const rgbl_keypoint* d_kp, const int32_t* d_n, int kp_cap, const float* d_kpun_x, float* d_depth,
                            float* d_uright, float* d_processed) {
  RGBL_TRY(rgbl_depth_project_batch_device(e, d_cloud, batch, n, ld, cloud_stride, w, h, d_processed));
  return rgbl_depth_gather_batch_device(e, batch, w, h, d_kp, d_n, kp_cap, d_kpun_x, d_depth, d_uright);
}

void* rgbl_depth_stream(rgbl_depth* e) { return e ? (void*)e->stream : nullptr; }

int rgbl_depth_sync(rgbl_depth* e) {
  if (!e) { set_error("null handle"); return RGBL_ERR_INVALID; }
  RGBL_HIP(hipSetDevice(e->device));
  RGBL_HIP(hipStreamSynchronize(e->stream));
  e->timer.collect();
  return RGBL_OK;
}
int rgbl_depth_set_stream(rgbl_depth* e, void* hip_stream) {
  if (!e) { set_error("null handle"); return RGBL_ERR_INVALID; }
  RGBL_HIP(hipStreamSynchronize(e->stream));
  e->stream = hip_stream ? (hipStream_t)hip_stream : e->own_stream;
  return RGBL_OK;
}
int rgbl_depth_set_sparse(rgbl_depth* e, int enable) {
  if (!e) { set_error("null handle"); return RGBL_ERR_INVALID; }
  e->sparse = enable != 0;
  return RGBL_OK;
}
int rgbl_depth_profile(rgbl_depth* e, int enable) {
  if (!e) { set_error("null handle"); return RGBL_ERR_INVALID; }
  RGBL_HIP(hipStreamSynchronize(e->stream));
  e->timer.reset();
  e->timer.enabled = enable != 0;
  return RGBL_OK;
}
int rgbl_depth_profile_read(rgbl_depth* e, const char** names, double* total_ms, long* launches, int cap) {
  if (!e) return 0;
  (void)hipStreamSynchronize(e->stream);
  e->timer.collect();
  const int n = (int)e->timer.names.size();
  for (int i = 0; i < n && i < cap; ++i) {
    if (names) names[i] = e->timer.names[i].c_str();
    if (total_ms) total_ms[i] = e->timer.total_ms[i];
    if (launches) launches[i] = e->timer.count[i];
  }
  return n;
}
int rgbl_depth_profile_samples(rgbl_depth* e, int kernel, float* ms, int cap) {
  if (!e || (cap > 0 && !ms)) return 0;
  (void)hipStreamSynchronize(e->stream);
  e->timer.collect();
  return e->timer.read_samples(kernel, ms, cap);
}

}  // extern "C"

// not part of the C ABI: where the last host-pointer call left mvuRight (common.h)
int rgbl_internal_depth_uright(rgbl_depth* d, const float** d_uright, int* k, hipStream_t* stream) {
  if (!d) { set_error("null depth handle"); return RGBL_ERR_INVALID; }
  *d_uright = d->d_uright; *k = d->last_k; *stream = d->stream;
  return RGBL_OK;
}
