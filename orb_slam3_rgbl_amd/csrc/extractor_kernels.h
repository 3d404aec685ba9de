// extractor_kernels.h — HIP kernels of the ORB extraction path (gfx950, wave64).
//
// One batched launch per stage covers B independent frames (blockIdx.y / .z = frame):
//   k_resize_linear   level l from level l-1          replaces cv::resize        (ORBextractor.cc:1183)
//   k_fast_cells      FAST-9/16 score + per-cell NMS + two-threshold choice + ordered compaction
//                                                     replaces ~900 cv::FAST calls (ORBextractor.cc:806-872)
//   k_octree          quad-tree distribution          replaces DistributeOctTree (ORBextractor.cc:555-779)
//   k_gauss7          7x7 sigma-2 fixed-point blur     replaces cv::GaussianBlur  (ORBextractor.cc:1133)
//   k_orient_brief    IC_Angle + steered BRIEF + pack  replaces ORBextractor.cc:76-146,1149-1165
//   k_lapping_permute vLappingArea front/back packing  replaces ORBextractor.cc:1153-1162
// All arithmetic is integer except the orientation polynomial / steering (fp32, fp64 sincos), compiled
// with -ffp-contract=off.  Results are bit-exact with the CPU oracle.
#pragma once
#include "common.h"
#include "sincos_glibc.h"

namespace rgbl {

constexpr int kMaxLevels = 16;
constexpr int kMaxRoots = 16;
constexpr int kMinBorder = 16;  // EDGE_THRESHOLD - 3

struct ResizeTab {  // one output column / row of cv::resize's fixed-point tables
  int32_t sofs;     // first source index
  int16_t a0, a1;   // 11-bit weights (sum 2048)
};

struct LevelGeom {
  int w, h, pitch;            // level size; pitch of the pyramid / blur buffers (bytes)
  uint32_t img_off;           // byte offset of this level inside one frame's pyramid (and blur) buffer
  int quota;                  // mnFeaturesPerLevel[level]
  int kcap, koff;             // keypoint slots of this level inside one frame's keypoint arrays
  int n_cols, n_rows, w_cell, h_cell, max_bx, max_by;
  int n_cells, cell_off, cell_cap;
  uint32_t slot_off;          // first candidate slot (entries) inside one frame's slot array
  uint32_t key_off, key_cap;  // quad-tree key buffers (entries) inside one frame
  uint32_t node_off, node_cap;
  int n_ini;                  // number of root nodes
  int root_x[kMaxRoots + 1];  // root node x bounds
  int root_first[kMaxRoots + 1];  // [k], k >= 1: first x whose keys go to root k or beyond (vpIniNodes[kp.pt.x / hX]: float division)
  uint32_t m_wcell, m_hcell;  // ceil(2^20 / w_cell), ceil(2^20 / h_cell): cell of a candidate from its coordinates
  uint32_t rootx_off;         // byte offset into the root lookup table (index by x relative to minBorder)
  uint32_t xtab_off, ytab_off;
  float scale;                // mvScaleFactor[level]
  int patch_size;             // (int)(31 * scale)
};

struct UMax { int v[16]; };

struct QNode {  // quad-tree node, 16 bytes
  uint16_t x0, x1, y0, y1;
  uint32_t beg;  // first key
  uint32_t cnt;  // bit 31: keys live in buffer B
};
struct QDiv { uint32_t c[4]; };

// candidate / key packing: x (12 bit) | y (12 bit) << 12 | score << 24, coordinates relative to minBorder
__device__ __forceinline__ uint32_t pack_key(int x, int y, int s) { return (uint32_t)x | ((uint32_t)y << 12) | ((uint32_t)s << 24); }
__device__ __forceinline__ int key_x(uint32_t k) { return (int)(k & 0xfff); }
__device__ __forceinline__ int key_y(uint32_t k) { return (int)((k >> 12) & 0xfff); }
__device__ __forceinline__ int key_s(uint32_t k) { return (int)(k >> 24); }

// ------------------------------------------------------------------------------------------------
// cv::resize(INTER_LINEAR, CV_8UC1): 11-bit fixed-point bilinear, each work-item makes 4 output pixels of one
// row.  The 4 outputs read a short run of source pixels (about 6 at scale 1.2) from two rows: those are fetched
// as two 32-bit words per row (8 bytes from the first source column) instead of 16 byte loads; the coefficient
// table entries of the 4 columns are one 32-byte read.  grid = (ceil(dw / (4 kResizeLanes)), ceil(dh / 16), B), block = kResizeLanes x 4.
__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t* p) {
  uint32_t v;
  __builtin_memcpy(&v, p, 4);  // level 0 may have an odd row stride: gfx950 global loads need no alignment
  return v;
}

constexpr int kResizeRows = 4;  // output rows per work-item (1 / 2 / 4 measured: 4 is 6 % ahead of 1 on the whole step)
// A workgroup covers kResizeLanes * 4 output columns x 16 rows.  With 64 lanes per row (256 columns) the last workgroup
// column of a level is mostly empty (1034 columns = 4.04 workgroups: 83 % of the launched lanes work over the 7 levels);
// 32 lanes per row leave 92 % (0.595 -> 0.575 ms; 16 lanes per row: 0.58).
constexpr int kResizeLanes = 32, kResizeWG = kResizeLanes * 4;
// One group of 4 output columns, prepared by the host: the 8-byte source window (pulled back at the end of a row so that it
// stays inside it), per column the v_perm_b32 selector that pulls its two taps out of the window as two 16-bit halves and
// the two 11-bit weights packed the same way (one v_dot2_u32_u16 per source row and column).  sxa < 0: the taps of the
// group do not fit 8 bytes (scale factors above ~1.7) or the source is narrower than 8 px - byte path.
struct ResizeGroup { uint32_t sel[4], w[4]; };
static_assert(sizeof(ResizeGroup) == 32, "two 16-byte loads per group");

__global__ __launch_bounds__(kResizeWG) void k_resize_linear(const uint8_t* __restrict__ src, int spitch,
                                                       size_t sframe, int sw, int sh,
                                                       uint8_t* __restrict__ dst, int dpitch, size_t dframe,
                                                       int dw, int dh, const ResizeTab* __restrict__ xtab,
                                                       const ResizeGroup* __restrict__ xgroups, const int32_t* __restrict__ xsxa,
                                                       const ResizeTab* __restrict__ ytab, int tiles_x) {
  // grid = xcd_grid(tiles_x * tiles_y, B) (common.h)
  const int tx = threadIdx.x % kResizeLanes, ty = threadIdx.x / kResizeLanes;
  const int tile_y = xcd_item() / tiles_x, tile_x = xcd_item() - tile_y * tiles_x;
  const int dx0 = (tile_x * kResizeLanes + tx) * 4;
  const int dy0 = (tile_y * 4 + ty) * kResizeRows;
  if (dy0 >= dh || dx0 >= dw) return;
  const uint8_t* S = src + (size_t)xcd_frame() * sframe;
  uint8_t* D = dst + (size_t)xcd_frame() * dframe + (size_t)dy0 * dpitch;
  // the group's record and the row table are requested before either is used (one round trip, not two); the x table is
  // padded to a multiple of 4 entries per level with copies of the last entry
  const int sxa = xsxa[dx0 >> 2];
  const uint4* g4 = reinterpret_cast<const uint4*>(xgroups + (dx0 >> 2));
  const uint4 gs = g4[0], gw = g4[1];
  // the kResizeRows rows' vertical taps; rows past the image repeat the last one (computed, not stored)
  ResizeTab ry[kResizeRows];
#pragma unroll
  for (int r = 0; r < kResizeRows; ++r) ry[r] = ytab[imin(dy0 + r, dh - 1)];
  if (sxa >= 0) {
    // all 4 kResizeRows source windows are requested before the first one is used: a wave keeps 4 KB in flight, the
    // kernel is bound by the round trips per byte otherwise (one row per work-item ran at 1.2 TB/s).  The last (partial)
    // group of a row takes this path too and stores a full word (the level's row pitch is 64-byte aligned).
    uint32_t r0l[kResizeRows], r0h[kResizeRows], r1l[kResizeRows], r1h[kResizeRows];
#pragma unroll
    for (int r = 0; r < kResizeRows; ++r) {
      const int sy0 = imin(imax(ry[r].sofs, 0), sh - 1), sy1 = imin(imax(ry[r].sofs + 1, 0), sh - 1);
      const uint8_t* S0 = S + (size_t)sy0 * spitch + sxa;
      const uint8_t* S1 = S + (size_t)sy1 * spitch + sxa;
      r0l[r] = load_u32_unaligned(S0); r0h[r] = load_u32_unaligned(S0 + 4);
      r1l[r] = load_u32_unaligned(S1); r1h[r] = load_u32_unaligned(S1 + 4);
    }
    const uint32_t sel[4] = {gs.x, gs.y, gs.z, gs.w}, wt[4] = {gw.x, gw.y, gw.z, gw.w};
#pragma unroll
    for (int r = 0; r < kResizeRows; ++r) {
      const int b0 = ry[r].a0, b1 = ry[r].a1;
      uint32_t out = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int h0 = (int)udot2(perm_bytes(r0h[r], r0l[r], sel[i]), wt[i], 0u);  // p00 a0 + p01 a1
        const int h1 = (int)udot2(perm_bytes(r1h[r], r1l[r], sel[i]), wt[i], 0u);
        const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
        out |= (uint32_t)(v & 0xff) << (8 * i);
      }
      if (dy0 + r < dh) *reinterpret_cast<uint32_t*>(D + (size_t)r * dpitch + dx0) = out;  // dpitch and dx0 are multiples of 4
    }
    return;
  }
  ResizeTab rx[4];
  {
    const uint4* t4 = reinterpret_cast<const uint4*>(xtab + dx0);
    const uint4 q0 = t4[0], q1 = t4[1];
    rx[0].sofs = (int)q0.x; rx[0].a0 = (int16_t)(q0.y & 0xffff); rx[0].a1 = (int16_t)(q0.y >> 16);
    rx[1].sofs = (int)q0.z; rx[1].a0 = (int16_t)(q0.w & 0xffff); rx[1].a1 = (int16_t)(q0.w >> 16);
    rx[2].sofs = (int)q1.x; rx[2].a0 = (int16_t)(q1.y & 0xffff); rx[2].a1 = (int16_t)(q1.y >> 16);
    rx[3].sofs = (int)q1.z; rx[3].a0 = (int16_t)(q1.w & 0xffff); rx[3].a1 = (int16_t)(q1.w >> 16);
  }
  // large scale factors / sources narrower than 8 px: byte path (unrolled with compile-time r: a run-time index into ry[]
  // makes the compiler park the array in LDS, 16 KB per workgroup, for the common path as well)
#pragma unroll
  for (int r = 0; r < kResizeRows; ++r) {
    if (dy0 + r >= dh) break;
    const int sy0 = imin(imax(ry[r].sofs, 0), sh - 1), sy1 = imin(imax(ry[r].sofs + 1, 0), sh - 1);
    const uint8_t* S0 = S + (size_t)sy0 * spitch;
    const uint8_t* S1 = S + (size_t)sy1 * spitch;
    const int b0 = ry[r].a0, b1 = ry[r].a1;
    uint8_t* Dr = D + (size_t)r * dpitch;
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int dx = dx0 + i;
      if (dx >= dw) break;
      const int sx = rx[i].sofs, sx1 = imin(sx + 1, sw - 1);
      const int h0 = S0[sx] * rx[i].a0 + S0[sx1] * rx[i].a1;
      const int h1 = S1[sx] * rx[i].a0 + S1[sx1] * rx[i].a1;
      const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
      out |= (uint32_t)(v & 0xff) << (8 * i);
    }
    if (dx0 + 3 < dw) {
      *reinterpret_cast<uint32_t*>(Dr + dx0) = out;
    } else {
      for (int i = 0; dx0 + i < dw; ++i) Dr[dx0 + i] = (uint8_t)(out >> (8 * i));
    }
  }
}

// ------------------------------------------------------------------------------------------------
// FAST-9/16 on one detection cell per workgroup.
constexpr int kCellMax = 72;        // max scanned cell side handled (wCell/hCell <= 72: levels at least 35 px wide)
constexpr int kCellSmall = 48;      // the common case (cells of 35..48 px): 7.4 KB of LDS instead of 24 KB per workgroup

// Bresenham ring of radius 3, OpenCV order (modules/features2d/src/fast_score.cpp makeOffsets)
#define RGBL_RING(c, P, k)                                                                            \
  ((k) == 0 ? (c)[3 * (P)] : (k) == 1 ? (c)[3 * (P) + 1] : (k) == 2 ? (c)[2 * (P) + 2]                \
   : (k) == 3 ? (c)[(P) + 3] : (k) == 4 ? (c)[3] : (k) == 5 ? (c)[-(P) + 3] : (k) == 6 ? (c)[-2 * (P) + 2] \
   : (k) == 7 ? (c)[-3 * (P) + 1] : (k) == 8 ? (c)[-3 * (P)] : (k) == 9 ? (c)[-3 * (P) - 1]           \
   : (k) == 10 ? (c)[-2 * (P) - 2] : (k) == 11 ? (c)[-(P) - 3] : (k) == 12 ? (c)[-3]                  \
   : (k) == 13 ? (c)[(P) - 3] : (k) == 14 ? (c)[2 * (P) - 2] : (c)[3 * (P) - 1])

// ring pixel k of cv::FAST's 16-ring as a byte offset in a tile of pitch P
__device__ __forceinline__ int ring_off(int k, int P) {
  switch (k) {
    case 0: return 3 * P; case 1: return 3 * P + 1; case 2: return 2 * P + 2; case 3: return P + 3;
    case 4: return 3; case 5: return -P + 3; case 6: return -2 * P + 2; case 7: return -3 * P + 1;
    case 8: return -3 * P; case 9: return -3 * P - 1; case 10: return -2 * P - 2; case 11: return -P - 3;
    case 12: return -3; case 13: return P - 3; case 14: return 2 * P - 2; default: return 3 * P - 1;
  }
}

// packed signed 16-bit helpers of the score (v_pk_sub_i16, v_pk_min_i16, v_pk_max_i16; half swaps fold into op_sel)
#ifdef RGBL_EMU
__device__ __forceinline__ uint32_t pk_sub_i16(uint32_t a, uint32_t b) { return ((a - b) & 0xffffu) | (((a >> 16) - (b >> 16)) << 16); }
__device__ __forceinline__ uint32_t pk_min_i16(uint32_t a, uint32_t b) {
  const int16_t al = (int16_t)a, bl = (int16_t)b, ah = (int16_t)(a >> 16), bh = (int16_t)(b >> 16);
  return (uint16_t)(al < bl ? al : bl) | ((uint32_t)(uint16_t)(ah < bh ? ah : bh) << 16);
}
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b) {
  const int16_t al = (int16_t)a, bl = (int16_t)b, ah = (int16_t)(a >> 16), bh = (int16_t)(b >> 16);
  return (uint16_t)(al > bl ? al : bl) | ((uint32_t)(uint16_t)(ah > bh ? ah : bh) << 16);
}
__device__ __forceinline__ uint32_t pk_swap(uint32_t a) { return (a >> 16) | (a << 16); }
#else
typedef short rgbl_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ rgbl_s2 as_s2(uint32_t a) { rgbl_s2 x; __builtin_memcpy(&x, &a, 4); return x; }
__device__ __forceinline__ uint32_t from_s2(rgbl_s2 x) { uint32_t a; __builtin_memcpy(&a, &x, 4); return a; }
__device__ __forceinline__ uint32_t pk_sub_i16(uint32_t a, uint32_t b) { return from_s2(as_s2(a) - as_s2(b)); }
__device__ __forceinline__ uint32_t pk_min_i16(uint32_t a, uint32_t b) { return from_s2(__builtin_elementwise_min(as_s2(a), as_s2(b))); }
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b) { return from_s2(__builtin_elementwise_max(as_s2(a), as_s2(b))); }
__device__ __forceinline__ uint32_t pk_swap(uint32_t a) { return from_s2(as_s2(a).yx); }
#endif

// v_mul_u32_u24 spelled out: the compiler turns __umul24 of operands it cannot bound into a mask + v_mul_lo_u32, which issues at
// a quarter of the rate.  Both operands must be below 2^24, the second one wave-uniform.
__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) {
#ifdef RGBL_EMU
  return (a & 0xffffffu) * (b & 0xffffffu);
#else
  uint32_t r;
  asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b));   // b: wave-uniform
  return r;
#endif
}

// v + (pred ? 1 : 0) as ONE vector instruction: the predicate is a comparison's scalar mask already, v_addc_co takes it as
// the carry (the compiler's own choice was v_cndmask 0 / 1 + add).
__device__ __forceinline__ int add_flag(int v, unsigned long long mask) {  // mask: wave_ballot of bare comparisons, combined with & |
#ifdef RGBL_EMU
  return v + (int)((mask >> lane_id()) & 1ull);
#else
  int r;
  asm("v_addc_co_u32_e64 %0, vcc, 0, %1, %2" : "=v"(r) : "v"(v), "s"(mask) : "vcc");
  return r;
#endif
}
// Hides where an LDS index came from: the compiler otherwise folds "index - constant" back into the accesses' offsets, and
// every NEGATIVE offset then costs an address addition (DS instructions take unsigned immediate offsets only).
__device__ __forceinline__ int opaque(int v) {
#ifndef RGBL_EMU
  asm volatile("" : "+v"(v));
#endif
  return v;
}

// The FAST score of ONE polarity: the largest threshold t at which the pixel still has a 9-arc of that polarity (< 0: none) =
// cv cornerScore<16> for a corner of that polarity.  A 9-arc of darker pixels and a 9-arc of brighter ones exclude each other
// (9 + 9 > 16), and the pre-screen says which of the two a pixel can have - well below 5 % of its survivors pass for both.
// sgn = 0xffffffff: the dark arc, differences d_k = v - ring_k; sgn = 0x00010001: the bright arc, d_k = ring_k - v; either
// way the result is max_k min(d_k .. d_k+8) - 1.  Packed 16-bit arithmetic: register k holds d_k and d_{k+8} as two signed
// halves (one v_pk_mad_i16 per ring pair forms them), so ring position k + 8 is register k with its halves swapped (op_sel) and
// every sliding minimum is computed for two ring positions at once; log-step windows 2, 4, 8, + 1.
#ifdef RGBL_EMU
__device__ __forceinline__ uint32_t pk_mad_i16(uint32_t a, uint32_t b, uint32_t c) {
  const uint32_t lo = (uint32_t)((int)(int16_t)a * (int)(int16_t)b + (int)(int16_t)c) & 0xffffu;
  const uint32_t hi = (uint32_t)((int)(int16_t)(a >> 16) * (int)(int16_t)(b >> 16) + (int)(int16_t)(c >> 16)) & 0xffffu;
  return lo | (hi << 16);
}
#else
__device__ __forceinline__ uint32_t pk_mad_i16(uint32_t a, uint32_t b, uint32_t c) { return from_s2(as_s2(a) * as_s2(b) + as_s2(c)); }
#endif
__device__ __forceinline__ int fast_score_one(const uint8_t* tile, int t, int P, uint32_t sgn) {
  // the pixel at tile[t]; every read as a non-negative offset from the ring's first byte (row - 3, column - 3)
  const uint8_t* const o = tile + opaque(t - (3 * P + 3));
#define RGBL_O(k) o[ring_off((k), P) + 3 * P + 3]
  const uint32_t v = o[3 * P + 3];
  const uint32_t vs = pk_mad_i16(v | (v << 16), sgn ^ 0xfffefffeu, 0u);  // -sgn * v in both halves (sgn = -1 -> * 1, sgn = 1 -> * -1)
  uint32_t D[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const uint32_t a = RGBL_O(k), b = RGBL_O(k + 8);
    D[k] = pk_mad_i16(a | (b << 16), sgn, vs);  // sgn * ring - sgn * v
  }
#define RGBL_AT(A, k) ((k) < 8 ? (A)[(k) & 7] : pk_swap((A)[((k) - 8) & 7]))
  // A 9-window starts at an even position s - 1 or at the odd position s behind it; both contain the 8-window s .. s + 7, the
  // first adds d[s-1], the second d[s+8], and max(min(a, x), min(a, y)) = min(a, max(x, y)).  So only the 8-windows at the ODD
  // positions are needed: 4 registers (position s and s + 8 share one) instead of 8 at every log step - 23 packed
  // operations instead of 40.  (cv's cornerScore<16> walks k = 0, 2, .. 14 the same way.)
  uint32_t mn2[4], mn4[4], mn8[4];
#define RGBL_ODD(A, s) ((s) < 8 ? (A)[((s) >> 1) & 3] : pk_swap((A)[(((s) - 8) >> 1) & 3]))   /* register of the odd position s */
#pragma unroll
  for (int i = 0; i < 4; ++i) mn2[i] = pk_min_i16(D[2 * i + 1], RGBL_AT(D, 2 * i + 2));
#pragma unroll
  for (int i = 0; i < 4; ++i) mn4[i] = pk_min_i16(mn2[i], RGBL_ODD(mn2, 2 * i + 3));
#pragma unroll
  for (int i = 0; i < 4; ++i) mn8[i] = pk_min_i16(mn4[i], RGBL_ODD(mn4, 2 * i + 5));
  uint32_t best = 0x80008000u;
#pragma unroll
  for (int i = 0; i < 4; ++i) best = pk_max_i16(best, pk_min_i16(mn8[i], pk_max_i16(D[2 * i], pk_swap(D[2 * i + 1]))));
#undef RGBL_ODD
#undef RGBL_AT
#undef RGBL_O
  return imax((int)(int16_t)(best & 0xffffu), (int)(int16_t)(best >> 16)) - 1;
}

// Two adjacent aligned words of the LDS tile (one ds_read2_b32).  Unaligned LDS reads are legal on gfx950 but cost ~15 LDS
// cycles per instruction (SQ_LDS_UNALIGNED_STALL; measured: 68 % of the kernel's time went there), so the tile is only ever
// read through aligned words and the byte shifts happen in v_perm_b32 / v_alignbyte_b32.
struct alignas(4) LdsPair { uint32_t lo, hi; };
__device__ __forceinline__ LdsPair lds_pair(const uint8_t* p) { return *reinterpret_cast<const LdsPair*>(p); }

// grid = xcd_grid(cells per frame over all levels, B) (common.h), block = 256.  CM = compile-time bound of the scanned cell side: the
// LDS tiles are sized by it, and LDS is what limits the workgroups per CU (6 at CM = 72; 17 at CM = 48, where 16 workgroups of 128 fill the 32 wave slots).
// value of the wave's first lane, wave-uniform
__device__ __forceinline__ int wave_first(int v) {
#ifdef RGBL_EMU
  return __shfl(v, 0);
#else
  return __builtin_amdgcn_readfirstlane(v);
#endif
}

struct FastCell {  // one detection cell of a frame, prepared by the host (upload_tables)
  uint16_t ini_x, ini_y;     // first pixel of its tile (cell + 3-px ring margin) in the level
  uint16_t kx0, ky0;         // key coordinates (relative to minBorder) of its first scanned pixel
  uint8_t tw, th, l, skip;   // tile size, level; skip: the cell starts too close to the border (ORBextractor.cc:810-822)
  uint16_t pitch, cell_cap;
  uint32_t magic, wmagic;    // ceil(2^20 / scanned width), ceil(2^20 / tile words per row)
  uint32_t img_off, slot_base;
};
static_assert(sizeof(FastCell) == 32, "one s_load_dwordx8 per cell");

// lanes of the wave for which `pred` holds, as a wave-uniform mask (the condition's own scalar mask: __ballot(int) would
// first turn the predicate into a register and compare that again)
__device__ __forceinline__ unsigned long long wave_ballot(bool pred) {
#ifdef RGBL_EMU
  return __ballot(pred ? 1 : 0);
#else
  return __builtin_amdgcn_ballot_w64(pred);
#endif
}
// number of set bits of a wave mask below the calling lane (v_mbcnt_lo / v_mbcnt_hi)
__device__ __forceinline__ int wave_rank(unsigned long long mask) {
#ifdef RGBL_EMU
  return (int)__popcll(mask & lanemask_lt());
#else
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
#endif
}
__device__ __forceinline__ int wave_last(int v) {  // value of lane 63, wave-uniform
#ifdef RGBL_EMU
  return __shfl(v, 63);
#else
  return __builtin_amdgcn_readlane(v, 63);
#endif
}

// 16 bytes per lane from global memory straight into LDS (gfx950 LDS-DMA, global_load_lds_dwordx4): no VGPR round trip, no
// ds_write.  The destination is NOT per lane: the wave's 64 pieces land back to back from `lds_wave_base` (wave-uniform) in
// lane order, inactive lanes leave their 16 bytes untouched.  The data is ordered for LDS reads by vmcnt(0) + a barrier
// (__syncthreads() emits both while such a load is in flight).
__device__ __forceinline__ void lds_dma16(const uint8_t* gsrc, uint8_t* lds_wave_base) {
#ifdef RGBL_EMU
  __builtin_memcpy(lds_wave_base + 16 * lane_id(), gsrc, 16);
#else
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  __builtin_amdgcn_global_load_lds((gptr_t)gsrc, (lptr_t)lds_wave_base, 16, 0, 0);
#endif
}

// Bytes K and K + 2 of the 12 bytes w0 | w1 | w2 (numbered -4 .. 7) as the HIGH bytes of a register's two 16-bit halves.
// kClean: the low bytes are zero; otherwise they hold whatever is cheapest (nothing at all when K = 1 mod 4).
template <int K, bool kClean = false>
__device__ __forceinline__ uint32_t hi_pair(uint32_t w0, uint32_t w1, uint32_t w2) {
  constexpr int q = (K + 4) >> 2, r = (K + 4) & 3;
  static_assert(K >= -4 && K + 2 <= 7, "inside the three words");
  const uint32_t lo = q == 0 ? w0 : q == 1 ? w1 : w2, hi = q == 0 ? w1 : w2;
  if (!kClean && r == 1) return lo;
  if (!kClean && r == 0) return lo << 8;
  return perm_bytes(hi, lo, 0x000c000cu | ((uint32_t)r << 8) | ((uint32_t)(r + 2) << 24));
}

// One detection cell per workgroup; the reference's own control flow (ORBextractor.cc:826-846): cv::FAST at iniThFAST on the
// cell, and only if that finds nothing cv::FAST at minThFAST.  (Rounds 1 - 2 made one pass at minThFAST and derived the
// iniThFAST set from the same score map: on frames where most pre-screen survivors at the low threshold are not corners at
// the high one, the exact score - the expensive phase - was computed for pixels whose result the reference never looks at.)
template <int CM, int BS, int P>
__global__ __launch_bounds__(BS) void k_fast_cells(const FastCell* __restrict__ cells,
                                                    const uint8_t* __restrict__ img0, int pitch0,
                                                    size_t frame0, const uint8_t* __restrict__ pyr,
                                                    size_t pyr_frame, int ini_th, int min_th,
                                                    uint32_t* __restrict__ cell_cnt, size_t cells_frame,
                                                    uint32_t* __restrict__ slots, size_t slots_frame, int cell_begin,
                                                    const LevelGeom* __restrict__ geom, int n_levels, uint32_t* __restrict__ dense_keys,
                                                    size_t keys_frame, uint32_t* __restrict__ level_cnt) {
  // P = pitch of BOTH LDS tiles in bytes: whole 16-byte pieces (the LDS-DMA's granule) and at least the widest tile row
  // (cell width + 7).  48 for cells up to 41 px wide - every level of KITTI, EuRoC, VGA or 4K frames -: 12 banks per row, so
  // the rows of a cell rotate through all 32 LDS banks and the scattered reads of phases B and C (a wave's survivors lie in
  // a handful of neighbouring rows) spread out; with 64 (16 banks) every other row started on the same bank and the cell's 10
  // word columns used 20 of the 32 banks: LDS bank-conflict cycles 1.45e8 -> see DESIGN 9.  64 / 80 for wider cells.
  static_assert(P % 16 == 0 && P >= 48 && P <= 80, "tile pitch");
  constexpr int kPieces = P / 16;
  // A pixel is named by its byte offset t = ty * P + tx in the pixel tile (tile coordinates: scanned pixel (x, y) sits at
  // (x + 4, y + 3): the scanned area starts on a 4-byte boundary of the tile rows); its score sits at t - kScoreOff in the score tile (same pitch, 1-px zero frame), its bit in the bitmap at
  // b = t - kBitOff = y * P + x.  Only phase A, which walks the scanned area linearly, needs a division.
  constexpr int kScoreOff = 2 * P + 3, kBitOff = 3 * P + 4;
  constexpr int kScoreQuads = (CM + 2) * P / 16;      // the score tile in 16-byte pieces
  constexpr int kBitWords = (CM * P + 63) / 64 * 2;   // bitmap words, an even number: the compaction reads them in pairs
  __shared__ __attribute__((aligned(16))) uint32_t s_tile_w[(CM + 6) * P / 4 + 4];  // + slack: the last group of the last row reads one word on
  uint8_t* s_tile = reinterpret_cast<uint8_t*>(s_tile_w);
  __shared__ __attribute__((aligned(16))) uint32_t s_score_w[kScoreQuads * 4];
  uint8_t* s_score = reinterpret_cast<uint8_t*>(s_score_w);
  // survivors of the pre-screen as lists of (t | polarity << 15), one list per wave: a wave reserves the slots of a trip with one
  // scan over its work-items' entry counts and keeps its count in a scalar register - no LDS atomic and no round trip per trip.  Cells with more survivors
  // than a list holds (noise, checkerboards) are scored pixel by pixel, both polarities.  The corners phase B finds go to the
  // front of the same list: a wave compacts its own entries in place (a corner's slot is never behind the entry it came from).
  constexpr int kWaves = BS / 64;
  constexpr int kSurvCap = CM <= kCellSmall ? 1024 : CM * CM, kSurvPerWave = kSurvCap / kWaves;
  __shared__ uint16_t s_surv[kSurvCap];
  __shared__ int s_nsurv[kWaves], s_ncorner[kWaves], s_any;
  __shared__ uint32_t s_keep[kBitWords];

  const int tid = threadIdx.x;
  const int f = xcd_frame();
  const int bx = xcd_item() + cell_begin;  // the launch covers the cells cell_begin .. cell_begin + gridDim.x
  // everything a cell needs in one 32-byte scalar load (the level search, two divisions and the LevelGeom look-ups of a
  // one-cell workgroup were a chain of dependent scalar loads in front of its first pixel request)
  const FastCell C = cells[bx];
  uint32_t* my_cnt = cell_cnt + (size_t)f * cells_frame + bx;
  if (C.skip) {  // ORBextractor.cc:810-822: cells starting too close to the border are skipped
    if (tid == 0) *my_cnt = 0;
    return;
  }
  const int ini_x = C.ini_x, ini_y = C.ini_y, tw = C.tw, th = C.th;
  const int sw = tw - 6, sh = th - 6;  // scanned (candidate) area of cv::FAST on the sub-image
  const uint8_t* img = (C.l == 0) ? img0 + (size_t)f * frame0 : pyr + (size_t)f * pyr_frame + C.img_off;
  const int pitch = (C.l == 0) ? pitch0 : (int)C.pitch;

  // ---- stage the (sw+6) x (sh+6) pixel tile: a lane moves 16 bytes, four (five) lanes a tile row, the wave's pieces land in
  //      LDS back to back = rows of pitch P.  Tile column 0 is the pixel LEFT of the cell's sub-image (ini_x >= 13), so that the
  //      scanned area starts at column 4.  The last piece of a row reads up to 14 bytes past the tile: still inside the image
  //      (the tile ends >= 13 px left of the row's end, and never on the last row).
  const int npix = sw * sh;
  // p / sw == (p * ceil(2^20 / sw)) >> 20 while p * sw < 2^20 (p < 72 * 72, sw <= 72); the product stays below 2^27.
  // 24-bit multiplies: v_mul_u32_u24 issues at the full VALU rate, v_mul_hi_u32 / v_mul_lo_u32 at a quarter of it.
  const uint32_t magic = C.magic;
  // pixel p = y * sw + x of the scanned area -> its tile offset t
#define RGBL_T_OF(p) ((p) + (int)__umul24(__umul24((uint32_t)(p), magic) >> 20, (uint32_t)(P - sw)) + kBitOff)
  {
    const int used = (tw + 1 + 15) >> 4;  // pieces of a row that hold tile bytes (the tile starts one pixel left of the cell's sub-image)
    for (int j0 = 0; j0 < th * kPieces; j0 += BS) {
      const int j = j0 + tid;
      const int y = kPieces == 4 ? j >> 2 : (int)(__umul24((uint32_t)j, kPieces == 3 ? 0x5556u : 0x3334u) >> 16), k = j - y * kPieces;  // j / 3, j / 5 for j < 2^14
      if (y < th && k < used)
        lds_dma16(img + (__umul24((uint32_t)(ini_y + y), (uint32_t)pitch) + (uint32_t)(ini_x - 1 + 16 * k)), s_tile + (j0 + (tid & ~63)) * 16);
    }
  }
#ifdef RGBL_FAST_SKIP
  const int n_passes = 1;  // timing experiments: one pass (at ini_th) whatever it finds
#else
  const int n_passes = ini_th > min_th ? 2 : 1;
#endif
  for (int pass = 0; pass < n_passes; ++pass) {
#ifdef RGBL_FAST_SKIP
    const int thr = ini_th;   // timing experiments: the first pass only
#else
    const int thr = (pass == 0 && n_passes == 2) ? ini_th : min_th;   // this pass = cv::FAST(cell, thr, nonmax suppression)
#endif
    // ---- clear the score tile (the rows in use), the counters and the bitmap
    {
      uint4* q = reinterpret_cast<uint4*>(s_score_w);
      for (int i = tid; i < (sh + 2) * kPieces; i += BS) q[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (tid == 0) s_any = 0;
    for (int i = tid; i < kBitWords; i += BS) s_keep[i] = 0;
    __syncthreads();

    // ---- phase A: cheap necessary condition on the 4 axis/diagonal ring pairs, per polarity; survivors are listed with the
    //      polarity they can have (dark if both - the rare pixel that passes for both gets a second entry for the bright arc).
    //      A work-item tests FOUR pixels of a row: the scanned area starts on a word boundary of the tile, so the 4 centres are
    //      one aligned word and every ring byte has a fixed place in one of 11 aligned words (3 of the row, 3 each of the rows
    //      +-2, 1 each of the rows +-3) - the byte selects fold into the min / max instructions (SDWA).  11 word reads instead
    //      of 36 byte reads per 4 pixels, index arithmetic once per 4.
    const int wv = wave_id();
    {
      uint16_t* mine = s_surv + wv * kSurvPerWave;
      int n_mine = 0;  // wave-uniform
      const int gpr = (sw + 3) >> 2, ngroups = gpr * sh;   // groups of 4 pixels per row / per cell
      const uint32_t gmagic = (0x10000u + (uint32_t)gpr - 1u) / (uint32_t)gpr;  // g / gpr == (g * gmagic) >> 16 for g < 72 * 18, gpr <= 18
#if defined(RGBL_FAST_SKIP) && RGBL_FAST_SKIP >= 2
      for (int g0 = 0; g0 < 0; g0 += BS) {
#else
      // wave w takes the groups 64 w + 64 k BS/64 ..: its loop ends with ITS last group (the 315 groups of a 35 x 35 cell are
      // 5 wave trips, 3 + 2, not 2 x 3)
      for (int g0 = 64 * wv; g0 < ngroups; g0 += BS) {
#endif
        // no divergent region around the tests: a lane past the last group works on that group again with all four pixels
        // masked out (nvalid = 0), so that the flags stay scalar masks - the ballots below cost nothing
        const int gl = g0 + lane_id(), g = imin(gl, ngroups - 1);
        const int gy = (int)(mul24((uint32_t)g, gmagic) >> 16), gx = g - gy * gpr;
        const int wi = (gy + 3) * (P / 4) + 1 + gx;   // word of the group's 4 centres
        const int t0 = 4 * wi;
        const uint32_t* W = s_tile_w + (wi - 3 * (P / 4));   // row - 3
        constexpr int R = P / 4;
        const uint32_t c0 = W[3 * R - 1], c1 = W[3 * R], c2 = W[3 * R + 1];
        const uint32_t u0 = W[5 * R - 1], u1 = W[5 * R], u2 = W[5 * R + 1];   // row + 2
        const uint32_t d0 = W[R - 1], d1 = W[R], d2 = W[R + 1];               // row - 2
        const uint32_t u3 = W[6 * R], d3 = W[0];                              // rows +- 3
        const int nvalid = sw - 4 * gx;  // pixels of the group inside the scanned area (the last group of a row may hold fewer than 4)
        const bool live = gl < ngroups;
        const unsigned long long mlive = wave_ballot(gl < ngroups);
        // Two pixels per instruction: the ring values of the pixels (1, 3) resp. (0, 2) of the group sit in the HIGH bytes of the
        // two 16-bit halves of a register, whatever is in the low bytes (a minimum / maximum of such halves has the minimum /
        // maximum of the high bytes in its high byte) - pixels 1 and 3 of an aligned word ARE such a register, the others
        // cost one shift or v_perm_b32 for two pixels.  The bounds are clean: lo = (v - thr) << 8 saturated at 0, hi =
        // (v + thr) << 8 | 0xff saturated at 0xffff, so that "M < lo" and "m > hi" on the halves compare the high bytes
        // strictly (equal high bytes never pass, whatever the low ones hold).
        bool dk[4], br[4];
        unsigned long long mdk[4], mbr[4];   // the same conditions as wave masks (the comparisons' own scalar results)
        const uint32_t th2 = ((uint32_t)thr << 8) | ((uint32_t)thr << 24);
#pragma unroll
        for (int o = 0; o < 2; ++o) {  // o = 1: pixels 1 (low half) and 3 (high half); o = 0: pixels 0 and 2
          uint32_t c, a0, b0, a1, b1, a2, b2, a3, b3;
          if (o == 1) {
            c = hi_pair<1, true>(c0, c1, c2);
            a0 = hi_pair<1>(0u, u3, 0u); b0 = hi_pair<1>(0u, d3, 0u);                       // ring 0 / 8: (0, +3), (0, -3)
            a1 = hi_pair<4>(c0, c1, c2); b1 = hi_pair<-2>(c0, c1, c2);                      // ring 4 / 12: (+3, 0), (-3, 0)
            a2 = hi_pair<3>(u0, u1, u2); b2 = hi_pair<-1>(d0, d1, d2);                      // ring 2 / 10: (+2, +2), (-2, -2)
            a3 = hi_pair<3>(d0, d1, d2); b3 = hi_pair<-1>(u0, u1, u2);                      // ring 6 / 14: (+2, -2), (-2, +2)
          } else {
            c = hi_pair<0, true>(c0, c1, c2);
            a0 = hi_pair<0>(0u, u3, 0u); b0 = hi_pair<0>(0u, d3, 0u);
            a1 = hi_pair<3>(c0, c1, c2); b1 = hi_pair<-3>(c0, c1, c2);
            a2 = hi_pair<2>(u0, u1, u2); b2 = hi_pair<-2>(d0, d1, d2);
            a3 = hi_pair<2>(d0, d1, d2); b3 = hi_pair<-2>(u0, u1, u2);
          }
          const uint32_t M = pk_max_u16(pk_max_u16(pk_min_u16(a0, b0), pk_min_u16(a1, b1)), pk_max_u16(pk_min_u16(a2, b2), pk_min_u16(a3, b3)));   // every pair has a member below lo
          const uint32_t m = pk_min_u16(pk_min_u16(pk_max_u16(a0, b0), pk_max_u16(a1, b1)), pk_min_u16(pk_max_u16(a2, b2), pk_max_u16(a3, b3)));   // every pair has a member above hi
          const uint32_t lo = pk_subs_u16(c, th2), hi = pk_adds_u16(c, th2 | 0x00ff00ffu);
          dk[o] = (uint16_t)M < (uint16_t)lo; dk[o + 2] = (uint16_t)(M >> 16) < (uint16_t)(lo >> 16);
          br[o] = (uint16_t)m > (uint16_t)hi; br[o + 2] = (uint16_t)(m >> 16) > (uint16_t)(hi >> 16);
          mdk[o] = wave_ballot((uint16_t)M < (uint16_t)lo); mdk[o + 2] = wave_ballot((uint16_t)(M >> 16) < (uint16_t)(lo >> 16));
          mbr[o] = wave_ballot((uint16_t)m > (uint16_t)hi); mbr[o + 2] = wave_ballot((uint16_t)(m >> 16) > (uint16_t)(hi >> 16));
        }
        // A work-item has up to 8 list entries: per pixel one for the dark arc (t) and one for the bright arc (t | 0x8000),
        // for whichever passes.  c[k] = entries among its first k candidates: eight adds of a condition's scalar mask as carry.
        // One reservation per work-item and trip: the counts are scanned over the wave (DPP adds), entry k then goes to
        // (first slot of the work-item) + c[k] - one address and one value operation per entry, the condition is the write's
        // execution mask.  A trip that does not fit the list as a whole writes nothing (uniform) and only counts on: the
        // cell is then scored pixel by pixel.
        bool on[8];
        int c[9];
        c[0] = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool v = live && j < nvalid;
          const unsigned long long mv = mlive & wave_ballot(j < nvalid);
          on[2 * j] = v && dk[j]; on[2 * j + 1] = v && br[j];
          c[2 * j + 1] = add_flag(c[2 * j], mv & mdk[j]);
          c[2 * j + 2] = add_flag(c[2 * j + 1], mv & mbr[j]);
        }
        const int cnt = c[8];
        const int incl = wave_inclusive_scan(cnt), total = wave_last(incl);
        if (n_mine + total <= kSurvPerWave) {
          uint16_t* dst = mine + (n_mine + incl - cnt);
#pragma unroll
          for (int k = 0; k < 8; ++k)
            if (on[k]) dst[c[k]] = (uint16_t)(t0 + (k >> 1) + ((k & 1) << 15));   // t0 + j < 2^15
        }
        n_mine += total;
      }
      if (lane_id() == 0) s_nsurv[wv] = n_mine;
    }
    __syncthreads();

    // ---- phase B: exact score of the listed arcs; the ones that exist at this threshold are corners and are listed
    bool all = false;  // the pre-screen is a necessary condition only: scoring every pixel, both arcs, gives the same corners
#pragma unroll
    for (int w = 0; w < kWaves; ++w) all = all || s_nsurv[w] > kSurvPerWave;
    int cs_mine = s_nsurv[wv];
    int nsurv = all ? 2 * npix : 0;   // all: the pixels' arcs, everybody takes a share
#if defined(RGBL_FAST_SKIP) && RGBL_FAST_SKIP >= 1
    nsurv = 0;
    cs_mine = 0;
#endif
    uint16_t* const my_surv = s_surv + wv * kSurvPerWave;
    {
      int n_mine = 0;
      // A wave scores the entries of its OWN list (as many wave trips in all as over the concatenated lists, and no search for
      // the list an index falls into); `all`: everybody takes its share of all pixels, both arcs each.
      const int n_loop = all ? nsurv : cs_mine, step = all ? BS : 64;
      for (int i0 = 0; i0 < n_loop; i0 += step) {
        // (a lane past the end scores the last entry again and is masked out: no divergent region, the flag stays a scalar mask)
        const int il = i0 + (all ? tid : lane_id()), i = imin(il, n_loop - 1);
        int e;
        if (all) e = RGBL_T_OF(i >> 1) | ((i & 1) << 15);
        else e = my_surv[i];
        const int t = e & 0x7fff;
        const int sc = fast_score_one(s_tile, t, P, e < 0x8000 ? 0xffffffffu : 0x00010001u);
        const bool live = il < n_loop, hit = sc >= thr, corner = live && hit;  // at most one of a pixel's two arcs can exist
        // (the ballots in the block of their comparisons: across a branch the compiler rebuilds a mask from a 0 / 1 register)
        const unsigned long long m = wave_ballot(il < n_loop) & wave_ballot(sc >= thr);
        if (corner) {
          s_score[t - kScoreOff] = (uint8_t)sc;
          const int pos = n_mine + wave_rank(m);
          if (pos < kSurvPerWave) my_surv[pos] = (uint16_t)t;   // pos <= the entry's own index (entries of the list), or the list is not read (all)
        }
        n_mine += (int)__popcll(m);
      }
      if (lane_id() == 0) s_ncorner[wv] = n_mine;
    }
    __syncthreads();

    // ---- phase C: 3x3 strict NMS inside the cell over the pixels that have a score; its survivors set a bit in a
    //      row-major bitmap.  A wave checks the corners it listed itself; a list can only overflow when every pixel was scored
    //      (`all`: more corners than list slots) - everybody then looks at its share of all pixels.
    bool listed = true;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) listed = listed && s_ncorner[w] <= kSurvPerWave;
    const int ncheck = listed ? s_ncorner[wv] : nsurv;
    for (int i = listed ? lane_id() : tid; i < ncheck; i += listed ? 64 : BS) {
      int t;
      if (listed) t = my_surv[i];
      else t = RGBL_T_OF(i >> 1);
      const uint8_t* s = &s_score[t - kScoreOff - (P + 1)];  // the 3x3 neighbourhood's first byte
      const int v = s[P + 1];
      if (v != 0 && v > s[P] && v > s[P + 2] && v > s[0] && v > s[1] && v > s[2] && v > s[2 * P] && v > s[2 * P + 1] && v > s[2 * P + 2]) {
        const int b = t - kBitOff;
        atomicOr(&s_keep[b >> 5], 1u << (b & 31));  // (a pixel listed twice sets its bit twice)
        s_any = 1;
      }
    }
    __syncthreads();
    if (s_any) break;  // ORBextractor.cc:832: the second cv::FAST runs only when the first found nothing
    if (pass + 1 < n_passes) __syncthreads();  // everyone has read s_any before the next pass clears it
  }
#undef RGBL_T_OF
  // Ordered compaction by the first wave alone (no further barrier; the other waves are done): a lane owns 64 bitmap
  // bits, ascending bits = cv::FAST's row-major emission order.  The ordered pixel list goes through LDS (the survivor
  // list's space) so that the keys leave with one coalesced store per 64 keypoints.
  if (wave_id() != 0) return;
  {
    const int lane = lane_id();
    const uint32_t* keep = s_keep;
    const int npairs = (sh * P + 63) >> 6;  // <= 90
    uint16_t* s_list = s_surv;
    uint32_t total = 0;
    for (int w0 = 0; w0 < npairs; w0 += 64) {
      const int t = w0 + lane;
      unsigned long long word = 0;
      if (t < npairs) word = (unsigned long long)keep[2 * t] | ((unsigned long long)keep[2 * t + 1] << 32);
      const uint32_t cnt = (uint32_t)__popcll(word);
      const uint32_t incl = wave_inclusive_scan(cnt);  // DPP adds, no LDS round trips
      uint32_t pos = total + incl - cnt;
      while (word) {
        const int bit = __ffsll((long long)word) - 1;
        word &= word - 1;
        s_list[pos++] = (uint16_t)(t * 64 + bit);
      }
      total += (uint32_t)wave_last((int)incl);
    }
    wave_sync();
    const uint32_t n_out = total < (uint32_t)C.cell_cap ? total : (uint32_t)C.cell_cap;  // cap is a proven bound
    // Dense output (level_cnt != null): the cell reserves its range of the level's candidate list with one atomic; the
    // list is then in whatever order the cells finish, which the quad-tree kernel does not care about (octree_labels.h:
    // the reference's candidate order is a function of the coordinates).  Otherwise: the cell's own slots.
    uint32_t* out = slots + (size_t)f * slots_frame + C.slot_base;
    if (level_cnt) {
      uint32_t base = 0;
      if (lane == 0 && n_out) base = atomicAdd(&level_cnt[(size_t)f * n_levels + C.l], n_out);
      base = (uint32_t)wave_first((int)base);
      out = dense_keys + (size_t)f * keys_frame + geom[C.l].key_off + base;
    }
    for (uint32_t i = lane; i < n_out; i += 64) {
      const int b = s_list[i];
      const int y = b / P, x = b - y * P;
      out[i] = pack_key(C.kx0 + x, C.ky0 + y, s_score[b + P + 1]);
    }
    if (lane == 0) *my_cnt = n_out;
  }
}

// ------------------------------------------------------------------------------------------------
// k_selftest_wrappers: the hand-written instruction wrappers of this file against their plain expressions (the CPU
// emulation replaces them by plain C, so only a run on the hardware says anything about the code that ships): mul24
// (v_mul_u32_u24 with a scalar second operand), add_flag (v_addc_co_u32 with a 64-bit SGPR mask as carry-in), lds_dma16
// (global_load_lds_dwordx4 from unaligned sources), each in waves with some lanes switched off.
__global__ __launch_bounds__(256) void k_selftest_wrappers(const uint32_t* __restrict__ a, uint32_t b, const uint32_t* __restrict__ flags,
                                                            const uint8_t* __restrict__ src, uint32_t* __restrict__ out_mul,
                                                            int* __restrict__ out_add, uint8_t* __restrict__ out_dma, int n) {
  __shared__ __attribute__((aligned(16))) uint8_t s_buf[256 * 16];
  const int tid = threadIdx.x, i = blockIdx.x * 256 + tid;
  const uint32_t av = i < n ? a[i] : 0u;
  const bool active = i < n && (av & 3u) != 0u;    // a quarter of the lanes sits out
  for (int k = tid; k < 256 * 16 / 4; k += 256) reinterpret_cast<uint32_t*>(s_buf)[k] = 0xeeeeeeeeu;
  __syncthreads();
  const unsigned long long m = wave_ballot(i < n && (flags[i < n ? i : 0] & 1u) != 0u);   // computed by the whole wave
  if (active) {
    out_mul[i] = mul24(av & 0xffffffu, b);
    out_add[i] = add_flag((int)av, m);
  }
  if (i < n && (lane_id() % 5) != 4) lds_dma16(src + (size_t)16 * i + (i & 3), s_buf + wave_id() * 1024);   // sources 0 - 3 bytes off alignment
  __syncthreads();
  if (i < n)
    for (int k = 0; k < 16; ++k) out_dma[(size_t)16 * i + k] = s_buf[16 * tid + k];
}

// ------------------------------------------------------------------------------------------------
// k_compact_cells (round 4): the cells' own candidate slots -> the level's dense candidate list, for batches.
// k_fast_cells used to reserve every cell's range of that list with one returning atomicAdd on the (frame, level) counter.
// With the XCD-aware launch a XCD's ~670 resident cells belong to ONE frame, on a 4K frame to one LEVEL: all their atomics
// went to one address, a returning atomic on one address takes ~32 ns, and the FAST kernel ran at the rate of that counter
// (4K: 5.2 ms per 64 frames against 3.2 ms without the reservation; KITTI: 1.33 against 1.27 ms) - every wave holding its
// LDS and registers while it waited for its turn.  Now the FAST cells write their own slots and leave (stores only), and
// this kernel - one workgroup per group of up to 256 consecutive cells of a level - sums the group's counts, reserves the
// group's range with ONE atomic (36 per 4K level-0 frame instead of 9 216) and copies the keys: reads in runs of a cell's
// ~20 keys, writes coalesced.  The quad-tree kernel does not care in which order the groups arrive (octree_labels.h).
// grid = xcd_grid(groups of the launch, frames), block = 256.
struct CellGroup { uint32_t first, count; };   // cells [first, first + count) of the frame's cell table, one level
constexpr int kCompactCells = 256;

__global__ __launch_bounds__(256) void k_compact_cells(const CellGroup* __restrict__ groups, const FastCell* __restrict__ cells,
                                                        const uint32_t* __restrict__ cell_cnt, size_t cells_frame,
                                                        const uint32_t* __restrict__ slots, size_t slots_frame,
                                                        const LevelGeom* __restrict__ geom, int n_levels,
                                                        uint32_t* __restrict__ dense_keys, size_t keys_frame,
                                                        uint32_t* __restrict__ level_cnt, int group_begin) {
  __shared__ uint32_t s_pref[kCompactCells + 1];
  __shared__ uint32_t s_scan[8];
  __shared__ uint32_t s_base;
  const int tid = threadIdx.x, f = xcd_frame();
  const CellGroup G = groups[xcd_item() + group_begin];
  const FastCell first = cells[G.first];   // the group's cells lie back to back in the table and in the slot array
  const uint32_t cnt = tid < (int)G.count ? cell_cnt[(size_t)f * cells_frame + G.first + tid] : 0u;
  uint32_t total;
  const uint32_t ex = block_exclusive_scan<uint32_t>(cnt, s_scan, &total);
  s_pref[tid] = ex;
  if (tid == 0) {
    s_pref[kCompactCells] = total;
    s_base = total ? atomicAdd(&level_cnt[(size_t)f * n_levels + first.l], total) : 0u;
  }
  __syncthreads();
  if (total == 0) return;
  const uint32_t* src = slots + (size_t)f * slots_frame + first.slot_base;
  uint32_t* out = dense_keys + (size_t)f * keys_frame + geom[first.l].key_off + s_base;
  const uint32_t cell_cap = first.cell_cap;
  for (uint32_t i = (uint32_t)tid; i < total; i += 256u) {
    // the cell whose range holds list position i: the last c with pref[c] <= i (pref is non-decreasing, pref[256] = total > i)
    uint32_t lo = 0;
#pragma unroll
    for (uint32_t step = kCompactCells / 2; step >= 1; step >>= 1)
      if (s_pref[lo + step] <= i) lo += step;
    out[i] = src[lo * cell_cap + (i - s_pref[lo])];
  }
}

// ------------------------------------------------------------------------------------------------
// 7x7 Gaussian, sigma 2, OpenCV's 8.8 fixed-point kernel {18,34,48,56,48,34,18}; BORDER_REFLECT_101.
// Tile = 128 x 32 outputs per workgroup.  Pass 1 reads the source rows straight from HBM as three 32-bit
// words per 4 pixels and leaves the horizontal sums (exact 16-bit 8.8 values) in LDS; pass 2 gives every
// work-item a 4 x 4 output block: five 16-byte LDS reads, four 32-bit stores.
// grid = (tiles per frame over all levels, B), block = 256.
constexpr int kBlurTW = 128, kBlurTH = 32;
struct BlurTiles { int tile_off[kMaxLevels + 1]; int tiles_x[kMaxLevels]; };

// The filter runs on the VALU's integer dot products.  Horizontal: a work-item owns 4 output columns of two rows; the
// seven taps of an output are two v_dot4_u32_u8 over byte-aligned windows (v_alignbyte_b32) of the row's three words with
// the weights {18,34,48,56} / {48,34,18,0}.  The exact 16-bit sums of the two rows share a word (row pair j = rows 2j,
// 2j + 1), so the vertical pass is four v_dot2_u32_u16 per output over five such words, the rounding constant being the
// initial accumulator: even output rows take the weights (18,34)(48,56)(48,34)(18,0), odd ones (0,18)(34,48)(56,48)(34,18).
// About 10 VALU instructions per pixel (shifts, masks and multiply-adds on unpacked bytes took 25).
struct GaussTile { uint16_t x0, y0, w, h, pitch; uint8_t l, pad; uint32_t img_off; };  // one output tile, prepared by the host
static_assert(sizeof(GaussTile) == 16, "one s_load_dwordx4 per tile");

template <int BS>
__global__ __launch_bounds__(BS) void k_gauss7(const GaussTile* __restrict__ tiles,
                                                    const uint8_t* __restrict__ img0, int pitch0, size_t frame0,
                                                    const uint8_t* __restrict__ pyr, size_t pyr_frame,
                                                    uint8_t* __restrict__ blur, size_t blur_frame, int tile_begin) {
  constexpr int kPairs = (kBlurTH + 6) / 2;       // 19 row pairs
  __shared__ uint32_t s_p[kPairs * kBlurTW];      // [pair][column]: sum of row 2j | sum of row 2j + 1 << 16
  const int f = xcd_frame();
  const int tid = threadIdx.x, bx = xcd_item() + tile_begin;
  const GaussTile T = tiles[bx];  // level, origin and geometry in one scalar load (no level search, no division)
  const int x0 = T.x0, y0 = T.y0;
  const uint8_t* img = (T.l == 0) ? img0 + (size_t)f * frame0 : pyr + (size_t)f * pyr_frame + T.img_off;
  const int pitch = (T.l == 0) ? pitch0 : (int)T.pitch;
  const int W = T.w, H = T.h;
  const uint32_t kWA = 18u | (34u << 8) | (48u << 16) | (56u << 24), kWB = 48u | (34u << 8) | (18u << 16);

  // one row pair x four columns: seven-tap sums of both rows from their three words each, stored as one 128-bit LDS write
  auto h_sums = [&](const uint32_t (&w)[2][3], int j, int cg) {
    uint32_t hs[2][4];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {  // output x + i: pixels x+i-3 .. x+i+3 = bytes i+1 .. i+7 of the 12
        const uint32_t lo = i == 3 ? w[rr][1] : align_bytes(w[rr][1], w[rr][0], i + 1);
        const uint32_t hi = i == 3 ? w[rr][2] : align_bytes(w[rr][2], w[rr][1], i + 1);
        hs[rr][i] = udot4(lo, kWA, udot4(hi, kWB, 0u));
      }
    }
    uint4 v;
    v.x = hs[0][0] | (hs[1][0] << 16); v.y = hs[0][1] | (hs[1][1] << 16);
    v.z = hs[0][2] | (hs[1][2] << 16); v.w = hs[0][3] | (hs[1][3] << 16);
    *reinterpret_cast<uint4*>(&s_p[j * kBlurTW + 4 * cg]) = v;
  };
  // interior column groups: all six words requested before the first use.  The groups that touch the left or right border
  // of the level are left to a second, compacted pass - inside this loop two lanes per wave would drag the other 62 through
  // the byte-wise path in every iteration of every tile in the first and last tile column.
  const bool rows_inside = y0 >= 3 && y0 + kBlurTH + 3 <= H;
  for (int task = tid; task < kPairs * 32; task += BS) {
    const int j = task >> 5, cg = task & 31;
    const int x = x0 + 4 * cg;
    if (x < 4 || x + 8 > W) continue;
    // tiles whose rows y0 - 3 .. y0 + 34 all exist (every tile but the first and last tile row of a level) skip the reflection
    int ya = y0 + 2 * j - 3, yb = ya + 1;
    if (!rows_inside) { ya = reflect101(ya, H); yb = reflect101(yb, H); }   // uniform
    const uint8_t* row0 = img + __umul24((uint32_t)ya, (uint32_t)pitch) + x;
    const uint8_t* row1 = img + __umul24((uint32_t)yb, (uint32_t)pitch) + x;
    uint32_t w[2][3];
    w[0][0] = load_u32_unaligned(row0 - 4); w[0][1] = load_u32_unaligned(row0); w[0][2] = load_u32_unaligned(row0 + 4);
    w[1][0] = load_u32_unaligned(row1 - 4); w[1][1] = load_u32_unaligned(row1); w[1][2] = load_u32_unaligned(row1 + 4);
    h_sums(w, j, cg);
  }
  // border column groups of this tile: group 0 of the first tile column, and the groups with x + 8 > W (at most two)
  {
    const int cg_last = imin(31, (W - 1 - x0) >> 2);                    // last group that holds a pixel of the level
    const int cg_r0 = imax(0, imin(cg_last + 1, ((W - 8 - x0) >> 2) + 1));  // first group with x + 8 > W (W >= 8 always)
    const int n_right = (x0 + 4 * cg_r0 + 8 > W) ? cg_last - cg_r0 + 1 : 0;
    const int n_left = (x0 == 0 && cg_r0 > 0) ? 1 : 0;                  // x = 0 < 4 (when it is not a right-border group too)
    const int n_edge = n_left + n_right;
    for (int task = tid; task < kPairs * n_edge; task += BS) {
      const int j = task / n_edge, e = task - j * n_edge;
      const int cg = e < n_left ? 0 : cg_r0 + (e - n_left);
      const int x = x0 + 4 * cg;
      uint32_t w[2][3];
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const uint8_t* row = img + __umul24((uint32_t)reflect101(y0 + 2 * j + rr - 3, H), (uint32_t)pitch);
        w[rr][0] = w[rr][1] = w[rr][2] = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          w[rr][0] |= (uint32_t)row[reflect101(x - 4 + k, W)] << (8 * k);
          w[rr][1] |= (uint32_t)row[reflect101(x + k, W)] << (8 * k);
          w[rr][2] |= (uint32_t)row[reflect101(x + 4 + k, W)] << (8 * k);
        }
      }
      h_sums(w, j, cg);
    }
  }
  __syncthreads();
  for (int item = tid; item < 256; item += BS) {  // 32 column groups x 8 row groups
  const int cg = item & 31, rg = item >> 5;
  const int x = x0 + 4 * cg;
  if (x >= W) continue;
  uint32_t pv[5][4];  // row pairs 2 rg .. 2 rg + 4 = the tile rows 4 rg .. 4 rg + 9
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const uint4 v = *reinterpret_cast<const uint4*>(&s_p[(2 * rg + j) * kBlurTW + 4 * cg]);
    pv[j][0] = v.x; pv[j][1] = v.y; pv[j][2] = v.z; pv[j][3] = v.w;
  }
  const uint32_t kE0 = 18u | (34u << 16), kE1 = 48u | (56u << 16), kE2 = 48u | (34u << 16), kE3 = 18u;
  const uint32_t kO0 = 18u << 16, kO1 = 34u | (48u << 16), kO2 = 56u | (48u << 16), kO3 = 34u | (18u << 16);
  // the row pitch is a multiple of 64 and x a multiple of 4: a full word always fits the row (what lands in the padding right
  // of the last column is never read), so no byte-wise tail; acc < 2^24 and its byte 2 is the pixel: one v_perm_b32 per pixel
  uint8_t* D = blur + (size_t)f * blur_frame + T.img_off + (__umul24((uint32_t)(y0 + 4 * rg), (uint32_t)T.pitch) + (uint32_t)x);
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    if (y0 + 4 * rg + o >= H) break;
    const int b = o >> 1;  // first row pair of the window
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t acc = 32768u;
      if ((o & 1) == 0) {
        acc = udot2(pv[b][i], kE0, acc); acc = udot2(pv[b + 1][i], kE1, acc);
        acc = udot2(pv[b + 2][i], kE2, acc); acc = udot2(pv[b + 3][i], kE3, acc);
      } else {
        acc = udot2(pv[b][i], kO0, acc); acc = udot2(pv[b + 1][i], kO1, acc);
        acc = udot2(pv[b + 2][i], kO2, acc); acc = udot2(pv[b + 3][i], kO3, acc);
      }
      out = i == 0 ? acc >> 16 : perm_bytes(acc, out, i == 1 ? 0x0c0c0600u : i == 2 ? 0x0c060100u : 0x06020100u);
    }
    *reinterpret_cast<uint32_t*>(D + __umul24((uint32_t)o, (uint32_t)T.pitch)) = out;
  }
  }
}

// ------------------------------------------------------------------------------------------------
// libstdc++ std::sort (GCC 11 introsort) re-stated on two parallel arrays, ordering = ascending key.
// The reference sorts its expandable quad-tree nodes with std::sort + compareNodes (ORBextractor.cc:538-553,
// :700); nodes with equal (size, UL.x) end up in an order that only the algorithm's exact sequence of
// swaps determines, and that order decides which nodes get split.  Hence a literal restatement.
struct SortPair { uint64_t k; uint32_t v; };
struct SortView {
  uint64_t* key; uint32_t* val;
  __host__ __device__ SortPair get(int i) const { SortPair p; p.k = key[i]; p.v = val[i]; return p; }
  __host__ __device__ void set(int i, SortPair p) const { key[i] = p.k; val[i] = p.v; }
  __host__ __device__ void swap(int i, int j) const { SortPair a = get(i), b = get(j); set(i, b); set(j, a); }
  __host__ __device__ bool less(int i, int j) const { return key[i] < key[j]; }
};
__host__ __device__ inline void ss_push_heap(const SortView& a, int first, int hole, int top, SortPair value) {
  int parent = (hole - 1) / 2;
  while (hole > top && a.key[first + parent] < value.k) {
    a.set(first + hole, a.get(first + parent));
    hole = parent;
    parent = (hole - 1) / 2;
  }
  a.set(first + hole, value);
}
__host__ __device__ inline void ss_adjust_heap(const SortView& a, int first, int hole, int len, SortPair value) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (a.less(first + child, first + child - 1)) --child;
    a.set(first + hole, a.get(first + child));
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    a.set(first + hole, a.get(first + child - 1));
    hole = child - 1;
  }
  ss_push_heap(a, first, hole, top, value);
}
__host__ __device__ inline void ss_heap_sort(const SortView& a, int first, int last) {
  const int len = last - first;
  if (len >= 2) {
    int parent = (len - 2) / 2;
    for (;;) {
      SortPair v = a.get(first + parent);
      ss_adjust_heap(a, first, parent, len, v);
      if (parent == 0) break;
      --parent;
    }
  }
  while (last - first > 1) {
    --last;
    SortPair v = a.get(last);
    a.set(last, a.get(first));
    ss_adjust_heap(a, first, 0, last - first, v);
  }
}
__host__ __device__ inline void ss_unguarded_linear_insert(const SortView& a, int last) {
  SortPair v = a.get(last);
  int next = last - 1;
  while (v.k < a.key[next]) {
    a.set(last, a.get(next));
    last = next;
    --next;
  }
  a.set(last, v);
}
__host__ __device__ inline void ss_insertion_sort(const SortView& a, int first, int last) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (a.less(i, first)) {
      SortPair v = a.get(i);
      for (int j = i; j > first; --j) a.set(j, a.get(j - 1));
      a.set(first, v);
    } else {
      ss_unguarded_linear_insert(a, i);
    }
  }
}
__host__ __device__ inline void std_sort_restated(uint64_t* key, uint32_t* val, int n) {
  if (n <= 0) return;
  SortView a{key, val};
  int lg = 0;
  while ((n >> (lg + 1)) != 0) ++lg;
  int stack_first[64], stack_last[64], stack_depth[64], sp = 0;
  stack_first[0] = 0; stack_last[0] = n; stack_depth[0] = 2 * lg; sp = 1;
  while (sp > 0) {
    --sp;
    int first = stack_first[sp], last = stack_last[sp], depth = stack_depth[sp];
    while (last - first > 16) {
      if (depth == 0) { ss_heap_sort(a, first, last); break; }
      --depth;
      // __move_median_to_first(first, first+1, mid, last-1)
      const int mid = first + (last - first) / 2;
      const int A = first + 1, B = mid, C = last - 1;
      if (a.less(A, B)) {
        if (a.less(B, C)) a.swap(first, B);
        else if (a.less(A, C)) a.swap(first, C);
        else a.swap(first, A);
      } else if (a.less(A, C)) a.swap(first, A);
      else if (a.less(B, C)) a.swap(first, C);
      else a.swap(first, B);
      // __unguarded_partition(first+1, last, pivot = first)
      int lo = first + 1, hi = last;
      const uint64_t pivot = a.key[first];
      for (;;) {
        while (a.key[lo] < pivot) ++lo;
        --hi;
        while (pivot < a.key[hi]) --hi;
        if (!(lo < hi)) break;
        a.swap(lo, hi);
        ++lo;
      }
      stack_first[sp] = lo; stack_last[sp] = last; stack_depth[sp] = depth; ++sp;  // right part later
      last = lo;
    }
  }
  if (n > 16) {
    ss_insertion_sort(a, 0, 16);
    for (int i = 16; i < n; ++i) ss_unguarded_linear_insert(a, i);
  } else {
    ss_insertion_sort(a, 0, n);
  }
}

// ------------------------------------------------------------------------------------------------
// Quad-tree distribution, one workgroup (4 waves) per (level, frame).
struct OctreeBufs {
  const uint32_t* cell_cnt; size_t cells_frame;
  const uint32_t* slots; size_t slots_frame;
  uint32_t *keys_a, *keys_b; size_t keys_frame;
  QNode *list_a, *list_b; QDiv* div; uint32_t *todo_a, *todo_b; uint64_t* skey; uint32_t* sval;
  uint8_t* divided; size_t nodes_frame;
  const uint8_t* rootx;
  uint32_t* kp_key; int* kp_count; size_t kp_frame;  // outputs: selected keys per level, counts [B][L]
  int* err;
  uint32_t* level_cnt;        // [B][L] candidates per level when k_fast_cells wrote them densely (keys_a, any cell order); null: cell slots.
                              // The quad-tree workgroup of a (frame, level) reads its counter, leaves it at zero for the next
                              // extraction and keeps a copy in level_cnt_last (rgbl_extractor_get_candidates)
  uint32_t* level_cnt_last;
  unsigned long long* dbg;  // optional: 16 cycle-counter stamps per (frame, level) workgroup (diagnostics)
  int no_hist;              // RGBL_OCTREE_HIST=0: the breadth-first phase round by round over the keys (rounds 1 - 3) instead of on the cell pyramid
};

__device__ __forceinline__ int quadrant_of(uint32_t key, int mx, int my) {
  const bool left = key_x(key) < mx, top = key_y(key) < my;
  return left ? (top ? 0 : 2) : (top ? 1 : 3);
}
// wave-cooperative: counts of the node's keys per child quadrant (all lanes get the result)
__device__ __forceinline__ void node_count(const QNode& nd, const uint32_t* keys_a, const uint32_t* keys_b, uint32_t c[4]) {
  const uint32_t cnt = nd.cnt & 0x7fffffffu;
  const uint32_t* src = ((nd.cnt >> 31) ? keys_b : keys_a) + nd.beg;
  const int mx = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1), my = nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1);
  c[0] = c[1] = c[2] = c[3] = 0;
  const int lane = lane_id();
  for (uint32_t base = 0; base < cnt; base += 64) {
    const bool valid = base + lane < cnt;
    const int q = valid ? quadrant_of(src[base + lane], mx, my) : -1;
#pragma unroll
    for (int k = 0; k < 4; ++k) c[k] += (uint32_t)__popcll(__ballot(q == k));
  }
}
// wave-cooperative stable 4-way partition of the node's keys into the other buffer
__device__ __forceinline__ void node_place(const QNode& nd, uint32_t* keys_a, uint32_t* keys_b, const uint32_t c[4]) {
  const uint32_t cnt = nd.cnt & 0x7fffffffu;
  const bool in_b = (nd.cnt >> 31) != 0;
  const uint32_t* src = (in_b ? keys_b : keys_a) + nd.beg;
  uint32_t* dst = (in_b ? keys_a : keys_b) + nd.beg;
  const int mx = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1), my = nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1);
  uint32_t o[4] = {0, c[0], c[0] + c[1], c[0] + c[1] + c[2]};
  const int lane = lane_id();
  const unsigned long long lt = lanemask_lt();
  for (uint32_t base = 0; base < cnt; base += 64) {
    const bool valid = base + lane < cnt;
    const uint32_t key = valid ? src[base + lane] : 0u;
    const int q = valid ? quadrant_of(key, mx, my) : -1;
    uint32_t my_pos = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned long long m = __ballot(q == k);
      if (q == k) my_pos = o[k] + (uint32_t)__popcll(m & lt);
      o[k] += (uint32_t)__popcll(m);
    }
    if (valid) dst[my_pos] = key;
  }
}
// first 64 keys of a node, one per lane (0 beyond the end)
__device__ __forceinline__ uint32_t node_first_keys(const QNode& nd, const uint32_t* keys_a, const uint32_t* keys_b) {
  const uint32_t cnt = nd.cnt & 0x7fffffffu;
  const uint32_t* src = ((nd.cnt >> 31) ? keys_b : keys_a) + nd.beg;
  return (uint32_t)lane_id() < cnt ? src[lane_id()] : 0u;
}
// count + place of a node whose first 64 keys (key0) are already in registers: nodes of at most 64 keys - nearly all of
// them after the first two rounds - are split without touching their keys again
__device__ __forceinline__ void node_split(const QNode& nd, uint32_t key0, uint32_t* keys_a, uint32_t* keys_b, uint32_t c[4]) {
  const uint32_t cnt = nd.cnt & 0x7fffffffu;
  if (cnt > 64) {
    node_count(nd, keys_a, keys_b, c);
    node_place(nd, keys_a, keys_b, c);
    return;
  }
  uint32_t* dst = (((nd.cnt >> 31) != 0) ? keys_a : keys_b) + nd.beg;
  const int mx = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1), my = nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1);
  const int lane = lane_id();
  const bool valid = (uint32_t)lane < cnt;
  const int q = valid ? quadrant_of(key0, mx, my) : -1;
  const unsigned long long lt = lanemask_lt();
  unsigned long long m[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { m[k] = __ballot(q == k); c[k] = (uint32_t)__popcll(m[k]); }
  const uint32_t o[4] = {0, c[0], c[0] + c[1], c[0] + c[1] + c[2]};
  if (valid) dst[o[q] + (uint32_t)__popcll(m[q] & lt)] = key0;
}
// the two halves separately (the quota phase counts every expandable node first and places only some of them)
__device__ __forceinline__ void node_count_k0(const QNode& nd, uint32_t key0, const uint32_t* keys_a, const uint32_t* keys_b, uint32_t c[4]) {
  const uint32_t cnt = nd.cnt & 0x7fffffffu;
  if (cnt > 64) { node_count(nd, keys_a, keys_b, c); return; }
  const int mx = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1), my = nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1);
  const int q = (uint32_t)lane_id() < cnt ? quadrant_of(key0, mx, my) : -1;
#pragma unroll
  for (int k = 0; k < 4; ++k) c[k] = (uint32_t)__popcll(__ballot(q == k));
}
__device__ __forceinline__ void node_place_k0(const QNode& nd, uint32_t key0, uint32_t* keys_a, uint32_t* keys_b, const uint32_t c[4]) {
  const uint32_t cnt = nd.cnt & 0x7fffffffu;
  if (cnt > 64) { node_place(nd, keys_a, keys_b, c); return; }
  uint32_t* dst = (((nd.cnt >> 31) != 0) ? keys_a : keys_b) + nd.beg;
  const int mx = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1), my = nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1);
  const bool valid = (uint32_t)lane_id() < cnt;
  const int q = valid ? quadrant_of(key0, mx, my) : -1;
  const unsigned long long lt = lanemask_lt();
  const uint32_t o[4] = {0, c[0], c[0] + c[1], c[0] + c[1] + c[2]};
  uint32_t my_pos = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned long long m = __ballot(q == k);
    if (q == k) my_pos = o[k] + (uint32_t)__popcll(m & lt);
  }
  if (valid) dst[my_pos] = key0;
}
// Wave-strided walk over nodes with the dependent loads taken off the critical path: while node t is processed, the
// descriptor of node t + 2 strides and the keys of node t + 1 stride are already in flight (a node costs three dependent
// global loads otherwise: position -> descriptor -> keys).  pos_of(t) = list position of the t-th node; body(t, node, key0).
template <class PosOf, class Body>
__device__ __forceinline__ void for_nodes_pipelined(const QNode* cur, int first, int count, int stride, const uint32_t* keys_a,
                                                    const uint32_t* keys_b, PosOf&& pos_of, Body&& body) {
  QNode nd_a, nd_b;
  nd_a.x0 = nd_a.x1 = nd_a.y0 = nd_a.y1 = 0; nd_a.beg = 0; nd_a.cnt = 0;
  nd_b = nd_a;
  if (first < count) nd_a = cur[pos_of(first)];
  if (first + stride < count) nd_b = cur[pos_of(first + stride)];
  uint32_t key_a = first < count ? node_first_keys(nd_a, keys_a, keys_b) : 0u;
  for (int t = first; t < count; t += stride) {
    QNode nd_c = nd_a;
    if (t + 2 * stride < count) nd_c = cur[pos_of(t + 2 * stride)];
    const uint32_t key_b = t + stride < count ? node_first_keys(nd_b, keys_a, keys_b) : 0u;
    body(t, nd_a, key_a);
    nd_a = nd_b; key_a = key_b; nd_b = nd_c;
  }
}

__device__ __forceinline__ QNode child_node(const QNode& nd, int q, uint32_t beg, uint32_t cnt) {
  const int mx = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1), my = nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1);
  QNode c;
  c.x0 = (uint16_t)((q & 1) ? mx : nd.x0);
  c.x1 = (uint16_t)((q & 1) ? nd.x1 : mx);
  c.y0 = (uint16_t)((q & 2) ? my : nd.y0);
  c.y1 = (uint16_t)((q & 2) ? nd.y1 : my);
  c.beg = beg;
  c.cnt = cnt | ((nd.cnt & 0x80000000u) ^ 0x80000000u);  // children live in the other buffer
  return c;
}

// Whole-workgroup version of node_count + node_place for very large nodes (4K frames start with two root nodes of
// ~75 k keys each; one wave per node would leave the rest of the workgroup idle).  Counting is work-item strided
// with one reduction; placement walks the keys in chunks of one workgroup, the stable rank of a key inside its
// quadrant = running total + counts of the earlier waves of the chunk (LDS) + ballot rank inside the wave.
constexpr int kCoopMin = 2048;  // nodes with at least this many keys are split by the whole workgroup

template <int BS>
__device__ __forceinline__ void block_split(const QNode& nd, uint32_t* keys_a, uint32_t* keys_b, uint32_t c[4],
                                            unsigned long long* s_scan, uint32_t (*s_wcnt)[4]) {
  const int tid = threadIdx.x, lane = lane_id(), wave = wave_id();
  constexpr int NW = BS / 64;
  const uint32_t cnt = nd.cnt & 0x7fffffffu;
  const bool in_b = (nd.cnt >> 31) != 0;
  const uint32_t* src = (in_b ? keys_b : keys_a) + nd.beg;
  uint32_t* dst = (in_b ? keys_a : keys_b) + nd.beg;
  const int mx = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1), my = nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1);
  unsigned long long c01 = 0, c23 = 0;
  for (uint32_t i = tid; i < cnt; i += BS) {
    const int q = quadrant_of(src[i], mx, my);
    c01 += q == 0 ? 1ull : (q == 1 ? (1ull << 32) : 0ull);
    c23 += q == 2 ? 1ull : (q == 3 ? (1ull << 32) : 0ull);
  }
  unsigned long long t01, t23;
  block_exclusive_scan<unsigned long long>(c01, s_scan, &t01);
  block_exclusive_scan<unsigned long long>(c23, s_scan, &t23);
  c[0] = (uint32_t)(t01 & 0xffffffffu); c[1] = (uint32_t)(t01 >> 32);
  c[2] = (uint32_t)(t23 & 0xffffffffu); c[3] = (uint32_t)(t23 >> 32);
  uint32_t o[4] = {0, c[0], c[0] + c[1], c[0] + c[1] + c[2]};  // running write position per quadrant
  const unsigned long long lt = lanemask_lt();
  for (uint32_t base = 0; base < cnt; base += BS) {
    const uint32_t i = base + tid;
    const bool valid = i < cnt;
    const uint32_t key = valid ? src[i] : 0u;
    const int q = valid ? quadrant_of(key, mx, my) : -1;
    uint32_t rank_in_wave = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned long long mask = __ballot(q == k);
      if (q == k) rank_in_wave = (uint32_t)__popcll(mask & lt);
      if (lane == 0) s_wcnt[wave][k] = (uint32_t)__popcll(mask);
    }
    __syncthreads();
    uint32_t before = 0, chunk_total[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      for (int w = 0; w < NW; ++w) {
        const uint32_t v = s_wcnt[w][k];
        if (k == q && w < wave) before += v;
        chunk_total[k] += v;
      }
    if (valid) dst[o[q] + before + rank_in_wave] = key;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] += chunk_total[k];
    __syncthreads();
  }
}

// work-items per quad-tree workgroup: a (level, frame) problem is one dependent chain, so a wider group shortens the
// chain (fewer strides per pass) while a narrower one lets more problems share a CU (VGPRs allow 4 waves per SIMD):
// kOctWide when the batch cannot fill the chip anyway, kOctNarrow when there are plenty of (level, frame) problems.
constexpr int kOctWide = 512, kOctNarrow = 256;
constexpr int kOctKeysLds = 12288;   // single-frame quad-tree: candidate lists up to this long live in LDS (48 + 24 KB next to the 37 KB of nodes)
constexpr int kSortLds = 2048;     // largest expandable-node list sorted in LDS by the whole workgroup
constexpr int kSortRanges = 160;   // > kSortLds / 17: pending ranges of more than 16 elements are disjoint

template <int R>
struct SortRangesT { int first[R], last[R], depth[R]; };
using SortRanges = SortRangesT<kSortRanges>;

// Workgroup version of std_sort_restated().  libstdc++'s introsort partitions disjoint ranges independently, so
// every pending range is partitioned by its own work-item (rounds = recursion depth), and the closing insertion
// sort never moves an element out of its <= 16-element leaf (left part <= pivot <= right part), i.e. it is a
// stable sort of every leaf: done here as a rank computation, one work-item per element.
// Elements are single 64-bit words: sort key in bits 63..16, payload (list position) in bits 15..0; only the key
// takes part in comparisons.  w may be read up to 4 entries outside [0, m) (padding required on both sides).
__device__ __forceinline__ bool pk_less(uint64_t a, uint64_t b) { return (a >> 16) < (b >> 16); }

__device__ __forceinline__ void pk_push_heap(uint64_t* w, int first, int hole, int top, uint64_t value) {
  int parent = (hole - 1) / 2;
  while (hole > top && pk_less(w[first + parent], value)) {
    w[first + hole] = w[first + parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  w[first + hole] = value;
}
__device__ __forceinline__ void pk_adjust_heap(uint64_t* w, int first, int hole, int len, uint64_t value) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (pk_less(w[first + child], w[first + child - 1])) --child;
    w[first + hole] = w[first + child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    w[first + hole] = w[first + child - 1];
    hole = child - 1;
  }
  pk_push_heap(w, first, hole, top, value);
}
__device__ __forceinline__ void pk_heap_sort(uint64_t* w, int first, int last) {
  const int len = last - first;
  if (len >= 2) {
    int parent = (len - 2) / 2;
    for (;;) {
      pk_adjust_heap(w, first, parent, len, w[first + parent]);
      if (parent == 0) break;
      --parent;
    }
  }
  while (last - first > 1) {
    --last;
    const uint64_t v = w[last];
    w[last] = w[first];
    pk_adjust_heap(w, first, 0, last - first, v);
  }
}

// Hoare's partition visits every element once and never revisits a swapped slot, so its outcome is a function of
// the ORIGINAL range: with L = ascending positions whose key is >= pivot and R = descending positions whose key
// is <= pivot, it swaps the pairs (L_i, R_i) while L_i < R_i and returns the first unswapped stop of the left
// scan.  A wave therefore partitions a range with two ballot-compacted position lists and parallel swaps instead
// of a serial scan; the four waves of the workgroup take different ranges of the same recursion depth.
// seg_first / seg_last double as the L / R lists of the range being partitioned (a leaf is only labelled once
// its range is final).
template <int BS, int CAP = kSortLds, class Ranges = SortRanges>
__device__ __forceinline__ void block_sort_restated(uint64_t* w, int m, uint16_t* seg_first, uint16_t* seg_last,
                                                    Ranges* ra, Ranges* rb, int* s_cnt) {
  const int tid = threadIdx.x, lane = lane_id(), wave = wave_id();
  const int nw = (int)(blockDim.x >> 6);
  const unsigned long long lt = lanemask_lt();
  if (tid == 0) {
    s_cnt[0] = s_cnt[1] = 0;
    if (m > 16) {
      int lg = 0;
      while ((m >> (lg + 1)) != 0) ++lg;
      ra->first[0] = 0; ra->last[0] = m; ra->depth[0] = 2 * lg;
      s_cnt[0] = 1;
    }
  }
  if (m <= 16)
    for (int i = tid; i < m; i += BS) { seg_first[i] = 0; seg_last[i] = (uint16_t)m; }
  __syncthreads();
  Ranges* cur = ra;
  Ranges* nxt = rb;
  int ci = 0;
  for (;;) {
    const int nr = s_cnt[ci];
    if (nr == 0) break;
    for (int r = wave; r < nr; r += nw) {
      const int first = cur->first[r], last = cur->last[r];
      int depth = cur->depth[r];
      if (depth == 0) {  // __partial_sort(first, last, last): heap sort, the range is final afterwards
        if (lane == 0) pk_heap_sort(w, first, last);
        for (int i = first + lane; i < last; i += 64) { seg_first[i] = (uint16_t)i; seg_last[i] = (uint16_t)(i + 1); }
        continue;
      }
      --depth;
      // __move_median_to_first(first, first+1, mid, last-1), evaluated redundantly by every lane
      const int A = first + 1, B = first + (last - first) / 2, C = last - 1;
      const uint64_t wf = w[first], wa = w[A], wb = w[B], wc = w[C];
      int X;
      if (pk_less(wa, wb)) X = pk_less(wb, wc) ? B : (pk_less(wa, wc) ? C : A);
      else X = pk_less(wa, wc) ? A : (pk_less(wb, wc) ? C : B);
      const uint64_t pivot = X == A ? wa : (X == B ? wb : wc);
      wave_sync();
      if (lane == 0) { w[first] = pivot; w[X] = wf; }
      wave_sync();
      // L: ascending positions in (first, last) with !(key < pivot)
      int cnt_l = 0;
      for (int base = first + 1; base < last; base += 64) {
        const int p = base + lane;
        const bool f = p < last && !pk_less(w[p < last ? p : first], pivot);
        const unsigned long long mask = __ballot(f);
        if (f) seg_first[first + cnt_l + __popcll(mask & lt)] = (uint16_t)p;
        cnt_l += __popcll(mask);
      }
      // R: descending positions in (first, last) with !(pivot < key)
      int cnt_r = 0;
      for (int top = last - 1; top > first; top -= 64) {
        const int p = top - lane;
        const bool f = p > first && !pk_less(pivot, w[p > first ? p : first]);
        const unsigned long long mask = __ballot(f);
        if (f) seg_last[first + cnt_r + __popcll(mask & lt)] = (uint16_t)p;
        cnt_r += __popcll(mask);
      }
      wave_sync();
      // number of swaps: pairs with L_i < R_i (a prefix, both lists are monotone)
      const int lim = imin(cnt_l, cnt_r);
      int nswap = 0;
      for (int base = 0; base < lim; base += 64) {
        const int i = base + lane;
        const bool ok = i < lim && seg_first[first + (i < lim ? i : 0)] < seg_last[first + (i < lim ? i : 0)];
        const int c = __popcll(__ballot(ok));
        nswap += c;
        if (c < 64) break;
      }
      for (int i = lane; i < nswap; i += 64) {
        const int pl = seg_first[first + i], pr = seg_last[first + i];
        const uint64_t x = w[pl], y = w[pr];
        w[pl] = y;
        w[pr] = x;
      }
      int cut;
      if (nswap > 0) {
        const int pr = seg_last[first + nswap - 1];
        const int pl = nswap < cnt_l ? (int)seg_first[first + nswap] : 0x7fffffff;
        cut = pl < pr ? pl : pr;
      } else {
        cut = seg_first[first];
      }
      wave_sync();  // every lane is done with the L / R lists before leaves are labelled over them
      const int sub_first[2] = {first, cut}, sub_last[2] = {cut, last};
      for (int k = 0; k < 2; ++k) {
        const int f0 = sub_first[k], l0 = sub_last[k];
        if (l0 - f0 > 16) {
          if (lane == 0) {
            const int slot = atomicAdd(&s_cnt[1 - ci], 1);
            nxt->first[slot] = f0; nxt->last[slot] = l0; nxt->depth[slot] = depth;
          }
        } else {
          for (int i = f0 + lane; i < l0; i += 64) { seg_first[i] = (uint16_t)f0; seg_last[i] = (uint16_t)l0; }
        }
      }
    }
    __syncthreads();
    if (tid == 0) s_cnt[ci] = 0;
    { Ranges* t = cur; cur = nxt; nxt = t; }
    ci ^= 1;
    __syncthreads();
  }
  // stable sort of every leaf (== __final_insertion_sort)
  constexpr int kPer = (CAP + BS - 1) / BS;
  uint64_t mine[kPer];
  int dest[kPer];
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int i = tid + BS * k;
    dest[k] = -1;
    if (i < m) {
      const int f0 = seg_first[i], l0 = seg_last[i];
      const uint64_t kk = w[i];
      int rank = 0;
      for (int j = f0; j < l0; ++j) {
        const uint64_t kj = w[j];
        rank += (pk_less(kj, kk) || (!pk_less(kk, kj) && j < i)) ? 1 : 0;
      }
      mine[k] = kk;
      dest[k] = f0 + rank;
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kPer; ++k)
    if (dest[k] >= 0) w[dest[k]] = mine[k];
  __syncthreads();
}

// Rebuilds the node list after a set of nodes has been split, reproducing std::list push_front/erase:
//   new list = [children of the LAST processed node (n4..n1), ..., children of the FIRST processed node]
//              ++ [all nodes that were not split, in their old order]
// `proc(rho)` gives the old list position of the rho-th processed node (rho < P); div[rho] its child counts
// (all zero = node was not split).  Children with more than one key are appended to todo_out in creation
// order.  Returns (through LDS) the new size and the number of expandable children.
template <bool kIdentity, int BS>
__device__ __forceinline__ void rebuild_list(const QNode* cur, QNode* nxt, int n, const QDiv* div, int P,
                                             const uint32_t* sval, int m, uint8_t* divided, uint32_t* todo_out,
                                             unsigned long long* s_scan, int* s_newn, int* s_nexp) {
  const int tid = threadIdx.x;
  // pass 1: totals over processing ranks (children | expandable children << 32)
  unsigned long long carry = 0;
  // two sweeps: first totals, then placement (placement needs T)
  unsigned long long T2 = 0;
  for (int r0 = 0; r0 < P; r0 += BS) {
    const int rho = r0 + tid;
    unsigned long long v = 0;
    if (rho < P) {
      const QDiv d = div[rho];
      for (int q = 0; q < 4; ++q) v += (d.c[q] > 0 ? 1ull : 0ull) + (d.c[q] > 1 ? (1ull << 32) : 0ull);
    }
    unsigned long long tot;
    block_exclusive_scan<unsigned long long>(v, s_scan, &tot);
    T2 += tot;
  }
  const uint32_t T = (uint32_t)(T2 & 0xffffffffu);
  for (int r0 = 0; r0 < P; r0 += BS) {
    const int rho = r0 + tid;
    unsigned long long v = 0;
    QDiv d; d.c[0] = d.c[1] = d.c[2] = d.c[3] = 0;
    if (rho < P) {
      d = div[rho];
      for (int q = 0; q < 4; ++q) v += (d.c[q] > 0 ? 1ull : 0ull) + (d.c[q] > 1 ? (1ull << 32) : 0ull);
    }
    unsigned long long tot;
    const unsigned long long ex = carry + block_exclusive_scan<unsigned long long>(v, s_scan, &tot);
    carry += tot;
    if (rho < P && v != 0) {
      const int pos = kIdentity ? rho : (int)sval[m - 1 - rho];
      const QNode nd = cur[pos];
      uint32_t child_rank = (uint32_t)(ex & 0xffffffffu), exp_rank = (uint32_t)(ex >> 32);
      uint32_t beg = nd.beg;
      for (int q = 0; q < 4; ++q) {
        if (d.c[q] > 0) {
          const uint32_t idx = T - child_rank - 1;
          nxt[idx] = child_node(nd, q, beg, d.c[q]);
          if (d.c[q] > 1) todo_out[exp_rank++] = idx;
          ++child_rank;
        }
        beg += d.c[q];
      }
      divided[pos] = 1;
    }
  }
  __syncthreads();
  // pass 2: surviving nodes keep their relative order behind the children
  uint32_t kcarry = 0;
  for (int p0 = 0; p0 < n; p0 += BS) {
    const int pos = p0 + tid;
    const uint32_t keep = (pos < n && !divided[pos]) ? 1u : 0u;
    uint32_t tot;
    uint32_t* s32 = reinterpret_cast<uint32_t*>(s_scan);
    const uint32_t ex = kcarry + block_exclusive_scan<uint32_t>(keep, s32, &tot);
    kcarry += tot;
    if (keep) nxt[T + ex] = cur[pos];
  }
  if (tid == 0) { *s_newn = (int)(T + kcarry); *s_nexp = (int)(T2 >> 32); }
  __syncthreads();
}

template <int BS>
__global__ __launch_bounds__(BS, 4) void k_octree_moving(const LevelGeom* __restrict__ geom, int n_levels, OctreeBufs b, int level_begin) {
  __shared__ unsigned long long s_scan[32];
  __shared__ unsigned long long s_skey_pad[kSortLds + 8];  // 4 entries of read slack on both sides
  __shared__ uint32_t s_sval[kSortLds];
  __shared__ uint16_t s_seg_first[kSortLds], s_seg_last[kSortLds];
  __shared__ SortRanges s_ra, s_rb;
  __shared__ int s_sort_cnt[2];
  unsigned long long* s_skey = s_skey_pad + 4;
  __shared__ int s_newn, s_nexp, s_n, s_P, s_nbig;
  constexpr int kMaxBig = 64;
  __shared__ int s_big[kMaxBig];
  __shared__ uint32_t s_wcnt[BS / 64][4];
  __shared__ uint32_t s_rootcnt[kMaxRoots];
#define RGBL_STAMP(k) do { if (b.dbg && threadIdx.x == 0) b.dbg[((size_t)blockIdx.y * n_levels + blockIdx.x + level_begin) * 16 + (k)] = rgbl_clock(); } while (0)
  RGBL_STAMP(0);

  const int tid = threadIdx.x, lane = lane_id(), wave = wave_id();
  const int nw = (int)(blockDim.x >> 6);
  const int l = blockIdx.x + level_begin, f = blockIdx.y;  // the launch covers the levels level_begin .. level_begin + gridDim.x
  const LevelGeom& g = geom[l];
  uint32_t* keys_a = b.keys_a + (size_t)f * b.keys_frame + g.key_off;
  uint32_t* keys_b = b.keys_b + (size_t)f * b.keys_frame + g.key_off;
  const size_t nbase = (size_t)f * b.nodes_frame + g.node_off;
  QNode* cur = b.list_a + nbase;
  QNode* nxt = b.list_b + nbase;
  QDiv* div = b.div + nbase;
  uint32_t* todo = b.todo_a + nbase;
  uint32_t* todo_n = b.todo_b + nbase;
  uint8_t* divided = b.divided + nbase;
  const int N = g.quota;

  // ---- 0. gather the cell slots into one dense list in the reference's order (cell-major) -> keys_a
  uint32_t C = 0;
  {
    const uint32_t* ccnt = b.cell_cnt + (size_t)f * b.cells_frame + g.cell_off;
    const uint32_t* slots = b.slots + (size_t)f * b.slots_frame + g.slot_off;
    for (int c0 = 0; c0 < g.n_cells; c0 += BS) {
      const int c = c0 + tid;
      const uint32_t cnt = c < g.n_cells ? ccnt[c] : 0u;
      uint32_t tot;
      const uint32_t base = C + block_exclusive_scan<uint32_t>(cnt, reinterpret_cast<uint32_t*>(s_scan), &tot);
      for (uint32_t k = 0; k < cnt; ++k) keys_a[base + k] = slots[(size_t)c * g.cell_cap + k];
      C += tot;
    }
  }
  __syncthreads();

  RGBL_STAMP(1);
  // ---- 1. stable partition into the n_ini root nodes (ORBextractor.cc:582-586) -> keys_b
  const uint8_t* rootx = b.rootx + g.rootx_off;
  {
    uint32_t outpos = 0;
    for (int r = 0; r < g.n_ini; ++r) {
      const uint32_t beg = outpos;
      for (uint32_t base = 0; base < C; base += BS * 8) {
        const uint32_t i0 = base + (uint32_t)tid * 8;
        uint32_t mask = 0;
        for (int k = 0; k < 8; ++k)
          if (i0 + k < C && rootx[key_x(keys_a[i0 + k])] == r) mask |= 1u << k;
        uint32_t tot;
        uint32_t pos = outpos + block_exclusive_scan<uint32_t>((uint32_t)__popc(mask), reinterpret_cast<uint32_t*>(s_scan), &tot);
        for (int k = 0; k < 8; ++k)
          if ((mask >> k) & 1) keys_b[pos++] = keys_a[i0 + k];
        outpos += tot;
      }
      if (tid == 0) s_rootcnt[r] = outpos - beg;
    }
  }
  __syncthreads();
  if (tid == 0) {
    int n0 = 0;
    uint32_t beg = 0;
    for (int r = 0; r < g.n_ini; ++r) {
      const uint32_t cnt = s_rootcnt[r];
      if (cnt > 0) {  // empty roots are erased (ORBextractor.cc:597-598)
        QNode nd;
        nd.x0 = (uint16_t)g.root_x[r]; nd.x1 = (uint16_t)g.root_x[r + 1];
        nd.y0 = 0; nd.y1 = (uint16_t)(g.max_by - kMinBorder);
        nd.beg = beg; nd.cnt = cnt | 0x80000000u;
        cur[n0++] = nd;
      }
      beg += cnt;
    }
    s_n = n0;
  }
  __syncthreads();
  int n = s_n;
  int m = 0;

  RGBL_STAMP(2);
  // ---- 2. breadth-first splitting (ORBextractor.cc:608-686)
  bool finished = (n == 0);
  bool careful = false;
  while (!finished) {
    // very large nodes: one after the other, whole workgroup each
    if (tid == 0) s_nbig = 0;
    __syncthreads();
    for (int pos = tid; pos < n; pos += BS)
      if ((cur[pos].cnt & 0x7fffffffu) >= (uint32_t)kCoopMin) {
        const int k = atomicAdd(&s_nbig, 1);
        if (k < kMaxBig) s_big[k] = pos;
      }
    __syncthreads();
    const int nbig = s_nbig <= kMaxBig ? s_nbig : 0;  // more than kMaxBig big nodes: leave all of them to the waves
    for (int k = 0; k < nbig; ++k) {
      const int pos = s_big[k];
      const QNode nd = cur[pos];
      QDiv d;
      block_split<BS>(nd, keys_a, keys_b, d.c, s_scan, s_wcnt);
      if (tid == 0) { div[pos] = d; divided[pos] = 0; }
    }
    // everything else: one wave per node
    for_nodes_pipelined(cur, wave, n, nw, keys_a, keys_b, [](int t) { return t; }, [&](int pos, const QNode& nd, uint32_t key0) {
      const uint32_t cnt = nd.cnt & 0x7fffffffu;
      if (cnt >= (uint32_t)kCoopMin && nbig > 0) return;  // already split above
      QDiv d; d.c[0] = d.c[1] = d.c[2] = d.c[3] = 0;
      if (cnt > 1) node_split(nd, key0, keys_a, keys_b, d.c);
      if (lane == 0) { div[pos] = d; divided[pos] = 0; }
    });
    __syncthreads();
    rebuild_list<true, BS>(cur, nxt, n, div, n, nullptr, 0, divided, todo_n, s_scan, &s_newn, &s_nexp);
    const int newn = s_newn, nexp = s_nexp;
    { QNode* t = cur; cur = nxt; nxt = t; }
    { uint32_t* t = todo; todo = todo_n; todo_n = t; }
    const int prev = n;
    n = newn;
    m = nexp;
    if (n > (int)g.node_cap - 8) { if (tid == 0) atomicOr(b.err, 1); finished = true; }
    else if (n >= N || n == prev) finished = true;
    else if (n + 3 * nexp > N) { careful = true; break; }
  }

  RGBL_STAMP(3);
  // ---- 3. near the quota: split the most populated nodes first (ORBextractor.cc:689-753)
  while (careful && !finished) {
    const int prev = n;
    // compareNodes orders by (size, UL.x); equal keys end up in libstdc++'s introsort order
    uint32_t* sval = (m <= kSortLds) ? s_sval : b.sval + nbase;
    for (int p = tid; p < n; p += BS) divided[p] = 0;
    if (m <= kSortLds) {
      uint64_t* w = reinterpret_cast<uint64_t*>(s_skey);
      for (int j = tid; j < m; j += BS) {
        const uint32_t pos = todo[j];
        const QNode nd = cur[pos];
        w[j] = ((uint64_t)(nd.cnt & 0x7fffffffu) << 28) | ((uint64_t)nd.x0 << 16) | pos;  // x0 < 4096, pos < 65536
      }
      __syncthreads();
      RGBL_STAMP(8);
      block_sort_restated<BS>(w, m, s_seg_first, s_seg_last, &s_ra, &s_rb, s_sort_cnt);
      RGBL_STAMP(9);
      for (int j = tid; j < m; j += BS) sval[j] = (uint32_t)(w[j] & 0xffffu);
      __syncthreads();
    } else {
      uint64_t* skey = b.skey + nbase;
      for (int j = tid; j < m; j += BS) {
        const uint32_t pos = todo[j];
        const QNode nd = cur[pos];
        skey[j] = ((uint64_t)(nd.cnt & 0x7fffffffu) << 32) | nd.x0;
        sval[j] = pos;
      }
      __syncthreads();
      if (tid == 0) std_sort_restated(skey, sval, m);
      __syncthreads();
    }
    // child counts of every expandable node, rank rho = position counted from the back of the sorted array
    for_nodes_pipelined(cur, wave, m, nw, keys_a, keys_b, [&](int j) { return (int)sval[j]; }, [&](int j, const QNode& nd, uint32_t key0) {
      QDiv d;
      node_count_k0(nd, key0, keys_a, keys_b, d.c);
      if (lane == 0) div[m - 1 - j] = d;
    });
    if (tid == 0) s_P = m;
    __syncthreads();
    // first rank after which the list has reached the quota (the reference breaks out of its loop there)
    {
      long long carry = 0;
      for (int r0 = 0; r0 < m; r0 += BS) {
        const int rho = r0 + tid;
        long long v = 0;
        if (rho < m) {
          const QDiv d = div[rho];
          v = (long long)((d.c[0] > 0) + (d.c[1] > 0) + (d.c[2] > 0) + (d.c[3] > 0)) - 1;
        }
        unsigned long long tot;
        const unsigned long long ex = block_exclusive_scan<unsigned long long>((unsigned long long)v, s_scan, &tot);
        const long long size_after = (long long)n + carry + (long long)ex + v;
        if (rho < m && size_after >= N) atomicMin(&s_P, rho + 1);
        carry += (long long)tot;
        __syncthreads();
      }
    }
    __syncthreads();
    const int P = s_P;
    for_nodes_pipelined(cur, wave, P, nw, keys_a, keys_b, [&](int rho) { return (int)sval[m - 1 - rho]; },
                        [&](int rho, const QNode& nd, uint32_t key0) {
      const QDiv d = div[rho];
      node_place_k0(nd, key0, keys_a, keys_b, d.c);
    });
    __syncthreads();
    rebuild_list<false, BS>(cur, nxt, n, div, P, sval, m, divided, todo_n, s_scan, &s_newn, &s_nexp);
    { QNode* t = cur; cur = nxt; nxt = t; }
    { uint32_t* t = todo; todo = todo_n; todo_n = t; }
    n = s_newn;
    m = s_nexp;
    if (n > (int)g.node_cap - 8) { if (tid == 0) atomicOr(b.err, 1); finished = true; }
    else if (n >= N || n == prev) finished = true;
  }

  RGBL_STAMP(4);
  // ---- 4. keep the strongest key of every node, first one on ties (ORBextractor.cc:757-776)
  uint32_t* out = b.kp_key + (size_t)f * b.kp_frame + g.koff;
  if (n > g.kcap) { if (tid == 0) atomicOr(b.err, 2); n = g.kcap; }
  for (int pos = tid; pos < n; pos += BS) {
    const QNode nd = cur[pos];
    const uint32_t cnt = nd.cnt & 0x7fffffffu;
    const uint32_t* src = ((nd.cnt >> 31) ? keys_b : keys_a) + nd.beg;
    uint32_t best = src[0];
    for (uint32_t k = 1; k < cnt; ++k) {
      const uint32_t key = src[k];
      if (key_s(key) > key_s(best)) best = key;
    }
    out[pos] = best;
  }
  if (tid == 0) b.kp_count[(size_t)f * n_levels + l] = n;
  RGBL_STAMP(5);
  if (b.dbg && tid == 0) { b.dbg[((size_t)f * n_levels + l) * 16 + 6] = C; b.dbg[((size_t)f * n_levels + l) * 16 + 7] = (unsigned long long)n; }
}

// ------------------------------------------------------------------------------------------------
// cv::fastAtan2 (degrees), scalar generic path, fp32 without contraction
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
  const float scale = (float)(180.0 / 3.14159265358979323846);
  const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale,
              p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
  const float eps = (float)2.2204460492503131e-16;  // (float)DBL_EPSILON
  const float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = __fdiv_rn(ay, ax + eps);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = __fdiv_rn(ax, ay + eps);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// Orientation + descriptor.  A wave owns kKpPerWave consecutive keypoint slots.  The pixel work of a keypoint (the moments
// of the 31x31 circular patch, the 256 steered comparisons on the blurred 37x37 patch) is spread over the 64 lanes, one
// keypoint after the other; everything that is ONE value per keypoint - slot -> (level, index), fastAtan2, the
// double-precision sin / cos, the keypoint record - is evaluated once with lane k working for keypoint k, instead of
// being replicated over all lanes of a one-keypoint wave (it was two thirds of the issued instructions).
// Four __ballot results per keypoint are descriptor bytes 0-7, 8-15, 16-23, 24-31.
// grid = (ceil(kp_frame / (4 * kKpPerWave)), B), block = 256 (128: 144.4 k, 512: 143.5 k against 146.5 k frames/s).
// 8 since the end of round 6 (4 before: 10 registers of patch words per keypoint in flight; 2 / 4 / 6 / 8 / 12 / 16 on the KITTI step, twice each
// on one box: 142.1 / 144.9 / 146.3 / 146.3 / 144.2 / 141.6 k frames/s, the kernel 1.38 / 1.26 / 1.25 / 1.23 / 1.21 / 1.35 ms; single frames
// unchanged: 0.308 -> 0.309 ms through the drop-in classes).
constexpr int kKpPerWave = 8;

__device__ __forceinline__ int bcast_i(int v, int k) {  // value of lane k, wave-uniform (k is a constant after unrolling)
#ifdef RGBL_EMU
  return __shfl(v, k);
#else
  return __builtin_amdgcn_readlane(v, k);
#endif
}
__device__ __forceinline__ float bcast_f(float v, int k) { return __int_as_float(bcast_i(__float_as_int(v), k)); }

// (Round 5: more resident waves do not help here - amdgpu_waves_per_eu(6) gives 71 VGPRs / 7 waves per SIMD instead of 82 / 5 and
// the kernel takes 0.65 instead of 0.63 ms per 512 frames, (8) spills: 0.81 ms.  What paces it is the number of cache lines its
// patch gathers touch - ~90 per keypoint, one L1 tag look-up each - not the latency those waves could hide.)
template <int BS>
__global__ __launch_bounds__(BS) void k_orient_brief(const LevelGeom* __restrict__ geom, int n_levels, UMax umax,
                                                      const int8_t* __restrict__ pattern,
                                                      const uint8_t* __restrict__ img0, int pitch0, size_t frame0,
                                                      const uint8_t* __restrict__ pyr, size_t pyr_frame,
                                                      const uint8_t* __restrict__ blur, size_t blur_frame,
                                                      const uint32_t* __restrict__ kp_key,
                                                      const int* __restrict__ kp_count, size_t kp_frame,
                                                      rgbl_keypoint* __restrict__ out_kp,
                                                      uint8_t* __restrict__ out_desc, int cap,
                                                      int32_t* __restrict__ out_n, int32_t* __restrict__ out_mono,
                                                      int* __restrict__ err, int slot_begin, int slot_end, int write_total) {
  // The launch covers the keypoint slots slot_begin .. slot_end - 1 (slots are laid out level after level): level 0's
  // keypoints can be described while the quad-trees of the upper levels still run; write_total: this launch sees the final
  // counts of ALL levels and writes the frame's keypoint count.
  __shared__ unsigned long long s_patch_q[BS / 64][37 * 5];  // blurred 37x37 neighbourhood, 40-byte rows
  const int lane = lane_id(), wave = wave_id();
  const int bx = xcd_item(), f = xcd_frame();  // grid = xcd_grid(keypoint groups, B): an XCD's L2 keeps its frames' levels
  const int* cnts = kp_count + (size_t)f * n_levels;

  // ---- lane k < kKpPerWave: which (level, index) is slot s0 + k?  slots are laid out level after level, kcap entries each
  const int s0 = slot_begin + (bx * (BS / 64) + wave) * kKpPerWave;
  const int slot = s0 + (lane < kKpPerWave ? lane : 0);
  int l = 0;
  for (int j = 1; j < n_levels; ++j) l += (slot >= geom[j].koff) ? 1 : 0;  // koff ascends with the level
  const int idx = slot - geom[l].koff;
  bool valid = lane < kKpPerWave && slot < slot_end && idx < cnts[l];
  int dense = idx;  // position in the level-major output order
  for (int j = 0; j < n_levels; ++j) dense += (j < l) ? cnts[j] : 0;
  if (write_total && bx == 0 && wave == 0 && lane == 0) {
    int total = 0;
    for (int j = 0; j < n_levels; ++j) total += cnts[j];
    out_n[f] = total < cap ? total : cap;
    if (out_mono) out_mono[f] = total < cap ? total : cap;  // no lapping area: monoIndex == count
    if (total > cap) atomicOr(err, 4);
  }
  if (valid && dense >= cap) valid = false;
  const uint32_t key = valid ? kp_key[(size_t)f * kp_frame + slot] : 0u;
  const int kx = key_x(key) + kMinBorder, ky = key_y(key) + kMinBorder;
  const unsigned long long vmask = __ballot(valid);  // (no early exit of a wave without keypoints: workgroup barriers below)

  // the lane's four pattern pairs (x0, y0, x1, y1 as signed bytes), the same for every keypoint
  f32x2 px[4], py[4];  // (x0, x1) and (y0, y1) of the lane's pairs
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t pw = reinterpret_cast<const uint32_t*>(pattern)[k * 64 + lane];
    px[k] = f32x2{(float)(int8_t)(pw & 0xff), (float)(int8_t)((pw >> 16) & 0xff)};
    py[k] = f32x2{(float)(int8_t)((pw >> 8) & 0xff), (float)(int8_t)(pw >> 24)};
  }

  // ---- all patch words of all keypoints are requested first (one exposed memory latency per wave instead of one per
  //      keypoint and pass): 4 + 6 registers per keypoint.  Keypoints sit >= 19 px inside the level, so x-15..x+16 and
  //      x-18..x+21 stay inside the row pitch.
  // Wide loads: a patch row of 32 bytes is two lanes x 16 bytes (one global_load_dwordx4 per keypoint instead of four dword
  // loads), a blurred row of 40 bytes five lanes x 8 bytes (three dwordx2 loads instead of six dword loads) - the kernel is
  // bound by the number of vector-memory instructions its patches take, not by their bytes.
  uint32_t rawreg[kKpPerWave][4];
  unsigned long long blreg[kKpPerWave][3];
#pragma unroll
  for (int k = 0; k < kKpPerWave; ++k) {
    if (!((vmask >> k) & 1)) continue;  // wave-uniform
    const int lk = bcast_i(l, k), x = bcast_i(kx, k), y = bcast_i(ky, k);
    const LevelGeom& g = geom[lk];
    const uint8_t* img = (lk == 0) ? img0 + (size_t)f * frame0 : pyr + (size_t)f * pyr_frame + g.img_off;
    const int pitch = (lk == 0) ? pitch0 : g.pitch;
    const uint8_t* src = img + (size_t)(y - 15) * pitch + (x - 15);  // wave-uniform
    uint4 q;
    q.x = q.y = q.z = q.w = 0u;
    if (lane < 62) __builtin_memcpy(&q, src + (__umul24((uint32_t)(lane >> 1), (uint32_t)pitch) + (uint32_t)(16 * (lane & 1))), 16);
    rawreg[k][0] = q.x; rawreg[k][1] = q.y; rawreg[k][2] = q.z; rawreg[k][3] = q.w;
  }
#pragma unroll
  for (int k = 0; k < kKpPerWave; ++k) {
    if (!((vmask >> k) & 1)) continue;
    const int lk = bcast_i(l, k), x = bcast_i(kx, k), y = bcast_i(ky, k);
    const LevelGeom& g = geom[lk];
    const uint8_t* bl = blur + (size_t)f * blur_frame + g.img_off + (size_t)(y - 18) * g.pitch + (x - 18);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int i = lane + 64 * j;
      const int r = i / 5, c = i - r * 5;
      unsigned long long v = 0ull;
      if (i < 37 * 5) __builtin_memcpy(&v, bl + (__umul24((uint32_t)r, (uint32_t)g.pitch) + (uint32_t)(8 * c)), 8);
      blreg[k][j] = v;
    }
  }

  // what depends on the lane only: its row v and half, the bytes of its 16 that lie inside the circle (|u| <= umax[|v|];
  // u = 16 does not exist), the weights u + 15 resp. u of the bytes
  uint32_t ic_mask[4] = {0u, 0u, 0u, 0u}, ic_coef[4];
  int ic_off;
  {
    const int v = (lane >> 1) - 15, half = lane & 1;
    const int d = umax.v[lane < 62 ? (v < 0 ? -v : v) : 0];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const bool in = lane < 62 && (half ? (q + 1 <= d) : (15 - q <= d));
      ic_mask[q >> 2] |= in ? (0xffu << (8 * (q & 3))) : 0u;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) ic_coef[j] = 0x03020100u + 0x04040404u * (uint32_t)j + (half ? 0x01010101u : 0u);
    ic_off = half ? 0 : 15;
  }
  // ---- pass 1, keypoint after keypoint: IC_Angle moments (ORBextractor.cc:76-101) straight from the registers the patch
  //      was loaded into: two lanes per row, both sums are integer dot products (v_dot4_u32_u8) of the masked words
  int my_m10 = 0, my_m01 = 0;
#pragma unroll
  for (int k = 0; k < kKpPerWave; ++k) {
    if (!((vmask >> k) & 1)) continue;  // wave-uniform
    uint32_t sum0 = 0, sum1 = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t wm = rawreg[k][j] & ic_mask[j];
      sum0 = udot4(wm, 0x01010101u, sum0);
      sum1 = udot4(wm, ic_coef[j], sum1);
    }
    int m10 = (int)sum1 - ic_off * (int)sum0;   // sum of u p with u = q - 15 (half 0) or q + 1 (half 1)
    int m01 = ((lane >> 1) - 15) * (int)sum0;
    m10 = wave_sum_uniform(m10);
    m01 = wave_sum_uniform(m01);
    if (lane == k) { my_m10 = m10; my_m01 = m01; }
  }

  // ---- once per keypoint: orientation and the steering coefficients (fastAtan2, glibc's sinf / cosf in double precision:
  //      ~150 vector instructions that only need one lane per keypoint).  The first wave does it for the keypoints of ALL the
  //      workgroup's waves (lane j for the workgroup's j-th keypoint) instead of every wave for its own four.
  __shared__ int s_mom[BS / 64 * kKpPerWave][2];
  __shared__ float s_angle[BS / 64 * kKpPerWave], s_cos[BS / 64 * kKpPerWave], s_sin[BS / 64 * kKpPerWave];
  if (lane < kKpPerWave) { s_mom[wave * kKpPerWave + lane][0] = my_m10; s_mom[wave * kKpPerWave + lane][1] = my_m01; }
  __syncthreads();
  if (wave == 0 && lane < BS / 64 * kKpPerWave) {
    const float an = fast_atan2_deg((float)s_mom[lane][1], (float)s_mom[lane][0]);
    const float factor_pi = (float)(3.14159265358979323846 / 180.f);
    const float rad = an * factor_pi;
    s_angle[lane] = an; s_cos[lane] = glibc_cosf(rad); s_sin[lane] = glibc_sinf(rad);
  }
  __syncthreads();
  const int mine = wave * kKpPerWave + (lane < kKpPerWave ? lane : 0);
  const float angle = s_angle[mine], ca = s_cos[mine], sb = s_sin[mine];

  // ---- pass 2, keypoint after keypoint: steered BRIEF-256 on the blurred level through a 37x37 LDS patch
  const uint8_t* patch = reinterpret_cast<const uint8_t*>(s_patch_q[wave]);
#pragma unroll
  for (int k = 0; k < kKpPerWave; ++k) {
    if (!((vmask >> k) & 1)) continue;  // wave-uniform
    const int dk = bcast_i(dense, k);
    const float a = bcast_f(ca, k), b = bcast_f(sb, k);
    wave_sync();
#pragma unroll
    for (int j = 0; j < 3; ++j)
      if (lane + 64 * j < 37 * 5) s_patch_q[wave][lane + 64 * j] = blreg[k][j];
    wave_sync();
    // Two points of a pair side by side in packed fp32 (v_pk_mul_f32 / v_pk_add_f32: two IEEE operations per instruction,
    // no contraction): x a - y b and x b + y a as in computeOrbDescriptor (ORBextractor.cc:118-119).  cvRound = adding
    // 1.5 * 2^23 (round-half-even lands in the mantissa; |coordinate| < 2^22); the two biased integers go straight into
    // one 24-bit multiply-add, the biases leave with one constant.
    const f32x2 av = {a, a}, bv = {b, b}, magic = {12582912.0f, 12582912.0f};
    constexpr int kBias = 0x400000 * 40 + 0x4B400000;  // (low 24 bits of the biased row) * 40 + the biased column
    unsigned long long bits[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x2 rx = px[q] * av - py[q] * bv;
      const f32x2 ry = px[q] * bv + py[q] * av;
      const f32x2 cx = rx + magic, cy = ry + magic;
      const int i0 = __mul24(__float_as_int(cy[0]), 40) + __float_as_int(cx[0]) - kBias;
      const int i1 = __mul24(__float_as_int(cy[1]), 40) + __float_as_int(cx[1]) - kBias;
      const int t0 = patch[18 * 40 + 18 + i0];
      const int t1 = patch[18 * 40 + 18 + i1];
      bits[q] = __ballot(t0 < t1);
    }
    if (lane < 4) {
      unsigned long long* d = reinterpret_cast<unsigned long long*>(out_desc + ((size_t)f * cap + dk) * 32);
      d[lane] = bits[lane];
    }
  }
  if (valid) {
    const LevelGeom& g = geom[l];
    rgbl_keypoint kp;
    kp.x = (float)kx; kp.y = (float)ky;
    if (l != 0) { kp.x = kp.x * g.scale; kp.y = kp.y * g.scale; }
    kp.size = (float)g.patch_size;
    kp.angle = angle;
    kp.response = (float)key_s(key);
    kp.octave = l;
    kp.class_id = -1;
    out_kp[(size_t)f * cap + dense] = kp;
  }
}

// vLappingArea packing (ORBextractor.cc:1153-1162): keypoints with lap0 <= x <= lap1 fill the arrays from
// the back (in reverse order), all others from the front.  One workgroup per frame.
__global__ __launch_bounds__(256) void k_lapping_permute(const rgbl_keypoint* __restrict__ in_kp,
                                                         const uint8_t* __restrict__ in_desc, int in_cap,
                                                         rgbl_keypoint* __restrict__ out_kp,
                                                         uint8_t* __restrict__ out_desc, int cap,
                                                         const int32_t* __restrict__ n_per_frame, float lap0,
                                                         float lap1, int32_t* __restrict__ out_mono) {
  __shared__ uint32_t s_scan[8];
  const int tid = threadIdx.x, f = blockIdx.x;
  const int n = n_per_frame[f];
  uint32_t carry = 0;
  for (int i0 = 0; i0 < n; i0 += 256) {
    const int i = i0 + tid;
    rgbl_keypoint kp;
    kp.x = 0;
    uint32_t in_lap = 0;
    if (i < n) {
      kp = in_kp[(size_t)f * in_cap + i];
      in_lap = (kp.x >= lap0 && kp.x <= lap1) ? 1u : 0u;
    }
    uint32_t tot;
    const uint32_t before = carry + block_exclusive_scan<uint32_t>(in_lap, s_scan, &tot);
    carry += tot;
    if (i < n) {
      const int dst = in_lap ? (n - 1 - (int)before) : (i - (int)before);
      out_kp[(size_t)f * cap + dst] = kp;
      const uint4* s = reinterpret_cast<const uint4*>(in_desc + ((size_t)f * in_cap + i) * 32);
      uint4* d = reinterpret_cast<uint4*>(out_desc + ((size_t)f * cap + dst) * 32);
      d[0] = s[0];
      d[1] = s[1];
    }
  }
  if (tid == 0) out_mono[f] = n - (int)carry;
}

// test hook: the workgroup sort on plain arrays (n <= kSortLds, key < 2^48, val < 2^16)
__global__ __launch_bounds__(kOctWide) void k_test_block_sort(uint64_t* key, uint32_t* val, int n) {
  __shared__ unsigned long long s_skey_pad[kSortLds + 8];
  __shared__ uint16_t s_seg_first[kSortLds], s_seg_last[kSortLds];
  __shared__ SortRanges s_ra, s_rb;
  __shared__ int s_sort_cnt[2];
  uint64_t* w = reinterpret_cast<uint64_t*>(s_skey_pad + 4);
  for (int i = threadIdx.x; i < n; i += kOctWide) w[i] = (key[i] << 16) | (val[i] & 0xffffu);
  __syncthreads();
  block_sort_restated<kOctWide>(w, n, s_seg_first, s_seg_last, &s_ra, &s_rb, s_sort_cnt);
  for (int i = threadIdx.x; i < n; i += kOctWide) { key[i] = w[i] >> 16; val[i] = (uint32_t)(w[i] & 0xffffu); }
}

// ------------------------------------------------------------------------------------------------
// Frame::ComputeStereoMatches (/root/reference/src/Frame.cc:901-1071), SURVEY.md 8(f) row f1.
//   k_stereo_match   one work-item per left keypoint: candidates = right keypoints whose row band
//                    [floor(y - 2 s), ceil(y + 2 s)] contains the left row, octave within +-1, u inside the disparity
//                    window; visited in ascending index order with strict '<' (== the reference's row table);
//                    then the 11x11 SAD over 11 shifts on the left keypoint's pyramid level, parabola fit.
//   k_stereo_filter  one workgroup per frame: median of the accepted SADs by a two-level histogram, outlier cut.
struct ScaleTables { float scale[kMaxLevels], inv_scale[kMaxLevels]; };
struct PyrView { const uint8_t* img0; int pitch0; size_t frame0; const uint8_t* pyr; size_t pyr_frame; };
constexpr int kStereoTile = 2048;      // right keypoints staged in LDS per pass of k_stereo_match (18 KB)
constexpr int kStereoLanes = 4;         // work-items per left keypoint in the candidate search
constexpr int kStereoBatch = 16;        // candidates whose descriptors are fetched together
constexpr int kStereoRowBias = 4096;   // row bands are kept as two biased 16-bit halves of one word

__device__ __forceinline__ int pyr_px(const uint8_t* base, int pitch, int w, int h, int x, int y) {
  return base[(size_t)reflect101(y, h) * pitch + reflect101(x, w)];  // the reference reads a reflect-101 bordered buffer
}

__global__ __launch_bounds__(256) void k_stereo_match(const LevelGeom* __restrict__ geom, ScaleTables st, PyrView PL, PyrView PR,
                                                      const rgbl_keypoint* __restrict__ kpl, const uint8_t* __restrict__ dl,
                                                      const int32_t* __restrict__ nl, const rgbl_keypoint* __restrict__ kpr,
                                                      const uint8_t* __restrict__ dr, const int32_t* __restrict__ nr, int cap,
                                                      float mb, float mbf, int n_rows, float* __restrict__ uright,
                                                      float* __restrict__ depth, int32_t* __restrict__ sad_out) {
  // The right keypoints' row band [floor(y - 2 s), ceil(y + 2 s)], octave and u - what the reference's vRowIndices table and
  // the candidate tests read - are staged in LDS in tiles of kStereoTile, once per workgroup (round 6: every work-item had
  // fetched all Nr 28-byte records from global memory itself, one dependent load per candidate: 560 us for one KITTI pair).
  __shared__ __attribute__((aligned(16))) float s_u[kStereoTile];
  __shared__ __attribute__((aligned(16))) int s_band[kStereoTile];   // minr (low half, biased) | maxr (high half, biased)
  __shared__ __attribute__((aligned(16))) uint8_t s_oct[kStereoTile];
  // kStereoLanes work-items per left keypoint: each scans every kStereoLanes-th group of 32 right keypoints (the scan is
  // N_left x N_right cheap tests - ALU-bound at ~100 us per KITTI pair with one work-item per left keypoint, and a pair's 2000
  // left keypoints are 32 waves on 256 CUs), the best (distance, index) keys meet in a two-step lane exchange
  const int f = blockIdx.y;
  const int iL = (int)(blockIdx.x * blockDim.x + threadIdx.x) / kStereoLanes, part = threadIdx.x & (kStereoLanes - 1);
  const int N = nl[f], Nr = nr[f];
  if ((int)(blockIdx.x * blockDim.x) / kStereoLanes >= N) return;   // the whole workgroup is beyond the frame's keypoints (uniform: no barrier is skipped by a part of it)
  const bool live = iL < N;
  const size_t o = (size_t)f * cap + (live ? iL : 0);
  const rgbl_keypoint kL = kpl[o];
  float out_u = -1.0f, out_d = -1.0f;
  int out_sad = -1;
  const float minD = 0, maxD = __fdiv_rn(mbf, mb);
  const float uL = kL.x, vL = kL.y;
  const int levelL = kL.octave;
  const int row = (int)vL;
  const float minU = uL - maxD, maxU = uL - minD;
  const bool searching = live && !(maxU < 0) && row >= 0 && row < n_rows;
  const unsigned long long* D = reinterpret_cast<const unsigned long long*>(dl + o * 32);
  const unsigned long long q0 = D[0], q1 = D[1], q2 = D[2], q3 = D[3];
  int bestDist = 100 /* TH_HIGH */, bestIdxR = 0;
  const rgbl_keypoint* KR = kpr + (size_t)f * cap;
  const unsigned long long* DR = reinterpret_cast<const unsigned long long*>(dr + (size_t)f * cap * 32);
  const int rowb = row + kStereoRowBias;
  // candidates of the current tile that passed the cheap tests, per work-item, in index order: their descriptors are fetched
  // afterwards, kStereoBatch independent 32-byte gathers at a time (inside the scan every candidate's gather had to return
  // before the next band could be looked at)
  __shared__ uint16_t s_cand[kStereoBatch][64 * 4];   // [slot][work-item]: conflict-free columns
  const int me = threadIdx.x;
  auto flush = [&](int n, int t0) {
    for (int c0 = 0; c0 < n; c0 += 4) {   // four gathers in flight, then their distances in index order
      int iR[4];
      unsigned long long w[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        iR[u] = t0 + s_cand[imin(c0 + u, n - 1)][me];
        const unsigned long long* t = DR + 4 * (size_t)iR[u];
        w[u][0] = t[0]; w[u][1] = t[1]; w[u][2] = t[2]; w[u][3] = t[3];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int dist = __popcll(q0 ^ w[u][0]) + __popcll(q1 ^ w[u][1]) + __popcll(q2 ^ w[u][2]) + __popcll(q3 ^ w[u][3]);
        if (c0 + u < n && dist < bestDist) { bestDist = dist; bestIdxR = iR[u]; }
      }
    }
  };
  for (int t0 = 0; t0 < Nr; t0 += kStereoTile) {
    const int nt = imin(kStereoTile, Nr - t0);
    __syncthreads();
    for (int j = threadIdx.x; j < nt; j += blockDim.x) {
      const rgbl_keypoint k = KR[t0 + j];
      const float r = 2.0f * st.scale[k.octave];
      const int maxr = imin(imax((int)ceilf(k.y + r), -kStereoRowBias), 0x7fff - kStereoRowBias);
      const int minr = imin(imax((int)floorf(k.y - r), -kStereoRowBias), 0x7fff - kStereoRowBias);
      s_u[j] = k.x;
      s_band[j] = (minr + kStereoRowBias) | ((maxr + kStereoRowBias) << 16);
      s_oct[j] = (uint8_t)k.octave;
    }
    if (threadIdx.x < 4 && (nt & 3) && (nt & ~3) + (int)threadIdx.x >= nt) s_band[(nt & ~3) + threadIdx.x] = 0xffff;   // no row is below 0xffff
    __syncthreads();
    if (searching) {
      // Four right keypoints per trip, everything about them read up front (three wide LDS reads, one latency) and tested
      // without a branch.  The left keypoints of a wave lie on arbitrary rows, so SOME lane has a candidate in nearly every
      // group of four: an early exit per band never leaves the wave, and the three dependent LDS reads behind it - band,
      // octave, u - ran one after the other for every right keypoint (~1300 cycles per group of four, 270 us per frame).
      int nc = 0;
      for (int j32 = 32 * part; j32 < nt; j32 += 32 * kStereoLanes) {
        // 32 right keypoints -> one bit each (pure ALU behind three wide LDS reads per four), then only the set bits are
        // walked: the append with its rare flush stays out of the per-keypoint path
        uint32_t mask = 0;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          const int j4 = j32 + 4 * g;   // (entries behind nt up to the next multiple of 4 are padding, behind that stale: masked below)
          const int4 b4 = *reinterpret_cast<const int4*>(&s_band[j4]);
          const float4 u4 = *reinterpret_cast<const float4*>(&s_u[j4]);
          const uint32_t o4 = *reinterpret_cast<const uint32_t*>(&s_oct[j4]);
          const int bb[4] = {b4.x, b4.y, b4.z, b4.w};
          const float uu[4] = {u4.x, u4.y, u4.z, u4.w};
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int octR = (int)((o4 >> (8 * u)) & 0xffu);
            const bool pass = !(rowb < (bb[u] & 0xffff) || rowb > (int)((uint32_t)bb[u] >> 16)) && !(octR < levelL - 1 || octR > levelL + 1) &&
                              (uu[u] >= minU && uu[u] <= maxU);
            mask |= pass ? (1u << (4 * g + u)) : 0u;
          }
        }
        if (nt - j32 < 32) mask &= (1u << (nt - j32)) - 1u;
        while (mask) {
          const int u = __ffs((int)mask) - 1;
          mask &= mask - 1u;
          s_cand[nc][me] = (uint16_t)(j32 + u);
          if (++nc == kStereoBatch) { flush(nc, t0); nc = 0; }
        }
      }
      flush(nc, t0);
    }
  }
  {
    // strict '<' in index order == the smallest (distance, index) key: the lanes of a keypoint merge theirs
    uint32_t key = ((uint32_t)bestDist << 16) | (uint32_t)bestIdxR;
#pragma unroll
    for (int m = 1; m < kStereoLanes; m <<= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)key, m); key = o < key ? o : key; }
    bestDist = (int)(key >> 16); bestIdxR = (int)(key & 0xffffu);
  }
  if (!live || part != 0) return;
  if (searching) {
    if (bestDist < 75 /* (TH_HIGH + TH_LOW) / 2 */) {
      const float uR0 = KR[bestIdxR].x;
      const float scaleFactor = st.inv_scale[levelL];
      const float scaleduL = roundf(kL.x * scaleFactor), scaledvL = roundf(kL.y * scaleFactor);
      const float scaleduR0 = roundf(uR0 * scaleFactor);
      const LevelGeom& g = geom[levelL];
      const int w = 5, Lw = 5;
      const float iniu = scaleduR0 + Lw - w, endu = scaleduR0 + Lw + w + 1;
      if (!(iniu < 0 || endu >= (float)g.w)) {
        const uint8_t* IL = levelL == 0 ? PL.img0 + (size_t)f * PL.frame0 : PL.pyr + (size_t)f * PL.pyr_frame + g.img_off;
        const uint8_t* IR = levelL == 0 ? PR.img0 + (size_t)f * PR.frame0 : PR.pyr + (size_t)f * PR.pyr_frame + g.img_off;
        const int pl = levelL == 0 ? PL.pitch0 : g.pitch, pr = levelL == 0 ? PR.pitch0 : g.pitch;
        const int yl0 = (int)(scaledvL - w), xl0 = (int)(scaleduL - w), xr0 = (int)(scaleduR0 - Lw - w);
        int sad[11];
#pragma unroll
        for (int k = 0; k < 11; ++k) sad[k] = 0;
        // windows that lie inside the level (nearly all): a row of the left window is three unaligned words, of the right one six;
        // the 11 shifted comparisons are byte-aligned word triples (v_alignbyte_b32) and three v_sad_u8 each, the twelfth byte
        // masked on both sides (round 6; 32 byte loads and 121 subtract / absolute / add triples per row before)
        const bool inside = xl0 >= 0 && xl0 + 12 <= g.w && xr0 >= 0 && xr0 + 24 <= g.w && yl0 >= 0 && yl0 + 11 <= g.h;
        if (inside) {
          for (int yy = 0; yy < 11; ++yy) {
            const uint8_t* rl = IL + (size_t)(yl0 + yy) * pl + xl0;
            const uint8_t* rr = IR + (size_t)(yl0 + yy) * pr + xr0;
            const uint32_t a0 = load_u32_unaligned(rl), a1 = load_u32_unaligned(rl + 4), a2 = load_u32_unaligned(rl + 8) & 0x00ffffffu;
            uint32_t b[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) b[j] = load_u32_unaligned(rr + 4 * j);
#pragma unroll
            for (int k = 0; k < 11; ++k) {
              const int w0 = k >> 2, sh = k & 3;
              const uint32_t c0 = sh ? align_bytes(b[w0 + 1], b[w0], sh) : b[w0];
              const uint32_t c1 = sh ? align_bytes(b[w0 + 2], b[w0 + 1], sh) : b[w0 + 1];
              const uint32_t c2 = (sh ? align_bytes(b[w0 + 3], b[w0 + 2], sh) : b[w0 + 2]) & 0x00ffffffu;
              sad[k] = (int)sad_u8(a2, c2, sad_u8(a1, c1, sad_u8(a0, c0, (uint32_t)sad[k])));
            }
          }
        } else
        for (int yy = 0; yy < 11; ++yy) {
          int a[11], b[21];
#pragma unroll
          for (int xx = 0; xx < 11; ++xx) a[xx] = pyr_px(IL, pl, g.w, g.h, xl0 + xx, yl0 + yy);
#pragma unroll
          for (int xx = 0; xx < 21; ++xx) b[xx] = pyr_px(IR, pr, g.w, g.h, xr0 + xx, yl0 + yy);
#pragma unroll
          for (int k = 0; k < 11; ++k)
#pragma unroll
            for (int xx = 0; xx < 11; ++xx) { const int d = a[xx] - b[xx + k]; sad[k] += d < 0 ? -d : d; }
        }
        int best = 0x7fffffff, bestinc = 0;
#pragma unroll
        for (int k = 0; k < 11; ++k)
          if ((float)sad[k] < (float)best) { best = sad[k]; bestinc = k - Lw; }
        if (!(bestinc == -Lw || bestinc == Lw)) {
          float dist1 = 0, dist2 = 0, dist3 = 0;
#pragma unroll
          for (int k = 1; k < 10; ++k)
            if (k - Lw == bestinc) { dist1 = (float)sad[k - 1]; dist2 = (float)sad[k]; dist3 = (float)sad[k + 1]; }
          const float deltaR = __fdiv_rn(dist1 - dist3, 2.0f * (dist1 + dist3 - 2.0f * dist2));
          if (!(deltaR < -1 || deltaR > 1)) {
            float bestuR = st.scale[levelL] * (scaleduR0 + (float)bestinc + deltaR);
            float disparity = uL - bestuR;
            if (disparity >= minD && disparity < maxD) {
              if (disparity <= 0) {
                disparity = 0.01f;
                bestuR = (float)((double)uL - 0.01);
              }
              out_d = __fdiv_rn(mbf, disparity);
              out_u = bestuR;
              out_sad = best;
            }
          }
        }
      }
    }
  }
  uright[o] = out_u;
  depth[o] = out_d;
  sad_out[o] = out_sad;
}

// The median of the accepted SADs by two histogram passes (128-wide bins, then the values of the median's bin), each bin found
// by a workgroup prefix sum (round 6: work-item 0 walked the 256 resp. 128 counters one LDS read after the other, and the three
// passes each re-read the SADs from global memory - 14 us for a frame's 2000 keypoints; now the first 2048 stay in registers).
__global__ __launch_bounds__(256) void k_stereo_filter(const int32_t* __restrict__ nl, int cap, const int32_t* __restrict__ sad,
                                                       float* __restrict__ uright, float* __restrict__ depth) {
  __shared__ int s_hist[256];
  __shared__ int s_scan[8];
  __shared__ int s_bin, s_rank, s_nacc;
  const int f = blockIdx.x, tid = threadIdx.x;
  const int N = nl[f];
  const int32_t* S = sad + (size_t)f * cap;
  constexpr int kReg = 8;   // SADs per work-item kept in registers
  int v[kReg];
#pragma unroll
  for (int k = 0; k < kReg; ++k) { const int i = tid + 256 * k; v[k] = i < N ? S[i] : -1; }
  // f(value, index) for every keypoint of the frame this work-item looks after
  auto for_mine = [&](auto&& fn) {
#pragma unroll
    for (int k = 0; k < kReg; ++k) fn(v[k], tid + 256 * k);
    for (int i = tid + 256 * kReg; i < N; i += 256) fn(S[i], i);
  };
  // the bin that holds rank k of the histogram: the first b with hist[0] + .. + hist[b] > k (what the serial walk found)
  auto find_bin = [&](int k, int bins) {
    const int h = tid < bins ? s_hist[tid] : 0;
    int tot;
    const int ex = block_exclusive_scan<int>(h, s_scan, &tot);
    if (ex <= k && k < ex + h) { s_bin = tid; s_rank = k - ex; }
    __syncthreads();
  };
  s_hist[tid] = 0;
  if (tid == 0) s_nacc = 0;
  __syncthreads();
  int acc = 0;
  for_mine([&](int val, int) { if (val >= 0) { atomicAdd(&s_hist[imin(val >> 7, 255)], 1); ++acc; } });
  if (acc) atomicAdd(&s_nacc, acc);
  __syncthreads();
  const int nacc = s_nacc;
  if (nacc == 0) return;  // the reference would index an empty vector here (undefined behaviour)
  find_bin(nacc / 2, 256);
  const int b1 = s_bin, k1 = s_rank;
  __syncthreads();
  s_hist[tid] = 0;
  __syncthreads();
  for_mine([&](int val, int) { if (val >= 0 && imin(val >> 7, 255) == b1) atomicAdd(&s_hist[val & 127], 1); });
  __syncthreads();
  find_bin(k1, 128);
  const float median = (float)(b1 * 128 + s_bin);
  const float thDist = 1.5f * 1.4f * median;
  for_mine([&](int val, int i) {
    if (val >= 0 && !((float)val < thDist)) { uright[(size_t)f * cap + i] = -1.0f; depth[(size_t)f * cap + i] = -1.0f; }
  });
}

// ------------------------------------------------------------------------------------------------
// cv::cvtColor(COLOR_{RGB,BGR,RGBA,BGRA}2GRAY) on 8-bit images, the call in front of the extractor
// (Tracking::GrabImageRGBL, Tracking.cc:1567-1580).  OpenCV 4.x fixed point: 15-bit weights R 9798, G 19235, B 3735,
// gray = (c0*w0 + c1*GY + c2*w2 + 2^14) >> 15.  Work-item = 4 output pixels (12 or 16 source bytes as 32-bit words,
// one 32-bit store).  grid = (ceil(w/256), ceil(h/4), B), block = 256 (64 x 4).
template <int kChannels>
__global__ __launch_bounds__(256) void k_cvt_gray(const uint8_t* __restrict__ src, int spitch, size_t sframe, int w, int h,
                                                  int w0, int w2, uint8_t* __restrict__ dst, int dpitch, size_t dframe) {
  const int x0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x0 >= w || y >= h) return;
  const uint8_t* S = src + (size_t)blockIdx.z * sframe + (size_t)y * spitch + (size_t)x0 * kChannels;
  uint8_t* D = dst + (size_t)blockIdx.z * dframe + (size_t)y * dpitch + x0;
  constexpr int kGY = 19235;
  if (x0 + 3 < w) {
    uint32_t out = 0;
    if (kChannels == 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t q = load_u32_unaligned(S + 4 * i);
        const int v = (int)(((q & 0xff) * (uint32_t)w0 + ((q >> 8) & 0xff) * (uint32_t)kGY + ((q >> 16) & 0xff) * (uint32_t)w2 + (1u << 14)) >> 15);
        out |= (uint32_t)v << (8 * i);
      }
    } else {
      const uint32_t q0 = load_u32_unaligned(S), q1 = load_u32_unaligned(S + 4), q2 = load_u32_unaligned(S + 8);
      const uint32_t px[4] = {q0 & 0xffffffu, (q0 >> 24) | ((q1 & 0xffffu) << 8), (q1 >> 16) | ((q2 & 0xffu) << 16), q2 >> 8};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t q = px[i];
        const int v = (int)(((q & 0xff) * (uint32_t)w0 + ((q >> 8) & 0xff) * (uint32_t)kGY + ((q >> 16) & 0xff) * (uint32_t)w2 + (1u << 14)) >> 15);
        out |= (uint32_t)v << (8 * i);
      }
    }
    if ((dpitch & 3) == 0) *reinterpret_cast<uint32_t*>(D) = out;  // x0 is a multiple of 4
    else { D[0] = (uint8_t)out; D[1] = (uint8_t)(out >> 8); D[2] = (uint8_t)(out >> 16); D[3] = (uint8_t)(out >> 24); }
    return;
  }
  for (int i = 0; x0 + i < w; ++i) {  // row end: byte path, never reads past the last pixel
    const uint8_t* q = S + i * kChannels;
    D[i] = (uint8_t)(((uint32_t)q[0] * (uint32_t)w0 + (uint32_t)q[1] * (uint32_t)kGY + (uint32_t)q[2] * (uint32_t)w2 + (1u << 14)) >> 15);
  }
}

// unpacks candidate keys into rgbl_keypoint records (diagnostic path of rgbl_extractor_get_candidates)
// ------------------------------------------------------------------------------------------------
// Frame::UndistortKeyPoints (src/Frame.cc:837-870) = cv::undistortPoints(mat, mat, K, mDistCoef, cv::Mat(), mK):
// normalise, 5 fixed-point iterations of the inverse Brown-Conrady model (OpenCV's cvUndistortPointsInternal with its default
// TermCriteria(MAX_ITER, 5, 0.01), all in double, the operations in OpenCV's order), re-project with P = K.  SURVEY 8(f) row f3.
struct UndistortParams { double fx, fy, cx, cy, k[5]; };  // k1, k2, p1, p2, k3 (0 when mDistCoef has 4 entries)

__host__ __device__ __forceinline__ void undistort_point(const UndistortParams& U, float px, float py, float* ox, float* oy) {
  const double ifx = 1. / U.fx, ify = 1. / U.fy;
  double x = (double)px, y = (double)py;
  const double u = x, v = y;
  x = (x - U.cx) * ifx;
  y = (y - U.cy) * ify;
  const double x0 = x, y0 = y;
  for (int j = 0; j < 5; ++j) {
    const double r2 = x * x + y * y;
    // k[5..7] (the rational model) and k[8..11] (thin prism) are zero for the 4 / 5 coefficients ORB-SLAM3 passes
    const double icdist = (1 + ((0. * r2 + 0.) * r2 + 0.) * r2) / (1 + ((U.k[4] * r2 + U.k[1]) * r2 + U.k[0]) * r2);
    if (icdist < 0) { x = (u - U.cx) * ifx; y = (v - U.cy) * ify; break; }
    const double deltaX = 2 * U.k[2] * x * y + U.k[3] * (r2 + 2 * x * x) + 0. * r2 + 0. * r2 * r2;
    const double deltaY = U.k[2] * (r2 + 2 * y * y) + 2 * U.k[3] * x * y + 0. * r2 + 0. * r2 * r2;
    x = (x0 - deltaX) * icdist;
    y = (y0 - deltaY) * icdist;
  }
  // RR = P(3x3) * R = K: xx = fx x + 0 y + cx, yy = 0 x + fy y + cy, ww = 1 / (0 x + 0 y + 1)
  const double xx = U.fx * x + 0. * y + U.cx, yy = 0. * x + U.fy * y + U.cy, ww = 1. / (0. * x + 0. * y + 1.);
  *ox = (float)(xx * ww);
  *oy = (float)(yy * ww);
}

// rgbl_device_frame_capture: a frame's cv::KeyPoint records + descriptors + mvuRight as the structure-of-arrays the matcher
// kernels read (xy, octave, uright, descriptors).  One work-item per keypoint, the descriptor as two 16-byte words.
__global__ __launch_bounds__(256) void k_frame_capture(const rgbl_keypoint* __restrict__ kp, const uint8_t* __restrict__ desc, int n,
                                                       const float* __restrict__ uright, int write_xy, float* __restrict__ xy,
                                                       int32_t* __restrict__ oct, float* __restrict__ ur, uint8_t* __restrict__ out_desc) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n) return;
  const rgbl_keypoint k = kp[i];
  if (write_xy) { xy[2 * i] = k.x; xy[2 * i + 1] = k.y; }
  oct[i] = k.octave;
  ur[i] = uright ? uright[i] : -1.f;
  const uint4* src = reinterpret_cast<const uint4*>(desc + (size_t)i * 32);
  uint4* dst = reinterpret_cast<uint4*>(out_desc + (size_t)i * 32);
  dst[0] = src[0];
  dst[1] = src[1];
}

// in / out: strided (x, y) float pairs (stride in floats: 2 for plain arrays, 7 for rgbl_keypoint records); grid = (ceil(cap/256), B)
__global__ __launch_bounds__(256) void k_undistort(UndistortParams U, const float* __restrict__ in, int in_stride, size_t in_frame,
                                                   const int32_t* __restrict__ n_per_frame, int n_fixed, float* __restrict__ out,
                                                   int out_stride, size_t out_frame) {
  const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  const int n = n_per_frame ? n_per_frame[f] : n_fixed;
  if (i >= n) return;
  const float* p = in + (size_t)f * in_frame + (size_t)i * in_stride;
  float* q = out + (size_t)f * out_frame + (size_t)i * out_stride;
  float ox, oy;
  undistort_point(U, p[0], p[1], &ox, &oy);
  q[0] = ox;
  q[1] = oy;
}

__global__ void k_unpack_keys(const uint32_t* __restrict__ keys, int n, rgbl_keypoint* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t k = keys[i];
  rgbl_keypoint kp;
  kp.x = (float)key_x(k); kp.y = (float)key_y(k); kp.size = 7.f; kp.angle = -1.f;
  kp.response = (float)key_s(k); kp.octave = 0; kp.class_id = -1;
  out[i] = kp;
}

}  // namespace rgbl
