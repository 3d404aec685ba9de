// fused_level.h — one kernel per pyramid level that stages a tile of the level ONCE and produces from it
//   (a) the FAST candidates of the detection cells inside the tile   (ORBextractor.cc:781-896, cv::FAST 9/16 + NMS),
//   (b) the tile of the Gaussian working image                       (ORBextractor.cc:1132-1133, cv::GaussianBlur 7x7 s=2),
//   (c) the part of pyramid level l + 1 whose bilinear taps start in the tile (ORBextractor.cc:1170-1195, cv::resize).
// k_fast_cells, k_gauss7 and k_resize_linear (extractor_kernels.h) each staged the same pixels on their own; here the
// pixels cross HBM -> LDS once per level, and the whole is written for the VALU instruction count, which is what bounds it.
//
// Decomposition.  A tile = up to kFusedCells detection cells of one cell row (optionally, RGBL_FUSED_STRIPS=1, with cell-less
// strip tiles for the 19-px image border - fewer LDS bytes per workgroup, but measured slower, see extractor.hip).  Its cells' scanned areas
// [19 + j wCell, 19 + (j + 1) wCell) x [19 + i hCell, 19 + (i + 1) hCell) tile the FAST region without overlap, so every
// pixel's score is computed once; the Gaussian / resize "ownership" of a tile is the same rectangle with the left edge
// rounded down to a multiple of 4 (aligned 32-bit stores) and extended to the image border in the first / last tile
// column and row.  LDS holds the owned rectangle + 3 px (+ 8 px to the right for the resize windows), reflect-101 filled
// outside the image, and all three products read it with unaligned 32 / 64-bit LDS loads.
//
// Phases (one barrier each), each a sequence of flat task loops over all work-items (the second loop of a phase hands its
// first tasks to the work-items that idled in the last round of the first):
//   0  stage the tile in 16-byte pieces (all loads of a work-item requested before the first store); clear score map,
//      bitmaps, counters; copy the tile's resize table entries to LDS
//   1  FAST pre-screen, 4 pixels per task in packed 16-bit arithmetic (v_perm_b32 unpack, v_pk_min/max_u16)
//      | Gaussian rows, chunk 0: 7-tap sums of a row pair x 4 columns (v_dot4_u32_u8) -> LDS
//   2  exact FAST score of the survivors (packed 16-bit) | Gaussian columns, chunk 0 (v_dot2_u32_u16) -> HBM
//   3  3x3 NMS inside each cell over the survivors that got a score | resize tasks (4 output px: v_perm_b32 + v_dot2_u32_u16) -> HBM
//      | Gaussian rows, chunk 1
//   4  ordered compaction, one wave per cell | Gaussian columns, chunk 1
// The Gaussian of a tile runs in chunks of at most hCell owned rows: interior tiles have one chunk, the first and last
// tile row (which own the 19-px image border as well) two - the row-sum buffer stays small.
// Everything that is one value per tile column / tile row / level (extents, task counts, multiply-shift reciprocals) is
// computed on the host (extractor.hip: build_fused) and read through scalar loads.
// Arithmetic is the arithmetic of the three kernels this replaces (bit-exact with the oracle); the candidate lists leave in
// the same layout (cell_cnt / slots), so the quad-tree and descriptor kernels are unchanged.
#pragma once
#include "extractor_kernels.h"

namespace rgbl {

constexpr int kFusedCells = 4;      // max detection cells per tile (= waves that compact: one wave per cell)
constexpr int kFusedMaxTiles = 64;  // tile columns / rows per level
constexpr int kFusedHaloR = 8;      // columns staged right of the owned rectangle (resize windows are 8 bytes)

struct FusedCol {
  int xa, xb;        // owned image columns [xa, xb), xa % 4 == 0
  int j0, nc;        // detection cells j0 .. j0 + nc - 1
  int ga, gb;        // column groups (4 px) of level l + 1 whose first tap lies in [xa, xb)
  int sw, n_valid;   // scanned columns of the tile's valid cells side by side; number of valid cells
  int sx_t;          // tile column of the first scanned pixel
  int gpr;           // pre-screen groups per scanned row = ceil(sw / 4)
  int gw4;           // owned column groups = ceil((xb - xa) / 4)
  int q16, q_lo, q_hi;  // 16-byte pieces per tile row; [q_lo, q_hi) lie completely inside the image
  uint32_t m_gpr, m_sw, m_gw4, m_ng;  // multiply-shift reciprocals (div_magic) of gpr, sw, gw4, gb - ga
  int pad0, pad1;
};
struct FusedRow {
  int ya, yb;        // owned image rows
  int dya, dyb;      // rows of level l + 1 whose first tap lies in [ya, yb)
  int sh, sy_t;      // scanned rows of the cell row; tile row of the first scanned row
  int rows0, rows1;  // owned rows of the two Gaussian chunks
  int cell_row;      // detection-cell row of the tile row, -1 for a border strip (no cells)
  int pad0, pad1, pad2;
};
struct FusedTiles {
  int ntx, nty, ncx;
  // LDS layout of this level (bytes), sized for its largest tile
  int tile_pitch, tile_rows, off_tile;
  int hs_pitch, hs_pairs, off_hs, chunk_rows;
  int off_surv, off_score, score_pitch, off_keep, bit_words, off_misc, off_xt, off_xs, off_yt, lds_bytes;
  uint32_t m_wcell;
  FusedCol col[kFusedMaxTiles];
  FusedRow row[kFusedMaxTiles];
};

// resize table of the fused kernel, one entry per group of 4 output columns: the byte selectors that pull the two taps of
// an output out of the 8-byte source window (as two 16-bit halves) and the two 11-bit weights packed the same way
struct FusedXGroup { uint32_t sel[4]; uint32_t w[4]; };

struct FusedArgs {
  LevelGeom g;
  int next_w, next_h, next_pitch;                   // level l + 1 (next_w = 0: none)
  const FusedXGroup* xgrp; const int32_t* xsrc;      // per output column group: selectors / weights, first source column
  const ResizeTab* ytab;                             // cv::resize row table of level l + 1
  const uint8_t* src; int spitch; size_t sframe;
  uint8_t* next; size_t next_frame;
  uint8_t* blur; size_t blur_frame;
  uint32_t* cell_cnt; size_t cells_frame;
  uint32_t* slots; size_t slots_frame;
  int ini_th, min_th;
  const FusedTiles* tiles;
  unsigned long long* dbg;  // phase stamps (cycles summed over the workgroups of a launch), null unless RGBL_FUSED_STAMPS is set
};

// exact quotient x / d by a multiplication: floor(x * ceil(2^22 / d) / 2^22) == x / d while x * d < 2^22, and the 32-bit
// product does not overflow while x / d < 1000 (both checked on the host for every divisor a level uses)
__host__ __device__ inline uint32_t div_magic(uint32_t d) { return (0x400000u + d - 1u) / d; }
__device__ __forceinline__ int div_by(uint32_t x, uint32_t magic) { return (int)(__umul24(x, magic) >> 22); }

// FAST pre-screen of 4 horizontally adjacent pixels per task.  S = (tile column of the first scanned pixel) & 3, a
// compile-time constant per instantiation: the 4 bytes of ring sample (dx, dy) start S + dx bytes from an aligned word,
// so the pair of words that holds them and the byte selectors that split them into even / odd pixels (two 16-bit halves
// each) are all literals.
// The necessary condition of a 9-arc: of each of the four opposite ring pairs (0,8) (2,10) (4,12) (6,14) at least one
// pixel is darker than v - t (dark arc) resp. brighter than v + t.  With D = max over the pairs of min(a, b) and
// B = min over the pairs of max(a, b): survivor <=> v - D > t or B - v > t.
template <int NT, int TP, int S>
__device__ __forceinline__ void fused_prescreen(int first, int n_pre, int gpr, uint32_t m_gpr, int SW, uint32_t t2,
                                                const uint8_t* base /* aligned word of the first scanned pixel */,
                                                uint16_t* s_surv, int* s_count) {
  for (int task = first; task < n_pre; task += NT) {
    const int y = div_by((uint32_t)task, m_gpr), gx = task - (int)__umul24((uint32_t)y, (uint32_t)gpr);
    const uint8_t* c = base + y * TP + 4 * gx;
    uint32_t sv[2];
#define RGBL_SAMPLE(dx, dy, h)                                                                          \
  ([&]() -> uint32_t {                                                                                  \
    constexpr int b = S + (dx), q = (b + 4) / 4 - 1, r = b - 4 * q;                                     \
    const LdsPair w = lds_pair(c + (dy) * TP + 4 * q);                                                  \
    constexpr uint32_t sel = (uint32_t)(r + (h)) | (0x0cu << 8) | ((uint32_t)(r + (h) + 2) << 16) | (0x0cu << 24); \
    return perm_bytes(w.hi, w.lo, sel);                                                                 \
  })()
#pragma unroll
    for (int h = 0; h < 2; ++h) {  // even / odd pixels of the four as two 16-bit halves
      const uint32_t v = h ? RGBL_SAMPLE(0, 0, 1) : RGBL_SAMPLE(0, 0, 0);
      const uint32_t a0 = h ? RGBL_SAMPLE(0, 3, 1) : RGBL_SAMPLE(0, 3, 0), a8 = h ? RGBL_SAMPLE(0, -3, 1) : RGBL_SAMPLE(0, -3, 0);
      const uint32_t a4 = h ? RGBL_SAMPLE(3, 0, 1) : RGBL_SAMPLE(3, 0, 0), a12 = h ? RGBL_SAMPLE(-3, 0, 1) : RGBL_SAMPLE(-3, 0, 0);
      const uint32_t a2 = h ? RGBL_SAMPLE(2, 2, 1) : RGBL_SAMPLE(2, 2, 0), a10 = h ? RGBL_SAMPLE(-2, -2, 1) : RGBL_SAMPLE(-2, -2, 0);
      const uint32_t a6 = h ? RGBL_SAMPLE(2, -2, 1) : RGBL_SAMPLE(2, -2, 0), a14 = h ? RGBL_SAMPLE(-2, 2, 1) : RGBL_SAMPLE(-2, 2, 0);
      const uint32_t D = pk_max_u16(pk_max_u16(pk_min_u16(a0, a8), pk_min_u16(a4, a12)), pk_max_u16(pk_min_u16(a2, a10), pk_min_u16(a6, a14)));
      const uint32_t Bv = pk_min_u16(pk_min_u16(pk_max_u16(a0, a8), pk_max_u16(a4, a12)), pk_min_u16(pk_max_u16(a2, a10), pk_max_u16(a6, a14)));
      sv[h] = pk_subs_u16(pk_max_u16(pk_subs_u16(v, D), pk_subs_u16(Bv, v)), t2);
    }
#undef RGBL_SAMPLE
    if ((sv[0] | sv[1]) != 0) {
      const int x0 = 4 * gx, p0 = (int)__umul24((uint32_t)y, (uint32_t)SW) + x0;
      const bool f0 = (sv[0] & 0xffffu) != 0, f1 = (sv[1] & 0xffffu) != 0 && x0 + 1 < SW;
      const bool f2 = (sv[0] >> 16) != 0 && x0 + 2 < SW, f3 = (sv[1] >> 16) != 0 && x0 + 3 < SW;
      if (f0) s_surv[atomicAdd(s_count, 1)] = (uint16_t)p0;
      if (f1) s_surv[atomicAdd(s_count, 1)] = (uint16_t)(p0 + 1);
      if (f2) s_surv[atomicAdd(s_count, 1)] = (uint16_t)(p0 + 2);
      if (f3) s_surv[atomicAdd(s_count, 1)] = (uint16_t)(p0 + 3);
    }
  }
}

// NT = work-items per workgroup, TP = LDS tile pitch in bytes (a compile-time constant so that every ring / row offset is an
// immediate of the LDS instruction; the host picks the smallest instantiated pitch that holds the level's widest tile)
template <int NT, int TP>
__global__ __launch_bounds__(NT) void k_level_fused(FusedArgs A) {
  RGBL_DYN_SHARED(uint32_t, smem_w);
  uint8_t* smem = reinterpret_cast<uint8_t*>(smem_w);
  const FusedTiles& T = *A.tiles;
  const LevelGeom& g = A.g;
  const int tid = threadIdx.x, f = xcd_frame();  // grid = xcd_grid(tiles, B)
  const int rtid = NT - 1 - tid;  // second task loop of a phase: the work-items that idled last start first
  const int ntx = T.ntx;
  const int tr = xcd_item() / ntx, tc = xcd_item() - tr * ntx;
  const FusedCol& C = T.col[tc];
  const FusedRow& R = T.row[tr];
  const int H = g.h;
  uint8_t* s_tile = smem + T.off_tile;
  uint32_t* s_hs = reinterpret_cast<uint32_t*>(smem + T.off_hs);
  uint16_t* s_surv = reinterpret_cast<uint16_t*>(smem + T.off_surv);
  uint8_t* s_score = smem + T.off_score;
  uint32_t* s_keep = reinterpret_cast<uint32_t*>(smem + T.off_keep);  // [cell][2][bit_words]: min-threshold set, ini-threshold set
  uint4* s_xt = reinterpret_cast<uint4*>(smem + T.off_xt);
  int32_t* s_xs = reinterpret_cast<int32_t*>(smem + T.off_xs);
  ResizeTab* s_yt = reinterpret_cast<ResizeTab*>(smem + T.off_yt);
  int* s_misc = reinterpret_cast<int*>(smem + T.off_misc);            // 0 nsurv, 2.. any_ini per cell
  const int SP = T.score_pitch, BW = T.bit_words, GW = T.hs_pitch;

  const int xorg = C.xa - 4, yorg = R.ya - 3;
  const int own_h = R.yb - R.ya;
  const int rows_t = own_h + 6;
  const int SH = R.sh, SW = SH > 0 ? C.sw : 0;
  const int n_valid = SH > 0 ? C.n_valid : 0;
  const int sx_t = C.sx_t, sy_t = R.sy_t;
  const uint8_t* img = A.src + (size_t)f * A.sframe;
  unsigned long long t_prev = A.dbg ? rgbl_clock() : 0ull;
#define RGBL_FUSED_STAMP(i)                                                       \
  if (A.dbg && tid == 0) {                                                         \
    const unsigned long long t_now = rgbl_clock();                                 \
    atomicAdd(A.dbg + (i), t_now - t_prev);                                        \
    t_prev = t_now;                                                                \
  }

  // ---- phase 0: stage the tile.  16 lanes per tile row, 16 bytes per lane
  {
    constexpr int kRowsPerRound = NT / 16;
    const int q = tid & 15, y0 = tid >> 4;
    if (q >= C.q_lo && q < C.q_hi) {
      const uint8_t* gsrc = img + (xorg + 16 * q);
      uint8_t* ldst = s_tile + 16 * q;
      for (int yb = y0; yb < rows_t; yb += 4 * kRowsPerRound) {  // four rows per work-item and round, loads first
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int y = yb + u * kRowsPerRound;
          if (y < rows_t) {
            const int sy = reflect101(yorg + y, H);
            __builtin_memcpy(&v[u], gsrc + __umul24((uint32_t)sy, (uint32_t)A.spitch), 16);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int y = yb + u * kRowsPerRound;
          if (y < rows_t) *reinterpret_cast<uint4*>(ldst + y * TP) = v[u];
        }
      }
    } else if (q < C.q16) {
      // a piece that touches the left / right image border (first / last tile column only): reflected byte by byte
      const int W = g.w;
      for (int y = y0; y < rows_t; y += kRowsPerRound) {
        const uint8_t* row = img + __umul24((uint32_t)reflect101(yorg + y, H), (uint32_t)A.spitch);
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          w[k] = 0;
#pragma unroll
          for (int b = 0; b < 4; ++b) w[k] |= (uint32_t)row[reflect101(xorg + 16 * q + 4 * k + b, W)] << (8 * b);
        }
        *reinterpret_cast<uint4*>(s_tile + y * TP + 16 * q) = make_uint4(w[0], w[1], w[2], w[3]);
      }
    }
    for (int i = tid; i < ((SH + 2) * SP) >> 2; i += NT) reinterpret_cast<uint32_t*>(s_score)[i] = 0;
    for (int i = tid; i < C.nc * 2 * BW; i += NT) s_keep[i] = 0;
    if (tid < 8) s_misc[tid] = 0;
    if (A.next_w > 0) {  // resize table entries of the tile's output column groups and rows
      const int ng = C.gb - C.ga;
      const uint4* xsrc = reinterpret_cast<const uint4*>(A.xgrp + C.ga);
      for (int i = tid; i < 2 * ng; i += NT) s_xt[i] = xsrc[i];
      for (int i = tid; i < ng; i += NT) s_xs[i] = A.xsrc[C.ga + i];
      for (int i = tid; i < R.dyb - R.dya; i += NT) s_yt[i] = A.ytab[R.dya + i];
    }
  }
  __syncthreads();
  RGBL_FUSED_STAMP(0)

  // ---- Gaussian (arithmetic of k_gauss7): rows of a chunk -> 16-bit sums in LDS, then columns -> HBM
  const uint32_t kWA = 18u | (34u << 8) | (48u << 16) | (56u << 24), kWB = 48u | (34u << 8) | (18u << 16);
  const int gw4 = C.gw4;
  auto gauss_rows = [&](int first, int c0, int crows) {  // chunk = owned rows [c0, c0 + crows)
    const int n = ((crows + 7) >> 1) * gw4;               // row pairs x column groups
    for (int task = first; task < n; task += NT) {
      const int j = div_by((uint32_t)task, C.m_gw4), cg = task - (int)__umul24((uint32_t)j, (uint32_t)gw4);
      // pair j = tile rows c0 + 2 j, c0 + 2 j + 1 (tile row t = owned row t - 3); owned column 4 cg = tile column 4 + 4 cg
      const uint8_t* p = s_tile + (c0 + 2 * j) * TP + 4 * cg;
      uint32_t hs[2][4];
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const uint32_t w0 = *reinterpret_cast<const uint32_t*>(p + rr * TP), w1 = *reinterpret_cast<const uint32_t*>(p + rr * TP + 4),
                       w2 = *reinterpret_cast<const uint32_t*>(p + rr * TP + 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t lo = i == 3 ? w1 : align_bytes(w1, w0, i + 1);
          const uint32_t hi = i == 3 ? w2 : align_bytes(w2, w1, i + 1);
          hs[rr][i] = udot4(lo, kWA, udot4(hi, kWB, 0u));
        }
      }
      uint4 v;
      v.x = hs[0][0] | (hs[1][0] << 16); v.y = hs[0][1] | (hs[1][1] << 16);
      v.z = hs[0][2] | (hs[1][2] << 16); v.w = hs[0][3] | (hs[1][3] << 16);
      *reinterpret_cast<uint4*>(&s_hs[__umul24((uint32_t)j, (uint32_t)GW) + 4 * cg]) = v;
    }
  };
  auto gauss_cols = [&](int first, int c0, int crows) {
    const int n = ((crows + 3) >> 2) * gw4;
    const uint32_t kE0 = 18u | (34u << 16), kE1 = 48u | (56u << 16), kE2 = 48u | (34u << 16), kE3 = 18u;
    const uint32_t kO0 = 18u << 16, kO1 = 34u | (48u << 16), kO2 = 56u | (48u << 16), kO3 = 34u | (18u << 16);
    for (int task = first; task < n; task += NT) {
      const int rg = div_by((uint32_t)task, C.m_gw4), cg = task - (int)__umul24((uint32_t)rg, (uint32_t)gw4);
      uint32_t pv[5][4];
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const uint4 v = *reinterpret_cast<const uint4*>(&s_hs[__umul24((uint32_t)(2 * rg + j), (uint32_t)GW) + 4 * cg]);
        pv[j][0] = v.x; pv[j][1] = v.y; pv[j][2] = v.z; pv[j][3] = v.w;
      }
      // the row pitch is a multiple of 64 and the column a multiple of 4: a full word always fits the row (what lands in
      // the padding right of the last column is never read)
      uint8_t* D = A.blur + (size_t)f * A.blur_frame +
                   (__umul24((uint32_t)(R.ya + c0 + 4 * rg), (uint32_t)g.pitch) + (uint32_t)(C.xa + 4 * cg));
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        if (4 * rg + o >= crows) break;
        const int b = o >> 1;
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
          // acc < 2^24 and its byte 2 is the output pixel: one byte shuffle per pixel instead of shift + or
          out = i == 0 ? acc >> 16 : perm_bytes(acc, out, i == 1 ? 0x0c0c0600u : i == 2 ? 0x0c060100u : 0x06020100u);
        }
        *reinterpret_cast<uint32_t*>(D + __umul24((uint32_t)o, (uint32_t)g.pitch)) = out;
      }
    }
  };
  const int rows0 = R.rows0, rows1 = R.rows1;

  // ---- phase 1: FAST pre-screen (4 pixels per task) | Gaussian rows of chunk 0
  {
    const int n_pre = C.gpr * SH;
    const uint32_t t2 = (uint32_t)A.min_th | ((uint32_t)A.min_th << 16);
    const uint8_t* base = s_tile + sy_t * TP + (sx_t & ~3);
    switch (sx_t & 3) {  // uniform per tile
      case 0: fused_prescreen<NT, TP, 0>(tid, n_pre, C.gpr, C.m_gpr, SW, t2, base, s_surv, &s_misc[0]); break;
      case 1: fused_prescreen<NT, TP, 1>(tid, n_pre, C.gpr, C.m_gpr, SW, t2, base, s_surv, &s_misc[0]); break;
      case 2: fused_prescreen<NT, TP, 2>(tid, n_pre, C.gpr, C.m_gpr, SW, t2, base, s_surv, &s_misc[0]); break;
      default: fused_prescreen<NT, TP, 3>(tid, n_pre, C.gpr, C.m_gpr, SW, t2, base, s_surv, &s_misc[0]); break;
    }
    gauss_rows(rtid, 0, rows0);
  }
  __syncthreads();
  RGBL_FUSED_STAMP(1)

  // ---- phase 2: exact score of the survivors | Gaussian columns of chunk 0
  const int nsurv = s_misc[0];
  for (int task = tid; task < nsurv; task += NT) {
    const int p = s_surv[task];
    const int y = div_by((uint32_t)p, C.m_sw), x = p - (int)__umul24((uint32_t)y, (uint32_t)SW);
    const int sc = fast_true_score_pk(s_tile + (sy_t + y) * TP + sx_t + x, TP);
    if (sc >= A.min_th) s_score[__umul24((uint32_t)(y + 1), (uint32_t)SP) + x + 1] = (uint8_t)sc;
  }
  gauss_cols(rtid, 0, rows0);
  __syncthreads();
  RGBL_FUSED_STAMP(2)
  if (A.dbg && tid == 0) atomicAdd(A.dbg + 6, (unsigned long long)nsurv);

  // ---- phase 3: NMS inside each cell | resize tasks | Gaussian rows of chunk 1
  {
    for (int task = tid; task < nsurv; task += NT) {  // the survivors that got a score are the corners
      const int p = s_surv[task];
      const int y = div_by((uint32_t)p, C.m_sw), x = p - (int)__umul24((uint32_t)y, (uint32_t)SW);
      const uint8_t* s = s_score + __umul24((uint32_t)(y + 1), (uint32_t)SP) + x + 1;
      const int v = s[0];
      if (v == 0) continue;
      const int k = div_by((uint32_t)x, T.m_wcell);  // cell of the tile
      const int kx = (int)__umul24((uint32_t)k, (uint32_t)g.w_cell);
      const int xc = x - kx;
      const int swc = imin(g.w_cell, SW - kx);
      // neighbours outside the pixel's own cell count as 0 (cv::FAST ran on the cell's sub-image)
      const bool lft = xc > 0, rgt = xc < swc - 1;
      bool keep = v > s[-SP] && v > s[SP];
      if (lft) keep = keep && v > s[-1] && v > s[-SP - 1] && v > s[SP - 1];
      if (rgt) keep = keep && v > s[1] && v > s[-SP + 1] && v > s[SP + 1];
      if (keep) {
        const int bit = (int)__umul24((uint32_t)y, (uint32_t)swc) + xc;
        atomicOr(&s_keep[(2 * k) * BW + (bit >> 5)], 1u << (bit & 31));
        if (v >= A.ini_th) { atomicOr(&s_keep[(2 * k + 1) * BW + (bit >> 5)], 1u << (bit & 31)); s_misc[2 + k] = 1; }
      }
    }
    if (A.next_w > 0) {
      // cv::resize INTER_LINEAR, 4 output pixels per task (arithmetic of k_resize_linear): the two taps of an output are
      // pulled out of the 8-byte source window as 16-bit halves and weighted by one v_dot2_u32_u16 per source row
      const int n_groups = C.gb - C.ga, n_rs = n_groups * (R.dyb - R.dya);
      uint8_t* dst = A.next + (size_t)f * A.next_frame + 4 * C.ga;
      for (int task = rtid; task < n_rs; task += NT) {
        const int ry = div_by((uint32_t)task, C.m_ng), gi = task - (int)__umul24((uint32_t)ry, (uint32_t)n_groups);
        const uint4 sel = s_xt[2 * gi], wt = s_xt[2 * gi + 1];
        const int sxa = s_xs[gi];
        const ResizeTab ty = s_yt[ry];
        const int sy0 = imin(imax(ty.sofs, 0), H - 1), sy1 = imin(imax(ty.sofs + 1, 0), H - 1);
        const int col = sxa - xorg, sh = col & 3;  // the 8-byte window starts sh bytes into an aligned word
        const uint8_t* S0 = s_tile + (sy0 - yorg) * TP + (col & ~3);
        const uint8_t* S1 = s_tile + (sy1 - yorg) * TP + (col & ~3);
        const LdsPair p0 = lds_pair(S0), p1 = lds_pair(S1);
        const uint32_t t0 = *reinterpret_cast<const uint32_t*>(S0 + 8), t1 = *reinterpret_cast<const uint32_t*>(S1 + 8);
        const uint32_t r0l = align_bytes(p0.hi, p0.lo, sh), r0h = align_bytes(t0, p0.hi, sh);
        const uint32_t r1l = align_bytes(p1.hi, p1.lo, sh), r1h = align_bytes(t1, p1.hi, sh);
        const uint32_t b0 = (uint32_t)ty.a0, b1 = (uint32_t)ty.a1;
        const uint32_t sl[4] = {sel.x, sel.y, sel.z, sel.w}, wl[4] = {wt.x, wt.y, wt.z, wt.w};
        uint32_t out = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t h0 = udot2(perm_bytes(r0h, r0l, sl[i]), wl[i], 0u);
          const uint32_t h1 = udot2(perm_bytes(r1h, r1l, sl[i]), wl[i], 0u);
          const uint32_t v = ((__umul24(b0, h0 >> 4) >> 16) + (__umul24(b1, h1 >> 4) >> 16) + 2u) >> 2;
          out |= (v & 0xffu) << (8 * i);
        }
        *reinterpret_cast<uint32_t*>(dst + (__umul24((uint32_t)(R.dya + ry), (uint32_t)A.next_pitch) + (uint32_t)(4 * gi))) = out;
      }
    }
    gauss_rows(tid, rows0, rows1);
  }
  __syncthreads();
  RGBL_FUSED_STAMP(3)

  // ---- phase 4: ordered compaction.  cv::FAST emits a cell's keypoints in row-major order = ascending bit index of the
  // cell's bitmap, so a kept corner's slot is the number of set bits before its own: (a) one wave per cell turns the word
  // popcounts of the chosen bitmap (two-threshold rule, ORBextractor.cc:826-846: the ini-threshold set if it is not empty,
  // else the min-threshold set) into exclusive prefixes, (b) after a barrier every work-item ranks the corners among its
  // survivors with prefix + popcount(lower bits of the word) and writes their keys.  No serial bit loops.
  // | Gaussian columns of chunk 1 next to (a)
  const int wave = wave_id(), lane = lane_id();
  uint32_t* s_prefix = s_keep + 2 * kFusedCells * BW;  // [cell][bit_words]
  if (wave < C.nc && R.cell_row >= 0) {
    const int k = wave;
    const int ci = R.cell_row * g.n_cols + C.j0 + k;
    uint32_t* my_cnt = A.cell_cnt + (size_t)f * A.cells_frame + g.cell_off + ci;
    if (k >= n_valid) {
      if (lane == 0) *my_cnt = 0;
    } else {
      const int swc = imin(g.w_cell, SW - k * g.w_cell);
      const int nwords = (swc * SH + 31) >> 5;
      const uint32_t* keep = s_keep + (2 * k + (s_misc[2 + k] ? 1 : 0)) * BW;
      uint32_t total = 0;
      for (int w0 = 0; w0 < nwords; w0 += 64) {
        const int t = w0 + lane;
        const uint32_t cnt = t < nwords ? (uint32_t)__popc(keep[t]) : 0u;
        uint32_t incl = cnt;
        for (int d = 1; d < 64; d <<= 1) {
          const uint32_t up = __shfl_up(incl, d);
          if (lane >= d) incl += up;
        }
        if (t < nwords) s_prefix[k * BW + t] = total + incl - cnt;
        total += __shfl(incl, 63);
      }
      const uint32_t cap = (uint32_t)g.cell_cap;  // a proven bound of strict 3x3 NMS
      if (lane == 0) *my_cnt = total < cap ? total : cap;
    }
  }
  if (rows1 > 0) gauss_cols(rtid, rows0, rows1);
  __syncthreads();
  RGBL_FUSED_STAMP(4)
  for (int task = tid; task < nsurv; task += NT) {
    const int p = s_surv[task];
    const int y = div_by((uint32_t)p, C.m_sw), x = p - (int)__umul24((uint32_t)y, (uint32_t)SW);
    const int sc = s_score[__umul24((uint32_t)(y + 1), (uint32_t)SP) + x + 1];
    if (sc == 0) continue;
    const int k = div_by((uint32_t)x, T.m_wcell);
    const int kx = (int)__umul24((uint32_t)k, (uint32_t)g.w_cell);
    const int xc = x - kx;
    const int swc = imin(g.w_cell, SW - kx);
    const int bit = (int)__umul24((uint32_t)y, (uint32_t)swc) + xc;
    const uint32_t word = s_keep[(2 * k + (s_misc[2 + k] ? 1 : 0)) * BW + (bit >> 5)];
    if (!((word >> (bit & 31)) & 1u)) continue;
    const uint32_t pos = s_prefix[k * BW + (bit >> 5)] + (uint32_t)__popc(word & ((1u << (bit & 31)) - 1u));
    if (pos < (uint32_t)g.cell_cap) {
      const int ci = R.cell_row * g.n_cols + C.j0 + k;
      A.slots[(size_t)f * A.slots_frame + g.slot_off + (size_t)ci * g.cell_cap + pos] =
          pack_key((C.j0 + k) * g.w_cell + 3 + xc, R.cell_row * g.h_cell + 3 + y, sc);
    }
  }
  RGBL_FUSED_STAMP(5)
  if (A.dbg && tid == 0) atomicAdd(A.dbg + 7, 1ull);
#undef RGBL_FUSED_STAMP
}

}  // namespace rgbl
