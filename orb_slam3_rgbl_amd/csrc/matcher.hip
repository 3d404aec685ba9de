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
//   k_search_triangulation one workgroup per BoW node shared by both key-frames, both buckets staged in LDS, one
//                         query feature per wave, the candidates over its lanes (ties -> later candidate wins, as
//                         in the reference) with the eligibility masks and epipolar test.
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <utility>
#include <vector>

#include "common.h"

namespace rgbl {

__device__ __forceinline__ int hamming256(const unsigned long long q[4], const unsigned long long* __restrict__ t) {
  // one accumulating chain of eight 32-bit popcounts (v_bcnt_u32_b32 adds its second operand for free); four 64-bit
  // popcounts cost the same eight v_bcnt plus two adds for the partial sums
  // (written as inline assembly: the compiler re-balances a chain of ctpop + add into a tree with three v_add3 per distance)
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned long long x = q[k] ^ t[k];
#ifdef RGBL_EMU
    acc += (uint32_t)__builtin_popcountll(x);
#else
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(acc) : "v"((uint32_t)x), "v"(acc));
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(acc) : "v"((uint32_t)(x >> 32)), "v"(acc));
#endif
  }
  return (int)acc;
}

// A workgroup owns 64 query descriptors (one per lane, 4 x u64 in VGPRs).  Its four waves split the train set
// into four contiguous index ranges; inside a wave the train descriptor address is wave-uniform, so it is
// fetched with scalar loads and broadcast to the 64 lanes.  Best and second-best are tracked on packed words
// (distance << 16 | train index): min() then implements the reference's "strict '<', first minimum wins" and the
// second-best is the second smallest word, whose distance part is the second smallest distance counted with
// multiplicity.  The four partial results are merged through LDS with the same two operations.
// grid = (ceil(cap/64), n_pairs), block = 256.  Requires train counts < 65536.
__device__ __forceinline__ void bf_track(uint32_t& best, uint32_t& second, uint32_t cur) {
  // best <= second always: the new second is the median of (best, cur, second) - one v_med3_u32 instead of max + min
#ifdef RGBL_EMU
  const uint32_t hi = best > cur ? best : cur;
  second = second < hi ? second : hi;
#else
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(second) : "v"(best), "v"(cur), "v"(second));
#endif
  best = best < cur ? best : cur;
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
    unsigned long long tt[32];  // the eight train descriptors first (wave-uniform: four s_load_dwordx16), then the arithmetic
#pragma unroll
    for (int u = 0; u < 32; ++u) tt[u] = t[u];
    uint32_t e[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) e[u] = ((uint32_t)hamming256(q, tt + 4 * u) << 16) | (uint32_t)(j + u);
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

// ------------------------------------------------------------------------------------------------
// The same all-pairs scan on the matrix cores.  With the bits of a descriptor expanded to signed bytes,
//   train bit a -> 64 (1 - 2a),   query bit b -> -64 (1 - 2b),
// a 256-long i8 dot product is 4096 (#different - #equal) = 8192 * hamming - 2^20: v_mfma_i32_32x32x32_i8 delivers, per
// instruction, 32 of the 256 bit positions of 32 train x 32 query pairs, eight chained instructions the whole distance.
// The accumulator starts at 2^20 + (row of the tile), so what comes out IS the tracking key of k_hamming_bf - distance
// above the index, here (distance << 13 | row) - and the epilogue is v_med3 + v_min per pair, nothing else.
//   * Tile t's rows are 32 t + row; instead of adding 32 t to 16 accumulators the two running keys of a lane are lowered by
//     32 per tile (signed compares), which orders (distance, index) pairs exactly as the absolute keys would; 13 index bits
//     = sweeps of 8192 train descriptors, decoded and merged into (distance << 16 | index) words after each sweep.
//   * The position of a bit inside the K = 256 sum is free as long as both operands agree, so the expansion is chosen for
//     the VALU: dword w of a descriptor feeds MFMA step w, and byte c of its q-th expanded dword is bit 8c + q of it:
//     ((x << (7 - q)) & 0x80808080) | 0x40404040 (0xC0 = -64, 0x40 = +64), two operations per four operand bytes.
//   * Query tiles (the B operand, the D columns) live in VGPRs for the whole scan: 64 queries per wave, 256 per workgroup.
//     Train rows are expanded once per workgroup into LDS, 64 rows per stage, double buffered (one barrier per stage), rows
//     272 B apart so that the 16-byte fragment reads of a wave spread over all banks; every fragment feeds two MFMAs.
// D layout (gfx950, all 32 x 32 shapes): lane l holds column l & 31, register r row (r & 3) + 8 (r >> 2) + 4 (l >> 5).
// grid = (ceil(cap / 256), n_pairs), block = 256.  Same arguments and results as k_hamming_bf.
constexpr int kBfQueriesPerBlock = 256;
constexpr int kBfStageRows = 64;
constexpr int kBfRowBytes = 272;
constexpr int kBfSweep = 8192;
constexpr int kBfIdle = 0x3fffffff;

__device__ __forceinline__ v4i bf_expand4(uint32_t x, int q0) {
  v4i r;
  r[0] = (int)(((x << (7 - q0)) & 0x80808080u) | 0x40404040u);
  r[1] = (int)(((x << (6 - q0)) & 0x80808080u) | 0x40404040u);
  r[2] = (int)(((x << (5 - q0)) & 0x80808080u) | 0x40404040u);
  r[3] = (int)(((x << (4 - q0)) & 0x80808080u) | 0x40404040u);
  return r;
}
__device__ __forceinline__ void bf_track_rel(int& best, int& second, int cur) {
#ifdef RGBL_EMU
  const int hi = best > cur ? best : cur;
  second = second < hi ? second : hi;
#else
  asm("v_med3_i32 %0, %1, %2, %3" : "=v"(second) : "v"(best), "v"(cur), "v"(second));
#endif
  best = best < cur ? best : cur;
}
__device__ __forceinline__ void bf_merge(uint32_t& best, uint32_t& second, uint32_t ob, uint32_t os) {
  const uint32_t hi = best > ob ? best : ob;
  best = best < ob ? best : ob;
  second = second < os ? second : os;
  second = second < hi ? second : hi;
}

__global__ __launch_bounds__(256) void k_hamming_mfma(const uint8_t* __restrict__ desc, const int32_t* __restrict__ n_rows,
                                                      int cap, const int32_t* __restrict__ pair_a,
                                                      const int32_t* __restrict__ pair_b, int32_t* __restrict__ best_idx,
                                                      int32_t* __restrict__ best_dist, int32_t* __restrict__ second_dist) {
  __shared__ __attribute__((aligned(16))) uint8_t s_rows[2][kBfStageRows * kBfRowBytes];
  // grid = xcd_grid(query blocks, pairs) (common.h): the query blocks of a pair share an XCD's L2 and with it the train set
  const int p = xcd_frame();
  const int fa = pair_a ? pair_a[p] : 0, fb = pair_b ? pair_b[p] : 1;
  const int na = n_rows[fa], nb = n_rows[fb];
  const int q_base = xcd_item() * kBfQueriesPerBlock;
  if (q_base >= na) return;
  const int tid = threadIdx.x, lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(wave_id());
  const int col = lane & 31, half = lane >> 5;
  const bool wave_has_queries = q_base + wave * 64 < na;

  // query fragments: two 32-column tiles, eight K steps each
  v4i bq[2][8];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int qi = q_base + wave * 64 + ct * 32 + col;
    const uint4* src = reinterpret_cast<const uint4*>(desc + ((size_t)fa * cap + (qi < na ? qi : 0)) * 32);
    const uint4 lo = src[0], hi = src[1];
    const uint32_t x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int s = 0; s < 8; ++s) bq[ct][s] = bf_expand4(~x[s], 4 * half);
  }
  v16i c_init;
#pragma unroll
  for (int r = 0; r < 16; ++r) c_init[r] = (1 << 20) + (r & 3) + 8 * (r >> 2) + 4 * half;

  const uint32_t* __restrict__ train = reinterpret_cast<const uint32_t*>(desc + (size_t)fb * cap * 32);
  const uint32_t kNone = (256u << 16) | 0xffffu;
  uint32_t out_best[2] = {kNone, kNone}, out_second[2] = {kNone, kNone};

  // stage s of the whole scan = train rows [64 s, 64 s + 64); work-item tid expands dwords tid and tid + 256 of it
  const int n_stages = (nb + kBfStageRows - 1) / kBfStageRows;
  auto load_stage = [&](int stage, uint32_t& x0, uint32_t& x1) {
    const int d0 = stage * (kBfStageRows * 8) + tid, d1 = d0 + 256;
    x0 = (d0 >> 3) < nb ? train[d0] : 0u;
    x1 = (d1 >> 3) < nb ? train[d1] : 0u;
  };
  uint32_t nx0 = 0, nx1 = 0;
  if (n_stages > 0) load_stage(0, nx0, nx1);
  int best[2] = {kBfIdle, kBfIdle}, second[2] = {kBfIdle, kBfIdle};
  int tiles_in_sweep = 0;
  auto close_sweep = [&](int sweep_base) {
    // relative keys -> (distance << 16 | absolute index), folded into the results of the earlier sweeps
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int off = 32 * (tiles_in_sweep - 1);
      const int kb = best[ct] + off, ks = second[ct] + off;
      const uint32_t wb = kb >= (257 << 13) ? kNone : ((uint32_t)(kb >> 13) << 16) | (uint32_t)(sweep_base + (kb & 8191));
      const uint32_t ws = ks >= (257 << 13) ? kNone : ((uint32_t)(ks >> 13) << 16) | (uint32_t)(sweep_base + (ks & 8191));
      bf_merge(out_best[ct], out_second[ct], wb, ws);
      best[ct] = second[ct] = kBfIdle;
    }
    tiles_in_sweep = 0;
  };
  for (int stage = 0; stage < n_stages; ++stage) {
    uint8_t* buf = s_rows[stage & 1];
    const uint32_t x0 = nx0, x1 = nx1;
    if (stage + 1 < n_stages) load_stage(stage + 1, nx0, nx1);
    {
      uint8_t* r0 = buf + (tid >> 3) * kBfRowBytes + (tid & 7) * 32;
      uint8_t* r1 = r0 + 32 * kBfRowBytes;
      *reinterpret_cast<v4i*>(r0) = bf_expand4(x0, 0);
      *reinterpret_cast<v4i*>(r0 + 16) = bf_expand4(x0, 4);
      *reinterpret_cast<v4i*>(r1) = bf_expand4(x1, 0);
      *reinterpret_cast<v4i*>(r1 + 16) = bf_expand4(x1, 4);
    }
    __syncthreads();
    if (!wave_has_queries) continue;
#pragma unroll 1
    for (int rt = 0; rt < 2; ++rt) {
      const int row0 = stage * kBfStageRows + rt * 32;  // first train row of the tile
      if (row0 >= nb) break;
      const uint8_t* frag = buf + (rt * 32 + col) * kBfRowBytes + half * 16;
      v16i acc0 = c_init, acc1 = c_init;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const v4i a = *reinterpret_cast<const v4i*>(frag + s * 32);
        acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[0][s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[1][s], acc1, 0, 0, 0);
      }
      best[0] -= 32; second[0] -= 32; best[1] -= 32; second[1] -= 32;
      ++tiles_in_sweep;
      if (row0 + 32 <= nb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { bf_track_rel(best[0], second[0], acc0[r]); bf_track_rel(best[1], second[1], acc1[r]); }
      } else {  // the last tile of the train set: rows beyond it never win
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool live = row0 + (r & 3) + 8 * (r >> 2) + 4 * half < nb;
          bf_track_rel(best[0], second[0], live ? acc0[r] : kBfIdle);
          bf_track_rel(best[1], second[1], live ? acc1[r] : kBfIdle);
        }
      }
      if (((row0 + 32) & (kBfSweep - 1)) == 0) close_sweep(row0 + 32 - kBfSweep);
    }
  }
  if (!wave_has_queries) return;
  if (tiles_in_sweep > 0) close_sweep(((nb - 1) / kBfSweep) * kBfSweep);
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    // the two halves of the wave hold the same columns, interleaved groups of four rows
    const uint32_t ob = __shfl_xor(out_best[ct], 32), os = __shfl_xor(out_second[ct], 32);
    bf_merge(out_best[ct], out_second[ct], ob, os);
    const int qi = q_base + wave * 64 + ct * 32 + col;
    if (half == 0 && qi < na) {
      const size_t o = (size_t)p * cap + qi;
      best_idx[o] = out_best[ct] == kNone ? -1 : (int32_t)(out_best[ct] & 0xffffu);
      best_dist[o] = (int32_t)(out_best[ct] >> 16);
      if (second_dist) second_dist[o] = (int32_t)(out_second[ct] >> 16);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The same scan once more, on gfx950's block-scaled FP4 matrix-core instruction (v_mfma_scale_f32_32x32x64_f8f6f4 with
// E2M1 operands: 64 bit positions per instruction, twice the i8 rate, half the operand bytes).  A descriptor bit becomes the
// four-bit float +-4 (0x6 / 0xE): train bit a -> 4 (1 - 2a), query bit b -> -4 (1 - 2b); both operands carry the block scale
// 2^4 (E8M0 byte 131), so a product is +-4096 and the 256-long sum is 8192 * hamming - 2^20 - the numbers of the i8 form,
// exact in the f32 accumulator (all values are integers below 2^22).  The accumulator starts at 2^20 + row, the tracking
// key comes out as a float and is tracked with v_med3_f32 / v_min_f32.  Dword w of a descriptor is the 32 values one lane
// group feeds to step w / 2 (group w & 1); which bit lands in which nibble of the group does not matter as long as train
// and query agree.  Everything else - query tiles in VGPRs, 64 train rows per double-buffered LDS stage, relative keys,
// sweeps - as in k_hamming_mfma.  grid = (ceil(cap / 256), n_pairs), block = 256.
constexpr int kBfRowBytesF4 = 144;  // 128 bytes of nibbles + 16: the 16-byte fragment reads of a wave spread over the banks
constexpr float kBfIdleF = 1073741824.f;

__device__ __forceinline__ v4i bf_expand_fp4(uint32_t x) {  // bit j of x -> nibble j: 0x6 (+4) or 0xE (-4)
  v4i r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint32_t t = (x >> (8 * i)) & 0xffu;
    t = (t | (t << 12)) & 0x000f000fu;
    t = (t | (t << 6)) & 0x03030303u;
    t = (t | (t << 3)) & 0x11111111u;
    r[i] = (int)(0x66666666u | (t << 3));
  }
  return r;
}
__device__ __forceinline__ void bf_track_rel_f(float& best, float& second, float cur) {
#ifdef RGBL_EMU
  const float hi = best > cur ? best : cur;
  second = second < hi ? second : hi;
#else
  asm("v_med3_f32 %0, %1, %2, %3" : "=v"(second) : "v"(best), "v"(cur), "v"(second));
#endif
  best = fminf(best, cur);  // one v_min_f32 (a compare + select under strict floating-point rules otherwise); no NaNs here
}

__global__ __launch_bounds__(256) void k_hamming_fp4(const uint8_t* __restrict__ desc, const int32_t* __restrict__ n_rows,
                                                     int cap, const int32_t* __restrict__ pair_a,
                                                     const int32_t* __restrict__ pair_b, int32_t* __restrict__ best_idx,
                                                     int32_t* __restrict__ best_dist, int32_t* __restrict__ second_dist,
                                                     int splits, uint32_t* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) uint8_t s_rows[2][kBfStageRows * kBfRowBytesF4];
  __shared__ uint32_t s_lut[256];  // byte -> its eight nibbles: the train rows are expanded with four table reads per dword
  // splits > 1 (one pair per call, rgbl_hamming_bf): the launch's frame index is a slice of the train set instead of a pair;
  // every slice leaves its packed best / second per query in `partial`, k_hamming_merge folds them (a frame against a frame is
  // 8 query blocks: alone they occupy 8 of 256 CUs for 43 us)
  const int split = splits > 1 ? xcd_frame() : 0;
  const int p = splits > 1 ? 0 : xcd_frame();
  const int fa = pair_a ? pair_a[p] : 0, fb = pair_b ? pair_b[p] : 1;
  const int na = n_rows[fa], nb = n_rows[fb];
  const int q_base = xcd_item() * kBfQueriesPerBlock;
  if (q_base >= na) return;
  const int tid = threadIdx.x, lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(wave_id());
  const int col = lane & 31, half = lane >> 5;
  const bool wave_has_queries = q_base + wave * 64 < na;
  constexpr int kScale = 131;  // E8M0: 2^(131 - 127) = 16 per operand
  s_lut[tid] = (uint32_t)bf_expand_fp4((uint32_t)tid)[0];  // the first barrier of the stage loop publishes it

  // query fragments: two 32-column tiles, four K steps each
  v8i bq[2][4];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int qi = q_base + wave * 64 + ct * 32 + col;
    const uint4* src = reinterpret_cast<const uint4*>(desc + ((size_t)fa * cap + (qi < na ? qi : 0)) * 32);
    const uint4 lo = src[0], hi = src[1];
    const uint32_t x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const v4i e = bf_expand_fp4(~(half ? x[2 * s + 1] : x[2 * s]));
      bq[ct][s] = v8i{e[0], e[1], e[2], e[3], 0, 0, 0, 0};
    }
  }
  v16f c_init;
#pragma unroll
  for (int r = 0; r < 16; ++r) c_init[r] = (float)((1 << 20) + (r & 3) + 8 * (r >> 2) + 4 * half);

  const uint32_t* __restrict__ train = reinterpret_cast<const uint32_t*>(desc + (size_t)fb * cap * 32);
  const uint32_t kNone = (256u << 16) | 0xffffu;
  uint32_t out_best[2] = {kNone, kNone}, out_second[2] = {kNone, kNone};

  // stage s of the whole scan = train rows [64 s, 64 s + 64); work-item tid expands dwords tid and tid + 256 of it
  const int all_stages = (nb + kBfStageRows - 1) / kBfStageRows;
  const int stages_per = (all_stages + splits - 1) / (splits > 1 ? splits : 1);
  const int s_begin = split * stages_per, n_stages = imin(all_stages, s_begin + stages_per);  // this launch slice: stages [s_begin, n_stages)
  auto load_stage = [&](int stage, uint32_t& x0, uint32_t& x1) {
    const int d0 = stage * (kBfStageRows * 8) + tid, d1 = d0 + 256;
    x0 = (d0 >> 3) < nb ? train[d0] : 0u;
    x1 = (d1 >> 3) < nb ? train[d1] : 0u;
  };
  uint32_t nx0 = 0, nx1 = 0;
  if (n_stages > s_begin) load_stage(s_begin, nx0, nx1);
  float best[2] = {kBfIdleF, kBfIdleF}, second[2] = {kBfIdleF, kBfIdleF};
  int tiles_in_sweep = 0;
  int sweep_start = s_begin * kBfStageRows;  // first train row of the sweep in progress
  auto close_sweep = [&](int sweep_base) {
    // relative keys -> (distance << 16 | absolute index), folded into the results of the earlier sweeps
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const float off = (float)(32 * (tiles_in_sweep - 1));
      const int kb = (int)(best[ct] + off), ks = (int)(second[ct] + off);  // exact: integers below 2^31 / idle 2^30 + small
      const uint32_t wb = kb >= (257 << 13) ? kNone : ((uint32_t)(kb >> 13) << 16) | (uint32_t)(sweep_base + (kb & 8191));
      const uint32_t ws = ks >= (257 << 13) ? kNone : ((uint32_t)(ks >> 13) << 16) | (uint32_t)(sweep_base + (ks & 8191));
      bf_merge(out_best[ct], out_second[ct], wb, ws);
      best[ct] = second[ct] = kBfIdleF;
    }
    tiles_in_sweep = 0;
  };
  for (int stage = s_begin; stage < n_stages; ++stage) {
    uint8_t* buf = s_rows[stage & 1];
    const uint32_t x0 = nx0, x1 = nx1;
    if (stage + 1 < n_stages) load_stage(stage + 1, nx0, nx1);
    {
      uint8_t* r0 = buf + (tid >> 3) * kBfRowBytesF4 + (tid & 7) * 16;
      if (stage == s_begin) {  // the table is not published yet
        *reinterpret_cast<v4i*>(r0) = bf_expand_fp4(x0);
        *reinterpret_cast<v4i*>(r0 + 32 * kBfRowBytesF4) = bf_expand_fp4(x1);
      } else {
        *reinterpret_cast<v4i*>(r0) = v4i{(int)s_lut[x0 & 0xff], (int)s_lut[(x0 >> 8) & 0xff], (int)s_lut[(x0 >> 16) & 0xff], (int)s_lut[x0 >> 24]};
        *reinterpret_cast<v4i*>(r0 + 32 * kBfRowBytesF4) = v4i{(int)s_lut[x1 & 0xff], (int)s_lut[(x1 >> 8) & 0xff], (int)s_lut[(x1 >> 16) & 0xff], (int)s_lut[x1 >> 24]};
      }
    }
    __syncthreads();
    if (!wave_has_queries) continue;
#pragma unroll 1
    for (int rt = 0; rt < 2; ++rt) {
      const int row0 = stage * kBfStageRows + rt * 32;  // first train row of the tile
      if (row0 >= nb) break;
      const uint8_t* frag = buf + (rt * 32 + col) * kBfRowBytesF4 + half * 16;
      v16f acc0 = c_init, acc1 = c_init;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const v4i a4 = *reinterpret_cast<const v4i*>(frag + s * 32);
        const v8i a = v8i{a4[0], a4[1], a4[2], a4[3], 0, 0, 0, 0};
        acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, bq[0][s], acc0, 4, 4, 0, kScale, 0, kScale);
        acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, bq[1][s], acc1, 4, 4, 0, kScale, 0, kScale);
      }
      best[0] -= 32.f; second[0] -= 32.f; best[1] -= 32.f; second[1] -= 32.f;
      ++tiles_in_sweep;
      if (row0 + 32 <= nb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { bf_track_rel_f(best[0], second[0], acc0[r]); bf_track_rel_f(best[1], second[1], acc1[r]); }
      } else {  // the last tile of the train set: rows beyond it never win
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool live = row0 + (r & 3) + 8 * (r >> 2) + 4 * half < nb;
          bf_track_rel_f(best[0], second[0], live ? acc0[r] : kBfIdleF);
          bf_track_rel_f(best[1], second[1], live ? acc1[r] : kBfIdleF);
        }
      }
      if (row0 + 32 - sweep_start == kBfSweep) { close_sweep(sweep_start); sweep_start = row0 + 32; }
    }
  }
  if (!wave_has_queries) return;
  if (tiles_in_sweep > 0) close_sweep(sweep_start);
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    // the two halves of the wave hold the same columns, interleaved groups of four rows
    const uint32_t ob = __shfl_xor(out_best[ct], 32), os = __shfl_xor(out_second[ct], 32);
    bf_merge(out_best[ct], out_second[ct], ob, os);
    const int qi = q_base + wave * 64 + ct * 32 + col;
    if (half == 0 && qi < na) {
      if (splits > 1) {
        uint32_t* o = partial + ((size_t)split * na + qi) * 2;
        o[0] = out_best[ct]; o[1] = out_second[ct];
      } else {
        const size_t o = (size_t)p * cap + qi;
        best_idx[o] = out_best[ct] == kNone ? -1 : (int32_t)(out_best[ct] & 0xffffu);
        best_dist[o] = (int32_t)(out_best[ct] >> 16);
        if (second_dist) second_dist[o] = (int32_t)(out_second[ct] >> 16);
      }
    }
  }
}

// folds the slices' packed (distance << 16 | index) best / second of every query (k_hamming_fp4 with splits > 1)
__global__ __launch_bounds__(256) void k_hamming_merge(const uint32_t* __restrict__ partial, int na, int splits, int32_t* __restrict__ best_idx,
                                                       int32_t* __restrict__ best_dist, int32_t* __restrict__ second_dist) {
  const int qi = blockIdx.x * 256 + threadIdx.x;
  if (qi >= na) return;
  const uint32_t kNone = (256u << 16) | 0xffffu;
  uint32_t b = kNone, s2 = kNone;
  for (int sp = 0; sp < splits; ++sp) {
    const uint32_t* o = partial + ((size_t)sp * na + qi) * 2;
    bf_merge(b, s2, o[0], o[1]);
  }
  best_idx[qi] = b == kNone ? -1 : (int32_t)(b & 0xffffu);
  best_dist[qi] = (int32_t)(b >> 16);
  if (second_dist) second_dist[qi] = (int32_t)(s2 >> 16);
}


// ------------------------------------------------------------------------------------------------
// MapPoint::ComputeDistinctiveDescriptors (/root/reference/src/MapPoint.cc:329-403) for a batch of map points: among the
// N descriptors that observe a point, the one with the least median Hamming distance to all N (its own 0 included; median =
// sorted row[(size_t)(0.5 * (N - 1))]; strict '<' over the rows, so the first minimum wins).
// One wave per point, one row per lane (rows in chunks of 64); the other descriptor of a pair is wave-uniform and comes
// through scalar loads.  The k-th smallest of a row is found by bisection on the value range 0..256 (9 counting passes that
// recompute the distances: N is a few dozen, nothing is stored).
// grid = points, block = 64
__global__ __launch_bounds__(64) void k_distinctive(const uint8_t* __restrict__ desc, const int32_t* __restrict__ off, int32_t* __restrict__ best) {
  const int p = blockIdx.x, lane = threadIdx.x;
  const int b = off[p], n = off[p + 1] - b;
  if (n <= 0) { if (lane == 0) best[p] = -1; return; }
  const unsigned long long* D = reinterpret_cast<const unsigned long long*>(desc) + 4 * (size_t)b;
  const int k = (n - 1) >> 1;  // (size_t)(0.5 * (N - 1))
  uint32_t wbest = 0xffffffffu;  // median << 16 | row
  for (int r0 = 0; r0 < n; r0 += 64) {
    const int i = r0 + lane;
    const bool live = i < n;
    const unsigned long long* Q = D + 4 * (size_t)(live ? i : 0);
    const unsigned long long q[4] = {Q[0], Q[1], Q[2], Q[3]};
    int lo = 0, hi = 256;
    for (int it = 0; it < 9; ++it) {  // 257 possible values
      const int mid = (lo + hi) >> 1;
      int c = 0;
      for (int j = 0; j < n; ++j) c += hamming256(q, D + 4 * (size_t)j) <= mid ? 1 : 0;
      if (c >= k + 1) hi = mid; else lo = mid + 1;
    }
    uint32_t key = live ? ((uint32_t)lo << 16) | (uint32_t)i : 0xffffffffu;
    for (int m = 32; m >= 1; m >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)key, m); key = o < key ? o : key; }
    wbest = key < wbest ? key : wbest;
  }
  if (lane == 0) best[p] = (int32_t)(wbest & 0xffffu);
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
// A first-image feature keeps the LAST second-image feature of the node's bucket with the smallest distance <= TH_LOW among those that
// pass the tests (":1075 if(dist>TH_LOW || dist>bestDist) continue" lets an equal distance replace the best) - the tests do not
// depend on what was found before, and nothing is handed from one first-image feature to the next (vbMatched2 is never set in this
// version, :1135): every pair is independent.  Round 6: a work-item per first-image feature walked the bucket through ~4 dependent
// global loads per candidate (38 us for 2000 x 2000 features in ~100 nodes, 20 work-items of a workgroup busy).  Now both buckets
// are staged in LDS by all 256 work-items at once (kTriCap entries each; longer buckets read the rest from global memory), a WAVE
// takes a first-image feature, its lanes the candidates, and one DPP minimum over (distance << 26 | 2^26 - 1 - position) names the match.
constexpr int kTriCap = 256;
struct TriEntry {
  unsigned long long d[4];
  float x, y;
  int32_t idx;
  int32_t oct_flags;  // octave | stereo << 8 | skip << 9 (holds a map point, or not stereo under only_stereo)
};
__device__ __forceinline__ TriEntry tri_load(const TriDev& T, bool second, int idx) {
  TriEntry e;
  const unsigned long long* D = reinterpret_cast<const unsigned long long*>((second ? T.desc2 : T.desc1) + (size_t)idx * 32);
  e.d[0] = D[0]; e.d[1] = D[1]; e.d[2] = D[2]; e.d[3] = D[3];
  const float* xy = second ? T.xy2 : T.xy1;
  e.x = xy[2 * idx]; e.y = xy[2 * idx + 1];
  e.idx = idx;
  const bool stereo = (second ? T.ur2 : T.ur1)[idx] >= 0;
  const bool skip = (second ? T.mp2 : T.mp1)[idx] != 0 || (T.only_stereo && !stereo);
  e.oct_flags = (second ? T.oct2[idx] : 0) | (stereo ? 0x100 : 0) | (skip ? 0x200 : 0);
  return e;
}
__global__ __launch_bounds__(256) void k_search_triangulation(TriDev T) {
  __shared__ TriEntry s_1[kTriCap], s_2[kTriCap];
  const int np = blockIdx.x, tid = threadIdx.x, lane = lane_id();
  const int a = T.pair_n1[np], b = T.pair_n2[np];
  const int b1 = T.off1[a], n1 = T.off1[a + 1] - b1, b2 = T.off2[b], n2 = T.off2[b + 1] - b2;
  if (tid < n1) s_1[tid] = tri_load(T, false, T.feat1[b1 + tid]);
  if (tid < n2) s_2[tid] = tri_load(T, true, T.feat2[b2 + tid]);
  __syncthreads();
  for (int p = wave_id(); p < n1; p += 4) {   // wave-uniform
    TriEntry e1;
    if (p < kTriCap) e1 = s_1[p]; else e1 = tri_load(T, false, T.feat1[b1 + p]);
    if (e1.oct_flags & 0x200) continue;
    const bool stereo1 = (e1.oct_flags & 0x100) != 0;
    // epipolar line of kp1 in image 2 (Pinhole.cpp:115-117), constant over the candidates
    const float la = e1.x * T.F[0] + e1.y * T.F[3] + T.F[6];
    const float lb = e1.x * T.F[1] + e1.y * T.F[4] + T.F[7];
    const float lc = e1.x * T.F[2] + e1.y * T.F[5] + T.F[8];
    const float den = la * la + lb * lb;
    uint32_t best = 0xffffffffu;
    for (int j = lane; j < n2; j += 64) {
      TriEntry e2;
      if (j < kTriCap) e2 = s_2[j]; else e2 = tri_load(T, true, T.feat2[b2 + j]);
      if (e2.oct_flags & 0x200) continue;
      const int dist = hamming256(e1.d, e2.d);
      if (dist > 50 /* TH_LOW */) continue;
      const int oct2 = e2.oct_flags & 0xff;
      if (!stereo1 && !(e2.oct_flags & 0x100)) {
        const float ex = T.ep[0] - e2.x, ey = T.ep[1] - e2.y;
        if (ex * ex + ey * ey < 100 * T.scale2[oct2]) continue;
      }
      bool ok = T.coarse != 0;
      if (!ok && den != 0) {
        const float num = la * e2.x + lb * e2.y + lc;
        const float dsqr = __fdiv_rn(num * num, den);
        ok = (double)dsqr < 3.84 * (double)T.sigma2[oct2];  // 3.84 is a double literal in the reference
      }
      if (!ok) continue;
      const uint32_t key = ((uint32_t)dist << 26) | (uint32_t)(0x3ffffff - j);   // dist <= 50: six bits
      best = key < best ? key : best;
    }
    best = wave_min_uniform(best);
    if (lane == 0 && best != 0xffffffffu) {
      const int j = 0x3ffffff - (int)(best & 0x3ffffffu);
      T.matches12[e1.idx] = j < kTriCap ? s_2[j].idx : T.feat2[b2 + j];
    }
  }
}


// ------------------------------------------------------------------------------------------------
// ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono)
// (/root/reference/src/ORBmatcher.cc:1676-1887, single camera) with Frame::AssignFeaturesToGrid,
// PosInGrid and GetFeaturesInArea (src/Frame.cc:475-506, 815-825, 747-813).
//
// The reference walks the LastFrame map points in index order, and a point with observations BLOCKS the feature it is
// assigned to for every later point: a sequential greedy assignment.  Three kernels:
//   k_proj_grid        one workgroup: AssignFeaturesToGrid as a counting sort
//   k_proj_candidates  work-item per map point: projection, search window, Hamming distance to every feature in the
//                      window; the VIABLE candidates (distance <= TH_HIGH - nothing else can ever be assigned) are kept
//                      as (distance, cell order, feature) keys - the minimum key is the reference's "first minimum"
//   k_proj_resolve     one workgroup resolves the greedy order in rounds: an unresolved point takes the smallest key
//                      whose feature is not held by a lower-index blocker; it is FINAL once no unresolved lower-index
//                      blocker still has that feature among its viable candidates (min_unres).  Finals commit, the
//                      rest try again.  The lowest unresolved index is final in every round, so the loop terminates,
//                      and induction over the index shows the result is the sequential one.
constexpr int kProjMaxLevels = 16;
constexpr int kProjCand = 16;  // viable candidates kept per point; a point with more re-scans its window (kRefOverflow)
constexpr uint32_t kRefOverflow = 1u << 31;
__host__ __device__ __forceinline__ int ref_n(uint32_t r) { return (int)((r >> 26) & 31u); }
__host__ __device__ __forceinline__ uint32_t ref_start(uint32_t r) { return r & 0x3ffffffu; }
// an entry of a point's candidate list as the resolve kernels read it
__host__ __device__ __forceinline__ uint32_t cand_entry(int dist, int level, int c) { return ((uint32_t)dist << 20) | ((uint32_t)level << 16) | (uint32_t)c; }
__host__ __device__ __forceinline__ int entry_dist(uint32_t e) { return (int)(e >> 20); }
__host__ __device__ __forceinline__ int entry_level(uint32_t e) { return (int)((e >> 16) & 15u); }
__host__ __device__ __forceinline__ int entry_feature(uint32_t e) { return (int)(e & 0xffffu); }
struct ProjDev {
  int n1, n2;
  const uint8_t* valid1;
  const float* wpos1;
  const uint8_t* mpdesc1;
  const uint8_t* obs1;
  const int32_t* oct1;
  const float* xy2;
  const int32_t* oct2;
  const float* ur2;
  const uint8_t* desc2;
  float grid[6], q[4], t[3], K[4], mbf, th, scale[kProjMaxLevels];
  int forward, backward;
  int skip_behind;           // 1: points with 1/z < 0 are skipped (Frame-to-Frame overload, :1709-1712); the key-frame overload has no such test
  int max_dist;              // a candidate is viable up to this Hamming distance (TH_HIGH / ORBdist)
  int sim3_mode;             // SearchByProjection(pKF, Scw, ...): wpos1 = camera-frame points, depth z < 0 skipped, KeyFrame::IsInImage,
                             // octaves predicted - 1 ... predicted; 1: Pinhole::project, 2: invz = 1 / z, fx (x invz) + cx
  // the vpMapPoints overload (ORBmatcher.cc:43-213) instead of wpos1 / oct1 / q / t / K:
  const float* proj1;        // mTrackProjX, mTrackProjY, mTrackProjXR
  const int32_t* level1;     // mnTrackScaleLevel
  const float* viewcos1;     // mTrackViewCos
  const uint8_t* blocked2;   // feature holds an observed map point on entry (nullptr: none)
  float nnratio;
  // scratch
  uint32_t* cell_start;      // kGridCells + 1
  uint16_t* cell_items;      // n2 feature indices grouped by grid cell
  int init_taken;            // 1: the grid comes from a resident frame, the candidate kernel initialises taken_by
  int32_t* taken_by;         // n2: index of the blocker holding the feature (INT_MAX = free)
  int32_t* min_unres;        // n2
  int32_t* owner;            // n2: SearchForInitialization's vnMatches21
  float4* win;               // n1: u, v, radius, ur (= u - mbf * invzc)
  int4* rng;                 // n1: cell x range, cell y range (lo | hi << 8), minLevel, maxLevel
  unsigned long long* cand;  // n1 keys: k_fuse_search's result (best key per point)
  uint32_t* clist;           // n1 slots of kProjCand 32-bit entries (cand_entry): the points' candidate lists
  uint32_t* cref;            // n1: where a point's list starts (i * kProjCand; the resolve kernels' LDS copy: where they packed it) | entries << 26 | kRefOverflow
  uint8_t* state;            // n1: 0 unresolved, 1 resolved, 2 final this round (k_init_resolve); | kStateObs
  int32_t* choice;           // n1: best CurrentFrame feature or -1
};
constexpr int kGridCols = 64, kGridRows = 48, kGridCells = kGridCols * kGridRows;  // FRAME_GRID_COLS / ROWS (Frame.h:46-47)

__host__ __device__ __forceinline__ void quat_rotate(const float q[4], float px, float py, float pz, float* rx, float* ry, float* rz) {
  // Eigen QuaternionBase::_transformVector: uv = 2 (q.vec x p); p + w uv + q.vec x uv
  float ux = q[1] * pz - q[2] * py, uy = q[2] * px - q[0] * pz, uz = q[0] * py - q[1] * px;
  ux = ux + ux; uy = uy + uy; uz = uz + uz;
  const float cx = q[1] * uz - q[2] * uy, cy = q[2] * ux - q[0] * uz, cz = q[0] * uy - q[1] * ux;
  *rx = px + q[3] * ux + cx; *ry = py + q[3] * uy + cy; *rz = pz + q[3] * uz + cz;
}

// the static tests of GetFeaturesInArea + the stereo check (ORBmatcher.cc:1749-1755) for feature c of a point's window
__device__ __forceinline__ bool window_accepts(const ProjDev& P, const float4& w, const int4& r, bool check_levels, int c) {
  if (check_levels) {
    const int o = P.oct2[c];
    if (o < r.z) return false;
    if (r.w >= 0 && o > r.w) return false;
  }
  const float dx = P.xy2[2 * c] - w.x, dy = P.xy2[2 * c + 1] - w.y;
  if (!(fabsf(dx) < w.z && fabsf(dy) < w.z)) return false;
  const float ur = P.ur2 ? P.ur2[c] : -1.f;  // no stereo coordinate test in the key-frame overload
  if (ur > 0 && fabsf(w.w - ur) > w.z) return false;
  return true;
}
// the same tests on values the caller has loaded (wave_candidates requests a cell's features four at a time)
__device__ __forceinline__ bool window_accepts_v(const float4& w, const int4& r, bool check_levels, int o, float x, float y, float ur) {
  if (check_levels) {
    if (o < r.z) return false;
    if (r.w >= 0 && o > r.w) return false;
  }
  const float dx = x - w.x, dy = y - w.y;
  if (!(fabsf(dx) < w.z && fabsf(dy) < w.z)) return false;
  if (ur > 0 && fabsf(w.w - ur) > w.z) return false;
  return true;
}
// visits the candidates of a point that pass those tests, in the reference's traversal order (one work-item walks the window)
template <class F>
__device__ __forceinline__ void for_candidates(const ProjDev& P, const float4& w, const int4& r, F&& f) {
  const int x0 = r.x & 0xff, x1 = r.x >> 8, y0 = r.y & 0xff, y1 = r.y >> 8;
  const bool check_levels = (r.z > 0) || (r.w >= 0);
  for (int ix = x0; ix <= x1; ++ix)
    for (int iy = y0; iy <= y1; ++iy) {
      const int cell = ix * kGridRows + iy;
      const uint32_t kb = P.cell_start[cell], ke = P.cell_start[cell + 1];
      for (uint32_t k = kb; k < ke; ++k) {
        const int c = P.cell_items[k];
        if (window_accepts(P, w, r, check_levels, c)) f(c, cell);
      }
    }
}
// The same walk by a whole WAVE (round 6: a work-item per point ran ~100 dependent gathers one after the other - 200 us for a
// frame's 2000 points): the window's cells go to the lanes in traversal order (cell t of the walk: ix = x0 + t / ny,
// iy = y0 + t % ny), 64 at a time, a lane takes its cell's features.  key_of(c, cell) is the feature's key, or ~0 when it is no
// candidate; the candidates' ranks in traversal order come from one DPP prefix sum per 64 cells and the first kProjCand keys are
// stored at their ranks: the list a single work-item would have written.  A lane keeps the first two keys of its cell in
// registers (a cell of the 64 x 48 grid holds 0.7 features of a KITTI frame on average), so the cell is walked once; a cell with
// more candidates is walked again.  Returns the number of candidates (wave-uniform).  All 64 lanes must call it.
template <class KeyOf>
__device__ __forceinline__ int wave_candidates(const ProjDev& P, const float4& w, const int4& r, unsigned long long* list, KeyOf&& key_of) {
  const int lane = lane_id();
  const int x0 = r.x & 0xff, x1 = r.x >> 8, y0 = r.y & 0xff, y1 = r.y >> 8, ny = y1 - y0 + 1, ncells = (x1 - x0 + 1) * ny;
  const bool check_levels = (r.z > 0) || (r.w >= 0);
  int base = 0;
  for (int t0 = 0; t0 < ncells; t0 += 64) {
    const int t = t0 + lane;
    uint32_t kb = 0, ke = 0;
    int cell = 0;
    if (t < ncells) {
      cell = (x0 + t / ny) * kGridRows + y0 + t % ny;
      kb = P.cell_start[cell]; ke = P.cell_start[cell + 1];
    }
    int cnt = 0;
    unsigned long long k0 = ~0ull, k1 = ~0ull;
    // four features of the cell at a time: their indices, then their positions / octaves / stereo coordinates, then the keys (the
    // descriptors) are requested back to back - device clock stamps: the walk was 12 000 of a wave's 17 000 cycles, three dependent
    // global reads per feature one feature after the other, and the lane with the fullest cell sets the wave's time
    for (uint32_t q0 = kb; q0 < ke; q0 += 4) {
      const int m = (int)(ke - q0);
      const int c0 = P.cell_items[q0], c1 = m > 1 ? (int)P.cell_items[q0 + 1] : c0, c2 = m > 2 ? (int)P.cell_items[q0 + 2] : c0,
                c3 = m > 3 ? (int)P.cell_items[q0 + 3] : c0;
      const int cs[4] = {c0, c1, c2, c3};
      float fx[4], fy[4], fu[4];
      int fo[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        fx[j] = P.xy2[2 * cs[j]]; fy[j] = P.xy2[2 * cs[j] + 1];
        fo[j] = check_levels ? P.oct2[cs[j]] : 0;
        fu[j] = P.ur2 ? P.ur2[cs[j]] : -1.f;
      }
      unsigned long long keys[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        keys[j] = (j < m && window_accepts_v(w, r, check_levels, fo[j], fx[j], fy[j], fu[j])) ? key_of(cs[j], cell) : ~0ull;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (keys[j] != ~0ull) {
          if (cnt == 0) k0 = keys[j]; else if (cnt == 1) k1 = keys[j];
          ++cnt;
        }
    }
    const int incl = wave_inclusive_scan(cnt);
    int pos = base + incl - cnt;
    base += __shfl(incl, 63);
    if (cnt > 0 && pos < kProjCand) {
      if (cnt <= 2) {
        list[pos] = k0;
        if (cnt == 2 && pos + 1 < kProjCand) list[pos + 1] = k1;
      } else {
        for (uint32_t k = kb; k < ke && pos < kProjCand; ++k) {
          const int c = P.cell_items[k];
          const unsigned long long key = window_accepts(P, w, r, check_levels, c) ? key_of(c, cell) : ~0ull;
          if (key != ~0ull) list[pos++] = key;
        }
      }
    }
  }
  return base;
}
__device__ __forceinline__ unsigned long long proj_key(int dist, int cell, int c) {
  // strict '<' over the traversal (cells x-major, feature index inside a cell) == minimum of this key
  return ((unsigned long long)dist << 32) | ((unsigned long long)cell << 16) | (unsigned)c;
}

// The wave that walked a point's window hands its list to the resolve kernel (round 6: the resolve rounds were chains of dependent
// global reads of 8-byte keys - ~8 us per round for a frame's 2000 points).  The keys the walk left in the wave's LDS slots
// (traversal order) become 32-bit entries in the point's 16-entry slot of P.clist, 64 bytes the resolve kernel fetches with four
// independent 16-byte loads per point and packs into LDS: its rounds never touch global memory.  (A dense list reserved with one
// atomic per wave was tried first: ~3000 returning atomics on one address cost the candidate kernels 10 - 20 us.)
// `sorted`: the entries are written in ascending key order (best-only searches: a point's choice is then the first entry still
// available).  All 64 lanes call it; returns the point's cref word (wave-uniform).
template <class Enc>
__device__ __forceinline__ uint32_t publish_candidates(const ProjDev& P, int i, const unsigned long long* keys, int total, bool sorted, Enc&& enc) {
  wave_sync();
  const int lane = lane_id();
  if (total > kProjCand) return kRefOverflow;
  const int n = total;
  const unsigned long long key = lane < n ? keys[lane] : ~0ull;
  int rank = lane;
  if (sorted) {
    rank = 0;
    for (int j = 0; j < n; ++j) rank += keys[j] < key ? 1 : 0;
  }
  const uint32_t start = (uint32_t)i * kProjCand;
  if (lane < n) P.clist[start + rank] = enc(key);
  return start | ((uint32_t)n << 26);
}
// what the candidate kernels leave in P.state: 0 = unresolved, 1 = resolved | kStateObs: the point blocks the feature it takes
constexpr uint8_t kStateObs = 0x80;

// grid = 1, block = kGridBS.  LDS = true (frames of up to kGridLdsN2 features): a feature's cell is computed once and kept in
// LDS, the cells' item lists are built and put in ascending order there and leave with one coalesced pass - the first form ran
// count, scatter and the per-cell ordering as chains of dependent global reads (20 us for a frame's 2000 features, round 6).
constexpr int kGridBS = 1024, kGridLdsN2 = 8192;
template <bool LDS>
__global__ __launch_bounds__(kGridBS) void k_proj_grid(ProjDev P) {
  __shared__ uint32_t s_fill[kGridCells], s_start[LDS ? kGridCells : 1];
  __shared__ uint32_t s_scan[kGridBS / 64 + 1];
  __shared__ uint16_t s_cell[LDS ? kGridLdsN2 : 1], s_items[LDS ? kGridLdsN2 : 1];
  const int tid = threadIdx.x;
  auto cell_of = [&](int c) {
    const int px = (int)roundf((P.xy2[2 * c] - P.grid[0]) * P.grid[4]), py = (int)roundf((P.xy2[2 * c + 1] - P.grid[1]) * P.grid[5]);
    return (px >= 0 && px < kGridCols && py >= 0 && py < kGridRows) ? px * kGridRows + py : -1;
  };
  for (int c = tid; c < kGridCells; c += kGridBS) s_fill[c] = 0;
  __syncthreads();
  for (int c = tid; c < P.n2; c += kGridBS) {
    const int cell = cell_of(c);
    if (LDS) s_cell[c] = (uint16_t)cell;   // 0xffff: outside the grid
    if (cell >= 0) atomicAdd(&s_fill[cell], 1u);
    P.taken_by[c] = (P.blocked2 && P.blocked2[c]) ? -1 : INT_MAX;  // -1: held by a point from before the call
  }
  __syncthreads();
  // a work-item owns three consecutive cells: ONE workgroup prefix sum for the 3072 cells (three chunks of 1024 took three)
  static_assert(kGridCells == 3 * kGridBS, "three cells per work-item");
  uint32_t carry;
  {
    const uint32_t v0 = s_fill[3 * tid], v1 = s_fill[3 * tid + 1], v2 = s_fill[3 * tid + 2];
    const uint32_t ex = block_exclusive_scan<uint32_t>(v0 + v1 + v2, s_scan, &carry);
    const uint32_t st[3] = {ex, ex + v0, ex + v0 + v1};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      P.cell_start[3 * tid + q] = st[q]; s_fill[3 * tid + q] = st[q];
      if (LDS) s_start[3 * tid + q] = st[q];
    }
  }
  if (tid == 0) P.cell_start[kGridCells] = carry;
  __syncthreads();
  uint16_t* items = LDS ? s_items : P.cell_items;
  for (int c = tid; c < P.n2; c += kGridBS) {
    const int cell = LDS ? (s_cell[c] == 0xffffu ? -1 : (int)s_cell[c]) : cell_of(c);
    if (cell >= 0) items[atomicAdd(&s_fill[cell], 1u)] = (uint16_t)c;
  }
  __syncthreads();
  // ascending feature index inside every cell = the push_back order of AssignFeaturesToGrid (cells hold a handful of items)
  for (int cell = tid; cell < kGridCells; cell += kGridBS) {
    const uint32_t b = LDS ? s_start[cell] : P.cell_start[cell], e = s_fill[cell];
    for (uint32_t i = b + 1; i < e; ++i) {
      const uint16_t v = items[i];
      uint32_t j = i;
      while (j > b && items[j - 1] > v) { items[j] = items[j - 1]; --j; }
      items[j] = v;
    }
  }
  if (LDS) {
    __syncthreads();
    for (uint32_t k = tid; k < carry; k += kGridBS) P.cell_items[k] = s_items[k];
  }
}
static void launch_proj_grid(const ProjDev& P, hipStream_t s) {
  if (P.n2 <= kGridLdsN2) hipLaunchKernelGGL(k_proj_grid<true>, dim3(1), dim3(kGridBS), 0, s, P);
  else hipLaunchKernelGGL(k_proj_grid<false>, dim3(1), dim3(kGridBS), 0, s, P);
}

// With the grid taken from a resident frame nobody runs k_proj_grid: its other job - who holds a feature before the call - is
// done by the candidate kernels' workgroups on the way (the resolve kernel behind them is the first reader).
__device__ __forceinline__ void init_taken_by(const ProjDev& P) {
  if (!P.init_taken) return;
  for (int c = (int)(blockIdx.x * blockDim.x + threadIdx.x); c < P.n2; c += (int)(gridDim.x * blockDim.x))
    P.taken_by[c] = (P.blocked2 && P.blocked2[c]) ? -1 : INT_MAX;
}

// grid = ceil(n1 / 4), block = 256 = one wave per LastFrame map point: projection and search window (ORBmatcher.cc:1696-1735,
// computed by every lane alike), then its viable candidates by wave_candidates
constexpr int kPointsPerBlock = 4;
__global__ __launch_bounds__(256) void k_proj_candidates(ProjDev P) {
  __shared__ unsigned long long s_keys[kPointsPerBlock][kProjCand];
  init_taken_by(P);
  const int i = blockIdx.x * kPointsPerBlock + wave_id();
  if (i >= P.n1) return;
  const int lane = lane_id();
  uint8_t st = 1;
  uint32_t ref = 0;
  // everything the point needs is requested up front (it was four dependent global reads: valid -> position -> octave -> descriptor)
  const uint8_t valid = P.valid1[i];
  const float wx = P.wpos1[3 * i], wy = P.wpos1[3 * i + 1], wz = P.wpos1[3 * i + 2];
  const int oct_i = P.oct1[i];
  const unsigned long long* D = reinterpret_cast<const unsigned long long*>(P.mpdesc1 + (size_t)i * 32);
  const unsigned long long d[4] = {D[0], D[1], D[2], D[3]};
  if (valid) {
    float x, y, z;
    if (P.sim3_mode) {
      x = wx; y = wy; z = wz;
    } else {
      quat_rotate(P.q, wx, wy, wz, &x, &y, &z);
      x += P.t[0]; y += P.t[1]; z += P.t[2];
    }
    const float invzc = (float)(1.0 / (double)z);
    float u, v;
    if (P.sim3_mode == 2) {
      const float invz = __fdiv_rn(1.0f, z);
      u = P.K[0] * (x * invz) + P.K[2]; v = P.K[1] * (y * invz) + P.K[3];
    } else {
      u = __fdiv_rn(P.K[0] * x, z) + P.K[2]; v = __fdiv_rn(P.K[1] * y, z) + P.K[3];
    }
    // NaN / inf coordinates never produce candidates in the reference either (empty cell range)
    const bool in_view = P.sim3_mode ? (!(z < 0.0f) && u >= P.grid[0] && u < P.grid[2] && v >= P.grid[1] && v < P.grid[3])
                                     : (!(P.skip_behind && invzc < 0) && u == u && v == v && !(u < P.grid[0] || u > P.grid[2]) &&
                                        !(v < P.grid[1] || v > P.grid[3]));
    if (in_view) {
      const int oct = oct_i;
      const float radius = P.th * P.scale[oct];
      int min_level, max_level;
      if (P.sim3_mode) { min_level = oct - 1; max_level = oct; }
      else if (P.forward) { min_level = oct; max_level = -1; }
      else if (P.backward) { min_level = 0; max_level = oct; }
      else { min_level = oct - 1; max_level = oct + 1; }
      const int x0 = imax(0, (int)floorf((u - P.grid[0] - radius) * P.grid[4]));
      const int x1 = imin(kGridCols - 1, (int)ceilf((u - P.grid[0] + radius) * P.grid[4]));
      const int y0 = imax(0, (int)floorf((v - P.grid[1] - radius) * P.grid[5]));
      const int y1 = imin(kGridRows - 1, (int)ceilf((v - P.grid[1] + radius) * P.grid[5]));
      if (x0 < kGridCols && x1 >= 0 && y0 < kGridRows && y1 >= 0) {
        const float4 w = make_float4(u, v, radius, u - P.mbf * invzc);
        const int4 r = make_int4(x0 | (x1 << 8), y0 | (y1 << 8), min_level, max_level);
        auto dist_of = [&](int c) { return hamming256(d, reinterpret_cast<const unsigned long long*>(P.desc2 + (size_t)c * 32)); };
        const int total = wave_candidates(P, w, r, s_keys[wave_id()], [&](int c, int cell) {
          const int dist = dist_of(c);
          return dist <= P.max_dist ? proj_key(dist, cell, c) : ~0ull;
        });
        if (lane == 0) { P.win[i] = w; P.rng[i] = r; }
        ref = publish_candidates(P, i, s_keys[wave_id()], total, true,
                                 [](unsigned long long key) { return cand_entry((int)(key >> 32), 0, (int)(key & 0xffffu)); });
        st = total > 0 ? 0 : 1;  // no viable candidate: bestDist > TH_HIGH whatever the others do
      }
    }
  }
  if (lane == 0) { P.choice[i] = -1; P.cref[i] = ref; P.state[i] = st | (P.obs1[i] ? kStateObs : 0); }
}

// the feature point i takes among those not held by a lower-index blocker: the smallest key (-1 = none)
__device__ __forceinline__ int proj_best(const ProjDev& P, const int32_t* taken_by, const uint32_t* cref, const uint32_t* clist, int i) {
  const uint32_t r = cref[i];
  if (!(r & kRefOverflow)) {
    const uint32_t* e = clist + ref_start(r);
    for (int k = 0, n = ref_n(r); k < n; ++k) {   // ascending keys
      const int c = entry_feature(e[k]);
      if (taken_by[c] >= i) return c;
    }
    return -1;
  }
  unsigned long long best = ~0ull;
  const unsigned long long* D = reinterpret_cast<const unsigned long long*>(P.mpdesc1 + (size_t)i * 32);
  const unsigned long long d[4] = {D[0], D[1], D[2], D[3]};
  for_candidates(P, P.win[i], P.rng[i], [&](int c, int cell) {
    if (taken_by[c] < i) return;
    const int dist = hamming256(d, reinterpret_cast<const unsigned long long*>(P.desc2 + (size_t)c * 32));
    if (dist > P.max_dist) return;
    const unsigned long long key = proj_key(dist, cell, c);
    if (key < best) best = key;
  });
  return best == ~0ull ? -1 : (int)(best & 0xffffu);
}

// The resolve kernels' view of the lists: packed into LDS when the call's points (cref) and entries fit, else where the candidate kernels left them
constexpr int kResolveLdsN2 = 6144, kResolveLdsN1 = 8192, kResolveLdsList = 12288;
constexpr int kUCap = 4096;   // unresolved points a round can hand to the next one as a list (resolve_rounds)
constexpr int kResolveBS = 1024;
struct ListView { const uint32_t* cref; const uint32_t* clist; };
// (the copies of taken_by and state ride in the same trips: three independent loads per trip instead of three loops of dependent ones)
template <bool LDS>
__device__ __forceinline__ ListView stage_lists(const ProjDev& P, int32_t* s_taken, uint8_t* s_state, uint32_t* s_ref, uint32_t* s_list, int* s_total) {
  ListView v{P.cref, P.clist};
  if (LDS) {
    const int tid = threadIdx.x, lane = lane_id();
    if (tid == 0) *s_total = 0;
    __syncthreads();
    // where a point's entries go: any packing will do, so a wave's points take one range (one LDS atomic per wave and trip)
    for (int i0 = 0; i0 < P.n1 || i0 < P.n2; i0 += kResolveBS) {
      const int i = i0 + tid;
      const uint32_t r = i < P.n1 ? P.cref[i] : 0u;
      const uint8_t st = i < P.n1 ? P.state[i] : (uint8_t)0;
      const int32_t tk = i < P.n2 ? P.taken_by[i] : 0;
      if (i < P.n1) s_state[i] = st;
      if (i < P.n2) s_taken[i] = tk;
      const int n = (r & kRefOverflow) ? 0 : ref_n(r);
      const int incl = wave_inclusive_scan(n), tot = __shfl(incl, 63);
      int base = 0;
      if (lane == 0 && tot > 0) base = atomicAdd(s_total, tot);
      base = __shfl(base, 0);
      if (i < P.n1) s_ref[i] = (r & kRefOverflow) | ((uint32_t)n << 26) | (uint32_t)(base + incl - n);
    }
    __syncthreads();
    if (*s_total <= kResolveLdsList) {
      for (int i = tid; i < P.n1; i += kResolveBS) {
        const uint32_t r = s_ref[i];
        const int n = ref_n(r);
        if (n == 0) continue;
        const uint4* src = reinterpret_cast<const uint4*>(P.clist + (size_t)i * kProjCand);
        const uint4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
        const uint32_t e[kProjCand] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
        uint32_t* dst = s_list + ref_start(r);
#pragma unroll
        for (int k = 0; k < kProjCand; ++k)
          if (k < n) dst[k] = e[k];
      }
      v.clist = s_list;
    } else {
      for (int i = tid; i < P.n1; i += kResolveBS) s_ref[i] = P.cref[i];
    }
    v.cref = s_ref;
  }
  return v;
}

// grid = 1, block = kResolveBS.  LDS = true (frames of up to kResolveLdsN2 features, kResolveLdsN1 points: every KITTI-size call):
// who holds a feature, the lowest unresolved blocker per feature and the points' states live in LDS for the whole kernel - a
// round is three passes over them, and with the arrays in global memory every pass was a chain of dependent ~1 us loads
// (round 6: 8 rounds of 17 us for a frame's 2000 points).
// min_unres[c] = (stamp << 20 | lowest unresolved blocker that may still take c), stamp = kStampMax - round: a round's entries are
// smaller than everything older rounds left behind, so the array is not cleared between rounds (one pass and one barrier per
// round less); it is refilled with INT_MAX every kStampMax rounds (the worst case - a chain of n1 blockers - takes n1 rounds).
constexpr int kStampMax = 2047;
__device__ __forceinline__ bool lower_unresolved(int m, int stamp, int i) { return (m >> 20) == stamp && (m & 0xfffff) < i; }
// The round loop of the three resolve kernels.  announce(i, stamp): point i, if it is an unresolved blocker, publishes itself on what it
// may still take; decide(i, stamp): point i becomes final (and commits) or not - true while it stays unresolved.  Rounds: all points;
// from the second round on the LIST of the points the round before left unresolved (up to kUCap of them) - a wave runs a point's chain
// of dependent LDS accesses as soon as ONE of its lanes holds an unresolved point, and the few dozen points of the late rounds,
// scattered over all waves, cost every wave its full time; and once at most 64 are left, ONE wave finishes them with wave-level
// synchronisation only: what is left then is a chain of points waiting for each other, one short round per link (a KITTI frame: ~10
// rounds, 3.5 us each while all 16 waves and three workgroup barriers took part in every one).
template <bool LDS, class Announce, class Decide>
__device__ __forceinline__ void resolve_rounds(int n1, int n2, int32_t* min_unres, int* s_unres, uint16_t (*s_ulist)[LDS ? kUCap : 1],
                                               Announce&& announce, Decide&& decide) {
  const int tid = threadIdx.x;
  if (tid < 2) s_unres[tid] = 0;
  bool use_list = false;
  int n_walk = n1;   // points this round looks at: all, or the entries of s_ulist[b ^ 1]
  int round = 0, left = 0;
  for (; round <= n1; ++round) {
    const int stamp = kStampMax - round % kStampMax, b = round & 1;
    if (round % kStampMax == 0) {
      __syncthreads();
      for (int c = tid; c < n2; c += kResolveBS) min_unres[c] = INT_MAX;
    }
    __syncthreads();
    for (int t = tid; t < n_walk; t += kResolveBS) announce(use_list ? (int)s_ulist[b ^ 1][t] : t, stamp);
    __syncthreads();
    for (int t = tid; t < n_walk; t += kResolveBS) {
      const int i = use_list ? (int)s_ulist[b ^ 1][t] : t;
      if (decide(i, stamp)) {   // counted, and listed for the next round
        const int pos = atomicAdd(&s_unres[b], 1);
        if (LDS && pos < kUCap) s_ulist[b][pos] = (uint16_t)i;
      }
    }
    if (tid == 0) s_unres[b ^ 1] = 0;
    __syncthreads();
    left = s_unres[b];
    if (left == 0) return;
    if (LDS && left <= 64) break;
    use_list = LDS && left <= kUCap;
    n_walk = use_list ? left : n1;
  }
  if (!LDS || wave_id() != 0) return;
  const int lane = lane_id();
  int mine = lane < left ? (int)s_ulist[round & 1][lane] : -1;
  for (++round; round <= n1 + 1; ++round) {
    const int stamp = kStampMax - round % kStampMax;
    if (round % kStampMax == 0) {
      wave_sync();
      for (int c = lane; c < n2; c += 64) min_unres[c] = INT_MAX;
    }
    wave_sync();
    if (mine >= 0) announce(mine, stamp);
    wave_sync();
    if (mine >= 0 && !decide(mine, stamp)) mine = -1;
    wave_sync();
    if (__ballot(mine >= 0) == 0ull) break;
  }
}

template <bool LDS>
__global__ __launch_bounds__(kResolveBS) void k_proj_resolve(ProjDev P) {
  __shared__ int s_unres[2], s_total;
  __shared__ int32_t s_taken[LDS ? kResolveLdsN2 : 1], s_min[LDS ? kResolveLdsN2 : 1];
  __shared__ uint8_t s_state[LDS ? kResolveLdsN1 : 1];
  int32_t* taken_by = LDS ? s_taken : P.taken_by;
  int32_t* min_unres = LDS ? s_min : P.min_unres;
  uint8_t* state = LDS ? s_state : P.state;
  __shared__ uint32_t s_ref[LDS ? kResolveLdsN1 : 1], s_list[LDS ? kResolveLdsList : 1];
  __shared__ uint16_t s_ulist[2][LDS ? kUCap : 1];
  const ListView staged = stage_lists<LDS>(P, s_taken, s_state, s_ref, s_list, &s_total);
  // Instantiated twice, for the lists in LDS and for the lists where the candidate kernel left them: with ONE pointer chosen at run
  // time the entries were read with flat loads
  auto run = [&](const ListView L) {
    // every unresolved blocker announces itself on the features it may still take
    auto announce = [&](int i, int stamp) {
      if (state[i] != kStateObs) return;   // unresolved and a blocker
      const uint32_t r = L.cref[i];
      const int me = (stamp << 20) | i;
      if (!(r & kRefOverflow)) {
        const uint32_t* e = L.clist + ref_start(r);
        for (int k = 0, n = ref_n(r); k < n; ++k) {
          const int c = entry_feature(e[k]);
          if (taken_by[c] >= i) atomicMin(&min_unres[c], me);
        }
      } else {
        const unsigned long long* D = reinterpret_cast<const unsigned long long*>(P.mpdesc1 + (size_t)i * 32);
        const unsigned long long d[4] = {D[0], D[1], D[2], D[3]};
        for_candidates(P, P.win[i], P.rng[i], [&](int c, int) {
          if (taken_by[c] < i) return;
          if (hamming256(d, reinterpret_cast<const unsigned long long*>(P.desc2 + (size_t)c * 32)) <= P.max_dist) atomicMin(&min_unres[c], me);
        });
      }
    };
    // decide AND commit in one pass: a point whose best available feature no lower unresolved blocker can still take is final
    // and occupies it at once.  What a concurrent work-item sees of that store does not matter: a HIGHER point that misses it
    // picks the same feature, finds this point's announcement in front of it and waits a round; one that sees it takes its next
    // choice, which is what the sequential loop would have given it; LOWER points never want a feature that becomes final here
    // (their announcement would have held this point back).  Two blockers never become final on one feature in the same round.
    auto decide = [&](int i, int stamp) {
      const uint8_t st = state[i];
      if (st & 0x7f) return false;
      const int c = proj_best(P, taken_by, L.cref, L.clist, i);
      if (c < 0) { state[i] = st | 1; P.choice[i] = -1; return false; }   // everything viable is taken: no match
      if (lower_unresolved(min_unres[c], stamp, i)) return true;           // somebody in front of i can still take c
      state[i] = st | 1; P.choice[i] = c;
      if (st & kStateObs) taken_by[c] = i;
      return false;
    };
    resolve_rounds<LDS>(P.n1, P.n2, min_unres, s_unres, s_ulist, announce, decide);
  };
  if (LDS && staged.clist != P.clist) run(ListView{s_ref, s_list});
  else run(ListView{LDS ? s_ref : P.cref, P.clist});
}

// ------------------------------------------------------------------------------------------------
// The search of ORBmatcher::Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, th, bRight = false)
// (/root/reference/src/ORBmatcher.cc:1148-1338, single camera) with KeyFrame::GetFeaturesInArea / IsInImage
// (src/KeyFrame.cc:704-753): every map point looks for its best feature on its own - what the loop does with a match
// (Replace / AddObservation / AddMapPoint) changes MapPoint and KeyFrame objects and stays with the caller.
// Uses ProjDev: wpos1, mpdesc1, oct1 (= predicted level), valid1, the grid of k_proj_grid, q / t / K, mbf, th, scale;
// cand[i] receives the best (distance, cell order, feature) key or ~0.
struct FuseDev {
  float inv_sigma2[kProjMaxLevels];
  int cam_frame;  // 1: wpos1 already holds camera-frame coordinates (the caller applied its SE3 / Sim3)
  int proj_form;  // 0: Pinhole::project, fx x / z + cx; 1: invz = (float)(1.0 / z), fx (x invz) + cx (SearchBySim3, :1508-1513)
  int chi2_gate;  // 1: the reprojection-error gates of Fuse(pKF, vpMapPoints, th)
};

// grid = ceil(n1 / 4), block = 256 = one wave per map point (round 6; a work-item per point walked ~100 dependent gathers: 30 us
// for 2500 points): the window's cells go to the lanes in traversal order, a lane keeps the smallest (distance, cell, feature)
// key of its cells - the cell number grows along the traversal, so the smallest key IS the reference's first minimum - and the
// wave's minimum over (distance, cell) names the one lane that holds it.
__global__ __launch_bounds__(256) void k_fuse_search(ProjDev P, FuseDev Fz) {
  const int i = blockIdx.x * kPointsPerBlock + wave_id();
  if (i >= P.n1) return;
  const int lane = lane_id();
  unsigned long long best = ~0ull;
  if (P.valid1[i]) {
    float x, y, z;
    if (Fz.cam_frame) {
      x = P.wpos1[3 * i]; y = P.wpos1[3 * i + 1]; z = P.wpos1[3 * i + 2];
    } else {
      quat_rotate(P.q, P.wpos1[3 * i], P.wpos1[3 * i + 1], P.wpos1[3 * i + 2], &x, &y, &z);
      x += P.t[0]; y += P.t[1]; z += P.t[2];
    }
    if (!(z < 0.0f)) {  // depth must be positive (:1201-1206)
      const float invz = Fz.proj_form == 1 ? (float)(1.0 / (double)z) : __fdiv_rn(1.0f, z);
      float u, v;
      if (Fz.proj_form == 1) { u = P.K[0] * (x * invz) + P.K[2]; v = P.K[1] * (y * invz) + P.K[3]; }
      else { u = __fdiv_rn(P.K[0] * x, z) + P.K[2]; v = __fdiv_rn(P.K[1] * y, z) + P.K[3]; }
      if (u >= P.grid[0] && u < P.grid[2] && v >= P.grid[1] && v < P.grid[3]) {  // KeyFrame::IsInImage
        const float ur = u - P.mbf * invz;
        const int level = P.oct1[i];
        const float radius = P.th * P.scale[level];
        const int x0 = imax(0, (int)floorf((u - P.grid[0] - radius) * P.grid[4]));
        const int x1 = imin(kGridCols - 1, (int)ceilf((u - P.grid[0] + radius) * P.grid[4]));
        const int y0 = imax(0, (int)floorf((v - P.grid[1] - radius) * P.grid[5]));
        const int y1 = imin(kGridRows - 1, (int)ceilf((v - P.grid[1] + radius) * P.grid[5]));
        if (x0 < kGridCols && x1 >= 0 && y0 < kGridRows && y1 >= 0) {
          const unsigned long long* D = reinterpret_cast<const unsigned long long*>(P.mpdesc1 + (size_t)i * 32);
          const unsigned long long d[4] = {D[0], D[1], D[2], D[3]};
          const int ny = y1 - y0 + 1, ncells = (x1 - x0 + 1) * ny;
          for (int t = lane; t < ncells; t += 64) {
            const int cell = (x0 + t / ny) * kGridRows + y0 + t % ny;
            const uint32_t kb = P.cell_start[cell], ke = P.cell_start[cell + 1];
            for (uint32_t k = kb; k < ke; ++k) {
              const int c = P.cell_items[k];
              const float kpx = P.xy2[2 * c], kpy = P.xy2[2 * c + 1];
              if (!(fabsf(kpx - u) < radius && fabsf(kpy - v) < radius)) continue;
              const int kl = P.oct2[c];
              if (kl < level - 1 || kl > level) continue;
              // reprojection error gate, chi-square at 95 % with 3 / 2 degrees of freedom (:1262-1288)
              const float ex = u - kpx, ey = v - kpy;
              const float kpr = P.ur2 ? P.ur2[c] : -1.f;
              if (!Fz.chi2_gate) {
              } else if (kpr >= 0) {
                const float er = ur - kpr;
                const float e2 = ex * ex + ey * ey + er * er;
                if ((double)(e2 * Fz.inv_sigma2[kl]) > 7.8) continue;
              } else {
                const float e2 = ex * ex + ey * ey;
                if ((double)(e2 * Fz.inv_sigma2[kl]) > 5.99) continue;
              }
              const int dist = hamming256(d, reinterpret_cast<const unsigned long long*>(P.desc2 + (size_t)c * 32));
              const unsigned long long key = proj_key(dist, cell, c);
              if (key < best) best = key;  // strict '<' over the traversal order
            }
          }
        }
      }
    }
  }
  // (distance << 16 | cell) of the wave's smallest key: a cell belongs to one lane, so exactly one lane matches - or all do, with ~0
  const uint32_t m = wave_min_uniform((uint32_t)(best >> 16));
  if ((uint32_t)(best >> 16) == m && (best != ~0ull || lane == 0)) P.cand[i] = best;
}

// ------------------------------------------------------------------------------------------------
// ORBmatcher::SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, th, ...) (ORBmatcher.cc:43-213; single camera),
// the matcher of Tracking::SearchLocalPoints.  Same greedy order and blocking rule as above, but a point's decision is
// best AND second-best distance with their pyramid levels (ratio test only inside one level), so the candidates are
// kept in traversal order (distance << 32 | level << 16 | feature) and the sequential scan is replayed on the available
// ones; a point is final once no unresolved lower-index blocker can still take ANY of its available candidates.
struct LocalScan {
  int best_dist = 256, best_level = -1, best_dist2 = 256, best_level2 = -1, best_idx = -1;
  __device__ __forceinline__ void visit(int dist, int level, int c) {  // ORBmatcher.cc:104-122
    if (dist < best_dist) { best_dist2 = best_dist; best_dist = dist; best_level2 = best_level; best_level = level; best_idx = c; }
    else if (dist < best_dist2) { best_level2 = level; best_dist2 = dist; }
  }
  __device__ __forceinline__ int accept(float nnratio) const {  // ORBmatcher.cc:125-141; feature index or -1
    if (best_dist > 100 /* TH_HIGH */) return -1;
    if (best_level == best_level2 && (float)best_dist > nnratio * (float)best_dist2) return -1;
    return best_idx;
  }
};

// grid = ceil(n1 / 4), block = 256 = one wave per map point
__global__ __launch_bounds__(256) void k_local_candidates(ProjDev P) {
  __shared__ unsigned long long s_keys[kPointsPerBlock][kProjCand];
  init_taken_by(P);
  const int i = blockIdx.x * kPointsPerBlock + wave_id();
  if (i >= P.n1) return;
  const int lane = lane_id();
  uint8_t st = 1;
  uint32_t ref = 0;
  // (requested up front, as in k_proj_candidates)
  const uint8_t valid = P.valid1[i];
  const int level_i = P.level1[i];
  const float viewcos = P.viewcos1[i], px = P.proj1[3 * i], py = P.proj1[3 * i + 1], pxr = P.proj1[3 * i + 2];
  const unsigned long long* D = reinterpret_cast<const unsigned long long*>(P.mpdesc1 + (size_t)i * 32);
  const unsigned long long d[4] = {D[0], D[1], D[2], D[3]};
  if (valid) {
    const int level = level_i;
    float r = (double)viewcos > 0.998 ? 2.5f : 4.0f;  // RadiusByViewingCos: the literal is a double in the reference
    if (P.th != 1.0f) r *= P.th;
    const float x = px, y = py;
    const float radius = r * P.scale[level];
    if (x == x && y == y) {
      const int x0 = imax(0, (int)floorf((x - P.grid[0] - radius) * P.grid[4]));
      const int x1 = imin(kGridCols - 1, (int)ceilf((x - P.grid[0] + radius) * P.grid[4]));
      const int y0 = imax(0, (int)floorf((y - P.grid[1] - radius) * P.grid[5]));
      const int y1 = imin(kGridRows - 1, (int)ceilf((y - P.grid[1] + radius) * P.grid[5]));
      if (x0 < kGridCols && x1 >= 0 && y0 < kGridRows && y1 >= 0) {
        const float4 w = make_float4(x, y, radius, pxr);
        const int4 rg = make_int4(x0 | (x1 << 8), y0 | (y1 << 8), level - 1, level);
        const int total = wave_candidates(P, w, rg, s_keys[wave_id()], [&](int c, int) {
          const int dist = hamming256(d, reinterpret_cast<const unsigned long long*>(P.desc2 + (size_t)c * 32));
          return ((unsigned long long)dist << 32) | ((unsigned long long)P.oct2[c] << 16) | (unsigned)c;
        });
        if (lane == 0) { P.win[i] = w; P.rng[i] = rg; }
        ref = publish_candidates(P, i, s_keys[wave_id()], total, false, [](unsigned long long key) {   // traversal order: the scan is replayed
          return cand_entry((int)(key >> 32), (int)((key >> 16) & 0xffffu), (int)(key & 0xffffu));
        });
        st = total > 0 ? 0 : 1;
      }
    }
  }
  if (lane == 0) { P.choice[i] = -1; P.cref[i] = ref; P.state[i] = st | (P.obs1[i] ? kStateObs : 0); }
}

// visits the candidates of point i that no lower-index blocker holds, in traversal order: f(dist, level, c)
template <class F>
__device__ __forceinline__ void local_available(const ProjDev& P, const int32_t* taken_by, const ListView& L, int i, F&& f) {
  const uint32_t r = L.cref[i];
  if (!(r & kRefOverflow)) {
    const uint32_t* e = L.clist + ref_start(r);
    for (int k = 0, n = ref_n(r); k < n; ++k) {
      const int c = entry_feature(e[k]);
      if (taken_by[c] >= i) f(entry_dist(e[k]), entry_level(e[k]), c);
    }
    return;
  }
  const unsigned long long* D = reinterpret_cast<const unsigned long long*>(P.mpdesc1 + (size_t)i * 32);
  const unsigned long long d[4] = {D[0], D[1], D[2], D[3]};
  for_candidates(P, P.win[i], P.rng[i], [&](int c, int) {
    if (taken_by[c] < i) return;
    f(hamming256(d, reinterpret_cast<const unsigned long long*>(P.desc2 + (size_t)c * 32)), P.oct2[c], c);
  });
}

// grid = 1, block = kResolveBS; LDS, round stamps and the merged decide + commit pass as for k_proj_resolve
template <bool LDS>
__global__ __launch_bounds__(kResolveBS) void k_local_resolve(ProjDev P) {
  __shared__ int s_unres[2], s_total;
  __shared__ int32_t s_taken[LDS ? kResolveLdsN2 : 1], s_min[LDS ? kResolveLdsN2 : 1];
  __shared__ uint8_t s_state[LDS ? kResolveLdsN1 : 1];
  int32_t* taken_by = LDS ? s_taken : P.taken_by;
  int32_t* min_unres = LDS ? s_min : P.min_unres;
  uint8_t* state = LDS ? s_state : P.state;
  __shared__ uint32_t s_ref[LDS ? kResolveLdsN1 : 1], s_list[LDS ? kResolveLdsList : 1];
  __shared__ uint16_t s_ulist[2][LDS ? kUCap : 1];
  const ListView staged = stage_lists<LDS>(P, s_taken, s_state, s_ref, s_list, &s_total);
  // Instantiated twice, for the lists in LDS and for the lists where the candidate kernel left them: with ONE pointer chosen at run
  // time the entries were read with flat loads
  auto run = [&](const ListView L) {
    auto announce = [&](int i, int stamp) {
      if (state[i] != kStateObs) return;   // unresolved and a blocker
      const int me = (stamp << 20) | i;
      local_available(P, taken_by, L, i, [&](int dist, int, int c) { if (dist <= 100) atomicMin(&min_unres[c], me); });
    };
    // (a settled point's available candidates carry no announcement of a lower point, so no lower point can occupy one of them in
    // this pass; a higher point's store leaves taken_by[c] >= i: still available to i)
    auto decide = [&](int i, int stamp) {
      const uint8_t st = state[i];
      if (st & 0x7f) return false;
      LocalScan sc;
      bool settled = true;
      local_available(P, taken_by, L, i, [&](int dist, int level, int c) {
        sc.visit(dist, level, c);
        if (lower_unresolved(min_unres[c], stamp, i)) settled = false;  // somebody in front of i may still take this feature
      });
      if (!settled) return true;
      const int c = sc.accept(P.nnratio);
      state[i] = st | 1; P.choice[i] = c;
      if (c >= 0 && (st & kStateObs)) taken_by[c] = i;
      return false;
    };
    resolve_rounds<LDS>(P.n1, P.n2, min_unres, s_unres, s_ulist, announce, decide);
  };
  if (LDS && staged.clist != P.clist) run(ListView{s_ref, s_list});
  else run(ListView{LDS ? s_ref : P.cref, P.clist});
}

// ------------------------------------------------------------------------------------------------
// ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize)
// (/root/reference/src/ORBmatcher.cc:648-763; Tracking::MonocularInitialization, Tracking.cc:2526) with
// Frame::GetFeaturesInArea(x, y, windowSize, 0, 0) (src/Frame.cc:747-813).  The loop is sequential through
// vMatchedDistance: a level-0 feature of F1 skips every F2 candidate that an earlier feature already holds at a distance
// <= its own, and takes a candidate over from its holder otherwise.  vMatchedDistance only ever decreases, and only the
// candidates below `max_dist + 1` = the smallest d > TH_LOW with (float)d * mfNNratio > TH_LOW can change a decision (a larger
// second-best passes the ratio test like INT_MAX does, a larger best fails TH_LOW), so: the candidates of that range are
// kept in traversal order (ProjDev: proj1 = vbPrevMatched (2 floats), mpdesc1 = F1 descriptors, oct1 = F1 octaves,
// taken_by = vMatchedDistance, owner = vnMatches21), and a feature is final once no unresolved lower-index feature shares
// one of its still-available candidates - two features that share one resolve in index order, exactly the order of the loop.
template <class F>
__device__ __forceinline__ void init_available(const ProjDev& P, const int32_t* taken_by, const ListView& L, int i, F&& f) {  // f(dist, c) in traversal order
  const uint32_t r = L.cref[i];
  if (!(r & kRefOverflow)) {
    const uint32_t* e = L.clist + ref_start(r);
    for (int k = 0, n = ref_n(r); k < n; ++k) {
      const int c = entry_feature(e[k]), dist = entry_dist(e[k]);
      if (taken_by[c] > dist) f(dist, c);  // vMatchedDistance[i2] <= dist: continue (:689-690)
    }
    return;
  }
  const unsigned long long* D = reinterpret_cast<const unsigned long long*>(P.mpdesc1 + (size_t)i * 32);
  const unsigned long long d[4] = {D[0], D[1], D[2], D[3]};
  for_candidates(P, P.win[i], P.rng[i], [&](int c, int) {
    const int dist = hamming256(d, reinterpret_cast<const unsigned long long*>(P.desc2 + (size_t)c * 32));
    if (dist <= P.max_dist && taken_by[c] > dist) f(dist, c);
  });
}

// grid = ceil(n1 / 64), block = 64
__global__ __launch_bounds__(256) void k_init_candidates(ProjDev P) {   // one wave per F1 feature (wave_candidates), grid = ceil(n1 / 4)
  __shared__ unsigned long long s_keys[kPointsPerBlock][kProjCand];
  const int i = blockIdx.x * kPointsPerBlock + wave_id();
  if (i >= P.n1) return;
  const int lane = lane_id();
  uint8_t st = 1;
  uint32_t ref = 0;
  const float x = P.proj1[2 * i], y = P.proj1[2 * i + 1];
  if (P.oct1[i] == 0 && x == x && y == y) {  // level1 > 0: continue (:664-666)
    const float radius = P.th;
    const int x0 = imax(0, (int)floorf((x - P.grid[0] - radius) * P.grid[4]));
    const int x1 = imin(kGridCols - 1, (int)ceilf((x - P.grid[0] + radius) * P.grid[4]));
    const int y0 = imax(0, (int)floorf((y - P.grid[1] - radius) * P.grid[5]));
    const int y1 = imin(kGridRows - 1, (int)ceilf((y - P.grid[1] + radius) * P.grid[5]));
    if (x0 < kGridCols && x1 >= 0 && y0 < kGridRows && y1 >= 0) {
      const float4 w = make_float4(x, y, radius, 0.f);
      const int4 rg = make_int4(x0 | (x1 << 8), y0 | (y1 << 8), 0, 0);  // minLevel = maxLevel = level1 = 0
      const unsigned long long* D = reinterpret_cast<const unsigned long long*>(P.mpdesc1 + (size_t)i * 32);
      const unsigned long long d[4] = {D[0], D[1], D[2], D[3]};
      auto dist_of = [&](int c) { return hamming256(d, reinterpret_cast<const unsigned long long*>(P.desc2 + (size_t)c * 32)); };
      const int total = wave_candidates(P, w, rg, s_keys[wave_id()], [&](int c, int) {
        const int dist = dist_of(c);
        return dist <= P.max_dist ? (((unsigned long long)dist << 32) | (unsigned)c) : ~0ull;
      });
      if (lane == 0) { P.win[i] = w; P.rng[i] = rg; }
      ref = publish_candidates(P, i, s_keys[wave_id()], total, false,
                               [](unsigned long long key) { return cand_entry((int)(key >> 32), 0, (int)(key & 0xffffu)); });
      st = total > 0 ? 0 : 1;
    }
  }
  if (lane == 0) { P.choice[i] = -1; P.cref[i] = ref; P.state[i] = st; }
}

// grid = 1, block = kResolveBS; state, lists, round stamps and the merged decide + commit pass as for k_proj_resolve (taken_by holds
// vMatchedDistance here).  Merging is safe for the same reason: two features that are final in one round share no available
// candidate (the higher one would have found the lower one's announcement), and lowering vMatchedDistance[c] can only take c away
// from features that were going to lose it to this one anyway - they wait behind its announcement.
template <bool LDS>
__global__ __launch_bounds__(kResolveBS) void k_init_resolve(ProjDev P) {
  __shared__ int s_unres[2], s_total;
  __shared__ int32_t s_taken[LDS ? kResolveLdsN2 : 1], s_min[LDS ? kResolveLdsN2 : 1];
  __shared__ uint8_t s_state[LDS ? kResolveLdsN1 : 1];
  __shared__ uint32_t s_ref[LDS ? kResolveLdsN1 : 1], s_list[LDS ? kResolveLdsList : 1];
  __shared__ uint16_t s_ulist[2][LDS ? kUCap : 1];
  int32_t* taken_by = LDS ? s_taken : P.taken_by;
  int32_t* min_unres = LDS ? s_min : P.min_unres;
  uint8_t* state = LDS ? s_state : P.state;
  const int tid = threadIdx.x;
  for (int c = tid; c < P.n2; c += kResolveBS) P.owner[c] = -1;
  const ListView staged = stage_lists<LDS>(P, s_taken, s_state, s_ref, s_list, &s_total);
  // Instantiated twice, for the lists in LDS and for the lists where the candidate kernel left them: with ONE pointer chosen at run
  // time the entries were read with flat loads
  auto run = [&](const ListView L) {
    auto announce = [&](int i, int stamp) {
      if (state[i] != 0) return;
      const int me = (stamp << 20) | i;
      init_available(P, taken_by, L, i, [&](int, int c) { atomicMin(&min_unres[c], me); });
    };
    auto decide = [&](int i, int stamp) {
      if (state[i] != 0) return false;
      int best = INT_MAX, best2 = INT_MAX, best_idx = -1;
      bool settled = true;
      init_available(P, taken_by, L, i, [&](int dist, int c) {
        if (dist < best) { best2 = best; best = dist; best_idx = c; }  // :692-701
        else if (dist < best2) best2 = dist;
        if (lower_unresolved(min_unres[c], stamp, i)) settled = false;
      });
      if (!settled) return true;
      const bool ok = best <= 50 /* TH_LOW */ && (float)best < (float)best2 * P.nnratio;  // :704-706
      state[i] = 1;
      P.choice[i] = ok ? best_idx : -1;
      if (ok) { taken_by[best_idx] = best; P.owner[best_idx] = i; }  // vMatchedDistance, vnMatches21 (:713-715)
      return false;
    };
    resolve_rounds<LDS>(P.n1, P.n2, min_unres, s_unres, s_ulist, announce, decide);
  };
  if (LDS && staged.clist != P.clist) run(ListView{s_ref, s_list});
  else run(ListView{LDS ? s_ref : P.cref, P.clist});
}

// ------------------------------------------------------------------------------------------------
// DBoW2 vocabulary descent: TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup)
// (/root/reference/Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1208-1255) with FORB::distance (FORB.cpp:79-98), the
// per-feature part of Frame::ComputeBoW (src/Frame.cc:828-835).  Work-item per feature: L levels, at each level the
// child with the smallest Hamming distance (strict '<', first child wins ties).  The tree is resident on the device:
// children CSR in file order, 32-byte node descriptors.
struct VocDev {
  int n_nodes, L;
  const int32_t* child_off;
  const int32_t* child;
  const uint8_t* desc;
  const double* weight;
  const int32_t* word_id;
};
// grid = (ceil(cap / 256), frames), block = 256; d_n == nullptr: every frame holds `cap` features
__global__ __launch_bounds__(256) void k_bow_descend(VocDev V, const uint8_t* __restrict__ desc, const int32_t* __restrict__ d_n,
                                                     int cap, int levelsup, int32_t* __restrict__ word, double* __restrict__ weight,
                                                     int32_t* __restrict__ node) {
  const int i = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
  const int n = d_n ? d_n[f] : cap;
  if (i >= n || i >= cap) return;
  const size_t o = (size_t)f * cap + i;
  const unsigned long long* D = reinterpret_cast<const unsigned long long*>(desc + o * 32);
  const unsigned long long d[4] = {D[0], D[1], D[2], D[3]};
  const int nid_level = V.L - levelsup;
  int nid = 0, final_id = 0, current_level = 0;
  int b = V.child_off[0], e = V.child_off[1];
  while (b != e) {  // an inner node
    ++current_level;
    int best_d = 257;
    for (int k = b; k < e; ++k) {
      const int id = V.child[k];
      const int dist = hamming256(d, reinterpret_cast<const unsigned long long*>(V.desc + (size_t)id * 32));
      if (dist < best_d) { best_d = dist; final_id = id; }
    }
    if (current_level == nid_level) nid = final_id;
    b = V.child_off[final_id];
    e = V.child_off[final_id + 1];
  }
  word[o] = V.word_id[final_id];
  weight[o] = V.weight[final_id];
  node[o] = nid;
}

// ------------------------------------------------------------------------------------------------
// ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches) (ORBmatcher.cc:223-425, single
// camera), the matcher of Tracking::TrackReferenceKeyFrame.  The vocabulary nodes shared by both FeatureVectors are
// independent problems (a feature lives in exactly one node), inside a node the key-frame features are visited in
// bucket order and a frame feature that got a match is skipped by all later ones.  One wave per shared node: the
// key-frame features one after the other, the frame bucket spread over the lanes; best / second-best with the
// reference's order rules = minimum of (distance << 16 | bucket position) and the second order statistic.
struct BowDev {
  const uint8_t* desc1; const uint8_t* desc2;
  const uint8_t* valid1;
  const uint8_t* valid2;  // candidates must hold a good map point as well (the key-frame / key-frame overload), or nullptr
  const int32_t *off1, *feat1, *off2, *feat2, *pair_n1, *pair_n2;
  float nnratio;
  int max_best;           // largest accepted best distance: TH_LOW ("<=", :315) or TH_LOW - 1 ("<", :854)
  int n_left2;            // F.Nleft of a two-camera frame (k_search_by_bow_rig): features from n_left2 on belong to the right camera
  int32_t* match1;  // per key-frame feature: the frame feature it took, or -1
  int32_t* match2;  // per frame feature: the key-frame feature, or -1
};
constexpr int kBowBucket = 16384;  // frame features of one node tracked in LDS (one byte each)

// The node's work for a frame bucket of up to 64 * kRegTrips features (kRegTrips = 4: any size, positions from 256 on are read from
// memory): instantiated per trip count so that a node pays for the lanes' registers it uses - the serial chain of the LARGEST node
// is the kernel's time, and with one body for all sizes every step computed four trips' distances.
template <int kRegTrips>
__device__ __forceinline__ void bow_node(const BowDev& T, uint8_t* s_taken, unsigned long long (*s_q)[4], int* s_idx1, int lane, int b1, int e1, int b2, int n2) {
  // the frame features at bucket positions lane, lane + 64, lane + 128, lane + 192 live in the lane's registers (index,
  // descriptor and whether the feature is taken): buckets of up to 256 features - a KITTI frame's largest hold ~100 - cost
  // neither a global nor an LDS access inside the serial loop
  for (int j = lane + 64 * kRegTrips; j < n2; j += 64) s_taken[j] = (T.valid2 && !T.valid2[T.feat2[b2 + j]]) ? 1 : 0;   // beyond the register trips
  unsigned long long t0[kRegTrips][4];
  int my_idx2[kRegTrips];
  bool gone[kRegTrips];   // taken, masked out, or no such position
#pragma unroll
  for (int r = 0; r < kRegTrips; ++r) {
    my_idx2[r] = -1;
    gone[r] = true;
    t0[r][0] = t0[r][1] = t0[r][2] = t0[r][3] = 0ull;
    if (lane + 64 * r < n2) {
      my_idx2[r] = T.feat2[b2 + lane + 64 * r];
      gone[r] = T.valid2 && !T.valid2[my_idx2[r]];
      const unsigned long long* D2 = reinterpret_cast<const unsigned long long*>(T.desc2 + (size_t)my_idx2[r] * 32);
      t0[r][0] = D2[0]; t0[r][1] = D2[1]; t0[r][2] = D2[2]; t0[r][3] = D2[3];
    }
  }
  for (int p0 = b1; p0 < e1; p0 += 64) {
    wave_sync();
    {
      const int p = p0 + lane;
      int idx1 = -1;
      if (p < e1) { idx1 = T.feat1[p]; if (!T.valid1[idx1]) idx1 = -1; }
      s_idx1[lane] = idx1;
      if (idx1 >= 0) {
        const unsigned long long* D1 = reinterpret_cast<const unsigned long long*>(T.desc1 + (size_t)idx1 * 32);
        s_q[lane][0] = D1[0]; s_q[lane][1] = D1[1]; s_q[lane][2] = D1[2]; s_q[lane][3] = D1[3];
      }
    }
    wave_sync();
    const int cnt = imin(64, e1 - p0);
    // the next key-frame feature's descriptor is fetched from LDS while the current one is worked on
    int idx1_n = s_idx1[0];
    unsigned long long qn[4] = {s_q[0][0], s_q[0][1], s_q[0][2], s_q[0][3]};
    for (int k = 0; k < cnt; ++k) {
      const int idx1 = idx1_n;
      const unsigned long long q[4] = {qn[0], qn[1], qn[2], qn[3]};
      if (k + 1 < cnt) {
        idx1_n = s_idx1[k + 1];
        qn[0] = s_q[k + 1][0]; qn[1] = s_q[k + 1][1]; qn[2] = s_q[k + 1][2]; qn[3] = s_q[k + 1][3];
      }
      if (idx1 < 0) continue;  // wave-uniform
      uint32_t best = 0xffffffffu;  // dist << 16 | bucket position
      uint32_t second = 256;
      // best / second-best without a branch (the loop body is what a node's serial chain is made of): the smaller key stays, the
      // larger one's distance competes for second place - the key it displaces when it is the new best (second >= best's distance
      // always), itself otherwise; an absent candidate is the key 0xffffffff and changes nothing
      auto visit = [&](uint32_t key) {
        const uint32_t lo = key < best ? key : best, hi = key < best ? best : key;
        second = (hi >> 16) < second ? (hi >> 16) : second;
        best = lo;
      };
#pragma unroll
      for (int r = 0; r < kRegTrips; ++r) {
        if (64 * r >= n2) break;  // wave-uniform
        const uint32_t key = ((uint32_t)hamming256(q, t0[r]) << 16) | (uint32_t)(lane + 64 * r);
        visit(gone[r] ? 0xffffffffu : key);
      }
      for (int j = lane + 64 * kRegTrips; j < n2; j += 64) {
        if (s_taken[j]) continue;
        const int idx2 = T.feat2[b2 + j];
        visit(((uint32_t)hamming256(q, reinterpret_cast<const unsigned long long*>(T.desc2 + (size_t)idx2 * 32)) << 16) | (uint32_t)j);
      }
      // wave-wide: the smallest key, and the smallest distance among everything else
      const uint32_t wbest = wave_min_uniform(best);   // register-file DPP steps: the two reductions sit in the node's serial loop
      const int other = (int)wave_min_uniform(best == wbest ? second : (best == 0xffffffffu ? 256u : best >> 16));
      if (wbest == 0xffffffffu) continue;
      const int best_dist = (int)(wbest >> 16), pos = (int)(wbest & 0xffffu);
      if (best_dist <= T.max_best && (float)best_dist < T.nnratio * (float)other) {
        if (lane == (pos & 63)) {   // the lane that holds the feature's index
          int idx2 = -1;
#pragma unroll
          for (int r = 0; r < kRegTrips; ++r)
            if ((pos >> 6) == r) { idx2 = my_idx2[r]; gone[r] = true; }
          if (pos >= 64 * kRegTrips) { idx2 = T.feat2[b2 + pos]; s_taken[pos] = 1; }
          T.match1[idx1] = idx2;
          T.match2[idx2] = idx1;
        }
        if (pos >= 64 * kRegTrips) wave_sync();   // wave-uniform: the other lanes read s_taken
      }
    }
  }
}

// grid = shared nodes, block = 64.  The node's key-frame features are taken one after the other (a frame feature taken by an
// earlier one is gone for the later ones: ORBmatcher.cc:296-299), so what sits inside that serial loop decides the kernel's
// time.  Round 6: the key-frame features' descriptors are staged in LDS 64 at a time and a lane keeps the descriptor of ITS
// frame feature (bucket position = lane, every bucket of a KITTI frame's ~100 nodes fits) in registers - the loop body is LDS
// reads and register work instead of three dependent global loads per key-frame feature (100 -> ~15 us for a frame pair).
__global__ __launch_bounds__(64) void k_search_by_bow(BowDev T) {
  __shared__ uint8_t s_taken[kBowBucket];
  __shared__ unsigned long long s_q[64][4];
  __shared__ int s_idx1[64];
  const int np = blockIdx.x, lane = threadIdx.x;
  const int a = T.pair_n1[np], b = T.pair_n2[np];
  const int b1 = T.off1[a], e1 = T.off1[a + 1], b2 = T.off2[b], n2 = T.off2[b + 1] - b2;
  if (n2 <= 64) bow_node<1>(T, s_taken, s_q, s_idx1, lane, b1, e1, b2, n2);
  else if (n2 <= 128) bow_node<2>(T, s_taken, s_q, s_idx1, lane, b1, e1, b2, n2);
  else if (n2 <= 192) bow_node<3>(T, s_taken, s_q, s_idx1, lane, b1, e1, b2, n2);
  else bow_node<4>(T, s_taken, s_q, s_idx1, lane, b1, e1, b2, n2);
}


// Two-camera frames (F.Nleft != -1, ORBmatcher.cc:298-326, 357-386): a key-frame feature keeps a best / second best among the
// node's LEFT frame features and a best among its RIGHT ones; the left one is taken under the usual tests, the right one - only if
// the left best passed `<= TH_LOW` - whenever its own distance does (`|| true`: no ratio test).  Both are gone for the later
// key-frame features.  One wave per node as above; the frame features of a lane live in registers for buckets of up to 256.
__device__ __forceinline__ void bow_node_rig(const BowDev& T, uint8_t* s_taken, unsigned long long (*s_q)[4], int* s_idx1, int lane, int b1, int e1, int b2, int n2) {
  constexpr int kRegTrips = 4;
  for (int j = lane + 64 * kRegTrips; j < n2; j += 64) s_taken[j] = 0;
  unsigned long long t0[kRegTrips][4];
  int my_idx2[kRegTrips];
  bool gone[kRegTrips], right[kRegTrips];
#pragma unroll
  for (int r = 0; r < kRegTrips; ++r) {
    my_idx2[r] = -1;
    gone[r] = true;
    right[r] = false;
    t0[r][0] = t0[r][1] = t0[r][2] = t0[r][3] = 0ull;
    if (lane + 64 * r < n2) {
      my_idx2[r] = T.feat2[b2 + lane + 64 * r];
      gone[r] = false;
      right[r] = my_idx2[r] >= T.n_left2;
      const unsigned long long* D2 = reinterpret_cast<const unsigned long long*>(T.desc2 + (size_t)my_idx2[r] * 32);
      t0[r][0] = D2[0]; t0[r][1] = D2[1]; t0[r][2] = D2[2]; t0[r][3] = D2[3];
    }
  }
  for (int p0 = b1; p0 < e1; p0 += 64) {
    wave_sync();
    {
      const int p = p0 + lane;
      int idx1 = -1;
      if (p < e1) { idx1 = T.feat1[p]; if (!T.valid1[idx1]) idx1 = -1; }
      s_idx1[lane] = idx1;
      if (idx1 >= 0) {
        const unsigned long long* D1 = reinterpret_cast<const unsigned long long*>(T.desc1 + (size_t)idx1 * 32);
        s_q[lane][0] = D1[0]; s_q[lane][1] = D1[1]; s_q[lane][2] = D1[2]; s_q[lane][3] = D1[3];
      }
    }
    wave_sync();
    const int cnt = imin(64, e1 - p0);
    for (int k = 0; k < cnt; ++k) {
      const int idx1 = s_idx1[k];
      if (idx1 < 0) continue;  // wave-uniform
      const unsigned long long q[4] = {s_q[k][0], s_q[k][1], s_q[k][2], s_q[k][3]};
      uint32_t best = 0xffffffffu, second = 256;   // left camera: dist << 16 | bucket position, and the runner-up's distance
      uint32_t best_r = 0xffffffffu;               // right camera: the first minimum is all that is used
      auto visit = [&](uint32_t key, bool is_right) {
        const uint32_t kl = is_right ? 0xffffffffu : key, kr = is_right ? key : 0xffffffffu;
        const uint32_t lo = kl < best ? kl : best, hi = kl < best ? best : kl;
        second = (hi >> 16) < second ? (hi >> 16) : second;
        best = lo;
        best_r = kr < best_r ? kr : best_r;
      };
#pragma unroll
      for (int r = 0; r < kRegTrips; ++r) {
        if (64 * r >= n2) break;  // wave-uniform
        const uint32_t key = ((uint32_t)hamming256(q, t0[r]) << 16) | (uint32_t)(lane + 64 * r);
        visit(gone[r] ? 0xffffffffu : key, right[r]);
      }
      for (int j = lane + 64 * kRegTrips; j < n2; j += 64) {
        if (s_taken[j]) continue;
        const int idx2 = T.feat2[b2 + j];
        visit(((uint32_t)hamming256(q, reinterpret_cast<const unsigned long long*>(T.desc2 + (size_t)idx2 * 32)) << 16) | (uint32_t)j, idx2 >= T.n_left2);
      }
      const uint32_t wbest = wave_min_uniform(best);
      const int other = (int)wave_min_uniform(best == wbest ? second : (best == 0xffffffffu ? 256u : best >> 16));
      const uint32_t wbest_r = wave_min_uniform(best_r);
      if (wbest == 0xffffffffu || (int)(wbest >> 16) > T.max_best) continue;   // :315: the right camera's match sits inside this test
      const int best_dist = (int)(wbest >> 16);
      const bool take_l = (float)best_dist < T.nnratio * (float)other;
      const bool take_r = wbest_r != 0xffffffffu && (int)(wbest_r >> 16) <= T.max_best;
      bool beyond = false;
#pragma unroll
      for (int side = 0; side < 2; ++side) {
        if (!(side == 0 ? take_l : take_r)) continue;   // wave-uniform
        const int pos = (int)((side == 0 ? wbest : wbest_r) & 0xffffu);
        if (lane == (pos & 63)) {
          int idx2 = -1;
#pragma unroll
          for (int r = 0; r < kRegTrips; ++r)
            if ((pos >> 6) == r) { idx2 = my_idx2[r]; gone[r] = true; }
          if (pos >= 64 * kRegTrips) { idx2 = T.feat2[b2 + pos]; s_taken[pos] = 1; }
          if (side == 0) T.match1[idx1] = idx2;   // the result is match2; match1 keeps the left camera's feature
          T.match2[idx2] = idx1;
        }
        beyond = beyond || pos >= 64 * kRegTrips;
      }
      if (beyond) wave_sync();   // wave-uniform: the other lanes read s_taken
    }
  }
}

__global__ __launch_bounds__(64) void k_search_by_bow_rig(BowDev T) {
  __shared__ uint8_t s_taken[kBowBucket];
  __shared__ unsigned long long s_q[64][4];
  __shared__ int s_idx1[64];
  const int np = blockIdx.x, lane = threadIdx.x;
  const int a = T.pair_n1[np], b = T.pair_n2[np];
  const int b1 = T.off1[a], e1 = T.off1[a + 1], b2 = T.off2[b], n2 = T.off2[b + 1] - b2;
  bow_node_rig(T, s_taken, s_q, s_idx1, lane, b1, e1, b2, n2);
}

}  // namespace rgbl

using namespace rgbl;

struct rgbl_matcher {
  int device = 0;
  hipStream_t stream = nullptr, own_stream = nullptr;
  KernelTimer timer;
  uint8_t* d_buf = nullptr;  // grow-only staging arena for the host entry points
  size_t buf_size = 0;
  uint8_t* h_pin = nullptr;  // grow-only page-locked mirror of the brute-force scan's inputs / outputs (rgbl_hamming_bf)
  size_t pin_size = 0;
  // tuning switches, read ONCE at rgbl_matcher_create (include/rgbl_frontend.h lists them): never on a launch path
  bool bf_matrix = true;   // RGBL_BF_MFMA=0: the VALU popcount scan (k_hamming_bf)
  bool bf_fp4 = true;      // RGBL_BF_MFMA=i8: v_mfma_i32_32x32x32_i8 (k_hamming_mfma) instead of the block-scaled FP4 instruction
  bool bf_split = true;    // RGBL_BF_SPLIT=0: one pair per call without train-set slices
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
// RGBL_BF_MFMA=0 selects the VALU popcount scan (k_hamming_bf) instead of the matrix-core one, =i8 the i8 instruction
// (k_hamming_mfma) instead of the block-scaled FP4 one (k_hamming_fp4, the default): read when the handle is created.
inline bool bf_on_matrix_cores(const rgbl_matcher* m) { return m->bf_matrix; }
inline bool bf_on_fp4(const rgbl_matcher* m) { return m->bf_fp4; }
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
int ensure_pin(rgbl_matcher* m, size_t bytes) {
  if (bytes <= m->pin_size) return RGBL_OK;
  if (m->h_pin) { (void)hipHostFree(m->h_pin); m->h_pin = nullptr; m->pin_size = 0; }
  const size_t sz = std::max(bytes, (size_t)1 << 20);
  RGBL_HIP(hipHostMalloc(reinterpret_cast<void**>(&m->h_pin), sz, hipHostMallocDefault));
  m->pin_size = sz;
  return RGBL_OK;
}
// One call of a host-pointer entry point.  The device arena and its page-locked mirror share ONE layout: what the call
// uploads is copied into the mirror at the offsets its device copies have and goes up with ONE hipMemcpyAsync (round 5
// queued one per array, 8 - 17 per call, from pageable memory); the result arrays are taken back to back and come back
// with one.  Arrays of a frame that is resident on the device (rgbl_device_frame) are not staged at all.
struct HostCall {
  rgbl_matcher* m = nullptr;
  hipStream_t s = nullptr;
  Arena A{nullptr};
  size_t up_end = 0, res_begin = ~(size_t)0, res_end = 0;
  int begin(rgbl_matcher* mm, size_t need) {
    m = mm; s = mm->stream;
    RGBL_TRY(ensure_arena(m, need));
    RGBL_TRY(ensure_pin(m, need));
    A = Arena{m->d_buf};
    return RGBL_OK;
  }
  size_t off_of(const void* d) const { return (size_t)(reinterpret_cast<const uint8_t*>(d) - m->d_buf); }
  template <class E> int put(const E** dst, const E* src, size_t count) {
    E* p = A.take<E>(count);
    *dst = p;
    if (count) memcpy(m->h_pin + off_of(p), src, count * sizeof(E));
    up_end = A.off;
    return RGBL_OK;
  }
  template <class E> E* put_fill(size_t count, int byte) {   // an array the device starts from (e.g. all -1) travels with the upload
    E* p = A.take<E>(count);
    if (count) memset(m->h_pin + off_of(p), byte, count * sizeof(E));
    up_end = A.off;
    return p;
  }
  int upload() {
    if (up_end) RGBL_HIP(hipMemcpyAsync(m->d_buf, m->h_pin, up_end, hipMemcpyHostToDevice, s));
    return RGBL_OK;
  }
  template <class E> E* stage(size_t count) { E* p = A.take<E>(count); up_end = A.off; return p; }   // the caller fills mirror(p)
  template <class E> E* mirror(E* d) { return reinterpret_cast<E*>(m->h_pin + off_of(d)); }
  template <class E> E* scratch(size_t count) { return A.take<E>(count); }
  template <class E> void mark_result(E* p, size_t count) {
    res_begin = std::min(res_begin, off_of(p));
    res_end = std::max(res_end, off_of(p) + count * sizeof(E));
  }
  template <class E> E* result(size_t count) { E* p = A.take<E>(count); mark_result(p, count); return p; }
  int fetch() {   // the call's only wait
    if (res_end > res_begin) RGBL_HIP(hipMemcpyAsync(m->h_pin + res_begin, m->d_buf + res_begin, res_end - res_begin, hipMemcpyDeviceToHost, s));
    RGBL_HIP(hipStreamSynchronize(s));
    m->timer.collect();
    return RGBL_OK;
  }
  template <class E> const E* host(const E* d) const { return reinterpret_cast<const E*>(m->h_pin + off_of(d)); }
  // a frame's resident arrays: the stream waits for whatever filled them last
  int use(const rgbl_device_frame* f, int n) {
    if (f->device != m->device || f->n != n) { set_error("device frame: %d features on device %d, the call says %d on device %d", f->n, f->device, n, m->device); return RGBL_ERR_INVALID; }
    RGBL_HIP(hipStreamWaitEvent(s, f->ready, 0));
    return RGBL_OK;
  }
};
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
  if (const char* v = getenv("RGBL_BF_MFMA")) { m->bf_matrix = v[0] != '0'; m->bf_fp4 = v[0] != 'i'; }
  if (const char* v = getenv("RGBL_BF_SPLIT")) m->bf_split = v[0] != '0';
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
  if (m->h_pin) (void)hipHostFree(m->h_pin);
  if (m->own_stream) (void)hipStreamDestroy(m->own_stream);
  delete m;
}

// Pool of matcher handles.  The reference builds its ORBmatcher as a function-local object in every Tracking / LocalMapping /
// LoopClosing call (Tracking.cc:2525, 2761, 2890, 3424, 3662, 3701; LocalMapping.cc:412): a drop-in class that created a HIP
// stream (and later freed a device arena, which synchronises the whole device) per object would pay that several times per
// frame.  acquire() hands out an idle handle of the device or creates one; release() parks it again - stream and arena stay
// alive for the life of the process.  Handles in use at the same time are distinct, so concurrent callers keep overlapping.
namespace {
struct MatcherPool {
  std::mutex mu;
  std::vector<rgbl_matcher*> idle;
};
MatcherPool* matcher_pool_ptr() {
  static MatcherPool* pool = new MatcherPool;  // never destroyed: no HIP calls from static destructors
  return pool;
}
}  // namespace

int rgbl_matcher_acquire(int device, rgbl_matcher** out) {
  if (!out) { set_error("null argument"); return RGBL_ERR_INVALID; }
  {
    MatcherPool& P = *matcher_pool_ptr();
    std::lock_guard<std::mutex> lock(P.mu);
    for (size_t i = 0; i < P.idle.size(); ++i)
      if (P.idle[i]->device == device) {
        *out = P.idle[i];
        P.idle.erase(P.idle.begin() + (long)i);
        return RGBL_OK;
      }
  }
  return rgbl_matcher_create(device, out);
}

void rgbl_matcher_release(rgbl_matcher* m) {
  if (!m) return;
  if (m->stream != m->own_stream) {  // a borrowed stream must not outlive its owner's wishes
    (void)hipStreamSynchronize(m->stream);
    m->stream = m->own_stream;
  }
  if (m->timer.enabled || !m->timer.recs.empty()) {  // the next owner must not inherit profiling brackets
    (void)hipStreamSynchronize(m->stream);
    m->timer.collect();
    m->timer.enabled = false;
  }
  m->timer.reset();
  MatcherPool& P = *matcher_pool_ptr();
  std::lock_guard<std::mutex> lock(P.mu);
  P.idle.push_back(m);
}

int rgbl_matcher_pool_clear(void) {
  std::vector<rgbl_matcher*> idle;
  {
    MatcherPool& P = *matcher_pool_ptr();
    std::lock_guard<std::mutex> lock(P.mu);
    idle.swap(P.idle);
  }
  for (rgbl_matcher* m : idle) rgbl_matcher_destroy(m);
  return (int)idle.size();
}

int rgbl_matcher_pool_size(void) {
  MatcherPool& P = *matcher_pool_ptr();
  std::lock_guard<std::mutex> lock(P.mu);
  return (int)P.idle.size();
}

// ---- frames resident on the device (include/rgbl_frontend.h) -----------------------------------------------------------
int rgbl_device_frame_create(int device, int capacity, rgbl_device_frame** out) {
  if (!out || capacity < 1 || capacity > 65535) { set_error("device frame: capacity 1 .. 65535"); return RGBL_ERR_INVALID; }
  *out = nullptr;
  if (rgbl_device_count() <= device || device < 0) {
    set_error("no usable HIP device %d (this library has no CPU fallback)", device);
    return RGBL_ERR_NO_DEVICE;
  }
  RGBL_HIP(hipSetDevice(device));
  rgbl_device_frame* f = new rgbl_device_frame;
  f->device = device;
  f->cap = (capacity + 63) / 64 * 64;   // every array starts 256-byte aligned
  const size_t cap = (size_t)f->cap;
  if (hipMalloc(&f->block, cap * 48) != hipSuccess || hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&f->ready, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    rgbl_device_frame_destroy(f);
    set_error("device frame: allocation failed");
    return RGBL_ERR_HIP;
  }
  f->d_desc = f->block;
  f->d_xy = reinterpret_cast<float*>(f->block + cap * 32);
  f->d_oct = reinterpret_cast<int32_t*>(f->block + cap * 40);
  f->d_ur = reinterpret_cast<float*>(f->block + cap * 44);
  RGBL_HIP(hipEventRecord(f->ready, f->stream));
  *out = f;
  return RGBL_OK;
}

void rgbl_device_frame_destroy(rgbl_device_frame* f) {
  if (!f) return;
  (void)hipSetDevice(f->device);
  if (f->ready) { (void)hipEventSynchronize(f->ready); (void)hipEventDestroy(f->ready); }
  if (f->stream) { (void)hipStreamSynchronize(f->stream); (void)hipStreamDestroy(f->stream); }
  if (f->block) (void)hipFree(f->block);
  if (f->d_fv) (void)hipFree(f->d_fv);
  if (f->d_cell_start) (void)hipFree(f->d_cell_start);
  if (f->d_cell_items) (void)hipFree(f->d_cell_items);
  if (f->d_grid_scratch) (void)hipFree(f->d_grid_scratch);
  if (f->h_pin) (void)hipHostFree(f->h_pin);
  delete f;
}

static int frame_pin(rgbl_device_frame* f, size_t bytes) {
  if (bytes <= f->pin_size) return RGBL_OK;
  if (f->h_pin) { (void)hipHostFree(f->h_pin); f->h_pin = nullptr; f->pin_size = 0; }
  RGBL_HIP(hipHostMalloc(reinterpret_cast<void**>(&f->h_pin), bytes, hipHostMallocDefault));
  f->pin_size = bytes;
  return RGBL_OK;
}

int rgbl_device_frame_upload(rgbl_device_frame* f, int n, const uint8_t* desc, const float* kp_xy, const int32_t* kp_octave,
                             const float* uright) {
  if (!f || n < 0 || n > f->cap || (n > 0 && (!desc || !kp_xy || !kp_octave))) { set_error("device frame upload: invalid argument / more features than the capacity"); return RGBL_ERR_INVALID; }
  RGBL_HIP(hipSetDevice(f->device));
  const size_t cap = (size_t)f->cap;
  RGBL_TRY(frame_pin(f, cap * 48));
  (void)hipEventSynchronize(f->ready);   // nothing still fills the arrays from elsewhere
  memcpy(f->h_pin, desc, (size_t)n * 32);
  memcpy(f->h_pin + cap * 32, kp_xy, (size_t)n * 8);
  memcpy(f->h_pin + cap * 40, kp_octave, (size_t)n * 4);
  float* ur = reinterpret_cast<float*>(f->h_pin + cap * 44);
  if (uright) memcpy(ur, uright, (size_t)n * 4);
  else for (int i = 0; i < n; ++i) ur[i] = -1.f;
  // the mirror has the block's layout: one request for all four arrays
  RGBL_HIP(hipMemcpyAsync(f->block, f->h_pin, cap * 44 + (size_t)n * 4, hipMemcpyHostToDevice, f->stream));
  RGBL_HIP(hipEventRecord(f->ready, f->stream));
  RGBL_HIP(hipStreamSynchronize(f->stream));   // the mirror is reused by the next upload; the caller's arrays are free at once anyway
  f->n = n;
  f->has_grid = false;   // new keypoints: rgbl_device_frame_set_grid again
  return RGBL_OK;
}

int rgbl_device_frame_set_feature_vector(rgbl_device_frame* f, int n_nodes, const int32_t* node_off, const int32_t* node_feat) {
  if (!f || n_nodes < 0 || !node_off || node_off[0] != 0) { set_error("device frame: invalid FeatureVector"); return RGBL_ERR_INVALID; }
  const int nf = node_off[n_nodes];
  if (nf < 0 || nf > f->cap || (nf > 0 && !node_feat)) { set_error("device frame: FeatureVector with %d entries, capacity %d", nf, f->cap); return RGBL_ERR_INVALID; }
  RGBL_HIP(hipSetDevice(f->device));
  const int words = n_nodes + 1 + nf;
  if (words > f->fv_cap) {
    if (f->d_fv) { RGBL_HIP(hipStreamSynchronize(f->stream)); (void)hipFree(f->d_fv); f->d_fv = nullptr; f->fv_cap = 0; }
    const int cap = std::max(words, 2 * f->cap + 2);
    RGBL_HIP(hipMalloc(&f->d_fv, sizeof(int32_t) * (size_t)cap));
    f->fv_cap = cap;
  }
  RGBL_TRY(frame_pin(f, std::max((size_t)f->cap * 48, sizeof(int32_t) * (size_t)words)));
  (void)hipEventSynchronize(f->ready);
  int32_t* h = reinterpret_cast<int32_t*>(f->h_pin);
  memcpy(h, node_off, sizeof(int32_t) * ((size_t)n_nodes + 1));
  if (nf) memcpy(h + n_nodes + 1, node_feat, sizeof(int32_t) * (size_t)nf);
  RGBL_HIP(hipMemcpyAsync(f->d_fv, h, sizeof(int32_t) * (size_t)words, hipMemcpyHostToDevice, f->stream));
  RGBL_HIP(hipEventRecord(f->ready, f->stream));
  RGBL_HIP(hipStreamSynchronize(f->stream));
  f->n_nodes = n_nodes; f->nf = nf;
  return RGBL_OK;
}

int rgbl_device_frame_set_grid(rgbl_device_frame* f, const float grid[6]) {
  if (!f || !grid) { set_error("device frame: null argument"); return RGBL_ERR_INVALID; }
  RGBL_HIP(hipSetDevice(f->device));
  if (!f->d_cell_start) {
    RGBL_HIP(hipMalloc(&f->d_cell_start, sizeof(uint32_t) * (kGridCells + 1)));
    RGBL_HIP(hipMalloc(&f->d_cell_items, sizeof(uint16_t) * (size_t)f->cap));
    RGBL_HIP(hipMalloc(&f->d_grid_scratch, sizeof(int32_t) * (size_t)f->cap));
  }
  ProjDev P;
  memset(&P, 0, sizeof(P));
  P.n2 = f->n; P.xy2 = f->d_xy;
  memcpy(P.grid, grid, sizeof(P.grid));
  P.cell_start = f->d_cell_start; P.cell_items = f->d_cell_items; P.taken_by = f->d_grid_scratch;
  RGBL_HIP(hipStreamWaitEvent(f->stream, f->ready, 0));   // the keypoints may still be on their way (capture)
  launch_proj_grid(P, f->stream);
  RGBL_HIP(hipGetLastError());
  RGBL_HIP(hipEventRecord(f->ready, f->stream));
  RGBL_HIP(hipStreamSynchronize(f->stream));
  memcpy(f->grid, grid, sizeof(f->grid));
  f->has_grid = true;
  return RGBL_OK;
}

int rgbl_device_frame_size(const rgbl_device_frame* f) { return f ? f->n : 0; }

int rgbl_device_frame_download(rgbl_device_frame* f, uint8_t* desc, float* kp_xy, int32_t* kp_octave, float* uright) {
  if (!f) { set_error("null frame"); return RGBL_ERR_INVALID; }
  RGBL_HIP(hipSetDevice(f->device));
  RGBL_HIP(hipEventSynchronize(f->ready));
  const size_t n = (size_t)f->n;
  if (desc && n) RGBL_HIP(hipMemcpy(desc, f->d_desc, n * 32, hipMemcpyDeviceToHost));
  if (kp_xy && n) RGBL_HIP(hipMemcpy(kp_xy, f->d_xy, n * 8, hipMemcpyDeviceToHost));
  if (kp_octave && n) RGBL_HIP(hipMemcpy(kp_octave, f->d_oct, n * 4, hipMemcpyDeviceToHost));
  if (uright && n) RGBL_HIP(hipMemcpy(uright, f->d_ur, n * 4, hipMemcpyDeviceToHost));
  return RGBL_OK;
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
int rgbl_matcher_profile_samples(rgbl_matcher* m, int kernel, float* ms, int cap) {
  if (!m || (cap > 0 && !ms)) return 0;
  (void)hipStreamSynchronize(m->stream);
  m->timer.collect();
  return m->timer.read_samples(kernel, ms, cap);
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
  if (bf_on_matrix_cores(m)) {
    m->timer.begin(bf_on_fp4(m) ? "k_hamming_fp4" : "k_hamming_mfma", m->stream);
    if (bf_on_fp4(m)) hipLaunchKernelGGL(k_hamming_fp4, xcd_grid(true, (cap + kBfQueriesPerBlock - 1) / kBfQueriesPerBlock, n_pairs), dim3(256), 0, m->stream,
                       d_desc, d_n, cap, d_pair_a, d_pair_b, d_best_idx, d_best_dist, d_second_dist, 1, (uint32_t*)nullptr);
    else hipLaunchKernelGGL(k_hamming_mfma, xcd_grid(true, (cap + kBfQueriesPerBlock - 1) / kBfQueriesPerBlock, n_pairs), dim3(256), 0, m->stream,
                       d_desc, d_n, cap, d_pair_a, d_pair_b, d_best_idx, d_best_dist, d_second_dist);
  } else {
    m->timer.begin("k_hamming_bf", m->stream);
    hipLaunchKernelGGL(k_hamming_bf, dim3((cap + 63) / 64, n_pairs), dim3(256), 0, m->stream, d_desc, d_n, cap, d_pair_a,
                       d_pair_b, d_best_idx, d_best_dist, d_second_dist);
  }
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
  StreamDrain drain(m->stream);  // error returns included
  const int cap = std::max(std::max(na, nb), 1);
  // One pair per call: the query blocks alone (8 for 2000 descriptors) would leave most of the chip idle, so the train set is
  // cut into slices of whole 64-row stages, one launch slice each, folded by k_hamming_merge (FP4 kernel only).
  const int qblocks = (na + kBfQueriesPerBlock - 1) / kBfQueriesPerBlock, stages = (nb + kBfStageRows - 1) / kBfStageRows;
  int splits = 1;
  if (bf_on_matrix_cores(m) && bf_on_fp4(m) && m->bf_split)
    splits = std::max(1, std::min(std::min(16, stages / 4), 128 / std::max(qblocks, 1)));
  HostCall hc;
  RGBL_TRY(hc.begin(m, pad256((size_t)2 * cap * 32) + pad256(8) + 3 * pad256((size_t)na * 4) + pad256((size_t)splits * na * 8)));
  hipStream_t s = hc.s;
  // both descriptor sets and the counts in one block, ONE request up; the three result arrays come back with one
  uint8_t* d_desc = hc.stage<uint8_t>((size_t)2 * cap * 32);
  memcpy(hc.mirror(d_desc), desc_a, (size_t)na * 32);
  if (nb > 0) memcpy(hc.mirror(d_desc) + (size_t)cap * 32, desc_b, (size_t)nb * 32);
  const int32_t counts[2] = {na, nb};
  const int32_t* d_n = nullptr;
  RGBL_TRY(hc.put(&d_n, counts, 2));
  RGBL_TRY(hc.upload());
  int32_t* d_bi = hc.result<int32_t>(na);
  int32_t* d_bd = hc.result<int32_t>(na);
  int32_t* d_sd = hc.result<int32_t>(na);
  uint32_t* d_partial = hc.scratch<uint32_t>((size_t)splits * na * 2);
  // pair_a/pair_b == NULL selects the fixed pair (frame 0 -> frame 1)
  if (bf_on_matrix_cores(m)) {
    m->timer.begin(bf_on_fp4(m) ? "k_hamming_fp4" : "k_hamming_mfma", s);
    if (bf_on_fp4(m)) {
      hipLaunchKernelGGL(k_hamming_fp4, xcd_grid(false, qblocks, splits), dim3(256), 0, s, d_desc, d_n, cap, (const int32_t*)nullptr,
                         (const int32_t*)nullptr, d_bi, d_bd, d_sd, splits, d_partial);
      if (splits > 1) hipLaunchKernelGGL(k_hamming_merge, dim3((na + 255) / 256), dim3(256), 0, s, d_partial, na, splits, d_bi, d_bd, d_sd);
    } else {
      hipLaunchKernelGGL(k_hamming_mfma, xcd_grid(false, qblocks, 1), dim3(256), 0, s, d_desc, d_n, cap, (const int32_t*)nullptr,
                         (const int32_t*)nullptr, d_bi, d_bd, d_sd);
    }
  } else {
    m->timer.begin("k_hamming_bf", s);
    hipLaunchKernelGGL(k_hamming_bf, dim3((na + 63) / 64, 1), dim3(256), 0, s, d_desc, d_n, cap, (const int32_t*)nullptr,
                       (const int32_t*)nullptr, d_bi, d_bd, d_sd);
  }
  m->timer.end(s);
  RGBL_HIP(hipGetLastError());
  RGBL_TRY(hc.fetch());
  memcpy(best_idx, hc.host(d_bi), sizeof(int32_t) * na);
  memcpy(best_dist, hc.host(d_bd), sizeof(int32_t) * na);
  if (second_dist) memcpy(second_dist, hc.host(d_sd), sizeof(int32_t) * na);
  return RGBL_OK;
}

int rgbl_stereo_fisheye_matches(rgbl_matcher* m, const uint8_t* desc_left, int n_left, int mono_left, const uint8_t* desc_right,
                                int n_right, int mono_right, int32_t* left_to_right, int32_t* best_dist, int32_t* second_dist) {
  // Frame::ComputeStereoFishEyeMatches (/root/reference/src/Frame.cc:1256-1296) up to the triangulation: the lapping-area
  // subsets [mono, n) of both descriptor sets, cv::BFMatcher(NORM_HAMMING).knnMatch(k = 2) = best and second-best train
  // row of every query (strict '<': the lower index stays in front on ties), then Lowe's ratio
  // `matches.size() >= 2 && best < second * 0.7` (float times double, compared in double)
  if (!m || n_left < 0 || n_right < 0 || mono_left < 0 || mono_right < 0 || mono_left > n_left || mono_right > n_right ||
      (n_left > 0 && (!desc_left || !left_to_right)) || (n_right > 0 && !desc_right)) {
    set_error("invalid argument");
    return RGBL_ERR_INVALID;
  }
  for (int i = 0; i < n_left; ++i) {
    left_to_right[i] = -1;
    if (best_dist) best_dist[i] = 256;
    if (second_dist) second_dist[i] = 256;
  }
  const int nq = n_left - mono_left, nt = n_right - mono_right;
  if (nq == 0 || nt == 0) return RGBL_OK;
  std::vector<int32_t> bi(nq), bd(nq), sd(nq);
  RGBL_TRY(rgbl_hamming_bf(m, desc_left + (size_t)mono_left * 32, nq, desc_right + (size_t)mono_right * 32, nt, bi.data(), bd.data(),
                           sd.data()));
  for (int i = 0; i < nq; ++i) {
    if (best_dist) best_dist[mono_left + i] = bd[i];
    if (second_dist) second_dist[mono_left + i] = nt >= 2 ? sd[i] : 256;
    if (nt >= 2 && (double)(float)bd[i] < (double)(float)sd[i] * 0.7) left_to_right[mono_left + i] = bi[i] + mono_right;
  }
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
  StreamDrain drain(m->stream);  // error returns included
    const int nf1 = k1->node_off[k1->n_nodes], nf2 = k2->node_off[k2->n_nodes];
    size_t need = pad256((size_t)n1 * 32) + pad256((size_t)n2 * 32) + pad256((size_t)n1 * 8) + pad256((size_t)n2 * 8) +
                  pad256((size_t)n2 * 4) + pad256((size_t)n1 * 4) + pad256((size_t)n2 * 4) + pad256(n1) + pad256(n2) +
                  pad256((size_t)(k1->n_nodes + 1) * 4) + pad256((size_t)nf1 * 4) + pad256((size_t)(k2->n_nodes + 1) * 4) +
                  pad256((size_t)nf2 * 4) + 2 * pad256((size_t)npairs * 4) + 2 * pad256((size_t)prm->n_levels * 4) +
                  pad256((size_t)n1 * 4);
    HostCall hc;
    RGBL_TRY(hc.begin(m, need));
    hipStream_t s = hc.s;
    TriDev T;
    const rgbl_keyframe_view* kv[2] = {k1, k2};
    const uint8_t** desc[2] = {&T.desc1, &T.desc2};
    const float** xy[2] = {&T.xy1, &T.xy2};
    const float** ur[2] = {&T.ur1, &T.ur2};
    const int32_t **off[2] = {&T.off1, &T.off2}, **feat[2] = {&T.feat1, &T.feat2};
    const int32_t* oct1_unused = nullptr;
    const int32_t** oct[2] = {&oct1_unused, &T.oct2};
    for (int k = 0; k < 2; ++k) {
      const rgbl_keyframe_view* v = kv[k];
      const int nf = v->node_off[v->n_nodes];
      if (const rgbl_device_frame* f = v->device) {
        RGBL_TRY(hc.use(f, v->n));
        *desc[k] = f->d_desc; *xy[k] = f->d_xy; *oct[k] = f->d_oct; *ur[k] = f->d_ur;
      } else {
        RGBL_TRY(hc.put(desc[k], v->desc, (size_t)v->n * 32));
        RGBL_TRY(hc.put(xy[k], v->kp_xy, (size_t)v->n * 2));
        if (k == 1) RGBL_TRY(hc.put(oct[k], v->kp_octave, (size_t)v->n));
        RGBL_TRY(hc.put(ur[k], v->uright, (size_t)v->n));
      }
      if (v->device && v->device->n_nodes == v->n_nodes && v->device->nf == nf) {
        *off[k] = v->device->d_fv; *feat[k] = v->device->d_fv + v->n_nodes + 1;
      } else {
        RGBL_TRY(hc.put(off[k], v->node_off, (size_t)v->n_nodes + 1));
        RGBL_TRY(hc.put(feat[k], v->node_feat, (size_t)nf));
      }
    }
    RGBL_TRY(hc.put(&T.mp1, k1->has_mappoint, (size_t)n1));
    RGBL_TRY(hc.put(&T.mp2, k2->has_mappoint, (size_t)n2));
    RGBL_TRY(hc.put(&T.pair_n1, pa.data(), (size_t)npairs));
    RGBL_TRY(hc.put(&T.pair_n2, pb.data(), (size_t)npairs));
    RGBL_TRY(hc.put(&T.scale2, prm->scale_factors2, (size_t)prm->n_levels));
    RGBL_TRY(hc.put(&T.sigma2, prm->level_sigma2_2, (size_t)prm->n_levels));
    T.matches12 = hc.put_fill<int32_t>(n1, 0xff);   // all -1: travels with the upload
    hc.mark_result(T.matches12, n1);
    RGBL_TRY(hc.upload());
    memcpy(T.F, prm->F12, sizeof(T.F));
    T.ep[0] = prm->epipole[0];
    T.ep[1] = prm->epipole[1];
    T.only_stereo = prm->only_stereo;
    T.coarse = prm->coarse;
    m->timer.begin("k_search_triangulation", s);
    hipLaunchKernelGGL(k_search_triangulation, dim3(npairs), dim3(256), 0, s, T);
    m->timer.end(s);
    RGBL_HIP(hipGetLastError());
    RGBL_TRY(hc.fetch());
    memcpy(matches12, hc.host(T.matches12), sizeof(int32_t) * n1);
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

// rotation-consistency bins that survive: ORBmatcher::ComputeThreeMaxima (ORBmatcher.cc:2012-2053)
static void three_maxima(const std::vector<int>* hist, int L, int& i1, int& i2, int& i3) {
  int max1 = 0, max2 = 0, max3 = 0;
  i1 = i2 = i3 = -1;
  for (int i = 0; i < L; ++i) {
    const int sz = (int)hist[i].size();
    if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; i3 = i2; i2 = i1; i1 = i; }
    else if (sz > max2) { max3 = max2; max2 = sz; i3 = i2; i2 = i; }
    else if (sz > max3) { max3 = sz; i3 = i; }
  }
  if (max2 < 0.1f * (float)max1) { i2 = -1; i3 = -1; }
  else if (max3 < 0.1f * (float)max1) { i3 = -1; }
}

// what the two greedy, best-only projection matchers (Frame-to-Frame and key-frame-to-Frame) hand to the kernels
namespace {
struct ProjHost {
  int n1, n2, n_levels;
  const uint8_t *valid1, *obs1 /* nullptr: every point blocks */, *mpdesc1, *desc2, *blocked2 /* nullable */;
  const float *wpos1, *angle1, *xy2, *angle2, *ur2 /* nullable */, *scale_factors;
  const int32_t *oct1, *oct2;
  const float *grid, *Tcw_q, *Tcw_t, *K;
  float mbf, th;
  int forward, backward, skip_behind, max_dist, check_orientation;
  int want_ur2;   // with dev2: the overload tests the stereo coordinate (ur2 is then the resident array)
  int sim3_mode;  // 0, or 1 / 2 = the projection form of the SearchByProjection(pKF, Scw, ...) overloads (wpos1 = camera-frame points)
  const rgbl_device_frame* dev2;  // nullable: the second frame's xy / octave / uright / descriptors resident on the device
};

// the second frame's per-feature arrays: from the resident copy, or staged with the rest of the call's upload
int put_frame2(HostCall& hc, ProjDev& P, const rgbl_device_frame* dev2, int n2, const float* xy2, const int32_t* oct2, const float* ur2,
               bool want_ur, const uint8_t* desc2) {
  if (dev2) {
    RGBL_TRY(hc.use(dev2, n2));
    P.xy2 = dev2->d_xy; P.oct2 = dev2->d_oct; P.desc2 = dev2->d_desc;
    if (want_ur) P.ur2 = dev2->d_ur;
    return RGBL_OK;
  }

  RGBL_TRY(hc.put(&P.xy2, xy2, (size_t)n2 * 2));
  RGBL_TRY(hc.put(&P.oct2, oct2, (size_t)n2));
  if (want_ur) RGBL_TRY(hc.put(&P.ur2, ur2, (size_t)n2));
  RGBL_TRY(hc.put(&P.desc2, desc2, (size_t)n2 * 32));
  return RGBL_OK;
}

// Frame::AssignFeaturesToGrid for the call: the resident frame's own grid when it has one for these image bounds
// (rgbl_device_frame_set_grid), else k_proj_grid on the call's scratch.  P.grid, P.xy2, P.blocked2 must be set.
void grid_for_call(rgbl_matcher* m, hipStream_t s, ProjDev& P, const rgbl_device_frame* dev2) {
  if (dev2 && dev2->has_grid && memcmp(dev2->grid, P.grid, sizeof(P.grid)) == 0) {
    P.cell_start = dev2->d_cell_start;
    P.cell_items = dev2->d_cell_items;
    P.init_taken = 1;
    return;
  }
  P.init_taken = 0;
  m->timer.begin("k_proj_grid", s);
  launch_proj_grid(P, s);
  m->timer.end(s);
}

int projection_core(rgbl_matcher* m, const ProjHost& in, int32_t* match2, int* out_nmatches) {
  *out_nmatches = 0;
  const int n1 = in.n1, n2 = in.n2;
  for (int i = 0; i < n2; ++i) match2[i] = -1;
  if (n1 == 0 || n2 == 0) return RGBL_OK;
  if (n1 >= (1 << 20)) { set_error("at most 2^20 - 1 map points per call (a point's index travels in 20 bits of the resolve kernel's announcements)"); return RGBL_ERR_INVALID; }
  for (int i = 0; i < n1; ++i)
    if (in.valid1[i] && (in.oct1[i] < 0 || in.oct1[i] >= in.n_levels)) { set_error("octave out of range"); return RGBL_ERR_INVALID; }
  RGBL_HIP(hipSetDevice(m->device));
  StreamDrain drain(m->stream);  // error returns included
  size_t need = pad256(n1) * 2 + pad256((size_t)n1 * 12) + pad256((size_t)n1 * 32) + pad256((size_t)n1 * 4) + pad256((size_t)n2 * 8) +
                pad256((size_t)n2 * 4) * 2 + pad256((size_t)n2 * 32) + pad256((size_t)n2 * 2) + pad256((size_t)n2 * 4) * 3 + pad256(n2) +
                pad256((size_t)n1 * 16) * 2 + pad256(n1) * 2 + pad256((size_t)n1 * 4) + pad256((size_t)n1 * kProjCand * 4) + pad256((size_t)n1 * 4) + 256 +
                pad256((size_t)(kGridCells + 1) * 4);
  HostCall hc;
  RGBL_TRY(hc.begin(m, need));
  hipStream_t s = hc.s;
  ProjDev P;
  P.n1 = n1; P.n2 = n2;
  P.proj1 = nullptr; P.level1 = nullptr; P.viewcos1 = nullptr; P.blocked2 = nullptr; P.ur2 = nullptr; P.nnratio = 0.f;
  RGBL_TRY(hc.put(&P.valid1, in.valid1, (size_t)n1));
  if (in.obs1) RGBL_TRY(hc.put(&P.obs1, in.obs1, (size_t)n1));
  else P.obs1 = hc.put_fill<uint8_t>(n1, 1);   // every point blocks
  RGBL_TRY(hc.put(&P.wpos1, in.wpos1, (size_t)n1 * 3));
  RGBL_TRY(hc.put(&P.mpdesc1, in.mpdesc1, (size_t)n1 * 32));
  RGBL_TRY(hc.put(&P.oct1, in.oct1, (size_t)n1));
  RGBL_TRY(put_frame2(hc, P, in.dev2, n2, in.xy2, in.oct2, in.ur2, in.ur2 != nullptr || (in.dev2 && in.want_ur2), in.desc2));
  if (in.blocked2) RGBL_TRY(hc.put(&P.blocked2, in.blocked2, (size_t)n2));
  RGBL_TRY(hc.upload());
  P.choice = hc.result<int32_t>(n1);
  P.cell_start = hc.scratch<uint32_t>(kGridCells + 1);
  P.cell_items = hc.scratch<uint16_t>(n2);
  P.taken_by = hc.scratch<int32_t>(n2);
  P.min_unres = hc.scratch<int32_t>(n2);
  P.win = hc.scratch<float4>(n1);
  P.rng = hc.scratch<int4>(n1);
  P.clist = hc.scratch<uint32_t>((size_t)n1 * kProjCand);
  P.cref = hc.scratch<uint32_t>(n1);
  P.state = hc.scratch<uint8_t>(n1);
  memcpy(P.grid, in.grid, sizeof(P.grid));
  memcpy(P.q, in.Tcw_q, sizeof(P.q));
  memcpy(P.t, in.Tcw_t, sizeof(P.t));
  memcpy(P.K, in.K, sizeof(P.K));
  P.mbf = in.mbf;
  P.th = in.th;
  for (int l = 0; l < kProjMaxLevels; ++l) P.scale[l] = l < in.n_levels ? in.scale_factors[l] : 1.f;
  P.forward = in.forward; P.backward = in.backward;
  P.skip_behind = in.skip_behind; P.max_dist = in.max_dist;
  P.sim3_mode = in.sim3_mode;
  grid_for_call(m, s, P, in.dev2);
  m->timer.begin("k_proj_candidates", s);
  hipLaunchKernelGGL(k_proj_candidates, dim3((n1 + kPointsPerBlock - 1) / kPointsPerBlock), dim3(256), 0, s, P);
  m->timer.end(s);
  m->timer.begin("k_proj_resolve", s);
  if (n2 <= kResolveLdsN2 && n1 <= kResolveLdsN1) hipLaunchKernelGGL(k_proj_resolve<true>, dim3(1), dim3(kResolveBS), 0, s, P);
  else hipLaunchKernelGGL(k_proj_resolve<false>, dim3(1), dim3(kResolveBS), 0, s, P);
  m->timer.end(s);
  RGBL_HIP(hipGetLastError());
  RGBL_TRY(hc.fetch());
  const int32_t* choice = hc.host(P.choice);
  // what the loop leaves in CurrentFrame.mvpMapPoints (a later point overwrites an unobserved earlier one), the match
  // count, and the rotation-consistency pass (ORBmatcher.cc:1768-1790, 1860-1884 / 1961-2006)
  int nmatches = 0;
  std::vector<int> hist[30];
  const float factor = 1.0f / 30;
  for (int i = 0; i < n1; ++i) {
    const int c = choice[i];
    if (c < 0) continue;
    match2[c] = i;
    ++nmatches;
    if (in.check_orientation) {
      float rot = in.angle1[i] - in.angle2[c];
      if (rot < 0.0) rot += 360.0f;
      int bin = (int)roundf(rot * factor);
      if (bin == 30) bin = 0;
      if (bin >= 0 && bin < 30) hist[bin].push_back(c);
    }
  }
  if (in.check_orientation) {
    int i1, i2, i3;
    three_maxima(hist, 30, i1, i2, i3);
    for (int i = 0; i < 30; ++i) {
      if (i == i1 || i == i2 || i == i3) continue;
      for (int c : hist[i]) { match2[c] = -1; --nmatches; }  // a feature chosen twice is un-counted twice, as in the reference
    }
  }
  *out_nmatches = nmatches;
  return RGBL_OK;
}
}  // namespace

int rgbl_search_by_projection(rgbl_matcher* m, const rgbl_projection_input* in, int32_t* match2, int* out_nmatches) {
  if (!m || !in || !match2 || !out_nmatches || in->n1 < 0 || in->n2 < 0 || in->n2 > 65535 || in->n_levels < 1 ||
      in->n_levels > kProjMaxLevels) {
    set_error("invalid argument (CurrentFrame may hold at most 65535 features, %d pyramid levels)", kProjMaxLevels);
    return RGBL_ERR_INVALID;
  }
  ProjHost h{};
  h.n1 = in->n1; h.n2 = in->n2; h.n_levels = in->n_levels;
  h.valid1 = in->valid1; h.obs1 = in->mp_observed1; h.mpdesc1 = in->mp_desc1; h.desc2 = in->desc2; h.blocked2 = nullptr;
  h.wpos1 = in->world_pos1; h.angle1 = in->angle1; h.xy2 = in->kp2_xy; h.angle2 = in->kp2_angle; h.ur2 = in->uright2;
  h.scale_factors = in->scale_factors; h.oct1 = in->octave1; h.oct2 = in->kp2_octave;
  h.grid = in->grid; h.Tcw_q = in->Tcw_q; h.Tcw_t = in->Tcw_t; h.K = in->K;
  h.mbf = in->mbf; h.th = in->th;
  {
    // bForward / bBackward (ORBmatcher.cc:1686-1694): tlc = Tlw * Tcw.inverse().translation(), Sophus / Eigen arithmetic
    const float qi[4] = {-in->Tcw_q[0], -in->Tcw_q[1], -in->Tcw_q[2], in->Tcw_q[3]};
    float wx, wy, wz, lx, ly, lz;
    quat_rotate(qi, -in->Tcw_t[0], -in->Tcw_t[1], -in->Tcw_t[2], &wx, &wy, &wz);
    quat_rotate(in->Tlw_q, wx, wy, wz, &lx, &ly, &lz);
    lz += in->Tlw_t[2];
    (void)lx; (void)ly;
    h.forward = (lz > in->mb && !in->mono) ? 1 : 0;
    h.backward = (-lz > in->mb && !in->mono) ? 1 : 0;
  }
  h.skip_behind = 1;
  h.max_dist = 100;  // TH_HIGH
  h.check_orientation = in->check_orientation;
  h.dev2 = in->device2; h.want_ur2 = 1;
  return projection_core(m, h, match2, out_nmatches);
}

// int ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, th, ORBdist)
// (src/ORBmatcher.cc:1889-2010): the same greedy best-only search; every matched feature is occupied for the points after it
// (CurrentFrame.mvpMapPoints[i2] != NULL, :1949-1950), as are the features that hold a map point on entry; levels
// [predicted - 1, predicted + 1], radius th * scale[predicted], no stereo-coordinate test and no test of the sign of the depth.
int rgbl_search_by_projection_keyframe(rgbl_matcher* m, const rgbl_keyframe_projection_input* in, int32_t* match2,
                                       int* out_nmatches) {
  if (!m || !in || !match2 || !out_nmatches || in->n1 < 0 || in->n2 < 0 || in->n2 > 65535 || in->n_levels < 1 ||
      in->n_levels > kProjMaxLevels || in->orb_dist < 0 || in->orb_dist > 255) {
    set_error("invalid argument (CurrentFrame may hold at most 65535 features, %d pyramid levels, ORBdist 0..255)", kProjMaxLevels);
    return RGBL_ERR_INVALID;
  }
  ProjHost h{};
  h.n1 = in->n1; h.n2 = in->n2; h.n_levels = in->n_levels;
  h.valid1 = in->valid1; h.obs1 = nullptr; h.mpdesc1 = in->mp_desc1; h.desc2 = in->desc2; h.blocked2 = in->occupied2;
  h.wpos1 = in->world_pos1; h.angle1 = in->angle1; h.xy2 = in->kp2_xy; h.angle2 = in->kp2_angle; h.ur2 = nullptr;
  h.scale_factors = in->scale_factors; h.oct1 = in->level1; h.oct2 = in->kp2_octave;
  h.grid = in->grid; h.Tcw_q = in->Tcw_q; h.Tcw_t = in->Tcw_t; h.K = in->K;
  h.mbf = 0.f; h.th = in->th;
  h.forward = 0; h.backward = 0;  // levels predicted - 1 ... predicted + 1
  h.skip_behind = 0;
  h.max_dist = in->orb_dist;
  h.check_orientation = in->check_orientation;
  h.dev2 = in->device2; h.want_ur2 = 0;
  return projection_core(m, h, match2, out_nmatches);
}

int rgbl_distinctive_descriptors(rgbl_matcher* m, const uint8_t* desc, const int32_t* off, int n_points, int32_t* best) {
  if (!m || !off || !best || n_points < 0) { set_error("null argument"); return RGBL_ERR_INVALID; }
  if (n_points == 0) return RGBL_OK;
  const int total = off[n_points];
  for (int p = 0; p < n_points; ++p)
    if (off[p + 1] < off[p] || off[p + 1] - off[p] > 65535) { set_error("offsets must ascend, at most 65535 observations per point"); return RGBL_ERR_INVALID; }
  if (off[0] != 0 || (total > 0 && !desc)) { set_error("offsets start at 0; descriptors missing"); return RGBL_ERR_INVALID; }
  RGBL_HIP(hipSetDevice(m->device));
  StreamDrain drain(m->stream);  // error returns included
  HostCall hc;
  RGBL_TRY(hc.begin(m, pad256((size_t)total * 32) + pad256((size_t)(n_points + 1) * 4) + pad256((size_t)n_points * 4)));
  hipStream_t s = hc.s;
  const uint8_t* d_desc = nullptr;
  const int32_t* d_off = nullptr;
  if (total > 0) RGBL_TRY(hc.put(&d_desc, desc, (size_t)total * 32));
  RGBL_TRY(hc.put(&d_off, off, (size_t)n_points + 1));
  RGBL_TRY(hc.upload());
  int32_t* d_best = hc.result<int32_t>(n_points);
  m->timer.begin("k_distinctive", s);
  hipLaunchKernelGGL(k_distinctive, dim3(n_points), dim3(64), 0, s, d_desc, d_off, d_best);
  m->timer.end(s);
  RGBL_HIP(hipGetLastError());
  RGBL_TRY(hc.fetch());
  memcpy(best, hc.host(d_best), sizeof(int32_t) * n_points);
  return RGBL_OK;
}

static int fuse_core(rgbl_matcher* m, const rgbl_fuse_input* in, int cam_frame, int proj_form, int chi2_gate, int max_dist,
                     int want_ur2, int32_t* best_idx, int32_t* best_dist) {
  if (!m || !in || !best_idx || in->n1 < 0 || in->n2 < 0 || in->n2 > 65535 || in->n_levels < 1 || in->n_levels > kProjMaxLevels) {
    set_error("invalid argument (the key frame may hold at most 65535 features, %d pyramid levels)", kProjMaxLevels);
    return RGBL_ERR_INVALID;
  }
  const int n1 = in->n1, n2 = in->n2;
  for (int i = 0; i < n1; ++i) { best_idx[i] = -1; if (best_dist) best_dist[i] = 256; }
  if (n1 == 0 || n2 == 0) return RGBL_OK;
  for (int i = 0; i < n1; ++i)
    if (in->valid1[i] && (in->level1[i] < 0 || in->level1[i] >= in->n_levels)) { set_error("predicted level out of range"); return RGBL_ERR_INVALID; }
  RGBL_HIP(hipSetDevice(m->device));
  StreamDrain drain(m->stream);  // error returns included
  size_t need = pad256(n1) + pad256((size_t)n1 * 12) + pad256((size_t)n1 * 32) + pad256((size_t)n1 * 4) + pad256((size_t)n2 * 8) +
                pad256((size_t)n2 * 4) * 2 + pad256((size_t)n2 * 32) + pad256((size_t)n2 * 2) + pad256((size_t)n2 * 4) * 2 +
                pad256((size_t)n1 * 8) + pad256((size_t)(kGridCells + 1) * 4);
  HostCall hc;
  RGBL_TRY(hc.begin(m, need));
  hipStream_t s = hc.s;
  ProjDev P{};
  P.n1 = n1; P.n2 = n2;
  RGBL_TRY(hc.put(&P.valid1, in->valid1, (size_t)n1));
  RGBL_TRY(hc.put(&P.wpos1, in->world_pos1, (size_t)n1 * 3));
  RGBL_TRY(hc.put(&P.mpdesc1, in->mp_desc1, (size_t)n1 * 32));
  RGBL_TRY(hc.put(&P.oct1, in->level1, (size_t)n1));
  RGBL_TRY(put_frame2(hc, P, in->device2, n2, in->kp2_xy, in->kp2_octave, in->uright2, want_ur2 != 0, in->desc2));
  RGBL_TRY(hc.upload());
  P.cand = hc.result<unsigned long long>(n1);
  P.cell_start = hc.scratch<uint32_t>(kGridCells + 1);
  P.cell_items = hc.scratch<uint16_t>(n2);
  P.taken_by = hc.scratch<int32_t>(n2);  // written by k_proj_grid, not used by the fuse search
  memcpy(P.grid, in->grid, sizeof(P.grid));
  memcpy(P.q, in->Tcw_q, sizeof(P.q));
  memcpy(P.t, in->Tcw_t, sizeof(P.t));
  memcpy(P.K, in->K, sizeof(P.K));
  P.mbf = in->bf;
  P.th = in->th;
  FuseDev Fz;
  Fz.cam_frame = cam_frame; Fz.proj_form = proj_form; Fz.chi2_gate = chi2_gate;
  for (int l = 0; l < kProjMaxLevels; ++l) {
    P.scale[l] = l < in->n_levels ? in->scale_factors[l] : 1.f;
    Fz.inv_sigma2[l] = l < in->n_levels ? in->inv_level_sigma2[l] : 1.f;
  }
  grid_for_call(m, s, P, in->device2);
  m->timer.begin("k_fuse_search", s);
  hipLaunchKernelGGL(k_fuse_search, dim3((n1 + kPointsPerBlock - 1) / kPointsPerBlock), dim3(256), 0, s, P, Fz);
  m->timer.end(s);
  RGBL_HIP(hipGetLastError());
  RGBL_TRY(hc.fetch());
  const unsigned long long* keys = hc.host(P.cand);
  for (int i = 0; i < n1; ++i) {
    if (keys[i] == ~0ull) continue;
    const int dist = (int)(keys[i] >> 32);
    if (best_dist) best_dist[i] = dist;
    if (dist <= max_dist) best_idx[i] = (int)(keys[i] & 0xffffu);
  }
  return RGBL_OK;
}

// int ORBmatcher::SearchByProjection(KeyFrame* pKF, Sophus::Sim3f& Scw, const vector<MapPoint*>& vpPoints, vector<MapPoint*>&
// vpMatched, int th, float ratioHamming) (src/ORBmatcher.cc:427-532) and its overload with vpPointsKFs / vpMatchedKF (:534-646):
// the greedy best-only search again - a matched feature (vpMatched[idx] != NULL, on entry or set by an earlier point) is
// skipped by the points after it - on camera-frame points.
int rgbl_search_by_projection_sim3(rgbl_matcher* m, const rgbl_project_search_input* in, const uint8_t* matched2, int32_t* match2,
                                   int* out_nmatches) {
  if (!m || !in || !match2 || !out_nmatches || in->n1 < 0 || in->n2 < 0 || in->n2 > 65535 || in->n_levels < 1 ||
      in->n_levels > kProjMaxLevels || (in->proj_form != 0 && in->proj_form != 2) || in->max_dist < 0 || in->max_dist > 255) {
    set_error("invalid argument (at most 65535 features, %d pyramid levels, proj_form 0 / 2, max_dist 0..255)", kProjMaxLevels);
    return RGBL_ERR_INVALID;
  }
  const float ident_q[4] = {0.f, 0.f, 0.f, 1.f}, zero_t[3] = {0.f, 0.f, 0.f};
  ProjHost h{};
  h.n1 = in->n1; h.n2 = in->n2; h.n_levels = in->n_levels;
  h.valid1 = in->valid1; h.obs1 = nullptr; h.mpdesc1 = in->mp_desc1; h.desc2 = in->desc2; h.blocked2 = matched2;
  h.wpos1 = in->cam_pos1; h.angle1 = nullptr; h.xy2 = in->kp2_xy; h.angle2 = nullptr; h.ur2 = nullptr;
  h.scale_factors = in->scale_factors; h.oct1 = in->level1; h.oct2 = in->kp2_octave;
  h.grid = in->grid; h.Tcw_q = ident_q; h.Tcw_t = zero_t; h.K = in->K;
  h.mbf = 0.f; h.th = in->th;
  h.forward = 0; h.backward = 0; h.skip_behind = 0;
  h.max_dist = in->max_dist;
  h.check_orientation = 0;
  h.sim3_mode = in->proj_form == 0 ? 1 : 2;
  h.dev2 = in->device2; h.want_ur2 = 0;
  return projection_core(m, h, match2, out_nmatches);
}

int rgbl_fuse_search(rgbl_matcher* m, const rgbl_fuse_input* in, int32_t* best_idx, int32_t* best_dist) {
  return fuse_core(m, in, 0, 0, 1, 50 /* TH_LOW */, (in && (in->uright2 || in->device2)) ? 1 : 0, best_idx, best_dist);
}

// The per-point search shared by ORBmatcher::Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) (src/ORBmatcher.cc:1340-1455:
// Pinhole::project, best <= TH_LOW) and by both directions of SearchBySim3 (:1457-1674: invz = 1.0 / z, best <= TH_HIGH): the
// caller has already moved the points into the key frame's camera frame with its SE3 / Sim3 objects.
int rgbl_project_search(rgbl_matcher* m, const rgbl_project_search_input* in, int32_t* best_idx, int32_t* best_dist) {
  if (!in || (in->proj_form != 0 && in->proj_form != 1) || in->max_dist < 0 || in->max_dist > 255) {
    set_error("invalid argument (proj_form 0 / 1, max_dist 0..255)");
    return RGBL_ERR_INVALID;
  }
  rgbl_fuse_input f{};
  f.n1 = in->n1; f.valid1 = in->valid1; f.world_pos1 = in->cam_pos1; f.mp_desc1 = in->mp_desc1; f.level1 = in->level1;
  f.n2 = in->n2; f.kp2_xy = in->kp2_xy; f.kp2_octave = in->kp2_octave; f.uright2 = nullptr; f.desc2 = in->desc2;
  memcpy(f.grid, in->grid, sizeof(f.grid));
  f.Tcw_q[3] = 1.f;
  memcpy(f.K, in->K, sizeof(f.K));
  f.bf = 0.f;
  f.scale_factors = in->scale_factors;
  f.inv_level_sigma2 = in->scale_factors;  // not read without the chi-square gate
  f.n_levels = in->n_levels;
  f.th = in->th;
  f.device2 = in->device2;
  return fuse_core(m, &f, 1, in->proj_form, 0, in->max_dist, 0, best_idx, best_dist);
}

int rgbl_search_local_points(rgbl_matcher* m, const rgbl_local_points_input* in, int32_t* match2, int* out_nmatches) {
  if (!m || !in || !match2 || !out_nmatches || in->n1 < 0 || in->n2 < 0 || in->n2 > 65535 || in->n_levels < 1 ||
      in->n_levels > kProjMaxLevels) {
    set_error("invalid argument (the frame may hold at most 65535 features, %d pyramid levels)", kProjMaxLevels);
    return RGBL_ERR_INVALID;
  }
  if (in->n1 >= (1 << 20)) { set_error("at most 2^20 - 1 map points per call (a point's index travels in 20 bits of the resolve kernel's announcements)"); return RGBL_ERR_INVALID; }
  *out_nmatches = 0;
  const int n1 = in->n1, n2 = in->n2;
  for (int i = 0; i < n2; ++i) match2[i] = -1;
  if (n1 == 0 || n2 == 0) return RGBL_OK;
  for (int i = 0; i < n1; ++i)
    if (in->valid1[i] && (in->level1[i] < 0 || in->level1[i] >= in->n_levels)) { set_error("predicted level out of range"); return RGBL_ERR_INVALID; }
  RGBL_HIP(hipSetDevice(m->device));
  StreamDrain drain(m->stream);  // error returns included
  size_t need = pad256(n1) * 2 + pad256((size_t)n1 * 12) + pad256((size_t)n1 * 32) + pad256((size_t)n1 * 4) * 2 + pad256((size_t)n2 * 8) +
                pad256((size_t)n2 * 4) * 2 + pad256((size_t)n2 * 32) + pad256(n2) + pad256((size_t)n2 * 2) + pad256((size_t)n2 * 4) * 2 +
                pad256((size_t)n1 * 16) * 2 + pad256(n1) * 2 + pad256((size_t)n1 * 4) + pad256((size_t)n1 * kProjCand * 4) + pad256((size_t)n1 * 4) + 256 +
                pad256((size_t)(kGridCells + 1) * 4);
  HostCall hc;
  RGBL_TRY(hc.begin(m, need));
  hipStream_t s = hc.s;
  ProjDev P;
  memset(&P, 0, sizeof(P));
  P.n1 = n1; P.n2 = n2;
  RGBL_TRY(hc.put(&P.valid1, in->valid1, (size_t)n1));
  RGBL_TRY(hc.put(&P.obs1, in->mp_observed1, (size_t)n1));
  RGBL_TRY(hc.put(&P.proj1, in->proj1, (size_t)n1 * 3));
  RGBL_TRY(hc.put(&P.mpdesc1, in->mp_desc1, (size_t)n1 * 32));
  RGBL_TRY(hc.put(&P.level1, in->level1, (size_t)n1));
  RGBL_TRY(hc.put(&P.viewcos1, in->view_cos1, (size_t)n1));
  RGBL_TRY(put_frame2(hc, P, in->device2, n2, in->kp2_xy, in->kp2_octave, in->uright2, true, in->desc2));
  if (in->blocked2) RGBL_TRY(hc.put(&P.blocked2, in->blocked2, (size_t)n2));
  RGBL_TRY(hc.upload());
  P.choice = hc.result<int32_t>(n1);
  P.cell_start = hc.scratch<uint32_t>(kGridCells + 1);
  P.cell_items = hc.scratch<uint16_t>(n2);
  P.taken_by = hc.scratch<int32_t>(n2);
  P.min_unres = hc.scratch<int32_t>(n2);
  P.win = hc.scratch<float4>(n1);
  P.rng = hc.scratch<int4>(n1);
  P.clist = hc.scratch<uint32_t>((size_t)n1 * kProjCand);
  P.cref = hc.scratch<uint32_t>(n1);
  P.state = hc.scratch<uint8_t>(n1);
  memcpy(P.grid, in->grid, sizeof(P.grid));
  P.th = in->th;
  P.nnratio = in->nnratio;
  for (int l = 0; l < kProjMaxLevels; ++l) P.scale[l] = l < in->n_levels ? in->scale_factors[l] : 1.f;
  grid_for_call(m, s, P, in->device2);
  m->timer.begin("k_local_candidates", s);
  hipLaunchKernelGGL(k_local_candidates, dim3((n1 + kPointsPerBlock - 1) / kPointsPerBlock), dim3(256), 0, s, P);
  m->timer.end(s);
  m->timer.begin("k_local_resolve", s);
  if (n2 <= kResolveLdsN2 && n1 <= kResolveLdsN1) hipLaunchKernelGGL(k_local_resolve<true>, dim3(1), dim3(kResolveBS), 0, s, P);
  else hipLaunchKernelGGL(k_local_resolve<false>, dim3(1), dim3(kResolveBS), 0, s, P);
  m->timer.end(s);
  RGBL_HIP(hipGetLastError());
  RGBL_TRY(hc.fetch());
  const int32_t* choice = hc.host(P.choice);
  int nmatches = 0;
  for (int i = 0; i < n1; ++i)
    if (choice[i] >= 0) { match2[choice[i]] = i; ++nmatches; }  // a later point overwrites an unobserved earlier one
  *out_nmatches = nmatches;
  return RGBL_OK;
}

int rgbl_search_for_initialization(rgbl_matcher* m, const rgbl_initialization_input* in, float* prev_matched, int32_t* matches12,
                                   int* out_nmatches) {
  if (!m || !in || !out_nmatches || in->n1 < 0 || in->n1 >= (1 << 20) || in->n2 < 0 || in->n2 > 65535 || (in->n1 > 0 && (!prev_matched || !matches12))) {
    set_error("invalid argument (the second frame may hold at most 65535 features, the first 2^20 - 1)");
    return RGBL_ERR_INVALID;
  }
  *out_nmatches = 0;
  const int n1 = in->n1, n2 = in->n2;
  for (int i = 0; i < n1; ++i) matches12[i] = -1;
  if (n1 == 0 || n2 == 0) return RGBL_OK;
  for (int i = 0; i < n1; ++i)
    if (in->kp1_octave[i] < 0) { set_error("negative octave"); return RGBL_ERR_INVALID; }
  RGBL_HIP(hipSetDevice(m->device));
  StreamDrain drain(m->stream);  // error returns included
  size_t need = pad256((size_t)n1 * 8) + pad256((size_t)n1 * 32) + pad256((size_t)n1 * 4) + pad256((size_t)n2 * 8) + pad256((size_t)n2 * 4) +
                pad256((size_t)n2 * 32) + pad256((size_t)n2 * 2) + pad256((size_t)n2 * 4) * 3 + pad256((size_t)n1 * 16) * 2 + pad256(n1) * 2 +
                pad256((size_t)n1 * 4) + pad256((size_t)n1 * kProjCand * 4) + pad256((size_t)n1 * 4) + 256 + pad256((size_t)(kGridCells + 1) * 4);
  HostCall hc;
  RGBL_TRY(hc.begin(m, need));
  hipStream_t s = hc.s;
  ProjDev P;
  memset(&P, 0, sizeof(P));
  P.n1 = n1; P.n2 = n2;
  RGBL_TRY(hc.put(&P.proj1, (const float*)prev_matched, (size_t)n1 * 2));
  RGBL_TRY(hc.put(&P.mpdesc1, in->desc1, (size_t)n1 * 32));
  RGBL_TRY(hc.put(&P.oct1, in->kp1_octave, (size_t)n1));
  RGBL_TRY(put_frame2(hc, P, nullptr, n2, in->kp2_xy, in->kp2_octave, nullptr, false, in->desc2));
  RGBL_TRY(hc.upload());
  P.choice = hc.result<int32_t>(n1);
  P.owner = hc.result<int32_t>(n2);
  P.cell_start = hc.scratch<uint32_t>(kGridCells + 1);
  P.cell_items = hc.scratch<uint16_t>(n2);
  P.taken_by = hc.scratch<int32_t>(n2);
  P.min_unres = hc.scratch<int32_t>(n2);
  P.win = hc.scratch<float4>(n1);
  P.rng = hc.scratch<int4>(n1);
  P.clist = hc.scratch<uint32_t>((size_t)n1 * kProjCand);
  P.cref = hc.scratch<uint32_t>(n1);
  P.state = hc.scratch<uint8_t>(n1);
  memcpy(P.grid, in->grid, sizeof(P.grid));
  P.th = (float)in->window_size;  // GetFeaturesInArea takes r as const float&
  P.nnratio = in->nnratio;
  // candidates at or above v cannot change a decision: as best they fail TH_LOW, as second-best they pass the ratio test
  // for every best <= TH_LOW
  int v = 51;
  while (v <= 256 && !((float)v * in->nnratio > 50.0f)) ++v;
  P.max_dist = v - 1;
  grid_for_call(m, s, P, nullptr);
  m->timer.begin("k_init_candidates", s);
  hipLaunchKernelGGL(k_init_candidates, dim3((n1 + kPointsPerBlock - 1) / kPointsPerBlock), dim3(256), 0, s, P);
  m->timer.end(s);
  m->timer.begin("k_init_resolve", s);
  if (n2 <= kResolveLdsN2 && n1 <= kResolveLdsN1) hipLaunchKernelGGL(k_init_resolve<true>, dim3(1), dim3(kResolveBS), 0, s, P);
  else hipLaunchKernelGGL(k_init_resolve<false>, dim3(1), dim3(kResolveBS), 0, s, P);
  m->timer.end(s);
  RGBL_HIP(hipGetLastError());
  RGBL_TRY(hc.fetch());
  const int32_t *choice = hc.host(P.choice), *owner = hc.host(P.owner);
  // a feature whose match was taken over later stays in the rotation histogram (pushed at match time, :720-730) but
  // holds no match any more (:708-712); nmatches always equals the number of entries >= 0
  std::vector<int> hist[30];
  const float factor = 1.0f / 30;
  for (int i = 0; i < n1; ++i) {
    const int c = choice[i];
    if (c < 0) continue;
    if (owner[c] == i) matches12[i] = c;
    if (in->check_orientation) {
      float rot = in->kp1_angle[i] - in->kp2_angle[c];
      if (rot < 0.0) rot += 360.0f;
      int bin = (int)roundf(rot * factor);
      if (bin == 30) bin = 0;
      if (bin >= 0 && bin < 30) hist[bin].push_back(i);
    }
  }
  if (in->check_orientation) {
    int i1, i2, i3;
    three_maxima(hist, 30, i1, i2, i3);
    for (int b = 0; b < 30; ++b) {
      if (b == i1 || b == i2 || b == i3) continue;
      for (int i : hist[b]) matches12[i] = -1;
    }
  }
  int nmatches = 0;
  for (int i = 0; i < n1; ++i)
    if (matches12[i] >= 0) {  // "Update prev matched" (:757-760)
      ++nmatches;
      prev_matched[2 * i] = in->kp2_xy[2 * matches12[i]];
      prev_matched[2 * i + 1] = in->kp2_xy[2 * matches12[i] + 1];
    }
  *out_nmatches = nmatches;
  return RGBL_OK;
}

// ---- vocabulary handle -----------------------------------------------------------------------------------------------
struct rgbl_vocabulary {
  int device = 0, k = 0, L = 0, n_nodes = 0, n_words = 0;
  VocDev dev{};
  std::vector<void*> allocs;
  hipStream_t stream = nullptr;
  // staging for the host entry point (grown on demand): ONE device block desc [cap x 32] | weight [cap] | word [cap] | node [cap]
  // and its page-locked mirror - one request up, one back
  uint8_t* d_stage = nullptr; uint8_t* h_stage = nullptr; int stage_cap = 0;
};

static int voc_upload(rgbl_vocabulary* v, int n_nodes, int L, const int32_t* child_off, const int32_t* child, const uint8_t* desc,
                      const double* weight, const int32_t* word_id) {
  auto up = [&](const void* src, size_t bytes, const void** dst) -> int {
    void* p = nullptr;
    RGBL_HIP(hipMalloc(&p, std::max<size_t>(bytes, 16)));
    v->allocs.push_back(p);
    RGBL_HIP(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
    *dst = p;
    return RGBL_OK;
  };
  v->n_nodes = n_nodes; v->L = L;
  v->dev.n_nodes = n_nodes; v->dev.L = L;
  RGBL_TRY(up(child_off, sizeof(int32_t) * ((size_t)n_nodes + 1), (const void**)&v->dev.child_off));
  RGBL_TRY(up(child, sizeof(int32_t) * (size_t)std::max(n_nodes - 1, 1), (const void**)&v->dev.child));
  RGBL_TRY(up(desc, (size_t)n_nodes * 32, (const void**)&v->dev.desc));
  RGBL_TRY(up(weight, sizeof(double) * (size_t)n_nodes, (const void**)&v->dev.weight));
  RGBL_TRY(up(word_id, sizeof(int32_t) * (size_t)n_nodes, (const void**)&v->dev.word_id));
  RGBL_HIP(hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking));
  return RGBL_OK;
}

static int voc_new(int device, rgbl_vocabulary** out) {
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { set_error("no HIP device: librgbl_frontend has no CPU fallback"); return RGBL_ERR_NO_DEVICE; }
  if (device < 0 || device >= ndev) { set_error("device %d out of range", device); return RGBL_ERR_INVALID; }
  RGBL_HIP(hipSetDevice(device));
  rgbl_vocabulary* v = new rgbl_vocabulary;
  v->device = device;
  *out = v;
  return RGBL_OK;
}

void rgbl_vocabulary_destroy(rgbl_vocabulary* v) {
  if (!v) return;
  (void)hipSetDevice(v->device);
  for (void* p : v->allocs) (void)hipFree(p);
  if (v->d_stage) (void)hipFree(v->d_stage);
  if (v->h_stage) (void)hipHostFree(v->h_stage);
  if (v->stream) (void)hipStreamDestroy(v->stream);
  delete v;
}

int rgbl_vocabulary_create(int n_nodes, int L, const int32_t* child_off, const int32_t* child, const uint8_t* desc,
                           const double* weight, const int32_t* word_id, int device, rgbl_vocabulary** out) {
  if (!out) { set_error("null argument"); return RGBL_ERR_INVALID; }
  *out = nullptr;
  if (n_nodes < 2 || L < 1 || !child_off || !child || !desc || !weight || !word_id || child_off[0] != 0 ||
      child_off[n_nodes] != n_nodes - 1 || child_off[1] == 0) {
    set_error("vocabulary: the tree needs a root with children and n_nodes - 1 child entries");
    return RGBL_ERR_INVALID;
  }
  rgbl_vocabulary* v = nullptr;
  RGBL_TRY(voc_new(device, &v));
  int nw = 0, kmax = 0;
  for (int i = 0; i < n_nodes; ++i) {
    kmax = std::max(kmax, child_off[i + 1] - child_off[i]);
    nw += child_off[i + 1] == child_off[i];
  }
  v->k = kmax; v->n_words = nw;
  const int rc = voc_upload(v, n_nodes, L, child_off, child, desc, weight, word_id);
  if (rc != RGBL_OK) { rgbl_vocabulary_destroy(v); return rc; }
  *out = v;
  return RGBL_OK;
}

// ORBVocabulary::loadFromTextFile (TemplatedVocabulary.h:1338-1425): "k L scoring weighting", then one node per line
// "parent is_leaf d0 .. d31 weight", node ids in file order starting at 1, word ids handed out to the leaves in file order.
// Only L1_NORM / TF_IDF files (ORBvoc.txt: "10 6 0 0") are accepted.  A trailing empty line is ignored here (the reference
// turns it into an extra child of the root with an unset descriptor).
int rgbl_vocabulary_load_text(const char* path, int device, rgbl_vocabulary** out) {
  if (!out || !path) { set_error("null argument"); return RGBL_ERR_INVALID; }
  *out = nullptr;
  FILE* f = fopen(path, "r");
  if (!f) { set_error("cannot open vocabulary file %s", path); return RGBL_ERR_INVALID; }
  int k = 0, L = 0, n1 = -1, n2 = -1;
  if (fscanf(f, "%d %d %d %d", &k, &L, &n1, &n2) != 4 || k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) {
    fclose(f);
    set_error("Vocabulary loading failure: This is not a correct text file!");
    return RGBL_ERR_INVALID;
  }
  if (n1 != 0 || n2 != 0) { fclose(f); set_error("only L1_NORM / TF_IDF vocabularies (scoring 0, weighting 0) are supported"); return RGBL_ERR_INVALID; }
  std::vector<int32_t> parent(1, 0), word(1, -1);
  std::vector<uint8_t> desc(32, 0), leaf(1, 0);
  std::vector<double> weight(1, 0.0);
  int nw = 0;
  for (;;) {
    int pid, is_leaf;
    if (fscanf(f, "%d %d", &pid, &is_leaf) != 2) break;
    const int nid = (int)parent.size();
    if (pid < 0 || pid >= nid) { fclose(f); set_error("vocabulary: node %d names parent %d", nid, pid); return RGBL_ERR_INVALID; }
    uint8_t d[32];
    for (int j = 0; j < 32; ++j) { int b; if (fscanf(f, "%d", &b) != 1) { fclose(f); set_error("vocabulary: truncated node %d", nid); return RGBL_ERR_INVALID; } d[j] = (uint8_t)b; }
    double w;
    if (fscanf(f, "%lf", &w) != 1) { fclose(f); set_error("vocabulary: truncated node %d", nid); return RGBL_ERR_INVALID; }
    parent.push_back(pid); leaf.push_back(is_leaf > 0); weight.push_back(w);
    desc.insert(desc.end(), d, d + 32);
    word.push_back(is_leaf > 0 ? nw++ : -1);
  }
  fclose(f);
  const int n = (int)parent.size();
  if (n < 2) { set_error("vocabulary: no nodes"); return RGBL_ERR_INVALID; }
  std::vector<int32_t> off(n + 1, 0), child(n - 1), fill(n, 0);
  for (int i = 1; i < n; ++i) ++off[parent[i] + 1];
  for (int i = 0; i < n; ++i) off[i + 1] += off[i];
  for (int i = 1; i < n; ++i) child[off[parent[i]] + fill[parent[i]]++] = i;
  for (int i = 1; i < n; ++i)
    if (leaf[i] && off[i + 1] != off[i]) { set_error("vocabulary: leaf %d has children", i); return RGBL_ERR_INVALID; }
  const int rc = rgbl_vocabulary_create(n, L, off.data(), child.data(), desc.data(), weight.data(), word.data(), device, out);
  if (rc == RGBL_OK) (*out)->k = k;
  return rc;
}

int rgbl_vocabulary_info(const rgbl_vocabulary* v, int* k, int* L, int* n_nodes, int* n_words) {
  if (!v) { set_error("null handle"); return RGBL_ERR_INVALID; }
  if (k) *k = v->k;
  if (L) *L = v->L;
  if (n_nodes) *n_nodes = v->n_nodes;
  if (n_words) *n_words = v->n_words;
  return RGBL_OK;
}

int rgbl_bow_descend_batch_device(rgbl_vocabulary* v, void* hip_stream, const uint8_t* d_desc, const int32_t* d_n, int batch, int cap,
                                  int levelsup, int32_t* d_word, double* d_weight, int32_t* d_node) {
  if (!v || !d_desc || !d_word || !d_weight || !d_node || batch < 1 || cap < 1) { set_error("invalid argument"); return RGBL_ERR_INVALID; }
  RGBL_HIP(hipSetDevice(v->device));
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : v->stream;
  hipLaunchKernelGGL(k_bow_descend, dim3((cap + 255) / 256, batch), dim3(256), 0, s, v->dev, d_desc, d_n, cap, levelsup, d_word,
                     d_weight, d_node);
  RGBL_HIP(hipGetLastError());
  return RGBL_OK;
}

// void TemplatedVocabulary::transform(const std::vector<TDescriptor>& features, BowVector& v, FeatureVector& fv, int levelsup)
// (TemplatedVocabulary.h:1127-1192) for TF_IDF / L1_NORM: BowVector::addWeight in feature order, L1 normalisation summed in
// ascending word order, FeatureVector::addFeature with ascending feature indices; stopped words (weight 0) are dropped.
static int bow_transform_core(rgbl_vocabulary* v, const rgbl_device_frame* frame, const uint8_t* desc, int n, int levelsup,
                              uint32_t* word_id, double* word_val, int cap_words, int* n_words, uint32_t* node_id, int32_t* node_off,
                              uint32_t* node_feat, int cap_nodes, int* n_nodes) {
  if (!v || !n_words || !n_nodes || n < 0 || (n > 0 && !desc && !frame)) { set_error("invalid argument"); return RGBL_ERR_INVALID; }
  if (frame && frame->device != v->device) { set_error("device frame and vocabulary live on different devices"); return RGBL_ERR_INVALID; }
  *n_words = 0; *n_nodes = 0;
  if (n == 0) { if (node_off && cap_nodes >= 0) node_off[0] = 0; return RGBL_OK; }
  RGBL_HIP(hipSetDevice(v->device));
  if (n > v->stage_cap) {
    if (v->d_stage) { (void)hipFree(v->d_stage); (void)hipHostFree(v->h_stage); }
    v->d_stage = nullptr; v->h_stage = nullptr; v->stage_cap = 0;
    const int cap = std::max(n, 4096);
    RGBL_HIP(hipMalloc(&v->d_stage, (size_t)cap * 48));
    RGBL_HIP(hipHostMalloc(reinterpret_cast<void**>(&v->h_stage), (size_t)cap * 48, hipHostMallocDefault));
    v->stage_cap = cap;
  }
  const size_t cap = (size_t)v->stage_cap;   // results back to back: weight [n] | word [n] | node [n]
  const uint8_t* d_desc = v->d_stage;
  double* d_weight = reinterpret_cast<double*>(v->d_stage + cap * 32);
  int32_t* d_word = reinterpret_cast<int32_t*>(d_weight + n);
  int32_t* d_node = d_word + n;
  hipStream_t s = v->stream;
  StreamDrain drain(s);  // error returns included
  if (frame) {   // the frame's descriptors are resident: nothing goes up
    RGBL_HIP(hipStreamWaitEvent(s, frame->ready, 0));
    d_desc = frame->d_desc;
  } else {
    memcpy(v->h_stage, desc, (size_t)n * 32);
    RGBL_HIP(hipMemcpyAsync(v->d_stage, v->h_stage, (size_t)n * 32, hipMemcpyHostToDevice, s));
  }
  hipLaunchKernelGGL(k_bow_descend, dim3((n + 255) / 256, 1), dim3(256), 0, s, v->dev, d_desc, (const int32_t*)nullptr, n, levelsup,
                     d_word, d_weight, d_node);
  RGBL_HIP(hipGetLastError());
  RGBL_HIP(hipMemcpyAsync(v->h_stage + cap * 32, v->d_stage + cap * 32, (size_t)n * 16, hipMemcpyDeviceToHost, s));
  RGBL_HIP(hipStreamSynchronize(s));
  const double* weight = reinterpret_cast<const double*>(v->h_stage + cap * 32);
  const int32_t* word = reinterpret_cast<const int32_t*>(weight + n);
  const int32_t* node = word + n;
  // the two std::map containers of the reference, as sorted (key << 32 | feature) runs: one unstable sort of 64-bit integers each
  // (the feature index makes the keys distinct, so the order inside a word / node is the order of the features) on buffers the
  // thread keeps - round 6: 2 x stable_sort of pairs in freshly grown vectors was ~25 us of a 72-us call
  static thread_local std::vector<unsigned long long> wf, nf;
  wf.clear(); nf.clear();
  for (int i = 0; i < n; ++i)
    if (weight[i] > 0) {
      wf.push_back(((unsigned long long)(uint32_t)word[i] << 32) | (uint32_t)i);
      nf.push_back(((unsigned long long)(uint32_t)node[i] << 32) | (uint32_t)i);
    }
  std::sort(wf.begin(), wf.end());
  std::sort(nf.begin(), nf.end());
  int nw = 0, nn = 0;
  for (size_t a = 0; a < wf.size();) {
    size_t b = a;
    double acc = 0.0;
    bool first = true;
    for (; b < wf.size() && (wf[b] >> 32) == (wf[a] >> 32); ++b) {
      const double w = weight[(uint32_t)wf[b]];
      acc = first ? w : acc + w;
      first = false;
    }
    if (nw < cap_words && word_id && word_val) { word_id[nw] = (uint32_t)(wf[a] >> 32); word_val[nw] = acc; }
    ++nw;
    a = b;
  }
  for (size_t a = 0; a < nf.size();) {
    size_t b = a;
    while (b < nf.size() && (nf[b] >> 32) == (nf[a] >> 32)) ++b;
    if (nn < cap_nodes && node_id && node_off && node_feat) {
      node_id[nn] = (uint32_t)(nf[a] >> 32);
      node_off[nn] = (int32_t)a;
      for (size_t q = a; q < b; ++q) node_feat[q] = (uint32_t)nf[q];
    }
    ++nn;
    a = b;
  }
  *n_words = nw; *n_nodes = nn;
  if (nw > cap_words || nn > cap_nodes) { set_error("BoW transform: %d words / %d nodes exceed the output capacity", nw, nn); return RGBL_ERR_CAPACITY; }
  if (node_off) node_off[nn] = (int32_t)nf.size();
  double norm = 0.0;  // BowVector::normalize(L1)
  for (int i = 0; i < nw; ++i) norm += fabs(word_val[i]);
  if (norm > 0.0)
    for (int i = 0; i < nw; ++i) word_val[i] /= norm;
  return RGBL_OK;
}

int rgbl_bow_transform(rgbl_vocabulary* v, const uint8_t* desc, int n, int levelsup, uint32_t* word_id, double* word_val,
                       int cap_words, int* n_words, uint32_t* node_id, int32_t* node_off, uint32_t* node_feat, int cap_nodes,
                       int* n_nodes) {
  return bow_transform_core(v, nullptr, desc, n, levelsup, word_id, word_val, cap_words, n_words, node_id, node_off, node_feat, cap_nodes,
                            n_nodes);
}

int rgbl_bow_transform_frame(rgbl_vocabulary* v, const rgbl_device_frame* frame, int levelsup, uint32_t* word_id, double* word_val,
                             int cap_words, int* n_words, uint32_t* node_id, int32_t* node_off, uint32_t* node_feat, int cap_nodes,
                             int* n_nodes) {
  if (!frame) { set_error("null frame"); return RGBL_ERR_INVALID; }
  return bow_transform_core(v, frame, nullptr, frame->n, levelsup, word_id, word_val, cap_words, n_words, node_id, node_off, node_feat,
                            cap_nodes, n_nodes);
}

// device part of both SearchByBoW overloads: match1[idx1] = feature of `fr` taken by key-frame feature idx1 (or -1), match2 the
// inverse; pa / pb = the shared vocabulary nodes in merge order
static int bow_core(rgbl_matcher* m, const rgbl_keyframe_view* kf, const rgbl_keyframe_view* fr, float nnratio, bool second_needs_mp,
                    int max_best, std::vector<int32_t>& pa, std::vector<int32_t>& pb, std::vector<int32_t>& match1,
                    std::vector<int32_t>& match2, int n_left2 = -1) {
  const int n1 = kf->n, n2 = fr->n;
  match1.assign(n1, -1);
  match2.assign(n2, -1);
  // merge walk of the two sorted FeatureVectors (ORBmatcher.cc:243-246, 388-401)
  pa.clear(); pb.clear();
  for (int a = 0, b = 0; a < kf->n_nodes && b < fr->n_nodes;) {
    if (kf->node_id[a] == fr->node_id[b]) { pa.push_back(a++); pb.push_back(b++); }
    else if (kf->node_id[a] < fr->node_id[b]) ++a;
    else ++b;
  }
  const int npairs = (int)pa.size();
  if (npairs == 0 || n1 == 0 || n2 == 0) return RGBL_OK;
  for (int p = 0; p < npairs; ++p)
    if (fr->node_off[pb[p] + 1] - fr->node_off[pb[p]] > kBowBucket) {
      set_error("SearchByBoW: a vocabulary node holds more than %d features of the second set", kBowBucket);
      return RGBL_ERR_CAPACITY;
    }
  RGBL_HIP(hipSetDevice(m->device));
  StreamDrain drain(m->stream);  // error returns included
  const int nf1 = kf->node_off[kf->n_nodes], nf2 = fr->node_off[fr->n_nodes];
  size_t need = pad256((size_t)n1 * 32) + pad256((size_t)n2 * 32) + pad256(n1) + pad256(n2) + pad256((size_t)(kf->n_nodes + 1) * 4) +
                pad256((size_t)nf1 * 4) + pad256((size_t)(fr->n_nodes + 1) * 4) + pad256((size_t)nf2 * 4) +
                2 * pad256((size_t)npairs * 4) + pad256((size_t)n1 * 4) + pad256((size_t)n2 * 4);
  HostCall hc;
  RGBL_TRY(hc.begin(m, need));
  hipStream_t s = hc.s;
  BowDev T;
  T.valid2 = nullptr;
  const rgbl_keyframe_view* kv[2] = {kf, fr};
  const uint8_t** desc[2] = {&T.desc1, &T.desc2};
  const int32_t **off[2] = {&T.off1, &T.off2}, **feat[2] = {&T.feat1, &T.feat2};
  for (int k = 0; k < 2; ++k) {
    const rgbl_keyframe_view* v = kv[k];
    const int nf = v->node_off[v->n_nodes];
    if (const rgbl_device_frame* f = v->device) {
      RGBL_TRY(hc.use(f, v->n));
      *desc[k] = f->d_desc;
    } else {
      RGBL_TRY(hc.put(desc[k], v->desc, (size_t)v->n * 32));
    }
    if (v->device && v->device->n_nodes == v->n_nodes && v->device->nf == nf) {
      *off[k] = v->device->d_fv; *feat[k] = v->device->d_fv + v->n_nodes + 1;
    } else {
      RGBL_TRY(hc.put(off[k], v->node_off, (size_t)v->n_nodes + 1));
      RGBL_TRY(hc.put(feat[k], v->node_feat, (size_t)nf));
    }
  }
  RGBL_TRY(hc.put(&T.valid1, kf->has_mappoint, (size_t)n1));
  if (second_needs_mp) RGBL_TRY(hc.put(&T.valid2, fr->has_mappoint, (size_t)n2));
  RGBL_TRY(hc.put(&T.pair_n1, pa.data(), (size_t)npairs));
  RGBL_TRY(hc.put(&T.pair_n2, pb.data(), (size_t)npairs));
  T.match1 = hc.put_fill<int32_t>(n1, 0xff);   // all -1: travel with the upload
  T.match2 = hc.put_fill<int32_t>(n2, 0xff);
  hc.mark_result(T.match1, n1);
  hc.mark_result(T.match2, n2);
  RGBL_TRY(hc.upload());
  T.nnratio = nnratio;
  T.max_best = max_best;
  T.n_left2 = n_left2;
  if (n_left2 >= 0) {
    m->timer.begin("k_search_by_bow_rig", s);
    hipLaunchKernelGGL(k_search_by_bow_rig, dim3(npairs), dim3(64), 0, s, T);
  } else {
    m->timer.begin("k_search_by_bow", s);
    hipLaunchKernelGGL(k_search_by_bow, dim3(npairs), dim3(64), 0, s, T);
  }
  m->timer.end(s);
  RGBL_HIP(hipGetLastError());
  RGBL_TRY(hc.fetch());
  memcpy(match1.data(), hc.host(T.match1), sizeof(int32_t) * n1);
  memcpy(match2.data(), hc.host(T.match2), sizeof(int32_t) * n2);
  return RGBL_OK;
}

int rgbl_search_by_bow(rgbl_matcher* m, const rgbl_keyframe_view* kf, const rgbl_keyframe_view* fr, float nnratio,
                       int check_orientation, int32_t* match_f, int* out_nmatches) {
  return rgbl_search_by_bow_rig(m, kf, fr, -1, nnratio, check_orientation, match_f, out_nmatches);
}

// The same with F.Nleft (-1: a single camera): a key-frame feature may hand its map point to a left AND a right frame feature
// (ORBmatcher.cc:298-326, 357-386), so the rotation histogram is filled from the frame's side - its bins' SIZES are all that
// ComputeThreeMaxima reads (:403-421), and every matched frame feature is in exactly one bin.
int rgbl_search_by_bow_rig(rgbl_matcher* m, const rgbl_keyframe_view* kf, const rgbl_keyframe_view* fr, int frame_n_left, float nnratio,
                           int check_orientation, int32_t* match_f, int* out_nmatches) {
  if (!m || !kf || !fr || !match_f || !out_nmatches || kf->n < 0 || fr->n < 0 || frame_n_left < -1 || frame_n_left > fr->n) {
    set_error("invalid argument");
    return RGBL_ERR_INVALID;
  }
  *out_nmatches = 0;
  const int n2 = fr->n;
  for (int i = 0; i < n2; ++i) match_f[i] = -1;
  std::vector<int32_t> pa, pb, match1, match2;
  RGBL_TRY(bow_core(m, kf, fr, nnratio, false, 50 /* <= TH_LOW */, pa, pb, match1, match2, frame_n_left));
  int nmatches = 0;
  for (int i = 0; i < n2; ++i) { match_f[i] = match2[i]; nmatches += match_f[i] >= 0; }
  if (check_orientation) {
    std::vector<int> hist[30];
    const float factor = 1.0f / 30;
    for (int idx2 = 0; idx2 < n2; ++idx2) {
      const int idx1 = match2[idx2];
      if (idx1 < 0) continue;
      float rot = kf->kp_angle[idx1] - fr->kp_angle[idx2];
      if (rot < 0.0) rot += 360.0f;
      int bin = (int)roundf(rot * factor);
      if (bin == 30) bin = 0;
      if (bin >= 0 && bin < 30) hist[bin].push_back(idx2);
    }
    int i1, i2, i3;
    three_maxima(hist, 30, i1, i2, i3);
    for (int i = 0; i < 30; ++i) {
      if (i == i1 || i == i2 || i == i3) continue;
      for (int idx2 : hist[i]) { match_f[idx2] = -1; --nmatches; }
    }
  }
  *out_nmatches = nmatches;
  return RGBL_OK;
}

// int ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12) (src/ORBmatcher.cc:765-905):
// both sides need a good map point, the best distance must be < TH_LOW, the result is indexed by the first key frame.
int rgbl_search_by_bow_keyframes(rgbl_matcher* m, const rgbl_keyframe_view* kf1, const rgbl_keyframe_view* kf2, float nnratio,
                                 int check_orientation, int32_t* match12, int* out_nmatches) {
  if (!m || !kf1 || !kf2 || !match12 || !out_nmatches || kf1->n < 0 || kf2->n < 0) { set_error("invalid argument"); return RGBL_ERR_INVALID; }
  *out_nmatches = 0;
  const int n1 = kf1->n;
  for (int i = 0; i < n1; ++i) match12[i] = -1;
  std::vector<int32_t> pa, pb, match1, match2;
  RGBL_TRY(bow_core(m, kf1, kf2, nnratio, true, 49 /* < TH_LOW */, pa, pb, match1, match2));
  const int npairs = (int)pa.size();
  int nmatches = 0;
  for (int i = 0; i < n1; ++i) { match12[i] = match1[i]; nmatches += match12[i] >= 0; }
  if (check_orientation) {
    std::vector<int> hist[30];
    const float factor = 1.0f / 30;
    for (int p = 0; p < npairs; ++p)
      for (int q = kf1->node_off[pa[p]]; q < kf1->node_off[pa[p] + 1]; ++q) {
        const int idx1 = kf1->node_feat[q];
        const int idx2 = match1[idx1];
        if (idx2 < 0) continue;
        float rot = kf1->kp_angle[idx1] - kf2->kp_angle[idx2];
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)roundf(rot * factor);
        if (bin == 30) bin = 0;
        if (bin >= 0 && bin < 30) hist[bin].push_back(idx1);
      }
    int i1, i2, i3;
    three_maxima(hist, 30, i1, i2, i3);
    for (int i = 0; i < 30; ++i) {
      if (i == i1 || i == i2 || i == i3) continue;
      for (int idx1 : hist[i]) { match12[idx1] = -1; --nmatches; }
    }
  }
  *out_nmatches = nmatches;
  return RGBL_OK;
}

}  // extern "C"
