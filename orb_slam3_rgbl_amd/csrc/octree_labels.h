// Quad-tree distribution (ORBextractor.cc:555-779, DistributeOctTree) without moving a key.
//
// The reference keeps a std::list of nodes, each owning a std::vector of its keypoints, and hands the vectors down to
// the children on every split.  What decides its result is (a) the ORDER of the node list (children are pushed to the
// front, parents erased), (b) the number of keys of every node (split or not, the sort of the expandable nodes near
// the quota) and (c) at the very end, per node, the key of the highest response - the first one in the node's vector
// on ties, and a node's vector always holds its keys in the order of the level's candidate list.  None of it needs the
// keys grouped in memory: here the candidate list stays where the gather put it, every key carries the list position
// of its node in a 16-bit label, and one round of splits is
//     rebuild the node list (prefix sums over the nodes: the std::list order, cf. qt_rebuild)
//   + one flat, fully parallel pass over the keys (new label; count the key into its new node's child quadrant),
// instead of one stable 4-way partition per node (k_octree_moving: a chain of dependent global loads per node, one
// wave per node, 450 k cycles for one KITTI level-0 problem - 2/3 of it waiting).  The node list, the per-child
// counters, the label maps and the sort of the quota phase live in LDS (NCAP nodes); (c) is an atomic maximum over
// (response, -candidate index).
//
// Launch: one workgroup of BS work-items per (level, frame), grid (levels, frames).
#pragma once

namespace rgbl {

template <int NCAP>
struct alignas(16) QtStore {
  uint2 geom[2][NCAP];        // x0 | x1 << 16, y0 | y1 << 16 (node rectangle, relative to minBorder)
  uint32_t cnt[2][NCAP][4];   // keys per child quadrant; the list that is not current doubles as the sort's scratch
  uint32_t map[NCAP];         // old list position -> bit 31: split, bits 30..0: rank of its first child | new position
  uint16_t todo[2][NCAP];     // expandable nodes in creation order (vSizeAndPointerToNode); at the end: best key per node
  uint16_t sval[NCAP];        // sorted expandable nodes (list positions)
};

__device__ __forceinline__ int qt_mid(uint32_t lohi) {  // lo + ceil((hi - lo) / 2), ORBextractor.cc:422-423
  const int lo = (int)(lohi & 0xffffu), hi = (int)(lohi >> 16);
  return lo + ((hi - lo + 1) >> 1);
}
__device__ __forceinline__ int qt_quadrant(uint32_t key, uint2 G) {  // n1 = 0 (UL), n2 = 1 (UR), n3 = 2 (BL), n4 = 3 (BR)
  return (key_x(key) < qt_mid(G.x) ? 0 : 1) | (key_y(key) < qt_mid(G.y) ? 0 : 2);
}
__device__ __forceinline__ uint2 qt_child(uint2 G, int q) {
  const uint32_t mx = (uint32_t)qt_mid(G.x), my = (uint32_t)qt_mid(G.y);
  uint2 c;
  c.x = (q & 1) ? (mx | (G.x & 0xffff0000u)) : ((G.x & 0xffffu) | (mx << 16));
  c.y = (q & 2) ? (my | (G.y & 0xffff0000u)) : ((G.y & 0xffffu) | (my << 16));
  return c;
}

// +1 on counters[slot] (slot < 0: nothing).  `few`: the wave's keys fall into a handful of counters (early rounds:
// 64 neighbouring candidates share one or two nodes) - one atomic per distinct counter instead of same-address
// atomics that the LDS serialises.  `few` must be wave-uniform.
__device__ __forceinline__ void qt_count(uint32_t* counters, int slot, bool few) {
  if (few) {
    unsigned long long left = __ballot(slot >= 0);
    while (left) {
      const int leader = __ffsll((long long)left) - 1;
      const int v = __shfl(slot, leader);
      const unsigned long long same = __ballot(slot == v);
      if (lane_id() == leader) atomicAdd(&counters[v], (uint32_t)__popcll(same));
      left &= ~same;
    }
  } else if (slot >= 0) {
    atomicAdd(&counters[slot], 1u);
  }
}

// New node list after the nodes of processing ranks 0 .. P-1 were offered for splitting (rank -> list position:
// identity in the breadth-first rounds, the sorted order from the back near the quota), as std::list push_front /
// erase leave it:
//   [children of the LAST processed node (n4..n1), ..., children of the FIRST processed node] ++ [nodes not split, old order]
// A node is split when it holds more than one key.  Writes list 1-p (rectangles; counters zeroed for children, copied
// for survivors), todo[1-p] (children with more than one key, creation order) and map[]; *s_T = number of children.
template <int BS, int NCAP>
__device__ __forceinline__ void qt_rebuild(QtStore<NCAP>& S, int p, int n, int P, int m, bool identity, int cap,
                                           unsigned long long* s_scan, int* s_newn, int* s_nexp, uint32_t* s_T) {
  const int tid = threadIdx.x;
  if (!identity)
    for (int pos = tid; pos < n; pos += BS) S.map[pos] = 0;
  // children | expandable children << 32 of rank rho
  auto offer = [&](int rho, int& pos, uint32_t c[4]) -> unsigned long long {
    pos = identity ? rho : (int)S.sval[m - 1 - rho];
    unsigned long long v = 0;
    uint32_t total = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      c[q] = S.cnt[p][pos][q];
      total += c[q];
      v += (c[q] > 0 ? 1ull : 0ull) + (c[q] > 1 ? (1ull << 32) : 0ull);
    }
    return total > 1 ? v : 0ull;
  };
  unsigned long long T2 = 0;
  for (int r0 = 0; r0 < P; r0 += BS) {
    const int rho = r0 + tid;
    int pos;
    uint32_t c[4];
    const unsigned long long v = rho < P ? offer(rho, pos, c) : 0ull;
    unsigned long long tot;
    block_exclusive_scan<unsigned long long>(v, s_scan, &tot);
    T2 += tot;
  }
  const uint32_t T = (uint32_t)(T2 & 0xffffffffu);
  unsigned long long carry = 0;
  for (int r0 = 0; r0 < P; r0 += BS) {
    const int rho = r0 + tid;
    int pos = 0;
    uint32_t c[4] = {0, 0, 0, 0};
    const unsigned long long v = rho < P ? offer(rho, pos, c) : 0ull;
    unsigned long long tot;
    const unsigned long long ex = carry + block_exclusive_scan<unsigned long long>(v, s_scan, &tot);
    carry += tot;
    if (rho < P) {
      if (v != 0) {
        const uint2 G = S.geom[p][pos];
        uint32_t child_rank = (uint32_t)(ex & 0xffffffffu), exp_rank = (uint32_t)(ex >> 32);
        S.map[pos] = 0x80000000u | child_rank;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (c[q] > 0) {
            const uint32_t idx = T - child_rank - 1;
            if (idx < (uint32_t)cap) {
              S.geom[1 - p][idx] = qt_child(G, q);
              S.cnt[1 - p][idx][0] = 0; S.cnt[1 - p][idx][1] = 0; S.cnt[1 - p][idx][2] = 0; S.cnt[1 - p][idx][3] = 0;
            }
            if (c[q] > 1) {
              if (exp_rank < (uint32_t)cap) S.todo[1 - p][exp_rank] = (uint16_t)idx;
              ++exp_rank;
            }
            ++child_rank;
          }
      } else if (identity) {
        S.map[pos] = 0;
      }
    }
  }
  __syncthreads();
  uint32_t kcarry = 0;
  for (int p0 = 0; p0 < n; p0 += BS) {
    const int pos = p0 + tid;
    const uint32_t keep = (pos < n && (S.map[pos] >> 31) == 0) ? 1u : 0u;
    uint32_t tot;
    const uint32_t ex = kcarry + block_exclusive_scan<uint32_t>(keep, reinterpret_cast<uint32_t*>(s_scan), &tot);
    kcarry += tot;
    if (keep) {
      const uint32_t np = T + ex;
      if (np < (uint32_t)cap) {
        S.geom[1 - p][np] = S.geom[p][pos];
#pragma unroll
        for (int q = 0; q < 4; ++q) S.cnt[1 - p][np][q] = S.cnt[p][pos][q];
      }
      S.map[pos] = np;
    }
  }
  if (tid == 0) { *s_newn = (int)(T + kcarry); *s_nexp = (int)(T2 >> 32); *s_T = T; }
  __syncthreads();
}

template <int NCAP> struct QtRanges { static constexpr int value = NCAP / 16 + 16; };  // pending ranges hold > 16 elements and are disjoint

template <int BS, int NCAP>
__global__ __launch_bounds__(BS) void k_octree(const LevelGeom* __restrict__ geom, int n_levels, OctreeBufs b, int level_begin) {
  using Ranges = SortRangesT<QtRanges<NCAP>::value>;
  __shared__ QtStore<NCAP> S;
  __shared__ unsigned long long s_scan[32];
  __shared__ Ranges s_ra, s_rb;
  __shared__ int s_sort_cnt[2];
  __shared__ int s_newn, s_nexp, s_n, s_P;
  __shared__ uint32_t s_T;
  __shared__ uint32_t s_rootcnt[kMaxRoots];
  __shared__ int s_rootpos[kMaxRoots];
#define RGBL_STAMP(k) do { if (b.dbg && threadIdx.x == 0) b.dbg[((size_t)blockIdx.y * n_levels + blockIdx.x + level_begin) * 16 + (k)] = rgbl_clock(); } while (0)
  RGBL_STAMP(0);

  const int tid = threadIdx.x, lane = lane_id();
  const int l = blockIdx.x + level_begin, f = blockIdx.y;  // the launch covers the levels level_begin .. level_begin + gridDim.x
  const LevelGeom& g = geom[l];
  uint32_t* keys = b.keys_a + (size_t)f * b.keys_frame + g.key_off;
  uint16_t* label = reinterpret_cast<uint16_t*>(b.keys_b + (size_t)f * b.keys_frame + g.key_off);
  const int N = g.quota;
  const int cap = (int)g.node_cap < NCAP ? (int)g.node_cap : NCAP;

  // ---- 0. the cells' candidates as one dense list in the reference's order (cell-major), each labelled with its root
  //         node (ORBextractor.cc:582-586).  Output position j -> cell by bisection of the cells' prefix sums.
  uint32_t* s_pref = &S.cnt[0][0][0];
  constexpr int kPrefCap = 8 * NCAP;
  uint32_t C = 0;
  {
    const uint32_t* ccnt = b.cell_cnt + (size_t)f * b.cells_frame + g.cell_off;
    const uint32_t* slots = b.slots + (size_t)f * b.slots_frame + g.slot_off;
    const uint8_t* rootx = b.rootx + g.rootx_off;
    if (tid < kMaxRoots) s_rootcnt[tid] = 0;
    for (int c_lo = 0; c_lo < g.n_cells; c_lo += kPrefCap) {
      const int c_hi = c_lo + kPrefCap < g.n_cells ? c_lo + kPrefCap : g.n_cells;
      const int nc = c_hi - c_lo;
      __syncthreads();  // the previous chunk's bisections are done
      uint32_t run = 0;
      for (int c0 = c_lo; c0 < c_hi; c0 += BS) {
        const int c = c0 + tid;
        const uint32_t cnt = c < c_hi ? ccnt[c] : 0u;
        uint32_t tot;
        const uint32_t ex = run + block_exclusive_scan<uint32_t>(cnt, reinterpret_cast<uint32_t*>(s_scan), &tot);
        if (c < c_hi) s_pref[c - c_lo] = ex;
        run += tot;
      }
      __syncthreads();
      for (uint32_t j0 = 0; j0 < run; j0 += BS) {
        const uint32_t j = j0 + (uint32_t)tid;
        int r = -1;
        if (j < run) {
          int lo = 0, hi = nc;  // s_pref[lo] <= j, and (hi == nc or s_pref[hi] > j): the last cell that starts at or before j
          while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_pref[mid] <= j) lo = mid; else hi = mid;
          }
          const uint32_t key = slots[(size_t)(c_lo + lo) * g.cell_cap + (j - s_pref[lo])];
          keys[C + j] = key;
          r = rootx[key_x(key)];
          label[C + j] = (uint16_t)r;
        }
        for (int k = 0; k < g.n_ini; ++k) {
          const int c = __popcll(__ballot(r == k));
          if (lane == 0 && c) atomicAdd(&s_rootcnt[k], (uint32_t)c);
        }
      }
      C += run;
    }
  }
  __syncthreads();
  RGBL_STAMP(1);

  // ---- 1. root nodes; empty ones are erased (ORBextractor.cc:566-606)
  if (tid == 0) {
    int n0 = 0;
    for (int r = 0; r < g.n_ini; ++r) {
      s_rootpos[r] = n0;
      if (s_rootcnt[r] > 0) {
        uint2 G;
        G.x = (uint32_t)g.root_x[r] | ((uint32_t)g.root_x[r + 1] << 16);
        G.y = (uint32_t)(g.max_by - kMinBorder) << 16;
        S.geom[0][n0] = G;
        S.cnt[0][n0][0] = 0; S.cnt[0][n0][1] = 0; S.cnt[0][n0][2] = 0; S.cnt[0][n0][3] = 0;
        ++n0;
      }
    }
    s_n = n0;
  }
  __syncthreads();
  int n = s_n;
  for (uint32_t i0 = 0; i0 < C; i0 += BS) {
    const uint32_t i = i0 + (uint32_t)tid;
    int slot = -1;
    if (i < C) {
      const int pos = s_rootpos[label[i]];
      label[i] = (uint16_t)pos;
      slot = pos * 4 + qt_quadrant(keys[i], S.geom[0][pos]);
    }
    qt_count(&S.cnt[0][0][0], slot, true);
  }
  __syncthreads();
  RGBL_STAMP(2);

  // ---- 2. rounds of splits: breadth-first (ORBextractor.cc:608-686), then - once another full round would overshoot
  //         the quota - the most populated nodes first, one at a time in the reference, up to the node that reaches
  //         the quota (ORBextractor.cc:689-753)
  uint32_t* best = reinterpret_cast<uint32_t*>(&S.todo[0][0]);
  int p = 0, m = 0;
  bool finished = (n == 0), careful = false, stamped = false;
  while (!finished) {
    const int prev = n;
    int P = n;
    if (careful) {
      // compareNodes orders by (size, UL.x); equal keys end up in libstdc++'s introsort order
      uint64_t* w = reinterpret_cast<uint64_t*>(&S.cnt[1 - p][0][0]) + 4;  // 4 entries of read slack on both sides
      uint16_t* seg_first = reinterpret_cast<uint16_t*>(w + NCAP + 4);
      uint16_t* seg_last = seg_first + NCAP;
      static_assert((NCAP + 8) * 8 + NCAP * 4 <= NCAP * 16, "the sort's scratch has to fit the idle counters");
      for (int j = tid; j < m; j += BS) {
        const uint32_t pos = S.todo[p][j];
        const uint32_t total = S.cnt[p][pos][0] + S.cnt[p][pos][1] + S.cnt[p][pos][2] + S.cnt[p][pos][3];
        w[j] = ((uint64_t)total << 28) | ((uint64_t)(S.geom[p][pos].x & 0xffffu) << 16) | pos;  // x0 < 4096, pos < 65536
      }
      __syncthreads();
      RGBL_STAMP(8);
      block_sort_restated<BS, NCAP, Ranges>(w, m, seg_first, seg_last, &s_ra, &s_rb, s_sort_cnt);
      RGBL_STAMP(9);
      for (int j = tid; j < m; j += BS) S.sval[j] = (uint16_t)(w[j] & 0xffffu);
      if (tid == 0) s_P = m;
      __syncthreads();
      // first rank (from the back of the sorted array) after which the list has reached the quota: the reference
      // breaks out of its loop there
      long long carry = 0;
      for (int r0 = 0; r0 < m; r0 += BS) {
        const int rho = r0 + tid;
        long long v = 0;
        if (rho < m) {
          const uint32_t* c = S.cnt[p][S.sval[m - 1 - rho]];
          v = (long long)((c[0] > 0) + (c[1] > 0) + (c[2] > 0) + (c[3] > 0)) - 1;
        }
        unsigned long long tot;
        const unsigned long long ex = block_exclusive_scan<unsigned long long>((unsigned long long)v, s_scan, &tot);
        const long long size_after = (long long)n + carry + (long long)ex + v;
        if (rho < m && size_after >= N) atomicMin(&s_P, rho + 1);
        carry += (long long)tot;
      }
      __syncthreads();
      P = s_P;
    }
    qt_rebuild<BS, NCAP>(S, p, n, P, m, !careful, cap, s_scan, &s_newn, &s_nexp, &s_T);
    n = s_newn;
    m = s_nexp;
    const uint32_t T = s_T;
    if (n > cap - 8) { if (tid == 0) atomicOr(b.err, 1); finished = true; }
    else if (n >= N || n == prev) finished = true;
    else if (!careful && n + 3 * m > N) careful = true;
    if (!stamped && (careful || finished)) { RGBL_STAMP(3); stamped = true; }
    if (finished) {
      for (int pos = tid; pos < NCAP; pos += BS) best[pos] = 0;
      __syncthreads();
    }
    // every key: the list position of its node in the new list; unless this was the last round, counted into its
    // node's child quadrant (nodes that were not split keep their counters)
    const bool few = n <= 64;
    for (uint32_t i0 = 0; i0 < C; i0 += BS) {
      const uint32_t i = i0 + (uint32_t)tid;
      int slot = -1;
      if (i < C) {
        const uint32_t key = keys[i];
        const uint32_t old = label[i];
        const uint32_t mp = S.map[old];
        uint32_t idx = mp;
        if (mp >> 31) {
          const uint2 G = S.geom[p][old];
          const int q = qt_quadrant(key, G);
          const uint32_t* c = S.cnt[p][old];
          const uint32_t c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3];
          const uint32_t before = (q > 0 && c0 > 0 ? 1u : 0u) + (q > 1 && c1 > 0 ? 1u : 0u) + (q > 2 && c2 > 0 ? 1u : 0u);
          idx = T - 1u - ((mp & 0x7fffffffu) + before);
          const uint32_t mine = q == 0 ? c0 : (q == 1 ? c1 : (q == 2 ? c2 : c3));
          if (!finished && mine > 1 && idx < (uint32_t)cap) slot = (int)idx * 4 + qt_quadrant(key, qt_child(G, q));
          label[i] = (uint16_t)idx;
        } else if (idx != old) {
          label[i] = (uint16_t)idx;
        }
        // ---- 3. the strongest key of every node, the first one of the candidate list on ties (ORBextractor.cc:757-776)
        if (finished && idx < (uint32_t)cap) atomicMax(&best[idx], ((uint32_t)key_s(key) << 24) | (0xffffffu - i));
      }
      if (!finished) qt_count(&S.cnt[1 - p][0][0], slot, few);
    }
    p ^= 1;
    __syncthreads();
  }
  if (!stamped) RGBL_STAMP(3);
  RGBL_STAMP(4);

  uint32_t* out = b.kp_key + (size_t)f * b.kp_frame + g.koff;
  if (n > g.kcap) { if (tid == 0) atomicOr(b.err, 2); n = g.kcap; }
  if (n > cap) n = cap;
  for (int pos = tid; pos < n; pos += BS) {
    const uint32_t v = best[pos];
    out[pos] = v ? keys[0xffffffu - (v & 0xffffffu)] : 0u;
  }
  if (tid == 0) b.kp_count[(size_t)f * n_levels + l] = n;
  RGBL_STAMP(5);
  if (b.dbg && tid == 0) { b.dbg[((size_t)f * n_levels + l) * 16 + 6] = C; b.dbg[((size_t)f * n_levels + l) * 16 + 7] = (unsigned long long)n; }
#undef RGBL_STAMP
}

}  // namespace rgbl
