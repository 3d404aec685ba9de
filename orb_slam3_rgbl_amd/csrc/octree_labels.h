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
// Input: normally one list per (level, frame) that k_fast_cells' cells appended to in whatever order they finished (the
// reference's candidate order - cell after cell, row-major inside a cell - is a function of a key's coordinates and only
// matters for (c)); with RGBL_DENSE=0 the cells' own slots, gathered here in that order.
//
// Launch: one workgroup of BS work-items per (level, frame), grid (levels, frames).
#pragma once

namespace rgbl {

template <int NCAP>
struct alignas(16) QtStore {
  uint2 geom[2][NCAP];        // x0 | x1 << 16, y0 | y1 << 16 (node rectangle, relative to minBorder)
  uint32_t cnt[2][NCAP][4];   // keys per child quadrant; the list that is not current doubles as the sort's scratch
  uint2 ctab[NCAP];           // old list position -> four 16-bit entries, one per quadrant: new list position of the keys
                              // of that quadrant | 0x8000 when they are to be counted into their new node's quadrants
                              // (a child with more than one key); a node that was not split: its new position, four times
  uint32_t mid[2][NCAP];      // split point of a node's rectangle: x | y << 16
  uint16_t todo[2][NCAP];     // expandable nodes in creation order (vSizeAndPointerToNode)
  uint16_t sval[NCAP];        // sorted expandable nodes (list positions)
};

__device__ __forceinline__ int qt_mid(uint32_t lohi) {  // lo + ceil((hi - lo) / 2), ORBextractor.cc:422-423
  const int lo = (int)(lohi & 0xffffu), hi = (int)(lohi >> 16);
  return lo + ((hi - lo + 1) >> 1);
}
__device__ __forceinline__ uint32_t qt_mid2(uint2 G) { return (uint32_t)qt_mid(G.x) | ((uint32_t)qt_mid(G.y) << 16); }
// n1 = 0 (UL), n2 = 1 (UR), n3 = 2 (BL), n4 = 3 (BR) of a node split at `mid`
// (bits 12 .. 15 of a stored split point carry the node's depth below its root, qt_mid_depth: coordinates stay below 4096)
__device__ __forceinline__ int qt_quadrant(uint32_t key, uint32_t mid) {
  return (key_x(key) < (int)(mid & 0xfffu) ? 0 : 1) | (key_y(key) < (int)(mid >> 16) ? 0 : 2);
}
__device__ __forceinline__ uint32_t qt_mid_depth(uint2 G, int depth) { return qt_mid2(G) | ((uint32_t)(depth < 15 ? depth : 15) << 12); }
__device__ __forceinline__ int qt_depth_of(uint32_t mid) { return (int)((mid >> 12) & 0xfu); }
__device__ __forceinline__ uint2 qt_child(uint2 G, int q) {
  const uint32_t mx = (uint32_t)qt_mid(G.x), my = (uint32_t)qt_mid(G.y);
  uint2 c;
  c.x = (q & 1) ? (mx | (G.x & 0xffff0000u)) : ((G.x & 0xffffu) | (mx << 16));
  c.y = (q & 2) ? (my | (G.y & 0xffff0000u)) : ((G.y & 0xffffu) | (my << 16));
  return c;
}
constexpr uint32_t kQtNotSplit = 0xffffffffu;  // ctab[].x of a node the current round has not split (yet)

// +1 on counters[slot] (slot < 0: nothing).  Plain LDS atomics: peeling the distinct counters of a wave (one atomic per
// counter, ballot + readlane per distinct value) was measured slower in every round, 0.35 vs 0.27 ms per 512 frames.
__device__ __forceinline__ void qt_count(uint32_t* counters, int slot) {
  if (slot >= 0) atomicAdd(&counters[slot], 1u);
}
// While the list is short, thousands of keys meet in a few dozen counters and the LDS serialises same-address atomics (a
// dense KITTI level 0: 8 600 keys into 12 counters took 160 k cycles with four workgroups sharing the CU's LDS).  Those
// rounds count into 2^rlog copies of every counter, the copy chosen by the lane, and sum the copies afterwards.
__device__ __forceinline__ void qt_count_rep(uint32_t* copies, int slot, int rlog) {
  if (slot >= 0) atomicAdd(&copies[(slot << rlog) + (lane_id() & ((1 << rlog) - 1))], 1u);
}
__device__ __forceinline__ int qt_rep_log(int counters, int words) {  // copies per counter that fit `words`, at most 32
  int rlog = 0;
  while (rlog < 5 && (counters << (rlog + 1)) <= words) ++rlog;
  return rlog;
}

// New node list after the nodes of processing ranks 0 .. P-1 were offered for splitting (rank -> list position:
// identity in the breadth-first rounds, the sorted order from the back near the quota), as std::list push_front /
// erase leave it:
//   [children of the LAST processed node (n4..n1), ..., children of the FIRST processed node] ++ [nodes not split, old order]
// A node is split when it holds more than one key.  Writes list 1-p (rectangles, split points; counters zeroed for
// children, copied for survivors), todo[1-p] (children with more than one key, creation order) and ctab[].
template <int BS, int NCAP>
__device__ __forceinline__ void qt_rebuild(QtStore<NCAP>& S, int p, int n, int P, int m, bool identity, int cap,
                                           uint32_t* s_scan, int* s_newn, int* s_nexp, int* s_nchild = nullptr) {
  const int tid = threadIdx.x;
  constexpr int kPer = (NCAP + BS - 1) / BS;  // ranks per work-item: their counters stay in registers between the two steps
  if (!identity)
    for (int pos = tid; pos < n; pos += BS) S.ctab[pos].x = kQtNotSplit;
  // step 1: children | expandable children << 16 per rank, exclusive prefix (both fit 16 bits: at most 4 per node)
  int pos_r[kPer];
  uint32_t c_r[kPer][4], v_r[kPer], ex_r[kPer];
  uint32_t carry = 0;
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    if (k * BS >= P) break;  // uniform
    const int rho = k * BS + tid;
    pos_r[k] = 0; v_r[k] = 0;
    c_r[k][0] = c_r[k][1] = c_r[k][2] = c_r[k][3] = 0;
    if (rho < P) {
      pos_r[k] = identity ? rho : (int)S.sval[m - 1 - rho];
      uint32_t total = 0, v = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        c_r[k][q] = S.cnt[p][pos_r[k]][q];
        total += c_r[k][q];
        v += (c_r[k][q] > 0 ? 1u : 0u) + (c_r[k][q] > 1 ? 0x10000u : 0u);
      }
      v_r[k] = total > 1 ? v : 0u;
    }
    uint32_t tot;
    ex_r[k] = carry + block_exclusive_scan<uint32_t>(v_r[k], s_scan, &tot);
    carry += tot;
  }
  const uint32_t T = carry & 0xffffu, E = carry >> 16;
  // step 2: the children, last processed node first
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    if (k * BS >= P) break;
    const int rho = k * BS + tid;
    if (rho < P) {
      const int pos = pos_r[k];
      if (v_r[k] != 0) {
        const uint2 G = S.geom[p][pos];
        uint32_t child_rank = ex_r[k] & 0xffffu, exp_rank = ex_r[k] >> 16;
        uint32_t e[4] = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (c_r[k][q] > 0) {
            const uint32_t idx = T - child_rank - 1;
            e[q] = idx;
            if (idx < (uint32_t)cap) {
              const uint2 Gc = qt_child(G, q);
              S.geom[1 - p][idx] = Gc;
              S.mid[1 - p][idx] = qt_mid_depth(Gc, qt_depth_of(S.mid[p][pos]) + 1);
              S.cnt[1 - p][idx][0] = 0; S.cnt[1 - p][idx][1] = 0; S.cnt[1 - p][idx][2] = 0; S.cnt[1 - p][idx][3] = 0;
            }
            if (c_r[k][q] > 1) {
              e[q] |= 0x8000u;
              if (exp_rank < (uint32_t)cap) S.todo[1 - p][exp_rank] = (uint16_t)idx;
              ++exp_rank;
            }
            ++child_rank;
          }
        uint2 t;
        t.x = e[0] | (e[1] << 16);
        t.y = e[2] | (e[3] << 16);
        S.ctab[pos] = t;
      } else if (identity) {
        S.ctab[pos].x = kQtNotSplit;
      }
    }
  }
  __syncthreads();
  // step 3: the nodes that were not split keep their order behind the children
  uint32_t kcarry = 0;
  for (int p0 = 0; p0 < n; p0 += BS) {
    const int pos = p0 + tid;
    const uint32_t keep = (pos < n && S.ctab[pos].x == kQtNotSplit) ? 1u : 0u;
    uint32_t tot;
    const uint32_t ex = kcarry + block_exclusive_scan<uint32_t>(keep, s_scan, &tot);
    kcarry += tot;
    if (keep) {
      const uint32_t np = T + ex;
      if (np < (uint32_t)cap) {
        S.geom[1 - p][np] = S.geom[p][pos];
        S.mid[1 - p][np] = S.mid[p][pos];
#pragma unroll
        for (int q = 0; q < 4; ++q) S.cnt[1 - p][np][q] = S.cnt[p][pos][q];
      }
      uint2 t;
      t.x = t.y = np | (np << 16);
      S.ctab[pos] = t;
    }
  }
  if (tid == 0) { *s_newn = (int)(T + kcarry); *s_nexp = (int)E; if (s_nchild) *s_nchild = (int)T; }
  __syncthreads();
}

// ---- The breadth-first phase without passes over the keys (round 4) ---------------------------------------------------
// A node's rectangle is a function of its root and of the path of quadrants that leads to it - the split points (qt_mid) do
// not depend on the keys, and x and y split independently.  So ONE pass can give every key the path it would take through
// HD rounds of splits (root | x path | y path = its cell of the 2^HD x 2^HD subdivision of its root) and count the keys per
// cell; the counts of every shallower node are sums of 2 x 2 blocks (a pyramid of HD levels).  While the reference is in
// its breadth-first phase (every node with more than one key splits, ORBextractor.cc:608-686) a round then only rebuilds
// the node list - the children's per-quadrant counts come out of the pyramid - and no key is touched.  The keys meet their
// nodes again through a table cell -> list position when the first round needs more than the pyramid holds (depth > HD - 2),
// when the quota phase starts (its rounds split only some nodes) or at the very end.
// level d (1 .. D) of the pyramid: entry (root << 2 d) | (x path << d) | y path, behind the entries of the levels above it
__device__ __forceinline__ int qt_hist_off(int n_ini, int d) { return n_ini * (((1 << (2 * d)) - 4) / 3); }
// the path of a coordinate through `depth` splits of [lo, hi]: bit = 1 where it lies right of (at) the split point
__device__ __forceinline__ int qt_path(int v, int lo, int hi, int depth) {
  int path = 0;
  for (int d = 0; d < depth; ++d) {
    const int mid = lo + ((hi - lo + 1) >> 1);
    const int bit = v >= mid ? 1 : 0;
    lo = bit ? mid : lo;
    hi = bit ? hi : mid;
    path = (path << 1) | bit;
  }
  return path;
}
// depth and paths of a NODE rectangle (x0 | x1 << 16, y0 | y1 << 16) below its root's: descends until the interval is met
__device__ __forceinline__ int qt_node_path(uint2 G, int rx0, int rx1, int ry1, int max_depth, int* xp, int* yp) {
  int xlo = rx0, xhi = rx1, ylo = 0, yhi = ry1, px = 0, py = 0, d = 0;
  const int x0 = (int)(G.x & 0xffffu), x1 = (int)(G.x >> 16), y0 = (int)(G.y & 0xffffu), y1 = (int)(G.y >> 16);
  while (d < max_depth && !(xlo == x0 && xhi == x1 && ylo == y0 && yhi == y1)) {
    const int mx = xlo + ((xhi - xlo + 1) >> 1), my = ylo + ((yhi - ylo + 1) >> 1);
    const int bx = x0 >= mx ? 1 : 0, by = y0 >= my ? 1 : 0;
    xlo = bx ? mx : xlo; xhi = bx ? xhi : mx;
    ylo = by ? my : ylo; yhi = by ? yhi : my;
    px = (px << 1) | bx; py = (py << 1) | by;
    ++d;
  }
  *xp = px; *yp = py;
  return d;
}

constexpr int kQtBatch = 4;   // keys per work-item and trip of the passes over the candidate list (independent loads in flight)

template <int NCAP> struct QtRanges { static constexpr int value = NCAP / 16 + 16; };  // pending ranges hold > 16 elements and are disjoint

// KEYCAP > 0 (round 4, single frames): candidate lists of up to KEYCAP keys are copied into LDS by the first pass, together
// with their labels, and every later pass reads them there.  ONE (level, frame) problem is a chain of ~8 passes over its keys,
// and with only 8 problems on the GPU nothing hides a pass's global-memory round trips (~17 keys per work-item in batches of
// 4: five dependent L2 latencies per pass, 24 of the 62 us of a KITTI level-0 problem).  Occupancy does not matter there, so
// the workgroup also takes 1024 work-items.  Batches keep the global lists (37 KB of LDS, four problems per CU).
template <int BS, int NCAP, int KEYCAP = 0>
__global__ __launch_bounds__(BS) void k_octree(const LevelGeom* __restrict__ geom, int n_levels, OctreeBufs b, int level_begin) {
  static_assert(NCAP <= 0x4000, "list positions travel in 15 bits of the ctab entries");
  __shared__ uint32_t s_keys[KEYCAP > 0 ? KEYCAP : 1];
  __shared__ uint16_t s_label[KEYCAP > 0 ? KEYCAP : 2];
  // the pyramid of per-cell key counts (see above): as deep as its space allows - 4 roots x 3 levels next to the 37 KB of a
  // batch workgroup (four of them still share a CU), 5 levels where LDS is not what limits the workgroups per CU
  constexpr int kHistEntries = KEYCAP > 0 ? 1 : (NCAP > 512 ? 2 * 1364 : 4 * 84);   // (the single-frame instantiation does without, see below)
  __shared__ uint32_t s_hist[kHistEntries];
  __shared__ int s_rootx[kMaxRoots + 1];
  __shared__ int s_nchild;
  using Ranges = SortRangesT<QtRanges<NCAP>::value>;
  __shared__ QtStore<NCAP> S;
  __shared__ uint32_t s_scan[32];
  __shared__ Ranges s_ra, s_rb;
  __shared__ int s_sort_cnt[2];
  __shared__ int s_newn, s_nexp, s_n, s_P;
  __shared__ int s_rootpos[kMaxRoots], s_rootfirst[kMaxRoots + 1];
#define RGBL_STAMP(k) do { if (b.dbg && threadIdx.x == 0) b.dbg[((size_t)blockIdx.y * n_levels + blockIdx.x + level_begin) * 16 + (k)] = rgbl_clock(); } while (0)
  RGBL_STAMP(0);

  const int tid = threadIdx.x;
  const int l = blockIdx.x + level_begin, f = blockIdx.y;  // the launch covers the levels level_begin .. level_begin + gridDim.x
  const LevelGeom& g = geom[l];
  uint32_t* keys = b.keys_a + (size_t)f * b.keys_frame + g.key_off;
  uint16_t* label = reinterpret_cast<uint16_t*>(b.keys_b + (size_t)f * b.keys_frame + g.key_off);
  // scalars of the level up front: behind the stores below the compiler would have to re-read them from memory
  const int N = g.quota, n_cells = g.n_cells, cell_cap = g.cell_cap, n_ini = g.n_ini, kcap = g.kcap;
  const int cap = (int)g.node_cap < NCAP ? (int)g.node_cap : NCAP;
  if (tid <= kMaxRoots) { s_rootfirst[tid] = g.root_first[tid]; s_rootx[tid] = g.root_x[tid]; }
  // root of a key (ORBextractor.cc:585, vpIniNodes[kp.pt.x / hX]): root_first[k] = first x that lands in root k or beyond
  auto root_of = [&](uint32_t key) {
    int r = 0;
    for (int k = 1; k < n_ini; ++k) r += key_x(key) >= s_rootfirst[k] ? 1 : 0;
    return r;
  };

  // ---- 0. + 1.  The cells' candidates as one dense list in the reference's order (cell-major), each labelled with its
  //         root node and counted into its root's quadrants on the way (ORBextractor.cc:566-606).  Half a wave
  //         copies one cell at a time (its slots are contiguous: whole cache lines in, whole lines out - with four
  //         2 048 workgroups at once, 8-lane groups cost 2.5 x the time); four cells per trip, 8 requests per lane in flight.
  //         Roots are numbered as if none were empty; the rare frame with an empty root is fixed up afterwards.
  uint32_t* s_pref = &S.cnt[0][kMaxRoots][0];  // behind the root nodes' counters
  constexpr int kPrefCap = 8 * NCAP - 4 * kMaxRoots - 1;
  constexpr int kGrp = 32, kGroups = BS / kGrp, kPre = 2, kTrip = 4;
  if (tid < n_ini) {
    uint2 G;
    G.x = (uint32_t)g.root_x[tid] | ((uint32_t)g.root_x[tid + 1] << 16);
    G.y = (uint32_t)(g.max_by - kMinBorder) << 16;
    S.geom[0][tid] = G;
    S.mid[0][tid] = qt_mid_depth(G, 0);
    S.cnt[0][tid][0] = 0; S.cnt[0][tid][1] = 0; S.cnt[0][tid][2] = 0; S.cnt[0][tid][3] = 0;
  }
  uint32_t* s_rep0 = reinterpret_cast<uint32_t*>(&S.ctab[0]);  // 2 NCAP words, idle until the first rebuild
  const int rlog0 = qt_rep_log(4 * n_ini, 2 * NCAP);
  for (int i = tid; i < ((4 * n_ini) << rlog0); i += BS) s_rep0[i] = 0;
  uint32_t C = 0;
  const bool dense = b.level_cnt != nullptr;  // k_fast_cells left the level's candidates as one list (any cell order)
  // depth of the pyramid: what fits its space for this level's number of roots (0 = the round-by-round passes of rounds 1 - 3)
  int hist_depth = 0;
  // Not for the handful-of-problems instantiation (KEYCAP > 0): with the keys in LDS a pass over them is cheap, and what the
  // pyramid adds - the paths of 8 600 keys in the first pass, the cell table in front of the quota phase - cost a KITTI level-0
  // problem more (+42 k cycles) than the four passes it saves (-26 k); measured, round 4.  Batches and 4K frames gain.
  if (dense && !b.no_hist && KEYCAP == 0) {
    // (cells of at least 2 x 2 px at the deepest level: no interval of a node that the tables meet has shrunk to a point, so
    // a node's left edge names its root and its path)
    int min_w = g.max_by - kMinBorder;
    for (int r = 0; r < n_ini; ++r) min_w = imin(min_w, g.root_x[r + 1] - g.root_x[r]);
    for (int d = 2; d <= 5; ++d)
      if (qt_hist_off(n_ini, d + 1) <= kHistEntries && (n_ini << (2 * d)) <= 0x10000 && (min_w >> d) >= 2) hist_depth = d;
  }
  bool from_codes = false;   // the keys' labels are cell codes (the breadth-first phase runs on the pyramid), not list positions
  if (!dense) {
    const uint32_t* ccnt = b.cell_cnt + (size_t)f * b.cells_frame + g.cell_off;
    const uint32_t* slots = b.slots + (size_t)f * b.slots_frame + g.slot_off;
    const int grp = tid / kGrp, gl = tid % kGrp;
    for (int c_lo = 0; c_lo < n_cells; c_lo += kPrefCap) {
      const int c_hi = c_lo + kPrefCap < n_cells ? c_lo + kPrefCap : n_cells;
      const int nc = c_hi - c_lo;
      __syncthreads();  // the previous chunk's readers are done (and the root nodes are set)
      uint32_t run = 0;
      for (int c0 = c_lo; c0 < c_hi; c0 += BS) {
        const int c = c0 + tid;
        const uint32_t cnt = c < c_hi ? ccnt[c] : 0u;
        uint32_t tot;
        const uint32_t ex = run + block_exclusive_scan<uint32_t>(cnt, s_scan, &tot);
        if (c < c_hi) s_pref[c - c_lo] = ex;
        run += tot;
      }
      if (tid == 0) s_pref[nc] = run;
      __syncthreads();
      auto take = [&](uint32_t key, uint32_t at) {
        const int r = root_of(key);
        keys[at] = key;
        label[at] = (uint16_t)r;
        qt_count_rep(s_rep0, r * 4 + qt_quadrant(key, S.mid[0][r]), rlog0);
      };
      for (int c0 = grp; c0 < nc; c0 += kTrip * kGroups) {
        // the first kPre * kGrp candidates of kTrip cells are requested before any is used (a cell holds ~30, rarely more
        // than 64): the slots come from HBM, a trip that waited for them piece by piece cost microseconds
        uint32_t bq[kTrip], nq[kTrip], kq[kTrip][kPre];
        const uint32_t* sq[kTrip];
#pragma unroll
        for (int t = 0; t < kTrip; ++t) {
          const int c = c0 + t * kGroups;
          const bool on = c < nc;
          bq[t] = on ? s_pref[c] : 0u;
          nq[t] = on ? s_pref[c + 1] - bq[t] : 0u;
          sq[t] = slots + (size_t)(c_lo + (on ? c : c0)) * cell_cap;
        }
#pragma unroll
        for (int t = 0; t < kTrip; ++t)
#pragma unroll
          for (int j = 0; j < kPre; ++j) {
            const uint32_t k = (uint32_t)(gl + kGrp * j);
            kq[t][j] = k < nq[t] ? sq[t][k] : 0u;
          }
#pragma unroll
        for (int t = 0; t < kTrip; ++t) {
#pragma unroll
          for (int j = 0; j < kPre; ++j) {
            const uint32_t k = (uint32_t)(gl + kGrp * j);
            if (k < nq[t]) take(kq[t][j], C + bq[t] + k);
          }
          for (uint32_t k = (uint32_t)(gl + kGrp * kPre); k < nq[t]; k += kGrp) take(sq[t][k], C + bq[t] + k);
        }
      }
      C += run;
    }
  } else {
    C = b.level_cnt[(size_t)f * n_levels + l];
    if (C > g.key_cap) C = g.key_cap;
    __syncthreads();  // the root nodes and the counter copies are set; everybody has read the counter
    if (tid == 0) { b.level_cnt_last[(size_t)f * n_levels + l] = C; b.level_cnt[(size_t)f * n_levels + l] = 0; }
    const uint32_t* kin = keys;   // the list k_fast_cells left in global memory
    const bool in_lds = KEYCAP > 0 && C <= (uint32_t)KEYCAP;   // workgroup-uniform
    if (in_lds) { keys = s_keys; label = s_label; }            // every later pass works on the LDS copy
    if (hist_depth >= 2) {
      // every key's cell at depth D = hist_depth as its label, counted into the pyramid's deepest level
      for (int i = tid; i < qt_hist_off(n_ini, hist_depth + 1); i += BS) s_hist[i] = 0;
      __syncthreads();
      const int offD = qt_hist_off(n_ini, hist_depth), ry1 = g.max_by - kMinBorder, rx1 = g.max_bx - kMinBorder;
      // The paths are functions of ONE coordinate each: a table per column (root | x path) and per row (y path), built once in
      // the second node list's counters - idle until the first rebuild - makes a key's code two LDS reads instead of 2 D rounds of
      // split-point arithmetic (a 4K level-0 problem: 159 k keys against 3 809 + 2 129 table entries).
      uint16_t* lut_x = reinterpret_cast<uint16_t*>(&S.cnt[1][0][0]);
      uint16_t* lut_y = lut_x + ((rx1 + 2) & ~1);
      const bool use_lut = (size_t)(((rx1 + 2) & ~1) + ry1 + 1) * 2 <= sizeof(S.cnt[1]);
      if (use_lut) {
        for (int x = tid; x <= rx1; x += BS) {
          int r = 0;
          for (int k = 1; k < n_ini; ++k) r += x >= s_rootfirst[k] ? 1 : 0;
          lut_x[x] = (uint16_t)((r << (2 * hist_depth)) | (qt_path(x, s_rootx[r], s_rootx[r + 1], hist_depth) << hist_depth));
        }
        for (int y = tid; y <= ry1; y += BS) lut_y[y] = (uint16_t)qt_path(y, 0, ry1, hist_depth);
        __syncthreads();
      }
      for (uint32_t i0 = 0; i0 < C; i0 += BS * kQtBatch) {
        uint32_t key[kQtBatch];
#pragma unroll
        for (int u = 0; u < kQtBatch; ++u) {
          const uint32_t i = i0 + (uint32_t)(u * BS + tid);
          key[u] = i < C ? kin[i] : 0xffffffffu;
        }
#pragma unroll
        for (int u = 0; u < kQtBatch; ++u)
          if (key[u] != 0xffffffffu) {
            int code;
            if (use_lut) {
              code = (int)lut_x[imin(key_x(key[u]), rx1)] | (int)lut_y[imin(key_y(key[u]), ry1)];
            } else {
              const int r = root_of(key[u]);
              code = (r << (2 * hist_depth)) | (qt_path(key_x(key[u]), s_rootx[r], s_rootx[r + 1], hist_depth) << hist_depth) |
                     qt_path(key_y(key[u]), 0, ry1, hist_depth);
            }
            if (in_lds) keys[i0 + (uint32_t)(u * BS + tid)] = key[u];
            label[i0 + (uint32_t)(u * BS + tid)] = (uint16_t)code;
            atomicAdd(&s_hist[offD + code], 1u);
          }
      }
      __syncthreads();   // (the tables' last readers are done before anybody writes that list's counters)
      // the shallower levels: sums of 2 x 2 cells
      for (int d = hist_depth - 1; d >= 1; --d) {
        const int off = qt_hist_off(n_ini, d), offc = qt_hist_off(n_ini, d + 1), cells = n_ini << (2 * d);
        for (int e = tid; e < cells; e += BS) {
          const int r = e >> (2 * d), xp = (e >> d) & ((1 << d) - 1), yp = e & ((1 << d) - 1);
          const int c = offc + (r << (2 * d + 2)) + ((2 * xp) << (d + 1)) + 2 * yp;
          s_hist[off + e] = s_hist[c] + s_hist[c + 1] + s_hist[c + (1 << (d + 1))] + s_hist[c + (1 << (d + 1)) + 1];
        }
        __syncthreads();
      }
      if (tid < 4 * n_ini) S.cnt[0][tid >> 2][tid & 3] = s_hist[((tid >> 2) << 2) + ((tid & 1) << 1) + ((tid >> 1) & 1)];   // level 1: (root << 2) | (x bit << 1) | y bit
      from_codes = true;
    } else
    for (uint32_t i0 = 0; i0 < C; i0 += BS * kQtBatch) {
      uint32_t key[kQtBatch];
#pragma unroll
      for (int u = 0; u < kQtBatch; ++u) {
        const uint32_t i = i0 + (uint32_t)(u * BS + tid);
        key[u] = i < C ? kin[i] : 0xffffffffu;
      }
#pragma unroll
      for (int u = 0; u < kQtBatch; ++u)
        if (key[u] != 0xffffffffu) {
          const int r = root_of(key[u]);
          if (in_lds) keys[i0 + (uint32_t)(u * BS + tid)] = key[u];
          label[i0 + (uint32_t)(u * BS + tid)] = (uint16_t)r;
          qt_count_rep(s_rep0, r * 4 + qt_quadrant(key[u], S.mid[0][r]), rlog0);
        }
    }
  }
  __syncthreads();
  if (!from_codes && tid < 4 * n_ini) {
    uint32_t v = 0;
    for (int j = 0; j < (1 << rlog0); ++j) v += s_rep0[(tid << rlog0) + j];
    S.cnt[0][tid >> 2][tid & 3] = v;
  }
  __syncthreads();
  RGBL_STAMP(1);
  if (tid == 0) {  // empty roots are erased (ORBextractor.cc:597-606)
    int n0 = 0;
    for (int r = 0; r < n_ini; ++r) {
      s_rootpos[r] = n0;
      if (S.cnt[0][r][0] + S.cnt[0][r][1] + S.cnt[0][r][2] + S.cnt[0][r][3] > 0) {
        if (n0 != r) {
          S.geom[0][n0] = S.geom[0][r];
          S.mid[0][n0] = S.mid[0][r];
          S.cnt[0][n0][0] = S.cnt[0][r][0]; S.cnt[0][n0][1] = S.cnt[0][r][1]; S.cnt[0][n0][2] = S.cnt[0][r][2]; S.cnt[0][n0][3] = S.cnt[0][r][3];
        }
        ++n0;
      }
    }
    s_n = n0;
  }
  __syncthreads();
  int n = s_n;
  if (n != n_ini && !from_codes) {  // wave-uniform, rare: the labels of the roots behind an empty one move up (cell codes name the root itself)
    for (uint32_t i = (uint32_t)tid; i < C; i += BS) label[i] = (uint16_t)s_rootpos[label[i]];
  }
  __syncthreads();
  RGBL_STAMP(2);

  // ---- 2. rounds of splits: breadth-first (ORBextractor.cc:608-686), then - once another full round would overshoot
  //         the quota - the most populated nodes first, one at a time in the reference, up to the node that reaches
  //         the quota (ORBextractor.cc:689-753)
  unsigned long long* best = reinterpret_cast<unsigned long long*>(&S.cnt[1][0][0]);
  int p = 0, m = 0;
  int bfs_depth = 0;   // depth of the expandable nodes while the rounds run on the pyramid
  bool finished = (n == 0), careful = false, stamped = false;
  const int w_cell = g.w_cell, h_cell = g.h_cell, n_cols = g.n_cols;
  const uint32_t m_wcell = g.m_wcell, m_hcell = g.m_hcell;
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
      int carry = 0;
      for (int r0 = 0; r0 < m; r0 += BS) {
        const int rho = r0 + tid;
        int v = 0;
        if (rho < m) {
          const uint32_t* c = S.cnt[p][S.sval[m - 1 - rho]];
          v = (int)((c[0] > 0) + (c[1] > 0) + (c[2] > 0) + (c[3] > 0)) - 1;
        }
        uint32_t tot;
        const uint32_t ex = block_exclusive_scan<uint32_t>((uint32_t)v, s_scan, &tot);  // modulo 2^32: the partial sums are small ints
        const int size_after = n + carry + (int)ex + v;
        if (rho < m && size_after >= N) atomicMin(&s_P, rho + 1);
        carry += (int)tot;
      }
      __syncthreads();
      P = s_P;
    }
    // A round on the pyramid alone: breadth-first (every expandable node splits, all of them at depth `bfs_depth`) and the
    // children's quadrant counts - level bfs_depth + 2 - still inside it.  Otherwise, if the labels are still cell codes, the
    // keys meet their nodes of the CURRENT list through the table first (its space: the pyramid's deepest level, read for the
    // last time by the previous round's children).
    const bool node_round = from_codes && !careful && bfs_depth + 2 <= hist_depth;
    const int offD = qt_hist_off(n_ini, hist_depth);
    auto cells_to_positions = [&](int list, int count) {   // every depth-D cell -> the list position of the node that covers it
      const int ry1 = g.max_by - kMinBorder, nw = BS / 64;
      for (int pos = wave_id(); pos < count; pos += nw) {
        const uint2 G = S.geom[list][pos];
        const int dn = qt_depth_of(S.mid[list][pos]), x0 = (int)(G.x & 0xffffu), y0 = (int)(G.y & 0xffffu);
        int r = 0;
        for (int k = 1; k < n_ini; ++k) r += x0 >= s_rootx[k] ? 1 : 0;
        const int sh = hist_depth - dn;   // the node covers 2^sh x 2^sh cells (dn <= hist_depth while the labels are codes)
        const int base = offD + (r << (2 * hist_depth)) + ((qt_path(x0, s_rootx[r], s_rootx[r + 1], dn) << sh) << hist_depth) + (qt_path(y0, 0, ry1, dn) << sh);
        for (int c = lane_id(); c < (1 << (2 * sh)); c += 64) s_hist[base + ((c >> sh) << hist_depth) + (c & ((1 << sh) - 1))] = (uint32_t)pos;
      }
      __syncthreads();
    };
    if (from_codes && !node_round) cells_to_positions(p, n);
    qt_rebuild<BS, NCAP>(S, p, n, P, m, !careful, cap, s_scan, &s_newn, &s_nexp, &s_nchild);
    n = s_newn;
    m = s_nexp;
    if (n > cap - 8) { if (tid == 0) atomicOr(b.err, 1); finished = true; }
    else if (n >= N || n == prev) finished = true;
    else if (!careful && n + 3 * m > N) careful = true;
    if (!stamped && (careful || finished)) { RGBL_STAMP(3); stamped = true; }
    if (node_round && !finished) {
      // the children of this round (the front of the new list) take their quadrant counts from the pyramid; no key is touched
      const int T = s_nchild, ry1 = g.max_by - kMinBorder, dc = bfs_depth + 1, offq = qt_hist_off(n_ini, dc + 1);
      for (int idx = tid; idx < T && idx < cap; idx += BS) {
        const uint2 G = S.geom[1 - p][idx];
        const int x0 = (int)(G.x & 0xffffu), y0 = (int)(G.y & 0xffffu);
        int r = 0;
        for (int k = 1; k < n_ini; ++k) r += x0 >= s_rootx[k] ? 1 : 0;
        const int xp = qt_path(x0, s_rootx[r], s_rootx[r + 1], dc), yp = qt_path(y0, 0, ry1, dc);
        const int c = offq + (r << (2 * dc + 2)) + ((2 * xp) << (dc + 1)) + 2 * yp;
        S.cnt[1 - p][idx][0] = s_hist[c]; S.cnt[1 - p][idx][2] = s_hist[c + 1];
        S.cnt[1 - p][idx][1] = s_hist[c + (1 << (dc + 1))]; S.cnt[1 - p][idx][3] = s_hist[c + (1 << (dc + 1)) + 1];
      }
      ++bfs_depth;
      p ^= 1;
      __syncthreads();
      continue;
    }
    // a pyramid round that ended the distribution: the keys meet the nodes of the NEW list directly
    const bool via_new_list = node_round;   // (finished)
    if (via_new_list) cells_to_positions(1 - p, n < cap ? n : cap);
    // copies of the new list's counters (in the old list's counters, which nobody reads any more) while it is short
    uint32_t* s_rep = &S.cnt[p][0][0];
    const int rlog = (finished || 4 * n > NCAP) ? 0 : qt_rep_log(4 * n, 4 * NCAP);
    // the last round: per node the strongest key, on ties the first one of the reference's candidate list (cell after
    // cell, row-major inside a cell): response << 56 | (2^28 - 1 - order) << 24 | coordinates, one LDS maximum per key
    best = reinterpret_cast<unsigned long long*>(&S.cnt[1 - p][0][0]);
    if (finished) {
      __syncthreads();  // qt_rebuild's last writes into that buffer are done
      for (int pos = tid; pos < NCAP; pos += BS) best[pos] = 0ull;
      __syncthreads();
    } else if (rlog > 0) {
      for (int i = tid; i < ((4 * n) << rlog); i += BS) s_rep[i] = 0;
      __syncthreads();
    }
    // every key: the list position of its node in the new list (one table entry per quadrant of its old node); keys
    // of a fresh child with more than one key are counted into that child's quadrants - unless this was the last round
    for (uint32_t i0 = 0; i0 < C; i0 += BS * kQtBatch) {
      uint32_t keyv[kQtBatch], oldv[kQtBatch];
#pragma unroll
      for (int u = 0; u < kQtBatch; ++u) {
        const uint32_t i = i0 + (uint32_t)(u * BS + tid);
        keyv[u] = i < C ? keys[i] : 0u;
        oldv[u] = i < C ? (uint32_t)label[i] : 0xffffffffu;
      }
      uint2 tabv[kQtBatch];
      uint32_t midv[kQtBatch];
      if (from_codes) {   // workgroup-uniform: the labels are cell codes, the table names the node (of the old list, or - via_new_list - of the new one)
#pragma unroll
        for (int u = 0; u < kQtBatch; ++u)
          if (oldv[u] != 0xffffffffu) oldv[u] = s_hist[offD + (int)oldv[u]];
      }
#pragma unroll
      for (int u = 0; u < kQtBatch; ++u) {
        const uint32_t o = oldv[u] & (uint32_t)(NCAP - 1);
        tabv[u] = S.ctab[o];
        midv[u] = S.mid[p][o];
      }
#pragma unroll
      for (int u = 0; u < kQtBatch; ++u) {
        const uint32_t i = i0 + (uint32_t)(u * BS + tid);
        const bool valid = oldv[u] != 0xffffffffu;
        const int q = qt_quadrant(keyv[u], midv[u]);
        const uint32_t e = via_new_list ? oldv[u] : (((q & 2) ? tabv[u].y : tabv[u].x) >> ((q & 1) * 16)) & 0xffffu;
        const uint32_t idx = e & 0x7fffu;
        if (valid && (from_codes || idx != oldv[u])) label[i] = (uint16_t)idx;
        if (finished) {
          // ---- 3. the strongest key of every node, the first one of the candidate list on ties (ORBextractor.cc:757-776)
          if (valid && idx < (uint32_t)cap) {
            uint32_t order = i;  // cell slots gathered in the reference's order
            if (dense) {
              // FAST scans a cell from its pixel 3: cell = (coordinate - 3) / cell size (the scanned width never exceeds it)
              const uint32_t x = (uint32_t)key_x(keyv[u]) - 3u, y = (uint32_t)key_y(keyv[u]) - 3u;
              const uint32_t col = __umul24(x, m_wcell) >> 20, row = __umul24(y, m_hcell) >> 20;
              order = ((row * (uint32_t)n_cols + col) << 14) | ((y - row * (uint32_t)h_cell) << 7) | (x - col * (uint32_t)w_cell);
            }
            atomicMax(&best[idx], ((unsigned long long)key_s(keyv[u]) << 56) | ((unsigned long long)(0xfffffffu - order) << 24) |
                                      (unsigned long long)(keyv[u] & 0xffffffu));
          }
        } else {
          int slot = -1;
          if (valid && (e & 0x8000u) && idx < (uint32_t)cap) slot = (int)idx * 4 + qt_quadrant(keyv[u], S.mid[1 - p][idx]);
          if (rlog > 0) qt_count_rep(s_rep, slot, rlog);
          else qt_count(&S.cnt[1 - p][0][0], slot);
        }
      }
    }
    if (rlog > 0) {  // wave-uniform
      __syncthreads();
      for (int i = tid; i < 4 * n; i += BS) {
        uint32_t v = 0;
        for (int j = 0; j < (1 << rlog); ++j) v += s_rep[(i << rlog) + j];
        if (v) S.cnt[1 - p][i >> 2][i & 3] += v;  // fresh children start at zero, nodes that were not split receive nothing
      }
    }
    from_codes = false;   // the labels are list positions from here on
    p ^= 1;
    __syncthreads();
  }
  if (!stamped) RGBL_STAMP(3);
  RGBL_STAMP(4);

  uint32_t* out = b.kp_key + (size_t)f * b.kp_frame + g.koff;
  if (n > kcap) { if (tid == 0) atomicOr(b.err, 2); n = kcap; }
  if (n > cap) n = cap;
  for (int pos = tid; pos < n; pos += BS) {
    const unsigned long long v = best[pos];
    out[pos] = (uint32_t)(v >> 56) << 24 | (uint32_t)(v & 0xffffffull);
  }
  if (tid == 0) b.kp_count[(size_t)f * n_levels + l] = n;
  RGBL_STAMP(5);
  if (b.dbg && tid == 0) { b.dbg[((size_t)f * n_levels + l) * 16 + 6] = C; b.dbg[((size_t)f * n_levels + l) * 16 + 7] = (unsigned long long)n; }
#undef RGBL_STAMP
}

}  // namespace rgbl
