// extractor.hip — host side of the ORB extractor: geometry/tables, HBM scratch, launch sequence, C ABI.
//
// Replaces ORB_SLAM3::ORBextractor (/root/reference/include/ORBextractor.h:49-83,
// /root/reference/src/ORBextractor.cc:409-469, 1086-1195).  The device kernels are in extractor_kernels.h.
//
// HBM layout (all buffers are per handle, sized for max_batch frames; frame f at base + f*frame_stride):
//   pyr    : levels 1..L-1 of the 8-bit pyramid, each level padded to a 64-byte pitch (level 0 is read
//            straight from the caller's image, it is never copied)
//   blur   : levels 0..L-1 of the Gaussian working images, same pitches
//   cells  : one candidate counter per FAST detection cell + cell_cap packed candidates per cell
//   keys   : two ping-pong arrays of packed keys per level for the quad-tree partitions
//   nodes  : quad-tree node lists (ping-pong), split counts, expandable lists, sort scratch
//   kp     : selected keys per level (kcap each) and the per-(frame,level) counts
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "brief_pattern.h"
#include "extractor_kernels.h"
#include "octree_labels.h"

namespace rgbl {

// ---- thread-local error message ------------------------------------------------------------------
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* last_error() { return g_err; }

static inline int round_even_f(float v) { return (int)lrintf(v); }
static inline int round_even_d(double v) { return (int)lrint(v); }
static inline int floor_f(float v) { int i = (int)v; return i - (v < (float)i); }
static inline int ceil_f(float v) { int i = (int)v; return i + (v > (float)i); }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace rgbl

using namespace rgbl;

struct rgbl_extractor {
  rgbl_extractor_cfg cfg;
  int device = 0;
  int L = 0;
  hipStream_t stream = nullptr, own_stream = nullptr;
  hipStream_t aux_stream = nullptr;  // the Gaussian working images only depend on the pyramid: they overlap FAST + quad-tree
  hipStream_t lvl_stream = nullptr;  // single frames: FAST + quad-tree of the levels 1 - 2 start behind their own resizes (RGBL_LEVEL_SPLIT=0: off)
  int level_split = 3;               // ... the main stream keeps the levels from this one on
  hipEvent_t ev_pyr = nullptr, ev_blur = nullptr, ev_start = nullptr, ev_fast0 = nullptr, ev_desc0 = nullptr, ev_r = nullptr, ev_fb = nullptr;
  int split_pyr = 0;  // batches: the pyramid levels k .. L - 1 and their FAST cells leave the main chain (default L / 2 from 6 levels on; RGBL_SPLIT_PYR=k, 0 = off)
  bool gauss_wg256 = false;     // RGBL_GAUSS_BS=256: four-wave workgroups for k_gauss7 (two waves measured faster)
  bool octree_stamps = false;   // RGBL_OCTREE_STAMPS: the quad-tree kernel leaves phase time stamps (tools/octree_stamps.py)
  bool octree_no_hist = false;  // RGBL_OCTREE_HIST=0: breadth-first rounds as passes over the keys instead of on the count pyramid
  KernelTimer timer;
  // hipGraph of the host-pointer path (all device pointers of that path are the handle's own buffers, so one captured
  // launch sequence can be replayed): key = (batch, row stride, lapping area, stream)
#ifndef RGBL_EMU
  hipGraphExec_t graph_exec = nullptr;
#endif
  int graph_batch = 0, graph_stride = 0, graph_lap0 = 0, graph_lap1 = 0;
  hipStream_t graph_stream = nullptr;
  bool graph_ok = true;  // RGBL_GRAPH=0 or a failed capture switch the replay off
  int octree_wg = 0;  // 0 = choose per launch; RGBL_OCTREE_WG=256|512 pins the quad-tree workgroup width (tuning / tests)
  bool octree_ldskeys = true;  // single frames: k_octree keeps the candidate lists in LDS (RGBL_OCTREE_LDSKEYS=0 switches it off)
  int octree_ncap = 0;  // LDS node capacity of the label-based quad-tree kernel (512 / 2048); 0 = key-moving kernel on global lists
  int max_cell = 0, max_cell_w = 0;  // largest detection-cell side / width over the levels: select the k_fast_cells instantiation
  int fast_waves = 0;  // waves per detection cell of k_fast_cells: 0 = by batch size, RGBL_FAST_BS=64 / 128 force 1 / 2
  bool xcd_map = true;  // XCD-aware workgroup -> (item, frame) mapping of the pixel kernels (common.h: xcd_item_frame); RGBL_XCD_MAP=0 switches it off
  std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
  std::vector<int> per_level;
  UMax umax;
  std::vector<LevelGeom> geom;
  BlurTiles blur_tiles;
  int cells_frame = 0, kp_frame = 0;
  size_t pyr_frame = 0, slots_frame = 0, keys_frame = 0, nodes_frame = 0, img_frame = 0;
  int img_pitch = 0;
  // device memory
  LevelGeom* d_geom = nullptr;
  FastCell* d_cells = nullptr;  // one record per detection cell of a frame (k_fast_cells)
  CellGroup* d_groups = nullptr;          // groups of up to 256 consecutive cells of one level (k_compact_cells)
  std::vector<int> group_off;             // [L + 1] first group of every level
  int compact_min_batch = 8;              // batches of at least this many frames: FAST cells write their own slots, k_compact_cells
                                          // builds the dense lists (RGBL_COMPACT=0: never, =1: always)
  GaussTile* d_gtiles = nullptr;  // one record per Gaussian output tile of a frame (k_gauss7)
  ResizeTab *d_xtab = nullptr, *d_ytab = nullptr;
  ResizeGroup* d_xgroups = nullptr;  // k_resize_linear: one record per 4 output columns (index xtab_off / 4 + group)
  int32_t* d_xsxa = nullptr;         // first source byte of the group's 8-byte window, -1 = byte path
  uint8_t* d_rootx = nullptr;
  int8_t* d_pattern = nullptr;
  uint8_t *d_img = nullptr, *d_pyr = nullptr, *d_blur = nullptr;
  uint32_t *d_cellcnt = nullptr, *d_slots = nullptr, *d_keys_a = nullptr, *d_keys_b = nullptr;
  QNode *d_list_a = nullptr, *d_list_b = nullptr;
  QDiv* d_div = nullptr;
  uint32_t *d_todo_a = nullptr, *d_todo_b = nullptr, *d_sval = nullptr;
  uint64_t* d_skey = nullptr;
  uint8_t* d_divided = nullptr;
  uint32_t* d_kpkey = nullptr;
  int* d_kpcount = nullptr;
  uint8_t* h_pinned = nullptr;      // page-locked result block of the host-pointer path: counts, flags, keypoints, descriptors of small batches
  size_t h_pinned_bytes = 0;
  uint32_t* d_levelcnt = nullptr;  // [B][L] candidates per level, counted by k_fast_cells' cells (dense candidate lists)
  bool dense = false;              // k_fast_cells writes a level's candidates as one list (label-based quad-tree kernel, separate pixel kernels)
  bool dense_dirty = false;        // an enqueue failed between the FAST and the quad-tree launches: the counters may hold leftovers
  int* d_err = nullptr;            // (the first word of d_stage)
  uint8_t* d_stage = nullptr;      // error flags | d_out_n | d_out_mono | d_out_kp | d_out_desc, the layout of h_pinned
  int32_t* d_stereo_sad = nullptr;  // ComputeStereoMatches scratch (grow-only)
  size_t stereo_sad_count = 0;
  void* d_stereo_stage = nullptr;
  void* d_color = nullptr;  // staging of one host colour frame (rgbl_extract_color), allocated on first use
  size_t color_bytes = 0;
  size_t stereo_stage_bytes = 0;
  uint8_t* h_stereo_stage = nullptr;   // page-locked mirror of d_stereo_stage (rgbl_stereo_matches: one request per direction)
  unsigned long long* d_dbg = nullptr;
  // staging for the host entry points and for the lapping permutation
  rgbl_keypoint *d_out_kp = nullptr, *d_tmp_kp = nullptr;
  uint8_t *d_out_desc = nullptr, *d_tmp_desc = nullptr;
  int32_t *d_out_n = nullptr, *d_out_mono = nullptr;
  int out_cap = 0;
  // rgbl_extract_begin: an extraction of one frame whose results are on their way into the page-locked block
  struct { bool active = false; const uint8_t* img = nullptr; int w = 0, h = 0, stride = 0, lap0 = 0, lap1 = 0; } pending;
  // last call (for get_level / get_candidates)
  const uint8_t* last_img0 = nullptr;
  int last_pitch0 = 0;
  size_t last_frame0 = 0;
  int last_batch = 0;
  std::vector<void*> allocs;
};

namespace {

template <class T>
int dev_alloc(rgbl_extractor* e, T** p, size_t count) {
  RGBL_HIP(hipMalloc(p, std::max<size_t>(count, 1) * sizeof(T)));
  e->allocs.push_back((void*)*p);
  return RGBL_OK;
}

// cv::resize's coefficient tables for one axis (modules/imgproc/src/resize.cpp, INTER_LINEAR, fixed point)
void build_resize_tab(int ssize, int dsize, bool clamp_x, std::vector<ResizeTab>& tab) {
  const double inv_scale = (double)dsize / ssize;
  const double scale = 1.0 / inv_scale;
  for (int d = 0; d < dsize; ++d) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = floor_f(f);
    f -= s;
    if (clamp_x) {
      if (s < 0) { f = 0; s = 0; }
      if (s >= ssize - 1) { f = 0; s = ssize - 1; }
    }
    ResizeTab t;
    t.sofs = s;
    t.a0 = (int16_t)round_even_f((1.f - f) * 2048);
    t.a1 = (int16_t)round_even_f(f * 2048);
    tab.push_back(t);
  }
}

int build_geometry(rgbl_extractor* e) {
  const rgbl_extractor_cfg& c = e->cfg;
  const int L = c.nlevels;
  e->L = L;
  // ---- ORBextractor.cc:414-445 (all fp32)
  e->scale.assign(L, 1.f); e->sigma2.assign(L, 1.f); e->inv_scale.assign(L, 1.f); e->inv_sigma2.assign(L, 1.f);
  for (int i = 1; i < L; ++i) {
    e->scale[i] = e->scale[i - 1] * c.scale_factor;
    e->sigma2[i] = e->scale[i] * e->scale[i];
  }
  for (int i = 0; i < L; ++i) {
    e->inv_scale[i] = 1.0f / e->scale[i];
    e->inv_sigma2[i] = 1.0f / e->sigma2[i];
  }
  e->per_level.assign(L, 0);
  const float factor = 1.0f / c.scale_factor;
  float desired = c.nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)L));
  int sum = 0;
  for (int l = 0; l < L - 1; ++l) {
    e->per_level[l] = round_even_f(desired);
    sum += e->per_level[l];
    desired *= factor;
  }
  e->per_level[L - 1] = std::max(c.nfeatures - sum, 0);
  // ---- umax, ORBextractor.cc:451-468
  {
    int* u = e->umax.v;
    for (int i = 0; i < 16; ++i) u[i] = 0;
    const int vmax = floor_f(15 * sqrtf(2.f) / 2 + 1), vmin = ceil_f(15 * sqrtf(2.f) / 2);
    for (int v = 0; v <= vmax; ++v) u[v] = round_even_d(sqrt(225.0 - v * v));
    for (int v = 15, v0 = 0; v >= vmin; --v) {
      while (u[v0] == u[v0 + 1]) ++v0;
      u[v] = v0;
      ++v0;
    }
  }
  // ---- per-level geometry
  e->geom.assign(L, LevelGeom());
  size_t img_off = 0, slot_off = 0, node_off = 0, rootx_off = 0, xtab_off = 0, ytab_off = 0;
  int cell_off = 0, koff = 0;
  e->img_pitch = (int)align_up(c.width, 64);
  e->img_frame = (size_t)e->img_pitch * c.height;
  int tile_off = 0;
  for (int l = 0; l < L; ++l) {
    LevelGeom& g = e->geom[l];
    memset(&g, 0, sizeof(g));
    g.w = round_even_f((float)c.width * e->inv_scale[l]);   // ORBextractor.cc:1174-1175
    g.h = round_even_f((float)c.height * e->inv_scale[l]);
    g.pitch = (int)align_up(g.w, 64);
    g.img_off = (uint32_t)img_off;
    img_off += (size_t)g.pitch * g.h;
    g.quota = e->per_level[l];
    g.scale = e->scale[l];
    g.patch_size = (int)(31 * e->scale[l]);
    // detection grid, ORBextractor.cc:789-803
    g.max_bx = g.w - 19 + 3;
    g.max_by = g.h - 19 + 3;
    const float width = (float)(g.max_bx - kMinBorder), height = (float)(g.max_by - kMinBorder);
    g.n_cols = (int)(width / 35.f);
    g.n_rows = (int)(height / 35.f);
    if (g.n_cols < 1 || g.n_rows < 1) {
      set_error("level %d (%dx%d) is smaller than one 35-px detection cell; the reference divides by zero here", l, g.w, g.h);
      return RGBL_ERR_INVALID;
    }
    g.w_cell = (int)ceilf(width / g.n_cols);
    g.h_cell = (int)ceilf(height / g.n_rows);
    if (g.w_cell > kCellMax || g.h_cell > kCellMax) {
      set_error("cell %dx%d exceeds the kernel tile", g.w_cell, g.h_cell);
      return RGBL_ERR_INVALID;
    }
    e->max_cell = std::max(e->max_cell, std::max(g.w_cell, g.h_cell));
    e->max_cell_w = std::max(e->max_cell_w, g.w_cell);
    g.n_cells = g.n_cols * g.n_rows;
    g.m_wcell = (0x100000u + (uint32_t)g.w_cell - 1u) / (uint32_t)g.w_cell;
    g.m_hcell = (0x100000u + (uint32_t)g.h_cell - 1u) / (uint32_t)g.h_cell;
    g.cell_off = cell_off;
    cell_off += g.n_cells;
    g.cell_cap = ((g.w_cell + 1) / 2) * ((g.h_cell + 1) / 2);  // strict 3x3 NMS keeps at most one pixel per 2x2
    g.slot_off = (uint32_t)slot_off;
    g.key_off = (uint32_t)slot_off;
    g.key_cap = (uint32_t)((size_t)g.n_cells * g.cell_cap);
    slot_off += (size_t)g.n_cells * g.cell_cap;
    // quad-tree roots, ORBextractor.cc:558-577
    g.n_ini = (int)roundf((float)(g.max_bx - kMinBorder) / (float)(g.max_by - kMinBorder));
    if (g.n_ini < 1 || g.n_ini > kMaxRoots) {
      set_error("aspect ratio gives %d quad-tree roots (supported 1..%d; the reference fails for portrait images)", g.n_ini, kMaxRoots);
      return RGBL_ERR_INVALID;
    }
    const float hX = (float)(g.max_bx - kMinBorder) / g.n_ini;
    for (int i = 0; i <= g.n_ini; ++i) g.root_x[i] = (int)(hX * (float)i);
    g.rootx_off = (uint32_t)rootx_off;
    rootx_off += align_up((size_t)(g.max_bx - kMinBorder) + 1, 16);
    g.kcap = std::max(g.quota, 4 * g.n_ini) + 8;
    g.koff = koff;
    koff += g.kcap;
    g.node_cap = (uint32_t)g.kcap + 16;
    if (g.node_cap > 65535u) {  // list positions travel as 16-bit payloads in the quad-tree's sort
      set_error("nfeatures too large: level %d would need %u quad-tree nodes (limit 65535)", l, g.node_cap);
      return RGBL_ERR_INVALID;
    }
    g.node_off = (uint32_t)node_off;
    node_off += g.node_cap;
    g.xtab_off = (uint32_t)xtab_off;
    g.ytab_off = (uint32_t)ytab_off;
    if (l > 0) { xtab_off += align_up(g.w, 4); ytab_off += g.h; }  // 32-byte aligned x tables (one read per 4 columns)
    e->blur_tiles.tile_off[l] = tile_off;
    e->blur_tiles.tiles_x[l] = (g.w + kBlurTW - 1) / kBlurTW;
    tile_off += e->blur_tiles.tiles_x[l] * ((g.h + kBlurTH - 1) / kBlurTH);
  }
  e->blur_tiles.tile_off[L] = tile_off;
  e->pyr_frame = align_up(img_off, 256);
  e->cells_frame = cell_off;
  e->slots_frame = slot_off;
  e->keys_frame = slot_off;
  e->nodes_frame = node_off;
  e->kp_frame = koff;
  return RGBL_OK;
}

int upload_tables(rgbl_extractor* e) {
  const int L = e->L;
  std::vector<ResizeTab> xt, yt;
  std::vector<uint8_t> rootx;
  for (int l = 0; l < L; ++l) {
    const LevelGeom& g = e->geom[l];
    if (l > 0) {
      xt.resize(g.xtab_off);
      build_resize_tab(e->geom[l - 1].w, g.w, true, xt);
      while (xt.size() % 4) xt.push_back(xt.back());  // the partial last group of a row computes (and stores into the row padding) copies of the last column
      build_resize_tab(e->geom[l - 1].h, g.h, false, yt);
    }
    // root node of every x (ORBextractor.cc:585: vpIniNodes[kp.pt.x / hX], float division, truncation)
    const int width = g.max_bx - kMinBorder;
    const float hX = (float)width / g.n_ini;
    rootx.resize(g.rootx_off + align_up((size_t)width + 1, 16), 0);
    LevelGeom& gm = e->geom[l];
    for (int k = 0; k <= kMaxRoots; ++k) gm.root_first[k] = 0x7fffffff;
    for (int x = 0; x <= width; ++x) {
      int r = (int)((float)x / hX);
      r = std::min(r, g.n_ini - 1);
      rootx[g.rootx_off + x] = (uint8_t)r;
      for (int k = 1; k <= r; ++k) gm.root_first[k] = std::min(gm.root_first[k], x);  // the table is monotone in x
    }
  }
  // detection cells (ORBextractor.cc:789-822), level after level
  std::vector<FastCell> cells((size_t)e->cells_frame);
  for (int l = 0; l < L; ++l) {
    const LevelGeom& g = e->geom[l];
    for (int ci = 0; ci < g.n_cells; ++ci) {
      FastCell c;
      memset(&c, 0, sizeof(c));
      const int row = ci / g.n_cols, col = ci - row * g.n_cols;
      const int ini_x = kMinBorder + col * g.w_cell, ini_y = kMinBorder + row * g.h_cell;
      const int max_x = std::min(ini_x + g.w_cell + 6, g.max_bx), max_y = std::min(ini_y + g.h_cell + 6, g.max_by);
      const int tw = max_x - ini_x, th = max_y - ini_y, sw = tw - 6, sh = th - 6;
      c.ini_x = (uint16_t)ini_x; c.ini_y = (uint16_t)ini_y;
      c.kx0 = (uint16_t)(col * g.w_cell + 3); c.ky0 = (uint16_t)(row * g.h_cell + 3);
      c.l = (uint8_t)l;
      c.skip = (ini_x >= g.max_bx - 6 || ini_y >= g.max_by - 3 || sw <= 0 || sh <= 0) ? 1 : 0;
      c.pitch = (uint16_t)g.pitch; c.cell_cap = (uint16_t)g.cell_cap;
      c.img_off = g.img_off;
      c.slot_base = g.slot_off + (uint32_t)ci * (uint32_t)g.cell_cap;
      if (!c.skip) {
        const uint32_t nwords = (uint32_t)(tw + 3) >> 2;
        c.tw = (uint8_t)tw; c.th = (uint8_t)th;
        c.magic = (0x100000u + (uint32_t)sw - 1u) / (uint32_t)sw;
        c.wmagic = (0x100000u + nwords - 1u) / nwords;
      }
      cells[(size_t)g.cell_off + ci] = c;
    }
  }
  std::vector<GaussTile> gtiles((size_t)e->blur_tiles.tile_off[L]);
  for (int l = 0; l < L; ++l) {
    const LevelGeom& g = e->geom[l];
    const int tx_n = e->blur_tiles.tiles_x[l], n = e->blur_tiles.tile_off[l + 1] - e->blur_tiles.tile_off[l];
    for (int t = 0; t < n; ++t) {
      GaussTile gt;
      memset(&gt, 0, sizeof(gt));
      gt.x0 = (uint16_t)((t % tx_n) * kBlurTW); gt.y0 = (uint16_t)((t / tx_n) * kBlurTH);
      gt.w = (uint16_t)g.w; gt.h = (uint16_t)g.h; gt.pitch = (uint16_t)g.pitch; gt.l = (uint8_t)l; gt.img_off = g.img_off;
      gtiles[(size_t)e->blur_tiles.tile_off[l] + t] = gt;
    }
  }
  RGBL_TRY(dev_alloc(e, &e->d_gtiles, gtiles.size()));
  RGBL_HIP(hipMemcpy(e->d_gtiles, gtiles.data(), sizeof(GaussTile) * gtiles.size(), hipMemcpyHostToDevice));
  RGBL_TRY(dev_alloc(e, &e->d_cells, cells.size()));
  RGBL_HIP(hipMemcpy(e->d_cells, cells.data(), sizeof(FastCell) * cells.size(), hipMemcpyHostToDevice));
  {
    // groups of consecutive cells for k_compact_cells: never across a level (one counter, one list per level) - the launch
    // ranges of the FAST kernel are whole levels, so they are whole groups as well
    std::vector<CellGroup> groups;
    e->group_off.assign(L + 1, 0);
    for (int l = 0; l < L; ++l) {
      const LevelGeom& g = e->geom[l];
      e->group_off[l] = (int)groups.size();
      for (int ci = 0; ci < g.n_cells; ci += kCompactCells)
        groups.push_back(CellGroup{(uint32_t)(g.cell_off + ci), (uint32_t)std::min(kCompactCells, g.n_cells - ci)});
    }
    e->group_off[L] = (int)groups.size();
    RGBL_TRY(dev_alloc(e, &e->d_groups, groups.size()));
    RGBL_HIP(hipMemcpy(e->d_groups, groups.data(), sizeof(CellGroup) * groups.size(), hipMemcpyHostToDevice));
    if (const char* v = getenv("RGBL_COMPACT")) e->compact_min_batch = atoi(v) ? 1 : 0x7fffffff;
  }
  RGBL_TRY(dev_alloc(e, &e->d_geom, L));
  RGBL_TRY(dev_alloc(e, &e->d_xtab, xt.size() + 8));
  RGBL_TRY(dev_alloc(e, &e->d_ytab, yt.size()));
  {
    std::vector<ResizeGroup> groups(xt.size() / 4 + 1);
    std::vector<int32_t> sxas(xt.size() / 4 + 1, -1);
    memset(groups.data(), 0, sizeof(ResizeGroup) * groups.size());
    for (int l = 1; l < L; ++l) {
      const int sw = e->geom[l - 1].w;
      const size_t g0 = e->geom[l].xtab_off / 4, g1 = (l + 1 < L ? e->geom[l + 1].xtab_off : xt.size()) / 4;
      for (size_t G = g0; G < g1; ++G) {
        const ResizeTab* X = xt.data() + 4 * G;
        ResizeGroup& R = groups[G];
        const int sxa = std::min((int)X[0].sofs, sw - 8);
        bool ok = sw >= 8 && sxa >= 0;
        for (int i = 0; i < 4; ++i) {
          ok = ok && X[i].sofs >= sxa && X[i].sofs + 1 - sxa <= 7;
          const uint32_t o = (uint32_t)std::min(std::max(X[i].sofs - sxa, 0), 6);
          R.sel[i] = o | (0x0cu << 8) | ((o + 1) << 16) | (0x0cu << 24);
          R.w[i] = (uint32_t)(uint16_t)X[i].a0 | ((uint32_t)(uint16_t)X[i].a1 << 16);
        }
        sxas[G] = ok ? sxa : -1;
      }
    }
    RGBL_TRY(dev_alloc(e, &e->d_xgroups, groups.size()));
    RGBL_TRY(dev_alloc(e, &e->d_xsxa, sxas.size()));
    RGBL_HIP(hipMemcpy(e->d_xgroups, groups.data(), sizeof(ResizeGroup) * groups.size(), hipMemcpyHostToDevice));
    RGBL_HIP(hipMemcpy(e->d_xsxa, sxas.data(), sizeof(int32_t) * sxas.size(), hipMemcpyHostToDevice));
  }
  RGBL_TRY(dev_alloc(e, &e->d_rootx, rootx.size()));
  RGBL_TRY(dev_alloc(e, &e->d_pattern, 1024));
  RGBL_HIP(hipMemcpy(e->d_geom, e->geom.data(), sizeof(LevelGeom) * L, hipMemcpyHostToDevice));
  if (!xt.empty()) RGBL_HIP(hipMemcpy(e->d_xtab, xt.data(), sizeof(ResizeTab) * xt.size(), hipMemcpyHostToDevice));
  if (!yt.empty()) RGBL_HIP(hipMemcpy(e->d_ytab, yt.data(), sizeof(ResizeTab) * yt.size(), hipMemcpyHostToDevice));
  RGBL_HIP(hipMemcpy(e->d_rootx, rootx.data(), rootx.size(), hipMemcpyHostToDevice));
  RGBL_HIP(hipMemcpy(e->d_pattern, kBriefPattern, 1024, hipMemcpyHostToDevice));
  if (const char* v = getenv("RGBL_XCD_MAP")) e->xcd_map = v[0] != '0';
  e->split_pyr = L >= 6 ? L / 2 : 0;   // 8 levels: the levels 4 - 7 (a fifth of the pixels) leave the main chain (round 5: the default)
  if (const char* v = getenv("RGBL_SPLIT_PYR")) e->split_pyr = atoi(v);
  if (const char* v = getenv("RGBL_FAST_BS")) e->fast_waves = atoi(v) == 64 ? 1 : atoi(v) == 128 ? 2 : 0;
  return RGBL_OK;
}

// layout of the host-pointer path's result block (device and page-locked): head | keypoints | descriptors, every part 256-byte aligned
static inline size_t stage_head(const rgbl_extractor* e) { return (256 + 2 * sizeof(int32_t) * (size_t)e->cfg.max_batch + 255) / 256 * 256; }
static inline size_t stage_kp_bytes(const rgbl_extractor* e, int batch) { return ((size_t)batch * e->out_cap * sizeof(rgbl_keypoint) + 255) / 256 * 256; }

int alloc_scratch(rgbl_extractor* e) {
  const size_t B = (size_t)e->cfg.max_batch;
  RGBL_TRY(dev_alloc(e, &e->d_img, B * e->img_frame + 256));
  RGBL_TRY(dev_alloc(e, &e->d_pyr, B * e->pyr_frame + 256));  // slack: 8-byte source reads may run past a row end
  RGBL_TRY(dev_alloc(e, &e->d_blur, B * e->pyr_frame + 256));
  RGBL_TRY(dev_alloc(e, &e->d_cellcnt, B * e->cells_frame));
  RGBL_TRY(dev_alloc(e, &e->d_slots, B * e->slots_frame));
  RGBL_TRY(dev_alloc(e, &e->d_keys_a, B * e->keys_frame));
  RGBL_TRY(dev_alloc(e, &e->d_keys_b, B * e->keys_frame));
  if (e->octree_ncap == 0) {  // global node lists of the key-moving quad-tree kernel
    RGBL_TRY(dev_alloc(e, &e->d_list_a, B * e->nodes_frame));
    RGBL_TRY(dev_alloc(e, &e->d_list_b, B * e->nodes_frame));
    RGBL_TRY(dev_alloc(e, &e->d_div, B * e->nodes_frame));
    RGBL_TRY(dev_alloc(e, &e->d_todo_a, B * e->nodes_frame));
    RGBL_TRY(dev_alloc(e, &e->d_todo_b, B * e->nodes_frame));
    RGBL_TRY(dev_alloc(e, &e->d_skey, B * e->nodes_frame));
    RGBL_TRY(dev_alloc(e, &e->d_sval, B * e->nodes_frame));
    RGBL_TRY(dev_alloc(e, &e->d_divided, B * e->nodes_frame));
  }
  RGBL_TRY(dev_alloc(e, &e->d_kpkey, B * (size_t)e->kp_frame));
  RGBL_TRY(dev_alloc(e, &e->d_kpcount, B * (size_t)e->L));
  RGBL_TRY(dev_alloc(e, &e->d_levelcnt, 2 * B * (size_t)e->L));  // counters | the last extraction's counts
  RGBL_HIP(hipMemset(e->d_levelcnt, 0, 2 * B * (size_t)e->L * sizeof(uint32_t)));
  RGBL_TRY(dev_alloc(e, &e->d_dbg, B * (size_t)e->L * 16));
  RGBL_HIP(hipMemset(e->d_dbg, 0, B * (size_t)e->L * 16 * sizeof(unsigned long long)));
  e->out_cap = e->kp_frame;
  // The host-pointer path's results live in ONE device block laid out like the page-locked block they are copied into:
  //   error flags (256 B) | counts [B] | monoIndex [B] | keypoints [B x out_cap] | descriptors [B x out_cap x 32]
  // so that a call that fills the whole handle (one frame per call above all) brings everything back with ONE copy - round 4
  // queued five (flag, counts, monoIndex, keypoints, descriptors: three copy kernels and two DMA transfers of ~5 us each at
  // the very end of a frame's critical path).
  {
    // head and keypoint ranges are rounded up to 256 bytes (in this block and in the page-locked one alike), so that the
    // descriptors - written with 16-byte and 8-byte stores by k_lapping_permute / k_orient_brief - start 256-byte aligned
    const size_t head = stage_head(e), kp_bytes = stage_kp_bytes(e, (int)B);
    uint8_t* blk = nullptr;
    RGBL_TRY(dev_alloc(e, &blk, head + kp_bytes + B * (size_t)e->out_cap * 32));
    e->d_stage = blk;
    e->d_err = reinterpret_cast<int*>(blk);
    e->d_out_n = reinterpret_cast<int32_t*>(blk + 256);
    e->d_out_mono = e->d_out_n + B;
    e->d_out_kp = reinterpret_cast<rgbl_keypoint*>(blk + head);
    e->d_out_desc = blk + head + kp_bytes;
    RGBL_HIP(hipMemset(blk, 0, head));
  }
  RGBL_TRY(dev_alloc(e, &e->d_tmp_kp, B * (size_t)e->out_cap));
  RGBL_TRY(dev_alloc(e, &e->d_tmp_desc, B * (size_t)e->out_cap * 32));
  // results of up to 4 frames per call come back through one page-locked block (run_staged)
  e->h_pinned_bytes = stage_head(e) + stage_kp_bytes(e, (int)std::min<size_t>(B, 4)) + std::min<size_t>(B, 4) * (size_t)e->out_cap * 32;
  if (hipHostMalloc(reinterpret_cast<void**>(&e->h_pinned), e->h_pinned_bytes, hipHostMallocDefault) != hipSuccess) { e->h_pinned = nullptr; e->h_pinned_bytes = 0; (void)hipGetLastError(); }
  return RGBL_OK;
}

// Enqueues the whole extraction of `batch` frames on e->stream. All pointers are device pointers.
int enqueue_extract(rgbl_extractor* e, const uint8_t* d_imgs, int batch, int stride, size_t frame_stride,
                    int lap0, int lap1, rgbl_keypoint* d_kp, uint8_t* d_desc, int cap, int32_t* d_n,
                    int32_t* d_mono) {
  const int L = e->L;
  hipStream_t s = e->stream;
  e->last_img0 = d_imgs; e->last_pitch0 = stride; e->last_frame0 = frame_stride; e->last_batch = batch;
  if (e->dense && e->dense_dirty) RGBL_HIP(hipMemsetAsync(e->d_levelcnt, 0, sizeof(uint32_t) * (size_t)e->cfg.max_batch * L, s));
  e->dense_dirty = e->dense;  // cleared at the end of a complete enqueue: the quad-tree workgroups leave the counters at zero

  // cells of at most kCellSmall px (every level of the usual image sizes) take the small-LDS instantiation: two waves per
  // cell (a cell is a chain of short phases; 16 workgroups of two waves per CU overlap better than 8 of four), tile pitch 48
  // bytes when no level's cells are wider than 41 px (cell + 7 bytes per tile row), else 64; bigger cells: four waves, pitch 80
  int fast_bs = e->max_cell <= kCellSmall ? 128 : 256;
  auto fast = e->max_cell <= kCellSmall ? (e->max_cell_w <= 41 ? k_fast_cells<kCellSmall, 128, 48> : k_fast_cells<kCellSmall, 128, 64>) : k_fast_cells<kCellMax, 256, 80>;
  // Batches: ONE wave per cell - no second wave's fixed work (6.0e8 instead of 6.7e8 vector instructions per 512-frame step),
  // the kernel's own time is the same (fewer waves to hide the LDS round trips) but what runs beside it gains; a single
  // frame is latency-bound and keeps two waves per cell (half the trips per wave).  RGBL_FAST_BS=64 / 128 overrides.
  const bool one_wave = e->fast_waves ? e->fast_waves == 1 : batch >= 8;
  if (one_wave && e->max_cell <= kCellSmall && e->max_cell_w <= 41) { fast = k_fast_cells<kCellSmall, 64, 48>; fast_bs = 64; }
  // Batches: the cells write their own slots (no reservation on the level's counter: on a 4K frame all resident cells of a
  // XCD hammered ONE address) and k_compact_cells builds the dense lists behind them, one atomic per 256 cells.  A single
  // frame keeps the reservation inside the FAST kernel: no contention to speak of, and one launch less on its critical path.
  const bool compact = e->dense && batch >= e->compact_min_batch;
  auto level_at = [&](int cell) { for (int l = 0; l < L; ++l) if ((int)e->geom[l].cell_off == cell) return l; return L; };
  auto launch_fast = [&](hipStream_t st, int cell_begin, int cell_end) {
    if (cell_end <= cell_begin) return;
    e->timer.begin("k_fast_cells", st);
    hipLaunchKernelGGL(fast, xcd_grid(e->xcd_map, cell_end - cell_begin, batch), dim3(fast_bs), 0, st, e->d_cells, d_imgs, stride, frame_stride,
                       e->d_pyr, e->pyr_frame, e->cfg.ini_th_fast, e->cfg.min_th_fast, e->d_cellcnt, (size_t)e->cells_frame,
                       e->d_slots, e->slots_frame, cell_begin, e->d_geom, L, e->d_keys_a, e->keys_frame, (e->dense && !compact) ? e->d_levelcnt : nullptr);
    e->timer.end(st);
    if (compact) {
      const int g0 = e->group_off[level_at(cell_begin)], g1 = e->group_off[level_at(cell_end)];   // the ranges are whole levels
      e->timer.begin("k_compact_cells", st);
      hipLaunchKernelGGL(k_compact_cells, xcd_grid(e->xcd_map, g1 - g0, batch), dim3(256), 0, st, e->d_groups, e->d_cells, e->d_cellcnt,
                         (size_t)e->cells_frame, e->d_slots, e->slots_frame, e->d_geom, L, e->d_keys_a, e->keys_frame, e->d_levelcnt, g0);
      e->timer.end(st);
    }
  };
  auto launch_gauss = [&](hipStream_t st, int tile_begin, int tile_end) {
    if (tile_end <= tile_begin) return;
    e->timer.begin("k_gauss7", st);
    // two passes with a barrier in between: 16 workgroups of two waves per CU interleave better than 8 of four (0.63 -> 0.54 ms)
    const bool g128 = !e->gauss_wg256;
    hipLaunchKernelGGL(g128 ? k_gauss7<128> : k_gauss7<256>, xcd_grid(e->xcd_map, tile_end - tile_begin, batch), dim3(g128 ? 128 : 256), 0, st, e->d_gtiles,
                       d_imgs, stride, frame_stride, e->d_pyr, e->pyr_frame, e->d_blur, e->pyr_frame, tile_begin);
    e->timer.end(st);
  };
  // buffers of the quad-tree distribution (ORBextractor.cc:555-779)
  OctreeBufs ob;
  ob.cell_cnt = e->d_cellcnt; ob.cells_frame = (size_t)e->cells_frame;
  ob.slots = e->d_slots; ob.slots_frame = e->slots_frame;
  ob.keys_a = e->d_keys_a; ob.keys_b = e->d_keys_b; ob.keys_frame = e->keys_frame;
  ob.list_a = e->d_list_a; ob.list_b = e->d_list_b; ob.div = e->d_div;
  ob.todo_a = e->d_todo_a; ob.todo_b = e->d_todo_b; ob.skey = e->d_skey; ob.sval = e->d_sval;
  ob.divided = e->d_divided; ob.nodes_frame = e->nodes_frame;
  ob.rootx = e->d_rootx;
  ob.kp_key = e->d_kpkey; ob.kp_count = e->d_kpcount; ob.kp_frame = (size_t)e->kp_frame;
  ob.level_cnt = e->dense ? e->d_levelcnt : nullptr;
  ob.level_cnt_last = e->d_levelcnt + (size_t)e->cfg.max_batch * L;
  ob.err = e->d_err;
  ob.dbg = e->octree_stamps ? e->d_dbg : nullptr;
  ob.no_hist = e->octree_no_hist ? 1 : 0;
  // narrow workgroups leave room for more (level, frame) problems per CU; small batches, which cannot fill the chip anyway,
  // take the wide group (shorter passes over the keys).  Node lists of up to 512 / 2048 entries live in LDS
  // (octree_labels.h); beyond that - more than ~9 000 features - the key-moving kernel on global lists takes over.
  const bool narrow = e->octree_wg ? e->octree_wg == kOctNarrow : (long)L * batch >= 1024;
  auto launch_octree = [&](hipStream_t st, int level_begin, int level_end) {
    if (level_end <= level_begin) return;
    e->timer.begin("k_octree", st);
    const dim3 grid(level_end - level_begin, batch);
    if (e->octree_ncap == 0) {
      if (narrow) hipLaunchKernelGGL(k_octree_moving<kOctNarrow>, grid, dim3(kOctNarrow), 0, st, e->d_geom, L, ob, level_begin);
      else hipLaunchKernelGGL(k_octree_moving<kOctWide>, grid, dim3(kOctWide), 0, st, e->d_geom, L, ob, level_begin);
    } else if (e->octree_ncap == 512) {
      // a handful of problems (a single frame): candidate lists in LDS, 1024 work-items each (octree_labels.h: KEYCAP)
      if (e->octree_ldskeys && !e->octree_wg && (long)(level_end - level_begin) * batch <= 64)
        hipLaunchKernelGGL((k_octree<1024, 512, kOctKeysLds>), grid, dim3(1024), 0, st, e->d_geom, L, ob, level_begin);
      else if (narrow) hipLaunchKernelGGL((k_octree<kOctNarrow, 512>), grid, dim3(kOctNarrow), 0, st, e->d_geom, L, ob, level_begin);
      else hipLaunchKernelGGL((k_octree<kOctWide, 512>), grid, dim3(kOctWide), 0, st, e->d_geom, L, ob, level_begin);
    } else {
      // 2048 nodes = 147 KB of LDS: ONE workgroup per CU whatever its width, so it takes all 16 wave slots of the CU - a 4K
      // level-0 problem is ~160 k candidates per pass (round 4: 512 -> 1024 work-items; RGBL_OCTREE_WG=256 / 512 still pin the others)
      if (e->octree_wg == kOctNarrow) hipLaunchKernelGGL((k_octree<kOctNarrow, 2048>), grid, dim3(kOctNarrow), 0, st, e->d_geom, L, ob, level_begin);
      else if (e->octree_wg == kOctWide) hipLaunchKernelGGL((k_octree<kOctWide, 2048>), grid, dim3(kOctWide), 0, st, e->d_geom, L, ob, level_begin);
      else hipLaunchKernelGGL((k_octree<1024, 2048>), grid, dim3(1024), 0, st, e->d_geom, L, ob, level_begin);
    }
    e->timer.end(st);
  };
  {
    // Level 0 of the pyramid is the input image itself: its FAST cells (a third of all pixels) and its Gaussian do not
    // wait for the resize chain - seven short dependent launches that leave most of the chip idle - but run next to it
    // on the auxiliary stream.  (While per-kernel timing is on, everything stays on one stream so that the event
    // brackets are not contended.)
    const bool overlap = !e->timer.enabled;
    hipStream_t bs = overlap ? e->aux_stream : s;
    const int cells0 = L > 1 ? e->geom[1].cell_off : e->cells_frame, tiles0 = e->blur_tiles.tile_off[1];
    // 1. pyramid: level l from level l-1 (ORBextractor.cc:1170-1195)
    auto launch_resize = [&](hipStream_t st, int l) {
      const LevelGeom& g = e->geom[l];
      const LevelGeom& p = e->geom[l - 1];
      const uint8_t* src = (l == 1) ? d_imgs : e->d_pyr + p.img_off;
      const int spitch = (l == 1) ? stride : p.pitch;
      const size_t sframe = (l == 1) ? frame_stride : e->pyr_frame;
      e->timer.begin("k_resize_linear", st);
      const int rtx = (g.w + 4 * kResizeLanes - 1) / (4 * kResizeLanes), rty = (g.h + 4 * kResizeRows - 1) / (4 * kResizeRows);
      hipLaunchKernelGGL(k_resize_linear, xcd_grid(e->xcd_map, rtx * rty, batch), dim3(kResizeWG), 0, st, src, spitch, sframe, p.w, p.h, e->d_pyr + g.img_off, g.pitch, e->pyr_frame, g.w, g.h,
                       e->d_xtab + g.xtab_off, e->d_xgroups + g.xtab_off / 4, e->d_xsxa + g.xtab_off / 4, e->d_ytab + g.ytab_off, rtx);
      e->timer.end(st);
    };
    const int sp = (overlap && batch >= 8 && e->split_pyr >= 2 && e->split_pyr < L) ? e->split_pyr : 0;
    if (sp) {
      // Batches (RGBL_SPLIT_PYR=k; default k = L / 2): the small levels k .. L - 1 - their resizes are the tail of a chain of
      // dependent launches, their FAST cells a fifth of the pixels - are produced on the auxiliary stream behind level 0's FAST
      // cells, so that the main stream's FAST launch (levels 1 .. k - 1) starts after k - 1 resizes instead of L - 1.
      // Round 3: k = 4 144.7 k, k = 3 144.3 k, k = 5 143.0 k against 142.9 - 143.3 k frames/s, left off.  Round 5, three A/B
      // calls on the KITTI step: k = 4 +1.8 ... +2.7 % (145.4 / 146.3 / 144.4 k against 142.8 / 142.9 / 140.6 k), k = 3 and 5
      // nothing; the small levels' quad-trees on the auxiliary stream as well: -2 %.  The default since.
      RGBL_HIP(hipEventRecord(e->ev_start, s));
      RGBL_HIP(hipStreamWaitEvent(bs, e->ev_start, 0));
      launch_fast(bs, 0, cells0);
      for (int l = 1; l < sp; ++l) launch_resize(s, l);
      RGBL_HIP(hipEventRecord(e->ev_r, s));
      RGBL_HIP(hipStreamWaitEvent(bs, e->ev_r, 0));
      for (int l = sp; l < L; ++l) launch_resize(bs, l);
      launch_fast(bs, e->geom[sp].cell_off, e->cells_frame);
      RGBL_HIP(hipEventRecord(e->ev_fb, bs));
      launch_gauss(bs, 0, tiles0);
      launch_octree(bs, 0, 1);
      RGBL_HIP(hipEventRecord(e->ev_fast0, bs));
      launch_gauss(bs, tiles0, e->blur_tiles.tile_off[L]);
      RGBL_HIP(hipEventRecord(e->ev_blur, bs));
      launch_fast(s, cells0, e->geom[sp].cell_off);
      RGBL_HIP(hipStreamWaitEvent(s, e->ev_fb, 0));
      launch_octree(s, 1, L);
      RGBL_HIP(hipStreamWaitEvent(s, e->ev_fast0, 0));
      RGBL_HIP(hipStreamWaitEvent(s, e->ev_blur, 0));
    } else if (overlap && batch < 8 && e->level_split >= 2 && e->level_split < L) {
      // A single frame is a race of dependent chains, and the longest one sets its latency.  Until round 4: main stream =
      // 7 resizes -> FAST of ALL upper levels -> their quad-trees (the level-1 problem: 53 us) -> descriptors.  Level 1 exists
      // after ONE resize: the levels 1 .. k - 1 (k = 3) now take a third stream - FAST and quad-tree behind their own resizes -
      // while the main stream finishes the pyramid and handles the small levels k .. L - 1; level 0 stays on the auxiliary
      // stream.  Critical path 29 + 14 + 55 us -> max(level 0: 14 + quad-tree, levels 1 - 2: 8 + 12 + quad-tree, rest: 29 + 8 + 32).
      const int k = e->level_split;
      hipStream_t ls = e->lvl_stream;
      // The captured graph dispatches its nodes in the order of capture, a few us apiece: level 0 first (it needs nothing but the
      // image), the Gaussian of the upper levels - read by the descriptors only - LAST, behind FAST and quad-tree of the small
      // levels (captured in front of them it delayed that branch, the last to finish: 0.332 -> 0.315 ms per frame on one box,
      // no change on a faster one; capturing the longest branch first cost 60 us: 0.31 -> 0.38 ms).
      RGBL_HIP(hipEventRecord(e->ev_start, s));
      RGBL_HIP(hipStreamWaitEvent(bs, e->ev_start, 0));
      launch_fast(bs, 0, cells0);
      launch_gauss(bs, 0, tiles0);
      launch_octree(bs, 0, 1);
      RGBL_HIP(hipEventRecord(e->ev_fast0, bs));
      for (int l = 1; l < k; ++l) launch_resize(s, l);
      RGBL_HIP(hipEventRecord(e->ev_r, s));
      RGBL_HIP(hipStreamWaitEvent(ls, e->ev_r, 0));
      launch_fast(ls, cells0, e->geom[k].cell_off);
      launch_octree(ls, 1, k);
      RGBL_HIP(hipEventRecord(e->ev_fb, ls));
      for (int l = k; l < L; ++l) launch_resize(s, l);
      RGBL_HIP(hipEventRecord(e->ev_pyr, s));
      launch_fast(s, e->geom[k].cell_off, e->cells_frame);
      launch_octree(s, k, L);
      RGBL_HIP(hipStreamWaitEvent(bs, e->ev_pyr, 0));
      launch_gauss(bs, tiles0, e->blur_tiles.tile_off[L]);
      RGBL_HIP(hipEventRecord(e->ev_blur, bs));
      RGBL_HIP(hipStreamWaitEvent(s, e->ev_fb, 0));
      RGBL_HIP(hipStreamWaitEvent(s, e->ev_fast0, 0));
      RGBL_HIP(hipStreamWaitEvent(s, e->ev_blur, 0));
    } else {
    if (overlap) {
      RGBL_HIP(hipEventRecord(e->ev_start, s));
      RGBL_HIP(hipStreamWaitEvent(bs, e->ev_start, 0));
      launch_fast(bs, 0, cells0);
      // level 0's Gaussian in front of its quad-tree launch: the quad-tree workgroups (37 KB of LDS each) only get placed as the
      // FAST cells of the upper levels drain anyway, the Gaussian fills the time until then (125.7 -> 126.8 k frames/s; the
      // quad-tree launch on the main stream in front of the upper levels' FAST cells instead: 122.3 k)
      launch_gauss(bs, 0, tiles0);
      launch_octree(bs, 0, 1);
      RGBL_HIP(hipEventRecord(e->ev_fast0, bs));
    }
    for (int l = 1; l < L; ++l) launch_resize(s, l);
    // 4. Gaussian working images (ORBextractor.cc:1132-1133) of the upper levels, on the auxiliary stream next to 2. and 3.
    if (overlap) {
      RGBL_HIP(hipEventRecord(e->ev_pyr, s));
      RGBL_HIP(hipStreamWaitEvent(bs, e->ev_pyr, 0));
      launch_gauss(bs, tiles0, e->blur_tiles.tile_off[L]);
      RGBL_HIP(hipEventRecord(e->ev_blur, bs));
    } else {
      launch_gauss(s, 0, e->blur_tiles.tile_off[L]);
    }
    // 2. FAST per detection cell (ORBextractor.cc:806-872)
    if (overlap) {
      launch_fast(s, cells0, e->cells_frame);
    } else {
      launch_fast(s, 0, e->cells_frame);
    }
    // 3. quad-tree distribution (ORBextractor.cc:555-779) of the remaining levels (level 0 went with its FAST cells above)
    if (overlap) {
      launch_octree(s, 1, L);
      RGBL_HIP(hipStreamWaitEvent(s, e->ev_fast0, 0));
    } else {
      launch_octree(s, 0, L);
    }
    // 5. orientation + descriptors + packing (ORBextractor.cc:894-895, 1136-1165); needs the blurred levels
    if (overlap) RGBL_HIP(hipStreamWaitEvent(s, e->ev_blur, 0));
    }
  }
  const bool lapping = lap1 >= 19 && lap1 >= lap0;  // keypoint x is always >= 19: nothing can fall into [lap0, lap1] otherwise
  rgbl_keypoint* kp_dst = lapping ? e->d_tmp_kp : d_kp;
  uint8_t* desc_dst = lapping ? e->d_tmp_desc : d_desc;
  int lap_cap = cap;
  if (lapping) lap_cap = std::min(cap, e->out_cap);
  // Batches: level 0's keypoints (its quad-tree and its Gaussian are done long before the upper levels' quad-trees) are
  // described on the auxiliary stream next to those quad-trees; the main stream takes the other levels and the frame totals.
  const bool split_desc = !e->timer.enabled && batch >= 8 && L > 1 && e->geom[1].koff > 0;
  const int slot_split = split_desc ? e->geom[1].koff : 0;
  auto launch_desc = [&](hipStream_t st, int slot_begin, int slot_end, int write_total) {
    if (slot_end <= slot_begin) return;
    e->timer.begin("k_orient_brief", st);
    // one-wave or two-wave workgroups were measured behind four-wave ones here (0.72 - 0.73 vs 0.70 ms): the waves are independent anyway
    hipLaunchKernelGGL(k_orient_brief<256>, xcd_grid(e->xcd_map, (slot_end - slot_begin + 4 * kKpPerWave - 1) / (4 * kKpPerWave), batch), dim3(256), 0, st, e->d_geom, L, e->umax,
                       e->d_pattern, d_imgs, stride, frame_stride, e->d_pyr, e->pyr_frame, e->d_blur, e->pyr_frame,
                       e->d_kpkey, e->d_kpcount, (size_t)e->kp_frame, kp_dst, desc_dst, lapping ? e->out_cap : cap, d_n,
                       lapping ? (int32_t*)nullptr : d_mono, e->d_err, slot_begin, slot_end, write_total);
    e->timer.end(st);
  };
  if (split_desc) {
    // (the auxiliary stream has level 0's quad-tree and every level's Gaussian behind it at this point)
    launch_desc(e->aux_stream, 0, slot_split, 0);
    RGBL_HIP(hipEventRecord(e->ev_desc0, e->aux_stream));
  }
  launch_desc(s, slot_split, e->kp_frame, 1);
  if (split_desc) RGBL_HIP(hipStreamWaitEvent(s, e->ev_desc0, 0));
  if (lapping) {
    (void)lap_cap;
    if (cap < e->out_cap) {
      set_error("vLappingArea packing needs cap >= %d", e->out_cap);
      return RGBL_ERR_CAPACITY;
    }
    e->timer.begin("k_lapping_permute", s);
    // the temporary arrays use out_cap as their frame stride, the destination uses cap
    hipLaunchKernelGGL(k_lapping_permute, dim3(batch), dim3(256), 0, s, e->d_tmp_kp, e->d_tmp_desc, e->out_cap, d_kp,
                       d_desc, cap, d_n, (float)lap0, (float)lap1, d_mono);
    e->timer.end(s);
  }
  RGBL_HIP(hipGetLastError());
  e->dense_dirty = false;
  return RGBL_OK;
}

int check_device_flags(rgbl_extractor* e) {
  int flags = 0;
  RGBL_HIP(hipMemcpy(&flags, e->d_err, sizeof(int), hipMemcpyDeviceToHost));
  if (flags) {
    RGBL_HIP(hipMemset(e->d_err, 0, sizeof(int)));
    RGBL_HIP(hipMemset(e->d_levelcnt, 0, sizeof(uint32_t) * (size_t)e->cfg.max_batch * e->L));  // whatever a failed extraction left behind
    if (flags & 3) { set_error("quad-tree scratch overflow (flags=%d)", flags); return RGBL_ERR_OVERFLOW; }
    set_error("more keypoints than the caller's capacity");
    return RGBL_ERR_CAPACITY;
  }
  return RGBL_OK;
}

}  // namespace

extern "C" {

const char* rgbl_last_error(void) { return rgbl::last_error(); }
const char* rgbl_backend(void) {
#ifdef RGBL_EMU
  return "emu";
#else
  return "hip:gfx950";
#endif
}
int rgbl_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int rgbl_extractor_create(const rgbl_extractor_cfg* cfg, int device, rgbl_extractor** out) {
  if (!cfg || !out) { set_error("null argument"); return RGBL_ERR_INVALID; }
  *out = nullptr;
  if (cfg->nlevels < 1 || cfg->nlevels > kMaxLevels || cfg->nfeatures < 1 || cfg->scale_factor < 1.0f ||
      cfg->width < 64 || cfg->height < 64 || cfg->width > 4096 || cfg->height > 4096 || cfg->max_batch < 1 ||
      cfg->min_th_fast < 1 || cfg->ini_th_fast < cfg->min_th_fast || cfg->ini_th_fast > 255) {
    set_error("invalid extractor configuration");
    return RGBL_ERR_INVALID;
  }
  if (rgbl_device_count() <= device || device < 0) {
    set_error("no usable HIP device %d (this library has no CPU fallback)", device);
    return RGBL_ERR_NO_DEVICE;
  }
  RGBL_HIP(hipSetDevice(device));
  rgbl_extractor* e = new rgbl_extractor;
  e->cfg = *cfg;
  // Tuning switches are read here, once per handle (include/rgbl_frontend.h lists them) - never on a launch path.
  if (const char* v = getenv("RGBL_GRAPH")) e->graph_ok = atoi(v) != 0;
  if (const char* v = getenv("RGBL_GAUSS_BS")) e->gauss_wg256 = atoi(v) == 256;
  if (getenv("RGBL_OCTREE_STAMPS")) e->octree_stamps = true;
  if (const char* v = getenv("RGBL_OCTREE_HIST")) e->octree_no_hist = v[0] == '0';
  if (const char* v = getenv("RGBL_OCTREE_LDSKEYS")) e->octree_ldskeys = atoi(v) != 0;
  if (const char* v = getenv("RGBL_LEVEL_SPLIT")) e->level_split = atoi(v);
  if (const char* v = getenv("RGBL_OCTREE_WG")) { const int wg = atoi(v); if (wg == kOctNarrow || wg == kOctWide) e->octree_wg = wg; }
  e->device = device;
  int rc = build_geometry(e);
  if (rc == RGBL_OK) {
    // quad-tree kernel: the smallest LDS node capacity that holds every level's list (RGBL_OCTREE_NCAP=0 forces the
    // key-moving kernel on global lists, 2048 the large instantiation - tests)
    uint32_t node_cap = 0, key_cap = 0;
    for (int l = 0; l < e->L; ++l) { node_cap = std::max(node_cap, e->geom[l].node_cap); key_cap = std::max(key_cap, e->geom[l].key_cap); }
    e->octree_ncap = node_cap <= 512 ? 512 : (node_cap <= 2048 ? 2048 : 0);
    if (key_cap >= (1u << 24)) e->octree_ncap = 0;  // candidate indices travel in 24 bits of its best-key word
    if (const char* v = getenv("RGBL_OCTREE_NCAP")) {
      const int want = atoi(v);
      if (want == 0 || (want == 2048 && node_cap <= 2048 && e->octree_ncap != 0)) e->octree_ncap = want;
    }
  }
  if (rc == RGBL_OK) rc = upload_tables(e);
  // dense candidate lists need the label-based quad-tree kernel (order-free) and the per-cell FAST kernel (RGBL_DENSE=0: cell slots)
  if (rc == RGBL_OK) e->dense = e->octree_ncap != 0 && !(getenv("RGBL_DENSE") && getenv("RGBL_DENSE")[0] == '0');
  if (rc == RGBL_OK) rc = alloc_scratch(e);
  if (rc == RGBL_OK && (hipStreamCreate(&e->own_stream) != hipSuccess || hipStreamCreate(&e->aux_stream) != hipSuccess || hipStreamCreate(&e->lvl_stream) != hipSuccess ||
                        hipEventCreateWithFlags(&e->ev_pyr, hipEventDisableTiming) != hipSuccess ||
                        hipEventCreateWithFlags(&e->ev_blur, hipEventDisableTiming) != hipSuccess ||
                        hipEventCreateWithFlags(&e->ev_start, hipEventDisableTiming) != hipSuccess ||
                        hipEventCreateWithFlags(&e->ev_fast0, hipEventDisableTiming) != hipSuccess ||
                        hipEventCreateWithFlags(&e->ev_desc0, hipEventDisableTiming) != hipSuccess ||
                        hipEventCreateWithFlags(&e->ev_r, hipEventDisableTiming) != hipSuccess ||
                        hipEventCreateWithFlags(&e->ev_fb, hipEventDisableTiming) != hipSuccess)) {
    set_error("hipStreamCreate failed");
    rc = RGBL_ERR_HIP;
  }
  if (rc != RGBL_OK) { rgbl_extractor_destroy(e); return rc; }
  e->stream = e->own_stream;
  *out = e;
  return RGBL_OK;
}

void rgbl_extractor_destroy(rgbl_extractor* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  e->timer.collect();
  for (void* p : e->allocs) (void)hipFree(p);
  if (e->d_stereo_sad) (void)hipFree(e->d_stereo_sad);
  if (e->d_stereo_stage) (void)hipFree(e->d_stereo_stage);
  if (e->h_stereo_stage) (void)hipHostFree(e->h_stereo_stage);
  if (e->d_color) (void)hipFree(e->d_color);
#ifndef RGBL_EMU
  if (e->graph_exec) (void)hipGraphExecDestroy(e->graph_exec);
#endif
  if (e->aux_stream) { (void)hipStreamSynchronize(e->aux_stream); (void)hipStreamDestroy(e->aux_stream); }
  if (e->lvl_stream) { (void)hipStreamSynchronize(e->lvl_stream); (void)hipStreamDestroy(e->lvl_stream); }
  if (e->ev_pyr) (void)hipEventDestroy(e->ev_pyr);
  if (e->ev_blur) (void)hipEventDestroy(e->ev_blur);
  if (e->h_pinned) (void)hipHostFree(e->h_pinned);
  if (e->ev_start) (void)hipEventDestroy(e->ev_start);
  if (e->ev_fast0) (void)hipEventDestroy(e->ev_fast0);
  if (e->ev_desc0) (void)hipEventDestroy(e->ev_desc0);
  if (e->ev_r) (void)hipEventDestroy(e->ev_r);
  if (e->ev_fb) (void)hipEventDestroy(e->ev_fb);
  if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
  delete e;
}

int rgbl_extractor_tables(const rgbl_extractor* e, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                          int* per_level, int* umax16) {
  if (!e) { set_error("null handle"); return RGBL_ERR_INVALID; }
  for (int i = 0; i < e->L; ++i) {
    if (scale) scale[i] = e->scale[i];
    if (inv_scale) inv_scale[i] = e->inv_scale[i];
    if (sigma2) sigma2[i] = e->sigma2[i];
    if (inv_sigma2) inv_sigma2[i] = e->inv_sigma2[i];
    if (per_level) per_level[i] = e->per_level[i];
  }
  if (umax16) memcpy(umax16, e->umax.v, sizeof(int) * 16);
  return RGBL_OK;
}

int rgbl_extractor_max_keypoints(const rgbl_extractor* e) { return e ? e->kp_frame : 0; }

int rgbl_extract_batch_device(rgbl_extractor* e, const uint8_t* d_imgs, int batch, int w, int h, int stride,
                              size_t frame_stride, int lap0, int lap1, rgbl_keypoint* d_kp, uint8_t* d_desc, int cap,
                              int32_t* d_n, int32_t* d_mono) {
  if (!e || !d_imgs || !d_kp || !d_desc || !d_n || !d_mono) { set_error("null argument"); return RGBL_ERR_INVALID; }
  if (w != e->cfg.width || h != e->cfg.height || batch < 1 || batch > e->cfg.max_batch || stride < w || cap < 1 ||
      (batch > 1 && frame_stride < (size_t)stride * h)) {
    set_error("image %dx%d / batch %d does not match the handle (%dx%d, max batch %d)", w, h, batch, e->cfg.width,
              e->cfg.height, e->cfg.max_batch);
    return RGBL_ERR_INVALID;
  }
  RGBL_HIP(hipSetDevice(e->device));
  if (e->pending.active) { e->pending.active = false; RGBL_HIP(hipStreamSynchronize(e->stream)); }  // a begun extraction is overtaken
  return enqueue_extract(e, d_imgs, batch, stride, frame_stride, lap0, lap1, d_kp, d_desc, cap, d_n, d_mono);
}

int rgbl_extractor_sync(rgbl_extractor* e) {
  if (!e) { set_error("null handle"); return RGBL_ERR_INVALID; }
  RGBL_HIP(hipSetDevice(e->device));
  RGBL_HIP(hipStreamSynchronize(e->stream));
  e->timer.collect();
  return check_device_flags(e);
}

// The host-pointer path launches ~15 short kernels on two streams per call; for a single frame the gaps between dependent
// launches are a fifth of the latency.  The sequence is captured once into a hipGraph (both streams: the event fork / join
// inside enqueue_extract becomes graph edges) and replayed on the following calls.
static int enqueue_extract_staged(rgbl_extractor* e, int batch, int dev_stride, int lap0, int lap1) {
#ifndef RGBL_EMU
  hipStream_t s = e->stream;
  if (e->graph_ok && !e->timer.enabled) {
    const bool hit = e->graph_exec && e->graph_batch == batch && e->graph_stride == dev_stride && e->graph_lap0 == lap0 &&
                     e->graph_lap1 == lap1 && e->graph_stream == s;
    if (!hit) {
      if (e->graph_exec) { (void)hipGraphExecDestroy(e->graph_exec); e->graph_exec = nullptr; }
      hipGraph_t graph = nullptr;
      if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        const int rc = enqueue_extract(e, e->d_img, batch, dev_stride, e->img_frame, lap0, lap1, e->d_out_kp, e->d_out_desc, e->out_cap,
                                       e->d_out_n, e->d_out_mono);
        const hipError_t end = hipStreamEndCapture(s, &graph);
        if (rc == RGBL_OK && end == hipSuccess && graph && hipGraphInstantiate(&e->graph_exec, graph, nullptr, nullptr, 0) == hipSuccess) {
          e->graph_batch = batch; e->graph_stride = dev_stride; e->graph_lap0 = lap0; e->graph_lap1 = lap1; e->graph_stream = s;
        } else {
          e->graph_exec = nullptr;
          e->graph_ok = false;  // fall back to plain launches for the rest of the handle's life
          (void)hipGetLastError();
        }
        if (graph) (void)hipGraphDestroy(graph);
      } else {
        e->graph_ok = false;
        (void)hipGetLastError();
      }
    }
    if (e->graph_exec) {
      e->last_img0 = e->d_img; e->last_pitch0 = dev_stride; e->last_frame0 = e->img_frame; e->last_batch = batch;
      RGBL_HIP(hipGraphLaunch(e->graph_exec, s));
      return RGBL_OK;
    }
  }
#endif
  return enqueue_extract(e, e->d_img, batch, dev_stride, e->img_frame, lap0, lap1, e->d_out_kp, e->d_out_desc, e->out_cap, e->d_out_n,
                         e->d_out_mono);
}

// the frames already sit in e->d_img (row stride dev_stride): extraction, then the results back to the host.
// Small batches (one frame per call above all): counts, error flags, keypoints and descriptors travel into ONE page-locked
// block behind the kernels and the call synchronises ONCE - three blocking round trips (counts, the error flag, the
// arrays) and copies into pageable memory were a third of a single frame's extraction latency.
// an extraction begun with rgbl_extract_begin that another entry point now overtakes: wait for it, forget it
static int drop_pending(rgbl_extractor* e) {
  if (e->pending.active) { e->pending.active = false; RGBL_HIP(hipStreamSynchronize(e->stream)); }
  return RGBL_OK;
}

static bool staged_one_trip(const rgbl_extractor* e, int batch) {
  const size_t kp_bytes = stage_kp_bytes(e, batch), desc_bytes = (size_t)batch * e->out_cap * 32;
  const size_t head = stage_head(e);
  return e->h_pinned && head + kp_bytes + desc_bytes <= e->h_pinned_bytes;
}

// first half: the kernels and - for small batches - the copies of counts, error flag, keypoints and descriptors into the
// page-locked block, all queued, nothing waited for
static int staged_enqueue(rgbl_extractor* e, int batch, int dev_stride, int lap0, int lap1) {
  hipStream_t s = e->stream;
  RGBL_TRY(enqueue_extract_staged(e, batch, dev_stride, lap0, lap1));
  if (!staged_one_trip(e, batch)) return RGBL_OK;
  const size_t kp_bytes = stage_kp_bytes(e, batch), desc_bytes = (size_t)batch * e->out_cap * 32;
  const size_t head = stage_head(e);
  int32_t* p_err = reinterpret_cast<int32_t*>(e->h_pinned);
  int32_t* p_n = reinterpret_cast<int32_t*>(e->h_pinned + 256);
  int32_t* p_mono = p_n + e->cfg.max_batch;
  uint8_t* p_kp = e->h_pinned + head;
  uint8_t* p_desc = p_kp + kp_bytes;
  (void)p_err; (void)p_n; (void)p_mono;
  // device block and page-locked block share their layout up to the keypoints of `batch` frames; the descriptors follow the
  // keypoints of ALL max_batch frames on the device, of `batch` frames on the host: one copy when the call fills the handle
  if (batch == e->cfg.max_batch) {
    RGBL_HIP(hipMemcpyAsync(e->h_pinned, e->d_stage, head + kp_bytes + desc_bytes, hipMemcpyDeviceToHost, s));
  } else {
    RGBL_HIP(hipMemcpyAsync(e->h_pinned, e->d_stage, head + kp_bytes, hipMemcpyDeviceToHost, s));
    RGBL_HIP(hipMemcpyAsync(p_desc, e->d_out_desc, desc_bytes, hipMemcpyDeviceToHost, s));
  }
  (void)p_kp;
  return RGBL_OK;
}

// second half: wait, then the results from the page-locked block (or, for big batches, straight from the device) to the caller.
// Small batches (one frame per call above all): counts, error flags, keypoints and descriptors travel into ONE page-locked
// block behind the kernels and the call synchronises ONCE - three blocking round trips (counts, the error flag, the
// arrays) and copies into pageable memory were a third of a single frame's extraction latency.
static int staged_finish(rgbl_extractor* e, int batch, int lap1, rgbl_keypoint* out_kp, uint8_t* out_desc, int cap, int* out_n,
                         int* out_mono) {
  hipStream_t s = e->stream;
  if (staged_one_trip(e, batch)) {
    const size_t kp_bytes = stage_kp_bytes(e, batch);
    const size_t head = stage_head(e);
    int32_t* p_err = reinterpret_cast<int32_t*>(e->h_pinned);
    int32_t* p_n = reinterpret_cast<int32_t*>(e->h_pinned + 256);
    int32_t* p_mono = p_n + e->cfg.max_batch;
    uint8_t* p_kp = e->h_pinned + head;
    uint8_t* p_desc = p_kp + kp_bytes;
    RGBL_HIP(hipStreamSynchronize(s));
    e->timer.collect();
    if (*p_err) RGBL_TRY(check_device_flags(e));  // resets the flag and names the error
    int rc = RGBL_OK;
    for (int b = 0; b < batch; ++b) {
      const int n = p_n[b];
      out_n[b] = n; out_mono[b] = p_mono[b];
      const int ncopy = std::min(n, cap);
      if (n > cap) { set_error("frame %d has %d keypoints, capacity %d", b, n, cap); rc = RGBL_ERR_CAPACITY; }
      if (n > cap && lap1 >= 19) continue;  // a truncated lapping layout would be meaningless
      memcpy(out_kp + (size_t)b * cap, p_kp + (size_t)b * e->out_cap * sizeof(rgbl_keypoint), sizeof(rgbl_keypoint) * ncopy);
      memcpy(out_desc + (size_t)b * cap * 32, p_desc + (size_t)b * e->out_cap * 32, (size_t)ncopy * 32);
    }
    return rc;
  }
  RGBL_HIP(hipMemcpyAsync(out_n, e->d_out_n, sizeof(int32_t) * batch, hipMemcpyDeviceToHost, s));
  RGBL_HIP(hipMemcpyAsync(out_mono, e->d_out_mono, sizeof(int32_t) * batch, hipMemcpyDeviceToHost, s));
  RGBL_HIP(hipStreamSynchronize(s));
  e->timer.collect();
  RGBL_TRY(check_device_flags(e));
  int rc = RGBL_OK;
  for (int b = 0; b < batch; ++b) {
    const int n = out_n[b];
    const int ncopy = std::min(n, cap);
    if (n > cap) { set_error("frame %d has %d keypoints, capacity %d", b, n, cap); rc = RGBL_ERR_CAPACITY; }
    if (n > cap && lap1 >= 19) continue;  // a truncated lapping layout would be meaningless
    RGBL_HIP(hipMemcpyAsync(out_kp + (size_t)b * cap, e->d_out_kp + (size_t)b * e->out_cap, sizeof(rgbl_keypoint) * ncopy,
                            hipMemcpyDeviceToHost, s));
    RGBL_HIP(hipMemcpyAsync(out_desc + (size_t)b * cap * 32, e->d_out_desc + (size_t)b * e->out_cap * 32, (size_t)ncopy * 32,
                            hipMemcpyDeviceToHost, s));
  }
  RGBL_HIP(hipStreamSynchronize(s));
  return rc;
}

// the frames already sit in e->d_img (row stride dev_stride): extraction, then the results back to the host
static int run_staged(rgbl_extractor* e, int batch, int dev_stride, int lap0, int lap1, rgbl_keypoint* out_kp,
                      uint8_t* out_desc, int cap, int* out_n, int* out_mono) {
  RGBL_TRY(staged_enqueue(e, batch, dev_stride, lap0, lap1));
  return staged_finish(e, batch, lap1, out_kp, out_desc, cap, out_n, out_mono);
}

// host -> device staging of `batch` frames; returns the row stride the kernels read them with
static int upload_frames(rgbl_extractor* e, const uint8_t* imgs, int batch, int w, int h, int stride, size_t frame_stride, int* dev_stride) {
  hipStream_t s = e->stream;
  // The kernels accept any row stride, so a frame whose rows are at most img_pitch apart is moved with ONE linear copy and
  // read with the caller's stride (a 2-D copy from pageable memory degenerates into one transfer per row: 376 transfers
  // per KITTI frame).
  *dev_stride = e->img_pitch;
  if (stride <= e->img_pitch) {
    *dev_stride = stride;
    const size_t bytes = (size_t)(h - 1) * stride + w;  // never read past the caller's last row
    for (int b = 0; b < batch; ++b)
      RGBL_HIP(hipMemcpyAsync(e->d_img + (size_t)b * e->img_frame, imgs + (size_t)b * frame_stride, bytes, hipMemcpyHostToDevice, s));
  } else {
    for (int b = 0; b < batch; ++b)
      RGBL_HIP(hipMemcpy2DAsync(e->d_img + (size_t)b * e->img_frame, e->img_pitch, imgs + (size_t)b * frame_stride, stride,
                                w, h, hipMemcpyHostToDevice, s));
  }
  return RGBL_OK;
}

int rgbl_extract_begin(rgbl_extractor* e, const uint8_t* img, int w, int h, int stride, int lap0, int lap1) {
  if (!e) { set_error("null handle"); return RGBL_ERR_INVALID; }
  if (!img || w <= 0 || h <= 0) { set_error("empty image"); return RGBL_ERR_EMPTY; }
  if (w != e->cfg.width || h != e->cfg.height || stride < w) {
    set_error("image %dx%d does not match the handle (%dx%d)", w, h, e->cfg.width, e->cfg.height);
    return RGBL_ERR_INVALID;
  }
  if (!staged_one_trip(e, 1)) {
    set_error("rgbl_extract_begin needs the handle's page-locked result block (%zu bytes; one frame's results do not fit or the block could not be allocated): call rgbl_extract instead",
              e->h_pinned ? e->h_pinned_bytes : (size_t)0);
    return RGBL_ERR_CAPACITY;
  }
  RGBL_HIP(hipSetDevice(e->device));
  if (e->pending.active) { RGBL_HIP(hipStreamSynchronize(e->stream)); e->pending.active = false; }  // an extraction nobody collected
  int dev_stride = 0;
  RGBL_TRY(upload_frames(e, img, 1, w, h, stride, 0, &dev_stride));
  RGBL_TRY(staged_enqueue(e, 1, dev_stride, lap0, lap1));
  e->pending.active = true; e->pending.img = img; e->pending.w = w; e->pending.h = h; e->pending.stride = stride;
  e->pending.lap0 = lap0; e->pending.lap1 = lap1;
  return RGBL_OK;
}

// A begun frame is recognised by its host pointer and geometry only.  A caller that drops the frame without collecting it
// (rgbl_extract on the same buffer) says so here before the buffer is freed: the allocator may hand the address to the next frame.
int rgbl_extract_cancel(rgbl_extractor* e) {
  if (!e) { set_error("null handle"); return RGBL_ERR_INVALID; }
  RGBL_HIP(hipSetDevice(e->device));
  return drop_pending(e);
}

int rgbl_extract_batch(rgbl_extractor* e, const uint8_t* imgs, int batch, int w, int h, int stride, size_t frame_stride,
                       int lap0, int lap1, rgbl_keypoint* out_kp, uint8_t* out_desc, int cap, int* out_n,
                       int* out_mono) {
  if (out_n) for (int b = 0; b < std::max(batch, 0); ++b) out_n[b] = 0;
  if (out_mono) for (int b = 0; b < std::max(batch, 0); ++b) out_mono[b] = -1;
  if (!e) { set_error("null handle"); return RGBL_ERR_INVALID; }
  if (!imgs || w <= 0 || h <= 0) { set_error("empty image"); return RGBL_ERR_EMPTY; }  // ORBextractor.cc:1090-1091
  if (!out_kp || !out_desc || !out_n || !out_mono) { set_error("null output"); return RGBL_ERR_INVALID; }
  if (w != e->cfg.width || h != e->cfg.height || batch < 1 || batch > e->cfg.max_batch || stride < w || cap < 1) {
    set_error("image %dx%d / batch %d does not match the handle (%dx%d, max batch %d)", w, h, batch, e->cfg.width,
              e->cfg.height, e->cfg.max_batch);
    return RGBL_ERR_INVALID;
  }
  RGBL_HIP(hipSetDevice(e->device));
  if (e->pending.active) {
    // rgbl_extract_begin was given this very frame: its kernels and result copies are queued (or done) - collect them
    const bool same = batch == 1 && e->pending.img == imgs && e->pending.w == w && e->pending.h == h && e->pending.stride == stride &&
                      e->pending.lap0 == lap0 && e->pending.lap1 == lap1;
    e->pending.active = false;
    if (same) return staged_finish(e, 1, lap1, out_kp, out_desc, cap, out_n, out_mono);
    RGBL_HIP(hipStreamSynchronize(e->stream));  // another frame: the begun extraction is dropped
  }
  int dev_stride = 0;
  RGBL_TRY(upload_frames(e, imgs, batch, w, h, stride, frame_stride, &dev_stride));
  return run_staged(e, batch, dev_stride, lap0, lap1, out_kp, out_desc, cap, out_n, out_mono);
}

int rgbl_extract(rgbl_extractor* e, const uint8_t* img, int w, int h, int stride, int lap0, int lap1, rgbl_keypoint* out_kp,
                 uint8_t* out_desc, int cap, int* out_n, int* out_mono) {
  return rgbl_extract_batch(e, img, 1, w, h, stride, 0, lap0, lap1, out_kp, out_desc, cap, out_n, out_mono);
}

// ---- ingest: cv::cvtColor in front of the extractor --------------------------------------------
static int enqueue_cvt_gray(rgbl_extractor* e, const uint8_t* d_src, int batch, int channels, int blue_first, int w, int h,
                            int src_stride, size_t src_frame, uint8_t* d_gray, int gray_stride, size_t gray_frame) {
  // COLOR_BGR2GRAY weights channel 0 as blue, COLOR_RGB2GRAY as red (OpenCV 4.x: RY15 9798, GY15 19235, BY15 3735)
  const int w0 = blue_first ? 3735 : 9798, w2 = blue_first ? 9798 : 3735;
  const dim3 grid((w + 255) / 256, (h + 3) / 4, batch);
  e->timer.begin("k_cvt_gray", e->stream);
  if (channels == 3) hipLaunchKernelGGL(k_cvt_gray<3>, grid, dim3(256), 0, e->stream, d_src, src_stride, src_frame, w, h, w0, w2, d_gray, gray_stride, gray_frame);
  else hipLaunchKernelGGL(k_cvt_gray<4>, grid, dim3(256), 0, e->stream, d_src, src_stride, src_frame, w, h, w0, w2, d_gray, gray_stride, gray_frame);
  e->timer.end(e->stream);
  RGBL_HIP(hipGetLastError());
  return RGBL_OK;
}

int rgbl_cvt_gray_batch_device(rgbl_extractor* e, const uint8_t* d_src, int batch, int channels, int blue_first, int w, int h,
                               int src_stride, size_t src_frame_stride, uint8_t* d_gray, int gray_stride,
                               size_t gray_frame_stride) {
  if (!e || !d_src || !d_gray) { set_error("null argument"); return RGBL_ERR_INVALID; }
  if ((channels != 3 && channels != 4) || w < 1 || h < 1 || batch < 1 || src_stride < w * channels || gray_stride < w ||
      (batch > 1 && (src_frame_stride < (size_t)src_stride * h || gray_frame_stride < (size_t)gray_stride * h))) {
    set_error("cvtColor: %d channels, %dx%d, strides %d / %d are not a valid 8-bit colour batch", channels, w, h, src_stride, gray_stride);
    return RGBL_ERR_INVALID;
  }
  RGBL_HIP(hipSetDevice(e->device));
  return enqueue_cvt_gray(e, d_src, batch, channels, blue_first, w, h, src_stride, src_frame_stride, d_gray, gray_stride,
                          gray_frame_stride);
}

static int undistort_params(const float K[4], const float* dist, int n_dist, UndistortParams* U) {
  if (!K || !dist || (n_dist != 4 && n_dist != 5) || K[0] == 0.f || K[1] == 0.f) {
    set_error("undistort: K = (fx, fy, cx, cy) with non-zero focal lengths and 4 or 5 distortion coefficients (k1, k2, p1, p2[, k3])");
    return RGBL_ERR_INVALID;
  }
  U->fx = K[0]; U->fy = K[1]; U->cx = K[2]; U->cy = K[3];
  for (int i = 0; i < 5; ++i) U->k[i] = i < n_dist ? (double)dist[i] : 0.0;
  return RGBL_OK;
}

int rgbl_undistort_keypoints_batch_device(rgbl_extractor* e, const rgbl_keypoint* d_kp, const int32_t* d_n, int batch, int cap,
                                          const float K[4], const float* dist, int n_dist, float* d_xy_un) {
  if (!e || !d_kp || !d_n || !d_xy_un || batch < 1 || cap < 1) { set_error("null / empty argument"); return RGBL_ERR_INVALID; }
  UndistortParams U;
  RGBL_TRY(undistort_params(K, dist, n_dist, &U));
  RGBL_HIP(hipSetDevice(e->device));
  e->timer.begin("k_undistort", e->stream);
  hipLaunchKernelGGL(k_undistort, dim3((cap + 255) / 256, batch), dim3(256), 0, e->stream, U, reinterpret_cast<const float*>(d_kp), 7,
                     (size_t)cap * 7, d_n, 0, d_xy_un, 2, (size_t)cap * 2);
  e->timer.end(e->stream);
  RGBL_HIP(hipGetLastError());
  return RGBL_OK;
}

// Frame / KeyFrame arrays straight from the handles that produced them (include/rgbl_frontend.h: rgbl_device_frame_capture):
// mvKeysUn[].pt (= mvKeys[].pt, or cv::undistortPoints of it), .octave, mDescriptors, mvuRight - device to device.
int rgbl_device_frame_capture(rgbl_device_frame* f, rgbl_extractor* e, int frame, int n, rgbl_depth* depth, const float K[4],
                              const float* dist, int n_dist) {
  if (!f || !e || frame < 0 || frame >= e->cfg.max_batch || n < 0 || n > f->cap || n > e->out_cap) {
    set_error("device frame capture: invalid argument / more keypoints than the frame holds");
    return RGBL_ERR_INVALID;
  }
  if (f->device != e->device) { set_error("device frame and extractor live on different devices"); return RGBL_ERR_INVALID; }
  const float* d_ur = nullptr;
  hipStream_t ds = nullptr;
  if (depth) {
    int k = 0;
    RGBL_TRY(rgbl_internal_depth_uright(depth, &d_ur, &k, &ds));
    if (k != n) { set_error("device frame capture: the depth handle's last call had %d keypoints, the frame %d", k, n); return RGBL_ERR_INVALID; }
  }
  const bool undist = K && dist && n_dist > 0 && dist[0] != 0.0f;   // Frame.cc:839: nothing to do without distortion
  UndistortParams U;
  if (undist) RGBL_TRY(undistort_params(K, dist, n_dist, &U));
  RGBL_HIP(hipSetDevice(e->device));
  hipStream_t s = e->stream;
  RGBL_HIP(hipStreamWaitEvent(s, f->ready, 0));          // an upload of the frame's own still in flight
  if (ds && ds != s) RGBL_TRY(rgbl_stream_wait(s, ds));  // mvuRight is written on the depth handle's stream
  f->n = n;
  f->has_grid = false;   // new keypoints: rgbl_device_frame_set_grid again
  if (n > 0) {
    const rgbl_keypoint* kp = e->d_out_kp + (size_t)frame * e->out_cap;
    hipLaunchKernelGGL(k_frame_capture, dim3((n + 255) / 256), dim3(256), 0, s, kp, e->d_out_desc + (size_t)frame * e->out_cap * 32, n, d_ur,
                       undist ? 0 : 1, f->d_xy, f->d_oct, f->d_ur, f->d_desc);
    if (undist)
      hipLaunchKernelGGL(k_undistort, dim3((n + 255) / 256, 1), dim3(256), 0, s, U, reinterpret_cast<const float*>(kp), 7, (size_t)0,
                         (const int32_t*)nullptr, n, f->d_xy, 2, (size_t)0);
    RGBL_HIP(hipGetLastError());
  }
  RGBL_HIP(hipEventRecord(f->ready, s));
  return RGBL_OK;
}

int rgbl_undistort_points(rgbl_extractor* e, const float* xy, int n, const float K[4], const float* dist, int n_dist, float* out_xy) {
  if (!e || n < 0 || (n > 0 && (!xy || !out_xy))) { set_error("null argument"); return RGBL_ERR_INVALID; }
  UndistortParams U;
  RGBL_TRY(undistort_params(K, dist, n_dist, &U));
  if (n == 0) return RGBL_OK;
  // staging: the two scratch buffers of the lapping-area pass (the result buffers stay what the last extraction left in them -
  // rgbl_device_frame_capture reads them)
  const size_t room = (size_t)e->cfg.max_batch * e->out_cap * sizeof(rgbl_keypoint);
  if ((size_t)n * 2 * sizeof(float) > room) { set_error("undistort: at most %zu points per call with this handle", room / 8); return RGBL_ERR_CAPACITY; }
  RGBL_HIP(hipSetDevice(e->device));
  hipStream_t s = e->stream;
  float* d_in = reinterpret_cast<float*>(e->d_tmp_kp);
  float* d_out = reinterpret_cast<float*>(e->d_tmp_desc);
  RGBL_HIP(hipMemcpyAsync(d_in, xy, sizeof(float) * 2 * n, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_undistort, dim3((n + 255) / 256, 1), dim3(256), 0, s, U, d_in, 2, (size_t)0, (const int32_t*)nullptr, n, d_out, 2, (size_t)0);
  RGBL_HIP(hipGetLastError());
  RGBL_HIP(hipMemcpyAsync(out_xy, d_out, sizeof(float) * 2 * n, hipMemcpyDeviceToHost, s));
  RGBL_HIP(hipStreamSynchronize(s));
  return RGBL_OK;
}

int rgbl_extract_color(rgbl_extractor* e, const uint8_t* img, int channels, int blue_first, int w, int h, int stride, int lap0,
                       int lap1, rgbl_keypoint* out_kp, uint8_t* out_desc, int cap, int* out_n, int* out_mono,
                       uint8_t* out_gray, int gray_stride) {
  if (out_n) *out_n = 0;
  if (out_mono) *out_mono = -1;
  if (!e) { set_error("null handle"); return RGBL_ERR_INVALID; }
  if (!img || w <= 0 || h <= 0) { set_error("empty image"); return RGBL_ERR_EMPTY; }
  if (!out_kp || !out_desc || !out_n || !out_mono) { set_error("null output"); return RGBL_ERR_INVALID; }
  if (channels == 1) {  // GrabImageRGBL leaves single-channel input alone
    if (out_gray) for (int y = 0; y < h; ++y) memcpy(out_gray + (size_t)y * gray_stride, img + (size_t)y * stride, w);
    return rgbl_extract(e, img, w, h, stride, lap0, lap1, out_kp, out_desc, cap, out_n, out_mono);
  }
  if ((channels != 3 && channels != 4) || w != e->cfg.width || h != e->cfg.height || stride < w * channels || cap < 1 ||
      (out_gray && gray_stride < w)) {
    set_error("colour image %dx%dx%d does not match the handle (%dx%d)", w, h, channels, e->cfg.width, e->cfg.height);
    return RGBL_ERR_INVALID;
  }
  RGBL_HIP(hipSetDevice(e->device));
  RGBL_TRY(drop_pending(e));
  hipStream_t s = e->stream;
  const size_t need = (size_t)stride * h + 16;
  if (need > e->color_bytes) {  // first colour frame (or a wider stride): grown once, kept
    if (e->d_color) RGBL_HIP(hipFree(e->d_color));
    e->d_color = nullptr; e->color_bytes = 0;
    RGBL_HIP(hipMalloc(&e->d_color, need));
    e->color_bytes = need;
  }
  RGBL_HIP(hipMemcpyAsync(e->d_color, img, (size_t)(h - 1) * stride + (size_t)w * channels, hipMemcpyHostToDevice, s));
  RGBL_TRY(enqueue_cvt_gray(e, (const uint8_t*)e->d_color, 1, channels, blue_first, w, h, stride, 0, e->d_img, e->img_pitch, 0));
  if (out_gray) RGBL_HIP(hipMemcpy2DAsync(out_gray, gray_stride, e->d_img, e->img_pitch, w, h, hipMemcpyDeviceToHost, s));
  return run_staged(e, 1, e->img_pitch, lap0, lap1, out_kp, out_desc, cap, out_n, out_mono);
}

int rgbl_extractor_level_size(const rgbl_extractor* e, int level, int* w, int* h) {
  if (!e || level < 0 || level >= e->L) { set_error("bad level"); return RGBL_ERR_INVALID; }
  if (w) *w = e->geom[level].w;
  if (h) *h = e->geom[level].h;
  return RGBL_OK;
}

int rgbl_extractor_get_level(rgbl_extractor* e, int frame, int level, int blurred, int with_border, uint8_t* dst,
                             int dst_stride) {
  if (!e || !dst || level < 0 || level >= e->L || frame < 0 || frame >= e->last_batch) {
    set_error("bad level/frame");
    return RGBL_ERR_INVALID;
  }
  RGBL_HIP(hipSetDevice(e->device));
  const LevelGeom& g = e->geom[level];
  const uint8_t* src;
  int pitch;
  if (blurred) { src = e->d_blur + (size_t)frame * e->pyr_frame + g.img_off; pitch = g.pitch; }
  else if (level == 0) { src = e->last_img0 + (size_t)frame * e->last_frame0; pitch = e->last_pitch0; }
  else { src = e->d_pyr + (size_t)frame * e->pyr_frame + g.img_off; pitch = g.pitch; }
  const int border = (with_border && !blurred) ? 19 : 0;
  RGBL_HIP(hipStreamSynchronize(e->stream));
  uint8_t* inner = dst + (size_t)border * dst_stride + border;
  {
    // one linear device -> host transfer, rows are re-strided on the host (a 2-D copy into pageable memory is one
    // transfer per row)
    std::vector<uint8_t> tmp((size_t)pitch * (g.h - 1) + g.w);
    RGBL_HIP(hipMemcpyAsync(tmp.data(), src, tmp.size(), hipMemcpyDeviceToHost, e->stream));
    RGBL_HIP(hipStreamSynchronize(e->stream));
    for (int y = 0; y < g.h; ++y) memcpy(inner + (size_t)y * dst_stride, tmp.data() + (size_t)y * pitch, g.w);
  }
  if (border) {
    // copyMakeBorder(..., BORDER_REFLECT_101 [+BORDER_ISOLATED]) (ORBextractor.cc:1185-1191); host side, it is
    // only consumed by Frame::ComputeStereoMatches
    auto refl = [](int p, int len) { while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p; return p; };
    for (int y = 0; y < g.h; ++y) {
      uint8_t* row = inner + (size_t)y * dst_stride;
      for (int x = 1; x <= border; ++x) { row[-x] = row[refl(-x, g.w)]; row[g.w - 1 + x] = row[refl(g.w - 1 + x, g.w)]; }
    }
    for (int y = 1; y <= border; ++y) {
      memcpy(inner + (ptrdiff_t)(-y) * dst_stride - border, inner + (ptrdiff_t)refl(-y, g.h) * dst_stride - border, g.w + 2 * border);
      memcpy(inner + (ptrdiff_t)(g.h - 1 + y) * dst_stride - border, inner + (ptrdiff_t)refl(g.h - 1 + y, g.h) * dst_stride - border,
             g.w + 2 * border);
    }
  }
  return RGBL_OK;
}

int rgbl_extractor_get_candidates(rgbl_extractor* e, int frame, int level, rgbl_keypoint* out, int cap, int* out_n) {
  if (!e || level < 0 || level >= e->L || frame < 0 || frame >= e->last_batch || !out_n) {
    set_error("bad level/frame");
    return RGBL_ERR_INVALID;
  }
  RGBL_HIP(hipSetDevice(e->device));
  RGBL_HIP(hipStreamSynchronize(e->stream));
  const LevelGeom& g = e->geom[level];
  auto emit = [&](uint32_t key, int n) {
    if (out && n < cap) {
      rgbl_keypoint kp;
      kp.x = (float)(key & 0xfff); kp.y = (float)((key >> 12) & 0xfff); kp.size = 7.f; kp.angle = -1.f;
      kp.response = (float)(key >> 24); kp.octave = 0; kp.class_id = -1;
      out[n] = kp;
    }
  };
  int n = 0;
  if (e->dense) {
    // the level's list in the order the cells finished: back into the reference's order (cell after cell, row-major inside)
    uint32_t C = 0;
    RGBL_HIP(hipMemcpy(&C, e->d_levelcnt + ((size_t)e->cfg.max_batch + frame) * e->L + level, sizeof(uint32_t), hipMemcpyDeviceToHost));
    C = std::min(C, g.key_cap);
    std::vector<uint32_t> keys(C);
    if (C) RGBL_HIP(hipMemcpy(keys.data(), e->d_keys_a + (size_t)frame * e->keys_frame + g.key_off, sizeof(uint32_t) * C, hipMemcpyDeviceToHost));
    auto order = [&](uint32_t key) {
      const uint32_t x = (key & 0xfff) - 3u, y = ((key >> 12) & 0xfff) - 3u;
      const uint32_t col = x / (uint32_t)g.w_cell, row = y / (uint32_t)g.h_cell;
      return ((unsigned long long)(row * (uint32_t)g.n_cols + col) << 14) | ((y - row * g.h_cell) << 7) | (x - col * g.w_cell);
    };
    std::sort(keys.begin(), keys.end(), [&](uint32_t a, uint32_t b) { return order(a) < order(b); });
    for (uint32_t key : keys) emit(key, n++);
  } else {
  std::vector<uint32_t> cnt(g.n_cells);
  RGBL_HIP(hipMemcpy(cnt.data(), e->d_cellcnt + (size_t)frame * e->cells_frame + g.cell_off, sizeof(uint32_t) * g.n_cells,
                     hipMemcpyDeviceToHost));
  std::vector<uint32_t> slots((size_t)g.n_cells * g.cell_cap);
  RGBL_HIP(hipMemcpy(slots.data(), e->d_slots + (size_t)frame * e->slots_frame + g.slot_off, sizeof(uint32_t) * slots.size(),
                     hipMemcpyDeviceToHost));
  for (int c = 0; c < g.n_cells; ++c)
    for (uint32_t k = 0; k < cnt[c]; ++k, ++n) emit(slots[(size_t)c * g.cell_cap + k], n);
  }
  *out_n = n;
  return RGBL_OK;
}

int rgbl_extractor_debug_stamps(rgbl_extractor* e, unsigned long long* out, int count) {
  if (!e || !out) { set_error("null argument"); return RGBL_ERR_INVALID; }
  RGBL_HIP(hipStreamSynchronize(e->stream));
  const size_t n = std::min((size_t)count, (size_t)e->cfg.max_batch * e->L * 16);
  RGBL_HIP(hipMemcpy(out, e->d_dbg, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return RGBL_OK;
}

int rgbl_extractor_set_stream(rgbl_extractor* e, void* hip_stream) {
  if (!e) { set_error("null handle"); return RGBL_ERR_INVALID; }
  RGBL_HIP(hipStreamSynchronize(e->stream));
  e->stream = hip_stream ? (hipStream_t)hip_stream : e->own_stream;
  return RGBL_OK;
}

void* rgbl_extractor_stream(rgbl_extractor* e) { return e ? (void*)e->stream : nullptr; }
void* rgbl_extractor_aux_stream(rgbl_extractor* e) { return e ? (void*)e->aux_stream : nullptr; }

int rgbl_stream_wait(void* waiter, void* signaler) {
  // everything enqueued on `signaler` so far must finish before work enqueued on `waiter` after this call starts
  if (waiter == signaler) return RGBL_OK;
  hipEvent_t ev;
  RGBL_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  RGBL_HIP(hipEventRecord(ev, (hipStream_t)signaler));
  RGBL_HIP(hipStreamWaitEvent((hipStream_t)waiter, ev, 0));
  RGBL_HIP(hipEventDestroy(ev));  // released by the runtime once the recorded work has completed
  return RGBL_OK;
}

int rgbl_event_create(void** ev) {
  if (!ev) { set_error("null argument"); return RGBL_ERR_INVALID; }
  hipEvent_t e;
  RGBL_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  *ev = (void*)e;
  return RGBL_OK;
}
void rgbl_event_destroy(void* ev) { if (ev) (void)hipEventDestroy((hipEvent_t)ev); }
int rgbl_stream_create_on(int device, void** out, int priority) {
  if (!out) { set_error("null argument"); return RGBL_ERR_INVALID; }
  *out = nullptr;
  if (device < 0 || device >= rgbl_device_count()) { set_error("no usable HIP device %d", device); return RGBL_ERR_NO_DEVICE; }
  RGBL_HIP(hipSetDevice(device));
  return rgbl_stream_create(out, priority);
}
int rgbl_stream_create(void** out, int priority) {
  if (!out) { set_error("null argument"); return RGBL_ERR_INVALID; }
  hipStream_t st = nullptr;
#ifdef RGBL_EMU
  (void)priority;
  RGBL_HIP(hipStreamCreate(&st));
#else
  int least = 0, greatest = 0;
  RGBL_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
  RGBL_HIP(hipStreamCreateWithPriority(&st, hipStreamDefault, priority < 0 ? least : priority > 0 ? greatest : 0));   // 0 = the default priority
#endif
  *out = (void*)st;
  return RGBL_OK;
}
void rgbl_stream_destroy(void* st) {
  if (st) { (void)hipStreamSynchronize((hipStream_t)st); (void)hipStreamDestroy((hipStream_t)st); }
}
int rgbl_event_record(void* ev, void* stream) {
  if (!ev) { set_error("null event"); return RGBL_ERR_INVALID; }
  RGBL_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
  return RGBL_OK;
}
int rgbl_event_wait(void* stream, void* ev) {
  if (!ev) { set_error("null event"); return RGBL_ERR_INVALID; }
  RGBL_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0));
  return RGBL_OK;
}

// ---- Frame::ComputeStereoMatches (Frame.cc:901-1071) ---------------------------------------------------------
static int stereo_enqueue(rgbl_extractor* L, rgbl_extractor* R, int batch, const rgbl_keypoint* d_kpl, const uint8_t* d_dl,
                          const int32_t* d_nl, const rgbl_keypoint* d_kpr, const uint8_t* d_dr, const int32_t* d_nr, int cap,
                          float mb, float mbf, float* d_uright, float* d_depth) {
  if (L->cfg.width != R->cfg.width || L->cfg.height != R->cfg.height || L->L != R->L || L->device != R->device ||
      L->cfg.scale_factor != R->cfg.scale_factor) {
    set_error("left and right extractor differ in geometry");
    return RGBL_ERR_INVALID;
  }
  if (batch < 1 || batch > L->last_batch || batch > R->last_batch || !L->last_img0 || !R->last_img0) {
    set_error("stereo matching needs both extractors to have processed this batch (pyramids resident)");
    return RGBL_ERR_INVALID;
  }
  if (!(mb > 0) || cap < 1) { set_error("invalid stereo parameters"); return RGBL_ERR_INVALID; }
  RGBL_HIP(hipSetDevice(L->device));
  const size_t need = (size_t)batch * cap;
  if (need > L->stereo_sad_count) {
    RGBL_HIP(hipStreamSynchronize(L->stream));
    if (L->d_stereo_sad) RGBL_HIP(hipFree(L->d_stereo_sad));
    L->d_stereo_sad = nullptr;
    RGBL_HIP(hipMalloc(&L->d_stereo_sad, need * sizeof(int32_t)));
    L->stereo_sad_count = need;
  }
  RGBL_TRY(rgbl_stream_wait((void*)L->stream, (void*)R->stream));  // the right pyramid and keypoints must be complete
  ScaleTables st;
  for (int i = 0; i < kMaxLevels; ++i) { st.scale[i] = i < L->L ? L->scale[i] : 1.f; st.inv_scale[i] = i < L->L ? L->inv_scale[i] : 1.f; }
  PyrView pl{L->last_img0, L->last_pitch0, L->last_frame0, L->d_pyr, L->pyr_frame};
  PyrView pr{R->last_img0, R->last_pitch0, R->last_frame0, R->d_pyr, R->pyr_frame};
  hipStream_t s = L->stream;
  L->timer.begin("k_stereo_match", s);
  // four waves per workgroup: the right keypoints are staged per workgroup (kStereoTile at a time), a wide group shares that
  const int bs = 256;
  hipLaunchKernelGGL(k_stereo_match, dim3((cap * kStereoLanes + bs - 1) / bs, batch), dim3(bs), 0, s, L->d_geom, st, pl, pr, d_kpl, d_dl, d_nl,
                     d_kpr, d_dr, d_nr, cap, mb, mbf, L->cfg.height, d_uright, d_depth, L->d_stereo_sad);
  L->timer.end(s);
  L->timer.begin("k_stereo_filter", s);
  hipLaunchKernelGGL(k_stereo_filter, dim3(batch), dim3(256), 0, s, d_nl, cap, L->d_stereo_sad, d_uright, d_depth);
  L->timer.end(s);
  RGBL_HIP(hipGetLastError());
  return RGBL_OK;
}

int rgbl_stereo_matches_batch_device(rgbl_extractor* left, rgbl_extractor* right, int batch, const rgbl_keypoint* d_kp_left,
                                     const uint8_t* d_desc_left, const int32_t* d_n_left, const rgbl_keypoint* d_kp_right,
                                     const uint8_t* d_desc_right, const int32_t* d_n_right, int cap, float mb, float mbf,
                                     float* d_uright, float* d_depth) {
  if (!left || !right || !d_kp_left || !d_desc_left || !d_n_left || !d_kp_right || !d_desc_right || !d_n_right || !d_uright ||
      !d_depth) {
    set_error("null argument");
    return RGBL_ERR_INVALID;
  }
  return stereo_enqueue(left, right, batch, d_kp_left, d_desc_left, d_n_left, d_kp_right, d_desc_right, d_n_right, cap, mb, mbf,
                        d_uright, d_depth);
}

int rgbl_stereo_matches(rgbl_extractor* left, rgbl_extractor* right, const rgbl_keypoint* kp_left, const uint8_t* desc_left,
                        int n_left, const rgbl_keypoint* kp_right, const uint8_t* desc_right, int n_right, float mb, float mbf,
                        float* out_uright, float* out_depth) {
  if (!left || !right || n_left < 0 || n_right < 0 || (n_left > 0 && (!kp_left || !desc_left || !out_uright || !out_depth)) ||
      (n_right > 0 && (!kp_right || !desc_right))) {
    set_error("invalid argument");
    return RGBL_ERR_INVALID;
  }
  if (n_left == 0) return RGBL_OK;
  RGBL_HIP(hipSetDevice(left->device));
  const int cap = std::max(std::max(n_left, n_right), 1);
  const size_t rec = (size_t)cap * (sizeof(rgbl_keypoint) + 32);
  const size_t bytes = 2 * rec + 256 + 2 * (size_t)cap * sizeof(float) + 1024;
  if (bytes > left->stereo_stage_bytes) {
    RGBL_HIP(hipStreamSynchronize(left->stream));
    if (left->d_stereo_stage) RGBL_HIP(hipFree(left->d_stereo_stage));
    if (left->h_stereo_stage) RGBL_HIP(hipHostFree(left->h_stereo_stage));
    left->d_stereo_stage = nullptr; left->h_stereo_stage = nullptr; left->stereo_stage_bytes = 0;
    RGBL_HIP(hipMalloc(&left->d_stereo_stage, bytes));
    RGBL_HIP(hipHostMalloc(reinterpret_cast<void**>(&left->h_stereo_stage), bytes, hipHostMallocDefault));
    left->stereo_stage_bytes = bytes;
  }
  uint8_t* base = (uint8_t*)left->d_stereo_stage;
  rgbl_keypoint* d_kpl = (rgbl_keypoint*)base;
  rgbl_keypoint* d_kpr = (rgbl_keypoint*)(base + (size_t)cap * sizeof(rgbl_keypoint));
  uint8_t* d_dl = base + 2 * (size_t)cap * sizeof(rgbl_keypoint);
  uint8_t* d_dr = d_dl + (size_t)cap * 32;
  int32_t* d_n = (int32_t*)(d_dr + (size_t)cap * 32);
  float* d_u = (float*)((uint8_t*)d_n + 256);
  float* d_d = d_u + cap;
  hipStream_t s = left->stream;
  // the page-locked mirror has the device block's layout: keypoints and descriptors of both views and the counts go up with
  // ONE request, uRight and depth come back with one (round 5: five uploads and two read-backs from / to pageable memory)
  uint8_t* hb = left->h_stereo_stage;
  const int32_t counts[2] = {n_left, n_right};
  memcpy(hb, kp_left, sizeof(rgbl_keypoint) * n_left);
  memcpy(hb + ((uint8_t*)d_dl - base), desc_left, (size_t)32 * n_left);
  if (n_right > 0) {
    memcpy(hb + ((uint8_t*)d_kpr - base), kp_right, sizeof(rgbl_keypoint) * n_right);
    memcpy(hb + ((uint8_t*)d_dr - base), desc_right, (size_t)32 * n_right);
  }
  memcpy(hb + ((uint8_t*)d_n - base), counts, sizeof(counts));
  RGBL_HIP(hipMemcpyAsync(base, hb, (size_t)((uint8_t*)d_n - base) + sizeof(counts), hipMemcpyHostToDevice, s));
  RGBL_TRY(stereo_enqueue(left, right, 1, d_kpl, d_dl, d_n, d_kpr, d_dr, d_n + 1, cap, mb, mbf, d_u, d_d));
  float* h_u = reinterpret_cast<float*>(hb + ((uint8_t*)d_u - base));
  RGBL_HIP(hipMemcpyAsync(h_u, d_u, sizeof(float) * ((size_t)cap + n_left), hipMemcpyDeviceToHost, s));
  RGBL_HIP(hipStreamSynchronize(s));
  memcpy(out_uright, h_u, sizeof(float) * n_left);
  memcpy(out_depth, h_u + cap, sizeof(float) * n_left);
  left->timer.collect();
  return RGBL_OK;
}

int rgbl_extractor_profile(rgbl_extractor* e, int enable) {
  if (!e) { set_error("null handle"); return RGBL_ERR_INVALID; }
  RGBL_HIP(hipStreamSynchronize(e->stream));
  e->timer.reset();
  e->timer.enabled = enable != 0;
  return RGBL_OK;
}

int rgbl_extractor_profile_read(rgbl_extractor* e, const char** names, double* total_ms, long* launches, int cap) {
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
int rgbl_extractor_profile_samples(rgbl_extractor* e, int kernel, float* ms, int cap) {
  if (!e || (cap > 0 && !ms)) return 0;
  (void)hipStreamSynchronize(e->stream);
  e->timer.collect();
  return e->timer.read_samples(kernel, ms, cap);
}

// Self-test of the hand-written instruction wrappers on the device (see k_selftest_wrappers): n random inputs, every result
// compared with its plain expression on the host.  RGBL_OK, or RGBL_ERR_HIP with the first mismatch in rgbl_last_error().
int rgbl_selftest_wrappers(int device, int n, unsigned seed) {
  using namespace rgbl;
  if (n < 1 || n > (1 << 20)) { set_error("selftest: 1 <= n <= 2^20"); return RGBL_ERR_INVALID; }
  if (rgbl_device_count() <= device || device < 0) { set_error("no usable HIP device %d (this library has no CPU fallback)", device); return RGBL_ERR_NO_DEVICE; }
  RGBL_HIP(hipSetDevice(device));
  std::vector<uint32_t> a(n), fl(n);
  std::vector<uint8_t> src((size_t)16 * n + 32);
  uint32_t st = seed * 2654435761u + 12345u;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return st ^ (st >> 15); };
  for (int i = 0; i < n; ++i) { a[i] = rnd(); fl[i] = rnd(); }
  for (auto& v : src) v = (uint8_t)(rnd() >> 9);
  const uint32_t b = rnd() & 0xffffffu;
  uint32_t *d_a = nullptr, *d_fl = nullptr, *d_mul = nullptr;
  int* d_add = nullptr;
  uint8_t *d_src = nullptr, *d_dma = nullptr;
  std::vector<uint32_t> mul(n, 0xdeadbeefu);
  std::vector<int> add(n, 0x5a5a5a5a);
  std::vector<uint8_t> dma((size_t)16 * n, 0);
  int rc = RGBL_OK;
  auto run = [&]() -> int {
    RGBL_HIP(hipMalloc(&d_a, sizeof(uint32_t) * n)); RGBL_HIP(hipMalloc(&d_fl, sizeof(uint32_t) * n));
    RGBL_HIP(hipMalloc(&d_mul, sizeof(uint32_t) * n)); RGBL_HIP(hipMalloc(&d_add, sizeof(int) * n));
    RGBL_HIP(hipMalloc(&d_src, src.size())); RGBL_HIP(hipMalloc(&d_dma, dma.size()));
    RGBL_HIP(hipMemcpy(d_a, a.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice));
    RGBL_HIP(hipMemcpy(d_fl, fl.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice));
    RGBL_HIP(hipMemcpy(d_src, src.data(), src.size(), hipMemcpyHostToDevice));
    RGBL_HIP(hipMemcpy(d_mul, mul.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice));
    RGBL_HIP(hipMemcpy(d_add, add.data(), sizeof(int) * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_selftest_wrappers, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t) nullptr, d_a, b, d_fl, d_src, d_mul, d_add, d_dma, n);
    RGBL_HIP(hipGetLastError());
    RGBL_HIP(hipMemcpy(mul.data(), d_mul, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
    RGBL_HIP(hipMemcpy(add.data(), d_add, sizeof(int) * n, hipMemcpyDeviceToHost));
    RGBL_HIP(hipMemcpy(dma.data(), d_dma, dma.size(), hipMemcpyDeviceToHost));
    return RGBL_OK;
  };
  rc = run();
  for (void* p : {(void*)d_a, (void*)d_fl, (void*)d_mul, (void*)d_add, (void*)d_src, (void*)d_dma}) if (p) (void)hipFree(p);
  if (rc != RGBL_OK) return rc;
  for (int i = 0; i < n; ++i) {
    const bool active = (a[i] & 3u) != 0u;
    const uint32_t want_mul = active ? (a[i] & 0xffffffu) * b : 0xdeadbeefu;
    const int want_add = active ? (int)a[i] + (int)(fl[i] & 1u) : 0x5a5a5a5a;
    if (mul[i] != want_mul) { set_error("selftest: mul24(%#x, %#x) = %#x, expected %#x (lane %d)", a[i] & 0xffffffu, b, mul[i], want_mul, i); return RGBL_ERR_HIP; }
    if (add[i] != want_add) { set_error("selftest: add_flag(%d, %u) = %d, expected %d (lane %d)", (int)a[i], fl[i] & 1u, add[i], want_add, i); return RGBL_ERR_HIP; }
    const bool moved = (i % 64) % 5 != 4;
    for (int k = 0; k < 16; ++k) {
      const uint8_t want = moved ? src[(size_t)16 * i + (i & 3) + k] : (uint8_t)0xee;
      if (dma[(size_t)16 * i + k] != want) { set_error("selftest: lds_dma16 lane %d byte %d = %#x, expected %#x", i, k, dma[(size_t)16 * i + k], want); return RGBL_ERR_HIP; }
    }
  }
  return RGBL_OK;
}

#ifdef RGBL_EMU
// test hook (emulation build only): the libstdc++ introsort restatement on plain arrays
void rgbl_test_std_sort(uint64_t* key, uint32_t* val, int n) { rgbl::std_sort_restated(key, val, n); }
void rgbl_test_block_sort(uint64_t* key, uint32_t* val, int n) {
  hipLaunchKernelGGL(rgbl::k_test_block_sort, dim3(1), dim3(rgbl::kOctWide), 0, (hipStream_t) nullptr, key, val, n);
}
#endif

}  // extern "C"

#ifdef RGBL_EMU
// test hooks (emulation build only)
extern "C" void rgbl_test_sincosf(float x, float* s, float* c) { *s = rgbl::glibc_sinf(x); *c = rgbl::glibc_cosf(x); }
#endif
