// common.h — shared host/device helpers for the rgbl front-end kernels (gfx950 / wave64).
#pragma once

#ifdef RGBL_EMU
// CPU SIMT emulation used ONLY by the test-suite (tests/emu/hip_emu.h); never part of the product build.
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#endif

#include <stdint.h>
#include <stdio.h>

#include <string>
#include <vector>

#include "../../include/rgbl_frontend.h"

namespace rgbl {

// ---- error plumbing: the C ABI never throws; every entry point returns a status and records a message
void set_error(const char* fmt, ...);
const char* last_error();

#define RGBL_HIP(expr)                                                                        \
  do {                                                                                        \
    hipError_t e__ = (expr);                                                                  \
    if (e__ != hipSuccess) {                                                                  \
      rgbl::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
      return RGBL_ERR_HIP;                                                                    \
    }                                                                                         \
  } while (0)

#define RGBL_TRY(expr)            \
  do {                            \
    int rc__ = (expr);            \
    if (rc__ != RGBL_OK) return rc__; \
  } while (0)

constexpr int kWave = 64;  // gfx950 wavefront width; hard-coded on purpose

// Host-pointer entry points queue uploads from caller / stack buffers; whichever way such a function is left - also through
// an RGBL_HIP / RGBL_TRY error return - the stream is drained first, so nothing still reads memory the caller frees next.
struct StreamDrain {
  hipStream_t s;
  explicit StreamDrain(hipStream_t stream) : s(stream) {}
  ~StreamDrain() { (void)hipStreamSynchronize(s); }
  StreamDrain(const StreamDrain&) = delete;
  StreamDrain& operator=(const StreamDrain&) = delete;
};

// ---- per-kernel timing (HIP events on the launch stream), used by bench.py's roofline leg
struct KernelTimer {
  struct Rec { int id; hipEvent_t a, b; };
  bool enabled = false;
  std::vector<Rec> recs;
  std::vector<double> total_ms;
  std::vector<long> count;
  static constexpr size_t kMaxSamples = 1 << 16;
  std::vector<std::vector<float>> samples;   // every launch's duration, in launch order (rgbl_*_profile_samples), the first kMaxSamples of them
  std::vector<std::string> names;
  int id_of(const char* name) {
    for (size_t i = 0; i < names.size(); ++i)
      if (names[i] == name) return (int)i;
    names.push_back(name);
    total_ms.push_back(0);
    count.push_back(0);
    samples.emplace_back();
    return (int)names.size() - 1;
  }
  void begin(const char* name, hipStream_t s) {
    if (!enabled) return;
    Rec r;
    r.id = id_of(name);
    (void)hipEventCreate(&r.a);
    (void)hipEventCreate(&r.b);
    (void)hipEventRecord(r.a, s);
    recs.push_back(r);
  }
  void end(hipStream_t s) {
    if (!enabled) return;
    (void)hipEventRecord(recs.back().b, s);
  }
  // call after the stream has been synchronised
  void collect() {
    for (Rec& r : recs) {
      float ms = 0;
      (void)hipEventSynchronize(r.b);
      (void)hipEventElapsedTime(&ms, r.a, r.b);
      total_ms[r.id] += ms;
      count[r.id] += 1;
      if (samples[r.id].size() < kMaxSamples) samples[r.id].push_back(ms);   // profiling left on for hours must not grow without bound
      (void)hipEventDestroy(r.a);
      (void)hipEventDestroy(r.b);
    }
    recs.clear();
  }
  void reset() {
    collect();
    for (size_t i = 0; i < total_ms.size(); ++i) { total_ms[i] = 0; count[i] = 0; samples[i].clear(); }
  }
  // the durations of kernel `id`'s launches since the last reset (call collect() first); returns how many there are
  int read_samples(int id, float* out, int cap) const {
    if (id < 0 || id >= (int)samples.size()) return 0;
    const int n = (int)samples[id].size();
    for (int i = 0; i < n && i < cap; ++i) out[i] = samples[id][i];
    return n;
  }
};

}  // namespace rgbl

// ---- a frame / key frame resident on the device (include/rgbl_frontend.h: rgbl_device_frame_*) -------------------
// One allocation: descriptors [cap x 32] | xy [cap x 2] | octave [cap] | uright [cap]; the FeatureVector's CSR arrays in a
// second, grow-only one.  `ready` is recorded behind whatever filled the arrays last (an upload on the frame's own stream, a
// capture on the extractor's); a matcher call makes its stream wait for it.
struct rgbl_device_frame {
  int device = 0, cap = 0, n = 0;
  uint8_t* block = nullptr;
  uint8_t* d_desc = nullptr;
  float* d_xy = nullptr;
  int32_t* d_oct = nullptr;
  float* d_ur = nullptr;
  // Frame::AssignFeaturesToGrid, kept with the frame (rgbl_device_frame_set_grid): the projection searches skip their grid build
  uint32_t* d_cell_start = nullptr;   // 64 x 48 + 1
  uint16_t* d_cell_items = nullptr;   // cap
  int32_t* d_grid_scratch = nullptr;  // cap (what k_proj_grid initialises besides the grid)
  float grid[6] = {0, 0, 0, 0, 0, 0};
  bool has_grid = false;
  int32_t* d_fv = nullptr;      // node_off [n_nodes + 1] | node_feat [nf]
  int fv_cap = 0, n_nodes = -1, nf = 0;
  uint8_t* h_pin = nullptr;     // page-locked mirror of one upload
  size_t pin_size = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ready = nullptr;
};
// where the last host-pointer call of a handle left its results on the device (extractor.hip / depth.hip; not exported C ABI)
int rgbl_internal_depth_uright(rgbl_depth* d, const float** d_uright, int* k, hipStream_t* stream);

namespace rgbl {

// ---- device helpers ----------------------------------------------------------------------------
__device__ __forceinline__ int cv_round_f(float v) { return __float2int_rn(v); }  // round-half-even
__device__ __forceinline__ unsigned long long rgbl_clock() {
#ifdef RGBL_EMU
  return 0;
#else
  return (unsigned long long)__builtin_readcyclecounter();
#endif
}
#ifdef RGBL_EMU
typedef emu_v4i v4i;    // matrix-core operand: 16 signed bytes per lane
typedef emu_v16i v16i;  // matrix-core accumulator of a 32 x 32 tile: 16 x i32 per lane
typedef emu_v8i v8i;    // operand of the block-scaled f8f6f4 matrix-core instructions (FP4: the first four dwords)
typedef emu_v16f v16f;
#else
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
#endif
typedef float f32x2 __attribute__((vector_size(8)));  // two fp32 lanes of one packed VALU operation
// integer dot products and byte shuffles of the VALU (v_dot4_u32_u8, v_dot2_u32_u16, v_alignbyte_b32, v_perm_b32)
__device__ __forceinline__ uint32_t udot4(uint32_t a, uint32_t b, uint32_t c) {  // sum of the four byte products + c
#ifdef RGBL_EMU
  for (int k = 0; k < 4; ++k) c += ((a >> (8 * k)) & 0xffu) * ((b >> (8 * k)) & 0xffu);
  return c;
#else
  return __builtin_amdgcn_udot4(a, b, c, false);
#endif
}
__device__ __forceinline__ uint32_t udot2(uint32_t a, uint32_t b, uint32_t c) {  // sum of the two 16-bit products + c
#ifdef RGBL_EMU
  return c + (a & 0xffffu) * (b & 0xffffu) + (a >> 16) * (b >> 16);
#else
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  us2 x, y;
  __builtin_memcpy(&x, &a, 4);
  __builtin_memcpy(&y, &b, 4);
  return __builtin_amdgcn_udot2(x, y, c, false);
#endif
}
__device__ __forceinline__ uint32_t sad_u8(uint32_t a, uint32_t b, uint32_t c) {  // c + sum of |a.byte - b.byte| over the four bytes (v_sad_u8)
#ifdef RGBL_EMU
  for (int k = 0; k < 4; ++k) { const int d = (int)((a >> (8 * k)) & 0xffu) - (int)((b >> (8 * k)) & 0xffu); c += (uint32_t)(d < 0 ? -d : d); }
  return c;
#else
  return __builtin_amdgcn_sad_u8(a, b, c);
#endif
}
__device__ __forceinline__ uint32_t align_bytes(uint32_t hi, uint32_t lo, int shift) {  // bytes shift .. shift + 3 of hi:lo
#ifdef RGBL_EMU
  return (uint32_t)((((unsigned long long)hi << 32) | lo) >> (8 * (shift & 3)));
#else
  return __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)shift);
#endif
}
// v_perm_b32: byte i of the result is byte sel[i] of the 8 bytes hi:lo (0-3 = lo, 4-7 = hi), 0x0c gives 0x00
__device__ __forceinline__ uint32_t perm_bytes(uint32_t hi, uint32_t lo, uint32_t sel) {
#ifdef RGBL_EMU
  const unsigned long long v = ((unsigned long long)hi << 32) | lo;
  uint32_t r = 0;
  for (int i = 0; i < 4; ++i) {
    const uint32_t s = (sel >> (8 * i)) & 0xffu;
    const uint32_t b = s < 8 ? (uint32_t)(v >> (8 * s)) & 0xffu : s == 0x0c ? 0u : s > 0x0c ? 0xffu : ((v >> (16 * (s - 8) + 15)) & 1 ? 0xffu : 0u);
    r |= b << (8 * i);
  }
  return r;
#else
  return __builtin_amdgcn_perm(hi, lo, sel);
#endif
}
// packed 16-bit VALU operations on two unsigned halves of a word (v_pk_min_u16, v_pk_max_u16, v_pk_sub_u16 clamp)
#ifdef RGBL_EMU
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
  const uint32_t l = (a & 0xffffu) < (b & 0xffffu) ? (a & 0xffffu) : (b & 0xffffu), h = (a >> 16) < (b >> 16) ? (a >> 16) : (b >> 16);
  return l | (h << 16);
}
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
  const uint32_t l = (a & 0xffffu) > (b & 0xffffu) ? (a & 0xffffu) : (b & 0xffffu), h = (a >> 16) > (b >> 16) ? (a >> 16) : (b >> 16);
  return l | (h << 16);
}
__device__ __forceinline__ uint32_t pk_subs_u16(uint32_t a, uint32_t b) {  // saturating a - b per half
  const uint32_t al = a & 0xffffu, bl = b & 0xffffu, ah = a >> 16, bh = b >> 16;
  return (al > bl ? al - bl : 0u) | ((ah > bh ? ah - bh : 0u) << 16);
}
__device__ __forceinline__ uint32_t pk_adds_u16(uint32_t a, uint32_t b) {  // saturating a + b per half
  const uint32_t l = (a & 0xffffu) + (b & 0xffffu), h = (a >> 16) + (b >> 16);
  return (l > 0xffffu ? 0xffffu : l) | ((h > 0xffffu ? 0xffffu : h) << 16);
}
#else
typedef unsigned short rgbl_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ rgbl_us2 as_us2(uint32_t a) { rgbl_us2 x; __builtin_memcpy(&x, &a, 4); return x; }
__device__ __forceinline__ uint32_t from_us2(rgbl_us2 x) { uint32_t a; __builtin_memcpy(&a, &x, 4); return a; }
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) { return from_us2(__builtin_elementwise_min(as_us2(a), as_us2(b))); }
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) { return from_us2(__builtin_elementwise_max(as_us2(a), as_us2(b))); }
__device__ __forceinline__ uint32_t pk_subs_u16(uint32_t a, uint32_t b) { return from_us2(__builtin_elementwise_sub_sat(as_us2(a), as_us2(b))); }
__device__ __forceinline__ uint32_t pk_adds_u16(uint32_t a, uint32_t b) { return from_us2(__builtin_elementwise_add_sat(as_us2(a), as_us2(b))); }
#endif
// dynamic LDS of a kernel (sized per launch); the emulator gives every workgroup thread a buffer of the hardware's size
#ifdef RGBL_EMU
#define RGBL_DYN_SHARED(T, name) static thread_local T name[(160 * 1024) / sizeof(T)]
#else
#define RGBL_DYN_SHARED(T, name) extern __shared__ T name[]
#endif
__device__ __forceinline__ uint32_t load_u32_any(const uint8_t* p) {  // global or LDS, no alignment needed on gfx950
  uint32_t v;
  __builtin_memcpy(&v, p, 4);
  return v;
}
// XCD-aware launch geometry of the per-frame pixel kernels.  The hardware hands consecutive workgroups (x fastest, then y,
// then z) to the 8 XCDs round-robin, each XCD with its own L2: with a plain (items, frames) grid the workgroups that share
// halo pixels / cache lines of one frame sit on 8 different L2s and every line leaves the fabric up to 8 times.  With the
// grid (8, items, frames / 8) blockIdx.x IS the XCD, and XCD x works on the frames z * 8 + x: the neighbours of a frame
// share an L2 (FETCH_SIZE of k_fast_cells: 4.3 x the image bytes before, 0.9 x after).  Frame counts that are not a
// multiple of 8 (or RGBL_XCD_MAP=0) take the grid (1, items, frames).  grid.y is limited to 65535: more items per frame than
// that (a 4096 x 4096 image at scale 1.1 with 16 levels has more detection cells) take the plain grid (items, 1, frames),
// which the kernels recognise by gridDim.x > 8.
inline dim3 xcd_grid(bool enabled, unsigned items, unsigned frames) {
  if (items > 65535u) return dim3(items, 1, frames);
  return (enabled && frames % 8u == 0u) ? dim3(8, items, frames / 8u) : dim3(1, items, frames);
}
__device__ __forceinline__ int xcd_frame() { return gridDim.x > 8u ? (int)blockIdx.z : (int)(blockIdx.z * gridDim.x + blockIdx.x); }
__device__ __forceinline__ int xcd_item() { return gridDim.x > 8u ? (int)blockIdx.x : (int)blockIdx.y; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ int wave_id() { return (int)(threadIdx.x >> 6); }
__device__ __forceinline__ unsigned long long lanemask_lt() { return (1ull << lane_id()) - 1ull; }
__device__ __forceinline__ int reflect101(int p, int len) {
  // BORDER_REFLECT_101 for |overshoot| < len (always true for the 3/19-px borders used here)
  if (p < 0) p = -p;
  if (p >= len) p = 2 * (len - 1) - p;
  return imin(imax(p, 0), len - 1);  // far-out-of-range taps only feed outputs that are never stored
}

// Orders LDS traffic between the lanes of ONE wave (cross-lane hand-off through LDS without a workgroup barrier):
// on gfx950 the lanes run in lockstep, so this only has to stop the compiler from moving memory operations.
__device__ __forceinline__ void wave_sync() {
#ifdef RGBL_EMU
  (void)__ballot(1);  // the emulator's lanes are independent fibers: rendezvous
#else
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

template <class T>
__device__ __forceinline__ T wave_sum(T v) {
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

// Sum of an int over the 64 lanes as a wave-uniform value: four DPP steps inside the rows of 16 (quad swaps, half-row and
// row mirror), two row broadcasts, one v_readlane - register-file traffic only (wave_sum's __shfl_xor is six ds_bpermute
// round trips through the LDS crossbar).
__device__ __forceinline__ int wave_sum_uniform(int v) {
#ifdef RGBL_EMU
  return __shfl(wave_sum(v), 0);
#else
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);  // row_half_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);  // row_mirror: every lane holds its row's sum
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
  return __builtin_amdgcn_readlane(v, 63);
#endif
}

// Minimum of an unsigned value over the 64 lanes as a wave-uniform value, the same DPP ladder as wave_sum_uniform (lanes a
// step has no source for keep their own value: the identity 0xffffffff comes in as `old`).
__device__ __forceinline__ uint32_t wave_min_uniform(uint32_t v) {
#ifdef RGBL_EMU
  for (int m = 32; m >= 1; m >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)v, m); v = o < v ? o : v; }
  return v;
#else
  auto mn = [](uint32_t a, int b) { return (uint32_t)b < a ? (uint32_t)b : a; };
  v = mn(v, __builtin_amdgcn_update_dpp(-1, (int)v, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  v = mn(v, __builtin_amdgcn_update_dpp(-1, (int)v, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  v = mn(v, __builtin_amdgcn_update_dpp(-1, (int)v, 0x141, 0xF, 0xF, false));  // row_half_mirror
  v = mn(v, __builtin_amdgcn_update_dpp(-1, (int)v, 0x140, 0xF, 0xF, false));  // row_mirror: every lane holds its row's minimum
  v = mn(v, __builtin_amdgcn_update_dpp(-1, (int)v, 0x142, 0xA, 0xF, false));  // row_bcast:15 into rows 1 and 3
  v = mn(v, __builtin_amdgcn_update_dpp(-1, (int)v, 0x143, 0xC, 0xF, false));  // row_bcast:31 into rows 2 and 3
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
#endif
}

// Exclusive prefix sum over the 256 work-items of a workgroup (4 waves). `scratch` holds >= 8 values.
// Returns the exclusive prefix of `v`; *total receives the workgroup sum. Contains two barriers.
// Inclusive prefix sum over the 64 lanes.  32-bit values: Hillis-Steele inside the rows of 16 with four row_shr DPP adds,
// then the two row broadcasts - register-file traffic only; __shfl_up is a ds_bpermute round trip per step.
template <class T>
__device__ __forceinline__ T wave_inclusive_scan(T v) {
#ifndef RGBL_EMU
  if constexpr (sizeof(T) == 4) {
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);   // row_shr:1 (lanes without a source add 0)
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
    return (T)x;
  }
#endif
  const int lane = lane_id();
  T incl = v;
  for (int d = 1; d < 64; d <<= 1) {
    T up = __shfl_up(incl, d);
    if (lane >= d) incl += up;
  }
  return incl;
}

template <class T>
__device__ __forceinline__ T block_exclusive_scan(T v, T* scratch, T* total) {
  const int lane = lane_id(), w = wave_id();
  const T incl = wave_inclusive_scan(v);
  if (lane == 63) scratch[w] = incl;
  __syncthreads();
  const int nw = (int)((blockDim.x + 63) >> 6);
  T base = 0, tot = 0;
  for (int i = 0; i < nw; ++i) {
    const T s = scratch[i];
    if (i < w) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

}  // namespace rgbl
