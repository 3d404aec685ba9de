// hip_emu.h — a tiny SIMT emulator so the HIP kernel SOURCES of this repo can be executed on a CPU.
//
// TEST INFRASTRUCTURE ONLY.  This container has no GPU and GPU minutes are scarce, so the kernels in
// orb_slam3_rgbl_amd/csrc/*.hip are additionally compiled by g++ against this header (-DRGBL_EMU) into
// tests/_build/librgbl_frontend_emu.so, and the CPU test-suite checks their LOGIC against the oracle
// before anything is sent to a real MI355X.  The emulated library is never shipped, never loaded by the
// product package and never timed.
//
// Model: every workgroup runs on one OS thread; its work-items are ucontext fibers that are resumed
// round-robin.  __syncthreads() and the wave64 collectives (__ballot/__shfl*) are rendezvous points.
// The resume order flips between ascending and descending lane order at every rendezvous (or is
// shuffled, RGBL_EMU_ORDER=shuffle) so that code relying on an accidental execution order, i.e. a
// missing barrier, tends to fail here instead of passing silently.  Workgroups of one launch are
// spread over a pool of OS threads; `__shared__` is `static thread_local`, i.e. private to the
// workgroup currently executing on that OS thread.
#pragma once
#include <ucontext.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static thread_local
#define __launch_bounds__(...)
#define __restrict__

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_uint3 { unsigned x, y, z; };
struct uchar4 { unsigned char x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int4 { int x, y, z, w; };
inline float4 make_float4(float x, float y, float z, float w) { float4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
inline int4 make_int4(int x, int y, int z, int w) { int4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { uint4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }

namespace hipemu {

struct Rendezvous {
  int arrived = 0;
  unsigned gen = 0;
};

// Context switches: glibc's swapcontext / getcontext make a rt_sigprocmask system call each - with two switches per yield that
// was more than half of the emulated suite's time (20 of 60 CPU minutes in the kernel).  On x86-64 without Address- / Thread-
// Sanitizer (which follow swapcontext, not a hand-written switch) the fibers are switched by 14 instructions: the callee-saved
// registers on the fiber's own stack, the stack pointers exchanged.  Everything else keeps ucontext.
#if defined(__x86_64__) && !defined(__SANITIZE_ADDRESS__) && !defined(__SANITIZE_THREAD__) && !defined(HIPEMU_UCONTEXT)
#define HIPEMU_FAST_SWITCH 1
extern "C" void hipemu_switch(void** save_sp, void* const* load_sp);
#endif

struct Block;
struct Fiber {
  ucontext_t ctx;
  void* sp = nullptr;     // HIPEMU_FAST_SWITCH: the suspended fiber's stack pointer
  Block* blk = nullptr;
  int tid = 0;
  bool done = false;
  emu_uint3 tidx{0, 0, 0};
  char* stack = nullptr;
  void* tsan = nullptr;   // ThreadSanitizer's view of this fiber (-fsanitize=thread builds only)
};

struct Block {
  ucontext_t sched;
  void* sched_sp = nullptr;
  std::vector<Fiber> fibers;
  int nthreads = 0;
  int active = 0;               // fibers not yet finished
  std::vector<int> wave_active; // per wave
  Rendezvous bar;               // __syncthreads
  std::vector<Rendezvous> wbar; // per wave collectives
  std::vector<unsigned long long> slot;  // exchange slots, one per work-item
  std::vector<unsigned char> wide;       // 32-byte exchange slots, one per work-item (matrix-core operands)
  unsigned phase = 0;
  std::function<void()> body;
  Fiber* cur = nullptr;
  void* tsan_sched = nullptr;
};

extern thread_local Block* g_blk;

inline Block*& cur_block() { return g_blk; }

void yield_to_scheduler();
void run_block(Block& b, const dim3& block);
void launch(const dim3& grid, const dim3& block, const std::function<void()>& body);

}  // namespace hipemu

extern thread_local emu_uint3 threadIdx;
extern thread_local emu_uint3 blockIdx;
extern thread_local dim3 blockDim;
extern thread_local dim3 gridDim;

// ------------------------------------------------------------------ device intrinsics
namespace hipemu {
inline void rendezvous(Rendezvous& r, int& active) {
  unsigned my = r.gen;
  if (++r.arrived >= active) {
    r.arrived = 0;
    ++r.gen;
    ++cur_block()->phase;
    return;
  }
  while (r.gen == my) yield_to_scheduler();
}
inline int lane_id() { return cur_block()->cur->tid & 63; }
inline int wave_id() { return cur_block()->cur->tid >> 6; }

// all (still running) lanes of the wave deposit a value, then read any lane's value
template <class F>
inline unsigned long long wave_collective(unsigned long long mine, F reader) {
  Block* b = cur_block();
  const int w = wave_id(), tid = b->cur->tid;
  b->slot[tid] = mine;
  rendezvous(b->wbar[2 * w], b->wave_active[w]);
  unsigned long long res = reader(&b->slot[(size_t)w * 64]);
  rendezvous(b->wbar[2 * w + 1], b->wave_active[w]);
  return res;
}
inline unsigned long long lane_active_mask() {
  Block* b = cur_block();
  const int w = wave_id();
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l) {
    int t = w * 64 + l;
    if (t < b->nthreads && !b->fibers[t].done) m |= 1ull << l;
  }
  return m;
}
}  // namespace hipemu

inline void __syncthreads() {
  hipemu::Block* b = hipemu::cur_block();
  hipemu::rendezvous(b->bar, b->active);
}
inline unsigned long long __ballot(int pred) {
  const unsigned long long act = hipemu::lane_active_mask();
  return hipemu::wave_collective(pred ? 1ull : 0ull, [&](const unsigned long long* s) {
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l)
      if (((act >> l) & 1) && s[l]) m |= 1ull << l;
    return m;
  });
}
template <class T>
inline T emu_shfl_generic(T v, int src_lane) {
  static_assert(sizeof(T) <= 8, "shfl type too wide");
  unsigned long long bits = 0;
  std::memcpy(&bits, &v, sizeof(T));
  const int me = hipemu::lane_id();
  unsigned long long r = hipemu::wave_collective(bits, [&](const unsigned long long* s) {
    int sl = src_lane;
    if (sl < 0 || sl > 63) sl = me;
    return s[sl];
  });
  T out;
  std::memcpy(&out, &r, sizeof(T));
  return out;
}
template <class T> inline T __shfl(T v, int src, int width = 64) {
  const int me = hipemu::lane_id();
  const int base = me & ~(width - 1);
  return emu_shfl_generic(v, base + (src & (width - 1)));
}
template <class T> inline T __shfl_xor(T v, int mask, int width = 64) {
  const int me = hipemu::lane_id();
  int src = me ^ mask;
  if ((src & ~(width - 1)) != (me & ~(width - 1))) src = me;
  return emu_shfl_generic(v, src);
}
template <class T> inline T __shfl_up(T v, unsigned delta, int width = 64) {
  const int me = hipemu::lane_id();
  int src = me - (int)delta;
  if (src < (me & ~(width - 1))) src = me;
  return emu_shfl_generic(v, src);
}
template <class T> inline T __shfl_down(T v, unsigned delta, int width = 64) {
  const int me = hipemu::lane_id();
  int src = me + (int)delta;
  if (src > ((me & ~(width - 1)) + width - 1)) src = me;
  return emu_shfl_generic(v, src);
}

// v_mfma_i32_32x32x32_i8: D = A * B + C on a 32 x 32 tile with K = 32, one instruction per wave.  Operand layout as on
// gfx950: lane l holds 16 consecutive k of row (A) / column (B) l & 31, the k-block chosen by l >> 5; lane l receives
// column l & 31 of D, register r = row (r & 3) + 8 * (r >> 2) + 4 * (l >> 5).
typedef int emu_v4i __attribute__((vector_size(16)));
typedef int emu_v16i __attribute__((vector_size(64)));
inline emu_v16i __builtin_amdgcn_mfma_i32_32x32x32_i8(emu_v4i a, emu_v4i b, emu_v16i c, int, int, int) {
  hipemu::Block* blk = hipemu::cur_block();
  const int w = hipemu::wave_id(), lane = hipemu::lane_id();
  unsigned char* base = &blk->wide[(size_t)w * 64 * 32];
  std::memcpy(base + lane * 32, &a, 16);
  std::memcpy(base + lane * 32 + 16, &b, 16);
  hipemu::rendezvous(blk->wbar[2 * w], blk->wave_active[w]);
  emu_v16i d = c;
  const int col = lane & 31;
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    int acc = 0;
    for (int g = 0; g < 2; ++g) {
      const signed char* pa = (const signed char*)(base + (row + 32 * g) * 32);
      const signed char* pb = (const signed char*)(base + (col + 32 * g) * 32 + 16);
      for (int k = 0; k < 16; ++k) acc += (int)pa[k] * (int)pb[k];
    }
    d[r] += acc;
  }
  hipemu::rendezvous(blk->wbar[2 * w + 1], blk->wave_active[w]);
  return d;
}
// v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 (E2M1) operands (cbsz = blgp = 4): D = (A * 2^(scale_a - 127)) * (B * 2^(scale_b - 127)) + C on a
// 32 x 32 tile with K = 64.  Lane l holds 32 four-bit values (16 bytes, the first four dwords of the operand) of row (A) /
// column (B) l & 31, the K half chosen by l >> 5; the scale is byte 0 of the lane's scale operand (an E8M0 exponent, one
// per 32-value block).  D as for the other 32 x 32 shapes.  Which k a nibble stands for inside its half is the same for A
// and B, hence irrelevant for the sum.
typedef int emu_v8i __attribute__((vector_size(32)));
typedef float emu_v16f __attribute__((vector_size(64)));
inline emu_v16f __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(emu_v8i a, emu_v8i b, emu_v16f c, int cbsz, int blgp, int, int scale_a, int, int scale_b) {
  static const float kE2M1[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
  (void)cbsz; (void)blgp;  // FP4 only
  hipemu::Block* blk = hipemu::cur_block();
  const int w = hipemu::wave_id(), lane = hipemu::lane_id();
  unsigned char* base = &blk->wide[(size_t)w * 64 * 32];
  std::memcpy(base + lane * 32, &a, 16);
  std::memcpy(base + lane * 32 + 16, &b, 16);
  hipemu::rendezvous(blk->wbar[2 * w], blk->wave_active[w]);
  const double sc = std::ldexp(1.0, (scale_a & 0xff) - 127) * std::ldexp(1.0, (scale_b & 0xff) - 127);
  emu_v16f d = c;
  const int col = lane & 31;
  auto val = [&](const unsigned char* p, int j) {
    const int code = (p[j >> 1] >> (4 * (j & 1))) & 0xf;
    return (code & 8) ? -kE2M1[code & 7] : kE2M1[code & 7];
  };
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    double acc = 0.0;
    for (int g = 0; g < 2; ++g) {
      const unsigned char* pa = base + (row + 32 * g) * 32;
      const unsigned char* pb = base + (col + 32 * g) * 32 + 16;
      for (int k = 0; k < 32; ++k) acc += (double)val(pa, k) * (double)val(pb, k);
    }
    d[r] = (float)((double)d[r] + acc * sc);  // exact here: small integers times powers of two
  }
  hipemu::rendezvous(blk->wbar[2 * w + 1], blk->wave_active[w]);
  return d;
}
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }  // wave-uniform by construction where it is used
inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xffffffu) * (b & 0xffffffu); }
inline int __mul24(int a, int b) {  // signed 24-bit operands
  const int x = (int)((unsigned)a << 8) >> 8, y = (int)((unsigned)b << 8) >> 8;
  return (int)((unsigned)x * (unsigned)y);
}
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
inline int __float2int_rn(float v) { return (int)lrintf(v); }
inline float __fdiv_rn(float a, float b) { return a / b; }
inline float __fmul_rn(float a, float b) { return a * b; }
inline float __fadd_rn(float a, float b) { return a + b; }
inline float __fsub_rn(float a, float b) { return a - b; }
inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
inline int __float_as_int(float f) { int u; std::memcpy(&u, &f, 4); return u; }
inline float __int_as_float(int u) { float f; std::memcpy(&f, &u, 4); return f; }

template <class T> inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
template <class T> inline T atomicOr(T* p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
template <class T> inline T atomicAnd(T* p, T v) { return __atomic_fetch_and(p, v, __ATOMIC_RELAXED); }
template <class T> inline T atomicExch(T* p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
template <class T> inline T atomicCAS(T* p, T cmp, T v) {
  __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
  return cmp;
}
template <class T> inline T atomicMax(T* p, T v) {
  T old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
template <class T> inline T atomicMin(T* p, T v) {
  T old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old > v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline void __threadfence_block() {}

// ------------------------------------------------------------------ host runtime
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNoDevice = 100 };
typedef struct emu_stream* hipStream_t;
typedef struct emu_event { std::chrono::steady_clock::time_point t; }* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost,
                     hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipHostMallocDefault = 0, hipStreamNonBlocking = 1 };

inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipError(emu)"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
template <class T> inline hipError_t hipMalloc(T** p, size_t n) {
  void* q = nullptr;
  if (posix_memalign(&q, 256, n ? n : 1)) return hipErrorOutOfMemory;
  *p = (T*)q;
  return hipSuccess;
}
template <class T> inline hipError_t hipHostMalloc(T** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t = nullptr) {
  for (size_t y = 0; y < h; ++y) std::memcpy((char*)d + y * dp, (const char*)s + y * sp, w);
  return hipSuccess;
}
inline hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new emu_event; return hipSuccess; }
enum { hipEventDisableTiming = 2 };
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new emu_event; return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}

template <class K, class... Args>
inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t /*shmem*/, hipStream_t /*stream*/,
                               Args... args) {
  hipemu::launch(grid, block, [=]() { kernel(args...); });
}

#ifdef HIP_EMU_IMPLEMENTATION
// ThreadSanitizer does not follow swapcontext by itself: every switch is announced (tests/test_shim_threads.py builds the
// emulator with -fsanitize=thread to check the library's host code under the reference's concurrent callers)
#if defined(__SANITIZE_THREAD__)
extern "C" {
void* __tsan_get_current_fiber(void);
void* __tsan_create_fiber(unsigned flags);
void __tsan_destroy_fiber(void* fiber);
void __tsan_switch_to_fiber(void* fiber, unsigned flags);
}
#define HIPEMU_TSAN_SWITCH(f) __tsan_switch_to_fiber((f), 0)
#else
#define HIPEMU_TSAN_SWITCH(f) ((void)0)
#endif
thread_local emu_uint3 threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;
namespace hipemu {
thread_local Block* g_blk = nullptr;

#ifdef HIPEMU_FAST_SWITCH
asm(R"(
  .text
  .globl hipemu_switch
  .type hipemu_switch, @function
hipemu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq (%rsi), %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
  .size hipemu_switch, .-hipemu_switch
)");
static inline void to_scheduler(Block* b, Fiber* f) { hipemu_switch(&f->sp, &b->sched_sp); }
static inline void to_fiber(Block* b, Fiber* f) { hipemu_switch(&b->sched_sp, &f->sp); }
#else
static inline void to_scheduler(Block* b, Fiber* f) { swapcontext(&f->ctx, &b->sched); }
static inline void to_fiber(Block* b, Fiber* f) { swapcontext(&b->sched, &f->ctx); }
#endif

void yield_to_scheduler() {
  Block* b = cur_block();
  Fiber* f = b->cur;
  HIPEMU_TSAN_SWITCH(b->tsan_sched);
  to_scheduler(b, f);
}

static void fiber_entry() {
  Block* b = cur_block();
  Fiber* f = b->cur;
  b->body();
  f->done = true;
  // leaving work-items stop participating in barriers / collectives
  --b->active;
  const int w = f->tid >> 6;
  --b->wave_active[w];
  auto release = [&](Rendezvous& r, int act) {
    if (act > 0 && r.arrived >= act) { r.arrived = 0; ++r.gen; ++b->phase; }
  };
  release(b->bar, b->active);
  release(b->wbar[2 * w], b->wave_active[w]);
  release(b->wbar[2 * w + 1], b->wave_active[w]);
  HIPEMU_TSAN_SWITCH(b->tsan_sched);
  to_scheduler(b, f);
  __builtin_trap();   // a finished fiber is never resumed
}

static int order_mode() {
  static const int m = [] {   // initialised once, thread-safely: kernels of several host threads get here concurrently
    const char* e = getenv("RGBL_EMU_ORDER");
    return !e ? 0 : !strcmp(e, "asc") ? 1 : !strcmp(e, "desc") ? 2 : !strcmp(e, "shuffle") ? 3 : 0;
  }();
  return m;
}

void run_block(Block& b, const dim3& block) {
  const int T = (int)(block.x * block.y * block.z);
  const size_t STACK = 96 * 1024;
  if ((int)b.fibers.size() < T) {
    const size_t old = b.fibers.size();
    b.fibers.resize(T);
    for (size_t i = old; i < b.fibers.size(); ++i) b.fibers[i].stack = (char*)malloc(STACK);
  }
  b.nthreads = T;
  b.active = T;
  const int nw = (T + 63) / 64;
  b.wave_active.assign(nw, 0);
  b.wbar.assign(2 * nw, Rendezvous());
  b.bar = Rendezvous();
  b.slot.assign((size_t)nw * 64, 0);
  b.wide.assign((size_t)nw * 64 * 32, 0);
  b.phase = 0;
  for (int t = 0; t < T; ++t) {
    Fiber& f = b.fibers[t];
    f.blk = &b;
    f.tid = t;
    f.done = false;
    f.tidx.x = t % block.x;
    f.tidx.y = (t / block.x) % block.y;
    f.tidx.z = t / (block.x * block.y);
    ++b.wave_active[t >> 6];
#ifdef HIPEMU_FAST_SWITCH
    {
      // what hipemu_switch pops on the first switch: six registers, then `ret` into fiber_entry with the stack as after a call
      void** top = reinterpret_cast<void**>((reinterpret_cast<uintptr_t>(f.stack) + STACK) & ~(uintptr_t)15);
      top[-1] = nullptr;                                  // fiber_entry's "return address" (it never returns)
      top[-2] = reinterpret_cast<void*>(&fiber_entry);
      for (int r = 3; r <= 8; ++r) top[-r] = nullptr;     // rbp, rbx, r12 - r15
      f.sp = top - 8;
    }
#else
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack;
    f.ctx.uc_stack.ss_size = STACK;
    f.ctx.uc_link = nullptr;
    makecontext(&f.ctx, (void (*)())fiber_entry, 0);
#endif
#if defined(__SANITIZE_THREAD__)
    if (f.tsan) __tsan_destroy_fiber(f.tsan);   // a fresh context is a fresh fiber for the sanitizer
    f.tsan = __tsan_create_fiber(0);
#endif
  }
#if defined(__SANITIZE_THREAD__)
  b.tsan_sched = __tsan_get_current_fiber();
#endif
  cur_block() = &b;
  std::vector<int> order(T);
  unsigned rng = 12345u + blockIdx.x * 977u;
  while (b.active > 0) {
    const int mode = order_mode();
    const bool desc = mode == 2 || (mode == 0 && (b.phase & 1));
    for (int i = 0; i < T; ++i) order[i] = desc ? T - 1 - i : i;
    if (mode == 3)
      for (int i = T - 1; i > 0; --i) { rng = rng * 1664525u + 1013904223u; std::swap(order[i], order[(rng >> 8) % (i + 1)]); }
    const unsigned phase0 = b.phase;
    for (int i = 0; i < T; ++i) {
      Fiber& f = b.fibers[order[i]];
      if (f.done) continue;
      b.cur = &f;
      threadIdx = f.tidx;
      HIPEMU_TSAN_SWITCH(f.tsan);
      to_fiber(&b, &f);
      if (mode == 0 && b.phase != phase0) break;  // a rendezvous completed: flip the order
    }
  }
  cur_block() = nullptr;
}

struct Pool {
  std::mutex launch_mu;  // one launch at a time
  std::mutex mu;
  std::condition_variable cv_start, cv_done;
  std::vector<std::thread> threads;
  unsigned long long job_id = 0;
  int running = 0;
  // current job
  dim3 grid, block;
  const std::function<void()>* body = nullptr;
  std::atomic<size_t> next{0};
  size_t nblocks = 0;

  void work() {
    static thread_local Block blk;  // fiber stacks are reused across blocks and launches
    blk.body = *body;
    for (;;) {
      const size_t id = next.fetch_add(1);
      if (id >= nblocks) break;
      blockIdx.x = (unsigned)(id % grid.x);
      blockIdx.y = (unsigned)((id / grid.x) % grid.y);
      blockIdx.z = (unsigned)(id / ((size_t)grid.x * grid.y));
      blockDim = block;
      gridDim = grid;
      run_block(blk, block);
    }
  }
  void thread_main() {
    unsigned long long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_start.wait(lk, [&] { return job_id != seen; });
        seen = job_id;
      }
      work();
      {
        std::lock_guard<std::mutex> lk(mu);
        if (--running == 0) cv_done.notify_all();
      }
    }
  }
};

void launch(const dim3& grid, const dim3& block, const std::function<void()>& body) {
  const size_t nblocks = (size_t)grid.x * grid.y * grid.z;
  if (nblocks == 0) return;
  static Pool* pool = new Pool;  // leaked on purpose: worker threads live until process exit
  static int nthreads = [] {
    const char* e = getenv("RGBL_EMU_THREADS");
    int n = e ? atoi(e) : (int)std::thread::hardware_concurrency();
    return n < 1 ? 1 : n;
  }();
  std::lock_guard<std::mutex> guard(pool->launch_mu);
  if (pool->threads.empty())
    for (int i = 0; i < nthreads; ++i) {
      pool->threads.emplace_back([=] { pool->thread_main(); });
      pool->threads.back().detach();
    }
  {
    std::lock_guard<std::mutex> lk(pool->mu);
    pool->grid = grid;
    pool->block = block;
    pool->body = &body;
    pool->nblocks = nblocks;
    pool->next = 0;
    pool->running = nthreads;
    ++pool->job_id;
  }
  pool->cv_start.notify_all();
  std::unique_lock<std::mutex> lk(pool->mu);
  pool->cv_done.wait(lk, [&] { return pool->running == 0; });
}
}  // namespace hipemu
#endif  // HIP_EMU_IMPLEMENTATION
