// Issue rate of a few VALU instructions on gfx950 (wave64): 8 independent chains per lane, 4096 iterations, 8 waves per SIMD.
// Prints wave-instructions per cycle per SIMD (1/4 = "full rate" for a 16-lane SIMD).
// hipcc -O3 --offload-arch=gfx950 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t dot2(uint32_t a, uint32_t b, uint32_t c) { us2 x, y; __builtin_memcpy(&x, &a, 4); __builtin_memcpy(&y, &b, 4); return __builtin_amdgcn_udot2(x, y, c, false); }
__device__ __forceinline__ uint32_t pkmin(uint32_t a, uint32_t b) { us2 x, y; __builtin_memcpy(&x, &a, 4); __builtin_memcpy(&y, &b, 4); x = __builtin_elementwise_min(x, y) + y; uint32_t r; __builtin_memcpy(&r, &x, 4); return r; }
#define KERNEL(name, ...)                                                                    \
  __global__ __launch_bounds__(256) void name(uint32_t* out, uint32_t seed, int iters) {      \
    uint32_t a[8], b = seed + threadIdx.x, c = seed * 3u + 1u;                                 \
    for (int k = 0; k < 8; ++k) a[k] = seed + k + threadIdx.x;                                 \
    for (int i = 0; i < iters; ++i) {                                                          \
      _Pragma("unroll") for (int k = 0; k < 8; ++k) { __VA_ARGS__; }                                  \
    }                                                                                          \
    uint32_t r = 0;                                                                            \
    for (int k = 0; k < 8; ++k) r ^= a[k];                                                     \
    out[blockIdx.x * 256 + threadIdx.x] = r;                                                   \
  }
KERNEL(k_dot4, a[k] = __builtin_amdgcn_udot4(a[k], b, c, false))
KERNEL(k_dot2, a[k] = dot2(a[k], b, c))
KERNEL(k_mad24, a[k] = __umul24(a[k], b) + c)
KERNEL(k_align, a[k] = __builtin_amdgcn_alignbyte(a[k], b, 1))
KERNEL(k_bcnt, a[k] = __builtin_popcount(a[k] ^ b) + a[k])
KERNEL(k_perm, a[k] = __builtin_amdgcn_perm(a[k], b, 0x06020400u))
KERNEL(k_mullo, a[k] = a[k] * b)
KERNEL(k_add, a[k] = a[k] + b)
KERNEL(k_min3, a[k] = min(min(a[k], b), c) + 1u)
KERNEL(k_pkmin, a[k] = pkmin(a[k], b))
KERNEL(k_sad, a[k] = __builtin_amdgcn_sad_u8(a[k], b, c))
template <class K>
void run(const char* name, K kern, int ops_per_body) {
  uint32_t* out;
  const int blocks = 256 * 8, iters = 4096;
  hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 12345u, 16);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 12345u, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  int clk_khz = 0;
  hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  const double wave_instr = (double)blocks * 4 * iters * 8 * ops_per_body;
  const double cycles = ms * 1e-3 * clk_khz * 1e3;
  printf("%-8s %8.3f ms  %.3f wave-instr / cycle / SIMD (clock %d MHz)\n", name, ms, wave_instr / cycles / (256 * 4), clk_khz / 1000);
  hipFree(out);
}
int main() {
  run("add", k_add, 1); run("mad24", k_mad24, 1); run("dot4", k_dot4, 1); run("dot2", k_dot2, 1); run("align", k_align, 1);
  run("bcnt", k_bcnt, 2); run("perm", k_perm, 1); run("mul_lo", k_mullo, 1); run("min3", k_min3, 2); run("pk_min", k_pkmin, 2);
  run("sad_u8", k_sad, 1);
  return 0;
}
