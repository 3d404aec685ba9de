// Which XCDs / CUs does a stream created with hipExtStreamCreateWithCUMask reach on gfx950 (8 XCDs x 32 CUs)?
// For a few masks: launch 4096 one-wave workgroups that spin ~20 us each, record HW_REG_XCC_ID and HW_REG_HW_ID, print the
// histogram of workgroups per XCD, the number of distinct (XCD, SE, CU) places and which XCD workgroup b landed on (b % 8 rule?).
// hipcc -O3 --offload-arch=gfx950 -o cumask_probe cumask_probe.hip && ./cumask_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <set>
#include <vector>

__global__ void k_where(uint32_t* out, int spin) {
  // s_getreg_b32: simm16 = id | offset << 6 | (size - 1) << 11 ; HW_REG_HW_ID = 4, HW_REG_XCC_ID = 20
  const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));
  const uint32_t hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}

static void run(const char* what, const std::vector<uint32_t>& mask) {
  hipStream_t st;
  if (mask.empty()) hipStreamCreate(&st);
  else if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: create failed\n", what); return; }
  const int n = 4096;
  uint32_t* d;
  hipMalloc(&d, n * 8);
  hipLaunchKernelGGL(k_where, dim3(n), dim3(64), 0, st, d, 2000);  // 100 MHz wall clock: 20 us
  hipStreamSynchronize(st);
  std::vector<uint32_t> h(2 * n);
  hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
  int per_xcd[16] = {0}, rule = 0;
  std::set<uint32_t> places;
  for (int b = 0; b < n; ++b) {
    const uint32_t x = h[2 * b] & 15u, hw = h[2 * b + 1];
    per_xcd[x]++;
    places.insert((x << 16) | (hw & 0xff00u) | ((hw >> 13) & 7u) << 4);  // CU_ID [11:8], SH [12], SE_ID [15:13]
    rule += (int)x == b % 8;
  }
  printf("%-28s wgs/xcd:", what);
  for (int x = 0; x < 8; ++x) printf(" %4d", per_xcd[x]);
  printf("   places %3zu   wg b on xcd b%%8: %d/%d   first 16 xcds:", places.size(), rule, n);
  for (int b = 0; b < 16; ++b) printf(" %u", h[2 * b] & 15u);
  printf("\n");
  hipFree(d);
  hipStreamDestroy(st);
}

int main() {
  run("no mask", {});
  // hypothesis A: bit i -> XCD i % 8 (round-robin over the XCDs), CU i / 8 inside it
  for (uint32_t xs : {0x01u, 0x03u, 0x0fu, 0x3fu, 0xc0u, 0xf0u}) {
    std::vector<uint32_t> m(8, 0u);
    for (int i = 0; i < 256; ++i) if ((xs >> (i % 8)) & 1u) m[i / 32] |= 1u << (i % 32);
    char name[64]; snprintf(name, sizeof name, "A: bits i%%8 in 0x%02x", xs);
    run(name, m);
  }
  // hypothesis B: bits 32 x .. 32 x + 31 = XCD x
  for (uint32_t xs : {0x01u, 0x0fu}) {
    std::vector<uint32_t> m(8, 0u);
    for (int x = 0; x < 8; ++x) if ((xs >> x) & 1u) m[x] = 0xffffffffu;
    char name[64]; snprintf(name, sizeof name, "B: words in 0x%02x", xs);
    run(name, m);
  }
  // half of the CUs of every XCD (hypothesis A numbering: CU index i / 8 < 16)
  { std::vector<uint32_t> m(8, 0u); for (int i = 0; i < 128; ++i) m[i / 32] |= 1u << (i % 32); run("first 128 bits", m); }
  return 0;
}
