// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 (E2M1) operands on gfx950: scale encoding, the K coverage of a lane's
// 32 nibbles, the C/D layout.  hipcc -O2 --offload-arch=gfx950 -o /tmp/fp4probe tools/micro/mfma_fp4_probe.hip && /tmp/fp4probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void k(const uint32_t* a, const uint32_t* b, float* c, int SA, int SB) {
  const int l = threadIdx.x;
  v8i A = {(int)a[l * 4], (int)a[l * 4 + 1], (int)a[l * 4 + 2], (int)a[l * 4 + 3], 0, 0, 0, 0};
  v8i B = {(int)b[l * 4], (int)b[l * 4 + 1], (int)b[l * 4 + 2], (int)b[l * 4 + 3], 0, 0, 0, 0};
  v16f C = {0};
  C = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, C, 4, 4, 0, SA, 0, SB);
  for (int r = 0; r < 16; ++r) c[l * 16 + r] = C[r];
}

int main() {
  uint32_t ha[256], hb[256];
  float hc[1024];
  uint32_t *da, *db; float* dc;
  hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dc, sizeof(hc));
  auto run = [&](int which) {
    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    const int sc[5] = {0, 127, 131, (int)0x83838383u, (int)0x7f7f7f7fu};
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc, sc[which], sc[which]);
    hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
  };
  // 1. all +4 x all +4: 64 products of 16 = 1024 (x scale_a x scale_b)
  for (int i = 0; i < 256; ++i) { ha[i] = 0x66666666u; hb[i] = 0x66666666u; }
  for (int w = 0; w < 5; ++w) { run(w); printf("scale variant %d: C[0] = %g, C[lane 37 reg 5] = %g\n", w, hc[0], hc[37 * 16 + 5]); }
  // 2. K coverage: flip ONE nibble of row 3 (lane 3 or lane 35) to -4: C[3][*] must drop by 32 wherever the nibble sits
  for (int g = 0; g < 2; ++g)
    for (int j = 0; j < 32; j += 7) {
      for (int i = 0; i < 256; ++i) { ha[i] = 0x66666666u; hb[i] = 0x66666666u; }
      const int lane = 3 + 32 * g;
      ha[lane * 4 + j / 8] ^= 0x8u << (4 * (j % 8));
      run(1);
      // row 3 of C: reg r with (r&3)+8*(r>>2)+4*(l>>5) == 3 -> r = 3, lanes 0..31 (cols)
      printf("flip lane-group %d nibble %2d: C[row 3][col 0] = %g, C[row 3][col 17] = %g, C[row 4][col 0] = %g\n", g, j, hc[0 * 16 + 3], hc[17 * 16 + 3], hc[32 * 16 + 0]);
    }
  // 3. C layout: row i of A has i nibbles of -4 (in lanes i, first group): C[i][j] = 16 * (64 - 2 i) under the i8 map
  for (int i = 0; i < 256; ++i) { ha[i] = 0x66666666u; hb[i] = 0x66666666u; }
  for (int row = 0; row < 32; ++row)
    for (int n = 0; n < row; ++n) ha[row * 4 + n / 8] ^= 0x8u << (4 * (n % 8));
  run(1);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
      if (hc[l * 16 + r] != 16.f * (64 - 2 * row)) ++bad;
    }
  printf("C layout check (i8 map): %d mismatches; C[lane 0] regs:", bad);
  for (int r = 0; r < 16; ++r) printf(" %g", hc[r]);
  printf("\n");
  // 4. asymmetric B: column j of B has j nibbles of -4
  for (int i = 0; i < 256; ++i) { ha[i] = 0x66666666u; hb[i] = 0x66666666u; }
  for (int col = 0; col < 32; ++col)
    for (int n = 0; n < col; ++n) hb[col * 4 + n / 8] ^= 0x8u << (4 * (n % 8));
  run(1);
  bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 16; ++r)
      if (hc[l * 16 + r] != 16.f * (64 - 2 * (l & 31))) ++bad;
  printf("B column check: %d mismatches\n", bad);
  return 0;
}
