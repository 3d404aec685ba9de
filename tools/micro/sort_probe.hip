// Cycles of the quad-tree's emulated std::sort (block_sort_restated, extractor_kernels.h: one wave per pending range, a round per
// recursion depth) on quad-tree-like keys.  Round 5 measured a level-synchronous variant with it (all ranges of a depth partitioned
// together by the whole workgroup: one packed prefix sum for both scan lists, five barriers per depth; same permutation as std::sort
// on the emulator's test set and here): 512 work-items 32.6 k against 38.5 k cycles for 350 keys, 1024: 36.7 k against 39.1 k,
// 256: 50.2 k against 46.1 k - a barrier + LDS round trip costs ~600 cycles either way; not kept (profiles/r05_sort_probe.txt).
// (count << 12 | x0) with few distinct counts, one workgroup alone on the GPU.  hipcc -O3 --offload-arch=gfx950 -I../../orb_slam3_rgbl_amd/csrc -o sort_probe sort_probe.hip
#include "extractor_kernels.h"
#include <stdio.h>
#include <vector>
#include <algorithm>
using namespace rgbl;

template <int BS>
__global__ __launch_bounds__(BS) void k_probe(const uint64_t* key, int n, unsigned long long* cycles, uint64_t* out) {
  __shared__ unsigned long long s_skey_pad[kSortLds + 8];
  __shared__ uint16_t s_seg_first[kSortLds], s_seg_last[kSortLds];
  __shared__ SortRanges s_ra, s_rb;
  __shared__ int s_sort_cnt[2];
  uint64_t* w = reinterpret_cast<uint64_t*>(s_skey_pad + 4);
  for (int i = threadIdx.x; i < n; i += BS) w[i] = (key[i] << 16) | (uint64_t)i;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  block_sort_restated<BS>(w, n, s_seg_first, s_seg_last, &s_ra, &s_rb, s_sort_cnt);
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cycles[0] = t1 - t0;
  for (int i = threadIdx.x; i < n; i += BS) out[i] = w[i];
}

int main() {
  uint64_t *d_key, *d_out; unsigned long long* d_c;
  hipMalloc(&d_key, 8 * 2048); hipMalloc(&d_out, 8 * 2048); hipMalloc(&d_c, 8);
  for (int n : {64, 128, 200, 350, 500}) {
    std::vector<uint64_t> k(n);
    uint32_t s = 12345u + n;
    for (int i = 0; i < n; ++i) {
      s = s * 1664525u + 1013904223u;
      const uint32_t r = s >> 8;
      const uint32_t count = 2 + (r % 16 < 8 ? 0 : r % 16 < 12 ? 1 : r % 16 < 14 ? 2 : (r >> 4) % 12);
      const uint32_t x0 = ((r >> 12) % 40) * 30;
      k[i] = ((uint64_t)count << 12) | x0;
    }
    hipMemcpy(d_key, k.data(), 8 * n, hipMemcpyHostToDevice);
    unsigned long long c[3] = {0, 0, 0};
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(k_probe<256>, dim3(1), dim3(256), 0, 0, d_key, n, d_c, d_out); hipMemcpy(&c[0], d_c, 8, hipMemcpyDeviceToHost);
      hipLaunchKernelGGL(k_probe<512>, dim3(1), dim3(512), 0, 0, d_key, n, d_c, d_out); hipMemcpy(&c[1], d_c, 8, hipMemcpyDeviceToHost);
      hipLaunchKernelGGL(k_probe<1024>, dim3(1), dim3(1024), 0, 0, d_key, n, d_c, d_out); hipMemcpy(&c[2], d_c, 8, hipMemcpyDeviceToHost);
    }
    std::vector<uint64_t> o(n);
    hipMemcpy(o.data(), d_out, 8 * n, hipMemcpyDeviceToHost);
    // sanity: keys ascending
    bool ok = true; for (int i = 1; i < n; ++i) ok = ok && (o[i - 1] >> 16) <= (o[i] >> 16);
    std::vector<uint64_t> d = k; std::sort(d.begin(), d.end()); d.erase(std::unique(d.begin(), d.end()), d.end());
    printf("n %4d distinct keys %4zu  cycles: 256 thr %6llu  512 thr %6llu  1024 thr %6llu  %s\n", n, d.size(), c[0], c[1], c[2], ok ? "sorted" : "NOT SORTED");
  }
  return 0;
}
