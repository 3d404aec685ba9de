// hbm_calib.hip — known-byte-count kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
// (MI355X_MICROARCH.md: FETCH_SIZE reports half the bytes of a wide coalesced streaming read and is uncalibrated for
// other access widths).  Every kernel moves exactly kBytes of a buffer far larger than the 256 MB infinity cache,
// 4 or 16 bytes per lane, coalesced.  Build + run on the GPU box: tools/gpu_traffic.sh.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr size_t kBytes = (size_t)1 << 30;

__global__ void calib_read_b32(const uint32_t* __restrict__ p, size_t n, uint32_t* __restrict__ sink) {
  uint32_t acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void calib_read_b128(const uint4* __restrict__ p, size_t n, uint32_t* __restrict__ sink) {
  uint32_t acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = p[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void calib_write_b32(uint32_t* __restrict__ p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}
__global__ void calib_write_b128(uint4* __restrict__ p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}

int main() {
  void *a = nullptr, *sink = nullptr;
  if (hipMalloc(&a, kBytes) != hipSuccess || hipMalloc(&sink, 256) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(a, 1, kBytes);
  hipDeviceSynchronize();
  const dim3 grid(256 * 16), block(256);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(calib_read_b32, grid, block, 0, 0, (const uint32_t*)a, kBytes / 4, (uint32_t*)sink);
    hipLaunchKernelGGL(calib_read_b128, grid, block, 0, 0, (const uint4*)a, kBytes / 16, (uint32_t*)sink);
    hipLaunchKernelGGL(calib_write_b32, grid, block, 0, 0, (uint32_t*)a, kBytes / 4);
    hipLaunchKernelGGL(calib_write_b128, grid, block, 0, 0, (uint4*)a, kBytes / 16);
  }
  hipDeviceSynchronize();
  printf("calib bytes per kernel: %zu\n", kBytes);
  return 0;
}
