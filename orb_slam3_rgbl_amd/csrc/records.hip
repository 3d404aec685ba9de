// records.hip — compaction of a batch's per-frame results into the variable-length records of the multi-GPU gather
// (one record per keypoint = 28 B cv::KeyPoint + 32 B descriptor + 4 B depth + 4 B uRight = 68 B, frames
// back to back, frame f at record offset sum of the counts before it).  What leaves a GPU over xGMI is then
// 68 * (number of keypoints) bytes plus the count vector, not capacity-padded arrays.
#include "common.h"

namespace rgbl {

constexpr int kRecordWords = 17;  // 68 bytes

// grid = (frames), block = 256.  Every workgroup sums the counts before its frame itself (a few hundred loads).
__global__ __launch_bounds__(256) void k_pack_records(const int32_t* __restrict__ n_rows, const uint32_t* __restrict__ kp,
                                                      const uint32_t* __restrict__ desc, const uint32_t* __restrict__ depth,
                                                      const uint32_t* __restrict__ uright, int cap, long long first_record,
                                                      long long capacity_records, uint32_t* __restrict__ out,
                                                      long long* __restrict__ offsets, int* __restrict__ overflow) {
  __shared__ long long s_part[4];
  __shared__ long long s_base;
  const int f = blockIdx.x, tid = threadIdx.x;
  long long before = 0;
  for (int i = tid; i < f; i += 256) before += imin(imax(n_rows[i], 0), cap);
  for (int m = 32; m >= 1; m >>= 1) before += __shfl_xor(before, m);
  if (lane_id() == 0) s_part[wave_id()] = before;
  __syncthreads();
  if (tid == 0) s_base = s_part[0] + s_part[1] + s_part[2] + s_part[3];
  __syncthreads();
  const long long base = first_record + s_base;
  const int n = imin(imax(n_rows[f], 0), cap);
  if (tid == 0) {
    offsets[f] = base;
    if ((int)blockIdx.x == (int)gridDim.x - 1) offsets[f + 1] = base + n;
  }
  if (base + n > capacity_records) {
    if (tid == 0) *overflow = 1;
    return;
  }
  const uint32_t* K = kp + (size_t)f * cap * 7;
  const uint32_t* D = desc + (size_t)f * cap * 8;
  const uint32_t* Z = depth + (size_t)f * cap;
  const uint32_t* U = uright + (size_t)f * cap;
  uint32_t* O = out + (size_t)base * kRecordWords;
  for (int i = tid; i < n * kRecordWords; i += 256) {
    const int r = i / kRecordWords, w = i - r * kRecordWords;
    O[i] = w < 7 ? K[r * 7 + w] : w < 15 ? D[r * 8 + (w - 7)] : w == 15 ? Z[r] : U[r];
  }
}

// an empty batch still leaves its one offset behind (the header promises d_offsets[batch])
__global__ void k_pack_records_empty(long long first_record, long long* __restrict__ offsets) { offsets[0] = first_record; }

}  // namespace rgbl

extern "C" int rgbl_pack_records_device(void* hip_stream, const int32_t* d_n, const rgbl_keypoint* d_kp, const uint8_t* d_desc,
                                        const float* d_depth, const float* d_uright, int batch, int cap, long long first_record,
                                        long long capacity_records, uint8_t* d_out, long long* d_offsets, int* d_overflow) {
  using namespace rgbl;
  if (!d_n || !d_kp || !d_desc || !d_depth || !d_uright || !d_out || !d_offsets || !d_overflow || batch < 0 || cap < 1 ||
      first_record < 0 || capacity_records < first_record) {
    set_error("invalid argument");
    return RGBL_ERR_INVALID;
  }
  if (batch == 0) {
    hipLaunchKernelGGL(k_pack_records_empty, dim3(1), dim3(1), 0, (hipStream_t)hip_stream, first_record, d_offsets);
    RGBL_HIP(hipGetLastError());
    return RGBL_OK;
  }
  hipLaunchKernelGGL(k_pack_records, dim3(batch), dim3(256), 0, (hipStream_t)hip_stream, d_n,
                     reinterpret_cast<const uint32_t*>(d_kp), reinterpret_cast<const uint32_t*>(d_desc),
                     reinterpret_cast<const uint32_t*>(d_depth), reinterpret_cast<const uint32_t*>(d_uright), cap, first_record,
                     capacity_records, reinterpret_cast<uint32_t*>(d_out), d_offsets, d_overflow);
  RGBL_HIP(hipGetLastError());
  return RGBL_OK;
}
