// ref_depth_glue.cpp — C entry points around the reference's OWN ORB_SLAM3::DepthModule, compiled from
// /root/reference/src/DepthModule.cc (unmodified, where it lies) against oracle/cvcompat.  TEST INFRASTRUCTURE.
#include <string.h>

#include <vector>

#include "DepthModule.h"  // /root/reference/include

namespace {
struct Probe : ORB_SLAM3::DepthModule {  // the parse flags are protected members
  Probe(const std::string& path, int sensor) : ORB_SLAM3::DepthModule(path, sensor) {}
  bool lidar_ok() const { return b_parse_LiDAR; }
  bool upsampling_ok() const { return b_parse_LiDARUpsampling; }
};
}  // namespace

extern "C" {

void* ref_depth_create(const char* yaml_path, int sensor, int* parsed_lidar, int* parsed_upsampling, float proj[12]) {
  Probe* p = new Probe(yaml_path, sensor);
  *parsed_lidar = p->lidar_ok();
  *parsed_upsampling = p->upsampling_ok();
  const cv::Mat& M = p->LidarProjectionMatrix;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) proj[4 * r + c] = (M.rows >= 3 && M.cols == 4) ? M.at<float>(r, c) : 0.f;
  return p;
}

void ref_depth_destroy(void* h) { delete static_cast<Probe*>(h); }

// cloud: 4 x n row-major with leading dimension ld (rows x, y, z, 1) == the cv::Mat the reference's caller builds
int ref_depth_compute(void* h, const float* cloud, int n, int ld, int w, int height, const float* kp_xy, const float* kpun_x,
                      int k, float* out_depth, float* out_uright, float* out_raw, float* out_processed) {
  Probe* p = static_cast<Probe*>(h);
  cv::Mat pc(4, n, CV_32F);
  for (int r = 0; r < 4; ++r) memcpy(pc.ptr<float>(r), cloud + (size_t)r * ld, sizeof(float) * n);
  std::vector<cv::KeyPoint> keys(k), keys_un(k);
  for (int i = 0; i < k; ++i) {
    keys[i].pt.x = kp_xy[2 * i];
    keys[i].pt.y = kp_xy[2 * i + 1];
    keys_un[i].pt.x = kpun_x[i];
    keys_un[i].pt.y = kp_xy[2 * i + 1];
  }
  p->mvDepth.clear();
  p->mvuRight.clear();
  p->CalculateDepthFromPcd(keys, keys_un, pc, w, height);
  if ((int)p->mvDepth.size() != k || (int)p->mvuRight.size() != k) return -1;  // module disabled / method None
  memcpy(out_depth, p->mvDepth.data(), sizeof(float) * k);
  memcpy(out_uright, p->mvuRight.data(), sizeof(float) * k);
  auto dump = [&](const cv::Mat& m, float* dst) {
    if (!dst) return 0;
    if (m.rows != height || m.cols != w || m.type() != CV_32F) return 1;
    for (int y = 0; y < height; ++y) memcpy(dst + (size_t)y * w, m.ptr<float>(y), sizeof(float) * w);
    return 0;
  };
  int missing = dump(p->RawDepthMap, out_raw);
  missing |= dump(p->ProcessedDepthMap, out_processed) << 1;
  return missing;  // bit 1 set: ProcessedDepthMap not produced (NearestNeighborPixel never writes it)
}

}  // extern "C"
