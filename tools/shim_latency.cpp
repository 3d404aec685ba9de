// Latency of ONE RGB-L frame through the C++ drop-in classes, the way Frame's RGB-L constructor + a tracking match use them:
// ORBextractor::operator(), DepthModule::CalculateDepthFromPcd, an all-pairs Hamming scan against the previous frame - with
// and without the optional hooks (ORBextractor::Begin + DepthModule::PrefetchPointcloud).  Host pointers, pageable memory,
// synchronous calls: what a C++ caller sees (tools/host_api_latency.py measures the same through the Python mirror).
//   shim_latency <settings.yaml> <frames.raw (n x h x w bytes)> <n> <w> <h> <cloud.raw (4 x N floats)> <N>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "DepthModule.h"
#include "ORBextractor.h"
#include "rgbl_frontend.h"

static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
  if (argc < 8) return 2;
  const int n = atoi(argv[3]), w = atoi(argv[4]), h = atoi(argv[5]), np = atoi(argv[7]);
  std::vector<cv::Mat> frames;
  FILE* f = fopen(argv[2], "rb");
  for (int i = 0; i < n; ++i) {
    cv::Mat im(h, w, CV_8UC1);
    if (!f || fread(im.data, 1, (size_t)w * h, f) != (size_t)w * h) return 3;
    frames.push_back(im);
  }
  fclose(f);
  cv::Mat pcd(4, np, CV_32F);
  f = fopen(argv[6], "rb");
  if (!f || fread(pcd.ptr<float>(), sizeof(float), (size_t)4 * np, f) != (size_t)4 * np) return 4;
  fclose(f);
  ORB_SLAM3::ORBextractor ex(2000, 1.2f, 8, 12, 7);
  ORB_SLAM3::DepthModule dm(argv[1], 6);
  dm.downloadDenseMaps = false;
  rgbl_matcher* mt = nullptr;
  if (rgbl_matcher_acquire(0, &mt) != RGBL_OK) return 5;
  std::vector<int> lap = {0, 0};
  std::vector<cv::KeyPoint> keys;
  cv::Mat desc, prev;
  std::vector<int32_t> bi(8192), bd(8192), sd(8192);
  for (int hooks = 0; hooks < 2; ++hooks) {
    double best = 1e9, parts[3] = {0, 0, 0};
    for (int rep = 0; rep < 6; ++rep) {
      double t[3] = {0, 0, 0};
      const double a0 = now_ms();
      for (int i = 0; i < n; ++i) {
        const double a = now_ms();
        if (hooks) { ex.Begin(frames[i], lap); dm.PrefetchPointcloud(pcd, w, h); }
        ex(frames[i], cv::Mat(), keys, desc, lap);
        const double b = now_ms();
        dm.CalculateDepthFromPcd(keys, keys, pcd, w, h);
        const double c = now_ms();
        if (prev.rows > 0)
          rgbl_hamming_bf(mt, prev.data, prev.rows, desc.data, desc.rows, bi.data(), bd.data(), sd.data());
        const double d = now_ms();
        prev = desc.clone();
        t[0] += b - a; t[1] += c - b; t[2] += d - c;
      }
      const double tot = (now_ms() - a0) / n;
      if (rep > 0 && tot < best) { best = tot; for (int k = 0; k < 3; ++k) parts[k] = t[k] / n; }
    }
    printf("%s: %.3f ms per frame (extract %.3f, depth %.3f, match %.3f) -> %.0f frames/s, %d keypoints\n",
           hooks ? "C++ drop-in classes with Begin + PrefetchPointcloud" : "C++ drop-in classes", best, parts[0], parts[1], parts[2],
           1e3 / best, (int)keys.size());
  }
  rgbl_matcher_release(mt);
  return 0;
}
