// ref_glue.cpp — C entry point around the reference's OWN ORB_SLAM3::ORBextractor, compiled from
// /root/reference/src/ORBextractor.cc (unmodified, where it lies) against oracle/cvcompat.  TEST INFRASTRUCTURE.
#include <vector>

#include "ORBextractor.h"  // /root/reference/include

extern "C" int ref_extract(const uint8_t* img, int w, int h, int stride, int nfeatures, float scale_factor, int nlevels,
                           int ini_th, int min_th, int lap0, int lap1, orc_keypoint* kps, uint8_t* desc, int cap, int* n_out) {
  ORB_SLAM3::ORBextractor ex(nfeatures, scale_factor, nlevels, ini_th, min_th);
  cv::Mat image(h, w, CV_8UC1);
  for (int y = 0; y < h; ++y) memcpy(image.ptr(y), img + (size_t)y * stride, w);
  std::vector<cv::KeyPoint> keys;
  cv::Mat descriptors, mask;
  std::vector<int> lapping = {lap0, lap1};
  const int mono = ex(image, mask, keys, descriptors, lapping);
  const int n = (int)keys.size();
  *n_out = n;
  for (int i = 0; i < n && i < cap; ++i) {
    kps[i].x = keys[i].pt.x; kps[i].y = keys[i].pt.y; kps[i].size = keys[i].size; kps[i].angle = keys[i].angle;
    kps[i].response = keys[i].response; kps[i].octave = keys[i].octave; kps[i].class_id = keys[i].class_id;
    memcpy(desc + (size_t)i * 32, descriptors.ptr(i), 32);
  }
  return mono;
}
