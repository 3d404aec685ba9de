// ref_frame_glue.cpp — C entry point around the reference's OWN Frame::ComputeStereoMatches.  TEST INFRASTRUCTURE
// (oracle/Makefile target `ref`).  The function's lines (/root/reference/src/Frame.cc:901-1071) are extracted, unmodified,
// at build time into _ref/frame_stereo_matches.inc and compiled here as a member of the stand-in Frame of
// cvcompat/orbslam_types.h (force-included); the extractors whose mvImagePyramid it reads are the reference's own
// ORBextractor (src/ORBextractor.cc), its thresholds and DescriptorDistance the reference's own ORBmatcher (src/ORBmatcher.cc).
#include <limits.h>
#include <math.h>
#include <string.h>

#include <algorithm>
#include <chrono>

#include "ORBmatcher.h"  // /root/reference/include
#include "oracle.h"

// wall time of the last Frame::ComputeStereoMatches() call itself (bench.py: the CPU side of the stereo matcher)
static double g_stereo_call_seconds = 0.0;
extern "C" double ref_frame_last_call_seconds() { return g_stereo_call_seconds; }

namespace cv {
enum { NORM_L1 = 2 };
// cv::norm(a, b, NORM_L1) of two CV_8U views: the integer sum of absolute differences, returned as double
inline double norm(const Mat& a, const Mat& b, int type) {
  assert(type == NORM_L1 && a.rows == b.rows && a.cols == b.cols);
  long sum = 0;
  for (int y = 0; y < a.rows; ++y)
    for (int x = 0; x < a.cols; ++x) sum += std::abs((int)a.at<uchar>(y, x) - (int)b.at<uchar>(y, x));
  return (double)sum;
}
}  // namespace cv

namespace ORB_SLAM3 {
#include "_ref/frame_stereo_matches.inc"
}  // namespace ORB_SLAM3

static void copy_keys(const std::vector<cv::KeyPoint>& keys, const cv::Mat& desc, orc_keypoint* kps, uint8_t* out_desc, int cap) {
  for (int i = 0; i < (int)keys.size() && i < cap; ++i) {
    kps[i].x = keys[i].pt.x; kps[i].y = keys[i].pt.y; kps[i].size = keys[i].size; kps[i].angle = keys[i].angle;
    kps[i].response = keys[i].response; kps[i].octave = keys[i].octave; kps[i].class_id = keys[i].class_id;
    memcpy(out_desc + (size_t)i * 32, desc.ptr(i), 32);
  }
}

// Left / right extraction with the reference's ORBextractor, then the reference's ComputeStereoMatches.
extern "C" int ref_stereo_matches(const uint8_t* left, const uint8_t* right, int w, int h, int stride, int nfeatures,
                                  float scale_factor, int nlevels, int ini_th, int min_th, float mb, float mbf,
                                  orc_keypoint* kp_left, uint8_t* desc_left, orc_keypoint* kp_right, uint8_t* desc_right,
                                  int cap, int* n_left, int* n_right, float* out_uright, float* out_depth) {
  ORB_SLAM3::ORBextractor exl(nfeatures, scale_factor, nlevels, ini_th, min_th), exr(nfeatures, scale_factor, nlevels, ini_th, min_th);
  cv::Mat il(h, w, CV_8UC1), ir(h, w, CV_8UC1), mask;
  for (int y = 0; y < h; ++y) { memcpy(il.ptr(y), left + (size_t)y * stride, w); memcpy(ir.ptr(y), right + (size_t)y * stride, w); }
  ORB_SLAM3::Frame F;
  std::vector<int> lapping = {0, 0};
  exl(il, mask, F.mvKeys, F.mDescriptors, lapping);
  exr(ir, mask, F.mvKeysRight, F.mDescriptorsRight, lapping);
  F.N = (int)F.mvKeys.size();
  F.mb = mb; F.mbf = mbf;
  F.mvScaleFactors = exl.GetScaleFactors();
  F.mvInvScaleFactors = exl.GetInverseScaleFactors();
  F.mpORBextractorLeft = &exl;
  F.mpORBextractorRight = &exr;
  *n_left = F.N; *n_right = (int)F.mvKeysRight.size();
  if (F.N > cap || *n_right > cap) return -1;
  {
    const auto t0 = std::chrono::steady_clock::now();
    F.ComputeStereoMatches();
    g_stereo_call_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  copy_keys(F.mvKeys, F.mDescriptors, kp_left, desc_left, cap);
  copy_keys(F.mvKeysRight, F.mDescriptorsRight, kp_right, desc_right, cap);
  for (int i = 0; i < F.N; ++i) { out_uright[i] = F.mvuRight[i]; out_depth[i] = F.mvDepth[i]; }
  return 0;
}
