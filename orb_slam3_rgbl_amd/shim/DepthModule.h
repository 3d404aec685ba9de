// Drop-in replacement of /root/reference/include/DepthModule.h: same class name, constructor,
// CalculateDepthFromPcd and public result members (mvDepth, mvuRight, RawDepthMap, ProcessedDepthMap,
// LidarProjectionMatrix), running on librgbl_frontend.so.
#ifndef DEPTHMODULE_H
#define DEPTHMODULE_H

#include <map>
#include <string>
#include <vector>

#include "cv_compat.h"

struct rgbl_depth;

namespace ORB_SLAM3 {

class DepthModule {
 public:
  enum UpsamlingMethod { None = 0, NearestNeighborPixel = 1, AverageFiltering = 2, InverseDilation = 3, IPBasic = 5 };

  DepthModule(const std::string& strSettingPath, const int sensor);
  ~DepthModule();
  DepthModule(const DepthModule&) = delete;
  DepthModule& operator=(const DepthModule&) = delete;

  // Depth value handler, manages the depth calculation for each frame (DepthModule.cc:50-79).
  void CalculateDepthFromPcd(std::vector<cv::KeyPoint> mvKeys, std::vector<cv::KeyPoint> mvKeysUn,
                             const cv::Mat& PointCloud, const int imwidth, const int imheight);
  // Optional latency hook (rgbl_depth_prefetch): the part of CalculateDepthFromPcd that does not need the keypoints - upload of
  // the scan, projection, up-sampling - queued on the device and not waited for.  Called right after ORBextractor::Begin it runs
  // next to the extraction; CalculateDepthFromPcd on the SAME cv::Mat data then only gathers the keypoints' depths (with
  // downloadDenseMaps set the maps are computed again with the raw map).  The scan must stay unchanged in between.
  void PrefetchPointcloud(const cv::Mat& PointCloud, const int imwidth, const int imheight);
  // Forgets a prefetched scan that will not reach CalculateDepthFromPcd (rgbl_depth_prefetch_cancel): call before its cv::Mat is released.
  void CancelPrefetch();
  // Same, on the raw contents of a KITTI velodyne .bin file (nPoints x {x, y, z, reflectance}): replaces the example's
  // LoadPointcloudBinaryMat repack (Examples/RGB-L/rgbl_kitti.cc:151-185) + CalculateDepthFromPcd.
  void CalculateDepthFromKittiBin(const std::vector<cv::KeyPoint>& mvKeys, const std::vector<cv::KeyPoint>& mvKeysUn,
                                  const float* xyzi, const int nPoints, const int imwidth, const int imheight);

  cv::Mat LidarProjectionMatrix;  // 3x4, CV_32F
  cv::Mat RawDepthMap;
  cv::Mat ProcessedDepthMap;
  std::vector<float> mvuRight;
  std::vector<float> mvDepth;
  // The dense maps are only read by the viewer (Tracking.cc:1584 -> FrameDrawer.cc:374-376); switching this off
  // saves two device->host copies of h*w floats per frame and changes nothing else.
  bool downloadDenseMaps = true;
  int device = 0;

  // the C-ABI handle (mvuRight of the last CalculateDepthFromPcd stays resident in it: ORBextractor::CaptureDeviceFrame)
  rgbl_depth* Handle() const { return mpHandle; }

 protected:
  bool ParseRGBLParameters(const std::string& strSettingPath);
  bool ParseUpsamplingParameters(const std::string& strSettingPath);
  void EnsureHandle(int width, int height, int nPoints, int nKeys);
  void Compute(const std::vector<cv::KeyPoint>& mvKeys, const std::vector<cv::KeyPoint>& mvKeysUn, const float* cloud, int nPoints,
               int ld, bool xyzi, int imwidth, int imheight);

  bool b_parse_LiDAR, b_parse_LiDARUpsampling;
  float mbf;
  UpsamlingMethod SelectedUpsamlingMethod;
  float opt_min_dist, opt_max_dist;
  float ParamUpsampling_NearestNeighborPixel_SearchRadius = 0;
  int ParamUpsampling_AverageFilter_KernelSize = 0;
  bool ParamUpsampling_AverageFilter_bDoDilationPreprocessing = false;
  std::string ParamUpsampling_AverageFilter_DilationPreprocessing_KernelType;
  int ParamUpsampling_AverageFilter_DilationPreprocessing_KernelSize = 0;
  std::string ParamUpsampling_InverseDilation_KernelType;
  int ParamUpsampling_InverseDilation_KernelSize_u = 0, ParamUpsampling_InverseDilation_KernelSize_v = 0;

  rgbl_depth* mpHandle = nullptr;
  int mHandleW = 0, mHandleH = 0, mHandlePoints = 0, mHandleKeys = 0;
};

}  // namespace ORB_SLAM3
#endif
