// Drop-in replacement of /root/reference/include/ORBextractor.h: same class name, constructor, operator(),
// getters and public mvImagePyramid — the members System/Tracking/Frame touch — running on librgbl_frontend.so.
#ifndef ORBEXTRACTOR_H
#define ORBEXTRACTOR_H

#include <vector>

#include "cv_compat.h"

struct rgbl_extractor;
struct rgbl_device_frame;
struct rgbl_depth;

namespace ORB_SLAM3 {

class ORBextractor {
 public:
  enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

  ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
  ~ORBextractor();
  ORBextractor(const ORBextractor&) = delete;
  ORBextractor& operator=(const ORBextractor&) = delete;

  // Compute the ORB features and descriptors on an image (mask is ignored, as in the reference).
  // Returns monoIndex; -1 for an empty image.
  int operator()(cv::InputArray _image, cv::InputArray _mask, std::vector<cv::KeyPoint>& _keypoints,
                 cv::OutputArray _descriptors, std::vector<int>& vLappingArea);

  // Optional latency hook (rgbl_extract_begin): upload and extraction of `image` are queued and not waited for; the operator()
  // call that follows on the SAME cv::Mat data collects the results.  What is issued in between - above all
  // DepthModule::PrefetchPointcloud - runs next to the extraction.  Returns false when nothing was begun.
  bool Begin(cv::InputArray _image, std::vector<int>& vLappingArea);
  // Drops a begun frame that will not reach operator() (rgbl_extract_cancel): call before its cv::Mat is released.
  void CancelBegin();

  // cvtColor + operator() in one device round trip (Tracking::GrabImageRGBL, Tracking.cc:1567-1580): `data` is an
  // 8-bit image with 3 or 4 interleaved channels (1 = already gray), bRGB = Tracking::mbRGB.  imGray receives mImGray.
  int ExtractColor(const unsigned char* data, int channels, int step, int width, int height, bool bRGB, cv::Mat& imGray,
                   std::vector<cv::KeyPoint>& _keypoints, cv::Mat& _descriptors, std::vector<int>& vLappingArea);

  // Frame::UndistortKeyPoints (Frame.cc:837-870) on the device the keypoints came from: mvKeysUn = mvKeys with
  // cv::undistortPoints(pt, K, mDistCoef, noArray(), K) applied; a plain copy when mDistCoef[0] == 0, as in the reference.
  // K: 3x3 CV_32F (Pinhole::toK()), mDistCoef: 4x1 or 5x1 CV_32F.  Call after operator() (the handle exists by then).
  void UndistortKeyPoints(const std::vector<cv::KeyPoint>& mvKeys, const cv::Mat& K, const cv::Mat& mDistCoef,
                          std::vector<cv::KeyPoint>& mvKeysUn);

  // The frame this extractor has just produced, resident on the device for the matchers (rgbl_device_frame; INTEGRATION.md
  // "Frames resident on the device"): n = mvKeys.size() of the operator() call just made; depth = the DepthModule's handle
  // after CalculateDepthFromPcd (mvuRight), or NULL; K / mDistCoef as for UndistortKeyPoints (empty: mvKeysUn = mvKeys).
  // Device to device - nothing crosses PCIe.  `frame` is created on first use and reused (capacity grows as needed);
  // the owner (Frame / KeyFrame) destroys it with rgbl_device_frame_destroy.  Returns false (and says why) on failure.
  // grid6 (optional) = Frame::mnMinX, mnMinY, mnMaxX, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv: the frame then keeps its
  // AssignFeaturesToGrid as well and the projection searches skip theirs.
  bool CaptureDeviceFrame(rgbl_device_frame*& frame, int n, rgbl_depth* depth = nullptr, const cv::Mat& K = cv::Mat(),
                          const cv::Mat& mDistCoef = cv::Mat(), const float* grid6 = nullptr);

  int inline GetLevels() { return nlevels; }
  float inline GetScaleFactor() { return scaleFactor; }
  std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
  std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
  std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
  std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

  // Filled (device -> host, with the 19-px reflect-101 frame) only when keepPyramid is set: its single consumer
  // is Frame::ComputeStereoMatches (Frame.cc:908,998-1013); RGB-L / mono never read it.
  std::vector<cv::Mat> mvImagePyramid;
  bool keepPyramid = false;
  int device = 0;  // HIP device ordinal, may be changed before the first call
  // The C-ABI handle (pyramids of the last image stay resident in it): Frame::ComputeStereoMatches hands the left
  // and the right handle to rgbl_stereo_matches() instead of reading mvImagePyramid on the host.
  rgbl_extractor* Handle() const { return mpHandle; }

 protected:
  void EnsureHandle(int width, int height);
  void FillPyramid();

  int nfeatures;
  double scaleFactor;
  int nlevels;
  int iniThFAST;
  int minThFAST;
  std::vector<int> mnFeaturesPerLevel;
  std::vector<int> umax;
  std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;

  rgbl_extractor* mpHandle = nullptr;
  int mHandleW = 0, mHandleH = 0;
  std::vector<cv::Mat> mvPyramidStorage;
};

}  // namespace ORB_SLAM3
#endif
