// Shim of ORB_SLAM3::ORBextractor over the C ABI (include/rgbl_frontend.h). Replaces
// /root/reference/src/ORBextractor.cc; all pixel work happens in the HIP kernels.
#include "ORBextractor.h"

#include <math.h>

#include <iostream>

#include "../../include/rgbl_frontend.h"

namespace ORB_SLAM3 {

static_assert(sizeof(cv::KeyPoint) == sizeof(rgbl_keypoint), "cv::KeyPoint must be layout compatible with rgbl_keypoint");

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST) {
  // The per-level tables are pure host arithmetic (ORBextractor.cc:414-430) and must exist before the first
  // image arrives: every Frame copies them (Frame.cc:110-116).  The handle recomputes the same values.
  mvScaleFactor.resize(nlevels);
  mvLevelSigma2.resize(nlevels);
  mvScaleFactor[0] = 1.0f;
  mvLevelSigma2[0] = 1.0f;
  for (int i = 1; i < nlevels; i++) {
    mvScaleFactor[i] = mvScaleFactor[i - 1] * _scaleFactor;
    mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i];
  }
  mvInvScaleFactor.resize(nlevels);
  mvInvLevelSigma2.resize(nlevels);
  for (int i = 0; i < nlevels; i++) {
    mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i];
    mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i];
  }
  mvImagePyramid.resize(nlevels);
}

ORBextractor::~ORBextractor() { rgbl_extractor_destroy(mpHandle); }

void ORBextractor::EnsureHandle(int width, int height) {
  if (mpHandle && width == mHandleW && height == mHandleH) return;
  rgbl_extractor_destroy(mpHandle);
  mpHandle = nullptr;
  rgbl_extractor_cfg cfg;
  cfg.nfeatures = nfeatures;
  cfg.scale_factor = (float)scaleFactor;
  cfg.nlevels = nlevels;
  cfg.ini_th_fast = iniThFAST;
  cfg.min_th_fast = minThFAST;
  cfg.width = width;
  cfg.height = height;
  cfg.max_batch = 1;
  if (rgbl_extractor_create(&cfg, device, &mpHandle) != RGBL_OK) {
    std::cerr << "[ORBextractor] " << rgbl_last_error() << std::endl;  // the reference reports on stdout/stderr, never throws
    mpHandle = nullptr;
    return;
  }
  mHandleW = width;
  mHandleH = height;
  mnFeaturesPerLevel.resize(nlevels);
  umax.resize(16);
  rgbl_extractor_tables(mpHandle, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(),
                        mvInvLevelSigma2.data(), mnFeaturesPerLevel.data(), umax.data());
}

int ORBextractor::operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint>& _keypoints,
                             cv::OutputArray _descriptors, std::vector<int>& vLappingArea) {
#ifdef RGBL_HAVE_OPENCV
  if (_image.empty()) return -1;
  cv::Mat image = _image.getMat();
#else
  if (_image.empty()) return -1;
  const cv::Mat& image = _image;
#endif
  if (image.type() != CV_8UC1) {
    std::cerr << "[ORBextractor] image must be CV_8UC1" << std::endl;
    return -1;
  }
  EnsureHandle(image.cols, image.rows);
  if (!mpHandle) return -1;
  const int cap = rgbl_extractor_max_keypoints(mpHandle);
  _keypoints = std::vector<cv::KeyPoint>(cap);
  cv::Mat desc(cap, 32, CV_8U);
  int n = 0, mono = -1;
  const int lap0 = vLappingArea.size() > 0 ? vLappingArea[0] : 0, lap1 = vLappingArea.size() > 1 ? vLappingArea[1] : 0;
  const int rc = rgbl_extract(mpHandle, image.data, image.cols, image.rows, (int)image.step, lap0, lap1,
                              reinterpret_cast<rgbl_keypoint*>(_keypoints.data()), desc.data, cap, &n, &mono);
  if (rc != RGBL_OK) {
    std::cerr << "[ORBextractor] " << rgbl_last_error() << std::endl;
    _keypoints.clear();
    return -1;
  }
  _keypoints.resize(n);
#ifdef RGBL_HAVE_OPENCV
  if (n == 0) _descriptors.release();
  else desc.rowRange(0, n).copyTo(_descriptors);
#else
  if (n == 0) _descriptors.release();
  else {
    _descriptors.create(n, 32, CV_8U);
    memcpy(_descriptors.data, desc.data, (size_t)n * 32);
  }
#endif
  FillPyramid();
  return mono;
}

bool ORBextractor::Begin(cv::InputArray _image, std::vector<int>& vLappingArea) {
#ifdef RGBL_HAVE_OPENCV
  if (_image.empty()) return false;
  cv::Mat image = _image.getMat();
#else
  if (_image.empty()) return false;
  const cv::Mat& image = _image;
#endif
  if (image.type() != CV_8UC1) return false;
  EnsureHandle(image.cols, image.rows);
  if (!mpHandle) return false;
  const int lap0 = vLappingArea.size() > 0 ? vLappingArea[0] : 0, lap1 = vLappingArea.size() > 1 ? vLappingArea[1] : 0;
  return rgbl_extract_begin(mpHandle, image.data, image.cols, image.rows, (int)image.step, lap0, lap1) == RGBL_OK;
}

void ORBextractor::CancelBegin() {
  if (mpHandle) (void)rgbl_extract_cancel(mpHandle);
}

int ORBextractor::ExtractColor(const unsigned char* data, int channels, int step, int width, int height, bool bRGB,
                               cv::Mat& imGray, std::vector<cv::KeyPoint>& _keypoints, cv::Mat& _descriptors,
                               std::vector<int>& vLappingArea) {
  if (!data || width <= 0 || height <= 0) return -1;
  EnsureHandle(width, height);
  if (!mpHandle) return -1;
  const int cap = rgbl_extractor_max_keypoints(mpHandle);
  _keypoints = std::vector<cv::KeyPoint>(cap);
  cv::Mat desc(cap, 32, CV_8U);
  imGray.create(height, width, CV_8UC1);
  int n = 0, mono = -1;
  const int lap0 = vLappingArea.size() > 0 ? vLappingArea[0] : 0, lap1 = vLappingArea.size() > 1 ? vLappingArea[1] : 0;
  const int rc = rgbl_extract_color(mpHandle, data, channels, bRGB ? 0 : 1, width, height, step, lap0, lap1,
                                    reinterpret_cast<rgbl_keypoint*>(_keypoints.data()), desc.data, cap, &n, &mono, imGray.data,
                                    (int)imGray.step);
  if (rc != RGBL_OK) {
    std::cerr << "[ORBextractor] " << rgbl_last_error() << std::endl;
    _keypoints.clear();
    return -1;
  }
  _keypoints.resize(n);
  if (n == 0) _descriptors.release();
  else {
    _descriptors.create(n, 32, CV_8U);
    memcpy(_descriptors.data, desc.data, (size_t)n * 32);
  }
  FillPyramid();
  return mono;
}

void ORBextractor::UndistortKeyPoints(const std::vector<cv::KeyPoint>& mvKeys, const cv::Mat& K, const cv::Mat& mDistCoef,
                                      std::vector<cv::KeyPoint>& mvKeysUn) {
  mvKeysUn = mvKeys;
  const int n = (int)mvKeys.size(), nd = (int)mDistCoef.total();
  if (n == 0 || nd < 1 || mDistCoef.at<float>(0) == 0.0) return;
  if (!mpHandle || (nd != 4 && nd != 5)) {
    std::cerr << "[ORBextractor] UndistortKeyPoints needs an extractor that has processed an image and 4 or 5 coefficients" << std::endl;
    return;
  }
  std::vector<float> xy(2 * (size_t)n), dist(nd);
  for (int i = 0; i < n; ++i) { xy[2 * i] = mvKeys[i].pt.x; xy[2 * i + 1] = mvKeys[i].pt.y; }
  for (int i = 0; i < nd; ++i) dist[i] = mDistCoef.at<float>(i);
  const float k[4] = {K.at<float>(0, 0), K.at<float>(1, 1), K.at<float>(0, 2), K.at<float>(1, 2)};
  if (rgbl_undistort_points(mpHandle, xy.data(), n, k, dist.data(), nd, xy.data()) != RGBL_OK) {
    std::cerr << "[ORBextractor] " << rgbl_last_error() << std::endl;
    return;
  }
  for (int i = 0; i < n; ++i) { mvKeysUn[i].pt.x = xy[2 * i]; mvKeysUn[i].pt.y = xy[2 * i + 1]; }
}

bool ORBextractor::CaptureDeviceFrame(rgbl_device_frame*& frame, int n, rgbl_depth* depth, const cv::Mat& K, const cv::Mat& mDistCoef,
                                      const float* grid6) {
  if (!mpHandle) { std::cerr << "[ORBextractor] CaptureDeviceFrame needs an extractor that has processed an image" << std::endl; return false; }
  if (frame && rgbl_device_frame_size(frame) < 0) return false;
  if (!frame && rgbl_device_frame_create(device, std::max(rgbl_extractor_max_keypoints(mpHandle), 1), &frame) != RGBL_OK) {
    std::cerr << "[ORBextractor] " << rgbl_last_error() << std::endl;
    return false;
  }
  float k[4] = {0, 0, 0, 0}, dist[5] = {0, 0, 0, 0, 0};
  const int nd = (int)mDistCoef.total();
  const bool und = !K.empty() && (nd == 4 || nd == 5);
  if (und) {
    k[0] = K.at<float>(0, 0); k[1] = K.at<float>(1, 1); k[2] = K.at<float>(0, 2); k[3] = K.at<float>(1, 2);
    for (int i = 0; i < nd; ++i) dist[i] = mDistCoef.at<float>(i);
  }
  if (rgbl_device_frame_capture(frame, mpHandle, 0, n, depth, und ? k : nullptr, und ? dist : nullptr, und ? nd : 0) != RGBL_OK) {
    std::cerr << "[ORBextractor] " << rgbl_last_error() << std::endl;
    return false;
  }
  if (grid6 && rgbl_device_frame_set_grid(frame, grid6) != RGBL_OK) {
    std::cerr << "[ORBextractor] " << rgbl_last_error() << std::endl;
    return false;
  }
  return true;
}

void ORBextractor::FillPyramid() {
  if (keepPyramid) {
    mvPyramidStorage.resize(nlevels);
    for (int l = 0; l < nlevels; ++l) {
      int w = 0, h = 0;
      rgbl_extractor_level_size(mpHandle, l, &w, &h);
      cv::Mat& full = mvPyramidStorage[l];
      full.create(h + 38, w + 38, CV_8UC1);
      rgbl_extractor_get_level(mpHandle, 0, l, 0, 1, full.data, (int)full.step);
#ifdef RGBL_HAVE_OPENCV
      mvImagePyramid[l] = full(cv::Rect(19, 19, w, h));  // an ROI inside the bordered buffer, like the reference
#else
      mvImagePyramid[l] = cv::Mat(h, w, CV_8UC1, full.data + 19 * full.step + 19, full.step);
#endif
    }
  }
}

}  // namespace ORB_SLAM3
