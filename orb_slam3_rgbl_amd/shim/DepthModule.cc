// Shim of ORB_SLAM3::DepthModule over the C ABI (include/rgbl_frontend.h). Replaces
// /root/reference/src/DepthModule.cc: the YAML handling and the error/printing behaviour stay on the host, the
// per-frame arithmetic (projection, scatter, up-sampling, keypoint gather) runs in the HIP kernels.
#include "DepthModule.h"

#include <stdlib.h>

#include <algorithm>
#include <fstream>
#include <iostream>
#include <sstream>

#include "../../include/rgbl_frontend.h"

#ifdef RGBL_HAVE_OPENCV
#include <opencv2/core/persistence.hpp>
#endif

namespace ORB_SLAM3 {

namespace {
// Settings access with the reference's semantics: "exists and is a real number" / "non-empty string".
struct Settings {
#ifdef RGBL_HAVE_OPENCV
  cv::FileStorage fs;
  explicit Settings(const std::string& path) : fs(path, cv::FileStorage::READ) {}
  bool real(const char* key, double* v) { cv::FileNode n = fs[key]; if (n.empty() || !n.isReal()) return false; *v = n.real(); return true; }
  bool number(const char* key, double* v) { cv::FileNode n = fs[key]; if (n.empty() || !(n.isReal() || n.isInt())) return false; *v = n.real(); return true; }
  std::string str(const char* key) { cv::FileNode n = fs[key]; return n.empty() || !n.isString() ? std::string() : (std::string)n; }
#else
  // Minimal reader for the flat "Key: value  # comment" files of Examples/*/*.yaml (no OpenCV available).
  std::map<std::string, std::string> kv;
  explicit Settings(const std::string& path) {
    std::ifstream f(path.c_str());
    std::string line;
    while (std::getline(f, line)) {
      const size_t hash = line.find('#');
      if (hash != std::string::npos) line.erase(hash);
      const size_t colon = line.find(':');
      if (colon == std::string::npos || line.compare(0, 5, "%YAML") == 0) continue;
      auto trim = [](std::string s) {
        const char* ws = " \t\r\n\"";
        const size_t a = s.find_first_not_of(ws), b = s.find_last_not_of(ws);
        return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
      };
      kv[trim(line.substr(0, colon))] = trim(line.substr(colon + 1));
    }
  }
  bool number(const char* key, double* v) {
    auto it = kv.find(key);
    if (it == kv.end() || it->second.empty()) return false;
    char* end = nullptr;
    const double d = strtod(it->second.c_str(), &end);
    if (end == it->second.c_str() || *end != '\0') return false;
    *v = d;
    return true;
  }
  // cv::FileNode::isReal(): the literal must carry a '.' or an exponent ("5" is an int node, "5.0" a real one)
  bool real(const char* key, double* v) {
    auto it = kv.find(key);
    return it != kv.end() && it->second.find_first_of(".eE") != std::string::npos && number(key, v);
  }
  std::string str(const char* key) { auto it = kv.find(key); return it == kv.end() ? std::string() : it->second; }
#endif
};
}  // namespace

DepthModule::DepthModule(const std::string& strSettingPath, const int /*sensor*/) {
  b_parse_LiDAR = false;
  b_parse_LiDAR = ParseRGBLParameters(strSettingPath);
  if (!b_parse_LiDAR) std::cout << "*Error in the LiDAR parameters in the config file*" << std::endl;
  b_parse_LiDARUpsampling = false;
  b_parse_LiDARUpsampling = ParseUpsamplingParameters(strSettingPath);
  if (!b_parse_LiDARUpsampling) std::cout << "*Error in the LiDAR upsampling parameters in the config file*" << std::endl;
  // What the device path cannot do is reported ONCE, here, and switches the module off like a parse failure does - not
  // once per frame from CalculateDepthFromPcd (the reference accepts any structuring-element size; the kernels hold 9 x 9).
  if (b_parse_LiDARUpsampling && b_parse_LiDAR) {
    const char* why = nullptr;
    if (SelectedUpsamlingMethod == InverseDilation) {
      const std::string& t = ParamUpsampling_InverseDilation_KernelType;
      const int ku = ParamUpsampling_InverseDilation_KernelSize_u, kv = ParamUpsampling_InverseDilation_KernelSize_v;
      if (t == "Diamond") { if (ku != 3 && ku != 5 && ku != 7 && ku != 9) why = "Diamond kernels exist for sizes 3, 5, 7 and 9 (DepthModule.h:138-161)"; }
      else if ((t == "Rectangle" || t == "Cross" || t == "Ellipse") && (ku < 1 || kv < 1 || ku > 9 || kv > 9)) why = "structuring elements larger than 9 x 9 are not supported by the device kernels";
    } else if (SelectedUpsamlingMethod == AverageFiltering &&
               (ParamUpsampling_AverageFilter_KernelSize < 1 || ParamUpsampling_AverageFilter_KernelSize > 9)) {
      why = "averaging kernels larger than 9 x 9 are not supported by the device kernels";
    }
    if (why) {
      std::cout << "*LiDAR upsampling disabled: " << why << "*" << std::endl;
      b_parse_LiDARUpsampling = false;
    }
  }
}

DepthModule::~DepthModule() { rgbl_depth_destroy(mpHandle); }

bool DepthModule::ParseRGBLParameters(const std::string& strSettingPath) {
  Settings s(strSettingPath);
  float K[12] = {0}, Tr[16] = {0};
  double v;
  const char* cam[4] = {"Camera.fx", "Camera.fy", "Camera.cx", "Camera.cy"};
  const int cam_pos[4] = {0, 5, 2, 6};
  for (int i = 0; i < 4; ++i) {
    if (!s.real(cam[i], &v)) { std::cout << "*" << cam[i] << " parameter doesn't exist or is not a real number*" << std::endl; return false; }
    K[cam_pos[i]] = (float)v;
  }
  K[10] = 1;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) {
      const std::string key = "LiDAR.Tr" + std::to_string(r + 1) + std::to_string(c + 1);
      if (!s.real(key.c_str(), &v)) { std::cout << "*" << key << " parameter doesn't exist or is not a real number*" << std::endl; return false; }
      Tr[4 * r + c] = (float)v;
    }
  Tr[15] = 1;
  LidarProjectionMatrix = cv::Mat(3, 4, CV_32F);
  rgbl_projection_matrix(K, Tr, LidarProjectionMatrix.ptr<float>());  // CameraMatrix * RotationMatrix (DepthModule.cc:434)
  std::cout << std::endl << "LiDAR to Camera Projection Matrix: " << std::endl;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 4; ++c) std::cout << LidarProjectionMatrix.at<float>(r, c) << (c < 3 ? ", " : ";\n");
  }
  if (!s.real("LiDAR.min_dist", &v)) { std::cout << "*LiDAR.min_dist parameter doesn't exist or is not a real number*" << std::endl; return false; }
  opt_min_dist = (float)v;
  if (!s.real("LiDAR.max_dist", &v)) { std::cout << "*LiDAR.max_dist parameter doesn't exist or is not a real number*" << std::endl; return false; }
  opt_max_dist = (float)v;
  if (!s.real("Camera.bf", &v)) { std::cout << "*Camera.bf parameter doesn't exist or is not a real number*" << std::endl; return false; }
  mbf = (float)v;
  const std::string MethodName = s.str("LiDAR.Method");
  std::cout << "Lidar Method: " << MethodName << std::endl;
  if (MethodName.empty()) { std::cout << "*LiDAR.Method parameter doesn't exist*" << std::endl; return false; }
  else if (MethodName == "None") SelectedUpsamlingMethod = None;
  else if (MethodName == "NearestNeighborPixel") SelectedUpsamlingMethod = NearestNeighborPixel;
  else if (MethodName == "AverageFiltering") SelectedUpsamlingMethod = AverageFiltering;
  else if (MethodName == "InverseDilation") SelectedUpsamlingMethod = InverseDilation;
  else if (MethodName == "IPBasic") SelectedUpsamlingMethod = IPBasic;
  else { std::cout << "*LiDAR.Method parameter indicated an unknown option.*" << std::endl; return false; }
  return true;
}

bool DepthModule::ParseUpsamplingParameters(const std::string& strSettingPath) {
  if (!b_parse_LiDAR) return false;  // SelectedUpsamlingMethod would be uninitialised (the reference reads it anyway)
  Settings s(strSettingPath);
  double v;
  switch (SelectedUpsamlingMethod) {
    case None:
      return true;
    case NearestNeighborPixel:
      if (!s.real("LiDAR.MethodNearestNeighborPixel.SearchDistance", &v)) {
        std::cout << "*LiDAR.MethodNearestNeighborPixel.SearchDistance parameter doesn't exist or is not a real number*" << std::endl;
        return false;
      }
      ParamUpsampling_NearestNeighborPixel_SearchRadius = (float)v;
      return true;
    case AverageFiltering:
      if (!s.real("LiDAR.MethodAverageFiltering.KernelSize", &v)) {
        std::cout << "*LiDAR.MethodAverageFiltering.KernelSize parameter doesn't exist or is not a real number*" << std::endl;
        return false;
      }
      ParamUpsampling_AverageFilter_KernelSize = (int)v;
      if (s.number("LiDAR.MethodAverageFiltering.bDoDilationPreprocessing", &v) && v == 0) ParamUpsampling_AverageFilter_bDoDilationPreprocessing = false;
      else if (s.number("LiDAR.MethodAverageFiltering.bDoDilationPreprocessing", &v) && v == 1) ParamUpsampling_AverageFilter_bDoDilationPreprocessing = true;
      else {
        std::cout << "*LiDAR.MethodAverageFiltering.bDoDilationPreprocessing parameter doesn't exist or is not 0 (false) or 1 (true)*" << std::endl;
        return false;
      }
      if (ParamUpsampling_AverageFilter_bDoDilationPreprocessing) {
        // parsed, but never used by Upsample_AverageFiltering (reference quirk, DepthModule.cc:200-228)
        if (!s.real("LiDAR.MethodAverageFiltering.DilationPreprocessing_KernelSize", &v)) {
          std::cout << "*LiDAR.MethodAverageFiltering.DilationPreprocessing_KernelSize parameter doesn't exist or is not a real number*" << std::endl;
          return false;
        }
        ParamUpsampling_AverageFilter_DilationPreprocessing_KernelSize = (int)v;
        ParamUpsampling_AverageFilter_DilationPreprocessing_KernelType = s.str("LiDAR.MethodAverageFiltering.DilationPreprocessing_KernelType");
      }
      // no break in the reference (DepthModule.cc:562-565): AverageFiltering also requires the InverseDilation keys
      // fall through
    case InverseDilation:
      if (!s.real("LiDAR.MethodInverseDilation.KernelSize_u", &v)) {
        std::cout << "*LiDAR.MethodInverseDilation.KernelSize_u parameter doesn't exist or is not a real number*" << std::endl;
        return false;
      }
      ParamUpsampling_InverseDilation_KernelSize_u = (int)v;
      if (!s.real("LiDAR.MethodInverseDilation.KernelSize_v", &v)) {
        std::cout << "*LiDAR.MethodInverseDilation.KernelSize_v parameter doesn't exist or is not a real number*" << std::endl;
        return false;
      }
      ParamUpsampling_InverseDilation_KernelSize_v = (int)v;
      ParamUpsampling_InverseDilation_KernelType = s.str("LiDAR.MethodInverseDilation.KernelType");
      return true;
    default:
      return false;  // IPBasic: declared, never implemented
  }
}

void DepthModule::EnsureHandle(int width, int height, int nPoints, int nKeys) {
  if (mpHandle && width == mHandleW && height == mHandleH && nPoints <= mHandlePoints && nKeys <= mHandleKeys) return;
  rgbl_depth_destroy(mpHandle);
  mpHandle = nullptr;
  rgbl_depth_cfg cfg;
  memset(&cfg, 0, sizeof(cfg));
  memcpy(cfg.proj, LidarProjectionMatrix.ptr<float>(), sizeof(float) * 12);
  cfg.min_dist = opt_min_dist;
  cfg.max_dist = opt_max_dist;
  cfg.mbf = mbf;
  cfg.method = (int)SelectedUpsamlingMethod;
  cfg.kernel_w = cfg.kernel_h = 1;
  cfg.kernel[0] = 1;
  if (SelectedUpsamlingMethod == InverseDilation) {
    const std::string& t = ParamUpsampling_InverseDilation_KernelType;
    const int shape = t == "Rectangle" ? 0 : t == "Cross" ? 1 : t == "Ellipse" ? 2 : t == "Diamond" ? 3 : -1;
    const int ku = ParamUpsampling_InverseDilation_KernelSize_u;
    const int kv = shape == 3 ? ku : ParamUpsampling_InverseDilation_KernelSize_v;
    if (shape < 0) {
      std::cerr << "Could not perform dilation, invalid kernel type: " << t << std::endl;
      std::cout << "Valid and implemented kernel types are: Rectangle, Cross, Ellipse and Diamond" << std::endl;
      return;
    }
    if (rgbl_structuring_element(shape, ku, kv, cfg.kernel) != RGBL_OK) {
      std::cerr << "Could not perform dilation, " << rgbl_last_error() << std::endl;
      return;
    }
    cfg.kernel_w = ku;
    cfg.kernel_h = kv;
  }
  cfg.avg_kernel_size = ParamUpsampling_AverageFilter_KernelSize;
  cfg.nn_search_radius = ParamUpsampling_NearestNeighborPixel_SearchRadius;
  cfg.width = width;
  cfg.height = height;
  cfg.max_points = std::max(nPoints, 250000);  // the KITTI loader caps a scan at 250000 points (rgbl_kitti.cc:153)
  cfg.max_keypoints = std::max(nKeys, 4096) * 2;
  cfg.max_batch = 1;
  if (rgbl_depth_create(&cfg, device, &mpHandle) != RGBL_OK) {
    std::cout << "*" << rgbl_last_error() << "*" << std::endl;
    mpHandle = nullptr;
    return;
  }
  mHandleW = width; mHandleH = height; mHandlePoints = cfg.max_points; mHandleKeys = cfg.max_keypoints;
}

void DepthModule::CalculateDepthFromPcd(std::vector<cv::KeyPoint> mvKeys, std::vector<cv::KeyPoint> mvKeysUn,
                                        const cv::Mat& PointCloud, const int imwidth, const int imheight) {
  Compute(mvKeys, mvKeysUn, PointCloud.ptr<float>(), PointCloud.cols, (int)(PointCloud.step / sizeof(float)), false, imwidth, imheight);
}

void DepthModule::PrefetchPointcloud(const cv::Mat& PointCloud, const int imwidth, const int imheight) {
  if (!b_parse_LiDARUpsampling || !b_parse_LiDAR || SelectedUpsamlingMethod == None || SelectedUpsamlingMethod == IPBasic) return;
  EnsureHandle(imwidth, imheight, PointCloud.cols, std::max(mHandleKeys / 2, 4096));
  if (!mpHandle) return;
  (void)rgbl_depth_prefetch(mpHandle, PointCloud.ptr<float>(), PointCloud.cols, (int)(PointCloud.step / sizeof(float)), imwidth, imheight);
}

void DepthModule::CancelPrefetch() {
  if (mpHandle) (void)rgbl_depth_prefetch_cancel(mpHandle);
}

// The scan exactly as read from a KITTI velodyne .bin file (nPoints records x, y, z, reflectance): what the example's
// LoadPointcloudBinaryMat (Examples/RGB-L/rgbl_kitti.cc:151-185) repacks into the 4 x N matrix, without the repack.
void DepthModule::CalculateDepthFromKittiBin(const std::vector<cv::KeyPoint>& mvKeys, const std::vector<cv::KeyPoint>& mvKeysUn,
                                             const float* xyzi, const int nPoints, const int imwidth, const int imheight) {
  Compute(mvKeys, mvKeysUn, xyzi, nPoints, nPoints, true, imwidth, imheight);
}

void DepthModule::Compute(const std::vector<cv::KeyPoint>& mvKeys, const std::vector<cv::KeyPoint>& mvKeysUn, const float* cloud,
                          int nPoints, int ld, bool xyzi, int imwidth, int imheight) {
  if (!b_parse_LiDARUpsampling || !b_parse_LiDAR) {
    std::cout << "*Cannot perform LiDAR Upsampling since parameters were missing in the config file.*" << std::endl;
    return;
  }
  if (SelectedUpsamlingMethod == None) {
    // the reference only projects in this case (DepthModule.cc:58-61); mvDepth / mvuRight stay untouched
    return;
  }
  if (SelectedUpsamlingMethod == IPBasic) {
    std::cout << "*Desired Upsampling Method was not yet implemented.*";
    return;
  }
  const int N = (int)mvKeys.size();
  // every reachable method of the reference starts from N x -1 (DepthModule.cc:86-87, 163-164): Frame copies these vectors
  // right after the call (Frame.cc:332-333), so they must have the right size even when the device path fails below
  mvuRight = std::vector<float>(N, -1);
  mvDepth = std::vector<float>(N, -1);
  EnsureHandle(imwidth, imheight, nPoints, N);
  if (!mpHandle) return;
  std::vector<float> kp_xy(2 * (size_t)N), kpun_x(N);
  for (int i = 0; i < N; ++i) {
    kp_xy[2 * i] = mvKeys[i].pt.x;
    kp_xy[2 * i + 1] = mvKeys[i].pt.y;
    kpun_x[i] = mvKeysUn[i].pt.x;
  }
  float *raw = nullptr, *proc = nullptr;
  if (downloadDenseMaps) {
    RawDepthMap.create(imheight, imwidth, CV_32F);
    raw = RawDepthMap.ptr<float>();
    if (SelectedUpsamlingMethod != NearestNeighborPixel) {  // that method never writes ProcessedDepthMap (reference quirk)
      ProcessedDepthMap.create(imheight, imwidth, CV_32F);
      proc = ProcessedDepthMap.ptr<float>();
    }
  }
  const int rc = xyzi ? rgbl_depth_compute_xyzi(mpHandle, cloud, nPoints, imwidth, imheight, kp_xy.data(), kpun_x.data(), N,
                                                mvDepth.data(), mvuRight.data(), raw, proc)
                      : rgbl_depth_compute(mpHandle, cloud, nPoints, ld, imwidth, imheight, kp_xy.data(), kpun_x.data(), N,
                                           mvDepth.data(), mvuRight.data(), raw, proc);
  if (rc != RGBL_OK) std::cout << "*" << rgbl_last_error() << "*" << std::endl;
}

}  // namespace ORB_SLAM3
