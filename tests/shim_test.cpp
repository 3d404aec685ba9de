// Exercises the C++ drop-in shims (orb_slam3_rgbl_amd/shim) the way System/Tracking/Frame/LocalMapping use the
// reference classes, and dumps the results for the Python side to compare with the oracle.
//   shim_test <yaml> <image.raw> <w> <h> <cloud.raw> <n> <tri.bin> <out.bin>
//   shim_test --parse <yaml>      prints the DepthModule parse flags (bit 0 LiDAR, bit 1 up-sampling)
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <set>
#include <tuple>
#include <string>
#include <vector>

#include "DepthModule.h"
#include "ORBextractor.h"
#include "ORBmatcher.h"
#include "ORBVocabulary.h"

#include "shim_standins.h"

// ---- stand-ins for the Frame / MapPoint members SearchByProjection touches
struct Quat { float qx, qy, qz, qw; float x() const { return qx; } float y() const { return qy; } float z() const { return qz; } float w() const { return qw; } };
struct V3n : V3 {  // what Eigen's "a - b" offers the relocalisation matcher: norm() as sqrt of the unrolled sum x*x + (y*y + z*z)
  V3n(const V3& a) : V3(a) {}
  float norm() const { return sqrtf(v[0] * v[0] + (v[1] * v[1] + v[2] * v[2])); }
  float dot(const V3& o) const { return v[0] * o.v[0] + (v[1] * o.v[1] + v[2] * o.v[2]); }
};
static V3n operator-(const V3& a, const V3& b) { return V3n(V3{{a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2]}}); }
struct QPose {
  Quat q; V3 t;
  const Quat& unit_quaternion() const { return q; }
  const V3& translation() const { return t; }
  // Sophus SE3::inverse(): (q^-1, q^-1 * (-t)), the rotation as Eigen's QuaternionBase::_transformVector
  QPose inverse() const {
    const float c[4] = {-q.qx, -q.qy, -q.qz, q.qw};
    const float p[3] = {-t.v[0], -t.v[1], -t.v[2]};
    float ux = c[1] * p[2] - c[2] * p[1], uy = c[2] * p[0] - c[0] * p[2], uz = c[0] * p[1] - c[1] * p[0];
    ux = ux + ux; uy = uy + uy; uz = uz + uz;
    const float cx = c[1] * uz - c[2] * uy, cy = c[2] * ux - c[0] * uz, cz = c[0] * uy - c[1] * ux;
    QPose r;
    r.q = Quat{c[0], c[1], c[2], c[3]};
    r.t = V3{{p[0] + c[3] * ux + cx, p[1] + c[3] * uy + cy, p[2] + c[3] * uz + cz}};
    return r;
  }
};
struct TestFrame;
struct TrackedPoint {
  V3 pos; cv::Mat desc; int nObs = 0;
  float mfMinDistance = 0, mfMaxDistance = 0;
  float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; }
  float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }
  int PredictScale(const float& currentDist, TestFrame* pF);  // MapPoint.cc:531-546
  bool mbTrackInView = false, mbTrackInViewR = false, bad = false;
  float mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = 0, mTrackDepth = 0, mTrackViewCos = 0;
  int mnTrackScaleLevel = 0;
  bool isBad() { return bad; }
  V3 GetWorldPos() { return pos; }
  cv::Mat GetDescriptor() { return desc; }
  int Observations() { return nObs; }
};
struct TestFrame {
  int N = 0, Nleft = -1;
  float mb = 0, mbf = 0, mfLogScaleFactor = 0;
  int mnScaleLevels = 0;
  Camera* mpCamera = nullptr;
  std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
  std::vector<float> mvuRight, mvScaleFactors;
  std::vector<TrackedPoint*> mvpMapPoints;
  std::vector<bool> mvbOutlier;
  cv::Mat mDescriptors;
  QPose pose;
  QPose GetPose() const { return pose; }
  static float mnMinX, mnMinY, mnMaxX, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv;
};
float TestFrame::mnMinX, TestFrame::mnMinY, TestFrame::mnMaxX, TestFrame::mnMaxY, TestFrame::mfGridElementWidthInv,
    TestFrame::mfGridElementHeightInv;
int TrackedPoint::PredictScale(const float& currentDist, TestFrame* pF) {
  const float ratio = mfMaxDistance / currentDist;
  int nScale = ceil(log(ratio) / pF->mfLogScaleFactor);
  if (nScale < 0) nScale = 0;
  else if (nScale >= pF->mnScaleLevels) nScale = pF->mnScaleLevels - 1;
  return nScale;
}
// ---- stand-ins for what ORBmatcher::Fuse touches
struct FuseKF;
struct FusePoint {
  V3 pos, normal; cv::Mat desc; bool bad = false, in_kf = false; int nObs = 4;
  float mfMinDistance = 0, mfMaxDistance = 0;
  bool isBad() { return bad; }
  bool IsInKeyFrame(FuseKF*) { return in_kf; }
  V3 GetWorldPos() { return pos; }
  V3 GetNormal() { return normal; }
  cv::Mat GetDescriptor() { return desc; }
  int Observations() { return nObs; }
  float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; }
  float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }
  int PredictScale(const float& currentDist, FuseKF* pKF);
  void AddObservation(FuseKF*, int) { in_kf = true; ++nObs; }
  void Replace(FusePoint*) {}  // rewires the map; the test reads the fused feature from FuseKF::GetMapPoint's log
};
struct FuseKF {
  int N = 0, NLeft = -1;
  float mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0, mfGridElementWidthInv = 0, mfGridElementHeightInv = 0, mbf = 0, mfLogScaleFactor = 0;
  int mnScaleLevels = 0;
  Camera* mpCamera = nullptr;
  std::vector<cv::KeyPoint> mvKeysUn;
  std::vector<float> mvuRight, mvScaleFactors, mvInvLevelSigma2;
  std::vector<FusePoint*> mps;
  std::vector<int> queried;  // every GetMapPoint(idx) of the bookkeeping loop = the feature a point was fused with
  cv::Mat mDescriptors;
  QPose pose; V3 Ow;
  QPose GetPose() { return pose; }
  V3 GetCameraCenter() { return Ow; }
  FusePoint* GetMapPoint(size_t idx) { queried.push_back((int)idx); return mps[idx]; }
  void AddMapPoint(FusePoint* p, size_t idx) { mps[idx] = p; }
};
int FusePoint::PredictScale(const float& currentDist, FuseKF* pKF) {
  const float ratio = mfMaxDistance / currentDist;
  int nScale = ceil(log(ratio) / pKF->mfLogScaleFactor);
  if (nScale < 0) nScale = 0;
  else if (nScale >= pKF->mnScaleLevels) nScale = pKF->mnScaleLevels - 1;
  return nScale;
}
// ---- stand-ins for the Sim3 matchers (SearchBySim3, Fuse(pKF, Scw, ...)): matrix-form transforms, exact for the identities used
static V3 operator/(const V3& a, float s) { return V3{{a.v[0] / s, a.v[1] / s, a.v[2] / s}}; }
struct TestSE3 {
  M3 R; V3 t;
  TestSE3() { for (int i = 0; i < 9; ++i) R.m[i] = (i % 4 == 0) ? 1.f : 0.f; t = V3{{0, 0, 0}}; }
  TestSE3(const M3& R_, const V3& t_) : R(R_), t(t_) {}
  V3n operator*(const V3& p) const {
    V3 r;
    for (int i = 0; i < 3; ++i) r.v[i] = R.m[3 * i] * p.v[0] + R.m[3 * i + 1] * p.v[1] + R.m[3 * i + 2] * p.v[2] + t.v[i];
    return V3n(r);
  }
  TestSE3 inverse() const {
    TestSE3 r;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.R.m[3 * i + j] = R.m[3 * j + i];
    for (int i = 0; i < 3; ++i) r.t.v[i] = -(r.R.m[3 * i] * t.v[0] + r.R.m[3 * i + 1] * t.v[1] + r.R.m[3 * i + 2] * t.v[2]);
    return r;
  }
  V3 translation() const { return t; }
};
struct TestSim3 {
  TestSE3 T; float s = 1.f;
  M3 rotationMatrix() const { return T.R; }
  V3 translation() const { return T.t; }
  float scale() const { return s; }
  TestSim3 inverse() const { TestSim3 r; r.T = T.inverse(); r.s = 1.f / s; for (int i = 0; i < 3; ++i) r.T.t.v[i] *= r.s; return r; }
  V3n operator*(const V3& p) const {
    V3 q;
    for (int i = 0; i < 3; ++i) q.v[i] = s * (T.R.m[3 * i] * p.v[0] + T.R.m[3 * i + 1] * p.v[1] + T.R.m[3 * i + 2] * p.v[2]) + T.t.v[i];
    return V3n(q);
  }
};
struct SimKF;
struct SimPoint {
  V3 pos, normal; cv::Mat desc; bool bad = false; int nObs = 3;
  float mfMinDistance = 0, mfMaxDistance = 0;
  SimKF* home = nullptr; int home_idx = -1;
  bool isBad() { return bad; }
  V3 GetWorldPos() { return pos; }
  V3 GetNormal() { return normal; }
  cv::Mat GetDescriptor() { return desc; }
  float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; }
  float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }
  int PredictScale(const float& currentDist, SimKF* pKF);
  std::tuple<int, int> GetIndexInKeyFrame(SimKF* pKF) { return std::tuple<int, int>(pKF == home ? home_idx : -1, -1); }
  void AddObservation(SimKF* pKF, int idx) { home = pKF; home_idx = idx; }
};
struct SimKF {
  int N = 0;
  float mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0, mfGridElementWidthInv = 0, mfGridElementHeightInv = 0;
  float fx = 0, fy = 0, cx = 0, cy = 0, mfLogScaleFactor = 0;
  int mnScaleLevels = 0;
  std::vector<cv::KeyPoint> mvKeysUn;
  std::vector<float> mvScaleFactors;
  std::vector<SimPoint*> mps;
  std::vector<int> queried;
  cv::Mat mDescriptors;
  TestSE3 GetPose() { return TestSE3(); }
  std::vector<SimPoint*> GetMapPointMatches() { return mps; }
  std::set<SimPoint*> GetMapPoints() { std::set<SimPoint*> r; for (SimPoint* p : mps) if (p) r.insert(p); return r; }
  SimPoint* GetMapPoint(size_t idx) { queried.push_back((int)idx); return mps[idx]; }
  void AddMapPoint(SimPoint* p, size_t idx) { mps[idx] = p; }
};
int SimPoint::PredictScale(const float& currentDist, SimKF* pKF) {
  const float ratio = mfMaxDistance / currentDist;
  int nScale = ceil(log(ratio) / pKF->mfLogScaleFactor);
  if (nScale < 0) nScale = 0;
  else if (nScale >= pKF->mnScaleLevels) nScale = pKF->mnScaleLevels - 1;
  return nScale;
}
struct RelocKeyFrame {
  std::vector<cv::KeyPoint> mvKeysUn;
  std::vector<TrackedPoint*> mps;
  std::vector<TrackedPoint*> GetMapPointMatches() { return mps; }
};

struct DepthProbe : ORB_SLAM3::DepthModule {  // the parse flags are protected, as in the reference
  DepthProbe(const std::string& path) : ORB_SLAM3::DepthModule(path, 6) {}
  int flags() const { return (b_parse_LiDAR ? 1 : 0) | (b_parse_LiDARUpsampling ? 2 : 0); }
};

int main(int argc, char** argv) {
  if (argc == 3 && std::string(argv[1]) == "--parse") {  // settings-file outcomes only
    DepthProbe probe(argv[2]);
    printf("\nparse_flags %d\n", probe.flags());
    return 0;
  }
  if (argc < 9) return 2;
  const int w = atoi(argv[3]), h = atoi(argv[4]), n = atoi(argv[6]);
  cv::Mat im(h, w, CV_8UC1);
  FILE* f = fopen(argv[2], "rb");
  if (!f || !rd(f, im.data, (size_t)w * h)) return 3;
  fclose(f);
  cv::Mat pcd(4, n, CV_32F);
  f = fopen(argv[5], "rb");
  if (!f || !rd(f, pcd.ptr<float>(), (size_t)4 * n)) return 4;
  fclose(f);
  FILE* out = fopen(argv[8], "wb");

  // --- as Tracking::ParseORBParamFile + Frame::ExtractORB (Tracking.cc:1281, Frame.cc:508-515)
  ORB_SLAM3::ORBextractor extractor(2000, 1.2f, 8, 12, 7);
  extractor.keepPyramid = true;
  std::vector<cv::KeyPoint> keys;
  cv::Mat desc;
  std::vector<int> vLapping = {0, 0};
  const int mono = extractor(im, cv::Mat(), keys, desc, vLapping);
  const int nk = (int)keys.size();
  wr(out, &mono, 1); wr(out, &nk, 1);
  wr(out, keys.data(), nk); wr(out, desc.data, (size_t)nk * 32);
  std::vector<float> sf = extractor.GetScaleFactors();
  wr(out, sf.data(), sf.size());
  const cv::Mat& p3 = extractor.mvImagePyramid[3];
  wr(out, &p3.cols, 1); wr(out, &p3.rows, 1);
  for (int r = -19; r < p3.rows + 19; ++r) wr(out, p3.data + (ptrdiff_t)r * (ptrdiff_t)p3.step - 19, p3.cols + 38);  // incl. the border
  cv::Mat empty;
  std::vector<cv::KeyPoint> k2; cv::Mat d2;
  const int mono_empty = extractor(empty, cv::Mat(), k2, d2, vLapping);
  wr(out, &mono_empty, 1);

  // --- as System (System.cc:220-221) + Frame (Frame.cc:331-333)
  ORB_SLAM3::DepthModule depth(argv[1], 6);
  depth.CalculateDepthFromPcd(keys, keys, pcd, w, h);
  const int nd = (int)depth.mvDepth.size();
  wr(out, &nd, 1);
  wr(out, depth.mvDepth.data(), nd); wr(out, depth.mvuRight.data(), nd);
  wr(out, depth.ProcessedDepthMap.ptr<float>(), (size_t)w * h);
  wr(out, depth.LidarProjectionMatrix.ptr<float>(), 12);

  // --- ingest: the .bin layout straight into the depth module, cvtColor + extraction in one call
  {
    std::vector<float> xyzi((size_t)4 * n);
    for (int i = 0; i < n; ++i) {
      xyzi[4 * i] = pcd.ptr<float>(0)[i]; xyzi[4 * i + 1] = pcd.ptr<float>(1)[i]; xyzi[4 * i + 2] = pcd.ptr<float>(2)[i];
      xyzi[4 * i + 3] = 0.25f + (i % 7) * 0.1f;  // reflectance: must not matter
    }
    depth.CalculateDepthFromKittiBin(keys, keys, xyzi.data(), n, w, h);
    const int nb = (int)depth.mvDepth.size();
    wr(out, &nb, 1);
    wr(out, depth.mvDepth.data(), nb); wr(out, depth.mvuRight.data(), nb);
    std::vector<unsigned char> bgr((size_t)3 * w * h);
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) {
        const int g = im.data[(size_t)y * im.step + x];
        unsigned char* q = &bgr[((size_t)y * w + x) * 3];
        q[0] = (unsigned char)g; q[1] = (unsigned char)(255 - g); q[2] = (unsigned char)(g / 2 + (x % 50));
      }
    cv::Mat gray, dcol;
    std::vector<cv::KeyPoint> kcol;
    const int mono_col = extractor.ExtractColor(bgr.data(), 3, 3 * w, w, h, /*bRGB=*/true, gray, kcol, dcol, vLapping);
    const int nc = (int)kcol.size();
    wr(out, &mono_col, 1); wr(out, &nc, 1);
    wr(out, kcol.data(), nc); wr(out, dcol.data, (size_t)nc * 32);
    for (int y = 0; y < h; ++y) wr(out, gray.data + (size_t)y * gray.step, w);
    // --- as Frame::UndistortKeyPoints (Frame.cc:837-870) with the EuRoC cam0 model
    cv::Mat Kmat(3, 3, CV_32F), dist(4, 1, CV_32F);
    for (int i = 0; i < 9; ++i) Kmat.at<float>(i / 3, i % 3) = 0.f;
    Kmat.at<float>(0, 0) = 458.654f; Kmat.at<float>(1, 1) = 457.296f; Kmat.at<float>(0, 2) = 367.215f; Kmat.at<float>(1, 2) = 248.375f;
    Kmat.at<float>(2, 2) = 1.f;
    const float dc[4] = {-0.28340811f, 0.07395907f, 0.00019359f, 1.76187114e-05f};
    for (int i = 0; i < 4; ++i) dist.at<float>(i) = dc[i];
    std::vector<cv::KeyPoint> kun;
    extractor.UndistortKeyPoints(kcol, Kmat, dist, kun);
    for (int i = 0; i < nc; ++i) { wr(out, &kun[i].pt.x, 1); wr(out, &kun[i].pt.y, 1); }
  }

  // --- as LocalMapping::CreateNewMapPoints (LocalMapping.cc:412,466)
  f = fopen(argv[7], "rb");
  Camera cam; MapPoint some; KeyFrame kf1, kf2;
  if (!f || !rd(f, cam.p, 4) || !load_kf(f, kf1, &cam, &some) || !load_kf(f, kf2, &cam, &some)) return 5;
  fclose(f);
  ORB_SLAM3::ORBmatcher matcher(0.6, false);
  std::vector<std::pair<size_t, size_t> > pairs;
  const int nm = matcher.SearchForTriangulation(&kf1, &kf2, pairs, false, false);
  const int np = (int)pairs.size();
  wr(out, &nm, 1); wr(out, &np, 1);
  for (auto& pr : pairs) { int a = (int)pr.first, b = (int)pr.second; wr(out, &a, 1); wr(out, &b, 1); }
  const int dd = ORB_SLAM3::ORBmatcher::DescriptorDistance(kf1.mDescriptors.row(0), kf2.mDescriptors.row(0));
  wr(out, &dd, 1);
  {
    // --- the same call with both key frames resident on the device (KeyFrame::mpDeviceFrame): same pairs
    rgbl_device_frame* d[2] = {nullptr, nullptr};
    KeyFrame* kfs[2] = {&kf1, &kf2};
    for (int k = 0; k < 2; ++k) {
      KeyFrame& kf = *kfs[k];
      std::vector<float> xy(2 * (size_t)kf.N);
      std::vector<int32_t> oct(kf.N);
      for (int i = 0; i < kf.N; ++i) { xy[2 * i] = kf.mvKeysUn[i].pt.x; xy[2 * i + 1] = kf.mvKeysUn[i].pt.y; oct[i] = kf.mvKeysUn[i].octave; }
      if (rgbl_device_frame_create(0, std::max(kf.N, 1), &d[k]) != RGBL_OK ||
          rgbl_device_frame_upload(d[k], kf.N, kf.mDescriptors.data, xy.data(), oct.data(), kf.mvuRight.data()) != RGBL_OK) {
        fprintf(stderr, "device frame: %s\n", rgbl_last_error());
        return 20;
      }
      kf.mpDeviceFrame = d[k];
    }
    std::vector<std::pair<size_t, size_t> > pairs_res;
    const int nm_res = matcher.SearchForTriangulation(&kf1, &kf2, pairs_res, false, false);
    if (nm_res != nm || pairs_res != pairs) { fprintf(stderr, "SearchForTriangulation on resident key frames differs\n"); return 21; }
    kf1.mpDeviceFrame = kf2.mpDeviceFrame = nullptr;
    rgbl_device_frame_destroy(d[0]); rgbl_device_frame_destroy(d[1]);
    // --- and the frame the extractor + depth module have just produced, captured device to device
    std::vector<cv::KeyPoint> kc; cv::Mat dc;
    extractor(im, cv::Mat(), kc, dc, vLapping);
    depth.CalculateDepthFromPcd(kc, kc, pcd, w, h);
    rgbl_device_frame* cap = nullptr;
    if (!extractor.CaptureDeviceFrame(cap, (int)kc.size(), depth.Handle())) return 22;
    const int nc2 = rgbl_device_frame_size(cap);
    std::vector<uint8_t> gd((size_t)nc2 * 32);
    std::vector<float> gxy(2 * (size_t)nc2), gur(nc2);
    std::vector<int32_t> goct(nc2);
    if (nc2 != (int)kc.size() || rgbl_device_frame_download(cap, gd.data(), gxy.data(), goct.data(), gur.data()) != RGBL_OK) return 23;
    bool same = memcmp(gd.data(), dc.data, gd.size()) == 0;
    for (int i = 0; i < nc2; ++i)
      same = same && gxy[2 * i] == kc[i].pt.x && gxy[2 * i + 1] == kc[i].pt.y && goct[i] == kc[i].octave &&
             memcmp(&gur[i], &depth.mvuRight[i], 4) == 0;
    rgbl_device_frame_destroy(cap);
    if (!same) { fprintf(stderr, "captured device frame differs from what the drop-in classes returned\n"); return 24; }
  }
  {
    // --- as Tracking::TrackReferenceKeyFrame (Tracking.cc:2798-2810): matcher(0.7, true).SearchByBoW(mpReferenceKF, mCurrentFrame, vpMapPointMatches)
    kf2.mvKeys = kf2.mvKeysUn;
    ORB_SLAM3::ORBmatcher bow(0.7, true);
    std::vector<MapPoint*> vpMapPointMatches;
    std::vector<MapPoint> own(kf1.N);
    for (int i = 0; i < kf1.N; ++i) if (kf1.mvpMapPoints[i]) kf1.mvpMapPoints[i] = &own[i];  // distinct points: indices can be read back
    const int nbow = bow.SearchByBoW(&kf1, kf2, vpMapPointMatches);
    const int n2b = (int)vpMapPointMatches.size();
    wr(out, &nbow, 1); wr(out, &n2b, 1);
    for (int i = 0; i < n2b; ++i) { const int idx = vpMapPointMatches[i] ? (int)(vpMapPointMatches[i] - own.data()) : -1; wr(out, &idx, 1); }
    // --- the same with a two-camera frame (F.Nleft != -1, ORBmatcher.cc:298-326, 357-386): the second half of the features as the right camera's
    kf2.Nleft = kf2.N / 2;
    kf2.mvKeys.assign(kf2.mvKeysUn.begin(), kf2.mvKeysUn.begin() + kf2.Nleft);
    kf2.mvKeysRight.assign(kf2.mvKeysUn.begin() + kf2.Nleft, kf2.mvKeysUn.end());
    const int nrig = bow.SearchByBoW(&kf1, kf2, vpMapPointMatches);
    wr(out, &nrig, 1);
    for (int i = 0; i < n2b; ++i) { const int idx = vpMapPointMatches[i] ? (int)(vpMapPointMatches[i] - own.data()) : -1; wr(out, &idx, 1); }
    kf2.Nleft = -1;
    kf2.mvKeys = kf2.mvKeysUn;
    kf2.mvKeysRight.clear();
  }
  // --- as Tracking::TrackWithMotionModel (Tracking.cc:2913-2934): SearchByProjection(mCurrentFrame, mLastFrame, th, bMono)
  if (argc > 9) {
    f = fopen(argv[9], "rb");
    int n1 = 0, n2 = 0, mono = 0, ori = 0;
    float th = 0, hdr[6 + 7 + 7 + 4 + 2 + 8];
    if (!f || !rd(f, &n1, 1) || !rd(f, &n2, 1) || !rd(f, &th, 1) || !rd(f, &mono, 1) || !rd(f, &ori, 1) || !rd(f, hdr, 34)) return 6;
    std::vector<unsigned char> valid(n1), obs(n1), d1((size_t)n1 * 32), d2((size_t)n2 * 32);
    std::vector<float> pos((size_t)n1 * 3), ang1(n1), xy2((size_t)n2 * 2), ang2(n2), ur2(n2);
    std::vector<int> o1(n1), o2(n2);
    rd(f, valid.data(), n1); rd(f, pos.data(), (size_t)n1 * 3); rd(f, d1.data(), (size_t)n1 * 32); rd(f, obs.data(), n1);
    rd(f, o1.data(), n1); rd(f, ang1.data(), n1); rd(f, xy2.data(), (size_t)n2 * 2); rd(f, o2.data(), n2); rd(f, ang2.data(), n2);
    rd(f, ur2.data(), n2); rd(f, d2.data(), (size_t)n2 * 32);
    fclose(f);
    TestFrame::mnMinX = hdr[0]; TestFrame::mnMinY = hdr[1]; TestFrame::mnMaxX = hdr[2]; TestFrame::mnMaxY = hdr[3];
    TestFrame::mfGridElementWidthInv = hdr[4]; TestFrame::mfGridElementHeightInv = hdr[5];
    Camera pcam; for (int k = 0; k < 4; ++k) pcam.p[k] = hdr[20 + k];
    TestFrame last, cur;
    std::vector<TrackedPoint> pts(n1);
    last.N = n1; last.mvKeys.resize(n1); last.mvKeysUn.resize(n1); last.mvpMapPoints.assign(n1, nullptr); last.mvbOutlier.assign(n1, false);
    for (int i = 0; i < n1; ++i) {
      last.mvKeys[i].octave = o1[i]; last.mvKeysUn[i].angle = ang1[i];
      if (!valid[i]) { if (i % 2) last.mvbOutlier[i] = true, last.mvpMapPoints[i] = &pts[i]; continue; }  // outliers and NULLs
      pts[i].pos = V3{{pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]}};
      pts[i].desc.create(1, 32, CV_8U); memcpy(pts[i].desc.data, &d1[(size_t)i * 32], 32);
      pts[i].nObs = obs[i] ? 2 : 0;
      last.mvpMapPoints[i] = &pts[i];
    }
    last.pose.q = Quat{hdr[13], hdr[14], hdr[15], hdr[16]}; last.pose.t = V3{{hdr[17], hdr[18], hdr[19]}};
    cur.N = n2; cur.mpCamera = &pcam; cur.mb = hdr[24]; cur.mbf = hdr[25];
    cur.mvKeysUn.resize(n2);
    for (int i = 0; i < n2; ++i) { cur.mvKeysUn[i].pt.x = xy2[2 * i]; cur.mvKeysUn[i].pt.y = xy2[2 * i + 1]; cur.mvKeysUn[i].octave = o2[i]; cur.mvKeysUn[i].angle = ang2[i]; }
    cur.mvKeys = cur.mvKeysUn; cur.mvuRight = ur2;
    cur.mDescriptors.create(n2, 32, CV_8U); memcpy(cur.mDescriptors.data, d2.data(), (size_t)n2 * 32);
    cur.mvScaleFactors.assign(hdr + 26, hdr + 34);
    cur.mvpMapPoints.assign(n2, nullptr);
    cur.pose.q = Quat{hdr[6], hdr[7], hdr[8], hdr[9]}; cur.pose.t = V3{{hdr[10], hdr[11], hdr[12]}};
    ORB_SLAM3::ORBmatcher tracker(0.9, ori != 0);
    const int nproj = tracker.SearchByProjection(cur, last, th, mono != 0);
    wr(out, &nproj, 1); wr(out, &n2, 1);
    for (int i = 0; i < n2; ++i) { const int idx = cur.mvpMapPoints[i] ? (int)(cur.mvpMapPoints[i] - pts.data()) : -1; wr(out, &idx, 1); }
  }
  // --- as Tracking::SearchLocalPoints (Tracking.cc:3428-3447): matcher.SearchByProjection(mCurrentFrame, mvpLocalMapPoints, th, ...)
  if (argc > 10) {
    f = fopen(argv[10], "rb");
    int n1 = 0, n2 = 0;
    float th = 0, ratio = 0, hdr[6 + 8];
    if (!f || !rd(f, &n1, 1) || !rd(f, &n2, 1) || !rd(f, &th, 1) || !rd(f, &ratio, 1) || !rd(f, hdr, 14)) return 7;
    std::vector<unsigned char> valid(n1), obs(n1), d1((size_t)n1 * 32), d2((size_t)n2 * 32), blocked(n2);
    std::vector<float> proj((size_t)n1 * 3), vcos(n1), xy2((size_t)n2 * 2), ur2(n2);
    std::vector<int> lev(n1), o2(n2);
    rd(f, valid.data(), n1); rd(f, proj.data(), (size_t)n1 * 3); rd(f, lev.data(), n1); rd(f, vcos.data(), n1);
    rd(f, d1.data(), (size_t)n1 * 32); rd(f, obs.data(), n1); rd(f, xy2.data(), (size_t)n2 * 2); rd(f, o2.data(), n2);
    rd(f, ur2.data(), n2); rd(f, d2.data(), (size_t)n2 * 32); rd(f, blocked.data(), n2);
    fclose(f);
    TestFrame::mnMinX = hdr[0]; TestFrame::mnMinY = hdr[1]; TestFrame::mnMaxX = hdr[2]; TestFrame::mnMaxY = hdr[3];
    TestFrame::mfGridElementWidthInv = hdr[4]; TestFrame::mfGridElementHeightInv = hdr[5];
    std::vector<TrackedPoint> pts(n1);
    std::vector<TrackedPoint*> vp(n1);
    for (int i = 0; i < n1; ++i) {
      TrackedPoint& p = pts[i];
      p.mbTrackInView = valid[i] != 0;
      if (!valid[i] && i % 3 == 0) { p.mbTrackInView = true; p.bad = true; }  // bad points are skipped as well
      p.mTrackProjX = proj[3 * i]; p.mTrackProjY = proj[3 * i + 1]; p.mTrackProjXR = proj[3 * i + 2];
      p.mnTrackScaleLevel = lev[i]; p.mTrackViewCos = vcos[i];
      p.desc.create(1, 32, CV_8U); memcpy(p.desc.data, &d1[(size_t)i * 32], 32);
      p.nObs = obs[i] ? 4 : 0;
      vp[i] = &p;
    }
    TrackedPoint old_point; old_point.nObs = 7;
    TestFrame F;
    F.N = n2; F.mvKeysUn.resize(n2);
    for (int i = 0; i < n2; ++i) { F.mvKeysUn[i].pt.x = xy2[2 * i]; F.mvKeysUn[i].pt.y = xy2[2 * i + 1]; F.mvKeysUn[i].octave = o2[i]; }
    F.mvuRight = ur2;
    F.mDescriptors.create(n2, 32, CV_8U); memcpy(F.mDescriptors.data, d2.data(), (size_t)n2 * 32);
    F.mvScaleFactors.assign(hdr + 6, hdr + 14);
    F.mvpMapPoints.assign(n2, nullptr);
    for (int i = 0; i < n2; ++i) if (blocked[i]) F.mvpMapPoints[i] = &old_point;
    ORB_SLAM3::ORBmatcher local(ratio, true);
    const int nloc = local.SearchByProjection(F, vp, th);
    wr(out, &nloc, 1); wr(out, &n2, 1);
    for (int i = 0; i < n2; ++i) {
      TrackedPoint* p = F.mvpMapPoints[i];
      const int idx = (p && p != &old_point) ? (int)(p - pts.data()) : -1;
      wr(out, &idx, 1);
    }
  }
  // --- as Frame::ComputeBoW (Frame.cc:828-835): mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4)
  if (argc > 11) {
    ORB_SLAM3::DeviceORBVocabulary voc;
    if (!voc.loadFromTextFile(argv[11])) return 8;
    std::vector<cv::Mat> vCurrentDesc(nk);
    for (int i = 0; i < nk; ++i) vCurrentDesc[i] = cv::Mat(1, 32, CV_8U, desc.data + (size_t)i * 32, 32);
    std::map<unsigned, double> bow;
    std::map<unsigned, std::vector<unsigned> > featvec;
    voc.transform(vCurrentDesc, bow, featvec, 2);
    const int nw = (int)bow.size(), nn = (int)featvec.size();
    wr(out, &nw, 1);
    for (auto& kv : bow) { wr(out, &kv.first, 1); wr(out, &kv.second, 1); }
    wr(out, &nn, 1);
    for (auto& kv : featvec) { const int c = (int)kv.second.size(); wr(out, &kv.first, 1); wr(out, &c, 1); wr(out, kv.second.data(), c); }
  }
  // --- as Tracking::Relocalization (Tracking.cc:3723-3752): matcher2.SearchByProjection(mCurrentFrame, vpCandidateKFs[i], sFound, 10, 100)
  if (argc > 12) {
    f = fopen(argv[12], "rb");
    int n1 = 0, n2 = 0, orb_dist = 0, ori = 0;
    float th = 0, hdr[6 + 7 + 4 + 8 + 1];
    if (!f || !rd(f, &n1, 1) || !rd(f, &n2, 1) || !rd(f, &th, 1) || !rd(f, &orb_dist, 1) || !rd(f, &ori, 1) || !rd(f, hdr, 26)) return 9;
    std::vector<unsigned char> has(n1), bad(n1), found(n1), d1((size_t)n1 * 32), d2((size_t)n2 * 32), occ(n2);
    std::vector<float> pos((size_t)n1 * 3), mind(n1), maxd(n1), ang1(n1), xy2((size_t)n2 * 2), ang2(n2);
    std::vector<int> o2(n2);
    rd(f, has.data(), n1); rd(f, bad.data(), n1); rd(f, found.data(), n1); rd(f, pos.data(), (size_t)n1 * 3);
    rd(f, d1.data(), (size_t)n1 * 32); rd(f, mind.data(), n1); rd(f, maxd.data(), n1); rd(f, ang1.data(), n1);
    rd(f, xy2.data(), (size_t)n2 * 2); rd(f, o2.data(), n2); rd(f, ang2.data(), n2); rd(f, d2.data(), (size_t)n2 * 32); rd(f, occ.data(), n2);
    fclose(f);
    TestFrame::mnMinX = hdr[0]; TestFrame::mnMinY = hdr[1]; TestFrame::mnMaxX = hdr[2]; TestFrame::mnMaxY = hdr[3];
    TestFrame::mfGridElementWidthInv = hdr[4]; TestFrame::mfGridElementHeightInv = hdr[5];
    Camera pcam; for (int k = 0; k < 4; ++k) pcam.p[k] = hdr[13 + k];
    std::vector<TrackedPoint> pts(n1);
    TrackedPoint before;
    RelocKeyFrame kf;
    std::set<TrackedPoint*> sFound;
    kf.mvKeysUn.resize(n1); kf.mps.assign(n1, nullptr);
    for (int i = 0; i < n1; ++i) {
      kf.mvKeysUn[i].angle = ang1[i];
      if (!has[i]) continue;
      TrackedPoint& p = pts[i];
      p.pos = V3{{pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]}};
      p.desc.create(1, 32, CV_8U); memcpy(p.desc.data, &d1[(size_t)i * 32], 32);
      p.bad = bad[i] != 0; p.mfMinDistance = mind[i]; p.mfMaxDistance = maxd[i];
      kf.mps[i] = &p;
      if (found[i]) sFound.insert(&p);
    }
    TestFrame cur;
    cur.N = n2; cur.mpCamera = &pcam;
    cur.mvKeysUn.resize(n2);
    for (int i = 0; i < n2; ++i) { cur.mvKeysUn[i].pt.x = xy2[2 * i]; cur.mvKeysUn[i].pt.y = xy2[2 * i + 1]; cur.mvKeysUn[i].octave = o2[i]; cur.mvKeysUn[i].angle = ang2[i]; }
    cur.mDescriptors.create(n2, 32, CV_8U); memcpy(cur.mDescriptors.data, d2.data(), (size_t)n2 * 32);
    cur.mvScaleFactors.assign(hdr + 17, hdr + 25);
    cur.mnScaleLevels = 8; cur.mfLogScaleFactor = hdr[25];
    cur.mvpMapPoints.assign(n2, nullptr);
    for (int i = 0; i < n2; ++i) if (occ[i]) cur.mvpMapPoints[i] = &before;
    cur.pose.q = Quat{hdr[6], hdr[7], hdr[8], hdr[9]}; cur.pose.t = V3{{hdr[10], hdr[11], hdr[12]}};
    ORB_SLAM3::ORBmatcher matcher2(0.9, ori != 0);
    const int nreloc = matcher2.SearchByProjection(cur, &kf, sFound, th, orb_dist);
    wr(out, &nreloc, 1); wr(out, &n2, 1);
    for (int i = 0; i < n2; ++i) {
      TrackedPoint* p = cur.mvpMapPoints[i];
      const int idx = (p && p != &before) ? (int)(p - pts.data()) : -1;
      wr(out, &idx, 1);
    }
  }
  {
    // --- as LoopClosing::DetectCommonRegionsFromBoW (LoopClosing.cc): ORBmatcher(0.9 / 0.75, true).SearchByBoW(pKF1, pKF2, vpMatches12)
    std::vector<MapPoint> own1(kf1.N), own2(kf2.N);
    for (int i = 0; i < kf1.N; ++i) if (kf1.mvpMapPoints[i]) kf1.mvpMapPoints[i] = &own1[i];
    for (int i = 0; i < kf2.N; ++i) if (kf2.mvpMapPoints[i]) kf2.mvpMapPoints[i] = &own2[i];
    ORB_SLAM3::ORBmatcher loop(0.75, true);
    std::vector<MapPoint*> vpMatches12;
    const int nkk = loop.SearchByBoW(&kf1, &kf2, vpMatches12);
    const int n1k = (int)vpMatches12.size();
    wr(out, &nkk, 1); wr(out, &n1k, 1);
    for (int i = 0; i < n1k; ++i) { const int idx = vpMatches12[i] ? (int)(vpMatches12[i] - own2.data()) : -1; wr(out, &idx, 1); }
  }
  // --- as LocalMapping::SearchInNeighbors (LocalMapping.cc:737-879): matcher.Fuse(pKFi, vpMapPointMatches)
  if (argc > 13) {
    f = fopen(argv[13], "rb");
    int n1 = 0, n2 = 0;
    float th = 0, hdr[6 + 7 + 3 + 4 + 1 + 8 + 8 + 1];
    if (!f || !rd(f, &n1, 1) || !rd(f, &n2, 1) || !rd(f, &th, 1) || !rd(f, hdr, 38)) return 10;
    std::vector<unsigned char> has(n1), bad(n1), inkf(n1), d1((size_t)n1 * 32), d2((size_t)n2 * 32), state(n2);
    std::vector<float> pos((size_t)n1 * 3), nor((size_t)n1 * 3), mind(n1), maxd(n1), xy2((size_t)n2 * 2), ur2(n2);
    std::vector<int> o2(n2);
    rd(f, has.data(), n1); rd(f, bad.data(), n1); rd(f, inkf.data(), n1); rd(f, pos.data(), (size_t)n1 * 3); rd(f, nor.data(), (size_t)n1 * 3);
    rd(f, d1.data(), (size_t)n1 * 32); rd(f, mind.data(), n1); rd(f, maxd.data(), n1);
    rd(f, xy2.data(), (size_t)n2 * 2); rd(f, o2.data(), n2); rd(f, ur2.data(), n2); rd(f, d2.data(), (size_t)n2 * 32); rd(f, state.data(), n2);
    fclose(f);
    Camera pcam; for (int k = 0; k < 4; ++k) pcam.p[k] = hdr[16 + k];
    FuseKF kf;
    kf.N = n2; kf.mpCamera = &pcam;
    kf.mnMinX = hdr[0]; kf.mnMinY = hdr[1]; kf.mnMaxX = hdr[2]; kf.mnMaxY = hdr[3]; kf.mfGridElementWidthInv = hdr[4]; kf.mfGridElementHeightInv = hdr[5];
    kf.pose.q = Quat{hdr[6], hdr[7], hdr[8], hdr[9]}; kf.pose.t = V3{{hdr[10], hdr[11], hdr[12]}};
    kf.Ow = V3{{hdr[13], hdr[14], hdr[15]}};
    kf.mbf = hdr[20];
    kf.mvScaleFactors.assign(hdr + 21, hdr + 29); kf.mvInvLevelSigma2.assign(hdr + 29, hdr + 37);
    kf.mnScaleLevels = 8; kf.mfLogScaleFactor = hdr[37];
    kf.mvKeysUn.resize(n2);
    for (int i = 0; i < n2; ++i) { kf.mvKeysUn[i].pt.x = xy2[2 * i]; kf.mvKeysUn[i].pt.y = xy2[2 * i + 1]; kf.mvKeysUn[i].octave = o2[i]; }
    kf.mvuRight = ur2;
    kf.mDescriptors.create(n2, 32, CV_8U); memcpy(kf.mDescriptors.data, d2.data(), (size_t)n2 * 32);
    std::vector<FusePoint> owned(n2), pts(n1);
    kf.mps.assign(n2, nullptr);
    for (int i = 0; i < n2; ++i)
      if (state[i]) { owned[i].nObs = state[i] == 1 ? 9 : 1; owned[i].bad = state[i] == 3; owned[i].in_kf = true; kf.mps[i] = &owned[i]; }
    std::vector<FusePoint*> vp(n1, nullptr);
    for (int i = 0; i < n1; ++i) {
      if (!has[i]) continue;
      FusePoint& p = pts[i];
      p.pos = V3{{pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]}}; p.normal = V3{{nor[3 * i], nor[3 * i + 1], nor[3 * i + 2]}};
      p.desc.create(1, 32, CV_8U); memcpy(p.desc.data, &d1[(size_t)i * 32], 32);
      p.bad = bad[i] != 0; p.in_kf = inkf[i] != 0; p.mfMinDistance = mind[i]; p.mfMaxDistance = maxd[i];
      vp[i] = &p;
    }
    ORB_SLAM3::ORBmatcher fuser(0.6, true);
    const int nFused = fuser.Fuse(&kf, vp, th);
    const int nq = (int)kf.queried.size();
    wr(out, &nFused, 1); wr(out, &nq, 1); wr(out, kf.queried.data(), nq);
  }
  {
    // --- as MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:329-403) on 25 observed descriptors
    std::vector<cv::Mat> vDescriptors;
    for (int i = 0; i < 25 && i < kf1.N; ++i) vDescriptors.push_back(kf1.mDescriptors.row(i * 3 % kf1.N));
    ORB_SLAM3::ORBmatcher any(0.6, true);
    const int bestIdx = any.DistinctiveDescriptor(vDescriptors);
    wr(out, &bestIdx, 1);
  }
  // --- as LoopClosing: matcher.SearchBySim3(mpCurrentKF, pKF, vpMapPointMatches, gScm, 7.5) and matcher.Fuse(pKFi, Scw, vpPoints, 4, vpReplacePoints)
  if (argc > 14) {
    f = fopen(argv[14], "rb");
    int n = 0;
    float th_sim = 0, th_fuse = 0, hdr[4 + 6 + 8 + 1];
    if (!f || !rd(f, &n, 1) || !rd(f, &th_sim, 1) || !rd(f, &th_fuse, 1) || !rd(f, hdr, 19)) return 11;
    SimKF kf[2];
    std::vector<SimPoint> pts[2];
    for (int s2 = 0; s2 < 2; ++s2) {
      std::vector<float> xy(2 * (size_t)n), pos(3 * (size_t)n), nor(3 * (size_t)n), mind(n), maxd(n);
      std::vector<int> oct(n);
      std::vector<unsigned char> d((size_t)n * 32), st(n), md((size_t)n * 32);
      rd(f, xy.data(), 2 * (size_t)n); rd(f, oct.data(), n); rd(f, d.data(), (size_t)n * 32); rd(f, st.data(), n); rd(f, pos.data(), 3 * (size_t)n);
      rd(f, nor.data(), 3 * (size_t)n); rd(f, md.data(), (size_t)n * 32); rd(f, mind.data(), n); rd(f, maxd.data(), n);
      SimKF& k = kf[s2];
      k.N = n; k.fx = hdr[0]; k.fy = hdr[1]; k.cx = hdr[2]; k.cy = hdr[3];
      k.mnMinX = hdr[4]; k.mnMinY = hdr[5]; k.mnMaxX = hdr[6]; k.mnMaxY = hdr[7]; k.mfGridElementWidthInv = hdr[8]; k.mfGridElementHeightInv = hdr[9];
      k.mvScaleFactors.assign(hdr + 10, hdr + 18); k.mnScaleLevels = 8; k.mfLogScaleFactor = hdr[18];
      k.mvKeysUn.resize(n); k.mps.assign(n, nullptr);
      k.mDescriptors.create(n, 32, CV_8U); memcpy(k.mDescriptors.data, d.data(), (size_t)n * 32);
      pts[s2].assign(n, SimPoint());
      for (int i = 0; i < n; ++i) {
        k.mvKeysUn[i].pt.x = xy[2 * i]; k.mvKeysUn[i].pt.y = xy[2 * i + 1]; k.mvKeysUn[i].octave = oct[i];
        if (!st[i]) continue;
        SimPoint& p = pts[s2][i];
        p.bad = st[i] == 2; p.pos = V3{{pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]}}; p.normal = V3{{nor[3 * i], nor[3 * i + 1], nor[3 * i + 2]}};
        p.desc.create(1, 32, CV_8U); memcpy(p.desc.data, &md[(size_t)i * 32], 32);
        p.mfMinDistance = mind[i]; p.mfMaxDistance = maxd[i]; p.home = &k; p.home_idx = i;
        k.mps[i] = &p;
      }
    }
    std::vector<int> prior(n);
    rd(f, prior.data(), n);
    fclose(f);
    std::vector<SimPoint*> vpMatches12(n, nullptr);
    for (int i = 0; i < n; ++i) if (prior[i] >= 0) vpMatches12[i] = &pts[1][prior[i]];
    TestSim3 S12;
    ORB_SLAM3::ORBmatcher loop2(0.75, true);
    const int nFound = loop2.SearchBySim3(&kf[0], &kf[1], vpMatches12, S12, th_sim);
    wr(out, &nFound, 1); wr(out, &n, 1);
    for (int i = 0; i < n; ++i) { const int idx = vpMatches12[i] ? (int)(vpMatches12[i] - pts[1].data()) : -1; wr(out, &idx, 1); }
    // Fuse: the good and bad points of KF1 as candidates, projected into KF2
    std::vector<SimPoint*> vpPoints, vpReplace;
    for (int i = 0; i < n; ++i) if (kf[0].mps[i]) vpPoints.push_back(kf[0].mps[i]);
    vpReplace.assign(vpPoints.size(), nullptr);
    kf[1].queried.clear();
    TestSim3 Scw;
    const int nFusedS = loop2.Fuse(&kf[1], Scw, vpPoints, th_fuse, vpReplace);
    const int nq = (int)kf[1].queried.size();
    wr(out, &nFusedS, 1); wr(out, &nq, 1); wr(out, kf[1].queried.data(), nq);
    // --- as LoopClosing::FindMatchesByProjection: SearchByProjection(pKF, Scw, vpPoints, [vpPointsKFs,] vpMatched, [vpMatchedKF,] th, ratio)
    for (int form = 0; form < 2; ++form) {
      SimKF fresh = kf[1];                       // Fuse above added points to kf[1]: start from the features alone
      fresh.mps.assign(n, nullptr);
      std::vector<SimPoint*> vpMatched(n, nullptr);
      std::vector<SimKF*> vpPointsKFs(vpPoints.size(), &kf[0]), vpMatchedKF(n, nullptr);
      for (int i = 0; i < n; i += 9) vpMatched[i] = &pts[1][i];   // matched before the call
      TestSim3 Sid;
      const int nms = form == 0 ? loop2.SearchByProjection(&fresh, Sid, vpPoints, vpMatched, 8, 1.5f)
                                : loop2.SearchByProjection(&fresh, Sid, vpPoints, vpPointsKFs, vpMatched, vpMatchedKF, 30, 1.0f);
      wr(out, &nms, 1);
      for (int i = 0; i < n; ++i) {
        SimPoint* p = vpMatched[i];
        int idx = -1;
        if (p && !(i % 9 == 0 && p == &pts[1][i])) idx = (int)(p - pts[0].data());
        if (form == 1 && idx >= 0 && vpMatchedKF[i] != &kf[0]) idx = -3;
        wr(out, &idx, 1);
      }
    }
  }
  // --- as Tracking::MonocularInitialization (Tracking.cc:2525-2526): ORBmatcher(0.9, true).SearchForInitialization(F1, F2, prev, m12, 100)
  if (argc > 15) {
    f = fopen(argv[15], "rb");
    int n1 = 0, n2 = 0, window = 0;
    float hdr[6];
    if (!f || !rd(f, &n1, 1) || !rd(f, &n2, 1) || !rd(f, &window, 1) || !rd(f, hdr, 6)) return 8;
    std::vector<int> o1(n1), o2(n2);
    std::vector<float> a1(n1), prev((size_t)n1 * 2), xy2((size_t)n2 * 2), a2(n2);
    std::vector<unsigned char> d1((size_t)n1 * 32), d2((size_t)n2 * 32);
    rd(f, o1.data(), n1); rd(f, a1.data(), n1); rd(f, d1.data(), (size_t)n1 * 32); rd(f, prev.data(), (size_t)n1 * 2);
    rd(f, xy2.data(), (size_t)n2 * 2); rd(f, o2.data(), n2); rd(f, a2.data(), n2); rd(f, d2.data(), (size_t)n2 * 32);
    fclose(f);
    TestFrame::mnMinX = hdr[0]; TestFrame::mnMinY = hdr[1]; TestFrame::mnMaxX = hdr[2]; TestFrame::mnMaxY = hdr[3];
    TestFrame::mfGridElementWidthInv = hdr[4]; TestFrame::mfGridElementHeightInv = hdr[5];
    TestFrame F1, F2;
    F1.N = n1; F1.mvKeysUn.resize(n1);
    for (int i = 0; i < n1; ++i) { F1.mvKeysUn[i].pt.x = prev[2 * i]; F1.mvKeysUn[i].pt.y = prev[2 * i + 1]; F1.mvKeysUn[i].octave = o1[i]; F1.mvKeysUn[i].angle = a1[i]; }
    F1.mDescriptors.create(n1, 32, CV_8U); memcpy(F1.mDescriptors.data, d1.data(), (size_t)n1 * 32);
    F2.N = n2; F2.mvKeysUn.resize(n2);
    for (int i = 0; i < n2; ++i) { F2.mvKeysUn[i].pt.x = xy2[2 * i]; F2.mvKeysUn[i].pt.y = xy2[2 * i + 1]; F2.mvKeysUn[i].octave = o2[i]; F2.mvKeysUn[i].angle = a2[i]; }
    F2.mDescriptors.create(n2, 32, CV_8U); memcpy(F2.mDescriptors.data, d2.data(), (size_t)n2 * 32);
    std::vector<cv::Point2f> vbPrevMatched(n1);
    for (int i = 0; i < n1; ++i) { vbPrevMatched[i].x = prev[2 * i]; vbPrevMatched[i].y = prev[2 * i + 1]; }
    std::vector<int> vnMatches12;
    ORB_SLAM3::ORBmatcher init(0.9, true);
    const int ninit = init.SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, window);
    wr(out, &ninit, 1); wr(out, &n1, 1);
    wr(out, vnMatches12.data(), n1);
    for (int i = 0; i < n1; ++i) { wr(out, &vbPrevMatched[i].x, 1); wr(out, &vbPrevMatched[i].y, 1); }
  }
  // --- the optional latency hooks: Begin + PrefetchPointcloud right before the ordinary calls of
  //     Frame's RGB-L constructor must change nothing in what those calls return
  {
    ORB_SLAM3::ORBextractor ex2(2000, 1.2f, 8, 12, 7);
    ORB_SLAM3::DepthModule depth2(argv[1], 6);
    depth2.downloadDenseMaps = false;
    std::vector<cv::KeyPoint> keys2;
    cv::Mat desc2;
    int same = 1;
    for (int rep = 0; rep < 2; ++rep) {   // the second time round every handle exists already
      same &= ex2.Begin(im, vLapping) ? 1 : 0;
      depth2.PrefetchPointcloud(pcd, w, h);
      const int mono2 = ex2(im, cv::Mat(), keys2, desc2, vLapping);
      depth2.CalculateDepthFromPcd(keys2, keys2, pcd, w, h);
      same &= mono2 == mono && keys2.size() == keys.size() && memcmp(keys2.data(), keys.data(), sizeof(cv::KeyPoint) * keys.size()) == 0;
      same &= desc2.rows == desc.rows && memcmp(desc2.data, desc.data, (size_t)desc.rows * 32) == 0;
      same &= depth2.mvDepth.size() == depth.mvDepth.size() && memcmp(depth2.mvDepth.data(), depth.mvDepth.data(), sizeof(float) * depth.mvDepth.size()) == 0;
      same &= memcmp(depth2.mvuRight.data(), depth.mvuRight.data(), sizeof(float) * depth.mvuRight.size()) == 0;
    }
    wr(out, &same, 1);
  }
  fclose(out);
  printf("shim_test ok: %d keypoints, %d depths, %d triangulation matches\n", nk, nd, nm);
  return 0;
}
