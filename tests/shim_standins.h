// shim_standins.h - stand-ins for the Sophus / camera / KeyFrame members the drop-in ORBmatcher::SearchForTriangulation
// touches, and the loader of the key-frame files tests/shim_driver.py writes.  Shared by shim_test.cpp and
// shim_threads_test.cpp.  TEST INFRASTRUCTURE.
#pragma once
#include <math.h>
#include <stdio.h>

#include <map>
#include <vector>

#include "ORBextractor.h"

// ---- minimal stand-ins for the Sophus / camera / KeyFrame members SearchForTriangulation touches
struct V3 { float v[3]; float operator()(int i) const { return v[i]; } };
struct V2 { float v[2]; float operator()(int i) const { return v[i]; } };
struct M3 { float m[9]; float operator()(int i, int j) const { return m[3 * i + j]; } };
struct SE3 {
  M3 R; V3 t;
  M3 rotationMatrix() const { return R; }
  V3 translation() const { return t; }
  SE3 operator*(const SE3& o) const {
    SE3 r;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) r.R.m[3 * i + j] = R.m[3 * i] * o.R.m[j] + R.m[3 * i + 1] * o.R.m[3 + j] + R.m[3 * i + 2] * o.R.m[6 + j];
      r.t.v[i] = R.m[3 * i] * o.t.v[0] + R.m[3 * i + 1] * o.t.v[1] + R.m[3 * i + 2] * o.t.v[2] + t.v[i];
    }
    return r;
  }
  V3 operator*(const V3& p) const {
    V3 r;
    for (int i = 0; i < 3; ++i) r.v[i] = R.m[3 * i] * p.v[0] + R.m[3 * i + 1] * p.v[1] + R.m[3 * i + 2] * p.v[2] + t.v[i];
    return r;
  }
};
struct Camera {
  float p[4];
  float getParameter(int i) const { return p[i]; }
  V2 project(const V3& c) const { V2 r; r.v[0] = p[0] * c.v[0] / c.v[2] + p[2]; r.v[1] = p[1] * c.v[1] / c.v[2] + p[3]; return r; }
};
struct MapPoint { bool isBad() { return false; } };
struct KeyFrame {
  int N = 0, NLeft = -1, Nleft = -1;  // Nleft / mvKeys: the members SearchByBoW reads when a KeyFrame stands in for a Frame
  std::vector<cv::KeyPoint> mvKeys, mvKeysRight;
  std::vector<MapPoint*> GetMapPointMatches() { return mvpMapPoints; }
  Camera* mpCamera = nullptr; Camera* mpCamera2 = nullptr;
  std::map<unsigned, std::vector<unsigned> > mFeatVec;
  std::vector<cv::KeyPoint> mvKeysUn;
  std::vector<float> mvuRight, mvScaleFactors, mvLevelSigma2;
  std::vector<MapPoint*> mvpMapPoints;
  cv::Mat mDescriptors;
  SE3 Tcw, Twc;
  const rgbl_device_frame* mpDeviceFrame = nullptr;   // the optional resident copy the drop-in ORBmatcher looks for
  MapPoint* GetMapPoint(size_t i) { return mvpMapPoints[i]; }
  SE3 GetPose() { return Tcw; }
  SE3 GetPoseInverse() { return Twc; }
  V3 GetCameraCenter() { return Twc.t; }
};

template <class T> static bool rd(FILE* f, T* p, size_t n) { return fread(p, sizeof(T), n, f) == n; }
template <class T> static void wr(FILE* f, const T* p, size_t n) { fwrite(p, sizeof(T), n, f); }

static bool load_kf(FILE* f, KeyFrame& kf, Camera* cam, MapPoint* some) {
  int n, nn;
  if (!rd(f, &n, 1)) return false;
  kf.N = n; kf.mpCamera = cam;
  kf.mDescriptors.create(n, 32, CV_8U);
  rd(f, kf.mDescriptors.data, (size_t)n * 32);
  kf.mvKeysUn.resize(n); kf.mvuRight.resize(n); kf.mvpMapPoints.resize(n);
  std::vector<float> xy(2 * n), ang(n); std::vector<int> oct(n); std::vector<unsigned char> mp(n);
  rd(f, xy.data(), 2 * n); rd(f, oct.data(), n); rd(f, ang.data(), n); rd(f, kf.mvuRight.data(), n); rd(f, mp.data(), n);
  for (int i = 0; i < n; ++i) {
    kf.mvKeysUn[i].pt.x = xy[2 * i]; kf.mvKeysUn[i].pt.y = xy[2 * i + 1]; kf.mvKeysUn[i].octave = oct[i]; kf.mvKeysUn[i].angle = ang[i];
    kf.mvpMapPoints[i] = mp[i] ? some : nullptr;
  }
  rd(f, &nn, 1);
  std::vector<int> id(nn), off(nn + 1);
  rd(f, id.data(), nn); rd(f, off.data(), nn + 1);
  std::vector<int> feat(off[nn]);
  rd(f, feat.data(), off[nn]);
  for (int k = 0; k < nn; ++k) kf.mFeatVec[(unsigned)id[k]] = std::vector<unsigned>(feat.begin() + off[k], feat.begin() + off[k + 1]);
  kf.mvScaleFactors.resize(8); kf.mvLevelSigma2.resize(8);
  rd(f, kf.mvScaleFactors.data(), 8); rd(f, kf.mvLevelSigma2.data(), 8);
  rd(f, kf.Tcw.R.m, 9); rd(f, kf.Tcw.t.v, 3); rd(f, kf.Twc.R.m, 9); rd(f, kf.Twc.t.v, 3);
  return true;
}
