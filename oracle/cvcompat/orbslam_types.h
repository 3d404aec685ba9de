// Stand-ins for ORB_SLAM3::MapPoint / KeyFrame / Frame / GeometricCamera, force-included (-include) in front of the
// reference's OWN src/ORBmatcher.cc so that the file compiles unmodified where it lies (TEST INFRASTRUCTURE, see
// oracle/Makefile target `ref`).  The real headers of these classes drag in Eigen, Sophus, DBoW2, boost::serialization
// and the rest of the SLAM system; defining their include guards here turns them into empty files, and the plain
// data classes below provide exactly the members ORBmatcher.cc touches.  What this pins to the reference source:
// every matcher's control flow - candidate order, masks, thresholds, ratio and orientation tests, tie rules, the
// rotation histogram - for the functions the tests drive (SearchForTriangulation, SearchByProjection, DescriptorDistance).
#pragma once
#define MAPPOINT_H
#define KEYFRAME_H
#define FRAME_H

#include <map>
#include <set>
#include <tuple>
#include <vector>

#include <opencv2/core/core.hpp>

#include "ORBextractor.h"  // the reference's own header (needs OpenCV types only)
#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"
#include "sophus/sim3.hpp"

// the reference's headers leak these into every translation unit (ORBmatcher.h names vector / pair unqualified)
using namespace std;
#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW

#define FRAME_GRID_ROWS 48
#define FRAME_GRID_COLS 64

namespace ORB_SLAM3 {

class KeyFrame;
class Frame;

class GeometricCamera {
 public:
  float fx = 0, fy = 0, cx = 0, cy = 0;
  // Pinhole::project(const Eigen::Vector3f&) (src/CameraModels/Pinhole.cpp:48-55)
  Eigen::Vector2f project(const Eigen::Vector3f& v) const { return Eigen::Vector2f(fx * v[0] / v[2] + cx, fy * v[1] / v[2] + cy); }
  float getParameter(int i) const { return i == 0 ? fx : i == 1 ? fy : i == 2 ? cx : cy; }
  // Pinhole::toK_ (src/CameraModels/Pinhole.cpp:100-104)
  Eigen::Matrix3f toK_() { Eigen::Matrix3f K; K(0, 0) = fx; K(0, 2) = cx; K(1, 1) = fy; K(1, 2) = cy; K(2, 2) = 1.f; return K; }
  // Pinhole::epipolarConstrain (src/CameraModels/Pinhole.cpp:107-129): the reference's own lines, extracted at build time
  // (oracle/Makefile: _ref/pinhole_epipolar.inc) and compiled as a member of this class in ref_matcher_glue.cpp
  bool epipolarConstrain(GeometricCamera* other, const cv::KeyPoint& kp1, const cv::KeyPoint& kp2, const Eigen::Matrix3f& R12,
                         const Eigen::Vector3f& t12, const float sigmaLevel, const float unc);
};

class MapPoint {
 public:
  Eigen::Vector3f mWorldPos, mNormal;
  cv::Mat mDescriptor;
  int nObs = 0;
  bool bad = false;
  float mfMinDistance = 0, mfMaxDistance = 0;
  std::map<KeyFrame*, int> mObservations;
  // tracking scratch written by Frame::isInFrustum, read by SearchByProjection(Frame&, vector<MapPoint*>&, ...)
  float mTrackProjX = 0, mTrackProjY = 0, mTrackDepth = 0, mTrackDepthR = 0, mTrackProjXR = 0, mTrackProjYR = 0;
  bool mbTrackInView = false, mbTrackInViewR = false;
  int mnTrackScaleLevel = 0, mnTrackScaleLevelR = 0;
  float mTrackViewCos = 0, mTrackViewCosR = 0;

  Eigen::Vector3f GetWorldPos() { return mWorldPos; }
  Eigen::Vector3f GetNormal() { return mNormal; }
  cv::Mat GetDescriptor() { return mDescriptor; }
  int Observations() { return nObs; }
  bool isBad() { return bad; }
  float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; }
  float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }
  int PredictScale(const float& currentDist, KeyFrame* pKF);
  int PredictScale(const float& currentDist, Frame* pF);
  bool IsInKeyFrame(KeyFrame* pKF) { return mObservations.count(pKF) != 0; }
  std::tuple<int, int> GetIndexInKeyFrame(KeyFrame* pKF) { return mObservations.count(pKF) ? std::tuple<int, int>(mObservations[pKF], -1) : std::tuple<int, int>(-1, -1); }
  void AddObservation(KeyFrame* pKF, int idx) { mObservations[pKF] = idx; ++nObs; }
  // MapPoint::Replace rewires observations between map objects; the stand-in only records the call (who, with whom)
  static std::vector<std::pair<MapPoint*, MapPoint*>>* replace_log;
  void Replace(MapPoint* pMP) { if (replace_log) replace_log->push_back(std::make_pair(this, pMP)); }
};

// Frame::GetFeaturesInArea / KeyFrame::GetFeaturesInArea (src/Frame.cc:747-813, src/KeyFrame.cc:604-666) over a grid built
// like Frame::AssignFeaturesToGrid (src/Frame.cc:475-506)
struct FeatureGrid {
  float mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0, mfGridElementWidthInv = 0, mfGridElementHeightInv = 0;
  std::vector<std::size_t> cells[FRAME_GRID_COLS][FRAME_GRID_ROWS];
  void Build(const std::vector<cv::KeyPoint>& keysUn);
  std::vector<size_t> Query(const std::vector<cv::KeyPoint>& keysUn, float x, float y, float r, int minLevel, int maxLevel) const;
};

class Frame {
 public:
  int N = 0, Nleft = -1;
  float mb = 0, mbf = 0, mfLogScaleFactor = 0;
  int mnScaleLevels = 0;
  GeometricCamera* mpCamera = nullptr;
  GeometricCamera* mpCamera2 = nullptr;
  std::vector<cv::KeyPoint> mvKeys, mvKeysRight, mvKeysUn;
  std::vector<float> mvuRight, mvScaleFactors;
  std::vector<MapPoint*> mvpMapPoints;
  std::vector<bool> mvbOutlier;
  std::vector<int> mvLeftToRightMatch, mvRightToLeftMatch;
  cv::Mat mDescriptors;
  DBoW2::FeatureVector mFeatVec;
  Sophus::SE3f mTcw, mTrl;
  FeatureGrid grid;
  static float mnMinX, mnMinY, mnMaxX, mnMaxY;  // static members in the reference as well (Frame.h)
  static float mfGridElementWidthInv, mfGridElementHeightInv;
  // what Frame::ComputeStereoMatches (src/Frame.cc:901-1071) touches; its body is the reference's own lines, extracted at
  // build time (oracle/Makefile: _ref/frame_stereo_matches.inc) and compiled in ref_frame_glue.cpp
  std::vector<float> mvDepth, mvInvScaleFactors;
  cv::Mat mDescriptorsRight;
  ORBextractor* mpORBextractorLeft = nullptr;
  ORBextractor* mpORBextractorRight = nullptr;
  void ComputeStereoMatches();

  Sophus::SE3f GetPose() const { return mTcw; }
  Sophus::SE3f GetRelativePoseTrl() const { return mTrl; }
  std::vector<size_t> GetFeaturesInArea(const float& x, const float& y, const float& r, const int minLevel = -1,
                                        const int maxLevel = -1, const bool bRight = false) const {
    return grid.Query(mvKeysUn, x, y, r, minLevel, maxLevel);
  }
};

class KeyFrame {
 public:
  int N = 0, NLeft = -1;
  float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0, mfLogScaleFactor = 0;
  int mnScaleLevels = 0;
  GeometricCamera* mpCamera = nullptr;
  GeometricCamera* mpCamera2 = nullptr;
  std::vector<cv::KeyPoint> mvKeys, mvKeysRight, mvKeysUn;
  std::vector<float> mvuRight, mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
  std::vector<MapPoint*> mvpMapPoints;
  cv::Mat mDescriptors;
  DBoW2::FeatureVector mFeatVec;
  Sophus::SE3f mTcw, mTwc;
  FeatureGrid grid;
  int mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0;  // const members of the reference's KeyFrame (copies of the Frame statics)
  float mfGridElementWidthInv = 0, mfGridElementHeightInv = 0;

  Sophus::SE3f GetPose() { return mTcw; }
  Sophus::SE3f GetPoseInverse() { return mTwc; }
  Sophus::SE3f GetRightPose() { return mTcw; }
  Sophus::SE3f GetRightPoseInverse() { return mTwc; }
  Eigen::Vector3f GetCameraCenter() { return mTwc.translation(); }
  Eigen::Vector3f GetRightCameraCenter() { return mTwc.translation(); }
  MapPoint* GetMapPoint(const size_t& idx) { return mvpMapPoints[idx]; }
  std::vector<MapPoint*> GetMapPointMatches() { return mvpMapPoints; }
  std::set<MapPoint*> GetMapPoints() { std::set<MapPoint*> s; for (MapPoint* p : mvpMapPoints) if (p) s.insert(p); return s; }
  void AddMapPoint(MapPoint* pMP, const size_t& idx) { mvpMapPoints[idx] = pMP; }
  bool IsInImage(const float& x, const float& y) const { return x >= grid.mnMinX && x < grid.mnMaxX && y >= grid.mnMinY && y < grid.mnMaxY; }
  std::vector<size_t> GetFeaturesInArea(const float& x, const float& y, const float& r, const bool bRight = false) const {
    return grid.Query(mvKeysUn, x, y, r, -1, -1);
  }
};

}  // namespace ORB_SLAM3
