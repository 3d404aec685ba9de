// ref_matcher_glue.cpp — bodies of the stand-in classes of cvcompat/orbslam_types.h plus C entry points around the
// reference's OWN ORB_SLAM3::ORBmatcher, compiled from /root/reference/src/ORBmatcher.cc (unmodified, where it lies).
// TEST INFRASTRUCTURE (oracle/Makefile target `ref`).
#include <math.h>
#include <string.h>

#include <chrono>

#include "ORBmatcher.h"  // /root/reference/include (orbslam_types.h is force-included in front of it)
#include "oracle.h"

// Wall time of the last call INTO the reference's own function (not of the stand-in objects built around it): what
// bench.py reports as the CPU side of a matcher call (`cpu_baseline`, kind "reference").
namespace {
double g_last_call_seconds = 0.0;
struct CallTimer {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  ~CallTimer() { g_last_call_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
}  // namespace
extern "C" double ref_last_call_seconds() { return g_last_call_seconds; }

namespace ORB_SLAM3 {

float Frame::mnMinX = 0, Frame::mnMinY = 0, Frame::mnMaxX = 0, Frame::mnMaxY = 0;
float Frame::mfGridElementWidthInv = 0, Frame::mfGridElementHeightInv = 0;

// Pinhole::epipolarConstrain: lines 107-129 of /root/reference/src/CameraModels/Pinhole.cpp, unmodified, extracted by the
// Makefile into _ref/pinhole_epipolar.inc; `Pinhole` is the stand-in camera class here (Eigen / Sophus come from
// cvcompat/sophus/sim3.hpp, so Eigen's product / inverse arithmetic is restated, the function itself is the reference's)
#define Pinhole GeometricCamera
#include "_ref/pinhole_epipolar.inc"
#undef Pinhole

std::vector<std::pair<MapPoint*, MapPoint*>>* MapPoint::replace_log = nullptr;

// MapPoint::PredictScale (src/MapPoint.cc:514-546)
int MapPoint::PredictScale(const float& currentDist, KeyFrame* pKF) {
  const float ratio = mfMaxDistance / currentDist;
  int nScale = ceil(log(ratio) / pKF->mfLogScaleFactor);
  if (nScale < 0) nScale = 0;
  else if (nScale >= pKF->mnScaleLevels) nScale = pKF->mnScaleLevels - 1;
  return nScale;
}
int MapPoint::PredictScale(const float& currentDist, Frame* pF) {
  const float ratio = mfMaxDistance / currentDist;
  int nScale = ceil(log(ratio) / pF->mfLogScaleFactor);
  if (nScale < 0) nScale = 0;
  else if (nScale >= pF->mnScaleLevels) nScale = pF->mnScaleLevels - 1;
  return nScale;
}

// Frame::AssignFeaturesToGrid + PosInGrid (src/Frame.cc:475-506, 815-825)
void FeatureGrid::Build(const std::vector<cv::KeyPoint>& keysUn) {
  for (int i = 0; i < FRAME_GRID_COLS; ++i)
    for (int j = 0; j < FRAME_GRID_ROWS; ++j) cells[i][j].clear();
  for (size_t i = 0; i < keysUn.size(); ++i) {
    const cv::KeyPoint& kp = keysUn[i];
    const int posX = round((kp.pt.x - mnMinX) * mfGridElementWidthInv);
    const int posY = round((kp.pt.y - mnMinY) * mfGridElementHeightInv);
    if (posX < 0 || posX >= FRAME_GRID_COLS || posY < 0 || posY >= FRAME_GRID_ROWS) continue;
    cells[posX][posY].push_back(i);
  }
}

// Frame::GetFeaturesInArea (src/Frame.cc:747-813); KeyFrame::GetFeaturesInArea (src/KeyFrame.cc) is the same walk
// without the level test
std::vector<size_t> FeatureGrid::Query(const std::vector<cv::KeyPoint>& keysUn, float x, float y, float r, int minLevel,
                                       int maxLevel) const {
  std::vector<size_t> vIndices;
  const float factorX = r, factorY = r;
  const int nMinCellX = std::max(0, (int)floor((x - mnMinX - factorX) * mfGridElementWidthInv));
  if (nMinCellX >= FRAME_GRID_COLS) return vIndices;
  const int nMaxCellX = std::min((int)FRAME_GRID_COLS - 1, (int)ceil((x - mnMinX + factorX) * mfGridElementWidthInv));
  if (nMaxCellX < 0) return vIndices;
  const int nMinCellY = std::max(0, (int)floor((y - mnMinY - factorY) * mfGridElementHeightInv));
  if (nMinCellY >= FRAME_GRID_ROWS) return vIndices;
  const int nMaxCellY = std::min((int)FRAME_GRID_ROWS - 1, (int)ceil((y - mnMinY + factorY) * mfGridElementHeightInv));
  if (nMaxCellY < 0) return vIndices;
  const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
  for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
    for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
      const std::vector<size_t>& vCell = cells[ix][iy];
      for (size_t j = 0, jend = vCell.size(); j < jend; j++) {
        const cv::KeyPoint& kpUn = keysUn[vCell[j]];
        if (bCheckLevels) {
          if (kpUn.octave < minLevel) continue;
          if (maxLevel >= 0)
            if (kpUn.octave > maxLevel) continue;
        }
        const float distx = kpUn.pt.x - x, disty = kpUn.pt.y - y;
        if (fabs(distx) < factorX && fabs(disty) < factorY) vIndices.push_back(vCell[j]);
      }
    }
  return vIndices;
}

}  // namespace ORB_SLAM3

using namespace ORB_SLAM3;

namespace {
struct KfArrays {  // one key-frame as flat arrays (the layout of orc_tri_input / rgbl_keyframe_view)
  int n;
  const uint8_t* desc;
  const float* kp_xy;
  const int* kp_octave;
  const float* kp_angle;
  const float* uright;
  const uint8_t* has_mp;
  int nnodes;
  const int *node_id, *node_off, *node_feat;
};

void fill_keyframe(KeyFrame& kf, const KfArrays& a, GeometricCamera* cam, MapPoint* some, const float* scale_factors,
                   const float* level_sigma2, int n_levels, const float q[4], const float t[3]) {
  kf.N = a.n;
  kf.mpCamera = cam;
  kf.fx = cam->fx; kf.fy = cam->fy; kf.cx = cam->cx; kf.cy = cam->cy;
  kf.mDescriptors = cv::Mat(a.n, 32, CV_8U);
  if (a.n) memcpy(kf.mDescriptors.data, a.desc, (size_t)a.n * 32);
  kf.mvKeysUn.resize(a.n);
  kf.mvuRight.assign(a.uright, a.uright + a.n);
  kf.mvpMapPoints.resize(a.n);
  for (int i = 0; i < a.n; ++i) {
    kf.mvKeysUn[i].pt.x = a.kp_xy[2 * i];
    kf.mvKeysUn[i].pt.y = a.kp_xy[2 * i + 1];
    kf.mvKeysUn[i].octave = a.kp_octave[i];
    kf.mvKeysUn[i].angle = a.kp_angle[i];
    kf.mvpMapPoints[i] = a.has_mp[i] ? some : nullptr;
  }
  for (int k = 0; k < a.nnodes; ++k)
    kf.mFeatVec[(unsigned)a.node_id[k]] = std::vector<unsigned>(a.node_feat + a.node_off[k], a.node_feat + a.node_off[k + 1]);
  kf.mvScaleFactors.assign(scale_factors, scale_factors + n_levels);
  kf.mvLevelSigma2.assign(level_sigma2, level_sigma2 + n_levels);
  kf.mTcw = Sophus::SE3f(Eigen::Quaternionf(q[3], q[0], q[1], q[2]), Eigen::Vector3f(t[0], t[1], t[2]));
  kf.mTwc = kf.mTcw.inverse();
}
}  // namespace

extern "C" {

int ref_descriptor_distance(const uint8_t* a, const uint8_t* b) {
  cv::Mat ma(1, 32, CV_8U), mb(1, 32, CV_8U);
  memcpy(ma.data, a, 32);
  memcpy(mb.data, b, 32);
  return ORBmatcher::DescriptorDistance(ma, mb);
}

// ORBmatcher(nnratio, checkOri).SearchForTriangulation(pKF1, pKF2, vMatchedPairs, bOnlyStereo, bCoarse).
// Poses as unit quaternion (x, y, z, w) + translation of Tcw.  Besides the pairs it returns what the function derives
// from the poses with the stand-in SE3 arithmetic (R12, t12, the epipole), so that the restated oracle can be run on
// identical inputs.
int ref_search_triangulation(const KfArrays* a1, const KfArrays* a2, const float K[4], const float* scale_factors,
                             const float* level_sigma2, int n_levels, const float q1[4], const float t1[3], const float q2[4],
                             const float t2[3], int only_stereo, int coarse, int check_orientation, int* matches12,
                             float out_R12[9], float out_t12[3], float out_ep[2]) {
  GeometricCamera cam;
  cam.fx = K[0]; cam.fy = K[1]; cam.cx = K[2]; cam.cy = K[3];
  MapPoint some;
  KeyFrame kf1, kf2;
  fill_keyframe(kf1, *a1, &cam, &some, scale_factors, level_sigma2, n_levels, q1, t1);
  fill_keyframe(kf2, *a2, &cam, &some, scale_factors, level_sigma2, n_levels, q2, t2);
  ORBmatcher matcher(0.6f, check_orientation != 0);
  std::vector<std::pair<size_t, size_t> > pairs;
  int nm;
  { CallTimer timed; nm = matcher.SearchForTriangulation(&kf1, &kf2, pairs, only_stereo != 0, coarse != 0); }
  for (int i = 0; i < a1->n; ++i) matches12[i] = -1;
  for (const auto& pr : pairs) matches12[pr.first] = (int)pr.second;
  // the same expressions as ORBmatcher.cc:914-931
  const Sophus::SE3f T12 = kf1.GetPose() * kf2.GetPoseInverse();
  const Eigen::Matrix3f R12 = T12.rotationMatrix();
  memcpy(out_R12, R12.m, sizeof(float) * 9);
  memcpy(out_t12, T12.translation().v, sizeof(float) * 3);
  const Eigen::Vector2f ep = cam.project(kf2.GetPose() * kf1.GetCameraCenter());
  out_ep[0] = ep(0); out_ep[1] = ep(1);
  return nm;
}

}  // extern "C"

// ORBmatcher(0.9, checkOri).SearchByProjection(CurrentFrame, LastFrame, th, bMono) (ORBmatcher.cc:1676-1887), single camera.
// The flat layout is orc_projection_input's; one MapPoint object per valid LastFrame feature.
extern "C" int ref_search_by_projection(const orc_projection_input* in, int* match2) {
  GeometricCamera cam;
  cam.fx = in->K[0]; cam.fy = in->K[1]; cam.cx = in->K[2]; cam.cy = in->K[3];
  Frame last, cur;
  std::vector<MapPoint> points(in->n1);
  last.N = in->n1;
  last.mvKeys.resize(in->n1); last.mvKeysUn.resize(in->n1);
  last.mvpMapPoints.assign(in->n1, nullptr);
  last.mvbOutlier.assign(in->n1, false);
  for (int i = 0; i < in->n1; ++i) {
    last.mvKeys[i].octave = last.mvKeysUn[i].octave = in->octave1[i];
    last.mvKeys[i].angle = last.mvKeysUn[i].angle = in->angle1[i];
    if (!in->valid1[i]) continue;
    MapPoint& mp = points[i];
    mp.mWorldPos = Eigen::Vector3f(in->world_pos1[3 * i], in->world_pos1[3 * i + 1], in->world_pos1[3 * i + 2]);
    mp.mDescriptor = cv::Mat(1, 32, CV_8U);
    memcpy(mp.mDescriptor.data, in->mp_desc1 + 32 * (size_t)i, 32);
    mp.nObs = in->mp_observed1[i] ? 3 : 0;
    last.mvpMapPoints[i] = &mp;
  }
  last.mTcw = Sophus::SE3f(Eigen::Quaternionf(in->Tlw_q[3], in->Tlw_q[0], in->Tlw_q[1], in->Tlw_q[2]),
                           Eigen::Vector3f(in->Tlw_t[0], in->Tlw_t[1], in->Tlw_t[2]));
  cur.N = in->n2;
  cur.mpCamera = &cam;
  cur.mb = in->mb; cur.mbf = in->mbf;
  cur.mvKeysUn.resize(in->n2);
  for (int i = 0; i < in->n2; ++i) {
    cur.mvKeysUn[i].pt.x = in->kp2_xy[2 * i]; cur.mvKeysUn[i].pt.y = in->kp2_xy[2 * i + 1];
    cur.mvKeysUn[i].octave = in->kp2_octave[i]; cur.mvKeysUn[i].angle = in->kp2_angle[i];
  }
  cur.mvKeys = cur.mvKeysUn;
  cur.mvuRight.assign(in->uright2, in->uright2 + in->n2);
  cur.mDescriptors = cv::Mat(in->n2, 32, CV_8U);
  if (in->n2) memcpy(cur.mDescriptors.data, in->desc2, (size_t)in->n2 * 32);
  cur.mvScaleFactors.assign(in->scale_factors, in->scale_factors + in->n_levels);
  cur.mvpMapPoints.assign(in->n2, nullptr);  // Tracking::TrackWithMotionModel clears them before the call (Tracking.cc:2913)
  cur.mTcw = Sophus::SE3f(Eigen::Quaternionf(in->Tcw_q[3], in->Tcw_q[0], in->Tcw_q[1], in->Tcw_q[2]),
                          Eigen::Vector3f(in->Tcw_t[0], in->Tcw_t[1], in->Tcw_t[2]));
  Frame::mnMinX = cur.grid.mnMinX = in->grid[0]; Frame::mnMinY = cur.grid.mnMinY = in->grid[1];
  Frame::mnMaxX = cur.grid.mnMaxX = in->grid[2]; Frame::mnMaxY = cur.grid.mnMaxY = in->grid[3];
  cur.grid.mfGridElementWidthInv = in->grid[4]; cur.grid.mfGridElementHeightInv = in->grid[5];
  cur.grid.Build(cur.mvKeysUn);
  ORBmatcher matcher(0.9f, in->check_orientation != 0);
  int nm;
  { CallTimer timed; nm = matcher.SearchByProjection(cur, last, in->th, in->mono != 0); }
  for (int i = 0; i < in->n2; ++i) match2[i] = cur.mvpMapPoints[i] ? (int)(cur.mvpMapPoints[i] - points.data()) : -1;
  return nm;
}

// ORBmatcher(0.9, checkOri).SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (ORBmatcher.cc:1889-2010), single
// camera.  One MapPoint object per key-frame feature that has one; features occupied on entry hold a dummy point.
extern "C" int ref_search_by_projection_kf(const orc_kf_projection_input* in, int* match2) {
  GeometricCamera cam;
  cam.fx = in->K[0]; cam.fy = in->K[1]; cam.cx = in->K[2]; cam.cy = in->K[3];
  KeyFrame kf;
  Frame cur;
  std::vector<MapPoint> points(in->n1);
  MapPoint before;
  std::set<MapPoint*> found;
  kf.N = in->n1;
  kf.mvKeysUn.resize(in->n1);
  kf.mvpMapPoints.assign(in->n1, nullptr);
  for (int i = 0; i < in->n1; ++i) {
    kf.mvKeysUn[i].angle = in->angle1[i];
    if (!in->has_mp1[i]) continue;
    MapPoint& mp = points[i];
    mp.mWorldPos = Eigen::Vector3f(in->world_pos1[3 * i], in->world_pos1[3 * i + 1], in->world_pos1[3 * i + 2]);
    mp.mDescriptor = cv::Mat(1, 32, CV_8U);
    memcpy(mp.mDescriptor.data, in->mp_desc1 + 32 * (size_t)i, 32);
    mp.bad = in->bad1[i] != 0;
    mp.mfMinDistance = in->min_dist1[i];
    mp.mfMaxDistance = in->max_dist1[i];
    kf.mvpMapPoints[i] = &mp;
    if (in->found1[i]) found.insert(&mp);
  }
  kf.mvKeys = kf.mvKeysUn;
  cur.N = in->n2;
  cur.mpCamera = &cam;
  cur.mvKeysUn.resize(in->n2);
  for (int i = 0; i < in->n2; ++i) {
    cur.mvKeysUn[i].pt.x = in->kp2_xy[2 * i]; cur.mvKeysUn[i].pt.y = in->kp2_xy[2 * i + 1];
    cur.mvKeysUn[i].octave = in->kp2_octave[i]; cur.mvKeysUn[i].angle = in->kp2_angle[i];
  }
  cur.mvKeys = cur.mvKeysUn;
  cur.mvuRight.assign(in->n2, -1.f);
  cur.mDescriptors = cv::Mat(in->n2, 32, CV_8U);
  if (in->n2) memcpy(cur.mDescriptors.data, in->desc2, (size_t)in->n2 * 32);
  cur.mvScaleFactors.assign(in->scale_factors, in->scale_factors + in->n_levels);
  cur.mnScaleLevels = in->n_levels;
  cur.mfLogScaleFactor = in->log_scale_factor;
  cur.mvpMapPoints.assign(in->n2, nullptr);
  for (int i = 0; i < in->n2; ++i)
    if (in->occupied2 && in->occupied2[i]) cur.mvpMapPoints[i] = &before;
  cur.mTcw = Sophus::SE3f(Eigen::Quaternionf(in->Tcw_q[3], in->Tcw_q[0], in->Tcw_q[1], in->Tcw_q[2]),
                          Eigen::Vector3f(in->Tcw_t[0], in->Tcw_t[1], in->Tcw_t[2]));
  Frame::mnMinX = cur.grid.mnMinX = in->grid[0]; Frame::mnMinY = cur.grid.mnMinY = in->grid[1];
  Frame::mnMaxX = cur.grid.mnMaxX = in->grid[2]; Frame::mnMaxY = cur.grid.mnMaxY = in->grid[3];
  cur.grid.mfGridElementWidthInv = in->grid[4]; cur.grid.mfGridElementHeightInv = in->grid[5];
  cur.grid.Build(cur.mvKeysUn);
  ORBmatcher matcher(0.9f, in->check_orientation != 0);
  int nm;
  { CallTimer timed; nm = matcher.SearchByProjection(cur, &kf, found, in->th, in->orb_dist); }
  for (int i = 0; i < in->n2; ++i) {
    MapPoint* p = cur.mvpMapPoints[i];
    match2[i] = (p && p != &before) ? (int)(p - points.data()) : -1;
  }
  return nm;
}

// ORBmatcher(nnratio, true).SearchByProjection(F, vpMapPoints, th, bFarPoints = false, thFarPoints) (ORBmatcher.cc:43-213).
// valid1 folds mbTrackInView / isBad; features blocked on entry hold a dummy point with observations.
extern "C" int ref_search_local_points(const orc_local_points_input* in, int* match2) {
  GeometricCamera cam;
  Frame F;
  std::vector<MapPoint> points(in->n1);
  std::vector<MapPoint*> vp(in->n1);
  for (int i = 0; i < in->n1; ++i) {
    MapPoint& mp = points[i];
    mp.mbTrackInView = in->valid1[i] != 0;
    mp.mbTrackInViewR = false;
    mp.mTrackProjX = in->proj1[3 * i]; mp.mTrackProjY = in->proj1[3 * i + 1]; mp.mTrackProjXR = in->proj1[3 * i + 2];
    mp.mnTrackScaleLevel = in->level1[i];
    mp.mTrackViewCos = in->view_cos1[i];
    mp.mDescriptor = cv::Mat(1, 32, CV_8U);
    memcpy(mp.mDescriptor.data, in->mp_desc1 + 32 * (size_t)i, 32);
    mp.nObs = in->mp_observed1[i] ? 3 : 0;
    vp[i] = &mp;
  }
  MapPoint old_point;
  old_point.nObs = 5;
  F.N = in->n2;
  F.mpCamera = &cam;
  F.mvKeysUn.resize(in->n2);
  for (int i = 0; i < in->n2; ++i) {
    F.mvKeysUn[i].pt.x = in->kp2_xy[2 * i]; F.mvKeysUn[i].pt.y = in->kp2_xy[2 * i + 1];
    F.mvKeysUn[i].octave = in->kp2_octave[i];
  }
  F.mvKeys = F.mvKeysUn;
  F.mvuRight.assign(in->uright2, in->uright2 + in->n2);
  F.mDescriptors = cv::Mat(in->n2, 32, CV_8U);
  if (in->n2) memcpy(F.mDescriptors.data, in->desc2, (size_t)in->n2 * 32);
  F.mvScaleFactors.assign(in->scale_factors, in->scale_factors + in->n_levels);
  F.mvpMapPoints.assign(in->n2, nullptr);
  for (int i = 0; i < in->n2; ++i)
    if (in->blocked2[i]) F.mvpMapPoints[i] = &old_point;
  Frame::mnMinX = F.grid.mnMinX = in->grid[0]; Frame::mnMinY = F.grid.mnMinY = in->grid[1];
  Frame::mnMaxX = F.grid.mnMaxX = in->grid[2]; Frame::mnMaxY = F.grid.mnMaxY = in->grid[3];
  F.grid.mfGridElementWidthInv = in->grid[4]; F.grid.mfGridElementHeightInv = in->grid[5];
  F.grid.Build(F.mvKeysUn);
  ORBmatcher matcher(in->nnratio, true);
  int nm;
  { CallTimer timed; nm = matcher.SearchByProjection(F, vp, in->th, false, 50.0f); }
  for (int i = 0; i < in->n2; ++i) {
    MapPoint* p = F.mvpMapPoints[i];
    match2[i] = (p && p != &old_point) ? (int)(p - points.data()) : -1;
  }
  return nm;
}

// ORBmatcher(nnratio, checkOri).SearchByBoW(pKF, F, vpMapPointMatches) (ORBmatcher.cc:223-425), single camera.
extern "C" int ref_search_by_bow(const KfArrays* kfa, const KfArrays* fra, float nnratio, int check_orientation, int* match2) {
  GeometricCamera cam;
  std::vector<MapPoint> pts(kfa->n);
  KeyFrame kf;
  kf.N = kfa->n;
  kf.mpCamera = &cam;
  kf.mDescriptors = cv::Mat(kfa->n, 32, CV_8U);
  if (kfa->n) memcpy(kf.mDescriptors.data, kfa->desc, (size_t)kfa->n * 32);
  kf.mvKeysUn.resize(kfa->n);
  kf.mvpMapPoints.assign(kfa->n, nullptr);
  for (int i = 0; i < kfa->n; ++i) {
    kf.mvKeysUn[i].angle = kfa->kp_angle[i];
    if (kfa->has_mp[i] == 1) kf.mvpMapPoints[i] = &pts[i];
    else if (kfa->has_mp[i] == 2) { pts[i].bad = true; kf.mvpMapPoints[i] = &pts[i]; }  // present but bad: skipped as well
  }
  for (int k = 0; k < kfa->nnodes; ++k)
    kf.mFeatVec[(unsigned)kfa->node_id[k]] = std::vector<unsigned>(kfa->node_feat + kfa->node_off[k], kfa->node_feat + kfa->node_off[k + 1]);
  Frame F;
  F.N = fra->n;
  F.mpCamera = &cam;
  F.mvKeys.resize(fra->n);
  for (int i = 0; i < fra->n; ++i) F.mvKeys[i].angle = fra->kp_angle[i];
  F.mvKeysUn = F.mvKeys;
  F.mDescriptors = cv::Mat(fra->n, 32, CV_8U);
  if (fra->n) memcpy(F.mDescriptors.data, fra->desc, (size_t)fra->n * 32);
  for (int k = 0; k < fra->nnodes; ++k)
    F.mFeatVec[(unsigned)fra->node_id[k]] = std::vector<unsigned>(fra->node_feat + fra->node_off[k], fra->node_feat + fra->node_off[k + 1]);
  ORBmatcher matcher(nnratio, check_orientation != 0);
  std::vector<MapPoint*> vpMapPointMatches;
  int nm;
  { CallTimer timed; nm = matcher.SearchByBoW(&kf, F, vpMapPointMatches); }
  for (int i = 0; i < fra->n; ++i) match2[i] = vpMapPointMatches[i] ? (int)(vpMapPointMatches[i] - pts.data()) : -1;
  return nm;
}

// The same on a two-camera frame (F.Nleft = n_left, F.mpCamera2 set: ORBmatcher.cc:298-326, 357-386).  The angles of the flat
// arrays go where the reference reads them: F.mvKeys[i] for i < Nleft, F.mvKeysRight[i - Nleft] beyond; the key frame's into
// mvKeysUn (kf_n_left < 0: pKF->mpCamera2 == nullptr) or mvKeys / mvKeysRight (kf_n_left >= 0).
extern "C" int ref_search_by_bow_rig(const KfArrays* kfa, int kf_n_left, const KfArrays* fra, int n_left, float nnratio, int check_orientation,
                                     int* match2) {
  GeometricCamera cam, cam2;
  std::vector<MapPoint> pts(kfa->n);
  KeyFrame kf;
  kf.N = kfa->n;
  kf.mpCamera = &cam;
  kf.mDescriptors = cv::Mat(kfa->n, 32, CV_8U);
  if (kfa->n) memcpy(kf.mDescriptors.data, kfa->desc, (size_t)kfa->n * 32);
  kf.mvpMapPoints.assign(kfa->n, nullptr);
  if (kf_n_left >= 0) {
    kf.mpCamera2 = &cam2;
    kf.NLeft = kf_n_left;
    kf.mvKeys.resize(kf_n_left);
    kf.mvKeysRight.resize(kfa->n - kf_n_left);
    for (int i = 0; i < kfa->n; ++i) (i < kf_n_left ? kf.mvKeys[i] : kf.mvKeysRight[i - kf_n_left]).angle = kfa->kp_angle[i];
  } else {
    kf.mvKeysUn.resize(kfa->n);
    for (int i = 0; i < kfa->n; ++i) kf.mvKeysUn[i].angle = kfa->kp_angle[i];
  }
  for (int i = 0; i < kfa->n; ++i) {
    if (kfa->has_mp[i] == 1) kf.mvpMapPoints[i] = &pts[i];
    else if (kfa->has_mp[i] == 2) { pts[i].bad = true; kf.mvpMapPoints[i] = &pts[i]; }
  }
  for (int k = 0; k < kfa->nnodes; ++k)
    kf.mFeatVec[(unsigned)kfa->node_id[k]] = std::vector<unsigned>(kfa->node_feat + kfa->node_off[k], kfa->node_feat + kfa->node_off[k + 1]);
  Frame F;
  F.N = fra->n;
  F.Nleft = n_left;
  F.mpCamera = &cam;
  F.mpCamera2 = &cam2;
  F.mvKeys.resize(n_left);
  F.mvKeysRight.resize(fra->n - n_left);
  for (int i = 0; i < fra->n; ++i) (i < n_left ? F.mvKeys[i] : F.mvKeysRight[i - n_left]).angle = fra->kp_angle[i];
  F.mDescriptors = cv::Mat(fra->n, 32, CV_8U);
  if (fra->n) memcpy(F.mDescriptors.data, fra->desc, (size_t)fra->n * 32);
  for (int k = 0; k < fra->nnodes; ++k)
    F.mFeatVec[(unsigned)fra->node_id[k]] = std::vector<unsigned>(fra->node_feat + fra->node_off[k], fra->node_feat + fra->node_off[k + 1]);
  ORBmatcher matcher(nnratio, check_orientation != 0);
  std::vector<MapPoint*> vpMapPointMatches;
  int nm;
  { CallTimer timed; nm = matcher.SearchByBoW(&kf, F, vpMapPointMatches); }
  for (int i = 0; i < fra->n; ++i) match2[i] = vpMapPointMatches[i] ? (int)(vpMapPointMatches[i] - pts.data()) : -1;
  return nm;
}

// ORBmatcher(nnratio, checkOri).SearchByBoW(pKF1, pKF2, vpMatches12) (ORBmatcher.cc:765-905); has_mp: 0 none, 1 good, 2 bad.
extern "C" int ref_search_by_bow_kf(const KfArrays* a1, const KfArrays* a2, float nnratio, int check_orientation, int* match12) {
  GeometricCamera cam;
  std::vector<MapPoint> pts1(a1->n), pts2(a2->n);
  KeyFrame kf[2];
  const KfArrays* src[2] = {a1, a2};
  std::vector<MapPoint>* pts[2] = {&pts1, &pts2};
  for (int s = 0; s < 2; ++s) {
    const KfArrays* a = src[s];
    kf[s].N = a->n;
    kf[s].mpCamera = &cam;
    kf[s].mDescriptors = cv::Mat(a->n, 32, CV_8U);
    if (a->n) memcpy(kf[s].mDescriptors.data, a->desc, (size_t)a->n * 32);
    kf[s].mvKeysUn.resize(a->n);
    kf[s].mvpMapPoints.assign(a->n, nullptr);
    for (int i = 0; i < a->n; ++i) {
      kf[s].mvKeysUn[i].angle = a->kp_angle[i];
      if (a->has_mp[i] == 1) kf[s].mvpMapPoints[i] = &(*pts[s])[i];
      else if (a->has_mp[i] == 2) { (*pts[s])[i].bad = true; kf[s].mvpMapPoints[i] = &(*pts[s])[i]; }
    }
    for (int k = 0; k < a->nnodes; ++k)
      kf[s].mFeatVec[(unsigned)a->node_id[k]] = std::vector<unsigned>(a->node_feat + a->node_off[k], a->node_feat + a->node_off[k + 1]);
  }
  ORBmatcher matcher(nnratio, check_orientation != 0);
  std::vector<MapPoint*> vpMatches12;
  int nm;
  { CallTimer timed; nm = matcher.SearchByBoW(&kf[0], &kf[1], vpMatches12); }
  for (int i = 0; i < a1->n; ++i) match12[i] = vpMatches12[i] ? (int)(vpMatches12[i] - pts2.data()) : -1;
  return nm;
}

// ORBmatcher(0.6, true).Fuse(pKF, vpMapPoints, th) (ORBmatcher.cc:1148-1338) on stand-in objects.  kf_state2: 0 = the feature
// has no map point, 1 = a good one with more observations than the candidates, 2 = a good one with fewer, 3 = a bad one
// (all four branches of :1305-1322 are taken).  best_idx[i] = the feature the reference fused point i with, read back from
// what it did to the objects (AddObservation / Replace), or -1.
extern "C" int ref_fuse(const orc_fuse_input* in, const uint8_t* kf_state2, int* best_idx) {
  GeometricCamera cam;
  cam.fx = in->K[0]; cam.fy = in->K[1]; cam.cx = in->K[2]; cam.cy = in->K[3];
  KeyFrame kf;
  std::vector<MapPoint> points(in->n1), owned(in->n2);
  std::vector<MapPoint*> vp(in->n1, nullptr);
  kf.N = in->n2;
  kf.mpCamera = &cam;
  kf.fx = in->K[0]; kf.fy = in->K[1]; kf.cx = in->K[2]; kf.cy = in->K[3]; kf.mbf = in->bf;
  kf.mvKeysUn.resize(in->n2);
  kf.mvpMapPoints.assign(in->n2, nullptr);
  for (int i = 0; i < in->n2; ++i) {
    kf.mvKeysUn[i].pt.x = in->kp2_xy[2 * i]; kf.mvKeysUn[i].pt.y = in->kp2_xy[2 * i + 1];
    kf.mvKeysUn[i].octave = in->kp2_octave[i];
    if (kf_state2[i]) {
      owned[i].nObs = kf_state2[i] == 1 ? 9 : 1;
      owned[i].bad = kf_state2[i] == 3;
      kf.mvpMapPoints[i] = &owned[i];
    }
  }
  kf.mvKeys = kf.mvKeysUn;
  kf.mvuRight.assign(in->uright2, in->uright2 + in->n2);
  kf.mDescriptors = cv::Mat(in->n2, 32, CV_8U);
  if (in->n2) memcpy(kf.mDescriptors.data, in->desc2, (size_t)in->n2 * 32);
  kf.mvScaleFactors.assign(in->scale_factors, in->scale_factors + in->n_levels);
  kf.mvInvLevelSigma2.assign(in->inv_level_sigma2, in->inv_level_sigma2 + in->n_levels);
  kf.mnScaleLevels = in->n_levels;
  kf.mfLogScaleFactor = in->log_scale_factor;
  kf.mTcw = Sophus::SE3f(Eigen::Quaternionf(in->Tcw_q[3], in->Tcw_q[0], in->Tcw_q[1], in->Tcw_q[2]),
                         Eigen::Vector3f(in->Tcw_t[0], in->Tcw_t[1], in->Tcw_t[2]));
  kf.mTwc = Sophus::SE3f(Eigen::Quaternionf(1.f, 0.f, 0.f, 0.f), Eigen::Vector3f(in->Ow[0], in->Ow[1], in->Ow[2]));  // only its translation is read
  kf.grid.mnMinX = in->grid[0]; kf.grid.mnMinY = in->grid[1]; kf.grid.mnMaxX = in->grid[2]; kf.grid.mnMaxY = in->grid[3];
  kf.grid.mfGridElementWidthInv = in->grid[4]; kf.grid.mfGridElementHeightInv = in->grid[5];
  kf.grid.Build(kf.mvKeysUn);
  for (int i = 0; i < in->n1; ++i) {
    if (!in->has_mp1[i]) continue;
    MapPoint& mp = points[i];
    mp.mWorldPos = Eigen::Vector3f(in->world_pos1[3 * i], in->world_pos1[3 * i + 1], in->world_pos1[3 * i + 2]);
    mp.mNormal = Eigen::Vector3f(in->normal1[3 * i], in->normal1[3 * i + 1], in->normal1[3 * i + 2]);
    mp.mDescriptor = cv::Mat(1, 32, CV_8U);
    memcpy(mp.mDescriptor.data, in->mp_desc1 + 32 * (size_t)i, 32);
    mp.bad = in->bad1[i] != 0;
    mp.mfMinDistance = in->min_dist1[i];
    mp.mfMaxDistance = in->max_dist1[i];
    mp.nObs = 4;
    if (in->in_kf1[i]) mp.mObservations[&kf] = 0;
    vp[i] = &mp;
  }
  std::vector<std::pair<MapPoint*, MapPoint*>> log;
  MapPoint::replace_log = &log;
  ORBmatcher matcher(0.6f, true);
  int nFused;
  { CallTimer timed; nFused = matcher.Fuse(&kf, vp, in->th); }
  MapPoint::replace_log = nullptr;
  for (int i = 0; i < in->n1; ++i) {
    best_idx[i] = -1;
    if (!vp[i] || in->in_kf1[i]) continue;
    auto it = points[i].mObservations.find(&kf);
    if (it != points[i].mObservations.end()) best_idx[i] = it->second;  // AddObservation(pKF, bestIdx)
  }
  // pMP->Replace(pMPinKF) or pMPinKF->Replace(pMP): the one of the two that sits in the key frame (one of its own points, or
  // an earlier candidate that was added to a free feature) names the feature, the other one is the point being fused
  auto feature_of = [&](MapPoint* p) -> int {
    if (p >= owned.data() && p < owned.data() + in->n2) return (int)(p - owned.data());
    const int i = (int)(p - points.data());
    if (in->in_kf1[i]) return -1;
    auto it = p->mObservations.find(&kf);
    return (it != p->mObservations.end() && kf.mvpMapPoints[it->second] == p) ? it->second : -1;
  };
  for (auto& pr : log) {
    const int fa = feature_of(pr.first), fb = feature_of(pr.second);
    MapPoint* cand = fa >= 0 ? pr.second : pr.first;
    const int feat = fa >= 0 ? fa : fb;
    if (cand >= points.data() && cand < points.data() + in->n1 && best_idx[cand - points.data()] < 0) best_idx[cand - points.data()] = feat;
  }
  // a feature holding a bad map point: the match counts (nFused++) but nothing is recorded (best_idx stays -1)
  return nFused;
}

// One side of the Sim3 matchers: a key frame (features, grid) and the map points its features hold, already expressed in the
// common camera frame (all poses of the call are identities, so the world frame IS that frame and the Sophus / Eigen transform
// arithmetic - third-party code - is exact and out of the picture).
struct Sim3Side {
  int n;
  const float* kp_xy; const int32_t* kp_octave; const uint8_t* desc;
  const uint8_t* mp_state;    // per feature: 0 no map point, 1 good, 2 bad
  const float* mp_pos;        // 3 floats per feature
  const float* mp_normal;     // 3 floats per feature (Fuse only)
  const uint8_t* mp_desc;     // MapPoint::GetDescriptor(), 32 bytes per feature
  const float *mp_min_dist, *mp_max_dist;
};
static void build_side(const Sim3Side& a, const float K[4], const float grid[6], const float* scale_factors, int n_levels,
                       float log_scale_factor, GeometricCamera* cam, KeyFrame& kf, std::vector<MapPoint>& pts) {
  kf.N = a.n;
  kf.mpCamera = cam;
  kf.fx = K[0]; kf.fy = K[1]; kf.cx = K[2]; kf.cy = K[3];
  kf.mvKeysUn.resize(a.n);
  kf.mvpMapPoints.assign(a.n, nullptr);
  pts.assign(a.n, MapPoint());
  for (int i = 0; i < a.n; ++i) {
    kf.mvKeysUn[i].pt.x = a.kp_xy[2 * i]; kf.mvKeysUn[i].pt.y = a.kp_xy[2 * i + 1]; kf.mvKeysUn[i].octave = a.kp_octave[i];
    if (!a.mp_state[i]) continue;
    MapPoint& mp = pts[i];
    mp.bad = a.mp_state[i] == 2;
    mp.mWorldPos = Eigen::Vector3f(a.mp_pos[3 * i], a.mp_pos[3 * i + 1], a.mp_pos[3 * i + 2]);
    if (a.mp_normal) mp.mNormal = Eigen::Vector3f(a.mp_normal[3 * i], a.mp_normal[3 * i + 1], a.mp_normal[3 * i + 2]);
    mp.mDescriptor = cv::Mat(1, 32, CV_8U);
    memcpy(mp.mDescriptor.data, a.mp_desc + 32 * (size_t)i, 32);
    mp.mfMinDistance = a.mp_min_dist[i]; mp.mfMaxDistance = a.mp_max_dist[i];
    mp.nObs = 3;
    mp.mObservations[&kf] = i;
    kf.mvpMapPoints[i] = &mp;
  }
  kf.mvKeys = kf.mvKeysUn;
  kf.mvuRight.assign(a.n, -1.f);
  kf.mDescriptors = cv::Mat(a.n, 32, CV_8U);
  if (a.n) memcpy(kf.mDescriptors.data, a.desc, (size_t)a.n * 32);
  kf.mvScaleFactors.assign(scale_factors, scale_factors + n_levels);
  kf.mnScaleLevels = n_levels;
  kf.mfLogScaleFactor = log_scale_factor;
  kf.mTcw = Sophus::SE3f(Eigen::Quaternionf(1.f, 0.f, 0.f, 0.f), Eigen::Vector3f(0.f, 0.f, 0.f));
  kf.mTwc = kf.mTcw;
  kf.grid.mnMinX = grid[0]; kf.grid.mnMinY = grid[1]; kf.grid.mnMaxX = grid[2]; kf.grid.mnMaxY = grid[3];
  kf.grid.mfGridElementWidthInv = grid[4]; kf.grid.mfGridElementHeightInv = grid[5];
  kf.grid.Build(kf.mvKeysUn);
}

// ORBmatcher(0.75, true).SearchBySim3(pKF1, pKF2, vpMatches12, S12 = identity, th) (ORBmatcher.cc:1457-1674).
// prior12[i1] >= 0: vpMatches12[i1] holds the map point of KF2's feature prior12[i1] on entry.  match12[i1] = KF2 feature or -1.
extern "C" int ref_search_by_sim3(const Sim3Side* a1, const Sim3Side* a2, const float K[4], const float grid[6],
                                  const float* scale_factors, int n_levels, float log_scale_factor, float th, const int* prior12,
                                  int* match12) {
  GeometricCamera cam;
  cam.fx = K[0]; cam.fy = K[1]; cam.cx = K[2]; cam.cy = K[3];
  KeyFrame kf1, kf2;
  std::vector<MapPoint> p1, p2;
  build_side(*a1, K, grid, scale_factors, n_levels, log_scale_factor, &cam, kf1, p1);
  build_side(*a2, K, grid, scale_factors, n_levels, log_scale_factor, &cam, kf2, p2);
  std::vector<MapPoint*> vpMatches12(a1->n, nullptr);
  for (int i = 0; i < a1->n; ++i)
    if (prior12[i] >= 0) vpMatches12[i] = &p2[prior12[i]];
  Sophus::Sim3f S12(1.f, Eigen::Quaternionf(1.f, 0.f, 0.f, 0.f), Eigen::Vector3f(0.f, 0.f, 0.f));
  ORBmatcher matcher(0.75f, true);
  int nFound;
  { CallTimer timed; nFound = matcher.SearchBySim3(&kf1, &kf2, vpMatches12, S12, th); }
  for (int i = 0; i < a1->n; ++i) match12[i] = vpMatches12[i] ? (int)(vpMatches12[i] - p2.data()) : -1;
  return nFound;
}

// ORBmatcher(0.8).Fuse(pKF, Scw = identity, vpPoints, th, vpReplacePoint) (ORBmatcher.cc:1340-1455).  The candidate points are
// a Sim3Side of their own (mp_state 1 / 2 = good / bad; every entry is a point); kf_state2 as in ref_fuse.
// fused[i] = feature the point was fused with (AddObservation or vpReplacePoint), -1 otherwise.
extern "C" int ref_fuse_sim3(const Sim3Side* cand, const uint8_t* in_kf1, const Sim3Side* kfa, const uint8_t* kf_state2,
                             const float K[4], const float grid[6], const float* scale_factors, int n_levels,
                             float log_scale_factor, float th, int* fused) {
  GeometricCamera cam;
  cam.fx = K[0]; cam.fy = K[1]; cam.cx = K[2]; cam.cy = K[3];
  KeyFrame kf, other;
  std::vector<MapPoint> owned, pts;
  Sim3Side bare = *kfa;
  std::vector<uint8_t> none(kfa->n, 0);
  bare.mp_state = none.data();
  build_side(bare, K, grid, scale_factors, n_levels, log_scale_factor, &cam, kf, owned);
  for (int i = 0; i < kfa->n; ++i)
    if (kf_state2[i]) { owned[i].nObs = 5; owned[i].bad = kf_state2[i] == 3; kf.mvpMapPoints[i] = &owned[i]; }
  build_side(*cand, K, grid, scale_factors, n_levels, log_scale_factor, &cam, other, pts);  // only to construct the points
  std::vector<MapPoint*> vp(cand->n), vpReplacePoint(cand->n, nullptr);
  for (int i = 0; i < cand->n; ++i) {
    vp[i] = &pts[i];
    pts[i].mObservations.clear();
    if (in_kf1[i]) {  // a candidate that the key frame already holds
      const int slot = kfa->n - 1 - (i % kfa->n);
      kf.mvpMapPoints[slot] = &pts[i];
      pts[i].mObservations[&kf] = slot;
    }
  }
  Sophus::Sim3f Scw(1.f, Eigen::Quaternionf(1.f, 0.f, 0.f, 0.f), Eigen::Vector3f(0.f, 0.f, 0.f));
  ORBmatcher matcher(0.8f, true);
  int nFused;
  { CallTimer timed; nFused = matcher.Fuse(&kf, Scw, vp, th, vpReplacePoint); }
  for (int i = 0; i < cand->n; ++i) {
    fused[i] = -1;
    if (vpReplacePoint[i]) {
      MapPoint* q = vpReplacePoint[i];
      if (q >= owned.data() && q < owned.data() + kfa->n) fused[i] = (int)(q - owned.data());
      else { auto it = q->mObservations.find(&kf); fused[i] = it != q->mObservations.end() ? it->second : -3; }
    } else if (!in_kf1[i]) {
      auto it = pts[i].mObservations.find(&kf);
      if (it != pts[i].mObservations.end()) fused[i] = it->second;
    }
  }
  return nFused;
}

// ORBmatcher(0.75, true).SearchByProjection(pKF, Scw = identity, vpPoints, vpMatched, th, ratioHamming) (ORBmatcher.cc:427-532;
// with_kfs = 0) or the overload with vpPointsKFs / vpMatchedKF (:534-646; with_kfs = 1).  cand: the candidate points (every
// entry a point, mp_state 1 / 2); found1[i]: the point already sits in vpMatched on entry (at feature matched_slot);
// matched2[i]: vpMatched[i] != NULL on entry.  match2[i2] = candidate stored in vpMatched[i2] by the call, or -1.
extern "C" int ref_search_by_projection_sim3(const Sim3Side* cand, const uint8_t* found1, const Sim3Side* kfa, const uint8_t* matched2,
                                             const float K[4], const float grid[6], const float* scale_factors, int n_levels,
                                             float log_scale_factor, int th, float ratioHamming, int with_kfs, int* match2) {
  GeometricCamera cam;
  cam.fx = K[0]; cam.fy = K[1]; cam.cx = K[2]; cam.cy = K[3];
  KeyFrame kf, other, marker;
  std::vector<MapPoint> owned, pts;
  Sim3Side bare = *kfa;
  std::vector<uint8_t> none(kfa->n, 0);
  bare.mp_state = none.data();
  build_side(bare, K, grid, scale_factors, n_levels, log_scale_factor, &cam, kf, owned);
  build_side(*cand, K, grid, scale_factors, n_levels, log_scale_factor, &cam, other, pts);
  MapPoint before;
  std::vector<MapPoint*> vp(cand->n), vpMatched(kfa->n, nullptr);
  std::vector<KeyFrame*> vpKFs(cand->n, &other), vpMatchedKF(kfa->n, nullptr);
  for (int i = 0; i < kfa->n; ++i)
    if (matched2[i]) vpMatched[i] = &before;
  int slot = 0;
  for (int i = 0; i < cand->n; ++i) {
    vp[i] = &pts[i];
    if (found1[i]) {  // spAlreadyFound: put the point itself into an entry that is matched on entry
      while (slot < kfa->n && !matched2[slot]) ++slot;
      if (slot < kfa->n) vpMatched[slot++] = &pts[i];
    }
  }
  Sophus::Sim3f Scw(1.f, Eigen::Quaternionf(1.f, 0.f, 0.f, 0.f), Eigen::Vector3f(0.f, 0.f, 0.f));
  ORBmatcher matcher(0.75f, true);
  std::vector<MapPoint*> entry = vpMatched;
  const int nm = with_kfs ? matcher.SearchByProjection(&kf, Scw, vp, vpKFs, vpMatched, vpMatchedKF, th, ratioHamming)
                          : matcher.SearchByProjection(&kf, Scw, vp, vpMatched, th, ratioHamming);
  for (int i = 0; i < kfa->n; ++i) {
    match2[i] = -1;
    if (vpMatched[i] && vpMatched[i] != entry[i]) match2[i] = (int)(vpMatched[i] - pts.data());
    if (with_kfs && match2[i] >= 0 && vpMatchedKF[i] != &other) match2[i] = -3;  // the key frame of the point must come along
  }
  return nm;
}

// ORBmatcher(nnratio, checkOri).SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (ORBmatcher.cc:648-763)
extern "C" int ref_search_for_initialization(const orc_initialization_input* in, float* prev_matched, int* matches12) {
  Frame F1, F2;
  F1.N = in->n1;
  F1.mvKeysUn.resize(in->n1);
  for (int i = 0; i < in->n1; ++i) { F1.mvKeysUn[i].octave = in->kp1_octave[i]; F1.mvKeysUn[i].angle = in->kp1_angle[i]; }
  F1.mDescriptors = cv::Mat(in->n1, 32, CV_8U);
  if (in->n1) memcpy(F1.mDescriptors.data, in->desc1, (size_t)in->n1 * 32);
  F2.N = in->n2;
  F2.mvKeysUn.resize(in->n2);
  for (int i = 0; i < in->n2; ++i) {
    F2.mvKeysUn[i].pt.x = in->kp2_xy[2 * i]; F2.mvKeysUn[i].pt.y = in->kp2_xy[2 * i + 1];
    F2.mvKeysUn[i].octave = in->kp2_octave[i]; F2.mvKeysUn[i].angle = in->kp2_angle[i];
  }
  F2.mDescriptors = cv::Mat(in->n2, 32, CV_8U);
  if (in->n2) memcpy(F2.mDescriptors.data, in->desc2, (size_t)in->n2 * 32);
  Frame::mnMinX = F2.grid.mnMinX = in->grid[0]; Frame::mnMinY = F2.grid.mnMinY = in->grid[1];
  Frame::mnMaxX = F2.grid.mnMaxX = in->grid[2]; Frame::mnMaxY = F2.grid.mnMaxY = in->grid[3];
  F2.grid.mfGridElementWidthInv = in->grid[4]; F2.grid.mfGridElementHeightInv = in->grid[5];
  F2.grid.Build(F2.mvKeysUn);
  std::vector<cv::Point2f> prev(in->n1);
  for (int i = 0; i < in->n1; ++i) { prev[i].x = prev_matched[2 * i]; prev[i].y = prev_matched[2 * i + 1]; }
  std::vector<int> m12;
  ORBmatcher matcher(in->nnratio, in->check_orientation != 0);
  int nm;
  { CallTimer timed; nm = matcher.SearchForInitialization(F1, F2, prev, m12, in->window_size); }
  for (int i = 0; i < in->n1; ++i) { matches12[i] = m12[i]; prev_matched[2 * i] = prev[i].x; prev_matched[2 * i + 1] = prev[i].y; }
  return nm;
}
