// Drop-in replacement of the Hamming paths of /root/reference/include/ORBmatcher.h on librgbl_frontend.so:
//   static DescriptorDistance(a, b)                                  (ORBmatcher.h:43,  ORBmatcher.cc:2058-2074)
//   SearchForTriangulation(pKF1, pKF2, vMatchedPairs, bOnlyStereo, bCoarse)   (ORBmatcher.h:75-76, :907-1146)
//   SearchByProjection(CurrentFrame, LastFrame, th, bMono)           (ORBmatcher.h:48,  ORBmatcher.cc:1676-1887)
//   SearchByProjection(F, vpMapPoints, th, bFarPoints, thFarPoints)  (ORBmatcher.h:45,  ORBmatcher.cc:43-213)
//   SearchByBoW(pKF, F, vpMapPointMatches)                           (ORBmatcher.h:57,  ORBmatcher.cc:223-425)
// The other Search*/Fuse members of the reference class are untouched (SURVEY.md §8(f) lists them as "next");
// in the ORB_SLAM3 tree this header is merged into the existing one, see INTEGRATION.md.
//
// SearchForTriangulation is a member template so that this file does not need KeyFrame.h / Sophus / DBoW2 here:
// it only uses the public KeyFrame members the reference function itself touches (mFeatVec, GetMapPoint, mvuRight,
// mvKeysUn, mDescriptors, N, NLeft, mpCamera, mpCamera2, GetPose, GetPoseInverse, GetCameraCenter, mvScaleFactors,
// mvLevelSigma2), so `matcher.SearchForTriangulation(mpCurrentKeyFrame, pKF2, vMatchedIndices, false, bCoarse)`
// (LocalMapping.cc:466) compiles unchanged.
#ifndef ORBMATCHER_H
#define ORBMATCHER_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#include <iostream>
#include <mutex>
#include <utility>
#include <set>
#include <tuple>
#include <vector>

#include "../../include/rgbl_frontend.h"
#include "cv_compat.h"

namespace rgbl_shim {
// A Frame / KeyFrame class with a member `rgbl_device_frame* mpDeviceFrame` (one line in Frame.h / KeyFrame.h,
// filled by ORBextractor::CaptureDeviceFrame in the Frame constructor) is matched from its resident copy: descriptors,
// mvKeysUn and mvuRight are not uploaded again.  Classes without the member behave as before.
template <class T> auto device_frame_of(const T& f, int) -> decltype(static_cast<const rgbl_device_frame*>(f.mpDeviceFrame)) { return f.mpDeviceFrame; }
template <class T> const rgbl_device_frame* device_frame_of(const T&, long) { return nullptr; }
}  // namespace rgbl_shim

namespace ORB_SLAM3 {

class ORBmatcher {
 public:
  ORBmatcher(float nnratio = 0.6, bool checkOri = true, int device = 0) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {
    // the reference constructs this class on the stack in every call (Tracking.cc:2890 ...): the device handle comes from
    // the library's pool and goes back to it, no HIP stream / arena is created or freed per object
    if (rgbl_matcher_acquire(device, &mpHandle) != RGBL_OK) {
      std::cerr << "[ORBmatcher] " << rgbl_last_error() << std::endl;
      mpHandle = nullptr;
    }
  }
  ~ORBmatcher() { rgbl_matcher_release(mpHandle); }
  ORBmatcher(const ORBmatcher&) = delete;
  ORBmatcher& operator=(const ORBmatcher&) = delete;

  // Computes the Hamming distance between two ORB descriptors (rows of 32 bytes).
  static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b) {
    return rgbl_descriptor_distance(a.ptr<uint8_t>(), b.ptr<uint8_t>());
  }

  // Matching to triangulate new MapPoints. Check Epipolar Constraint.
  template <class KeyFrameT>
  int SearchForTriangulation(KeyFrameT* pKF1, KeyFrameT* pKF2, std::vector<std::pair<size_t, size_t> >& vMatchedPairs,
                             const bool bOnlyStereo, const bool bCoarse = false) {
    vMatchedPairs.clear();
    if (!mpHandle) return 0;
    if (pKF1->mpCamera2 || pKF2->mpCamera2 || pKF1->NLeft != -1 || pKF2->NLeft != -1) {
      std::cerr << "[ORBmatcher] SearchForTriangulation: fisheye stereo rigs (mpCamera2) are not covered by the device path" << std::endl;
      return 0;
    }
    // relative pose and epipole, ORBmatcher.cc:913-931
    auto T12 = pKF1->GetPose() * pKF2->GetPoseInverse();
    auto R = T12.rotationMatrix();
    auto t = T12.translation();
    auto C2 = pKF2->GetPose() * pKF1->GetCameraCenter();
    auto ep = pKF2->mpCamera->project(C2);
    float R12[9], t12[3], K1[4], K2[4];
    for (int i = 0; i < 3; ++i) {
      t12[i] = t(i);
      for (int j = 0; j < 3; ++j) R12[3 * i + j] = R(i, j);
    }
    for (int i = 0; i < 4; ++i) {
      K1[i] = pKF1->mpCamera->getParameter(i);
      K2[i] = pKF2->mpCamera->getParameter(i);
    }
    rgbl_triangulation_params prm;
    rgbl_fundamental(K1, K2, R12, t12, prm.F12);  // constant per key-frame pair (the reference rebuilds it per candidate)
    prm.epipole[0] = ep(0);
    prm.epipole[1] = ep(1);
    prm.scale_factors2 = pKF2->mvScaleFactors.data();
    prm.level_sigma2_2 = pKF2->mvLevelSigma2.data();
    prm.n_levels = (int)pKF2->mvScaleFactors.size();
    prm.only_stereo = bOnlyStereo;
    prm.coarse = bCoarse;
    prm.check_orientation = mbCheckOrientation;

    Flat f1, f2;
    Flatten(pKF1, f1);
    Flatten(pKF2, f2);
    std::vector<int32_t> vMatches12(f1.view.n, -1);
    int nmatches = 0;
    if (rgbl_search_triangulation(mpHandle, &f1.view, &f2.view, &prm, vMatches12.data(), &nmatches) != RGBL_OK) {
      std::cerr << "[ORBmatcher] " << rgbl_last_error() << std::endl;
      return 0;
    }
    vMatchedPairs.reserve(nmatches);
    for (size_t i = 0, iend = vMatches12.size(); i < iend; i++) {
      if (vMatches12[i] < 0) continue;
      vMatchedPairs.push_back(std::make_pair(i, (size_t)vMatches12[i]));
    }
    return nmatches;
  }

  // Search matches between MapPoints in a KeyFrame and ORB in a Frame.  Brute force constrained to ORB that belong to the same
  // vocabulary node (at a certain level).  Used in Relocalisation and Loop Detection (and Tracking::TrackReferenceKeyFrame,
  // Tracking.cc:2798-2810).  ORBmatcher.h:57, ORBmatcher.cc:223-425.
  template <class KeyFrameT, class FrameT, class MapPointT>
  int SearchByBoW(KeyFrameT* pKF, FrameT& F, std::vector<MapPointT*>& vpMapPointMatches) {
    const std::vector<MapPointT*> vpMapPointsKF = pKF->GetMapPointMatches();
    vpMapPointMatches = std::vector<MapPointT*>(F.N, static_cast<MapPointT*>(NULL));
    if (!mpHandle) return 0;
    Flat fk, ff;
    const int n1 = pKF->N, n2 = F.N;
    fk.xy.assign(2 * (size_t)n1, 0.f); fk.angle.resize(n1); fk.octave.assign(n1, 0); fk.has_mp.resize(n1);
    for (int i = 0; i < n1; ++i) {
      // the key point the reference reads the angle from, ORBmatcher.cc:335-338 / 362-365
      fk.angle[i] = !pKF->mpCamera2 ? pKF->mvKeysUn[i].angle : (i >= pKF->NLeft) ? pKF->mvKeysRight[i - pKF->NLeft].angle : pKF->mvKeys[i].angle;
      fk.has_mp[i] = (vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad()) ? 1 : 0;
    }
    ff.xy.assign(2 * (size_t)n2, 0.f); ff.angle.resize(n2); ff.octave.assign(n2, 0); ff.has_mp.assign(n2, 0);
    // :340-343 / :367-373: a two-camera frame's features from Nleft on are the right camera's key points
    for (int i = 0; i < n2; ++i) ff.angle[i] = (F.Nleft != -1 && i >= F.Nleft) ? F.mvKeysRight[i - F.Nleft].angle : F.mvKeys[i].angle;
    FlattenFeatVec(pKF->mFeatVec, fk);
    FlattenFeatVec(F.mFeatVec, ff);
    std::vector<float> ur1(n1, -1.f), ur2(n2, -1.f);
    FillView(fk, n1, pKF->mDescriptors.template ptr<uint8_t>(), ur1.data());
    FillView(ff, n2, F.mDescriptors.template ptr<uint8_t>(), ur2.data());
    std::vector<int32_t> match(n2, -1);
    int nmatches = 0;
    if (rgbl_search_by_bow_rig(mpHandle, &fk.view, &ff.view, F.Nleft, mfNNratio, mbCheckOrientation, match.data(), &nmatches) != RGBL_OK) {
      std::cerr << "[ORBmatcher] " << rgbl_last_error() << std::endl;
      return 0;
    }
    for (int i = 0; i < n2; ++i)
      if (match[i] >= 0) vpMapPointMatches[i] = vpMapPointsKF[match[i]];
    return nmatches;
  }

  // The table and the selection of MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:370-398): BestIdx among the observed
  // descriptors (1 x 32 CV_8U rows, in the order the reference collects them), -1 for an empty list.
  int DistinctiveDescriptor(const std::vector<cv::Mat>& vDescriptors) {
    if (!mpHandle || vDescriptors.empty()) return vDescriptors.empty() ? -1 : 0;
    const int n = (int)vDescriptors.size();
    std::vector<uint8_t> rows((size_t)n * 32);
    for (int i = 0; i < n; ++i) memcpy(&rows[(size_t)i * 32], vDescriptors[i].template ptr<uint8_t>(), 32);
    const int32_t off[2] = {0, n};
    int32_t best = 0;
    if (rgbl_distinctive_descriptors(mpHandle, rows.data(), off, 1, &best) != RGBL_OK) {
      std::cerr << "[ORBmatcher] " << rgbl_last_error() << std::endl;
      return 0;
    }
    return best;
  }

  // Matching between the MapPoints of two KeyFrames through the vocabulary (LoopClosing's place recognition and the merge /
  // relocalisation of the multi-map case).  ORBmatcher.h:58, ORBmatcher.cc:765-905.
  template <class KeyFrameT, class MapPointT>
  int SearchByBoW(KeyFrameT* pKF1, KeyFrameT* pKF2, std::vector<MapPointT*>& vpMatches12) {
    const std::vector<MapPointT*> vpMapPoints1 = pKF1->GetMapPointMatches();
    const std::vector<MapPointT*> vpMapPoints2 = pKF2->GetMapPointMatches();
    vpMatches12 = std::vector<MapPointT*>(vpMapPoints1.size(), static_cast<MapPointT*>(NULL));
    if (!mpHandle) return 0;
    if (pKF1->NLeft != -1 || pKF2->NLeft != -1) {
      std::cerr << "[ORBmatcher] SearchByBoW: fisheye stereo rigs are not covered by the device path" << std::endl;
      return 0;
    }
    Flat f1, f2;
    KeyFrameT* kfs[2] = {pKF1, pKF2};
    const std::vector<MapPointT*>* mps[2] = {&vpMapPoints1, &vpMapPoints2};
    Flat* flat[2] = {&f1, &f2};
    std::vector<float> ur[2];
    for (int s = 0; s < 2; ++s) {
      const int n = (int)mps[s]->size();
      Flat& f = *flat[s];
      f.xy.assign(2 * (size_t)n, 0.f); f.angle.resize(n); f.octave.assign(n, 0); f.has_mp.resize(n);
      for (int i = 0; i < n; ++i) {
        f.angle[i] = kfs[s]->mvKeysUn[i].angle;
        MapPointT* pMP = (*mps[s])[i];
        f.has_mp[i] = (pMP && !pMP->isBad()) ? 1 : 0;
      }
      FlattenFeatVec(kfs[s]->mFeatVec, f);
      ur[s].assign(n, -1.f);
      FillView(f, n, kfs[s]->mDescriptors.template ptr<uint8_t>(), ur[s].data());
    }
    std::vector<int32_t> match(vpMapPoints1.size(), -1);
    int nmatches = 0;
    if (rgbl_search_by_bow_keyframes(mpHandle, &f1.view, &f2.view, mfNNratio, mbCheckOrientation, match.data(), &nmatches) != RGBL_OK) {
      std::cerr << "[ORBmatcher] " << rgbl_last_error() << std::endl;
      return 0;
    }
    for (size_t i = 0; i < match.size(); ++i)
      if (match[i] >= 0) vpMatches12[i] = vpMapPoints2[match[i]];
    return nmatches;
  }

  // Project MapPoints tracked in last frame into the current frame and search matches.  Used to track from previous
  // frame (Tracking::TrackWithMotionModel, Tracking.cc:2917-2934).  ORBmatcher.h:48, ORBmatcher.cc:1676-1887.
  // A member template for the same reason as above; it reads the Frame / MapPoint members the reference loop reads
  // (mvpMapPoints, mvbOutlier, mvKeys, mvKeysUn, mvuRight, mDescriptors, mvScaleFactors, mb, mbf, Nleft, GetPose(),
  // mpCamera->getParameter(), the static grid bounds; MapPoint::GetWorldPos / GetDescriptor / Observations) and writes
  // CurrentFrame.mvpMapPoints, which must be all NULL on entry as Tracking.cc:2913 leaves it.
  template <class FrameT>
  int SearchByProjection(FrameT& CurrentFrame, const FrameT& LastFrame, const float th, const bool bMono) {
    if (!mpHandle) return 0;
    if (CurrentFrame.Nleft != -1 || LastFrame.Nleft != -1) {
      std::cerr << "[ORBmatcher] SearchByProjection: fisheye stereo rigs (Nleft != -1) are not covered by the device path" << std::endl;
      return 0;
    }
    const int n1 = LastFrame.N, n2 = CurrentFrame.N;
    std::vector<uint8_t> valid(n1, 0), observed(n1, 0), desc1((size_t)n1 * 32, 0);
    std::vector<float> pos((size_t)n1 * 3, 0.f), angle1(n1), xy2((size_t)n2 * 2), angle2(n2);
    std::vector<int32_t> oct1(n1), oct2(n2);
    for (int i = 0; i < n1; ++i) {
      oct1[i] = LastFrame.mvKeys[i].octave;
      angle1[i] = LastFrame.mvKeysUn[i].angle;
      auto* pMP = LastFrame.mvpMapPoints[i];
      if (!pMP || LastFrame.mvbOutlier[i]) continue;
      valid[i] = 1;
      const auto x3Dw = pMP->GetWorldPos();
      for (int k = 0; k < 3; ++k) pos[3 * (size_t)i + k] = x3Dw(k);
      const cv::Mat dMP = pMP->GetDescriptor();
      memcpy(&desc1[(size_t)i * 32], dMP.ptr<uint8_t>(), 32);
      observed[i] = pMP->Observations() > 0 ? 1 : 0;
    }
    for (int i = 0; i < n2; ++i) {
      xy2[2 * (size_t)i] = CurrentFrame.mvKeysUn[i].pt.x;
      xy2[2 * (size_t)i + 1] = CurrentFrame.mvKeysUn[i].pt.y;
      oct2[i] = CurrentFrame.mvKeysUn[i].octave;
      angle2[i] = CurrentFrame.mvKeysUn[i].angle;
    }
    rgbl_projection_input in{};
    in.n1 = n1; in.valid1 = valid.data(); in.world_pos1 = pos.data(); in.mp_desc1 = desc1.data();
    in.mp_observed1 = observed.data(); in.octave1 = oct1.data(); in.angle1 = angle1.data();
    in.n2 = n2; in.kp2_xy = xy2.data(); in.kp2_octave = oct2.data(); in.kp2_angle = angle2.data();
    in.uright2 = CurrentFrame.mvuRight.data(); in.desc2 = CurrentFrame.mDescriptors.template ptr<uint8_t>();
    in.grid[0] = FrameT::mnMinX; in.grid[1] = FrameT::mnMinY; in.grid[2] = FrameT::mnMaxX; in.grid[3] = FrameT::mnMaxY;
    in.grid[4] = FrameT::mfGridElementWidthInv; in.grid[5] = FrameT::mfGridElementHeightInv;
    const auto Tcw = CurrentFrame.GetPose();
    const auto Tlw = LastFrame.GetPose();
    in.Tcw_q[0] = Tcw.unit_quaternion().x(); in.Tcw_q[1] = Tcw.unit_quaternion().y(); in.Tcw_q[2] = Tcw.unit_quaternion().z();
    in.Tcw_q[3] = Tcw.unit_quaternion().w();
    in.Tlw_q[0] = Tlw.unit_quaternion().x(); in.Tlw_q[1] = Tlw.unit_quaternion().y(); in.Tlw_q[2] = Tlw.unit_quaternion().z();
    in.Tlw_q[3] = Tlw.unit_quaternion().w();
    for (int k = 0; k < 3; ++k) { in.Tcw_t[k] = Tcw.translation()(k); in.Tlw_t[k] = Tlw.translation()(k); }
    for (int k = 0; k < 4; ++k) in.K[k] = CurrentFrame.mpCamera->getParameter(k);
    in.mb = CurrentFrame.mb; in.mbf = CurrentFrame.mbf;
    in.scale_factors = CurrentFrame.mvScaleFactors.data();
    in.n_levels = (int)CurrentFrame.mvScaleFactors.size();
    in.th = th; in.mono = bMono; in.check_orientation = mbCheckOrientation;
    in.device2 = rgbl_shim::device_frame_of(CurrentFrame, 0);
    std::vector<int32_t> match2(n2, -1);
    int nmatches = 0;
    if (rgbl_search_by_projection(mpHandle, &in, match2.data(), &nmatches) != RGBL_OK) {
      std::cerr << "[ORBmatcher] " << rgbl_last_error() << std::endl;
      return 0;
    }
    for (int i2 = 0; i2 < n2; ++i2)
      if (match2[i2] >= 0) CurrentFrame.mvpMapPoints[i2] = LastFrame.mvpMapPoints[match2[i2]];
    return nmatches;
  }

  // Project MapPoints seen in KeyFrame into the Frame and search matches.  Used in relocalisation (Tracking::Relocalization,
  // Tracking.cc:3723-3752).  ORBmatcher.h:51, ORBmatcher.cc:1889-2010.  What needs the MapPoint objects is evaluated here
  // exactly as the reference loop does (isBad, sAlreadyFound, the scale-invariance range around |x3Dw - Ow|, PredictScale);
  // projection, window search and the assignment in key-frame index order run on the device.
  template <class FrameT, class KeyFrameT, class MapPointT>
  int SearchByProjection(FrameT& CurrentFrame, KeyFrameT* pKF, const std::set<MapPointT*>& sAlreadyFound, const float th,
                         const int ORBdist) {
    if (!mpHandle) return 0;
    if (CurrentFrame.Nleft != -1) {
      std::cerr << "[ORBmatcher] SearchByProjection: fisheye stereo rigs (Nleft != -1) are not covered by the device path" << std::endl;
      return 0;
    }
    const auto Tcw = CurrentFrame.GetPose();
    const auto Ow = Tcw.inverse().translation();
    const std::vector<MapPointT*> vpMPs = pKF->GetMapPointMatches();
    const int n1 = (int)vpMPs.size(), n2 = CurrentFrame.N;
    std::vector<uint8_t> valid(n1, 0), desc1((size_t)n1 * 32, 0), occupied(n2, 0);
    std::vector<float> pos((size_t)n1 * 3, 0.f), angle1(n1, 0.f), xy2((size_t)n2 * 2), angle2(n2);
    std::vector<int32_t> level1(n1, 0), oct2(n2);
    for (int i = 0; i < n1; ++i) {
      angle1[i] = pKF->mvKeysUn[i].angle;
      MapPointT* pMP = vpMPs[i];
      if (!pMP || pMP->isBad() || sAlreadyFound.count(pMP)) continue;
      const auto x3Dw = pMP->GetWorldPos();
      const auto PO = x3Dw - Ow;
      const float dist3D = PO.norm();
      if (dist3D < pMP->GetMinDistanceInvariance() || dist3D > pMP->GetMaxDistanceInvariance()) continue;
      valid[i] = 1;
      level1[i] = pMP->PredictScale(dist3D, &CurrentFrame);
      for (int k = 0; k < 3; ++k) pos[3 * (size_t)i + k] = x3Dw(k);
      const cv::Mat dMP = pMP->GetDescriptor();
      memcpy(&desc1[(size_t)i * 32], dMP.ptr<uint8_t>(), 32);
    }
    for (int i = 0; i < n2; ++i) {
      xy2[2 * (size_t)i] = CurrentFrame.mvKeysUn[i].pt.x;
      xy2[2 * (size_t)i + 1] = CurrentFrame.mvKeysUn[i].pt.y;
      oct2[i] = CurrentFrame.mvKeysUn[i].octave;
      angle2[i] = CurrentFrame.mvKeysUn[i].angle;
      occupied[i] = CurrentFrame.mvpMapPoints[i] ? 1 : 0;
    }
    rgbl_keyframe_projection_input in{};
    in.n1 = n1; in.valid1 = valid.data(); in.world_pos1 = pos.data(); in.mp_desc1 = desc1.data();
    in.level1 = level1.data(); in.angle1 = angle1.data();
    in.n2 = n2; in.kp2_xy = xy2.data(); in.kp2_octave = oct2.data(); in.kp2_angle = angle2.data();
    in.desc2 = CurrentFrame.mDescriptors.template ptr<uint8_t>(); in.occupied2 = occupied.data();
    in.grid[0] = FrameT::mnMinX; in.grid[1] = FrameT::mnMinY; in.grid[2] = FrameT::mnMaxX; in.grid[3] = FrameT::mnMaxY;
    in.grid[4] = FrameT::mfGridElementWidthInv; in.grid[5] = FrameT::mfGridElementHeightInv;
    in.Tcw_q[0] = Tcw.unit_quaternion().x(); in.Tcw_q[1] = Tcw.unit_quaternion().y(); in.Tcw_q[2] = Tcw.unit_quaternion().z();
    in.Tcw_q[3] = Tcw.unit_quaternion().w();
    for (int k = 0; k < 3; ++k) in.Tcw_t[k] = Tcw.translation()(k);
    for (int k = 0; k < 4; ++k) in.K[k] = CurrentFrame.mpCamera->getParameter(k);
    in.scale_factors = CurrentFrame.mvScaleFactors.data();
    in.n_levels = (int)CurrentFrame.mvScaleFactors.size();
    in.th = th; in.orb_dist = ORBdist; in.check_orientation = mbCheckOrientation;
    std::vector<int32_t> match2(n2, -1);
    int nmatches = 0;
    if (rgbl_search_by_projection_keyframe(mpHandle, &in, match2.data(), &nmatches) != RGBL_OK) {
      std::cerr << "[ORBmatcher] " << rgbl_last_error() << std::endl;
      return 0;
    }
    for (int i2 = 0; i2 < n2; ++i2)
      if (match2[i2] >= 0) CurrentFrame.mvpMapPoints[i2] = vpMPs[match2[i2]];
    return nmatches;
  }

  // Project MapPoints into KeyFrame and search for duplicated MapPoints (LocalMapping::SearchInNeighbors, LocalMapping.cc:737-879).
  // ORBmatcher.h:83, ORBmatcher.cc:1148-1338.  The tests that need the MapPoint object are evaluated here as the reference
  // loop does, the search of all points runs on the device (a point's search does not depend on what the loop did with the
  // points before it), then the loop's bookkeeping - Replace / AddObservation / AddMapPoint, ORBmatcher.cc:1303-1325 - is
  // applied in order; a point that an earlier step of that bookkeeping put into the key frame or made bad is skipped, as the
  // reference's tests at the top of its loop body would.
  template <class KeyFrameT, class MapPointT>
  int Fuse(KeyFrameT* pKF, const std::vector<MapPointT*>& vpMapPoints, const float th = 3.0, const bool bRight = false) {
    if (!mpHandle) return 0;
    if (bRight || pKF->NLeft != -1) {
      std::cerr << "[ORBmatcher] Fuse: fisheye stereo rigs are not covered by the device path" << std::endl;
      return 0;
    }
    const auto Tcw = pKF->GetPose();
    const auto Ow = pKF->GetCameraCenter();
    const int n1 = (int)vpMapPoints.size(), n2 = pKF->N;
    std::vector<uint8_t> valid(n1, 0), desc1((size_t)n1 * 32, 0);
    std::vector<float> pos((size_t)n1 * 3, 0.f), xy2((size_t)n2 * 2);
    std::vector<int32_t> level1(n1, 0), oct2(n2);
    for (int i = 0; i < n1; ++i) {
      MapPointT* pMP = vpMapPoints[i];
      if (!pMP || pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
      const auto p3Dw = pMP->GetWorldPos();
      const auto PO = p3Dw - Ow;
      const float dist3D = PO.norm();
      if (dist3D < pMP->GetMinDistanceInvariance() || dist3D > pMP->GetMaxDistanceInvariance()) continue;
      const auto Pn = pMP->GetNormal();
      if (PO.dot(Pn) < 0.5 * dist3D) continue;  // viewing angle must be less than 60 deg
      valid[i] = 1;
      level1[i] = pMP->PredictScale(dist3D, pKF);
      for (int k = 0; k < 3; ++k) pos[3 * (size_t)i + k] = p3Dw(k);
      const cv::Mat dMP = pMP->GetDescriptor();
      memcpy(&desc1[(size_t)i * 32], dMP.ptr<uint8_t>(), 32);
    }
    for (int i = 0; i < n2; ++i) {
      xy2[2 * (size_t)i] = pKF->mvKeysUn[i].pt.x;
      xy2[2 * (size_t)i + 1] = pKF->mvKeysUn[i].pt.y;
      oct2[i] = pKF->mvKeysUn[i].octave;
    }
    rgbl_fuse_input in{};
    in.n1 = n1; in.valid1 = valid.data(); in.world_pos1 = pos.data(); in.mp_desc1 = desc1.data(); in.level1 = level1.data();
    in.n2 = n2; in.kp2_xy = xy2.data(); in.kp2_octave = oct2.data(); in.uright2 = pKF->mvuRight.data();
    in.desc2 = pKF->mDescriptors.template ptr<uint8_t>();
    in.grid[0] = pKF->mnMinX; in.grid[1] = pKF->mnMinY; in.grid[2] = pKF->mnMaxX; in.grid[3] = pKF->mnMaxY;
    in.grid[4] = pKF->mfGridElementWidthInv; in.grid[5] = pKF->mfGridElementHeightInv;
    in.Tcw_q[0] = Tcw.unit_quaternion().x(); in.Tcw_q[1] = Tcw.unit_quaternion().y(); in.Tcw_q[2] = Tcw.unit_quaternion().z();
    in.Tcw_q[3] = Tcw.unit_quaternion().w();
    for (int k = 0; k < 3; ++k) in.Tcw_t[k] = Tcw.translation()(k);
    for (int k = 0; k < 4; ++k) in.K[k] = pKF->mpCamera->getParameter(k);
    in.bf = pKF->mbf;
    in.scale_factors = pKF->mvScaleFactors.data();
    in.inv_level_sigma2 = pKF->mvInvLevelSigma2.data();
    in.n_levels = (int)pKF->mvScaleFactors.size();
    in.th = th;
    std::vector<int32_t> best(n1, -1);
    if (rgbl_fuse_search(mpHandle, &in, best.data(), NULL) != RGBL_OK) {
      std::cerr << "[ORBmatcher] " << rgbl_last_error() << std::endl;
      return 0;
    }
    int nFused = 0;
    for (int i = 0; i < n1; ++i) {
      if (best[i] < 0) continue;
      MapPointT* pMP = vpMapPoints[i];
      if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
      // If there is already a MapPoint replace otherwise add new measurement
      MapPointT* pMPinKF = pKF->GetMapPoint(best[i]);
      if (pMPinKF) {
        if (!pMPinKF->isBad()) {
          if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
          else pMPinKF->Replace(pMP);
        }
      } else {
        pMP->AddObservation(pKF, best[i]);
        pKF->AddMapPoint(pMP, best[i]);
      }
      nFused++;
    }
    return nFused;
  }

  // Project MapPoints into KeyFrame using a given Sim3 and search for duplicated MapPoints (LoopClosing::SearchAndFuse).
  // ORBmatcher.h:86, ORBmatcher.cc:1340-1455.  The Sophus objects are applied here exactly as the reference does
  // (Tcw = SE3(Scw.rotationMatrix(), Scw.translation() / Scw.scale())), the search of all points runs on the device, the
  // loop's bookkeeping (vpReplacePoint / AddObservation / AddMapPoint) is applied in order.
  template <class KeyFrameT, class Sim3T, class MapPointT>
  int Fuse(KeyFrameT* pKF, Sim3T& Scw, const std::vector<MapPointT*>& vpPoints, float th, std::vector<MapPointT*>& vpReplacePoint) {
    if (!mpHandle) return 0;
    typedef decltype(pKF->GetPose()) SE3T;
    const SE3T Tcw(Scw.rotationMatrix(), Scw.translation() / Scw.scale());
    const auto Ow = Tcw.inverse().translation();
    const std::set<MapPointT*> spAlreadyFound = pKF->GetMapPoints();
    const int n1 = (int)vpPoints.size(), n2 = pKF->N;
    std::vector<uint8_t> valid(n1, 0), desc1((size_t)n1 * 32, 0);
    std::vector<float> pos((size_t)n1 * 3, 0.f);
    std::vector<int32_t> level1(n1, 0);
    for (int i = 0; i < n1; ++i) {
      MapPointT* pMP = vpPoints[i];
      if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
      const auto p3Dw = pMP->GetWorldPos();
      const auto p3Dc = Tcw * p3Dw;
      const auto PO = p3Dw - Ow;
      const float dist3D = PO.norm();
      if (dist3D < pMP->GetMinDistanceInvariance() || dist3D > pMP->GetMaxDistanceInvariance()) continue;
      const auto Pn = pMP->GetNormal();
      if (PO.dot(Pn) < 0.5 * dist3D) continue;
      valid[i] = 1;
      level1[i] = pMP->PredictScale(dist3D, pKF);
      for (int k = 0; k < 3; ++k) pos[3 * (size_t)i + k] = p3Dc(k);
      const cv::Mat dMP = pMP->GetDescriptor();
      memcpy(&desc1[(size_t)i * 32], dMP.ptr<uint8_t>(), 32);
    }
    std::vector<int32_t> best;
    if (!ProjectSearch(pKF, valid, pos, desc1, level1, th, 0, 50 /* TH_LOW */, best)) return 0;
    int nFused = 0;
    for (int i = 0; i < n1; ++i) {
      if (best[i] < 0) continue;
      MapPointT* pMP = vpPoints[i];
      MapPointT* pMPinKF = pKF->GetMapPoint(best[i]);
      if (pMPinKF) {
        if (!pMPinKF->isBad()) vpReplacePoint[i] = pMPinKF;
      } else {
        pMP->AddObservation(pKF, best[i]);
        pKF->AddMapPoint(pMP, best[i]);
      }
      nFused++;
    }
    (void)n2;
    return nFused;
  }

  // Project MapPoints using a Similarity Transformation and search matches.  Used in loop detection
  // (LoopClosing::FindMatchesByProjection).  ORBmatcher.h:61 / :65, ORBmatcher.cc:427-532 / 534-646.
  template <class KeyFrameT, class Sim3T, class MapPointT>
  int SearchByProjection(KeyFrameT* pKF, Sim3T& Scw, const std::vector<MapPointT*>& vpPoints, std::vector<MapPointT*>& vpMatched,
                         int th, float ratioHamming = 1.0) {
    std::vector<int32_t> match2;
    int nmatches = 0;
    if (!SearchSim3Greedy(pKF, Scw, vpPoints, vpMatched, th, ratioHamming, 0, match2, nmatches)) return 0;
    for (size_t i2 = 0; i2 < match2.size(); ++i2)
      if (match2[i2] >= 0) vpMatched[i2] = vpPoints[match2[i2]];
    return nmatches;
  }
  template <class KeyFrameT, class Sim3T, class MapPointT>
  int SearchByProjection(KeyFrameT* pKF, Sim3T& Scw, const std::vector<MapPointT*>& vpPoints, const std::vector<KeyFrameT*>& vpPointsKFs,
                         std::vector<MapPointT*>& vpMatched, std::vector<KeyFrameT*>& vpMatchedKF, int th, float ratioHamming = 1.0) {
    std::vector<int32_t> match2;
    int nmatches = 0;
    if (!SearchSim3Greedy(pKF, Scw, vpPoints, vpMatched, th, ratioHamming, 2, match2, nmatches)) return 0;
    for (size_t i2 = 0; i2 < match2.size(); ++i2)
      if (match2[i2] >= 0) { vpMatched[i2] = vpPoints[match2[i2]]; vpMatchedKF[i2] = vpPointsKFs[match2[i2]]; }
    return nmatches;
  }

  // Search matches between MapPoints seen in KF1 and KF2 transforming by a Sim3 [s12*R12|t12] (LoopClosing).
  // ORBmatcher.h:79, ORBmatcher.cc:1457-1674: two directed searches on the device, the mutual-agreement pass here.
  template <class KeyFrameT, class MapPointT, class Sim3T>
  int SearchBySim3(KeyFrameT* pKF1, KeyFrameT* pKF2, std::vector<MapPointT*>& vpMatches12, const Sim3T& S12, const float th) {
    if (!mpHandle) return 0;
    const auto T1w = pKF1->GetPose();
    const auto T2w = pKF2->GetPose();
    const Sim3T S21 = S12.inverse();
    const std::vector<MapPointT*> vpMapPoints1 = pKF1->GetMapPointMatches();
    const std::vector<MapPointT*> vpMapPoints2 = pKF2->GetMapPointMatches();
    const int N1 = (int)vpMapPoints1.size(), N2 = (int)vpMapPoints2.size();
    std::vector<bool> vbAlreadyMatched1(N1, false), vbAlreadyMatched2(N2, false);
    for (int i = 0; i < N1; i++) {
      MapPointT* pMP = vpMatches12[i];
      if (pMP) {
        vbAlreadyMatched1[i] = true;
        const int idx2 = std::get<0>(pMP->GetIndexInKeyFrame(pKF2));
        if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
      }
    }
    std::vector<int32_t> vnMatch1, vnMatch2;
    {
      std::vector<uint8_t> valid(N1, 0), desc((size_t)N1 * 32, 0);
      std::vector<float> pos((size_t)N1 * 3, 0.f);
      std::vector<int32_t> level(N1, 0);
      for (int i1 = 0; i1 < N1; i1++) {
        MapPointT* pMP = vpMapPoints1[i1];
        if (!pMP || vbAlreadyMatched1[i1] || pMP->isBad()) continue;
        const auto p3Dw = pMP->GetWorldPos();
        const auto p3Dc1 = T1w * p3Dw;
        const auto p3Dc2 = S21 * p3Dc1;
        const float dist3D = p3Dc2.norm();
        if (dist3D < pMP->GetMinDistanceInvariance() || dist3D > pMP->GetMaxDistanceInvariance()) continue;
        valid[i1] = 1;
        level[i1] = pMP->PredictScale(dist3D, pKF2);
        for (int k = 0; k < 3; ++k) pos[3 * (size_t)i1 + k] = p3Dc2(k);
        const cv::Mat dMP = pMP->GetDescriptor();
        memcpy(&desc[(size_t)i1 * 32], dMP.ptr<uint8_t>(), 32);
      }
      if (!ProjectSearch(pKF2, valid, pos, desc, level, th, 1, 100 /* TH_HIGH */, vnMatch1)) return 0;
    }
    {
      std::vector<uint8_t> valid(N2, 0), desc((size_t)N2 * 32, 0);
      std::vector<float> pos((size_t)N2 * 3, 0.f);
      std::vector<int32_t> level(N2, 0);
      for (int i2 = 0; i2 < N2; i2++) {
        MapPointT* pMP = vpMapPoints2[i2];
        if (!pMP || vbAlreadyMatched2[i2] || pMP->isBad()) continue;
        const auto p3Dw = pMP->GetWorldPos();
        const auto p3Dc2 = T2w * p3Dw;
        const auto p3Dc1 = S12 * p3Dc2;
        const float dist3D = p3Dc1.norm();
        if (dist3D < pMP->GetMinDistanceInvariance() || dist3D > pMP->GetMaxDistanceInvariance()) continue;
        valid[i2] = 1;
        level[i2] = pMP->PredictScale(dist3D, pKF1);
        for (int k = 0; k < 3; ++k) pos[3 * (size_t)i2 + k] = p3Dc1(k);
        const cv::Mat dMP = pMP->GetDescriptor();
        memcpy(&desc[(size_t)i2 * 32], dMP.ptr<uint8_t>(), 32);
      }
      if (!ProjectSearch(pKF1, valid, pos, desc, level, th, 1, 100 /* TH_HIGH */, vnMatch2)) return 0;
    }
    // Check agreement
    int nFound = 0;
    for (int i1 = 0; i1 < N1; i1++) {
      const int idx2 = vnMatch1[i1];
      if (idx2 >= 0) {
        const int idx1 = vnMatch2[idx2];
        if (idx1 == i1) {
          vpMatches12[i1] = vpMapPoints2[idx2];
          nFound++;
        }
      }
    }
    return nFound;
  }

  // Search matches between Frame keypoints and projected MapPoints.  Used to track the local map (Tracking::SearchLocalPoints,
  // Tracking.cc:3370-3450).  ORBmatcher.h:45, ORBmatcher.cc:43-213.  Reads what Frame::isInFrustum left in every MapPoint
  // (mbTrackInView, mTrackProjX / Y / XR, mnTrackScaleLevel, mTrackViewCos, mTrackDepth) and writes F.mvpMapPoints.
  template <class FrameT, class MapPointT>
  int SearchByProjection(FrameT& F, const std::vector<MapPointT*>& vpMapPoints, const float th = 3, const bool bFarPoints = false,
                         const float thFarPoints = 50.0f) {
    if (!mpHandle) return 0;
    if (F.Nleft != -1) {
      std::cerr << "[ORBmatcher] SearchByProjection: fisheye stereo rigs (Nleft != -1) are not covered by the device path" << std::endl;
      return 0;
    }
    const int n1 = (int)vpMapPoints.size(), n2 = F.N;
    std::vector<uint8_t> valid(n1, 0), observed(n1, 0), desc1((size_t)n1 * 32, 0), blocked(n2, 0);
    std::vector<float> proj((size_t)n1 * 3, 0.f), vcos(n1, 0.f), xy2((size_t)n2 * 2);
    std::vector<int32_t> level(n1, 0), oct2(n2);
    for (int i = 0; i < n1; ++i) {
      MapPointT* pMP = vpMapPoints[i];
      if (!pMP->mbTrackInView) continue;  // (mbTrackInViewR only matters for two-camera frames)
      if (bFarPoints && pMP->mTrackDepth > thFarPoints) continue;
      if (pMP->isBad()) continue;
      valid[i] = 1;
      proj[3 * (size_t)i] = pMP->mTrackProjX; proj[3 * (size_t)i + 1] = pMP->mTrackProjY; proj[3 * (size_t)i + 2] = pMP->mTrackProjXR;
      level[i] = pMP->mnTrackScaleLevel;
      vcos[i] = pMP->mTrackViewCos;
      const cv::Mat d = pMP->GetDescriptor();
      memcpy(&desc1[(size_t)i * 32], d.ptr<uint8_t>(), 32);
      observed[i] = pMP->Observations() > 0 ? 1 : 0;
    }
    for (int i = 0; i < n2; ++i) {
      xy2[2 * (size_t)i] = F.mvKeysUn[i].pt.x;
      xy2[2 * (size_t)i + 1] = F.mvKeysUn[i].pt.y;
      oct2[i] = F.mvKeysUn[i].octave;
      blocked[i] = (F.mvpMapPoints[i] && F.mvpMapPoints[i]->Observations() > 0) ? 1 : 0;
    }
    rgbl_local_points_input in{};
    in.n1 = n1; in.valid1 = valid.data(); in.proj1 = proj.data(); in.level1 = level.data(); in.view_cos1 = vcos.data();
    in.mp_desc1 = desc1.data(); in.mp_observed1 = observed.data();
    in.n2 = n2; in.kp2_xy = xy2.data(); in.kp2_octave = oct2.data(); in.uright2 = F.mvuRight.data();
    in.desc2 = F.mDescriptors.template ptr<uint8_t>(); in.blocked2 = blocked.data();
    in.grid[0] = FrameT::mnMinX; in.grid[1] = FrameT::mnMinY; in.grid[2] = FrameT::mnMaxX; in.grid[3] = FrameT::mnMaxY;
    in.grid[4] = FrameT::mfGridElementWidthInv; in.grid[5] = FrameT::mfGridElementHeightInv;
    in.scale_factors = F.mvScaleFactors.data();
    in.n_levels = (int)F.mvScaleFactors.size();
    in.th = th; in.nnratio = mfNNratio;
    in.device2 = rgbl_shim::device_frame_of(F, 0);
    std::vector<int32_t> match2(n2, -1);
    int nmatches = 0;
    if (rgbl_search_local_points(mpHandle, &in, match2.data(), &nmatches) != RGBL_OK) {
      std::cerr << "[ORBmatcher] " << rgbl_last_error() << std::endl;
      return 0;
    }
    for (int i2 = 0; i2 < n2; ++i2)
      if (match2[i2] >= 0) F.mvpMapPoints[i2] = vpMapPoints[match2[i2]];
    return nmatches;
  }

  // Matching for the Map Initialization (only used in the monocular case).  ORBmatcher.h:72, ORBmatcher.cc:648-763
  // (Tracking::MonocularInitialization, Tracking.cc:2525-2526).  Reads F1.mvKeysUn / mDescriptors and F2's, reads and updates
  // vbPrevMatched, fills vnMatches12.
  template <class FrameT>
  int SearchForInitialization(FrameT& F1, FrameT& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12,
                              int windowSize = 10) {
    const int n1 = (int)F1.mvKeysUn.size(), n2 = (int)F2.mvKeysUn.size();
    vnMatches12 = std::vector<int>(n1, -1);
    if (!mpHandle) return 0;
    if (F2.Nleft != -1) {
      std::cerr << "[ORBmatcher] SearchForInitialization: fisheye stereo rigs (Nleft != -1) are not covered by the device path" << std::endl;
      return 0;
    }
    std::vector<float> ang1(n1), xy2((size_t)n2 * 2), ang2(n2), prev((size_t)n1 * 2);
    std::vector<int32_t> oct1(n1), oct2(n2);
    for (int i = 0; i < n1; ++i) {
      oct1[i] = F1.mvKeysUn[i].octave; ang1[i] = F1.mvKeysUn[i].angle;
      prev[2 * (size_t)i] = vbPrevMatched[i].x; prev[2 * (size_t)i + 1] = vbPrevMatched[i].y;
    }
    for (int i = 0; i < n2; ++i) {
      xy2[2 * (size_t)i] = F2.mvKeysUn[i].pt.x; xy2[2 * (size_t)i + 1] = F2.mvKeysUn[i].pt.y;
      oct2[i] = F2.mvKeysUn[i].octave; ang2[i] = F2.mvKeysUn[i].angle;
    }
    rgbl_initialization_input in{};
    in.n1 = n1; in.kp1_octave = oct1.data(); in.kp1_angle = ang1.data(); in.desc1 = F1.mDescriptors.template ptr<uint8_t>();
    in.n2 = n2; in.kp2_xy = xy2.data(); in.kp2_octave = oct2.data(); in.kp2_angle = ang2.data();
    in.desc2 = F2.mDescriptors.template ptr<uint8_t>();
    in.grid[0] = FrameT::mnMinX; in.grid[1] = FrameT::mnMinY; in.grid[2] = FrameT::mnMaxX; in.grid[3] = FrameT::mnMaxY;
    in.grid[4] = FrameT::mfGridElementWidthInv; in.grid[5] = FrameT::mfGridElementHeightInv;
    in.window_size = windowSize; in.nnratio = mfNNratio; in.check_orientation = mbCheckOrientation ? 1 : 0;
    int nmatches = 0;
    std::vector<int32_t> m12(n1, -1);
    if (rgbl_search_for_initialization(mpHandle, &in, prev.data(), m12.data(), &nmatches) != RGBL_OK) {
      std::cerr << "[ORBmatcher] " << rgbl_last_error() << std::endl;
      return 0;
    }
    for (int i = 0; i < n1; ++i) {
      vnMatches12[i] = m12[i];
      if (m12[i] >= 0) vbPrevMatched[i] = F2.mvKeysUn[m12[i]].pt;  // "Update prev matched" (:757-760)
    }
    return nmatches;
  }

  static const int TH_LOW = 50;
  static const int TH_HIGH = 100;
  static const int HISTO_LENGTH = 30;

 protected:
  struct Flat {
    std::vector<float> xy, angle;
    std::vector<int32_t> octave, node_id, node_off, node_feat;
    std::vector<uint8_t> has_mp;
    rgbl_keyframe_view view{};
  };
  // Snapshot of the key-frame state the kernel needs; GetMapPoint() takes the key-frame's own mutex per call,
  // exactly as the reference's inner loops do (KeyFrame.cc:373-377).
  template <class KeyFrameT>
  static void Flatten(KeyFrameT* kf, Flat& f) {
    const int n = kf->N;
    f.xy.resize(2 * (size_t)n); f.angle.resize(n); f.octave.resize(n); f.has_mp.resize(n);
    for (int i = 0; i < n; ++i) {
      f.xy[2 * i] = kf->mvKeysUn[i].pt.x;
      f.xy[2 * i + 1] = kf->mvKeysUn[i].pt.y;
      f.angle[i] = kf->mvKeysUn[i].angle;
      f.octave[i] = kf->mvKeysUn[i].octave;
      f.has_mp[i] = kf->GetMapPoint(i) ? 1 : 0;
    }
    f.node_off.assign(1, 0);
    for (auto it = kf->mFeatVec.begin(); it != kf->mFeatVec.end(); ++it) {  // std::map: node ids ascend
      f.node_id.push_back((int32_t)it->first);
      for (size_t k = 0; k < it->second.size(); ++k) f.node_feat.push_back((int32_t)it->second[k]);
      f.node_off.push_back((int32_t)f.node_feat.size());
    }
    f.view.n = n;
    f.view.desc = kf->mDescriptors.template ptr<uint8_t>();
    f.view.kp_xy = f.xy.data();
    f.view.kp_octave = f.octave.data();
    f.view.kp_angle = f.angle.data();
    f.view.uright = kf->mvuRight.data();
    f.view.has_mappoint = f.has_mp.data();
    f.view.n_nodes = (int)f.node_id.size();
    f.view.node_id = f.node_id.data();
    f.view.node_off = f.node_off.data();
    f.view.node_feat = f.node_feat.data();
    f.view.device = rgbl_shim::device_frame_of(*kf, 0);
  }

  // the part the two SearchByProjection(pKF, Scw, ...) overloads share: the reference's tests on the MapPoint objects, then the
  // greedy search on the device (proj_form 0: Pinhole::project as in :465, 2: invz = 1 / p3Dc(2) as in :575-580)
  template <class KeyFrameT, class Sim3T, class MapPointT>
  bool SearchSim3Greedy(KeyFrameT* pKF, Sim3T& Scw, const std::vector<MapPointT*>& vpPoints, const std::vector<MapPointT*>& vpMatched,
                        int th, float ratioHamming, int proj_form, std::vector<int32_t>& match2, int& nmatches) {
    nmatches = 0;
    if (!mpHandle) return false;
    typedef decltype(pKF->GetPose()) SE3T;
    const SE3T Tcw(Scw.rotationMatrix(), Scw.translation() / Scw.scale());
    const auto Ow = Tcw.inverse().translation();
    std::set<MapPointT*> spAlreadyFound(vpMatched.begin(), vpMatched.end());
    spAlreadyFound.erase(static_cast<MapPointT*>(NULL));
    const int n1 = (int)vpPoints.size(), n2 = pKF->N;
    std::vector<uint8_t> valid(n1, 0), desc1((size_t)n1 * 32, 0), matched(n2, 0);
    std::vector<float> pos((size_t)n1 * 3, 0.f), xy2((size_t)n2 * 2);
    std::vector<int32_t> level1(n1, 0), oct2(n2);
    for (int i = 0; i < n1; ++i) {
      MapPointT* pMP = vpPoints[i];
      if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
      const auto p3Dw = pMP->GetWorldPos();
      const auto p3Dc = Tcw * p3Dw;
      const auto PO = p3Dw - Ow;
      const float dist = PO.norm();
      if (dist < pMP->GetMinDistanceInvariance() || dist > pMP->GetMaxDistanceInvariance()) continue;
      const auto Pn = pMP->GetNormal();
      if (PO.dot(Pn) < 0.5 * dist) continue;
      valid[i] = 1;
      level1[i] = pMP->PredictScale(dist, pKF);
      for (int k = 0; k < 3; ++k) pos[3 * (size_t)i + k] = p3Dc(k);
      const cv::Mat dMP = pMP->GetDescriptor();
      memcpy(&desc1[(size_t)i * 32], dMP.ptr<uint8_t>(), 32);
    }
    for (int i = 0; i < n2; ++i) {
      xy2[2 * (size_t)i] = pKF->mvKeysUn[i].pt.x;
      xy2[2 * (size_t)i + 1] = pKF->mvKeysUn[i].pt.y;
      oct2[i] = pKF->mvKeysUn[i].octave;
      matched[i] = vpMatched[i] ? 1 : 0;
    }
    rgbl_project_search_input in{};
    in.n1 = n1; in.valid1 = valid.data(); in.cam_pos1 = pos.data(); in.mp_desc1 = desc1.data(); in.level1 = level1.data();
    in.n2 = n2; in.kp2_xy = xy2.data(); in.kp2_octave = oct2.data(); in.desc2 = pKF->mDescriptors.template ptr<uint8_t>();
    in.grid[0] = pKF->mnMinX; in.grid[1] = pKF->mnMinY; in.grid[2] = pKF->mnMaxX; in.grid[3] = pKF->mnMaxY;
    in.grid[4] = pKF->mfGridElementWidthInv; in.grid[5] = pKF->mfGridElementHeightInv;
    in.K[0] = pKF->fx; in.K[1] = pKF->fy; in.K[2] = pKF->cx; in.K[3] = pKF->cy;
    in.scale_factors = pKF->mvScaleFactors.data();
    in.n_levels = (int)pKF->mvScaleFactors.size();
    in.th = (float)th;  // radius = th * scale: int times float in the reference
    in.proj_form = proj_form;
    const float limit = 50 /* TH_LOW */ * ratioHamming;  // bestDist <= TH_LOW * ratioHamming, an int against a float
    in.max_dist = limit >= 255.f ? 255 : (limit < 0.f ? -1 : (int)floorf(limit));
    match2.assign(n2, -1);
    if (in.max_dist < 0) return true;
    if (rgbl_search_by_projection_sim3(mpHandle, &in, matched.data(), match2.data(), &nmatches) != RGBL_OK) {
      std::cerr << "[ORBmatcher] " << rgbl_last_error() << std::endl;
      nmatches = 0;
      return false;
    }
    return true;
  }

  // camera-frame points against the features of pKF (rgbl_project_search); false on a device error
  template <class KeyFrameT>
  bool ProjectSearch(KeyFrameT* pKF, const std::vector<uint8_t>& valid, const std::vector<float>& pos, const std::vector<uint8_t>& desc,
                     const std::vector<int32_t>& level, float th, int proj_form, int max_dist, std::vector<int32_t>& best) {
    const int n1 = (int)valid.size(), n2 = pKF->N;
    std::vector<float> xy2((size_t)n2 * 2);
    std::vector<int32_t> oct2(n2);
    for (int i = 0; i < n2; ++i) {
      xy2[2 * (size_t)i] = pKF->mvKeysUn[i].pt.x;
      xy2[2 * (size_t)i + 1] = pKF->mvKeysUn[i].pt.y;
      oct2[i] = pKF->mvKeysUn[i].octave;
    }
    rgbl_project_search_input in{};
    in.n1 = n1; in.valid1 = valid.data(); in.cam_pos1 = pos.data(); in.mp_desc1 = desc.data(); in.level1 = level.data();
    in.n2 = n2; in.kp2_xy = xy2.data(); in.kp2_octave = oct2.data(); in.desc2 = pKF->mDescriptors.template ptr<uint8_t>();
    in.grid[0] = pKF->mnMinX; in.grid[1] = pKF->mnMinY; in.grid[2] = pKF->mnMaxX; in.grid[3] = pKF->mnMaxY;
    in.grid[4] = pKF->mfGridElementWidthInv; in.grid[5] = pKF->mfGridElementHeightInv;
    in.K[0] = pKF->fx; in.K[1] = pKF->fy; in.K[2] = pKF->cx; in.K[3] = pKF->cy;
    in.scale_factors = pKF->mvScaleFactors.data();
    in.n_levels = (int)pKF->mvScaleFactors.size();
    in.th = th; in.proj_form = proj_form; in.max_dist = max_dist;
    best.assign(n1, -1);
    if (rgbl_project_search(mpHandle, &in, best.data(), NULL) != RGBL_OK) {
      std::cerr << "[ORBmatcher] " << rgbl_last_error() << std::endl;
      return false;
    }
    return true;
  }

  template <class FeatVecT>
  static void FlattenFeatVec(const FeatVecT& fv, Flat& f) {
    f.node_id.clear(); f.node_feat.clear();
    f.node_off.assign(1, 0);
    for (auto it = fv.begin(); it != fv.end(); ++it) {  // std::map: node ids ascend
      f.node_id.push_back((int32_t)it->first);
      for (size_t k = 0; k < it->second.size(); ++k) f.node_feat.push_back((int32_t)it->second[k]);
      f.node_off.push_back((int32_t)f.node_feat.size());
    }
  }
  static void FillView(Flat& f, int n, const uint8_t* desc, const float* uright) {
    f.view.n = n;
    f.view.desc = desc;
    f.view.kp_xy = f.xy.data();
    f.view.kp_octave = f.octave.data();
    f.view.kp_angle = f.angle.data();
    f.view.uright = uright;
    f.view.has_mappoint = f.has_mp.data();
    f.view.n_nodes = (int)f.node_id.size();
    f.view.node_id = f.node_id.data();
    f.view.node_off = f.node_off.data();
    f.view.node_feat = f.node_feat.data();
  }

  float mfNNratio;
  bool mbCheckOrientation;
  rgbl_matcher* mpHandle = nullptr;
};

}  // namespace ORB_SLAM3
#endif
