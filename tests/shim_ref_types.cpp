// Compile-only (tests/test_shim_reference_types.py): every member template of the drop-in ORBmatcher instantiated with the
// stand-in classes that the REFERENCE's own ORBmatcher.cc is compiled against (oracle/cvcompat/orbslam_types.h:
// KeyFrame, Frame, MapPoint, GeometricCamera, Sophus::SE3f / Sim3f, DBoW2::FeatureVector, cv::Mat / KeyPoint) - i.e. with
// the call shapes of the reference's callers (Tracking.cc:2525, 2761, 2890, 3424, 3662, 3701; LocalMapping.cc:412, 466, 1035;
// LoopClosing.cc) rather than with the reduced types of tests/shim_test.cpp.
#include "orbslam_types.h"
#include "ORBmatcher.h"

using namespace ORB_SLAM3;

int instantiate_all(KeyFrame* k1, KeyFrame* k2, Frame& f1, Frame& f2, std::vector<MapPoint*>& mps, std::vector<KeyFrame*>& kfs,
                    Sophus::Sim3f& S, std::vector<cv::Point2f>& prev, std::vector<int>& m12) {
  ORB_SLAM3::ORBmatcher m(0.6f, true);
  std::vector<std::pair<size_t, size_t> > pairs;
  std::set<MapPoint*> found;
  std::vector<MapPoint*> replaced, matched;
  std::vector<KeyFrame*> matched_kf;
  int n = 0;
  n += m.SearchForTriangulation(k1, k2, pairs, false, false);                 // LocalMapping.cc:412
  n += m.SearchByProjection(f1, f2, 7.f, false);                              // Tracking.cc:2890
  n += m.SearchByProjection(f1, mps, 3.f, false, 50.f);                       // Tracking.cc:3424
  n += m.SearchByProjection(f1, k1, found, 15.f, 100);                        // Tracking.cc:3701
  n += m.SearchByBoW(k1, f1, mps);                                            // Tracking.cc:2761
  n += m.SearchByBoW(k1, k2, mps);                                            // LoopClosing
  n += m.Fuse(k1, mps, 3.f, false);                                           // LocalMapping.cc:1035
  n += m.Fuse(k1, S, mps, 4.f, replaced);                                     // LoopClosing::SearchAndFuse
  n += m.SearchByProjection(k1, S, mps, matched, 3, 1.5f);                    // LoopClosing::FindMatchesByProjection
  n += m.SearchByProjection(k1, S, mps, kfs, matched, matched_kf, 3, 1.5f);
  n += m.SearchBySim3(k1, k2, mps, S, 7.5f);                                  // LoopClosing
  n += m.SearchForInitialization(f1, f2, prev, m12, 100);                     // Tracking.cc:2525
  return n;
}
