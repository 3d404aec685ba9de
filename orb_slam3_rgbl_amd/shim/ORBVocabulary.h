// Drop-in for the two members of ORB_SLAM3::ORBVocabulary (= DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>,
// /root/reference/include/ORBVocabulary.h) that the tracking thread uses: loadFromTextFile (System.cc:115) and
// transform(features, BowVector, FeatureVector, levelsup) (Frame::ComputeBoW, src/Frame.cc:828-835; KeyFrame::ComputeBoW,
// src/KeyFrame.cc:98-107) - the vocabulary tree lives on the device, the per-feature descent runs there.
// transform() is a template over the two container types so that DBoW2's own BowVector / FeatureVector
// (std::map<WordId, WordValue>, std::map<NodeId, std::vector<unsigned>>) are filled directly; the scoring side of the
// vocabulary (score(), used by KeyFrameDatabase) stays with DBoW2.
#ifndef RGBL_ORBVOCABULARY_H
#define RGBL_ORBVOCABULARY_H

#include <iostream>
#include <string>
#include <vector>

#include "../../include/rgbl_frontend.h"
#include "cv_compat.h"

namespace ORB_SLAM3 {

class DeviceORBVocabulary {
 public:
  explicit DeviceORBVocabulary(int device = 0) : mDevice(device) {}
  ~DeviceORBVocabulary() { rgbl_vocabulary_destroy(mpHandle); }
  DeviceORBVocabulary(const DeviceORBVocabulary&) = delete;
  DeviceORBVocabulary& operator=(const DeviceORBVocabulary&) = delete;

  bool loadFromTextFile(const std::string& filename) {
    rgbl_vocabulary_destroy(mpHandle);
    mpHandle = nullptr;
    if (rgbl_vocabulary_load_text(filename.c_str(), mDevice, &mpHandle) != RGBL_OK) {
      std::cerr << rgbl_last_error() << std::endl;
      return false;
    }
    return true;
  }
  bool empty() const { return mpHandle == nullptr; }

  template <class BowVectorT, class FeatureVectorT>
  void transform(const std::vector<cv::Mat>& features, BowVectorT& v, FeatureVectorT& fv, int levelsup) const {
    v.clear();
    fv.clear();
    if (!mpHandle || features.empty()) return;
    const int n = (int)features.size();
    std::vector<uint8_t> desc((size_t)n * 32);
    for (int i = 0; i < n; ++i) memcpy(&desc[(size_t)i * 32], features[i].template ptr<uint8_t>(), 32);
    std::vector<uint32_t> wid(n), nid(n), nfeat(n);
    std::vector<double> wval(n);
    std::vector<int32_t> noff(n + 1);
    int nw = 0, nn = 0;
    if (rgbl_bow_transform(mpHandle, desc.data(), n, levelsup, wid.data(), wval.data(), n, &nw, nid.data(), noff.data(), nfeat.data(),
                           n, &nn) != RGBL_OK) {
      std::cerr << "[ORBVocabulary] " << rgbl_last_error() << std::endl;
      return;
    }
    for (int i = 0; i < nw; ++i) v.insert(v.end(), typename BowVectorT::value_type(wid[i], wval[i]));
    for (int i = 0; i < nn; ++i) {
      typename FeatureVectorT::iterator it = fv.insert(fv.end(), typename FeatureVectorT::value_type(nid[i], typename FeatureVectorT::mapped_type()));
      it->second.assign(nfeat.begin() + noff[i], nfeat.begin() + noff[i + 1]);
    }
  }

 protected:
  rgbl_vocabulary* mpHandle = nullptr;
  int mDevice;
};

}  // namespace ORB_SLAM3
#endif
