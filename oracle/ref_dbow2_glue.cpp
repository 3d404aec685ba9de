// ref_dbow2_glue.cpp — C entry points around the reference's OWN vendored DBoW2 (Thirdparty/DBoW2/DBoW2: TemplatedVocabulary.h,
// FORB.cpp, BowVector.cpp, FeatureVector.cpp, ScoringObject.cpp), compiled unmodified where they lie against oracle/cvcompat.
// TEST INFRASTRUCTURE (oracle/Makefile target `ref`).
#include <string.h>

#include <chrono>
#include <vector>

#include "Thirdparty/DBoW2/DBoW2/FORB.h"
#include "Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h"

typedef DBoW2::TemplatedVocabulary<DBoW2::FORB::TDescriptor, DBoW2::FORB> ORBVocabulary;  // include/ORBVocabulary.h

static double g_transform_seconds = 0.0;  // wall time of the last transform() call itself (bench.py)

extern "C" {

double ref_voc_last_call_seconds() { return g_transform_seconds; }

void* ref_voc_load_text(const char* path) {
  ORBVocabulary* v = new ORBVocabulary();
  if (!v->loadFromTextFile(path)) { delete v; return nullptr; }
  return v;
}
void ref_voc_destroy(void* h) { delete static_cast<ORBVocabulary*>(h); }

// Frame::ComputeBoW (src/Frame.cc:828-835): mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4).
// Outputs in map order: BowVector as (word id, value), FeatureVector as CSR (node id, offsets, feature indices).
int ref_voc_transform(void* h, const uint8_t* desc, int n, int levelsup, int cap_words, unsigned* word_id, double* word_val,
                      int* n_words, int cap_nodes, unsigned* node_id, int* node_off, unsigned* node_feat, int* n_nodes) {
  ORBVocabulary* v = static_cast<ORBVocabulary*>(h);
  std::vector<cv::Mat> features(n);
  for (int i = 0; i < n; ++i) {
    features[i] = cv::Mat(1, 32, CV_8U);
    memcpy(features[i].data, desc + (size_t)i * 32, 32);
  }
  DBoW2::BowVector bow;
  DBoW2::FeatureVector fv;
  {
    const auto t0 = std::chrono::steady_clock::now();
    v->transform(features, bow, fv, levelsup);
    g_transform_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  *n_words = (int)bow.size();
  *n_nodes = (int)fv.size();
  if ((int)bow.size() > cap_words || (int)fv.size() > cap_nodes) return -1;
  int k = 0;
  for (const auto& kv : bow) { word_id[k] = kv.first; word_val[k] = kv.second; ++k; }
  k = 0;
  int off = 0;
  for (const auto& kv : fv) {
    node_id[k] = kv.first;
    node_off[k] = off;
    for (unsigned f : kv.second) node_feat[off++] = f;
    ++k;
  }
  node_off[k] = off;
  return 0;
}

}  // extern "C"
