// shim_threads_test.cpp - the threading contract of the drop-in boundary (SURVEY.md 8(b) "Threading"), exercised the way the
// reference does it:
//   (i)  Frame::Frame (stereo, /root/reference/src/Frame.cc:122-125): two ORBextractor objects called concurrently from two
//        std::threads (`thread threadLeft(&Frame::ExtractORB, this, 0, imLeft, 0, 0); thread threadRight(...); join; join`),
//        `iters` frames in a row;
//   (ii) ORBmatcher used concurrently by the Tracking, LocalMapping and LoopClosing threads: three threads loop
//        SearchByProjection (rgbl_search_by_projection on a pooled handle, as the drop-in class acquires / releases one per
//        object), ORBmatcher::SearchForTriangulation (the drop-in class itself) and rgbl_hamming_bf while a fourth keeps
//        extracting; every thread also provokes an error now and then and must read ITS OWN message from rgbl_last_error().
// Every concurrent result must equal the sequential run of the same call; the sequential results are written to <out.bin>
// for the Python side to hold against the oracle.  Exit code 0 = all equal.
//   shim_threads_test <w> <h> <left.raw> <right.raw> <tri.bin> <proj.bin> <iters> <out.bin> [<nfeatures> <nlevels>]
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "ORBextractor.h"
#include "ORBmatcher.h"
#include "shim_standins.h"

namespace {
struct Extraction {
  int mono = 0;
  std::vector<cv::KeyPoint> keys;
  cv::Mat desc;
  bool operator==(const Extraction& o) const {
    if (mono != o.mono || keys.size() != o.keys.size()) return false;
    if (!keys.empty() && memcmp(keys.data(), o.keys.data(), keys.size() * sizeof(cv::KeyPoint)) != 0) return false;
    return keys.empty() || memcmp(desc.data, o.desc.data, keys.size() * 32) == 0;
  }
};
void extract(ORB_SLAM3::ORBextractor* ex, const cv::Mat* im, Extraction* out) {  // Frame::ExtractORB (Frame.cc:508-515)
  std::vector<int> lap = {0, 0};
  out->mono = (*ex)(*im, cv::Mat(), out->keys, out->desc, lap);
}

struct ProjCase {
  int n1 = 0, n2 = 0, mono = 0, ori = 0;
  float th = 0, hdr[34];
  std::vector<unsigned char> valid, obs, d1, d2;
  std::vector<float> pos, ang1, xy2, ang2, ur2;
  std::vector<int> o1, o2;
  bool load(const char* path) {  // the layout tests/shim_driver.py writes (proj.bin)
    FILE* f = fopen(path, "rb");
    if (!f || !rd(f, &n1, 1) || !rd(f, &n2, 1) || !rd(f, &th, 1) || !rd(f, &mono, 1) || !rd(f, &ori, 1) || !rd(f, hdr, 34)) return false;
    valid.resize(n1); obs.resize(n1); d1.resize((size_t)n1 * 32); d2.resize((size_t)n2 * 32);
    pos.resize((size_t)n1 * 3); ang1.resize(n1); xy2.resize((size_t)n2 * 2); ang2.resize(n2); ur2.resize(n2); o1.resize(n1); o2.resize(n2);
    bool ok = rd(f, valid.data(), n1) && rd(f, pos.data(), (size_t)n1 * 3) && rd(f, d1.data(), (size_t)n1 * 32) && rd(f, obs.data(), n1) &&
              rd(f, o1.data(), n1) && rd(f, ang1.data(), n1) && rd(f, xy2.data(), (size_t)n2 * 2) && rd(f, o2.data(), n2) &&
              rd(f, ang2.data(), n2) && rd(f, ur2.data(), n2) && rd(f, d2.data(), (size_t)n2 * 32);
    fclose(f);
    return ok;
  }
  void fill(rgbl_projection_input* in) const {
    memset(in, 0, sizeof(*in));
    in->n1 = n1; in->valid1 = valid.data(); in->world_pos1 = pos.data(); in->mp_desc1 = d1.data(); in->mp_observed1 = obs.data();
    in->octave1 = o1.data(); in->angle1 = ang1.data();
    in->n2 = n2; in->kp2_xy = xy2.data(); in->kp2_octave = o2.data(); in->kp2_angle = ang2.data(); in->uright2 = ur2.data(); in->desc2 = d2.data();
    memcpy(in->grid, hdr, 6 * sizeof(float));
    memcpy(in->Tcw_q, hdr + 6, 4 * sizeof(float)); memcpy(in->Tcw_t, hdr + 10, 3 * sizeof(float));
    memcpy(in->Tlw_q, hdr + 13, 4 * sizeof(float)); memcpy(in->Tlw_t, hdr + 17, 3 * sizeof(float));
    memcpy(in->K, hdr + 20, 4 * sizeof(float));
    in->mb = hdr[24]; in->mbf = hdr[25];
    in->scale_factors = hdr + 26; in->n_levels = 8;
    in->th = th; in->mono = mono; in->check_orientation = ori;
  }
};

// SearchByProjection the way the drop-in class runs it: a pooled handle per ORBmatcher object
int projection(const ProjCase& c, std::vector<int32_t>* match2, int* nm) {
  rgbl_matcher* h = nullptr;
  int rc = rgbl_matcher_acquire(0, &h);
  if (rc != RGBL_OK) return rc;
  rgbl_projection_input in;
  c.fill(&in);
  match2->assign(c.n2, -2);
  rc = rgbl_search_by_projection(h, &in, match2->data(), nm);
  rgbl_matcher_release(h);
  return rc;
}

bool provoke_own_error(const char* who) {
  // an invalid call: the message must be this thread's, whatever the other threads are doing
  rgbl_matcher* h = nullptr;
  if (rgbl_matcher_acquire(0, &h) != RGBL_OK) return false;
  std::vector<uint8_t> d(64);
  std::vector<int32_t> o(4);
  const int rc = rgbl_hamming_bf(h, d.data(), 1, d.data(), 70000, o.data(), o.data() + 1, o.data() + 2);  // nb > 65535
  const std::string msg = rgbl_last_error();
  rgbl_matcher_release(h);
  if (rc != RGBL_ERR_INVALID || msg.find("invalid argument") == std::string::npos) {
    fprintf(stderr, "%s: rc %d, message '%s'\n", who, rc, msg.c_str());
    return false;
  }
  return true;
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 9) return 2;
  const int w = atoi(argv[1]), h = atoi(argv[2]), iters = atoi(argv[7]);
  cv::Mat left(h, w, CV_8UC1), right(h, w, CV_8UC1);
  FILE* f = fopen(argv[3], "rb");
  if (!f || !rd(f, left.data, (size_t)w * h)) return 3;
  fclose(f);
  f = fopen(argv[4], "rb");
  if (!f || !rd(f, right.data, (size_t)w * h)) return 3;
  fclose(f);
  Camera cam; MapPoint some; KeyFrame kf1, kf2;
  f = fopen(argv[5], "rb");
  if (!f || !rd(f, cam.p, 4) || !load_kf(f, kf1, &cam, &some) || !load_kf(f, kf2, &cam, &some)) return 4;
  fclose(f);
  ProjCase pc;
  if (!pc.load(argv[6])) return 5;
  const int nfeat = argc > 9 ? atoi(argv[9]) : (w >= 1000 ? 2000 : 500), nlevels = argc > 10 ? atoi(argv[10]) : 8;

  // ---- sequential reference results of this very library
  ORB_SLAM3::ORBextractor exL(nfeat, 1.2f, nlevels, 20, 7), exR(nfeat, 1.2f, nlevels, 20, 7);  // Tracking.cc:1281-1287 (stereo thresholds)
  Extraction seqL, seqR;
  extract(&exL, &left, &seqL);
  extract(&exR, &right, &seqR);
  std::vector<std::pair<size_t, size_t> > seq_pairs;
  int seq_tri;
  { ORB_SLAM3::ORBmatcher m(0.6, false); seq_tri = m.SearchForTriangulation(&kf1, &kf2, seq_pairs, false, false); }
  std::vector<int32_t> seq_proj;
  int seq_proj_n = 0;
  if (projection(pc, &seq_proj, &seq_proj_n) != RGBL_OK) { fprintf(stderr, "projection: %s\n", rgbl_last_error()); return 6; }
  const int nbf = (int)seqL.keys.size(), nbt = (int)seqR.keys.size();
  std::vector<int32_t> seq_bi(nbf), seq_bd(nbf), seq_sd(nbf);
  {
    rgbl_matcher* hm = nullptr;
    if (rgbl_matcher_acquire(0, &hm) != RGBL_OK) return 7;
    if (rgbl_hamming_bf(hm, seqL.desc.data, nbf, seqR.desc.data, nbt, seq_bi.data(), seq_bd.data(), seq_sd.data()) != RGBL_OK) return 7;
    rgbl_matcher_release(hm);
  }
  FILE* out = fopen(argv[8], "wb");
  if (!out) return 8;
  for (const Extraction* e : {&seqL, &seqR}) {
    const int nk = (int)e->keys.size();
    wr(out, &e->mono, 1); wr(out, &nk, 1); wr(out, e->keys.data(), nk); wr(out, e->desc.data, (size_t)nk * 32);
  }
  const int np = (int)seq_pairs.size();
  wr(out, &seq_tri, 1); wr(out, &np, 1);
  for (auto& pr : seq_pairs) { int a = (int)pr.first, b = (int)pr.second; wr(out, &a, 1); wr(out, &b, 1); }
  wr(out, &seq_proj_n, 1); wr(out, &pc.n2, 1); wr(out, seq_proj.data(), pc.n2);
  wr(out, &nbf, 1); wr(out, seq_bi.data(), nbf); wr(out, seq_bd.data(), nbf); wr(out, seq_sd.data(), nbf);
  fclose(out);

  std::atomic<int> bad(0);
  // ---- (i) Frame.cc:122-125, `iters` stereo frames
  for (int it = 0; it < iters; ++it) {
    Extraction l, r;
    std::thread threadLeft(extract, &exL, &left, &l);
    std::thread threadRight(extract, &exR, &right, &r);
    threadLeft.join();
    threadRight.join();
    if (!(l == seqL) || !(r == seqR)) { fprintf(stderr, "stereo extraction %d differs from the sequential run\n", it); ++bad; }
  }
  // ---- (ii) three matcher threads + an extracting thread
  const int rounds = iters;
  std::thread tTrack([&] {
    for (int it = 0; it < rounds; ++it) {
      std::vector<int32_t> m2;
      int nm = 0;
      if (projection(pc, &m2, &nm) != RGBL_OK || nm != seq_proj_n || m2 != seq_proj) { fprintf(stderr, "SearchByProjection %d differs\n", it); ++bad; }
      if (it % 3 == 0 && !provoke_own_error("tracking thread")) ++bad;
    }
  });
  std::thread tMap([&] {
    for (int it = 0; it < rounds; ++it) {
      ORB_SLAM3::ORBmatcher m(0.6, false);   // function-local, as in LocalMapping::CreateNewMapPoints (LocalMapping.cc:412)
      std::vector<std::pair<size_t, size_t> > pairs;
      const int nm = m.SearchForTriangulation(&kf1, &kf2, pairs, false, false);
      if (nm != seq_tri || pairs != seq_pairs) { fprintf(stderr, "SearchForTriangulation %d differs\n", it); ++bad; }
      if (it % 4 == 1 && !provoke_own_error("mapping thread")) ++bad;
    }
  });
  std::thread tLoop([&] {
    rgbl_matcher* hm = nullptr;
    if (rgbl_matcher_create(0, &hm) != RGBL_OK) { ++bad; return; }
    std::vector<int32_t> bi(nbf), bd(nbf), sd(nbf);
    for (int it = 0; it < rounds; ++it) {
      if (rgbl_hamming_bf(hm, seqL.desc.data, nbf, seqR.desc.data, nbt, bi.data(), bd.data(), sd.data()) != RGBL_OK || bi != seq_bi || bd != seq_bd ||
          sd != seq_sd) { fprintf(stderr, "rgbl_hamming_bf %d differs\n", it); ++bad; }
      if (it % 5 == 2 && !provoke_own_error("loop-closing thread")) ++bad;
    }
    rgbl_matcher_destroy(hm);
  });
  std::thread tExtract([&] {
    for (int it = 0; it < rounds; ++it) {
      Extraction l;
      extract(&exL, &left, &l);
      if (!(l == seqL)) { fprintf(stderr, "extraction beside the matchers %d differs\n", it); ++bad; }
    }
  });
  tTrack.join(); tMap.join(); tLoop.join(); tExtract.join();
  printf("threads ok: %d stereo frames from two threads, %d rounds of three matcher threads + one extractor thread, pool holds %d idle handle(s), %d mismatches\n",
         iters, rounds, rgbl_matcher_pool_size(), bad.load());
  return bad.load() == 0 ? 0 : 1;
}
