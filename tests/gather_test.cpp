// gather_test.cpp - the multi-GPU gather driven from C++ through the C ABI alone (include/rgbl_frontend.h), the way a C++
// host in the style of the reference's Examples/RGB-L/rgbl_kitti.cc:87-125 would: one process per GPU, rank 0 draws the
// RCCL unique id and hands it to the others through a file, every rank packs its per-frame results each "step" and rank 0
// receives all of them (two-phase variable-length gather).  The per-frame results are synthetic (seeded by
// rank and step): this program tests the exchange, the kernels in front of it have their own tests.
//
//   gather_test <world> <rank> <id file> <step|final> <steps> [device]
//
// Built twice by tests/test_gather_cpp.py: against the CPU SIMT-emulation library (GATHER_TEST_EMU: "device" memory is host
// memory, RCCL is tests/emu/nccl_emu.cpp) and, for `-m gpu`, against the product library + libamdhip64 (real RCCL).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include <string>
#include <vector>

#include "rgbl_frontend.h"

#ifdef GATHER_TEST_EMU
static void* dev_alloc(size_t n) { return calloc(n ? n : 1, 1); }
static void dev_free(void* p) { free(p); }
static void h2d(void* d, const void* s, size_t n) { memcpy(d, s, n); }
static void d2h(void* d, const void* s, size_t n) { memcpy(d, s, n); }
static void set_device(int) {}
#else
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
static void* dev_alloc(size_t n) { void* p = nullptr; if (hipMalloc(&p, n ? n : 1) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); exit(2); } return p; }
static void dev_free(void* p) { (void)hipFree(p); }
static void h2d(void* d, const void* s, size_t n) { if (hipMemcpy(d, s, n, hipMemcpyHostToDevice) != hipSuccess) exit(2); }
static void d2h(void* d, const void* s, size_t n) { if (hipMemcpy(d, s, n, hipMemcpyDeviceToHost) != hipSuccess) exit(2); }
static void set_device(int d) { (void)hipSetDevice(d); }
#endif

#define CHECK(expr)                                                                             \
  do {                                                                                          \
    int rc__ = (expr);                                                                          \
    if (rc__ != RGBL_OK) { fprintf(stderr, "%s -> %d: %s\n", #expr, rc__, rgbl_last_error()); return 1; } \
  } while (0)

static const int kBatch = 6, kCap = 40, kRecord = 68;

struct StepData {
  std::vector<int32_t> n;
  std::vector<rgbl_keypoint> kp;
  std::vector<uint8_t> desc;
  std::vector<float> depth, uright;
};

static uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

// what rank `rank` "computed" in step `step`: counts between 0 and cap (frame 1 always empty, frame 2 always full)
static StepData make_step(int rank, int step) {
  StepData d;
  uint32_t s = 12345u + 1000u * (uint32_t)rank + 17u * (uint32_t)step;
  d.n.resize(kBatch); d.kp.resize((size_t)kBatch * kCap); d.desc.resize((size_t)kBatch * kCap * 32);
  d.depth.resize((size_t)kBatch * kCap); d.uright.resize((size_t)kBatch * kCap);
  for (int f = 0; f < kBatch; ++f) d.n[f] = f == 1 ? 0 : f == 2 ? kCap : (int)(lcg(s) % (kCap + 1));
  for (size_t i = 0; i < d.kp.size(); ++i) {
    rgbl_keypoint& k = d.kp[i];
    k.x = (float)(lcg(s) % 1241); k.y = (float)(lcg(s) % 376); k.size = 31.f; k.angle = (float)(lcg(s) % 360);
    k.response = (float)(lcg(s) % 255); k.octave = (int)(lcg(s) % 8); k.class_id = -1;
    d.depth[i] = (float)(lcg(s) % 8000) * 0.01f; d.uright[i] = k.x - 3.f;
  }
  for (size_t i = 0; i < d.desc.size(); ++i) d.desc[i] = (uint8_t)lcg(s);
  return d;
}

static std::vector<uint8_t> expected_records(const StepData& d, std::vector<int32_t>* counts) {
  std::vector<uint8_t> out;
  counts->assign(d.n.begin(), d.n.end());
  for (int f = 0; f < kBatch; ++f)
    for (int i = 0; i < d.n[f]; ++i) {
      const size_t j = (size_t)f * kCap + i;
      uint8_t rec[kRecord];
      memcpy(rec, &d.kp[j], 28); memcpy(rec + 28, &d.desc[j * 32], 32); memcpy(rec + 60, &d.depth[j], 4); memcpy(rec + 64, &d.uright[j], 4);
      out.insert(out.end(), rec, rec + kRecord);
    }
  return out;
}

int main(int argc, char** argv) {
  if (argc < 6) { fprintf(stderr, "usage: gather_test <world> <rank> <id file> <step|final> <steps> [device]\n"); return 2; }
  const int world = atoi(argv[1]), rank = atoi(argv[2]), steps = atoi(argv[5]);
  const std::string id_file = argv[3];
  const bool streaming = strcmp(argv[4], "step") == 0;
  const int device = argc > 6 ? atoi(argv[6]) : 0;
  static_assert(sizeof(rgbl_keypoint) == 28, "cv::KeyPoint layout");
  if (rgbl_device_count() <= device) { fprintf(stderr, "no device %d\n", device); return 3; }
  set_device(device);

  // ---- bootstrap: the unique id travels through a file (any out-of-band channel will do: MPI_Bcast, a socket, ...)
  uint8_t id[RGBL_COMM_ID_BYTES];
  if (rank == 0) {
    CHECK(rgbl_comm_unique_id(id));
    const std::string tmp = id_file + ".part";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) return 4;
    fclose(f);
    if (rename(tmp.c_str(), id_file.c_str()) != 0) return 4;
  } else {
    FILE* f = nullptr;
    const timespec nap = {0, 5000000};
    for (int i = 0; i < 20000 && !(f = fopen(id_file.c_str(), "rb")); ++i) nanosleep(&nap, nullptr);
    if (!f || fread(id, 1, sizeof id, f) != sizeof id) { fprintf(stderr, "no unique id\n"); return 4; }
    fclose(f);
  }
  rgbl_comm* comm = nullptr;
  CHECK(rgbl_comm_create(id, world, rank, device, &comm));
  int cw = 0, cr = 0, cd = 0, ver = 0;
  CHECK(rgbl_comm_info(comm, &cw, &cr, &cd, &ver));
  if (cw != world || cr != rank || cd != device) return 5;

  const int slots = streaming ? 2 : steps;
  rgbl_gather* g = nullptr;
  CHECK(rgbl_gather_create(comm, device, kBatch, kCap, slots, nullptr, &g));
  if (world == 1) CHECK(rgbl_gather_set_loopback(g, 1));   // the point-to-point path with one rank: send to itself

  // two output sets, like a pipeline that lets step k + 1 compute while step k's records travel
  void *d_n[2], *d_kp[2], *d_desc[2], *d_depth[2], *d_ur[2], *done[2];
  for (int i = 0; i < 2; ++i) {
    d_n[i] = dev_alloc(sizeof(int32_t) * kBatch); d_kp[i] = dev_alloc(sizeof(rgbl_keypoint) * kBatch * kCap);
    d_desc[i] = dev_alloc((size_t)kBatch * kCap * 32); d_depth[i] = dev_alloc(sizeof(float) * kBatch * kCap);
    d_ur[i] = dev_alloc(sizeof(float) * kBatch * kCap);
    CHECK(rgbl_event_create(&done[i]));
  }
  bool ok = true;
  long long records_seen = 0;
  auto verify = [&](int step) -> int {   // root: what the last exchange left behind == what every rank packed in `step`
    CHECK(rgbl_gather_sync(g));
    for (int r = 0; r < world; ++r) {
      const int32_t* counts = nullptr; const uint8_t* d_rec = nullptr; long long n = 0;
      CHECK(rgbl_gather_result(g, r, &counts, &d_rec, &n));
      std::vector<int32_t> ecounts;
      const std::vector<uint8_t> want = expected_records(make_step(r, step), &ecounts);
      std::vector<uint8_t> got((size_t)n * kRecord);
      if (n) d2h(got.data(), d_rec, got.size());
      const bool same = (size_t)n * kRecord == want.size() && memcmp(counts, ecounts.data(), sizeof(int32_t) * kBatch) == 0 &&
                        (want.empty() || memcmp(got.data(), want.data(), want.size()) == 0);
      if (!same) fprintf(stderr, "step %d rank %d: %lld records, expected %zu\n", step, r, n, want.size() / kRecord);
      ok = ok && same;
      records_seen += n;
    }
    return RGBL_OK;
  };
  int pending = -1, pending_step = -1;
  for (int k = 0; k < steps; ++k) {
    const int o = k & 1, slot = k % slots;
    const StepData d = make_step(rank, k);
    // (a real pipeline's kernels write these; the set is free again once `done` of its previous use has fired - the blocking
    //  copies below are ordered behind it by rgbl_gather_sync in verify / the stream order of the emulation)
    if (k >= 2) CHECK(rgbl_gather_sync(g));
    h2d(d_n[o], d.n.data(), sizeof(int32_t) * kBatch); h2d(d_kp[o], d.kp.data(), sizeof(rgbl_keypoint) * d.kp.size());
    h2d(d_desc[o], d.desc.data(), d.desc.size()); h2d(d_depth[o], d.depth.data(), sizeof(float) * d.depth.size());
    h2d(d_ur[o], d.uright.data(), sizeof(float) * d.uright.size());
    if (streaming && pending >= 0) {   // step k - 1 travels while step k "computes"
      CHECK(rgbl_gather_exchange(g, pending));
      if (rank == 0) CHECK(verify(pending_step));
    }
    CHECK(rgbl_gather_pack(g, slot, (const int32_t*)d_n[o], (const rgbl_keypoint*)d_kp[o], (const uint8_t*)d_desc[o],
                           (const float*)d_depth[o], (const float*)d_ur[o], nullptr, 0, done[o]));
    pending = slot; pending_step = k;
  }
  if (streaming) {
    CHECK(rgbl_gather_exchange(g, pending));
    if (rank == 0) CHECK(verify(pending_step));
  } else {
    for (int k = 0; k < steps; ++k) {
      CHECK(rgbl_gather_exchange(g, k % slots));
      if (rank == 0) CHECK(verify(k));
    }
  }
  CHECK(rgbl_gather_sync(g));
  // an exchange of a slot that holds nothing is an error, not a hang
  if (rgbl_gather_exchange(g, 0) != RGBL_ERR_INVALID) ok = false;
  rgbl_gather_destroy(g);
  rgbl_comm_destroy(comm);
  for (int i = 0; i < 2; ++i) { dev_free(d_n[i]); dev_free(d_kp[i]); dev_free(d_desc[i]); dev_free(d_depth[i]); dev_free(d_ur[i]); rgbl_event_destroy(done[i]); }
  if (rank == 0) printf("%s world %d mode %s steps %d records %lld rccl %d\n", ok ? "GATHER_CPP_OK" : "GATHER_CPP_MISMATCH", world, argv[4], steps, records_seen, ver);
  return ok ? 0 : 1;
}
