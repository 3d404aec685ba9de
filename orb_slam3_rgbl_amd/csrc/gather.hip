// gather.hip — the one exchange of the multi-GPU front end: the two-phase, variable-length gather of the
// per-frame results to rank 0, as C-ABI entry points over RCCL, on a stream this library controls.
//   phase 1  ncclAllGather of the per-frame keypoint counts of every rank (batch int32 each), copied into page-locked host
//            memory behind an event - the host reads them one step later, never waits for the step it has just queued;
//   phase 2  the compacted 68-byte records (records.hip), point to point: every non-root posts ONE ncclSend of its exact size,
//            the root one ncclRecv per peer, all inside one ncclGroup - on MI355X that is one xGMI link per peer, in parallel
//            (a ring collective would be bound by a single link and would move capacity-padded arrays).
// There is nothing to cite in the reference: it runs on one CPU.  What this serves is `north_star`'s "host stays C++ calling
// through a thin C-ABI ... RCCL over xGMI only for the final keypoint/descriptor gather" - a C++ host in the style of
// Examples/RGB-L/rgbl_kitti.cc:87-125 cannot use a Python process group.
//
// RCCL is not linked: it is looked up at the first rgbl_comm_* call (an RCCL already mapped into the process - PyTorch ships
// one - is reused, otherwise librccl.so.1 is dlopen'ed), so that the single-GPU library loads and runs without it.
// The CPU SIMT-emulation build (RGBL_EMU, tests only) binds the same calls to tests/emu/nccl_emu.cpp, a file-system mailbox
// with RCCL's signatures, so that this very file runs with two ranks on a GPU-less machine.
#include "common.h"

#include <stdlib.h>
#include <string.h>

#include <mutex>

#include <rccl/rccl.h>   // types and prototypes only (tests/emu/rccl/rccl.h under RGBL_EMU); no symbol of it is linked in the product

#ifndef RGBL_EMU
#include <dlfcn.h>
#include <link.h>
#endif

namespace rgbl {
namespace {

constexpr int kRecordBytes = 68;

struct NcclApi {
  decltype(&ncclGetVersion) GetVersion = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommDestroy) CommAbort = nullptr;   // ncclCommAbort has ncclCommDestroy's signature; optional (null: no abort on errors)
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  bool ok = false;
  std::string where, why;
};

#ifndef RGBL_EMU
int find_loaded_rccl(struct dl_phdr_info* info, size_t, void* data) {
  if (info->dlpi_name && strstr(info->dlpi_name, "librccl.so")) { *static_cast<std::string*>(data) = info->dlpi_name; return 1; }
  return 0;
}
#endif

NcclApi& nccl() {
  static NcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
#ifdef RGBL_EMU
    api.GetVersion = &ncclGetVersion; api.GetUniqueId = &ncclGetUniqueId; api.CommInitRank = &ncclCommInitRank;
    api.CommDestroy = &ncclCommDestroy; api.GetErrorString = &ncclGetErrorString; api.AllGather = &ncclAllGather;
    api.Send = &ncclSend; api.Recv = &ncclRecv; api.GroupStart = &ncclGroupStart; api.GroupEnd = &ncclGroupEnd;
    api.CommAbort = &ncclCommAbort;
    api.ok = true; api.where = "tests/emu/nccl_emu.cpp";
#else
    void* h = nullptr;
    std::string loaded;
    // 1. an RCCL that is already part of the process (PyTorch's bundled one): a second copy would bring a second set of
    //    kernels and, worse, might bind to another HIP runtime than the one serving this process
    dl_iterate_phdr(find_loaded_rccl, &loaded);
    if (!loaded.empty()) { h = dlopen(loaded.c_str(), RTLD_NOW | RTLD_NOLOAD); if (h) api.where = loaded; }
    // 2. RGBL_RCCL_LIB, then the system's
    const char* names[] = {getenv("RGBL_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      if (h) break;
      if (!n || !*n) continue;
      h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (h) api.where = n; else api.why = dlerror();
    }
    if (!h) return;
    bool all = true;
#define RGBL_SYM(field, name) do { api.field = reinterpret_cast<decltype(api.field)>(dlsym(h, name)); if (!api.field) { all = false; api.why = std::string("missing symbol ") + name; } } while (0)
    RGBL_SYM(GetVersion, "ncclGetVersion"); RGBL_SYM(GetUniqueId, "ncclGetUniqueId"); RGBL_SYM(CommInitRank, "ncclCommInitRank");
    RGBL_SYM(CommDestroy, "ncclCommDestroy"); RGBL_SYM(GetErrorString, "ncclGetErrorString"); RGBL_SYM(AllGather, "ncclAllGather");
    RGBL_SYM(Send, "ncclSend"); RGBL_SYM(Recv, "ncclRecv"); RGBL_SYM(GroupStart, "ncclGroupStart"); RGBL_SYM(GroupEnd, "ncclGroupEnd");
#undef RGBL_SYM
    api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(dlsym(h, "ncclCommAbort"));
    api.ok = all;
#endif
  });
  return api;
}

int need_nccl() {
  NcclApi& a = nccl();
  if (!a.ok) { set_error("RCCL is not available: %s (set RGBL_RCCL_LIB to a librccl.so)", a.why.empty() ? "no librccl.so found" : a.why.c_str()); return RGBL_ERR_COMM; }
  return RGBL_OK;
}

#define RGBL_NCCL(expr)                                                                                      \
  do {                                                                                                       \
    ncclResult_t r__ = (expr);                                                                               \
    if (r__ != ncclSuccess) {                                                                                \
      rgbl::set_error("%s failed: %s (%s:%d)", #expr, nccl().GetErrorString(r__), __FILE__, __LINE__);      \
      return RGBL_ERR_COMM;                                                                                  \
    }                                                                                                        \
  } while (0)

}  // namespace
}  // namespace rgbl

using namespace rgbl;

struct rgbl_comm {
  ncclComm_t comm = nullptr;
  int world = 1, rank = 0, device = 0, version = 0;
  // Lifetime: gather handles hold a reference.  rgbl_comm_destroy with gathers still alive only marks the handle;
  // the communicator goes with the last rgbl_gather_destroy - a gather never sees a dangling pointer.
  // lifetime: gather handles hold the communicator (users); rgbl_comm_destroy marks it released, the last one out frees it.
  // One mutex for both fields: gather_create / gather_destroy / comm_destroy may come from different threads.
  std::mutex mu;
  int users = 0;
  bool released = false;   // rgbl_comm_destroy was called
  bool aborted = false;    // a failed exchange aborted the communicator (ncclCommAbort): every later call returns RGBL_ERR_COMM
};

struct rgbl_gather {
  rgbl_comm* comm = nullptr;   // null: one rank, no RCCL
  int world = 1, rank = 0, device = 0;
  int batch = 0, cap = 0, slots = 0;
  bool loopback = false;
  bool failed = false;         // an exchange failed half-way (peers may be blocked): the handle refuses further collectives
  hipStream_t stream = nullptr, own_stream = nullptr;
  size_t slot_bytes = 0;       // batch * cap * 68
  // per slot
  std::vector<uint8_t*> d_send;
  std::vector<long long*> d_offsets;
  std::vector<int32_t*> d_counts;       // this rank's per-frame counts, copied out of the step's output set
  std::vector<int32_t*> d_all_counts;   // world x batch, gathered on the device
  std::vector<int32_t*> h_counts;       // the same in page-locked host memory, valid behind ev_counts
  std::vector<hipEvent_t> ev_counts;
  std::vector<char> packed;             // slot holds a packed step whose counts are on their way
  int* d_overflow = nullptr;
  int* h_overflow = nullptr;
  // root: two banks of one receive buffer per rank; what the last exchange left behind
  uint8_t* d_recv[2] = {nullptr, nullptr};
  long long n_exchanged = 0;
  int last_bank = -1;
  std::vector<long long> last_total;    // records per rank
  std::vector<int32_t> last_counts;     // world x batch
  std::vector<void*> dev_allocs, host_allocs;
};

namespace {
template <class T>
int galloc(rgbl_gather* g, T** p, size_t count) {
  RGBL_HIP(hipMalloc(p, std::max<size_t>(count, 1) * sizeof(T)));
  g->dev_allocs.push_back((void*)*p);
  RGBL_HIP(hipMemset(*p, 0, std::max<size_t>(count, 1) * sizeof(T)));
  return RGBL_OK;
}
template <class T>
int halloc(rgbl_gather* g, T** p, size_t count) {
  RGBL_HIP(hipHostMalloc(reinterpret_cast<void**>(p), std::max<size_t>(count, 1) * sizeof(T), hipHostMallocDefault));
  g->host_allocs.push_back((void*)*p);
  memset(*p, 0, std::max<size_t>(count, 1) * sizeof(T));
  return RGBL_OK;
}
}  // namespace

extern "C" {

int rgbl_comm_available(void) { return nccl().ok ? 1 : 0; }

int rgbl_comm_unique_id(uint8_t id[RGBL_COMM_ID_BYTES]) {
  static_assert(sizeof(ncclUniqueId) == RGBL_COMM_ID_BYTES, "ncclUniqueId is 128 opaque bytes");
  if (!id) { set_error("null argument"); return RGBL_ERR_INVALID; }
  RGBL_TRY(need_nccl());
  ncclUniqueId u;
  RGBL_NCCL(nccl().GetUniqueId(&u));
  memcpy(id, &u, sizeof u);
  return RGBL_OK;
}

int rgbl_comm_create(const uint8_t id[RGBL_COMM_ID_BYTES], int world, int rank, int device, rgbl_comm** out) {
  if (!id || !out || world < 1 || rank < 0 || rank >= world) { set_error("rgbl_comm_create: id, 0 <= rank < world"); return RGBL_ERR_INVALID; }
  *out = nullptr;
  if (device < 0 || device >= rgbl_device_count()) { set_error("no usable HIP device %d (this library has no CPU fallback)", device); return RGBL_ERR_NO_DEVICE; }
  RGBL_TRY(need_nccl());
  RGBL_HIP(hipSetDevice(device));
  ncclUniqueId u;
  memcpy(&u, id, sizeof u);
  ncclComm_t c = nullptr;
  RGBL_NCCL(nccl().CommInitRank(&c, world, u, rank));   // collective: returns when every rank has called it
  rgbl_comm* h = new rgbl_comm;
  h->comm = c; h->world = world; h->rank = rank; h->device = device;
  (void)nccl().GetVersion(&h->version);
  *out = h;
  return RGBL_OK;
}

namespace {
void comm_free(rgbl_comm* c) {
  (void)hipSetDevice(c->device);
  if (c->comm && !c->aborted) (void)nccl().CommDestroy(c->comm);
  delete c;
}
}  // namespace

void rgbl_comm_destroy(rgbl_comm* c) {
  if (!c) return;
  bool free_now = false;
  {
    std::lock_guard<std::mutex> lock(c->mu);
    if (c->released) return;          // a second call is valid only while gather handles keep the communicator alive (header)
    c->released = true;
    free_now = c->users == 0;         // otherwise the last rgbl_gather_destroy frees it
  }
  if (free_now) comm_free(c);
}

int rgbl_comm_info(const rgbl_comm* c, int* world, int* rank, int* device, int* rccl_version) {
  if (!c) { set_error("null handle"); return RGBL_ERR_INVALID; }
  if (world) *world = c->world;
  if (rank) *rank = c->rank;
  if (device) *device = c->device;
  if (rccl_version) *rccl_version = c->version;
  return RGBL_OK;
}

int rgbl_gather_create(rgbl_comm* comm, int device, int batch, int cap, int slots, void* hip_stream, rgbl_gather** out) {
  if (!out || batch < 1 || cap < 1 || slots < 1) { set_error("rgbl_gather_create: batch, cap, slots >= 1"); return RGBL_ERR_INVALID; }
  *out = nullptr;
  if (comm && comm->device != device) { set_error("the communicator lives on device %d, not %d", comm->device, device); return RGBL_ERR_INVALID; }
  if (comm && (comm->released || comm->aborted)) { set_error("the communicator was destroyed or aborted"); return RGBL_ERR_COMM; }
  if (device < 0 || device >= rgbl_device_count()) { set_error("no usable HIP device %d (this library has no CPU fallback)", device); return RGBL_ERR_NO_DEVICE; }
  RGBL_HIP(hipSetDevice(device));
  rgbl_gather* g = new rgbl_gather;
  g->comm = comm; g->device = device; g->batch = batch; g->cap = cap; g->slots = slots;
  if (comm) {
    std::lock_guard<std::mutex> lock(comm->mu);
    if (comm->released) { delete g; set_error("the communicator was destroyed"); return RGBL_ERR_COMM; }
    g->world = comm->world; g->rank = comm->rank; ++comm->users;
  }
  g->slot_bytes = (size_t)batch * cap * kRecordBytes;
  const int W = g->world;
  int rc = RGBL_OK;
  g->d_send.resize(slots); g->d_offsets.resize(slots); g->d_counts.resize(slots); g->d_all_counts.resize(slots);
  g->h_counts.resize(slots); g->ev_counts.assign(slots, nullptr); g->packed.assign(slots, 0);
  for (int s = 0; s < slots && rc == RGBL_OK; ++s) {
    rc = galloc(g, &g->d_send[s], g->slot_bytes);
    if (rc == RGBL_OK) rc = galloc(g, &g->d_offsets[s], (size_t)batch + 1);
    if (rc == RGBL_OK) rc = galloc(g, &g->d_counts[s], (size_t)batch);
    if (rc == RGBL_OK) rc = galloc(g, &g->d_all_counts[s], (size_t)W * batch);
    if (rc == RGBL_OK) rc = halloc(g, &g->h_counts[s], (size_t)W * batch);
    if (rc == RGBL_OK && hipEventCreateWithFlags(&g->ev_counts[s], hipEventDisableTiming) != hipSuccess) { set_error("hipEventCreate failed"); rc = RGBL_ERR_HIP; }
  }
  if (rc == RGBL_OK) rc = galloc(g, &g->d_overflow, 1);
  if (rc == RGBL_OK) rc = halloc(g, &g->h_overflow, 1);
  if (rc == RGBL_OK && g->rank == 0)
    for (int b = 0; b < 2 && rc == RGBL_OK; ++b) rc = galloc(g, &g->d_recv[b], (size_t)W * g->slot_bytes);
  if (rc == RGBL_OK) {
    if (hip_stream) g->stream = (hipStream_t)hip_stream;
    else {
      void* st = nullptr;
      rc = rgbl_stream_create_on(device, &st, -1);   // off the critical chain: lowest priority
      g->own_stream = g->stream = (hipStream_t)st;
    }
  }
  g->last_total.assign(W, 0);
  g->last_counts.assign((size_t)W * batch, 0);
  if (rc != RGBL_OK) { rgbl_gather_destroy(g); return rc; }
  *out = g;
  return RGBL_OK;
}

void rgbl_gather_destroy(rgbl_gather* g) {
  if (!g) return;
  (void)hipSetDevice(g->device);
  if (g->stream) (void)hipStreamSynchronize(g->stream);
  for (hipEvent_t e : g->ev_counts) if (e) (void)hipEventDestroy(e);
  for (void* p : g->dev_allocs) (void)hipFree(p);
  for (void* p : g->host_allocs) (void)hipHostFree(p);
  if (g->own_stream) rgbl_stream_destroy((void*)g->own_stream);
  if (g->comm) {
    bool free_now = false;
    {
      std::lock_guard<std::mutex> lock(g->comm->mu);
      free_now = --g->comm->users == 0 && g->comm->released;
    }
    if (free_now) comm_free(g->comm);
  }
  delete g;
}

void* rgbl_gather_stream(rgbl_gather* g) { return g ? (void*)g->stream : nullptr; }

int rgbl_gather_set_loopback(rgbl_gather* g, int enable) {
  if (!g) { set_error("null handle"); return RGBL_ERR_INVALID; }
  if (enable && (!g->comm || g->world != 1)) { set_error("loopback transfers need a one-rank communicator"); return RGBL_ERR_INVALID; }
  g->loopback = enable != 0;
  return RGBL_OK;
}

namespace {
// a collective of this handle may be half posted: nothing this rank could still do would match what the peers wait for
void fail_and_abort(rgbl_gather* g) {
  g->failed = true;
  if (g->comm && !g->comm->aborted && nccl().CommAbort) { (void)nccl().CommAbort(g->comm->comm); g->comm->aborted = true; }
}
}  // namespace

int rgbl_gather_pack(rgbl_gather* g, int slot, const int32_t* d_n, const rgbl_keypoint* d_kp, const uint8_t* d_desc,
                     const float* d_depth, const float* d_uright, void* const* wait_events, int n_wait, void* done_event) {
  if (!g || slot < 0 || slot >= g->slots || !d_n || !d_kp || !d_desc || !d_depth || !d_uright || n_wait < 0 || (n_wait > 0 && !wait_events)) {
    set_error("rgbl_gather_pack: invalid argument");
    return RGBL_ERR_INVALID;
  }
  if (g->failed || (g->comm && g->comm->aborted)) { set_error("rgbl_gather_pack: an earlier exchange failed, the handle is unusable"); return RGBL_ERR_COMM; }
  // a slot is free again once it has been exchanged: packing over a step that never left would lose its records silently and
  // leave this rank one all-gather ahead of the others
  if (g->packed[slot]) { set_error("rgbl_gather_pack: slot %d still holds a packed step (rgbl_gather_exchange it first)", slot); return RGBL_ERR_INVALID; }
  RGBL_HIP(hipSetDevice(g->device));
  hipStream_t s = g->stream;
  for (int i = 0; i < n_wait; ++i)
    if (wait_events[i]) RGBL_HIP(hipStreamWaitEvent(s, (hipEvent_t)wait_events[i], 0));
  RGBL_TRY(rgbl_pack_records_device((void*)s, d_n, d_kp, d_desc, d_depth, d_uright, g->batch, g->cap, 0, (long long)g->batch * g->cap,
                                    g->d_send[slot], g->d_offsets[slot], g->d_overflow));
  RGBL_HIP(hipMemcpyAsync(g->d_counts[slot], d_n, sizeof(int32_t) * g->batch, hipMemcpyDeviceToDevice, s));
  // the step's output set is free again once its records and counts are copied out
  if (done_event) RGBL_HIP(hipEventRecord((hipEvent_t)done_event, s));
  // phase 1.  From the all-gather on a failure cannot be retried: the collective may be posted and the peers sit in it, a
  // second pack of this slot would put this rank one all-gather ahead.  Same treatment as a failed exchange:
  // the handle fails, the communicator is aborted (the peers return with an error instead of hanging).
  const int rc = [&]() -> int {
    if (g->comm) RGBL_NCCL(nccl().AllGather(g->d_counts[slot], g->d_all_counts[slot], (size_t)g->batch, ncclInt32, g->comm->comm, s));
    else RGBL_HIP(hipMemcpyAsync(g->d_all_counts[slot], g->d_counts[slot], sizeof(int32_t) * g->batch, hipMemcpyDeviceToDevice, s));
    RGBL_HIP(hipMemcpyAsync(g->h_counts[slot], g->d_all_counts[slot], sizeof(int32_t) * (size_t)g->world * g->batch, hipMemcpyDeviceToHost, s));
    RGBL_HIP(hipEventRecord(g->ev_counts[slot], s));
    return RGBL_OK;
  }();
  if (rc != RGBL_OK) { fail_and_abort(g); return g->comm ? RGBL_ERR_COMM : rc; }
  g->packed[slot] = 1;
  return RGBL_OK;
}

namespace {
// phase 2 proper; any error return leaves transfers half posted - rgbl_gather_exchange turns that into a failed handle
int exchange_posted(rgbl_gather* g, int slot, const std::vector<long long>& total, int bank) {
  hipStream_t s = g->stream;
  const int W = g->world;
  if (g->rank == 0) {
    uint8_t* base = g->d_recv[bank];
    bool any = false;
    for (int r = 1; r < W; ++r) any = any || total[r] > 0;
    const bool self = g->loopback && total[0] > 0;
    if (any || self) {
      // one group: the transfers are posted together and progress in parallel (an error return inside it still closes the group)
      struct Group {
        bool open = false;
        ~Group() { if (open) (void)nccl().GroupEnd(); }
      } grp;
      RGBL_NCCL(nccl().GroupStart());
      grp.open = true;
      if (self) {
        RGBL_NCCL(nccl().Send(g->d_send[slot], (size_t)total[0] * kRecordBytes, ncclUint8, 0, g->comm->comm, s));
        RGBL_NCCL(nccl().Recv(base, (size_t)total[0] * kRecordBytes, ncclUint8, 0, g->comm->comm, s));
      }
      for (int r = 1; r < W; ++r)
        if (total[r] > 0) RGBL_NCCL(nccl().Recv(base + (size_t)r * g->slot_bytes, (size_t)total[r] * kRecordBytes, ncclUint8, r, g->comm->comm, s));
      grp.open = false;
      RGBL_NCCL(nccl().GroupEnd());
    }
    if (!self && total[0] > 0)
      RGBL_HIP(hipMemcpyAsync(base, g->d_send[slot], (size_t)total[0] * kRecordBytes, hipMemcpyDeviceToDevice, s));
  } else if (total[g->rank] > 0) {
    RGBL_NCCL(nccl().Send(g->d_send[slot], (size_t)total[g->rank] * kRecordBytes, ncclUint8, 0, g->comm->comm, s));
  }
  return RGBL_OK;
}
}  // namespace

int rgbl_gather_exchange(rgbl_gather* g, int slot) {
  if (!g || slot < 0 || slot >= g->slots) { set_error("rgbl_gather_exchange: invalid slot"); return RGBL_ERR_INVALID; }
  if (g->failed || (g->comm && g->comm->aborted)) { set_error("rgbl_gather_exchange: an earlier exchange failed, the handle is unusable"); return RGBL_ERR_COMM; }
  if (!g->packed[slot]) { set_error("rgbl_gather_exchange: slot %d holds no packed step", slot); return RGBL_ERR_INVALID; }
  RGBL_HIP(hipSetDevice(g->device));
  // the only host wait of the gather: the counts of THIS slot - in a streaming run an event of the step before the one just queued
  if (hipEventSynchronize(g->ev_counts[slot]) != hipSuccess) {   // the counts never arrived: the all-gather of this slot is lost
    (void)hipGetLastError();
    set_error("rgbl_gather_exchange: waiting for the counts of slot %d failed", slot);
    fail_and_abort(g);
    return RGBL_ERR_COMM;
  }
  const int W = g->world, B = g->batch;
  std::vector<long long> total(W, 0);
  for (int r = 0; r < W; ++r)
    for (int f = 0; f < B; ++f) total[r] += std::min(std::max(g->h_counts[slot][(size_t)r * B + f], 0), g->cap);
  const int bank = (int)((g->n_exchanged + 1) & 1);
  const int rc = exchange_posted(g, slot, total, bank);
  if (rc != RGBL_OK) {
    // Some transfers may be posted and the peers sit in their matching calls: nothing this rank could still do would match
    // them.  Abort the communicator (it unblocks the peers with an error instead of a hang) and refuse further collectives.
    fail_and_abort(g);
    return rc;
  }
  // book-keeping only once the whole group is posted: a failed exchange flips no bank and publishes no counts
  if (g->rank == 0) {
    ++g->n_exchanged;
    g->last_bank = bank;
    g->last_total = total;
    memcpy(g->last_counts.data(), g->h_counts[slot], sizeof(int32_t) * (size_t)W * B);
  }
  g->packed[slot] = 0;
  return RGBL_OK;
}

int rgbl_gather_sync(rgbl_gather* g) {
  if (!g) { set_error("null handle"); return RGBL_ERR_INVALID; }
  RGBL_HIP(hipSetDevice(g->device));
  RGBL_HIP(hipMemcpyAsync(g->h_overflow, g->d_overflow, sizeof(int), hipMemcpyDeviceToHost, g->stream));
  RGBL_HIP(hipStreamSynchronize(g->stream));
  if (*g->h_overflow) { set_error("record buffer overflow in rgbl_gather_pack (more than batch x cap records)"); return RGBL_ERR_OVERFLOW; }
  return RGBL_OK;
}

int rgbl_gather_result(rgbl_gather* g, int rank, const int32_t** h_counts, const uint8_t** d_records, long long* n_records) {
  if (!g || rank < 0 || rank >= g->world) { set_error("rgbl_gather_result: invalid rank"); return RGBL_ERR_INVALID; }
  if (g->rank != 0 || g->last_bank < 0) { set_error("rgbl_gather_result: only the root holds results, after an exchange"); return RGBL_ERR_INVALID; }
  if (h_counts) *h_counts = g->last_counts.data() + (size_t)rank * g->batch;
  if (d_records) *d_records = g->d_recv[g->last_bank] + (size_t)rank * g->slot_bytes;
  if (n_records) *n_records = g->last_total[rank];
  return RGBL_OK;
}

int rgbl_gather_copy_result(rgbl_gather* g, int rank, void* dst, long long capacity_bytes) {
  const uint8_t* src = nullptr;
  long long n = 0;
  RGBL_TRY(rgbl_gather_result(g, rank, nullptr, &src, &n));
  if (n * kRecordBytes > capacity_bytes || (n > 0 && !dst)) { set_error("rgbl_gather_copy_result: %lld records need %lld bytes", n, n * kRecordBytes); return RGBL_ERR_CAPACITY; }
  RGBL_HIP(hipSetDevice(g->device));
  if (n > 0) RGBL_HIP(hipMemcpyAsync(dst, src, (size_t)n * kRecordBytes, hipMemcpyDefault, g->stream));
  return RGBL_OK;
}

}  // extern "C"
