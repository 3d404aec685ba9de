// nccl_emu.cpp - TEST INFRASTRUCTURE ONLY.  RCCL's point-to-point and all-gather calls as a mailbox in the file system, for
// the CPU SIMT-emulation build of orb_slam3_rgbl_amd/csrc/gather.hip: every message is one file
//   $TMPDIR/rgbl_nccl_emu_<unique id>/<kind>_<src>_<dst>_<sequence number>
// written under a temporary name and renamed (atomic on POSIX), read by polling.  The emulator's streams are synchronous, so an
// operation happens when it is called: sends never block, receives wait (with a timeout) - inside ncclGroupStart / ncclGroupEnd
// the operations are collected and the sends of the group are executed before its receives, which is the one ordering
// guarantee of a real group that the gather relies on (a rank's send to itself next to the matching receive).
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <string>
#include <vector>

#include "rccl/rccl.h"

struct ncclComm {
  std::string dir;
  int world = 1, rank = 0;
  std::vector<unsigned long> send_seq, recv_seq;   // per peer
  unsigned long gather_seq = 0;
};

namespace {
struct PendingOp { bool send; void* buf; size_t bytes; int peer; ncclComm* comm; };
thread_local int g_group_depth = 0;
thread_local std::vector<PendingOp> g_pending;

size_t elem_size(ncclDataType_t t) { return t == ncclInt32 ? 4 : 1; }

bool write_file(const std::string& path, const void* data, size_t bytes) {
  const std::string tmp = path + ".part";
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) return false;
  const bool ok = bytes == 0 || fwrite(data, 1, bytes, f) == bytes;
  fclose(f);
  return ok && rename(tmp.c_str(), path.c_str()) == 0;
}

bool read_file(const std::string& path, void* data, size_t bytes, double timeout_s = 120.0) {
  const timespec nap = {0, 2000000};   // 2 ms
  for (double waited = 0; waited < timeout_s; waited += 0.002) {
    FILE* f = fopen(path.c_str(), "rb");
    if (f) {
      const bool ok = bytes == 0 || fread(data, 1, bytes, f) == bytes;
      fclose(f);
      unlink(path.c_str());
      return ok;
    }
    nanosleep(&nap, nullptr);
  }
  fprintf(stderr, "nccl_emu: timed out waiting for %s\n", path.c_str());
  return false;
}

std::string msg(const ncclComm* c, const char* kind, int src, int dst, unsigned long seq) {
  char b[96];
  snprintf(b, sizeof b, "/%s_%d_%d_%lu", kind, src, dst, seq);
  return c->dir + b;
}

ncclResult_t do_send(const PendingOp& o) {
  ncclComm* c = o.comm;
  return write_file(msg(c, "p2p", c->rank, o.peer, c->send_seq[o.peer]++), o.buf, o.bytes) ? ncclSuccess : ncclSystemError;
}
ncclResult_t do_recv(const PendingOp& o) {
  ncclComm* c = o.comm;
  return read_file(msg(c, "p2p", o.peer, c->rank, c->recv_seq[o.peer]++), o.buf, o.bytes) ? ncclSuccess : ncclSystemError;
}
}  // namespace

ncclResult_t ncclGetVersion(int* version) { if (version) *version = 0; return ncclSuccess; }   // 0: the emulation

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return ncclInvalidArgument;
  memset(id, 0, sizeof *id);
  unsigned long long r[2] = {0, 0};
  FILE* f = fopen("/dev/urandom", "rb");
  if (f) { (void)!fread(r, sizeof r, 1, f); fclose(f); }
  snprintf(id->internal, sizeof id->internal, "%016llx%016llx_%ld", r[0], r[1], (long)getpid());
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  id.internal[NCCL_UNIQUE_ID_BYTES - 1] = 0;
  const char* tmp = getenv("TMPDIR");
  ncclComm* c = new ncclComm;
  c->dir = std::string(tmp && *tmp ? tmp : "/tmp") + "/rgbl_nccl_emu_" + id.internal;
  c->world = nranks; c->rank = rank;
  c->send_seq.assign(nranks, 0); c->recv_seq.assign(nranks, 0);
  if (mkdir(c->dir.c_str(), 0700) != 0 && errno != EEXIST) { delete c; return ncclSystemError; }
  // rendezvous, as the real call: everybody announces itself and waits for the others
  char one = 1;
  for (int r = 0; r < nranks; ++r)
    if (r != rank && !write_file(msg(c, "init", rank, r, 0), &one, 1)) { delete c; return ncclSystemError; }
  for (int r = 0; r < nranks; ++r)
    if (r != rank && !read_file(msg(c, "init", r, rank, 0), &one, 1)) { delete c; return ncclSystemError; }
  *comm = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  rmdir(c->dir.c_str());   // succeeds for the last rank that leaves an empty mailbox
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclCommAbort(ncclComm_t c) { return ncclCommDestroy(c); }

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclSystemError: return "mailbox I/O failed or timed out (nccl_emu)";
    case ncclInvalidArgument: return "invalid argument (nccl_emu)";
    default: return "error (nccl_emu)";
  }
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t t, ncclComm_t c, hipStream_t) {
  if (!c || !sendbuff || !recvbuff) return ncclInvalidArgument;
  const size_t bytes = sendcount * elem_size(t);
  const unsigned long seq = c->gather_seq++;
  for (int r = 0; r < c->world; ++r)
    if (r != c->rank && !write_file(msg(c, "ag", c->rank, r, seq), sendbuff, bytes)) return ncclSystemError;
  memmove(static_cast<char*>(recvbuff) + (size_t)c->rank * bytes, sendbuff, bytes);
  for (int r = 0; r < c->world; ++r)
    if (r != c->rank && !read_file(msg(c, "ag", r, c->rank, seq), static_cast<char*>(recvbuff) + (size_t)r * bytes, bytes)) return ncclSystemError;
  return ncclSuccess;
}

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t) {
  if (!c || peer < 0 || peer >= c->world) return ncclInvalidArgument;
  PendingOp o{true, const_cast<void*>(sendbuff), count * elem_size(t), peer, c};
  if (g_group_depth > 0) { g_pending.push_back(o); return ncclSuccess; }
  return do_send(o);
}

ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t) {
  if (!c || peer < 0 || peer >= c->world) return ncclInvalidArgument;
  PendingOp o{false, recvbuff, count * elem_size(t), peer, c};
  if (g_group_depth > 0) { g_pending.push_back(o); return ncclSuccess; }
  return do_recv(o);
}

ncclResult_t ncclGroupStart() { ++g_group_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd() {
  if (g_group_depth <= 0) return ncclInvalidUsage;
  if (--g_group_depth > 0) return ncclSuccess;
  std::vector<PendingOp> ops;
  ops.swap(g_pending);
  ncclResult_t rc = ncclSuccess;
  for (const PendingOp& o : ops) if (o.send && rc == ncclSuccess) rc = do_send(o);
  for (const PendingOp& o : ops) if (!o.send && rc == ncclSuccess) rc = do_recv(o);
  return rc;
}
