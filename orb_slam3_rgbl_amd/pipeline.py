"""The batched front-end step (extract -> depth -> match) on device-resident inputs and the multi-GPU gather of its
results — the code `bench.py` times and `tests/test_distributed.py` drives (there on the CPU: the SIMT-emulated library,
CPU tensors and the gloo backend run the very same calls).

One process per GPU; frames / sequences are sharded over the ranks and nothing in the per-frame path depends on another
rank: the only exchange is the gather of the results to rank 0.  It is the two-phase variable-length
gather of 8(e):
  phase 1  the per-frame keypoint counts of every rank (all_gather of B int32 each),
  phase 2  the compacted records (68 B per keypoint, `rgbl_pack_records_device`) by point-to-point send / recv, the root
           posting one receive of the exact size per peer (on MI355X one xGMI link per peer, in parallel; a ring would be
           per-link bound and is the wrong shape).
`gather="step"` streams every step's records to the root while the next step computes.  Per step k, on the communication
stream and in this order: the pack of step k behind the step's last writers, the all-gather of its counts and their copy
into page-locked host memory (asynchronous, an event marks them).  The host touches those counts only one step later, after
step k + 1 has been enqueued, posts the exact-size transfers of step k and goes on: it never waits for the step it has just
queued, only for the one before, so the GPU always has a full step of work ahead of the host.  `gather="final"` keeps the
compacted records of every step in HBM and exchanges them once at the end (no host synchronisation inside the run);
`gather="none"` does no exchange.  With one rank the same choreography runs (pack on the communication stream, counts through
the page-locked block, the root's device copy, the events) without a peer - and through RCCL when a process group exists.
The communication stream is the low-priority stream of the Hamming scan (`rgbl_stream_create`, one lane): what is off the
chain resize -> FAST -> quad-tree -> descriptors that paces the steps only takes the slots the chain leaves, and a fifth stream
would not find a hardware queue of its own.  The exchange of step k - 1 therefore starts behind the scan of step k.

Two transports carry the same choreography:
  transport="abi"    (round 4, the default whenever a `comm` is given or there is one rank) - the library's own entry points
                     `rgbl_gather_pack` / `rgbl_gather_exchange` (csrc/gather.hip): ncclAllGather + grouped ncclSend / ncclRecv
                     straight on RCCL, queued on the stream handed to `rgbl_gather_create` - the low-priority stream above.  This
                     is what a C++ host calls; no framework process group and none of its internal streams.
  transport="torch"  torch.distributed (`all_gather`, `batch_isend_irecv`): the gloo backend of the CPU tests, and on GPUs
                     ProcessGroupNCCL, which runs its collectives on an internal stream of its own.
PyTorch is plumbing here: device memory, streams and torch.distributed.
"""
import ctypes as C
import os

import numpy as np

from . import _lib as L
from . import frontend as F

RECORD_BYTES = 68


def unpack_records(buf, counts):
    """Compacted records (bytes) + per-frame counts -> list of per-frame dicts of numpy arrays."""
    a = np.frombuffer(np.ascontiguousarray(buf), np.uint8)
    out, o = [], 0
    for n in counts:
        n = int(n)
        r = a[o * RECORD_BYTES:(o + n) * RECORD_BYTES].reshape(n, RECORD_BYTES)
        out.append(dict(n=n, kp=r[:, :28].copy(), desc=r[:, 28:60].copy(), depth=r[:, 60:64].copy().view(np.float32).reshape(n),
                        uright=r[:, 64:68].copy().view(np.float32).reshape(n)))
        o += n
    return out


class OutSet:
    """Outputs of one step (the pipeline ping-pongs between two sets) and the events that order their readers / writers."""

    def __init__(self, lib, torch, dev, B, cap):
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        self.kp = z((B, cap, 7), torch.float32)     # rgbl_keypoint records (28 B)
        self.desc = z((B, cap, 32), torch.uint8)
        self.n = z((B,), torch.int32)
        self.mono = z((B,), torch.int32)
        self.depth = z((B, cap), torch.float32)
        self.uright = z((B, cap), torch.float32)
        self.bi, self.bd, self.sd = (z((B, cap), torch.int32) for _ in range(3))
        self.ev = {}
        for name in ("extracted", "depth_done", "match_done", "comm_done"):
            e = C.c_void_p()
            L.check(lib, lib.rgbl_event_create(C.byref(e)))
            self.ev[name] = e


# A stream that PyTorch has wrapped (torch.cuda.ExternalStream, transport="torch") must outlive every tensor the caching
# allocator has seen on it, i.e. in practice the process: ONE such low-priority stream per device is created and shared by
# all pipelines of the process instead of leaking one per pipeline.
_WRAPPED_LOW_STREAMS = {}


def make_comm(lib, dist, device_index):
    """An RCCL communicator of the library (rgbl_comm_create) for the ranks of an initialised torch.distributed group:
    rank 0 draws the unique id, the group's object broadcast carries it - the only thing the framework is used for."""
    rank, world = dist.get_rank(), dist.get_world_size()
    ident = (C.c_uint8 * L.COMM_ID_BYTES)()
    if rank == 0:
        L.check(lib, lib.rgbl_comm_unique_id(ident))
    box = [bytes(ident)]
    dist.broadcast_object_list(box, src=0)
    ident = (C.c_uint8 * L.COMM_ID_BYTES).from_buffer_copy(box[0])
    comm = C.c_void_p()
    L.check(lib, lib.rgbl_comm_create(ident, world, rank, device_index, C.byref(comm)))
    return comm


class _Lane:
    """One set of handles (extractor, depth module, matcher) with their streams: the work of one step."""

    def __init__(self, lib, index, w, h, nfeatures, proj, n_points, batch, levels, scale, ini_th, min_th, serial, gather="none",
                 shared_low=False):
        self.own_stream = None
        self.owns_stream = True
        self.ex = F.ORBextractor(nfeatures, scale, levels, ini_th, min_th, w, h, max_batch=batch, device=index, lib=lib)
        self.dm = F.DepthModule(proj, w, h, max_points=n_points, max_keypoints=self.ex.max_keypoints, max_batch=batch, device=index, lib=lib)
        self.mt = F.ORBmatcher(0.6, False, device=index, lib=lib)
        one = C.c_void_p(lib.rgbl_extractor_stream(self.ex.h))
        if serial:
            L.check(lib, lib.rgbl_depth_set_stream(self.dm.h, one))
            L.check(lib, lib.rgbl_matcher_set_stream(self.mt.h, one))
        else:
            # The Hamming scan of step k is not on the chain resize -> FAST -> quad-tree -> descriptors that paces the steps: on
            # a LOW-PRIORITY stream of its own it runs next to the extraction of step k + 1 and only takes what that leaves
            # (142 - 144 k frames/s).  On the extractor's stream (RGBL_MATCHER_STREAM=shared, the rounds 1 - 3 default) it is a
            # link of the chain (136 - 138 k); on a stream of the default priority it competes with FAST for issue slots (130 - 132 k).
            # With the gather on, a communication stream of its own would be the FIFTH stream on the runtime's four hardware
            # queues, which costs more than the priority gains (123 k against 134 k at one rank with gather='step'): the
            # pipeline then uses this low-priority stream for the communication as well.
            mode = os.environ.get("RGBL_MATCHER_STREAM", "low")
            if mode == "shared":
                L.check(lib, lib.rgbl_matcher_set_stream(self.mt.h, one))
            elif mode == "low":
                if shared_low and index in _WRAPPED_LOW_STREAMS:
                    self.own_stream = _WRAPPED_LOW_STREAMS[index]
                else:
                    self.own_stream = C.c_void_p()
                    L.check(lib, lib.rgbl_stream_create_on(index, C.byref(self.own_stream), -1))
                    if shared_low:
                        _WRAPPED_LOW_STREAMS[index] = self.own_stream
                self.owns_stream = not shared_low
                L.check(lib, lib.rgbl_matcher_set_stream(self.mt.h, self.own_stream))
        self.streams(lib)

    def streams(self, lib):
        self.s_ex = C.c_void_p(lib.rgbl_extractor_stream(self.ex.h))
        self.s_dm = C.c_void_p(lib.rgbl_depth_stream(self.dm.h))
        self.s_mt = C.c_void_p(lib.rgbl_matcher_stream(self.mt.h))


class FrontEndPipeline:
    def __init__(self, lib, torch, dev, w, h, nfeatures, proj, n_points, batch, levels=8, scale=1.2, ini_th=12, min_th=7,
                 world=1, rank=0, gather=None, serial=False, keep_steps=0, log_steps=1,
                 sparse_depth=False, transport=None, comm=None, loopback=False, halo=0):
        self.lib, self.torch, self.dev = lib, torch, dev
        self.w, self.h, self.B, self.n_points = w, h, batch, n_points
        # halo = 1: ONE long sequence cut into contiguous chunks (sharding.frame_chunk): the rank owns `batch` frames per step and
        # also extracts the frame behind them - the first one of the next rank's chunk - so that its last frame can be matched
        # against its successor without any communication.  The halo frame is computed twice, gathered never (the next rank owns it).
        self.halo = int(halo)
        self.Bh = batch + self.halo
        # one rank and nobody asked for a gather: none (no buffers, no pack, no host event per step - ADVICE r3)
        gather = gather or ("step" if world > 1 else "none")
        self.world, self.rank, self.gather = world, rank, gather
        if transport is None:
            transport = "abi" if (comm is not None or world == 1) else "torch"
        if transport == "abi" and world > 1 and comm is None and gather != "none":
            # (no gather, no communicator needed: `--gather none` is the compute-only weak-scaling baseline - ADVICE r4)
            raise ValueError("transport='abi' with more than one rank needs an rgbl_comm (pipeline.make_comm)")
        self.transport, self.comm, self.g = transport, comm, None
        index = dev.index if (dev.type == "cuda" and dev.index is not None) else 0
        self.index = index
        # ONE set of handles (the "two steps in flight on two sets of handles" schedule of rounds 3 - 5 lost every measurement,
        # docs/history/, and is gone; the list stays a list for the callers that walk it)
        wrap_low = gather != "none" and transport == "torch" and dev.type == "cuda"
        self.lanes = [_Lane(lib, index, w, h, nfeatures, proj, n_points, self.Bh, levels, scale, ini_th, min_th, serial, gather,
                            shared_low=wrap_low)]
        self.ex, self.dm, self.mt = self.lanes[0].ex, self.lanes[0].dm, self.lanes[0].mt
        # sparse_depth: no dense ProcessedDepthMap (rgbl_depth_set_sparse) - the step never hands it out anyway
        for ln in self.lanes:
            ln.dm.SetSparseUpsampling(sparse_depth)
        self.cap = self.ex.max_keypoints
        if serial:
            # one stream for all handles and per-kernel HIP-event brackets on: every launch of the run is serialised - the
            # mode `rocprofv3 --kernel-trace --stats` is recorded in (profiles/)
            self.profile(True)
        self.sets = [OutSet(lib, torch, dev, self.Bh, self.cap) for _ in range(2)]
        self.pair_a = torch.arange(batch, dtype=torch.int32, device=dev)
        self.pair_b = (self.pair_a + 1) % self.Bh    # with a halo frame the last owned frame meets its true successor
        self.step_no = 0
        # ---- gather state
        self.cuda = dev.type == "cuda"
        self.comm_stream = None
        self.pending = None         # the step whose records are packed but not exchanged yet
        self.received = []          # root: per exchanged step, per rank (counts [B] int32 on the host, records uint8 tensor)
        self.keep = keep_steps      # root keeps the records of at most this many steps (0 = only the last)
        self.n_exchanged = 0
        self.dist = None
        self.n_slots = 2 if self.gather == "step" else max(log_steps, 1)   # 'final': one slot per step of the run
        low = self.lanes[0].own_stream if not serial else None
        if self.gather != "none" and transport == "abi":
            # the library's gather on the Hamming scan's low-priority stream
            self.g = C.c_void_p()
            L.check(lib, lib.rgbl_gather_create(comm, index, batch, self.cap, self.n_slots, low, C.byref(self.g)))
            if loopback:
                L.check(lib, lib.rgbl_gather_set_loopback(self.g, 1))
            self.s_comm = C.c_void_p(lib.rgbl_gather_stream(self.g))
        elif self.gather != "none" and self.cuda:
            # pack, counts and the exchange of step k - 1 behind the Hamming scan of step k on ONE low-priority stream (see _Lane)
            self.comm_stream = torch.cuda.ExternalStream(low.value, device=dev) if low is not None else torch.cuda.Stream(dev)
        if self.g is None:
            self.s_comm = C.c_void_p(self.comm_stream.cuda_stream) if self.comm_stream is not None else C.c_void_p(None)
        if self.gather != "none" and transport == "torch":
            import torch.distributed as dist
            if world > 1 or dist.is_initialized():
                self.dist = dist    # one rank with a process group: the collectives still go through the backend
            slots = self.n_slots
            rec_cap = batch * self.cap
            self.send = [torch.zeros(rec_cap * RECORD_BYTES, dtype=torch.uint8, device=dev) for _ in range(slots)]
            self.offsets = [torch.zeros(batch + 1, dtype=torch.int64, device=dev) for _ in range(slots)]
            self.counts = [torch.zeros(batch, dtype=torch.int32, device=dev) for _ in range(slots)]
            # counts of every rank: gathered on the device, read by the host from a page-locked copy behind an event
            self.all_counts = [[torch.zeros(batch, dtype=torch.int32, device=dev) for _ in range(world)] for _ in range(slots)]
            self.host_counts = [torch.zeros((world, batch), dtype=torch.int32, pin_memory=self.cuda) for _ in range(slots)]
            self.counts_ready = [torch.cuda.Event() if self.cuda else None for _ in range(slots)]
            self.overflow = torch.zeros(1, dtype=torch.int32, device=dev)
            self.recv = None
            if rank == 0:
                self.recv = [[torch.zeros(rec_cap * RECORD_BYTES, dtype=torch.uint8, device=dev) for _ in range(world)] for _ in range(2)]

    def profile(self, on):
        for ln in self.lanes:
            ln.ex.profile(on); ln.dm.profile(on); ln.mt.profile(on)

    def serialise(self):
        """One lane, all its handles on the extractor's stream (the per-kernel timing leg of bench.py)."""
        self.sync()
        ln = self.lanes[0]
        self.all_lanes = list(self.lanes)
        self.lanes = [ln]
        one = C.c_void_p(self.lib.rgbl_extractor_stream(ln.ex.h))
        L.check(self.lib, self.lib.rgbl_depth_set_stream(ln.dm.h, one))
        L.check(self.lib, self.lib.rgbl_matcher_set_stream(ln.mt.h, one))
        ln.streams(self.lib)

    def profile_read(self):
        k = {}
        for ln in self.lanes:
            for src in (ln.ex.profile_read(), ln.dm.profile_read(), ln.mt.profile_read()):
                for name, (ms, n) in src.items():
                    a = k.get(name, (0.0, 0))
                    k[name] = (a[0] + ms, a[1] + n)
        return k

    def profile_samples(self):
        """{kernel: [every launch's duration, ms]} of the (single) lane since profile(True)."""
        k = {}
        ln = self.lanes[0]
        for src in (ln.ex.profile_samples(), ln.dm.profile_samples(), ln.mt.profile_samples()):
            k.update(src)
        return k

    def set_inputs(self, d_imgs, d_cloud):
        self.d_imgs, self.d_cloud = d_imgs, d_cloud

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr())

    def step(self, active=True):
        """One step.  active=False: this rank has no frames for the step (BASELINE configs[3]: 11 sequences on 8 ranks - the
        ranks 3 .. 7 idle in the second round) but still takes part in the step's collectives with all-zero counts."""
        lib, p, B, w, h, cap = self.lib, self._p, self.Bh, self.w, self.h, self.cap   # extraction and depth include the halo frame
        o = self.sets[self.step_no % 2]
        ln = self.lanes[self.step_no % len(self.lanes)]
        # this set's readers of two steps ago must be done before the extractor overwrites it
        L.check(lib, lib.rgbl_event_wait(ln.s_ex, o.ev["depth_done"]))
        L.check(lib, lib.rgbl_event_wait(ln.s_ex, o.ev["match_done"]))
        if self.gather != "none":
            L.check(lib, lib.rgbl_event_wait(ln.s_ex, o.ev["comm_done"]))
        if not active:
            ts = C.c_void_p(self.torch.cuda.current_stream(self.dev).cuda_stream) if self.cuda else None
            if ts is not None:
                L.check(lib, lib.rgbl_stream_wait(ts, ln.s_ex))
            o.n.zero_()
            if ts is not None:
                L.check(lib, lib.rgbl_stream_wait(ln.s_ex, ts))
            for name in ("extracted", "depth_done", "match_done"):
                L.check(lib, lib.rgbl_event_record(o.ev[name], ln.s_ex))
            if self.gather != "none":
                prev = self.pending
                if self.gather == "step" and prev is not None:
                    self._exchange(prev)
                self._pack(o)
            self.step_no += 1
            return
        L.check(lib, lib.rgbl_extract_batch_device(ln.ex.h, p(self.d_imgs), B, w, h, w, w * h, 0, 0, p(o.kp), p(o.desc), cap, p(o.n), p(o.mono)))
        L.check(lib, lib.rgbl_event_record(o.ev["extracted"], ln.s_ex))
        # LiDAR projection + up-sampling: independent of the keypoints, runs concurrently on the depth stream
        n_points = self.n_points
        L.check(lib, lib.rgbl_depth_project_batch_device(ln.dm.h, p(self.d_cloud), B, n_points, n_points, 4 * n_points, w, h, None))
        L.check(lib, lib.rgbl_event_wait(ln.s_dm, o.ev["extracted"]))
        L.check(lib, lib.rgbl_depth_gather_batch_device(ln.dm.h, B, w, h, p(o.kp), p(o.n), cap, None, p(o.depth), p(o.uright)))
        L.check(lib, lib.rgbl_event_record(o.ev["depth_done"], ln.s_dm))
        L.check(lib, lib.rgbl_event_wait(ln.s_mt, o.ev["extracted"]))
        L.check(lib, lib.rgbl_hamming_bf_batch_device(ln.mt.h, p(o.desc), p(o.n), cap, p(self.pair_a), p(self.pair_b), self.B, p(o.bi),
                                                      p(o.bd), p(o.sd)))
        L.check(lib, lib.rgbl_event_record(o.ev["match_done"], ln.s_mt))
        if self.gather != "none":
            prev = self.pending
            if self.gather == "step" and prev is not None:
                # step k - 1's records travel while step k (queued above) and step k + 1 compute.  Posted BEFORE the pack of step k:
                # the communication stream is in order, and behind that pack the exchange would wait for the step just queued.
                self._exchange(prev)
            self._pack(o)
        self.step_no += 1

    # ---- gather ------------------------------------------------------------------------------------------------
    def _comm(self):
        return self.torch.cuda.stream(self.comm_stream) if self.comm_stream is not None else _Null()

    def _pack(self, o):
        """Compaction of the step's results on the communication stream, behind the step's last writers; phase 1 of the
        gather (the counts of every rank) follows it on the same stream and ends in page-locked host memory."""
        lib, p = self.lib, self._p
        slot = self.step_no % self.n_slots
        if self.g is not None:
            # pack + phase 1 (ncclAllGather of the counts, their copy into page-locked memory) in one call of the C ABI
            waits = (C.c_void_p * 2)(o.ev["depth_done"], o.ev["match_done"])
            L.check(lib, lib.rgbl_gather_pack(self.g, slot, p(o.n), p(o.kp), p(o.desc), p(o.depth), p(o.uright), waits, 2, o.ev["comm_done"]))
            self.pending = slot
            return
        L.check(lib, lib.rgbl_event_wait(self.s_comm, o.ev["depth_done"]))
        L.check(lib, lib.rgbl_event_wait(self.s_comm, o.ev["match_done"]))
        L.check(lib, lib.rgbl_pack_records_device(self.s_comm, p(o.n), p(o.kp), p(o.desc), p(o.depth), p(o.uright), self.B, self.cap, 0,
                                                  self.B * self.cap, p(self.send[slot]), p(self.offsets[slot]), p(self.overflow)))
        with self._comm():
            self.counts[slot].copy_(o.n[:self.B])  # the set is free again once the records and the counts are copied out
        L.check(lib, lib.rgbl_event_record(o.ev["comm_done"], self.s_comm))
        if self.gather == "step":
            self._gather_counts([slot])
        self.pending = slot

    def _gather_counts(self, slots):
        """Phase 1: per-frame counts of every rank for the given slots -> host_counts[slot] (asynchronous on a GPU)."""
        with self._comm():
            for slot in slots:
                if self.dist is not None:
                    self.dist.all_gather(self.all_counts[slot], self.counts[slot])
                else:
                    self.all_counts[slot][0].copy_(self.counts[slot])
                for r in range(self.world):
                    self.host_counts[slot][r].copy_(self.all_counts[slot][r], non_blocking=True)
                if self.cuda:
                    self.counts_ready[slot].record(self.comm_stream)

    def _exchange(self, slot):
        """Phase 2 of the gather of one packed step to rank 0: exact-size point-to-point transfers, one per
        peer, sized by the counts phase 1 left in page-locked memory."""
        torch, world, rank = self.torch, self.world, self.rank
        if self.g is not None:
            return self._exchange_abi(slot)
        if self.cuda:
            self.counts_ready[slot].synchronize()   # an event of an EARLIER step in gather='step': already signalled or about to be
        host_counts = [self.host_counts[slot][r].numpy().copy() for r in range(world)]
        totals = [int(np.minimum(np.maximum(c, 0), self.cap).sum()) for c in host_counts]
        with self._comm():
            ops = []
            if rank == 0:
                self.n_exchanged += 1
                bank = self.recv[self.n_exchanged % 2]
                for r in range(1, world):
                    if totals[r] > 0:
                        ops.append(self.dist.P2POp(self.dist.irecv, bank[r][:totals[r] * RECORD_BYTES], r))
                bank[0][:totals[0] * RECORD_BYTES].copy_(self.send[slot][:totals[0] * RECORD_BYTES])
            elif totals[rank] > 0:
                ops.append(self.dist.P2POp(self.dist.isend, self.send[slot][:totals[rank] * RECORD_BYTES], 0))
            if ops:
                for req in self.dist.batch_isend_irecv(ops):
                    req.wait()   # nccl: orders the communication stream behind the transfer, the host goes on
            if rank == 0:
                got = [(host_counts[r], bank[r][:totals[r] * RECORD_BYTES]) for r in range(world)]
                if self.keep:
                    got = [(c, t.clone()) for c, t in got]
                    self.received = (self.received + [got])[-self.keep:]
                else:
                    self.received = [got]
        if self.pending == slot:
            self.pending = None

    def _exchange_abi(self, slot):
        """Phase 2 through the C ABI: rgbl_gather_exchange waits for the slot's counts and posts the grouped ncclSend / ncclRecv."""
        lib, torch = self.lib, self.torch
        L.check(lib, lib.rgbl_gather_exchange(self.g, slot))
        if self.rank == 0:
            self.n_exchanged += 1
            got = []
            for r in range(self.world):
                h_counts, d_rec, n_rec = C.c_void_p(), C.c_void_p(), C.c_longlong()
                L.check(lib, lib.rgbl_gather_result(self.g, r, C.byref(h_counts), C.byref(d_rec), C.byref(n_rec)))
                counts = np.ctypeslib.as_array(C.cast(h_counts, C.POINTER(C.c_int32)), (self.B,)).copy()
                rec = None
                if self.keep:   # tests: a copy of the root's bank (the bank itself is reused two exchanges later)
                    # no fill kernel on torch's stream that could land behind the copy on the gather's stream, and both
                    # directions ordered by events: the block the caching allocator hands out may still be in use on torch's
                    # current stream, and the tensor's later readers run there
                    rec = torch.empty(n_rec.value * RECORD_BYTES, dtype=torch.uint8, device=self.dev)
                    ts = C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream) if self.cuda else None
                    if ts is not None:
                        L.check(lib, lib.rgbl_stream_wait(self.s_comm, ts))
                    L.check(lib, lib.rgbl_gather_copy_result(self.g, r, C.c_void_p(rec.data_ptr()), rec.numel()))
                    if ts is not None:
                        L.check(lib, lib.rgbl_stream_wait(ts, self.s_comm))
                got.append((counts, rec))
            self.received = (self.received + [got])[-self.keep:] if self.keep else [got]
        if self.pending == slot:
            self.pending = None

    def finish(self):
        """Flushes the gather: the last step's records (gather='step') or every kept step's (gather='final')."""
        if self.gather == "step" and self.pending is not None:
            self._exchange(self.pending)
        elif self.gather == "final":
            first = max(self.step_no - self.n_slots, 0)
            slots = [k % self.n_slots for k in range(first, self.step_no)]
            if self.g is None:
                self._gather_counts(slots)   # (the ABI's pack has already queued every step's all-gather)
            for slot in slots:
                self._exchange(slot)
        if self.g is not None:
            L.check(self.lib, self.lib.rgbl_gather_sync(self.g))
        elif self.comm_stream is not None:
            self.comm_stream.synchronize()

    def sync(self):
        if self.dev.type == "cuda":
            self.torch.cuda.synchronize(self.dev)
        for ln in self.lanes:
            L.check(self.lib, self.lib.rgbl_extractor_sync(ln.ex.h))  # also surfaces device-side overflow flags
        if self.g is not None:
            L.check(self.lib, self.lib.rgbl_gather_sync(self.g))   # RGBL_ERR_OVERFLOW if a pack did not fit
        elif self.gather != "none" and int(self.overflow.cpu()[0]) != 0:
            raise RuntimeError("record buffer overflow in rgbl_pack_records_device")

    def last(self):
        return self.sets[(self.step_no - 1) % 2]

    def close(self):
        if self.g is not None:
            self.lib.rgbl_gather_destroy(self.g)
            self.g = None
        for ln in getattr(self, "all_lanes", self.lanes):
            ln.ex.close(); ln.dm.close(); ln.mt.close()
            if ln.own_stream is not None:
                # a stream PyTorch has wrapped is the process-wide one of _WRAPPED_LOW_STREAMS and stays; the others go
                if ln.owns_stream:
                    self.lib.rgbl_stream_destroy(ln.own_stream)
                ln.own_stream = None


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
