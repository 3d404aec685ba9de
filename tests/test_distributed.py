"""The N > 1 path on CPU: two processes (gloo) run the SAME pipeline code bench.py times (orb_slam3_rgbl_amd/pipeline.py:
step = extract -> depth -> match, record packing, two-phase variable-length gather) with the SIMT-emulated library as the
per-rank front end; what rank 0 gathers must be byte-identical to what a single process computes for every rank's input,
and the records must decode to the oracle's keypoints / descriptors / depths (BASELINE configs[3])."""
import os
import subprocess
import sys

import numpy as np
import pytest

from orb_slam3_rgbl_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["RGBL_ROOT"])
import numpy as np, torch, torch.distributed as dist
from orb_slam3_rgbl_amd import _lib, synth, sharding
from orb_slam3_rgbl_amd import frontend as F
from orb_slam3_rgbl_amd.pipeline import FrontEndPipeline, unpack_records, make_comm

W, H, NF, LEVELS, B, STEPS, N_AZ = 200, 160, 300, 4, 3, 3, 240
MODE = os.environ["RGBL_GATHER"]
TRANSPORT = os.environ.get("RGBL_TRANSPORT", "torch")   # abi: the library's rgbl_gather_* over RCCL (nccl_emu on the CPU)
CUDA = os.environ.get("RGBL_DEVICE", "cpu") == "cuda"     # the same worker on real GPUs: backend nccl (= RCCL), product library
if CUDA:
    W, H, NF, LEVELS, B, STEPS, N_AZ = 640, 360, 800, 8, 16, 4, 600
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
DEV = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))) if CUDA else torch.device("cpu")

def inputs_of(rank):
    sq = synth.Sequence(100 + rank, W, H, n_frames=B)
    frames = np.stack([sq.frame(i) for i in range(B)])
    cloud = np.stack([synth.lidar_scan(100 * rank + i, n_az=N_AZ) for i in range(B)])
    return frames, cloud

def make(lib, rank, world, gather, keep=0, transport=None, comm=None):
    K = synth.KITTI_K.copy(); K[0, 2], K[1, 2] = W / 2.0, H / 2.0
    proj = F.projection_matrix(K, synth.KITTI_TR, lib)
    frames, cloud = inputs_of(rank)
    pipe = FrontEndPipeline(lib, torch, DEV, W, H, NF, proj, cloud.shape[2], B, levels=LEVELS, ini_th=20, min_th=7,
                            world=world, rank=rank, gather=gather, keep_steps=keep, log_steps=STEPS, transport=transport, comm=comm,
                            loopback=(transport == "abi" and world == 1 and comm is not None))
    pipe.set_inputs(torch.from_numpy(frames).to(DEV), torch.from_numpy(cloud).to(DEV))
    return pipe, frames, cloud, proj

def main():
    if CUDA:
        dist.init_process_group("nccl", device_id=DEV)
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    lib = _lib.load() if CUDA else _lib.bind(os.environ["RGBL_EMU_LIB"])
    if CUDA and os.environ.get("RGBL_SELF_P2P") == "1":
        # one rank: a grouped send / receive to itself is the only way its point-to-point path can run through RCCL
        a = torch.arange(1 << 20, dtype=torch.int32, device=DEV)
        b = torch.zeros_like(a)
        for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, a, rank), dist.P2POp(dist.irecv, b, rank)]):
            req.wait()
        torch.cuda.synchronize(DEV)
        assert torch.equal(a, b)
        print("SELF_P2P_OK")
    comm = make_comm(lib, dist, DEV.index if CUDA else 0) if TRANSPORT == "abi" else None
    pipe, _, _, _ = make(lib, rank, world, MODE, keep=STEPS, transport=TRANSPORT, comm=comm)
    for _ in range(STEPS):
        pipe.step()
    pipe.finish()
    pipe.sync()
    ok = True
    if rank == 0:
        from oracle import oracle_py as O   # the checker
        ok &= len(pipe.received) == STEPS
        for r in range(world):
            # what a single process produces for rank r's input
            solo, frames, cloud, proj = make(lib, r, 1, "none")
            solo.step(); solo.sync()
            o = solo.last()
            n = o.n.cpu().numpy()
            s_kp, s_desc, s_depth, s_uright = (t.cpu().numpy() for t in (o.kp, o.desc, o.depth, o.uright))
            for got in pipe.received:                      # every step processed the same resident batch
                counts, rec = got[r]
                ok &= np.array_equal(counts, n)
                fr = unpack_records(rec.cpu().numpy(), counts)
                for f in range(B):
                    m = int(n[f])
                    ok &= np.array_equal(fr[f]["kp"], s_kp[f, :m].view(np.uint8).reshape(m, 28))
                    ok &= np.array_equal(fr[f]["desc"], s_desc[f, :m])
                    ok &= np.array_equal(fr[f]["depth"].view(np.uint32), s_depth[f, :m].view(np.uint32))
                    ok &= np.array_equal(fr[f]["uright"].view(np.uint32), s_uright[f, :m].view(np.uint32))
            # and the records decode to the oracle's results
            orc = O.Extractor(NF, 1.2, LEVELS, 20, 7)
            P = O.make_depth_params(proj)
            fr = unpack_records(pipe.received[-1][r][1].cpu().numpy(), pipe.received[-1][r][0])
            for f in range(B):
                okps, odesc, _ = orc(frames[f])
                ok &= fr[f]["n"] == len(okps) and np.array_equal(fr[f]["kp"], okps.view(np.uint8).reshape(len(okps), 28))
                ok &= np.array_equal(fr[f]["desc"], odesc)
                od, our, _, _ = O.depth(P, cloud[f], W, H, np.stack([okps["x"], okps["y"]], 1), okps["x"], want_maps=False)
                ok &= np.array_equal(fr[f]["depth"].view(np.uint32), od.view(np.uint32))
                ok &= np.array_equal(fr[f]["uright"].view(np.uint32), our.view(np.uint32))
            solo.close()
        print("GATHER_OK" if ok else "GATHER_MISMATCH")
    pipe.close()
    if comm is not None:
        lib.rgbl_comm_destroy(comm)
    dist.barrier()
    dist.destroy_process_group()

main()
'''


CHUNK_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["RGBL_ROOT"])
import numpy as np, torch, torch.distributed as dist
from orb_slam3_rgbl_amd import _lib, synth, sharding
from orb_slam3_rgbl_amd import frontend as F
from orb_slam3_rgbl_amd.pipeline import FrontEndPipeline, unpack_records, make_comm

W, H, NF, LEVELS, N_AZ, N_FRAMES = 200, 160, 300, 4, 240, 7      # 7 frames over 2 ranks: chunks of 4 and 3
DEV = torch.device("cpu")

def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    lib = _lib.bind(os.environ["RGBL_EMU_LIB"])
    K = synth.KITTI_K.copy(); K[0, 2], K[1, 2] = W / 2.0, H / 2.0
    proj = F.projection_matrix(K, synth.KITTI_TR, lib)
    sq = synth.Sequence(300, W, H, n_frames=N_FRAMES)            # ONE long sequence, the same on every rank
    frames = np.stack([sq.frame(i) for i in range(N_FRAMES)])
    cloud = np.stack([synth.lidar_scan(300 + i, n_az=N_AZ) for i in range(N_FRAMES)])
    b, e, e_halo = sharding.frame_chunk(N_FRAMES, world, rank)
    own = e - b
    # the gather's step shape is the same on every rank: the largest chunk (ranks with a shorter one pad with an empty frame count)
    comm = make_comm(lib, dist, 0)
    pipe = FrontEndPipeline(lib, torch, DEV, W, H, NF, proj, cloud.shape[2], own, levels=LEVELS, ini_th=20, min_th=7, world=1, rank=0,
                            gather="none", halo=e_halo - e)
    pipe.set_inputs(torch.from_numpy(frames[b:e_halo].copy()), torch.from_numpy(cloud[b:e_halo].copy()))
    pipe.step(); pipe.sync()
    o = pipe.last()
    # what ONE process computes for the whole sequence
    solo = FrontEndPipeline(lib, torch, DEV, W, H, NF, proj, cloud.shape[2], N_FRAMES, levels=LEVELS, ini_th=20, min_th=7, world=1, rank=0, gather="none")
    solo.set_inputs(torch.from_numpy(frames), torch.from_numpy(cloud))
    solo.step(); solo.sync()
    s = solo.last()
    ok = True
    for t in range(b, e):
        n = int(s.n[t])
        ok &= int(o.n[t - b]) == n
        ok &= torch.equal(o.kp[t - b, :n].view(torch.int32), s.kp[t, :n].view(torch.int32)) and torch.equal(o.desc[t - b, :n], s.desc[t, :n])   # (class_id -1 reads as NaN in a float view)
        ok &= torch.equal(o.depth[t - b, :n].view(torch.int32), s.depth[t, :n].view(torch.int32))
        if t + 1 < N_FRAMES:   # every owned frame meets its TRUE successor - across the chunk boundary through the halo frame
            ok &= torch.equal(o.bi[t - b, :n], s.bi[t, :n]) and torch.equal(o.bd[t - b, :n], s.bd[t, :n]) and torch.equal(o.sd[t - b, :n], s.sd[t, :n])
    if e < N_FRAMES:
        ok &= e_halo == e + 1 and int(o.n[own]) == int(s.n[e])      # the halo frame is the next rank's first frame, recomputed here
    flags = [None] * world
    dist.all_gather_object(flags, bool(ok))
    if rank == 0:
        print("CHUNKS_OK" if all(flags) else "CHUNKS_MISMATCH %s" % flags)
    pipe.close(); solo.close(); lib.rgbl_comm_destroy(comm)
    dist.barrier(); dist.destroy_process_group()

main()
'''


def test_chunked_sequence_matches_across_the_chunk_boundary(tmp_path, oracle, emu_lib):
    """SURVEY 8(e) 'single long sequence': contiguous chunks per rank with a one-frame halo (sharding.frame_chunk,
    FrontEndPipeline(halo=1)).  Two gloo ranks; every owned frame's keypoints, depths and (t, t + 1) matches - the pair across
    the chunk boundary included - equal what one process computes for the whole sequence."""
    script = tmp_path / "chunk_worker.py"
    script.write_text(CHUNK_WORKER)
    env = dict(os.environ, RGBL_ROOT=ROOT, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", RGBL_EMU_THREADS="2", TMPDIR=str(tmp_path),
               RGBL_EMU_LIB=os.path.join(ROOT, "tests", "_build", "librgbl_frontend_emu.so"))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29525", str(script)],
                         env=env, capture_output=True, text=True, timeout=600)
    assert "CHUNKS_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


ROUND_ROBIN_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["RGBL_ROOT"])
import numpy as np, torch, torch.distributed as dist
from orb_slam3_rgbl_amd import _lib, synth, sharding
from orb_slam3_rgbl_amd import frontend as F
from orb_slam3_rgbl_amd.pipeline import FrontEndPipeline, unpack_records, make_comm

# BASELINE configs[3]: the 11 KITTI sequences 00-10 on the 8 GPUs of a node, sequence s -> rank s mod 8 (sharding.sequences_of_rank):
# the ranks 0 - 2 own two sequences, the others one and idle in the second round (all-zero counts, same collectives)
W, H, NF, LEVELS, B, N_AZ, N_SEQ = 160, 128, 200, 3, 2, 200, 11
DEV = torch.device("cpu")

def inputs_of(seq):
    sq = synth.Sequence(500 + seq, W, H, n_frames=B)
    frames = np.stack([sq.frame(i) for i in range(B)])
    cloud = np.stack([synth.lidar_scan(500 + 10 * seq + i, n_az=N_AZ) for i in range(B)])
    return frames, cloud

def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    lib = _lib.bind(os.environ["RGBL_EMU_LIB"])
    K = synth.KITTI_K.copy(); K[0, 2], K[1, 2] = W / 2.0, H / 2.0
    proj = F.projection_matrix(K, synth.KITTI_TR, lib)
    mine = sharding.sequences_of_rank(N_SEQ, world, rank)
    rounds = max(len(sharding.sequences_of_rank(N_SEQ, world, r)) for r in range(world))
    transport = os.environ.get("RGBL_TRANSPORT", "abi")
    comm = make_comm(lib, dist, 0) if transport == "abi" else None
    if comm is not None:
        cw = __import__("ctypes").c_int(0)
        lib.rgbl_comm_info(comm, __import__("ctypes").byref(cw), None, None, None)
        assert cw.value == world
    n_points = inputs_of(0)[1].shape[2]
    pipe = FrontEndPipeline(lib, torch, DEV, W, H, NF, proj, n_points, B, levels=LEVELS, ini_th=20, min_th=7, world=world, rank=rank,
                            gather=os.environ["RGBL_GATHER"], keep_steps=rounds, log_steps=rounds, transport=transport, comm=comm)
    for j in range(rounds):
        if j < len(mine):
            frames, cloud = inputs_of(mine[j])
            pipe.set_inputs(torch.from_numpy(frames), torch.from_numpy(cloud))
            pipe.step()
        else:
            pipe.step(active=False)
        pipe.sync()   # (the inputs of the next round replace this round's)
    pipe.finish()
    pipe.sync()
    ok = True
    if rank == 0:
        ok &= len(pipe.received) == rounds
        solo = FrontEndPipeline(lib, torch, DEV, W, H, NF, proj, n_points, B, levels=LEVELS, ini_th=20, min_th=7, world=1, rank=0, gather="none")
        seen = 0
        for j in range(rounds):
            for r in range(world):
                counts, rec = pipe.received[j][r]
                s = r + world * j
                if s >= N_SEQ:
                    ok &= int(np.abs(counts).sum()) == 0 and rec.numel() == 0      # an idle rank sends nothing
                    continue
                frames, cloud = inputs_of(s)
                solo.set_inputs(torch.from_numpy(frames), torch.from_numpy(cloud))
                solo.step(); solo.sync()
                o = solo.last()
                n = o.n.numpy()
                ok &= np.array_equal(counts, n) and int(n.min()) > 10
                fr = unpack_records(rec.numpy(), counts)
                for f in range(B):
                    m = int(n[f])
                    ok &= np.array_equal(fr[f]["kp"], o.kp[f, :m].numpy().view(np.uint8).reshape(m, 28))
                    ok &= np.array_equal(fr[f]["desc"], o.desc[f, :m].numpy())
                    ok &= np.array_equal(fr[f]["depth"].view(np.uint32), o.depth[f, :m].numpy().view(np.uint32))
                    ok &= np.array_equal(fr[f]["uright"].view(np.uint32), o.uright[f, :m].numpy().view(np.uint32))
                seen += 1
        ok &= seen == N_SEQ
        solo.close()
        print("ROUND_ROBIN_OK %d sequences on %d ranks" % (seen, world) if ok else "ROUND_ROBIN_MISMATCH")
    pipe.close()
    if comm is not None:
        lib.rgbl_comm_destroy(comm)
    dist.barrier()
    dist.destroy_process_group()

main()
'''


@pytest.mark.parametrize("mode,port,transport", [("step", 29551, "abi"), ("final", 29553, "abi")])
def test_eight_ranks_eleven_sequences_round_robin(tmp_path, oracle, emu_lib, mode, port, transport):
    """BASELINE configs[3] at the node's real width: 8 ranks (gloo bootstrap; the library's gather over the emulated RCCL, or
    torch.distributed), 11 sequences round-robin - two rounds, five ranks idle in the second one with all-zero counts.  What rank 0
    holds for every (round, rank) is byte-identical with a single-process run of that sequence."""
    script = tmp_path / "rr_worker.py"
    script.write_text(ROUND_ROBIN_WORKER)
    env = dict(os.environ, RGBL_ROOT=ROOT, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", RGBL_GATHER=mode, RGBL_EMU_THREADS="1",
               RGBL_TRANSPORT=transport, TMPDIR=str(tmp_path),
               RGBL_EMU_LIB=os.path.join(ROOT, "tests", "_build", "librgbl_frontend_emu.so"))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         env=env, capture_output=True, text=True, timeout=900)
    assert "ROUND_ROBIN_OK 11 sequences on 8 ranks" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_chunk_partition_covers_all_frames_with_halo():
    for n, world in ((4541, 8), (10, 3), (7, 8), (64, 1)):
        seen = []
        for r in range(world):
            b, e, e_halo = sharding.frame_chunk(n, world, r)
            seen.extend(range(b, e))
            assert e_halo == min(e + 1, n)
        assert seen == list(range(n))
    assert sharding.sequences_of_rank(11, 8, 2) == [2, 10]


def test_pack_unpack_roundtrip(emu_lib):
    import ctypes as C

    import torch

    from orb_slam3_rgbl_amd import _lib as L
    from orb_slam3_rgbl_amd.pipeline import RECORD_BYTES, unpack_records
    rng = np.random.default_rng(0)
    B, cap = 4, 50
    n = torch.tensor([50, 0, 17, 3], dtype=torch.int32)
    kp = torch.from_numpy(rng.standard_normal((B, cap, 7)).astype(np.float32))
    desc = torch.from_numpy(rng.integers(0, 256, (B, cap, 32), dtype=np.uint8))
    dep = torch.from_numpy(rng.standard_normal((B, cap)).astype(np.float32))
    ur = torch.from_numpy(rng.standard_normal((B, cap)).astype(np.float32))
    out = torch.zeros(B * cap * RECORD_BYTES, dtype=torch.uint8)
    off = torch.zeros(B + 1, dtype=torch.int64)
    ovf = torch.zeros(1, dtype=torch.int32)
    p = lambda t: C.c_void_p(t.data_ptr())
    L.check(emu_lib, emu_lib.rgbl_pack_records_device(None, p(n), p(kp), p(desc), p(dep), p(ur), B, cap, 0, B * cap, p(out), p(off), p(ovf)))
    assert off.tolist() == [0, 50, 50, 67, 70] and int(ovf[0]) == 0
    fr = unpack_records(out.numpy(), n.numpy())
    for i in range(B):
        m = int(n[i])
        assert fr[i]["n"] == m
        assert np.array_equal(fr[i]["desc"], desc[i, :m].numpy())
        assert np.array_equal(fr[i]["kp"], kp[i, :m].numpy().view(np.uint8).reshape(m, 28))
        assert np.array_equal(fr[i]["depth"], dep[i, :m].numpy()) and np.array_equal(fr[i]["uright"], ur[i, :m].numpy())
    # a buffer that is too small: the overflow flag is raised, nothing is written past the end
    small = torch.zeros(60 * RECORD_BYTES, dtype=torch.uint8)
    L.check(emu_lib, emu_lib.rgbl_pack_records_device(None, p(n), p(kp), p(desc), p(dep), p(ur), B, cap, 0, 60, p(small), p(off), p(ovf)))
    assert int(ovf[0]) == 1


@pytest.mark.parametrize("mode,port,transport", [("step", 29517, "torch"), ("final", 29519, "torch"),
                                                 ("step", 29521, "abi"), ("final", 29523, "abi")])
def test_two_rank_gather_equals_single_process(tmp_path, oracle, emu_lib, mode, port, transport):
    """transport torch: torch.distributed over gloo.  transport abi: the library's own rgbl_gather_* entry points
    (csrc/gather.hip) with tests/emu/nccl_emu.cpp standing in for RCCL - gloo only carries the unique id."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, RGBL_ROOT=ROOT, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", RGBL_GATHER=mode, RGBL_EMU_THREADS="2",
               RGBL_TRANSPORT=transport, TMPDIR=str(tmp_path),
               RGBL_EMU_LIB=os.path.join(ROOT, "tests", "_build", "librgbl_frontend_emu.so"))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         env=env, capture_output=True, text=True, timeout=600)
    assert "GATHER_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def _run_on_gpus(tmp_path, nproc, mode, port, self_p2p=False, transport="torch"):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, RGBL_ROOT=ROOT, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", RGBL_GATHER=mode, RGBL_DEVICE="cuda",
               HSA_ENABLE_IPC_MODE_LEGACY="0", RGBL_SELF_P2P="1" if self_p2p else "0", RGBL_TRANSPORT=transport)
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % nproc,
                           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                          env=env, capture_output=True, text=True, timeout=240)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,port", [("step", 29531), ("final", 29533)])
def test_rccl_gather_with_one_rank(tmp_path, oracle, gpu_lib, mode, port):
    """ONE rank, backend nccl: process-group initialisation, the all-gather of the counts and (a grouped send / receive of the
    rank to itself) the point-to-point path run through RCCL on the hardware, around the same pipeline the N > 1 bench runs.
    What the root holds must equal a plain single-process run and decode to the oracle's results."""
    out = _run_on_gpus(tmp_path, 1, mode, port, self_p2p=(mode == "step"))
    assert "GATHER_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
    if mode == "step":
        assert "SELF_P2P_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("mode,port", [("step", 29541), ("final", 29543)])
def test_rccl_gather_through_the_c_abi_with_one_rank(tmp_path, oracle, gpu_lib, mode, port):
    """ONE rank, the library's own gather (rgbl_comm_create / rgbl_gather_pack / rgbl_gather_exchange): ncclCommInitRank, the
    ncclAllGather of the counts and - loopback - a grouped ncclSend / ncclRecv of the rank's records to itself run through RCCL
    on the hardware, on the pipeline's low-priority stream; torch.distributed only hands the unique id around."""
    out = _run_on_gpus(tmp_path, 1, mode, port, transport="abi")
    assert "GATHER_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("mode,port,transport", [("step", 29535, "torch"), ("final", 29537, "torch"), ("step", 29545, "abi"), ("final", 29547, "abi")])
def test_rccl_gather_two_ranks(tmp_path, oracle, gpu_lib, mode, port, transport):
    """Two ranks on two GPUs over RCCL / xGMI (skipped on the one-GPU boxes of the pool)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    out = _run_on_gpus(tmp_path, 2, mode, port, transport=transport)
    assert "GATHER_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_gather_entry_points_reject_misuse(emu_lib):
    """Error behaviour of the C-ABI gather (csrc/gather.hip): status codes, never a hang or a crash."""
    import ctypes as C

    import numpy as np

    from orb_slam3_rgbl_amd import _lib as L
    lib = emu_lib
    g = C.c_void_p()
    assert lib.rgbl_gather_create(None, 0, 0, 10, 2, None, C.byref(g)) == L.ERR_INVALID          # batch 0
    assert lib.rgbl_gather_create(None, 99, 4, 10, 2, None, C.byref(g)) == L.ERR_NO_DEVICE       # no such device
    ident = (C.c_uint8 * L.COMM_ID_BYTES)()
    comm = C.c_void_p()
    assert lib.rgbl_comm_create(ident, 2, 2, 0, C.byref(comm)) == L.ERR_INVALID                   # rank >= world
    assert lib.rgbl_comm_create(None, 1, 0, 0, C.byref(comm)) == L.ERR_INVALID
    L.check(lib, lib.rgbl_gather_create(None, 0, 3, 8, 2, None, C.byref(g)))                      # one rank, no communicator
    assert lib.rgbl_gather_set_loopback(g, 1) == L.ERR_INVALID                                    # loopback needs a communicator
    assert lib.rgbl_gather_exchange(g, 0) == L.ERR_INVALID                                        # nothing packed
    assert lib.rgbl_gather_exchange(g, 5) == L.ERR_INVALID
    assert lib.rgbl_gather_result(g, 0, None, None, None) == L.ERR_INVALID                        # no exchange yet
    n = np.array([8, 0, 3], np.int32)
    kp = np.zeros((3, 8, 7), np.float32); desc = np.arange(3 * 8 * 32, dtype=np.uint8).reshape(3, 8, 32)
    dep = np.ones((3, 8), np.float32); ur = np.full((3, 8), 2.0, np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    assert lib.rgbl_gather_pack(g, 2, p(n), p(kp), p(desc), p(dep), p(ur), None, 0, None) == L.ERR_INVALID   # slot out of range
    L.check(lib, lib.rgbl_gather_pack(g, 1, p(n), p(kp), p(desc), p(dep), p(ur), None, 0, None))
    # packing over a step that was never exchanged would lose its records silently
    assert lib.rgbl_gather_pack(g, 1, p(n), p(kp), p(desc), p(dep), p(ur), None, 0, None) == L.ERR_INVALID
    assert b"still holds a packed step" in lib.rgbl_last_error()
    L.check(lib, lib.rgbl_gather_exchange(g, 1))
    L.check(lib, lib.rgbl_gather_sync(g))
    assert lib.rgbl_gather_result(g, 1, None, None, None) == L.ERR_INVALID                        # rank out of range
    counts, rec, cnt = C.c_void_p(), C.c_void_p(), C.c_longlong()
    L.check(lib, lib.rgbl_gather_result(g, 0, C.byref(counts), C.byref(rec), C.byref(cnt)))
    assert cnt.value == 11 and list(np.ctypeslib.as_array(C.cast(counts, C.POINTER(C.c_int32)), (3,))) == [8, 0, 3]
    small = np.zeros(10 * 68, np.uint8)
    assert lib.rgbl_gather_copy_result(g, 0, p(small), small.size) == L.ERR_CAPACITY
    full = np.zeros(11 * 68, np.uint8)
    L.check(lib, lib.rgbl_gather_copy_result(g, 0, p(full), full.size))
    L.check(lib, lib.rgbl_gather_sync(g))
    assert np.array_equal(full.reshape(11, 68)[:, 28:60], np.concatenate([desc[0, :8], desc[2, :3]]))
    assert lib.rgbl_gather_exchange(g, 1) == L.ERR_INVALID                                        # already exchanged
    lib.rgbl_gather_destroy(g)
    # the communicator's lifetime - destroyed by the caller while a gather handle still uses it: deferred to the
    # last rgbl_gather_destroy, the handle keeps working in between (one emulated rank, loopback transfers through the "RCCL")
    os.environ.setdefault("TMPDIR", "/tmp")
    L.check(lib, lib.rgbl_comm_unique_id(ident))
    L.check(lib, lib.rgbl_comm_create(ident, 1, 0, 0, C.byref(comm)))
    L.check(lib, lib.rgbl_gather_create(comm, 0, 3, 8, 2, None, C.byref(g)))
    L.check(lib, lib.rgbl_gather_set_loopback(g, 1))
    lib.rgbl_comm_destroy(comm)                       # too early: the gather still holds it
    lib.rgbl_comm_destroy(comm)                       # and twice
    L.check(lib, lib.rgbl_gather_pack(g, 0, p(n), p(kp), p(desc), p(dep), p(ur), None, 0, None))
    L.check(lib, lib.rgbl_gather_exchange(g, 0))
    L.check(lib, lib.rgbl_gather_sync(g))
    L.check(lib, lib.rgbl_gather_result(g, 0, C.byref(counts), C.byref(rec), C.byref(cnt)))
    assert cnt.value == 11
    g2 = C.c_void_p()
    assert lib.rgbl_gather_create(comm, 0, 3, 8, 2, None, C.byref(g2)) == L.ERR_COMM              # no new users of a released communicator
    lib.rgbl_gather_destroy(g)                        # frees the communicator as well
