"""The N > 1 path on CPU: two processes (gloo), frames sharded by rank, records gathered to rank 0, and the
gathered result must be byte-identical to what a single process produces for all frames (BASELINE configs[3])."""
import os
import subprocess
import sys

import numpy as np

from orb_slam3_rgbl_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["RGBL_ROOT"])
import numpy as np, torch, torch.distributed as dist
from orb_slam3_rgbl_amd import sharding, synth
from oracle import oracle_py as O   # stands in for the per-rank GPU front end: same record layout

def records_for(seq_ids, frames_per_seq, w, h, cap):
    ex = O.Extractor(300, 1.2, 4, 20, 7)
    n, kp, desc, dep, ur = [], [], [], [], []
    for s in seq_ids:
        sq = synth.Sequence(s, w, h, n_frames=frames_per_seq)
        for t in range(frames_per_seq):
            k, d, _ = ex(sq.frame(t))
            m = len(k)
            kk = np.zeros((cap, 7), np.float32); kk.view(np.uint8).reshape(cap, 28)[:m] = k.view(np.uint8).reshape(m, 28)
            dd = np.zeros((cap, 32), np.uint8); dd[:m] = d
            de = np.full(cap, -1, np.float32); de[:m] = k["response"]          # any per-keypoint float payload
            uu = np.full(cap, -1, np.float32); uu[:m] = k["x"] - 100.0 / np.maximum(k["response"], 1)
            n.append(m); kp.append(kk); desc.append(dd); dep.append(de); ur.append(uu)
    return (torch.tensor(n, dtype=torch.int32), torch.from_numpy(np.stack(kp)), torch.from_numpy(np.stack(desc)),
            torch.from_numpy(np.stack(dep)), torch.from_numpy(np.stack(ur)))

def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    w, h, cap, n_seq, fps = 200, 160, 400, 4, 2
    mine = sharding.sequences_of_rank(n_seq, world, rank)
    send = sharding.pack_records(*records_for(mine, fps, w, h, cap))
    got = sharding.gather_records(send, dst=0)
    if rank == 0:
        # single-process result for every sequence, in (rank, local order) order
        ok = True
        for r in range(world):
            ref = sharding.pack_records(*records_for(sharding.sequences_of_rank(n_seq, world, r), fps, w, h, cap))
            ok &= torch.equal(got[r], ref)
            frames = sharding.unpack_records(got[r], cap)
            ok &= all(f["n"] > 0 and f["desc"].shape == (f["n"], 32) for f in frames)
        print("GATHER_OK" if ok else "GATHER_MISMATCH")
    dist.barrier()
    dist.destroy_process_group()

main()
'''


def test_chunk_partition_covers_all_frames_with_halo():
    for n, world in ((4541, 8), (10, 3), (7, 8), (64, 1)):
        seen = []
        for r in range(world):
            b, e, e_halo = sharding.frame_chunk(n, world, r)
            seen.extend(range(b, e))
            assert e_halo == min(e + 1, n)
        assert seen == list(range(n))
    assert sharding.sequences_of_rank(11, 8, 2) == [2, 10]


def test_pack_unpack_roundtrip():
    import torch
    rng = np.random.default_rng(0)
    B, cap = 3, 50
    n = torch.tensor([50, 0, 17], dtype=torch.int32)
    kp = torch.from_numpy(rng.standard_normal((B, cap, 7)).astype(np.float32))
    desc = torch.from_numpy(rng.integers(0, 256, (B, cap, 32), dtype=np.uint8))
    dep = torch.from_numpy(rng.standard_normal((B, cap)).astype(np.float32))
    ur = torch.from_numpy(rng.standard_normal((B, cap)).astype(np.float32))
    buf = sharding.pack_records(n, kp, desc, dep, ur)
    assert buf.shape == (B, sharding.record_bytes(cap))
    fr = sharding.unpack_records(buf, cap)
    for i in range(B):
        m = int(n[i])
        assert fr[i]["n"] == m
        assert np.array_equal(fr[i]["desc"], desc[i, :m].numpy())
        assert np.array_equal(fr[i]["kp"], kp[i, :m].numpy().view(np.uint8).reshape(m, 28))
        assert np.array_equal(fr[i]["depth"], dep[i, :m].numpy()) and np.array_equal(fr[i]["uright"], ur[i, :m].numpy())


def test_two_rank_gather_equals_single_process(tmp_path, oracle):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, RGBL_ROOT=ROOT, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert "GATHER_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
