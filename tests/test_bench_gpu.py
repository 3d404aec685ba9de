"""bench.py's own contract on the hardware: one JSON line, the LAST thing on stdout, with the gather inside the timed steps
(a one-rank RCCL communicator through the library's rgbl_gather_* entry points) and the chunk-with-halo sharding."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "64",
                          "--no-extras", "--no-cpu-baseline"] + list(args), capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-3000:]
    lines = res.stdout.strip().splitlines()
    return json.loads(lines[-1]), lines      # the line is the last thing on stdout (RCCL's banner is flushed in front of it)


@pytest.mark.gpu
def test_bench_line_with_the_gather_over_rccl_at_one_rank(gpu_lib):
    d, _ = run_bench("--pg")
    assert d["n_gpus"] == 1 and d["metric"].startswith("RGB-L front-end frames/sec") and d["unit"] == "frames/s"
    assert d["parity_spot_check"].startswith("bit-exact"), d["parity_spot_check"]
    assert "rgbl_gather_* over RCCL" in d["config"]["gather_transport"] and d["config"]["gather"].startswith("step")
    assert d["roofline"]["kernel"] == "k_fast_cells" and d["value"] > 1000
    # what an N > 1 line must carry so that "RCCL saw N ranks" is on record - here with the one rank a box has
    m = d["multi_gpu"]
    assert m["rccl"]["world"] == 1 and m["rccl"]["version"] > 20000 and "rgbl_comm_info" in m["rccl"]["source"]
    assert len(m["per_rank_frames_per_s"]) == 1 and m["per_rank_frames_per_s"][0] > 1000
    assert m["gathered_bytes_per_step"] == m["gathered_bytes_per_step_per_rank"][0] > 64 * 500 * 68 and m["root_ingest_GB/s"] == 0.0
    st = d["roofline"]["kernels_ms_per_step_stats"]
    assert set(st) == set(d["roofline"]["kernels_ms_per_step"]) and all(v["min"] <= v["median"] <= v["max"] for v in st.values())


@pytest.mark.gpu
def test_bench_line_with_one_sequence_in_chunks(gpu_lib):
    d, _ = run_bench("--shard", "chunks")
    assert d["parity_spot_check"].startswith("bit-exact"), d["parity_spot_check"]      # the last owned frame met the halo frame
    assert d["config"]["parallelism"].startswith("one sequence of 64 frames in contiguous chunks")


@pytest.mark.gpu
def test_serialised_kernel_table_is_stable(gpu_lib):
    """the driver-visible per-kernel table (median of serialised steps) must not depend on what ran before
    it.  The Hamming scan was the outlier (0.643 ms in the driver's round-4 line against 0.335 under rocprofv3: a mean of three
    steps).  A fresh pipeline's table against the table of another fresh pipeline after the process has run gather legs and
    overlapped steps in between: every kernel above 0.1 ms within 15 %."""
    import numpy as np
    import torch

    sys.path.insert(0, ROOT)
    import bench
    from orb_slam3_rgbl_amd import frontend as F
    from orb_slam3_rgbl_amd import synth
    from orb_slam3_rgbl_amd.pipeline import FrontEndPipeline
    dev = torch.device("cuda", 0)
    w, h, nf, B = synth.KITTI_W, synth.KITTI_H, 2000, 128
    proj = F.projection_matrix(synth.KITTI_K, synth.KITTI_TR, gpu_lib)
    sq = synth.Sequence(0, w, h, n_frames=B, constant_density=True)
    frames = torch.from_numpy(np.stack([sq.frame(i) for i in range(B)])).to(dev)
    scans = [synth.lidar_scan(i, n_az=1900) for i in range(4)]
    cloud = torch.from_numpy(np.stack([scans[i % 4] for i in range(B)])).to(dev)
    n_points = scans[0].shape[1]

    def table(busy_first):
        pipe = FrontEndPipeline(gpu_lib, torch, dev, w, h, nf, proj, n_points, B, gather="step" if busy_first else "none")
        pipe.set_inputs(frames, cloud)
        for _ in range(10 if busy_first else 3):     # overlapped steps (with the one-rank gather choreography) in front of the leg
            pipe.step()
        pipe.finish(); pipe.sync()
        _, stats = bench.serial_kernel_leg(pipe, 12)
        pipe.close()
        return stats
    # the GPU is shared with the other pytest-xdist workers of a `-n 4` run: a comparison of two timings gets up to three tries
    for attempt in range(3):
        a = table(False)
        b = table(True)
        off = [(k, a[k], b[k]) for k in a if a[k]["median"] > 0.1 * B / 512 and abs(a[k]["median"] - b[k]["median"]) > 0.15 * a[k]["median"]]
        if not off:
            break
    assert not off, off
    # the outlier itself is real and stays visible as `max` / `max_at_step` (one step in twelve took 2.4 x when this test was
    # written); what the table reports - the median - sits within 15 % of the fastest step
    print("k_hamming_fp4 per-step stats: fresh", a["k_hamming_fp4"], "after load", b["k_hamming_fp4"])
    assert a["k_hamming_fp4"]["median"] <= 1.15 * a["k_hamming_fp4"]["min"], a["k_hamming_fp4"]
