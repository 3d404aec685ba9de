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


@pytest.mark.gpu
def test_bench_line_with_one_sequence_in_chunks(gpu_lib):
    d, _ = run_bench("--shard", "chunks")
    assert d["parity_spot_check"].startswith("bit-exact"), d["parity_spot_check"]      # the last owned frame met the halo frame
    assert d["config"]["parallelism"].startswith("one sequence of 64 frames in contiguous chunks")
