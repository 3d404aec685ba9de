"""bench.py's launch decision: a bare `python bench.py --gpus N` must either run N ranks or fail loudly -
never print a line whose n_gpus is not N - and SURVEY 8(d)'s step bytes are what `step_algorithmic_GB/s` is computed from."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_single_gpu_runs_in_place():
    assert bench.launch_plan(1, {}, 1, ["bench.py"]) == ("run", 1)
    assert bench.launch_plan(1, {}, 8, ["bench.py", "--steps", "3"]) == ("run", 1)


def test_bare_multi_gpu_command_line_reexecs_under_the_launcher():
    kind, argv = bench.launch_plan(8, {}, 8, ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "3"], port=29400)
    assert kind == "exec"
    assert argv[0] == sys.executable and argv[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in argv and argv[argv.index("--nproc-per-node") + 1] == "8"
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1" and argv[argv.index("--master-port") + 1] == "29400"
    assert argv[-7:] == ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "3"]   # the original command line, unchanged
    kind, argv = bench.launch_plan(2, {"PATH": "/bin"}, 4, ["bench.py", "--gpus", "2"])
    assert kind == "exec" and int(argv[argv.index("--master-port") + 1]) > 0


def test_too_few_gpus_is_an_error_not_a_smaller_run():
    kind, msg = bench.launch_plan(8, {}, 1, ["bench.py", "--gpus", "8"])
    assert kind == "error" and "8" in msg and "1" in msg
    kind, msg = bench.launch_plan(2, {}, 0, ["bench.py", "--gpus", "2"])
    assert kind == "error"


def test_under_the_launcher_the_world_must_be_what_gpus_says():
    env = {"WORLD_SIZE": "4", "RANK": "1", "LOCAL_RANK": "1"}
    assert bench.launch_plan(4, env, 8, ["bench.py", "--gpus", "4"]) == ("run", 4)
    kind, msg = bench.launch_plan(8, env, 8, ["bench.py", "--gpus", "8"])
    assert kind == "error" and "WORLD_SIZE=4" in msg
    kind, msg = bench.launch_plan(1, env, 8, ["bench.py"])          # torchrun with 4 ranks, --gpus left at its default
    assert kind == "error"
    kind, msg = bench.launch_plan(4, dict(env, LOCAL_RANK="3"), 2, ["bench.py", "--gpus", "4"])
    assert kind == "error" and "LOCAL_RANK" in msg
    assert bench.launch_plan(1, {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, 1, ["bench.py"]) == ("run", 1)


def test_command_line_fails_loudly_without_gpus():
    """Here (no GPU): `--gpus 2` exits non-zero with the reason and prints no JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env=env)
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        return   # a multi-GPU box: the run is the driver's business
    assert res.returncode != 0 and "not launching" in res.stderr and "{" not in res.stdout


def test_step_bytes_are_survey_8d():
    # K1 (1241 x 376, N = 121 600, K = 2000): B_ext 6.84 MB + B_dep 8.05 MB + B_m 0.144 MB = 15.0 MB
    b = bench.step_bytes_8d(1241, 376, 121600, 2000)
    px = [a * c for a, c in bench.level_sizes(1241, 376)]
    assert sum(px) == 1444097
    assert b == (5 * sum(px) - px[0] - px[-1] + 120000) + (20 * 121600 + 12 * 1241 * 376 + 24000) + 144000
    assert abs(b / 1e6 - 15.04) < 0.02
    assert abs(bench.step_bytes_8d(3840, 2160, 262144, 8000) / 1e6 - 225.4) < 0.6
