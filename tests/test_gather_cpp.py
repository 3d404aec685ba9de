"""The multi-GPU gather driven from C++ through the C ABI alone (tests/gather_test.cpp: rgbl_comm_* / rgbl_gather_*), one
process per rank, the RCCL unique id handed over through a file - the call sequence INTEGRATION.md shows for a C++ host.
CPU: two ranks against the SIMT-emulation library, whose RCCL is tests/emu/nccl_emu.cpp.  `-m gpu`: the product library and
the real RCCL with one rank (loopback send / receive), and with two ranks where the box has two GPUs."""
import fcntl
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "gather_test.cpp")


def build(libdir, libname, exe, emu):
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    deps = [SRC, os.path.join(ROOT, "include", "rgbl_frontend.h"), os.path.join(libdir, "lib%s.so" % libname)]
    with open(exe + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if os.path.exists(exe) and all(os.path.getmtime(exe) >= os.path.getmtime(d) for d in deps):
            return
        tmp = "%s.tmp.%d" % (exe, os.getpid())
        cmd = ["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), SRC, "-o", tmp, "-L" + libdir, "-l" + libname,
               "-Wl,-rpath," + libdir, "-pthread"]
        if emu:
            cmd.insert(1, "-DGATHER_TEST_EMU")
        else:
            cmd += ["-I/opt/rocm/include", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"]
        subprocess.check_call(cmd)
        os.replace(tmp, exe)


def run_ranks(exe, world, mode, steps, tmp_path, devices=None, env=None):
    id_file = str(tmp_path / ("id_%s_%d" % (mode, world)))
    procs = [subprocess.Popen([exe, str(world), str(r), id_file, mode, str(steps), str(devices[r] if devices else 0)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
    outs = [p.communicate(timeout=300) for p in procs]
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d: rc %d\n%s\n%s" % (r, p.returncode, so[-2000:], se[-3000:])
    assert "GATHER_CPP_OK" in outs[0][0], outs[0][0] + outs[0][1]
    return outs[0][0]


@pytest.mark.parametrize("world", [1, 2, 3, 8])     # 8: one rank per GPU of an MI355X node
@pytest.mark.parametrize("mode", ["step", "final"])
def test_gather_from_cpp_under_emulation(emu_lib, tmp_path, world, mode):
    exe = os.path.join(ROOT, "tests", "_build", "gather_test_emu")
    build(os.path.join(ROOT, "tests", "_build"), "rgbl_frontend_emu", exe, emu=True)
    run_ranks(exe, world, mode, 5, tmp_path, env=dict(os.environ, TMPDIR=str(tmp_path), RGBL_EMU_THREADS="1"))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["step", "final"])
def test_gather_from_cpp_over_rccl(gpu_lib, tmp_path, mode):
    """The product library + the real RCCL, driven from C++: one rank (its records travel through a grouped ncclSend / ncclRecv
    to itself), and two ranks on two GPUs where the box has them."""
    import torch
    exe = os.path.join(ROOT, "tests", "_build", "gather_test_gpu")
    build(os.path.join(ROOT, "orb_slam3_rgbl_amd"), "rgbl_frontend", exe, emu=False)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = run_ranks(exe, 1, mode, 5, tmp_path, env=env)
    assert " rccl 0" not in out        # a real RCCL reports its version
    if torch.cuda.device_count() >= 2:
        run_ranks(exe, 2, mode, 5, tmp_path, devices=[0, 1], env=env)
