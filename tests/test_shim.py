"""The C++ drop-in classes ORB_SLAM3::ORBextractor / DepthModule / ORBmatcher (orb_slam3_rgbl_amd/shim) used
the way the reference's System/Tracking/Frame/LocalMapping use theirs."""
import os

import pytest

import shim_driver
from yaml_cases import PARSE_CASES, yaml_with

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_shims_under_emulation(emu_lib, tmp_path):
    exe = os.path.join(ROOT, "tests", "_build", "shim_test_emu")
    shim_driver.build(os.path.join(ROOT, "tests", "_build"), "rgbl_frontend_emu", exe)
    shim_driver.run_and_check(exe, str(tmp_path))


@pytest.mark.gpu
def test_cpp_shims_on_mi355x(gpu_lib, tmp_path):
    exe = os.path.join(ROOT, "tests", "_build", "shim_test_gpu")
    shim_driver.build(os.path.join(ROOT, "orb_slam3_rgbl_amd"), "rgbl_frontend", exe)
    shim_driver.run_and_check(exe, str(tmp_path))


def test_shim_depthmodule_parse_outcomes(emu_lib, tmp_path):
    """Settings-file failure modes of the drop-in DepthModule == the reference parser's (same table that
    tests/test_reference_build.py checks against the reference source itself)."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "_build", "shim_test_emu")
    shim_driver.build(os.path.join(ROOT, "tests", "_build"), "rgbl_frontend_emu", exe)
    for i, (edits, drop, expect) in enumerate(PARSE_CASES):
        res = subprocess.run([exe, "--parse", yaml_with(tmp_path, "c%d.yaml" % i, edits, drop)], capture_output=True, text=True, timeout=120)
        assert res.returncode == 0, res.stdout + res.stderr
        flags = int(res.stdout.strip().splitlines()[-1].split()[1])
        assert bool(flags & 1) == expect[0], (edits, drop, res.stdout)
        if expect[1] is not None:
            assert bool(flags & 2) == expect[1], (edits, drop, res.stdout)
        else:
            assert not (flags & 1)


@pytest.mark.parametrize("name,expect", [("KITTI00-02.yaml", 3), ("KITTI04-12.yaml", 1), ("KITTIxx-03.yaml", 1)])
def test_shim_depthmodule_on_the_references_shipped_settings(emu_lib, name, expect):
    """The three settings files of Examples/RGB-L, read in place: the drop-in parser must end where the reference's own parser
    ends (tests/test_reference_build.py::test_reference_disables_upsampling_on_its_other_shipped_settings) - the files for
    sequences 03 - 12 carry a stale key, up-sampling stays disabled (DepthModule.cc:566-582)."""
    import subprocess
    path = os.path.join("/root/reference/Examples/RGB-L", name)
    if not os.path.exists(path):
        pytest.skip("the reference's Examples/RGB-L is not on this host")
    exe = os.path.join(ROOT, "tests", "_build", "shim_test_emu")
    shim_driver.build(os.path.join(ROOT, "tests", "_build"), "rgbl_frontend_emu", exe)
    res = subprocess.run([exe, "--parse", path], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    assert int(res.stdout.strip().splitlines()[-1].split()[1]) == expect, res.stdout
