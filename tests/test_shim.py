"""The C++ drop-in classes ORB_SLAM3::ORBextractor / DepthModule / ORBmatcher (orb_slam3_rgbl_amd/shim) used
the way the reference's System/Tracking/Frame/LocalMapping use theirs."""
import os

import pytest

import shim_driver

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
