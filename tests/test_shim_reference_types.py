"""VERDICT r1 "missing 8": the drop-in ORBmatcher's member templates instantiated with reference-shaped callers.

The reference's Frame.cc / LocalMapping.cc / LoopClosing.cc cannot be compiled here (the whole SLAM system, real Eigen /
Sophus / OpenCV / g2o behind them).  What CAN be done: every member template of orb_slam3_rgbl_amd/shim/ORBmatcher.h is
instantiated (compile only) with the very stand-in classes the reference's own ORBmatcher.cc is compiled against in
oracle/_ref (oracle/cvcompat/orbslam_types.h: KeyFrame, Frame, MapPoint, GeometricCamera, Sophus::SE3f / Sim3f,
DBoW2::FeatureVector, cv::Mat), with the argument shapes of the reference's call sites (tests/shim_ref_types.cpp).  A member
the shim reads that the reference's classes do not have, or has with another type, fails this build."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "include", "ORBextractor.h")),
                    reason="the stand-in types include the reference's ORBextractor.h; /root/reference is not here")
def test_shim_templates_compile_against_the_references_class_shapes():
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wno-unused-parameter", "-Wno-unused-variable", "-Wno-sign-compare",
           "-I" + os.path.join(ROOT, "oracle", "cvcompat"), "-I" + os.path.join(ROOT, "oracle"),
           "-I" + os.path.join(ROOT, "orb_slam3_rgbl_amd", "shim"), "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(REF, "include"), os.path.join(ROOT, "tests", "shim_ref_types.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
