"""Builds tools/shim_latency.cpp against the product library and runs it on synthetic KITTI frames (on an MI355X)."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from orb_slam3_rgbl_amd import synth
SHIM = os.path.join(ROOT, "orb_slam3_rgbl_amd", "shim")
tmp = tempfile.mkdtemp()
exe = os.path.join(tmp, "shim_latency")
subprocess.check_call(["g++", "-O2", "-std=c++17", "-DRGBL_FORCE_CV_COMPAT", "-I" + SHIM, "-I" + os.path.join(ROOT, "include"),
                       os.path.join(ROOT, "tools", "shim_latency.cpp"), os.path.join(SHIM, "ORBextractor.cc"), os.path.join(SHIM, "DepthModule.cc"),
                       "-o", exe, "-L" + os.path.join(ROOT, "orb_slam3_rgbl_amd"), "-lrgbl_frontend", "-Wl,-rpath," + os.path.join(ROOT, "orb_slam3_rgbl_amd"), "-pthread"])
w, h, n = synth.KITTI_W, synth.KITTI_H, 8
seq = synth.Sequence(0, w, h, n_frames=n)
np.stack([seq.frame(i) for i in range(n)]).tofile(os.path.join(tmp, "frames.raw"))
cloud = np.ascontiguousarray(synth.lidar_scan(0), np.float32)
cloud.tofile(os.path.join(tmp, "cloud.raw"))
print(subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "KITTI00-02.yaml"), os.path.join(tmp, "frames.raw"), str(n), str(w), str(h),
                      os.path.join(tmp, "cloud.raw"), str(cloud.shape[1])], capture_output=True, text=True).stdout)
