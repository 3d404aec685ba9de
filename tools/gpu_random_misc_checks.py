"""Randomised GPU-vs-oracle checks of the depth module (image sizes that are no multiples of the 64 x 32 dilation tile, every
Diamond size, the other structuring elements) and of the one-pair Hamming scan (train sets around the launch-slice and sweep
boundaries).  Usage (on an MI355X): python tools/gpu_random_misc_checks.py [seed] [cases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch  # noqa: F401  (HIP runtime order)
import parity_checks as pc
from orb_slam3_rgbl_amd import _lib as L, frontend as F

lib = L.load()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for i in range(n_cases):
    w, h = int(rng.integers(130, 1400)), int(rng.integers(70, 520))
    shape = int(rng.choice([F.KERNEL_DIAMOND, F.KERNEL_DIAMOND, F.KERNEL_RECT, F.KERNEL_CROSS, F.KERNEL_ELLIPSE]))
    ku = int(rng.choice([3, 5, 7, 9]))
    kv = int(rng.choice([3, 5, 7, 9]))
    method = int(rng.choice([F.UPS_INVERSE_DILATION] * 3 + [F.UPS_AVERAGE_FILTERING, F.UPS_NEAREST_NEIGHBOR_PIXEL]))
    n = pc.check_depth(lib, method, w=w, h=h, seed=int(rng.integers(0, 1000)), n_az=int(rng.integers(300, 2000)), kernel=(shape, ku, kv),
                       n_kp=int(rng.integers(1, 2500)))
    print("ok depth %4dx%-4d method %d kernel %d %dx%d -> %d keypoints with depth" % (w, h, method, shape, ku, kv, n), flush=True)
for i in range(n_cases):
    na = int(rng.choice([1, 63, 64, 255, 256, 257, 700, 2000, 2049, 5000]))
    nb = int(rng.choice([1, 63, 64, 65, 255, 256, 257, 511, 1024, 2000, 4095, 4097, 8191, 8192, 8193, 12000, 20000]))
    pc.check_matcher_bf(lib, na, nb, seed=int(rng.integers(0, 1000)))
    print("ok hamming %5d x %5d" % (na, nb), flush=True)
print("all cases bit-exact")
