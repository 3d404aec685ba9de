"""Randomised GPU-vs-oracle extractor checks (tests/fuzz_cases.py; the seeded version runs as tests/test_fuzz_gpu.py).
Usage (on an MI355X): python tools/gpu_random_extractor_checks.py [seed] [cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401  (HIP runtime order)
import fuzz_cases
from orb_slam3_rgbl_amd import _lib as L

lib = L.load()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for _ in range(n_cases):
    print("ok", fuzz_cases.extractor_case(lib, rng), flush=True)
for _ in range(max(n_cases // 3, 4)):
    print("ok", fuzz_cases.low_contrast_case(lib, rng), flush=True)
print("all cases bit-exact")
