"""Randomised shapes against the oracle on the hardware (the former tools/gpu_random_*_checks.py as tests).
Fixed seeds so that the driver's `-m gpu` run is reproducible; RGBL_FUZZ_SECONDS=<s> adds fresh, time-boxed seeds (printed, so a
failure can be replayed with RGBL_FUZZ_SEED)."""
import os
import time

import numpy as np
import pytest

import fuzz_cases

SEEDS = {"extractor": (1, 2, 3, 4), "low_contrast": (11, 12), "depth": (21, 22, 23, 24), "hamming": (31, 32, 33, 34),
         "greedy_search": (41, 42, 43, 44, 45, 46), "node_search": (51, 52, 53, 54)}
PER_SEED = {"extractor": 3, "low_contrast": 2, "depth": 3, "hamming": 4, "greedy_search": 4, "node_search": 4}


@pytest.mark.gpu
@pytest.mark.parametrize("kind,seed", [(k, s) for k in sorted(SEEDS) for s in SEEDS[k]])
def test_random_shapes_against_the_oracle(gpu_lib, kind, seed):
    rng = np.random.default_rng(seed)
    for _ in range(PER_SEED[kind]):
        print(fuzz_cases.CASES[kind](gpu_lib, rng))


@pytest.mark.gpu
def test_random_shapes_time_boxed(gpu_lib):
    budget = float(os.environ.get("RGBL_FUZZ_SECONDS", "0"))
    if budget <= 0:
        pytest.skip("set RGBL_FUZZ_SECONDS to run fresh random cases for that long")
    seed = int(os.environ.get("RGBL_FUZZ_SEED", str(int(time.time()))))
    print("RGBL_FUZZ_SEED=%d" % seed)
    rng = np.random.default_rng(seed)
    t0 = time.time()
    kinds = sorted(fuzz_cases.CASES)
    i = 0
    while time.time() - t0 < budget:
        print(fuzz_cases.CASES[kinds[i % len(kinds)]](gpu_lib, rng))
        i += 1
