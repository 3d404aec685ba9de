"""Natural-image fixtures (tests/golden/natural_*.png, made by tests/golden/make_natural_fixtures.py from two CC0 sample
images of scikit-image): real texture instead of the procedural generator's.  The committed golden vectors are the
oracle's own results - they catch oracle regressions; the kernels are held to the oracle on the same images."""
import hashlib
import json
import os

import numpy as np
import pytest
from PIL import Image

from oracle import oracle_py as O
from orb_slam3_rgbl_amd import frontend as F

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDEN = json.load(open(os.path.join(GOLD, "natural_golden.json")))
CONFIGS = {"kitti": (1000, 12, 7), "stereo": (1500, 20, 7)}


def load(name):
    img = np.asarray(Image.open(os.path.join(GOLD, "natural_%s.png" % name)), np.uint8)
    assert hashlib.sha256(img.tobytes()).hexdigest() == GOLDEN[name]["pixels_sha256"]
    return np.ascontiguousarray(img)


@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_oracle_reproduces_the_golden_vectors(oracle, name):
    img = load(name)
    for cfg, (nf, ini, mn) in CONFIGS.items():
        kps, desc, mono = O.Extractor(nf, 1.2, 8, ini, mn)(img)
        g = GOLDEN[name]["oracle"][cfg]
        assert len(kps) == g["n"] and mono == g["mono"]
        assert [int((kps["octave"] == l).sum()) for l in range(8)] == g["per_level"]
        assert hashlib.sha256(np.ascontiguousarray(kps).tobytes()).hexdigest() == g["keypoints_sha256"]
        assert hashlib.sha256(np.ascontiguousarray(desc).tobytes()).hexdigest() == g["descriptors_sha256"]


def check_device(lib, name):
    img = load(name)
    h, w = img.shape
    for cfg, (nf, ini, mn) in CONFIGS.items():
        ex = F.ORBextractor(nf, 1.2, 8, ini, mn, w, h, lib=lib)
        kps, desc, mono = ex(img)
        g = GOLDEN[name]["oracle"][cfg]
        assert len(kps) == g["n"] and mono == g["mono"]
        got = np.zeros(len(kps), O.KP_DTYPE)
        for f in O.KP_DTYPE.names:
            got[f] = kps[f]
        assert hashlib.sha256(got.tobytes()).hexdigest() == g["keypoints_sha256"], (name, cfg)
        assert hashlib.sha256(np.ascontiguousarray(desc).tobytes()).hexdigest() == g["descriptors_sha256"], (name, cfg)
        ex.close()


@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_emulated_kernels_on_natural_images(emu_lib, name):
    check_device(emu_lib, name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_gpu_on_natural_images(gpu_lib, name):
    check_device(gpu_lib, name)
