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


# ---- rows f1 / f3 on photographs read in place (tests/real_images.py) --------------------------------
def check_real_stereo_pair(lib):
    """Two extractors + Frame::ComputeStereoMatches on the Middlebury 'Motorcycle' pair, device against oracle, bit for bit
    (the oracle itself is held to the reference's own lines on this pair in tests/test_reference_build.py)."""
    import parity_checks as pc
    import real_images as R
    lrgb, rrgb, _ = R.stereo_pair()
    h, w = lrgb.shape[:2]
    nf, ini, mn, mb, mbf = 1500, 20, 7, R.MOTORCYCLE_MB, R.MOTORCYCLE_MBF
    exl = F.ORBextractor(nf, 1.2, 8, ini, mn, w, h, lib=lib)
    exr = F.ORBextractor(nf, 1.2, 8, ini, mn, w, h, lib=lib)
    kl, dl, _, gl = exl.extract_color(lrgb, True)      # Tracking::GrabImageStereo converts both images first (Tracking.cc:1478-1503)
    kr, dr, _, gr = exr.extract_color(rrgb, True)
    ogl, ogr = O.cvt_gray(lrgb, True), O.cvt_gray(rrgb, True)
    assert np.array_equal(gl, ogl) and np.array_equal(gr, ogr)
    ol, orr = O.Extractor(nf, 1.2, 8, ini, mn), O.Extractor(nf, 1.2, 8, ini, mn)
    okl, odl, _ = ol(ogl)
    okr, odr, _ = orr(ogr)
    pc.assert_keypoints_equal(kl, okl, "left")
    pc.assert_keypoints_equal(kr, okr, "right")
    assert np.array_equal(dl, odl) and np.array_equal(dr, odr)
    ur, dp = F.ComputeStereoMatches(exl, exr, kl, dl, kr, dr, mb, mbf)
    our, odp = O.stereo_matches(ol, orr, okl, odl, okr, odr, mb, mbf)
    assert np.array_equal(pc.bits(ur), pc.bits(our)), "mvuRight"
    assert np.array_equal(pc.bits(dp), pc.bits(odp)), "mvDepth"
    assert (our >= 0).sum() > 400
    exl.close(); exr.close()


def check_real_color_photo(lib, name):
    """cvtColor -> extract on a colour photograph (rgbl_extract_color) against oracle cvtColor + oracle extractor."""
    import parity_checks as pc
    import real_images as R
    rgb = R.color_photo(name)
    h, w = rgb.shape[:2]
    ex = F.ORBextractor(1200, 1.2, 8, 12, 7, w, h, lib=lib)
    orc = O.Extractor(1200, 1.2, 8, 12, 7)
    for mbRGB in (True, False):
        kps, desc, mono, gray = ex.extract_color(rgb, mbRGB)
        ogray = O.cvt_gray(rgb, mbRGB)
        assert np.array_equal(gray, ogray), "mImGray"
        okps, odesc, omono = orc(ogray)
        pc.assert_keypoints_equal(kps, okps, "%s RGB=%s" % (name, mbRGB))
        assert np.array_equal(desc, odesc) and mono == omono
        assert len(kps) > 800
    ex.close()


def test_emulated_kernels_on_a_real_stereo_pair(emu_lib):
    check_real_stereo_pair(emu_lib)


@pytest.mark.parametrize("name", ["astronaut.png", "coffee.png"])
def test_emulated_kernels_on_a_colour_photograph(emu_lib, name):
    check_real_color_photo(emu_lib, name)


@pytest.mark.gpu
def test_gpu_on_a_real_stereo_pair(gpu_lib):
    check_real_stereo_pair(gpu_lib)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["astronaut.png", "coffee.png"])
def test_gpu_on_a_colour_photograph(gpu_lib, name):
    check_real_color_photo(gpu_lib, name)
