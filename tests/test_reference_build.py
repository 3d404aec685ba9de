"""oracle/_ref: the reference's own src/ORBextractor.cc (compiled unmodified from /root/reference against
oracle/cvcompat, see oracle/Makefile `ref`) versus the restated oracle.  Pins everything in the extractor that is
not OpenCV-internal — detection-cell loop, two-threshold rule, the quad-tree with the real std::list / std::sort,
IC_Angle, steering and the pattern table, level scaling, vLappingArea packing — against the reference source itself.
The .so is built where /root/reference exists (this container); on the GPU box the prebuilt file is used."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle_py as O
from orb_slam3_rgbl_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_orbextractor.so")


@pytest.fixture(scope="module")
def ref(oracle):
    if os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    lib = C.CDLL(REF_SO)
    lib.ref_extract.restype = C.c_int
    lib.ref_extract.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    return lib


def run_ref(lib, img, nfeatures, nlevels, ini, mn, lap):
    cap = nfeatures * 2 + 4096
    kps = np.zeros(cap, O.KP_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    n = C.c_int(0)
    mono = lib.ref_extract(img.ctypes.data, img.shape[1], img.shape[0], img.strides[0], nfeatures, 1.2, nlevels, ini, mn,
                           lap[0], lap[1], kps.ctypes.data, desc.ctypes.data, cap, C.byref(n))
    return kps[:n.value], desc[:n.value], mono


@pytest.mark.parametrize("w,h,nf,nl,ini,mn,lap,seq", [
    (1241, 376, 2000, 8, 12, 7, (0, 0), 0),      # BASELINE cfg 2
    (1241, 376, 1000, 8, 12, 7, (0, 0), 1),      # cfg 1
    (1241, 376, 2000, 8, 20, 7, (0, 0), 2),      # stereo thresholds
    (752, 480, 1200, 8, 20, 7, (0, 400), 3),     # mono lapping area
    (1227, 370, 2000, 8, 12, 7, (0, 0), 4),      # one-pixel-wide border cells
    (640, 480, 5000, 5, 12, 7, (0, 0), 5),       # quota larger than the supply on the coarse levels
])
def test_reference_source_agrees_with_oracle(ref, w, h, nf, nl, ini, mn, lap, seq):
    img = synth.Sequence(seq, w, h, 1).frame(0)
    kps, desc, mono = run_ref(ref, img, nf, nl, ini, mn, lap)
    okps, odesc, omono = O.Extractor(nf, 1.2, nl, ini, mn)(img, lap)
    assert len(kps) == len(okps) and mono == omono
    for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        assert np.array_equal(kps[f].view(np.uint32), okps[f].view(np.uint32)), f
    assert np.array_equal(desc, odesc)


def test_reference_source_on_degenerate_images(ref):
    flat = np.full((200, 300), 90, np.uint8)
    assert len(run_ref(ref, flat, 500, 4, 20, 7, (0, 0))[0]) == 0
    yy, xx = np.mgrid[0:240, 0:320]
    chk = (((yy // 6) + (xx // 6)) % 2 * 255).astype(np.uint8)
    kps, desc, mono = run_ref(ref, chk, 800, 4, 20, 7, (0, 0))
    okps, odesc, omono = O.Extractor(800, 1.2, 4, 20, 7)(chk)
    assert len(kps) == len(okps) and np.array_equal(desc, odesc)
    assert np.array_equal(kps["x"], okps["x"]) and np.array_equal(kps["angle"].view(np.uint32), okps["angle"].view(np.uint32))
