"""The C-ABI library loads and exports every symbol include/rgbl_frontend.h declares (no compute calls)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "rgbl_frontend.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rgbl_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    from orb_slam3_rgbl_amd import _lib
    assert declared_symbols() == sorted(_lib.SYMBOLS)


def test_product_library_exports_every_declared_symbol():
    from orb_slam3_rgbl_amd import _lib
    import __graft_entry__ as g
    if not os.path.exists(_lib.LIB_PATH):
        g.build()
    lib = _lib.bind(_lib.LIB_PATH)  # raises AttributeError on a missing export
    for name in declared_symbols():
        assert hasattr(lib, name)
    assert lib.rgbl_backend() == b"hip:gfx950"


def test_no_cpu_fallback_without_a_device():
    """Without a GPU the product library must refuse to create handles instead of computing on the host."""
    from orb_slam3_rgbl_amd import _lib
    lib = _lib.load()
    if lib.rgbl_device_count() > 0:
        pytest.skip("a HIP device is visible")
    cfg = _lib.ExtractorCfg(1000, 1.2, 8, 20, 7, 640, 480, 1)
    h = C.c_void_p()
    assert lib.rgbl_extractor_create(C.byref(cfg), 0, C.byref(h)) == _lib.ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.rgbl_last_error()
    m = C.c_void_p()
    assert lib.rgbl_matcher_create(0, C.byref(m)) == _lib.ERR_NO_DEVICE


def test_product_never_imports_the_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "orb_slam3_rgbl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".cc", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle_py" not in txt and "liborbslam_oracle" not in txt and "oracle/" not in txt.replace("oracle/cvcompat", ""), f


def test_invalid_arguments_are_reported_not_thrown(emu_lib):
    from orb_slam3_rgbl_amd import _lib
    lib = emu_lib
    h = C.c_void_p()
    bad = _lib.ExtractorCfg(1000, 1.2, 40, 20, 7, 640, 480, 1)  # too many levels
    assert lib.rgbl_extractor_create(C.byref(bad), 0, C.byref(h)) == _lib.ERR_INVALID
    portrait = _lib.ExtractorCfg(1000, 1.2, 4, 20, 7, 200, 900, 1)  # round(w/h) == 0 roots: the reference divides by zero
    assert lib.rgbl_extractor_create(C.byref(portrait), 0, C.byref(h)) == _lib.ERR_INVALID
    small = _lib.ExtractorCfg(1000, 1.2, 8, 20, 7, 160, 120, 1)  # top levels smaller than one detection cell
    assert lib.rgbl_extractor_create(C.byref(small), 0, C.byref(h)) == _lib.ERR_INVALID
    d = _lib.DepthCfg()
    d.method, d.width, d.height, d.max_points, d.max_keypoints, d.max_batch, d.kernel_w, d.kernel_h = 5, 64, 64, 10, 10, 1, 5, 5
    assert lib.rgbl_depth_create(C.byref(d), 0, C.byref(h)) == _lib.ERR_INVALID  # IPBasic: never implemented upstream
    assert lib.rgbl_structuring_element(3, 4, 4, None) == _lib.ERR_INVALID


def test_invalid_arguments_of_the_matcher_entry_points(emu_lib, tmp_path):
    """The tracking matchers and the vocabulary reject bad input with a status code and a message (never throw)."""
    import numpy as np
    from orb_slam3_rgbl_amd import _lib, frontend as F, synth
    import parity_checks as pc
    lib = emu_lib
    mt = F.ORBmatcher(0.9, True, lib=lib)
    case = pc.make_projection_case(40, 50, 3)
    bad = dict(case, octave1=np.full(40, 9, np.int32))            # octave beyond the scale table
    with pytest.raises(_lib.RgblError):
        mt.SearchByProjection(bad, 7.0, False)
    lcase = pc.make_local_points_case(40, 50, 3)
    bad = dict(lcase, level1=np.full(40, -1, np.int32))
    with pytest.raises(_lib.RgblError):
        mt.SearchLocalPoints(bad, 1.0)
    n = _lib.C.c_int(0)
    assert lib.rgbl_search_by_projection(mt.h, None, None, _lib.C.byref(n)) == _lib.ERR_INVALID
    assert lib.rgbl_search_local_points(mt.h, None, None, _lib.C.byref(n)) == _lib.ERR_INVALID
    assert lib.rgbl_search_by_projection_keyframe(mt.h, None, None, _lib.C.byref(n)) == _lib.ERR_INVALID
    icase = pc.make_initialization_case(60, 3)
    with pytest.raises(_lib.RgblError):
        mt.SearchForInitialization(dict(icase, kp1_octave=np.full(60, -1, np.int32)), 100)
    assert lib.rgbl_search_for_initialization(mt.h, None, None, None, _lib.C.byref(n)) == _lib.ERR_INVALID
    rcase = pc.make_relocalization_case(40, 50, 3)
    rcase = dict(rcase, valid1=np.ones(40, np.uint8), level1=np.zeros(40, np.int32))
    with pytest.raises(_lib.RgblError):
        mt.SearchByProjectionKeyFrame(rcase, 10.0, 256)                      # ORBdist = 256 would index mvpMapPoints[-1] upstream
    with pytest.raises(_lib.RgblError):
        mt.SearchByProjectionKeyFrame(dict(rcase, level1=np.full(40, 8, np.int32)), 10.0, 100)
    mt.close()
    # vocabulary files
    h = _lib.C.c_void_p()
    p = tmp_path / "bad.txt"
    p.write_text("10 6 0 0 extra\n")
    assert lib.rgbl_vocabulary_load_text(str(p).encode(), 0, _lib.C.byref(h)) == _lib.ERR_INVALID      # no nodes
    p.write_text("40 6 0 0\n")
    assert lib.rgbl_vocabulary_load_text(str(p).encode(), 0, _lib.C.byref(h)) == _lib.ERR_INVALID      # k out of range, as upstream
    assert b"not a correct text file" in lib.rgbl_last_error()
    p.write_text("10 3 2 0\n0 1 " + " ".join(["0"] * 32) + " 1.0")
    assert lib.rgbl_vocabulary_load_text(str(p).encode(), 0, _lib.C.byref(h)) == _lib.ERR_INVALID      # scoring other than L1_NORM
    p.write_text("10 3 0 0\n5 1 " + " ".join(["0"] * 32) + " 1.0")
    assert lib.rgbl_vocabulary_load_text(str(p).encode(), 0, _lib.C.byref(h)) == _lib.ERR_INVALID      # parent that does not exist yet
    voc = synth.make_vocabulary(4, 2, 0)
    V = F.ORBVocabulary(lib=lib).from_arrays(synth.vocabulary_arrays(voc))
    nw, nn = _lib.C.c_int(0), _lib.C.c_int(0)
    desc = synth.descriptors(50, 1)
    wid, wval = np.zeros(2, np.uint32), np.zeros(2, np.float64)
    nid, noff, nfeat = np.zeros(50, np.uint32), np.zeros(51, np.int32), np.zeros(50, np.uint32)
    rc = lib.rgbl_bow_transform(V.h, desc.ctypes.data, 50, 1, wid.ctypes.data, wval.ctypes.data, 2, _lib.C.byref(nw), nid.ctypes.data,
                                noff.ctypes.data, nfeat.ctypes.data, 50, _lib.C.byref(nn))
    assert rc == _lib.ERR_CAPACITY and nw.value > 2      # capacity too small: reported, count returned
    V.close()
