"""ctypes binding of oracle/_ref - the reference's OWN ORBmatcher.cc / Frame::ComputeStereoMatches / DBoW2 compiled here
(oracle/Makefile `ref`).  TEST INFRASTRUCTURE: imported by tests/ and by bench.py's cpu_baseline legs only, never by the
product.  Every entry point reports the wall time of the call INTO the reference's function (`last_call_seconds`), not of the
stand-in objects the glue builds around it.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")


class KfArrays(C.Structure):
    _fields_ = [("n", C.c_int), ("desc", C.c_void_p), ("kp_xy", C.c_void_p), ("kp_octave", C.c_void_p), ("kp_angle", C.c_void_p),
                ("uright", C.c_void_p), ("has_mp", C.c_void_p), ("nnodes", C.c_int), ("node_id", C.c_void_p),
                ("node_off", C.c_void_p), ("node_feat", C.c_void_p)]


def kf_arrays(kf, keep):
    a = KfArrays()

    def arr(v, dt):
        x = np.ascontiguousarray(v, dt)
        keep.append(x)
        return x.ctypes.data
    a.n = len(kf["desc"])
    a.desc, a.kp_xy = arr(kf["desc"], np.uint8), arr(kf["xy"], np.float32)
    a.kp_octave, a.kp_angle = arr(kf["octave"], np.int32), arr(kf["angle"], np.float32)
    a.uright, a.has_mp = arr(kf["uright"], np.float32), arr(kf["has_mp"], np.uint8)
    a.nnodes = len(kf["node_id"])
    a.node_id, a.node_off, a.node_feat = arr(kf["node_id"], np.int32), arr(kf["node_off"], np.int32), arr(kf["node_feat"], np.int32)
    return a


def _load(name):
    path = os.path.join(REF_DIR, name)
    if not os.path.exists(path):
        return None
    try:
        return C.CDLL(path)
    except OSError:
        return None


def load_matcher():
    """libref_orbmatcher.so with argtypes set, or None when oracle/_ref is not built."""
    lib = _load("libref_orbmatcher.so")
    if lib is None:
        return None
    lib.ref_last_call_seconds.restype = C.c_double
    lib.ref_search_triangulation.restype = C.c_int
    lib.ref_search_triangulation.argtypes = [C.POINTER(KfArrays), C.POINTER(KfArrays)] + [C.c_void_p] * 3 + [C.c_int] + \
        [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_void_p] * 4
    for name in ("ref_search_by_projection", "ref_search_local_points", "ref_search_by_projection_kf"):
        getattr(lib, name).restype = C.c_int
        getattr(lib, name).argtypes = [C.c_void_p, C.c_void_p]
    lib.ref_fuse.restype = C.c_int
    lib.ref_fuse.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ref_search_for_initialization.restype = C.c_int
    lib.ref_search_for_initialization.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ref_search_by_bow_kf.restype = C.c_int
    lib.ref_search_by_bow_kf.argtypes = [C.POINTER(KfArrays), C.POINTER(KfArrays), C.c_float, C.c_int, C.c_void_p]
    lib.ref_search_by_bow.restype = C.c_int
    lib.ref_search_by_bow.argtypes = [C.POINTER(KfArrays), C.POINTER(KfArrays), C.c_float, C.c_int, C.c_void_p]
    lib.ref_search_by_bow_rig.restype = C.c_int
    lib.ref_search_by_bow_rig.argtypes = [C.POINTER(KfArrays), C.c_int, C.POINTER(KfArrays), C.c_int, C.c_float, C.c_int, C.c_void_p]
    return lib


def load_frame():
    lib = _load("libref_frame.so")
    if lib is None:
        return None
    lib.ref_frame_last_call_seconds.restype = C.c_double
    lib.ref_stereo_matches.restype = C.c_int
    lib.ref_stereo_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                                       C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                       C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_void_p]
    return lib


def load_dbow2():
    lib = _load("libref_dbow2.so")
    if lib is None:
        return None
    lib.ref_voc_last_call_seconds.restype = C.c_double
    lib.ref_voc_load_text.restype = C.c_void_p
    lib.ref_voc_load_text.argtypes = [C.c_char_p]
    lib.ref_voc_destroy.argtypes = [C.c_void_p]
    lib.ref_voc_transform.restype = C.c_int
    lib.ref_voc_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int),
                                      C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    return lib


def search_triangulation(lib, kf1, kf2, K, scale_factors, level_sigma2, q1, t1, q2, t2, only_stereo, coarse, check_orientation):
    """ORBmatcher(0.6, check_orientation).SearchForTriangulation(&kf1, &kf2, pairs, only_stereo, coarse) on stand-in key frames.
    Returns (matches12, nmatches, R12, t12, epipole, seconds inside the reference's function)."""
    keep = []
    a1, a2 = kf_arrays(kf1, keep), kf_arrays(kf2, keep)
    arrs = [np.ascontiguousarray(v, np.float32) for v in (K, scale_factors, level_sigma2, q1, t1, q2, t2)]
    m = np.zeros(a1.n, np.int32)
    R12, t12, ep = np.zeros(9, np.float32), np.zeros(3, np.float32), np.zeros(2, np.float32)
    nm = lib.ref_search_triangulation(C.byref(a1), C.byref(a2), arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data,
                                      len(arrs[1]), arrs[3].ctypes.data, arrs[4].ctypes.data, arrs[5].ctypes.data, arrs[6].ctypes.data,
                                      int(only_stereo), int(coarse), int(check_orientation), m.ctypes.data, R12.ctypes.data,
                                      t12.ctypes.data, ep.ctypes.data)
    return m, nm, R12, t12, ep, lib.ref_last_call_seconds()


def call_struct(lib, name, P, n_out):
    """ref_search_by_projection / ref_search_local_points on an oracle_py input block.  Returns (match, nmatches, seconds)."""
    m = np.zeros(n_out, np.int32)
    nm = getattr(lib, name)(C.byref(P), m.ctypes.data)
    return m, nm, lib.ref_last_call_seconds()


def search_by_bow(lib, kf, frame, nnratio, check_orientation):
    keep = []
    a1, a2 = kf_arrays(kf, keep), kf_arrays(frame, keep)
    m = np.zeros(a2.n, np.int32)
    nm = lib.ref_search_by_bow(C.byref(a1), C.byref(a2), C.c_float(nnratio), int(check_orientation), m.ctypes.data)
    return m, nm, lib.ref_last_call_seconds()


def search_by_bow_rig(lib, kf, kf_n_left, frame, n_left, nnratio, check_orientation):
    """The reference's SearchByBoW on a two-camera frame (F.Nleft = n_left, F.mpCamera2 set); kf_n_left >= 0: the key frame is a
    two-camera one as well (pKF->mpCamera2 set, pKF->NLeft = kf_n_left)."""
    keep = []
    a1, a2 = kf_arrays(kf, keep), kf_arrays(frame, keep)
    m = np.zeros(a2.n, np.int32)
    nm = lib.ref_search_by_bow_rig(C.byref(a1), int(kf_n_left), C.byref(a2), int(n_left), C.c_float(nnratio), int(check_orientation), m.ctypes.data)
    return m, nm, lib.ref_last_call_seconds()


def search_by_bow_kf(lib, kf1, kf2, nnratio, check_orientation):
    keep = []
    a1, a2 = kf_arrays(kf1, keep), kf_arrays(kf2, keep)
    m = np.zeros(a1.n, np.int32)
    nm = lib.ref_search_by_bow_kf(C.byref(a1), C.byref(a2), C.c_float(nnratio), int(check_orientation), m.ctypes.data)
    return m, nm, lib.ref_last_call_seconds()


def fuse(lib, P, kf_state2):
    """ORBmatcher().Fuse(pKF, vpMapPoints, th) on an oracle_py fuse input block.  Returns (best feature per point as far as the
    function left a trace, nFused, seconds)."""
    best = np.zeros(P.n1, np.int32)
    st = np.ascontiguousarray(kf_state2, np.uint8)
    nf = lib.ref_fuse(C.byref(P), st.ctypes.data, best.ctypes.data)
    return best, nf, lib.ref_last_call_seconds()


def search_for_initialization(lib, P, prev_matched):
    prev = np.ascontiguousarray(prev_matched, np.float32).copy()
    m = np.zeros(P.n1, np.int32)
    nm = lib.ref_search_for_initialization(C.byref(P), prev.ctypes.data, m.ctypes.data)
    return m, prev, nm, lib.ref_last_call_seconds()


def voc_transform(lib, h, desc, levelsup):
    desc = np.ascontiguousarray(desc, np.uint8)
    n = len(desc)
    wid, wval = np.zeros(n, np.uint32), np.zeros(n, np.float64)
    nid, noff, nfeat = np.zeros(n, np.uint32), np.zeros(n + 1, np.int32), np.zeros(n, np.uint32)
    nw, nn = C.c_int(0), C.c_int(0)
    rc = lib.ref_voc_transform(h, desc.ctypes.data, n, levelsup, n, wid.ctypes.data, wval.ctypes.data, C.byref(nw), n,
                               nid.ctypes.data, noff.ctypes.data, nfeat.ctypes.data, C.byref(nn))
    if rc != 0:
        raise RuntimeError("ref_voc_transform failed")
    return (wid[:nw.value], wval[:nw.value], nid[:nn.value], noff[:nn.value + 1], nfeat[:noff[nn.value]]), lib.ref_voc_last_call_seconds()


def stereo_matches(lib, left, right, nfeatures, ini, mn, mb, mbf, scale=1.2, levels=8):
    """The reference's two ORBextractor calls + Frame::ComputeStereoMatches.  Returns (mvuRight, mvDepth, n_left, n_right,
    seconds inside ComputeStereoMatches)."""
    h, w = left.shape
    cap = nfeatures * 2 + 4096
    from . import oracle_py as O
    kl, kr = np.zeros(cap, O.KP_DTYPE), np.zeros(cap, O.KP_DTYPE)
    dl, dr = np.zeros((cap, 32), np.uint8), np.zeros((cap, 32), np.uint8)
    ur, dp = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
    nl, nr = C.c_int(0), C.c_int(0)
    rc = lib.ref_stereo_matches(left.ctypes.data, right.ctypes.data, w, h, left.strides[0], nfeatures, scale, levels, ini, mn, mb, mbf,
                                kl.ctypes.data, dl.ctypes.data, kr.ctypes.data, dr.ctypes.data, cap, C.byref(nl), C.byref(nr),
                                ur.ctypes.data, dp.ctypes.data)
    if rc != 0:
        raise RuntimeError("ref_stereo_matches failed")
    return ur[:nl.value], dp[:nl.value], nl.value, nr.value, lib.ref_frame_last_call_seconds()
