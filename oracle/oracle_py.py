"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE — see oracle/oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (orb_slam3_rgbl_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liborbslam_oracle.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liborbslam_oracle.so"])


class DepthParams(C.Structure):
    _fields_ = [("proj", C.c_float * 12), ("min_dist", C.c_float), ("max_dist", C.c_float),
                ("mbf", C.c_float), ("method", C.c_int), ("kw", C.c_int), ("kh", C.c_int),
                ("kernel", C.c_uint8 * 81), ("avg_ksize", C.c_int), ("nn_radius", C.c_float)]


class ProjectionInput(C.Structure):
    """orc_projection_input == rgbl_projection_input (same layout)."""
    _fields_ = [("n1", C.c_int), ("valid1", C.c_void_p), ("world_pos1", C.c_void_p), ("mp_desc1", C.c_void_p),
                ("mp_observed1", C.c_void_p), ("octave1", C.c_void_p), ("angle1", C.c_void_p),
                ("n2", C.c_int), ("kp2_xy", C.c_void_p), ("kp2_octave", C.c_void_p), ("kp2_angle", C.c_void_p),
                ("uright2", C.c_void_p), ("desc2", C.c_void_p), ("grid", C.c_float * 6),
                ("Tcw_q", C.c_float * 4), ("Tcw_t", C.c_float * 3), ("Tlw_q", C.c_float * 4), ("Tlw_t", C.c_float * 3),
                ("K", C.c_float * 4), ("mb", C.c_float), ("mbf", C.c_float), ("scale_factors", C.c_void_p),
                ("n_levels", C.c_int), ("th", C.c_float), ("mono", C.c_int), ("check_orientation", C.c_int)]


def make_projection_input(case, th, mono, check_orientation, keep):
    """case: dict from tests/parity_checks.make_projection_case; keep: list that keeps the arrays alive."""
    def arr(v, dt):
        a = np.ascontiguousarray(v, dt)
        keep.append(a)
        return a.ctypes.data
    P = ProjectionInput()
    P.n1 = len(case["valid1"])
    P.valid1, P.world_pos1 = arr(case["valid1"], np.uint8), arr(case["world_pos1"], np.float32)
    P.mp_desc1, P.mp_observed1 = arr(case["mp_desc1"], np.uint8), arr(case["mp_observed1"], np.uint8)
    P.octave1, P.angle1 = arr(case["octave1"], np.int32), arr(case["angle1"], np.float32)
    P.n2 = len(case["kp2_xy"])
    P.kp2_xy, P.kp2_octave = arr(case["kp2_xy"], np.float32), arr(case["kp2_octave"], np.int32)
    P.kp2_angle, P.uright2, P.desc2 = arr(case["kp2_angle"], np.float32), arr(case["uright2"], np.float32), arr(case["desc2"], np.uint8)
    for name, n in (("grid", 6), ("Tcw_q", 4), ("Tcw_t", 3), ("Tlw_q", 4), ("Tlw_t", 3), ("K", 4)):
        for i in range(n):
            getattr(P, name)[i] = float(case[name][i])
    P.mb, P.mbf = float(case["mb"]), float(case["mbf"])
    P.scale_factors = arr(case["scale_factors"], np.float32)
    P.n_levels = len(case["scale_factors"])
    P.th, P.mono, P.check_orientation = float(th), int(mono), int(check_orientation)
    return P


def search_by_projection(case, th=7.0, mono=False, check_orientation=True):
    keep = []
    P = make_projection_input(case, th, mono, check_orientation, keep)
    m = np.zeros(P.n2, np.int32)
    n = lib().orc_search_by_projection(C.byref(P), _p(m))
    return m, n


class KfProjectionInput(C.Structure):
    """orc_kf_projection_input: the key frame's map points as the reference sees them."""
    _fields_ = [("n1", C.c_int), ("has_mp1", C.c_void_p), ("bad1", C.c_void_p), ("found1", C.c_void_p),
                ("world_pos1", C.c_void_p), ("mp_desc1", C.c_void_p), ("min_dist1", C.c_void_p), ("max_dist1", C.c_void_p),
                ("angle1", C.c_void_p), ("n2", C.c_int), ("kp2_xy", C.c_void_p), ("kp2_octave", C.c_void_p),
                ("kp2_angle", C.c_void_p), ("desc2", C.c_void_p), ("occupied2", C.c_void_p), ("grid", C.c_float * 6),
                ("Tcw_q", C.c_float * 4), ("Tcw_t", C.c_float * 3), ("K", C.c_float * 4), ("scale_factors", C.c_void_p),
                ("n_levels", C.c_int), ("log_scale_factor", C.c_float), ("th", C.c_float), ("orb_dist", C.c_int),
                ("check_orientation", C.c_int)]


def make_kf_projection_input(case, th, orb_dist, check_orientation, keep):
    """case: dict from tests/parity_checks.make_relocalization_case."""
    def arr(v, dt):
        a = np.ascontiguousarray(v, dt)
        keep.append(a)
        return a.ctypes.data
    P = KfProjectionInput()
    P.n1 = len(case["has_mp1"])
    P.has_mp1, P.bad1, P.found1 = arr(case["has_mp1"], np.uint8), arr(case["bad1"], np.uint8), arr(case["found1"], np.uint8)
    P.world_pos1, P.mp_desc1 = arr(case["world_pos1"], np.float32), arr(case["mp_desc1"], np.uint8)
    P.min_dist1, P.max_dist1 = arr(case["min_dist1"], np.float32), arr(case["max_dist1"], np.float32)
    P.angle1 = arr(case["angle1"], np.float32)
    P.n2 = len(case["kp2_xy"])
    P.kp2_xy, P.kp2_octave = arr(case["kp2_xy"], np.float32), arr(case["kp2_octave"], np.int32)
    P.kp2_angle, P.desc2 = arr(case["kp2_angle"], np.float32), arr(case["desc2"], np.uint8)
    P.occupied2 = arr(case["occupied2"], np.uint8)
    for name, n in (("grid", 6), ("Tcw_q", 4), ("Tcw_t", 3), ("K", 4)):
        for i in range(n):
            getattr(P, name)[i] = float(case[name][i])
    P.scale_factors = arr(case["scale_factors"], np.float32)
    P.n_levels = len(case["scale_factors"])
    P.log_scale_factor = float(case["log_scale_factor"])
    P.th, P.orb_dist, P.check_orientation = float(th), int(orb_dist), int(check_orientation)
    return P


def search_by_projection_kf(case, th=15.0, orb_dist=100, check_orientation=True):
    keep = []
    P = make_kf_projection_input(case, th, orb_dist, check_orientation, keep)
    m = np.zeros(P.n2, np.int32)
    n = lib().orc_search_by_projection_kf(C.byref(P), _p(m))
    return m, n


def kf_projection_prepass(case):
    """valid1 / level1 as the caller of rgbl_search_by_projection_keyframe has to provide them."""
    keep = []
    P = make_kf_projection_input(case, 1.0, 100, True, keep)
    valid = np.zeros(P.n1, np.uint8)
    level = np.zeros(P.n1, np.int32)
    lib().orc_kf_projection_prepass(C.byref(P), _p(valid), _p(level))
    return valid, level


class FuseInput(C.Structure):
    """orc_fuse_input: the candidate map points of ORBmatcher::Fuse as the reference sees them."""
    _fields_ = [("n1", C.c_int), ("has_mp1", C.c_void_p), ("bad1", C.c_void_p), ("in_kf1", C.c_void_p),
                ("world_pos1", C.c_void_p), ("normal1", C.c_void_p), ("mp_desc1", C.c_void_p), ("min_dist1", C.c_void_p),
                ("max_dist1", C.c_void_p), ("n2", C.c_int), ("kp2_xy", C.c_void_p), ("kp2_octave", C.c_void_p),
                ("uright2", C.c_void_p), ("desc2", C.c_void_p), ("grid", C.c_float * 6), ("Tcw_q", C.c_float * 4),
                ("Tcw_t", C.c_float * 3), ("Ow", C.c_float * 3), ("K", C.c_float * 4), ("bf", C.c_float),
                ("scale_factors", C.c_void_p), ("inv_level_sigma2", C.c_void_p), ("n_levels", C.c_int),
                ("log_scale_factor", C.c_float), ("th", C.c_float)]


def make_fuse_input(case, th, keep):
    """case: dict from tests/parity_checks.make_fuse_case."""
    def arr(v, dt):
        a = np.ascontiguousarray(v, dt)
        keep.append(a)
        return a.ctypes.data
    P = FuseInput()
    P.n1 = len(case["has_mp1"])
    P.has_mp1, P.bad1, P.in_kf1 = arr(case["has_mp1"], np.uint8), arr(case["bad1"], np.uint8), arr(case["in_kf1"], np.uint8)
    P.world_pos1, P.normal1 = arr(case["world_pos1"], np.float32), arr(case["normal1"], np.float32)
    P.mp_desc1 = arr(case["mp_desc1"], np.uint8)
    P.min_dist1, P.max_dist1 = arr(case["min_dist1"], np.float32), arr(case["max_dist1"], np.float32)
    P.n2 = len(case["kp2_xy"])
    P.kp2_xy, P.kp2_octave = arr(case["kp2_xy"], np.float32), arr(case["kp2_octave"], np.int32)
    P.uright2, P.desc2 = arr(case["uright2"], np.float32), arr(case["desc2"], np.uint8)
    for name, n in (("grid", 6), ("Tcw_q", 4), ("Tcw_t", 3), ("Ow", 3), ("K", 4)):
        for i in range(n):
            getattr(P, name)[i] = float(case[name][i])
    P.bf = float(case["bf"])
    P.scale_factors, P.inv_level_sigma2 = arr(case["scale_factors"], np.float32), arr(case["inv_level_sigma2"], np.float32)
    P.n_levels = len(case["scale_factors"])
    P.log_scale_factor = float(case["log_scale_factor"])
    P.th = float(th)
    return P


def fuse_search(case, th=3.0):
    keep = []
    P = make_fuse_input(case, th, keep)
    best = np.zeros(P.n1, np.int32)
    n = lib().orc_fuse_search(C.byref(P), _p(best))
    return best, n


def fuse_prepass(case):
    keep = []
    P = make_fuse_input(case, 3.0, keep)
    valid = np.zeros(P.n1, np.uint8)
    level = np.zeros(P.n1, np.int32)
    lib().orc_fuse_prepass(C.byref(P), _p(valid), _p(level))
    return valid, level


class LocalPointsInput(C.Structure):
    """orc_local_points_input == rgbl_local_points_input (same layout)."""
    _fields_ = [("n1", C.c_int), ("valid1", C.c_void_p), ("proj1", C.c_void_p), ("level1", C.c_void_p),
                ("view_cos1", C.c_void_p), ("mp_desc1", C.c_void_p), ("mp_observed1", C.c_void_p),
                ("n2", C.c_int), ("kp2_xy", C.c_void_p), ("kp2_octave", C.c_void_p), ("uright2", C.c_void_p),
                ("desc2", C.c_void_p), ("blocked2", C.c_void_p), ("grid", C.c_float * 6), ("scale_factors", C.c_void_p),
                ("n_levels", C.c_int), ("th", C.c_float), ("nnratio", C.c_float)]


def make_local_points_input(case, th, nnratio, keep):
    def arr(v, dt):
        a = np.ascontiguousarray(v, dt)
        keep.append(a)
        return a.ctypes.data
    P = LocalPointsInput()
    P.n1 = len(case["valid1"])
    P.valid1, P.proj1, P.level1 = arr(case["valid1"], np.uint8), arr(case["proj1"], np.float32), arr(case["level1"], np.int32)
    P.view_cos1, P.mp_desc1 = arr(case["view_cos1"], np.float32), arr(case["mp_desc1"], np.uint8)
    P.mp_observed1 = arr(case["mp_observed1"], np.uint8)
    P.n2 = len(case["kp2_xy"])
    P.kp2_xy, P.kp2_octave = arr(case["kp2_xy"], np.float32), arr(case["kp2_octave"], np.int32)
    P.uright2, P.desc2, P.blocked2 = arr(case["uright2"], np.float32), arr(case["desc2"], np.uint8), arr(case["blocked2"], np.uint8)
    for i in range(6):
        P.grid[i] = float(case["grid"][i])
    P.scale_factors = arr(case["scale_factors"], np.float32)
    P.n_levels = len(case["scale_factors"])
    P.th, P.nnratio = float(th), float(nnratio)
    return P


def search_local_points(case, th=1.0, nnratio=0.8):
    keep = []
    P = make_local_points_input(case, th, nnratio, keep)
    m = np.zeros(P.n2, np.int32)
    n = lib().orc_search_local_points(C.byref(P), _p(m))
    return m, n


class InitializationInput(C.Structure):
    """orc_initialization_input == rgbl_initialization_input (same layout)."""
    _fields_ = [("n1", C.c_int), ("kp1_octave", C.c_void_p), ("kp1_angle", C.c_void_p), ("desc1", C.c_void_p),
                ("n2", C.c_int), ("kp2_xy", C.c_void_p), ("kp2_octave", C.c_void_p), ("kp2_angle", C.c_void_p),
                ("desc2", C.c_void_p), ("grid", C.c_float * 6), ("window_size", C.c_int), ("nnratio", C.c_float),
                ("check_orientation", C.c_int)]


def make_initialization_input(case, window, nnratio, check_orientation, keep, cls=InitializationInput):
    def arr(v, dt):
        a = np.ascontiguousarray(v, dt)
        keep.append(a)
        return a.ctypes.data
    P = cls()
    P.n1 = len(case["kp1_octave"])
    P.kp1_octave, P.kp1_angle, P.desc1 = arr(case["kp1_octave"], np.int32), arr(case["kp1_angle"], np.float32), arr(case["desc1"], np.uint8)
    P.n2 = len(case["kp2_xy"])
    P.kp2_xy, P.kp2_octave = arr(case["kp2_xy"], np.float32), arr(case["kp2_octave"], np.int32)
    P.kp2_angle, P.desc2 = arr(case["kp2_angle"], np.float32), arr(case["desc2"], np.uint8)
    for i in range(6):
        P.grid[i] = float(case["grid"][i])
    P.window_size, P.nnratio, P.check_orientation = int(window), float(nnratio), int(check_orientation)
    return P


def search_for_initialization(case, window=100, nnratio=0.9, check_orientation=True):
    """Returns (vnMatches12, updated vbPrevMatched, nmatches)."""
    keep = []
    P = make_initialization_input(case, window, nnratio, check_orientation, keep)
    prev = np.ascontiguousarray(case["prev_matched"], np.float32).copy()
    m = np.zeros(P.n1, np.int32)
    n = lib().orc_search_for_initialization(C.byref(P), _p(prev), _p(m))
    return m, prev, n


class Vocabulary(C.Structure):
    _fields_ = [("n_nodes", C.c_int), ("L", C.c_int), ("child_off", C.c_void_p), ("child", C.c_void_p), ("desc", C.c_void_p),
                ("weight", C.c_void_p), ("word_id", C.c_void_p)]


def bow_from_descent(word, weight, node):
    """BowVector / FeatureVector of TemplatedVocabulary::transform(features, v, fv, levelsup) (TemplatedVocabulary.h:1127-1192,
    TF_IDF + L1_NORM as in ORBvoc.txt) from the per-feature descent: weights of a word accumulate in feature order
    (BowVector::addWeight), the L1 norm is summed in ascending word order (BowVector::normalize), features whose word is
    stopped (weight 0) are dropped, FeatureVector keeps ascending feature indices per node."""
    bow, fv = {}, {}
    for i, (w, wt, nd) in enumerate(zip(word, weight, node)):
        if wt > 0:
            bow[int(w)] = bow.get(int(w), 0.0) + float(wt)
            fv.setdefault(int(nd), []).append(i)
    ids = sorted(bow)
    vals = np.array([bow[k] for k in ids], np.float64)
    norm = 0.0
    for v in vals:
        norm += abs(float(v))
    if norm > 0.0:
        vals = vals / norm
    nids = sorted(fv)
    off = np.zeros(len(nids) + 1, np.int32)
    for j, k in enumerate(nids):
        off[j + 1] = off[j] + len(fv[k])
    feats = np.array([i for k in nids for i in fv[k]], np.uint32)
    return np.array(ids, np.uint32), vals, np.array(nids, np.uint32), off, feats


def bow_transform(varr, desc, levelsup=4):
    """varr: synth.vocabulary_arrays(...). Returns (word ids, values, node ids, node offsets, feature indices)."""
    desc = np.ascontiguousarray(desc, np.uint8)
    n = len(desc)
    keep = [np.ascontiguousarray(varr[k]) for k in ("child_off", "child", "desc", "weight", "word_id")]
    V = Vocabulary(varr["n_nodes"], varr["L"], *[a.ctypes.data for a in keep])
    word, weight, node = np.zeros(n, np.int32), np.zeros(n, np.float64), np.zeros(n, np.int32)
    lib().orc_bow_descend(C.byref(V), _p(desc), n, levelsup, _p(word), _p(weight), _p(node))
    return bow_from_descent(word, weight, node)


class TriInput(C.Structure):
    _fields_ = [("n1", C.c_int), ("n2", C.c_int),
                ("desc1", C.c_void_p), ("desc2", C.c_void_p),
                ("kp1_xy", C.c_void_p), ("kp2_xy", C.c_void_p),
                ("kp1_octave", C.c_void_p), ("kp2_octave", C.c_void_p),
                ("kp1_angle", C.c_void_p), ("kp2_angle", C.c_void_p),
                ("uright1", C.c_void_p), ("uright2", C.c_void_p),
                ("has_mp1", C.c_void_p), ("has_mp2", C.c_void_p),
                ("nnodes1", C.c_int), ("nnodes2", C.c_int),
                ("node_id1", C.c_void_p), ("node_off1", C.c_void_p), ("node_feat1", C.c_void_p),
                ("node_id2", C.c_void_p), ("node_off2", C.c_void_p), ("node_feat2", C.c_void_p),
                ("F12", C.c_float * 9), ("ep", C.c_float * 2),
                ("scale_factors2", C.c_void_p), ("level_sigma2_2", C.c_void_p),
                ("only_stereo", C.c_int), ("coarse", C.c_int), ("check_orientation", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_extractor_create.restype = C.c_void_p
        L.orc_extractor_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.orc_extractor_destroy.argtypes = [C.c_void_p]
        L.orc_extractor_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.orc_extract.restype = C.c_int
        L.orc_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.orc_level_size.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        for f in ("orc_level_image", "orc_level_blurred", "orc_level_bordered"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
            getattr(L, f).restype = None
        for f in ("orc_level_candidates", "orc_level_keypoints"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
            getattr(L, f).restype = C.c_int
        L.orc_stereo_matches.restype = None
        L.orc_stereo_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.orc_resize_linear_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                           C.c_int, C.c_int]
        L.orc_gaussian_blur7_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_cvt_gray.restype = None
        L.orc_cvt_gray.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_kitti_bin_to_cloud.restype = None
        L.orc_kitti_bin_to_cloud.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_bow_descend.restype = None
        L.orc_bow_descend.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_search_by_bow.restype = C.c_int
        L.orc_search_by_bow.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_void_p]
        L.orc_search_by_bow_rig.restype = C.c_int
        L.orc_search_by_bow_rig.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
        L.orc_search_for_initialization.restype = C.c_int
        L.orc_search_for_initialization.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_search_local_points.restype = C.c_int
        L.orc_search_local_points.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_search_by_projection.restype = C.c_int
        L.orc_search_by_projection.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_fast.restype = C.c_int
        L.orc_fast.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_fast_corner_score.restype = C.c_int
        L.orc_fast_corner_score.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_fast_atan2.restype = C.c_float
        L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orc_cv_round_f.restype = C.c_int
        L.orc_cv_round_f.argtypes = [C.c_float]
        L.orc_ic_angle.restype = C.c_float
        L.orc_ic_angle.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_brief.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]
        L.orc_distribute_octree.restype = C.c_int
        L.orc_distribute_octree.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_int, C.c_void_p, C.c_int]
        L.orc_depth.restype = C.c_int
        L.orc_depth.argtypes = [C.POINTER(DepthParams), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p]
        L.orc_projection_matrix.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_structuring_element.restype = C.c_int
        L.orc_structuring_element.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_descriptor_distance.restype = C.c_int
        L.orc_descriptor_distance.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_hamming_bf.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_void_p]
        L.orc_search_triangulation.restype = C.c_int
        L.orc_search_triangulation.argtypes = [C.POINTER(TriInput), C.c_void_p]
        L.orc_fundamental.argtypes = [C.c_void_p] * 5
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Extractor:
    """Mirror of ORB_SLAM3::ORBextractor on the oracle."""

    def __init__(self, nfeatures=2000, scale_factor=1.2, nlevels=8, ini_th=12, min_th=7):
        self.L = lib()
        self.nlevels = nlevels
        self.nfeatures = nfeatures
        self.h = self.L.orc_extractor_create(nfeatures, scale_factor, nlevels, ini_th, min_th)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_extractor_destroy(self.h)
            self.h = None

    def tables(self):
        n = self.nlevels
        sc, isc, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        per = np.zeros(n, np.int32)
        umax = np.zeros(16, np.int32)
        self.L.orc_extractor_tables(self.h, _p(sc), _p(isc), _p(s2), _p(is2), _p(per), _p(umax))
        return dict(scale=sc, inv_scale=isc, sigma2=s2, inv_sigma2=is2, per_level=per, umax=umax)

    def __call__(self, img, lapping=(0, 0), cap=None):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        cap = cap or (self.nfeatures * 2 + 4096)
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        mono = self.L.orc_extract(self.h, _p(img), w, h, img.strides[0], lapping[0], lapping[1],
                                  _p(kps), _p(desc), cap, C.byref(n))
        if mono == -2:
            return self(img, lapping, cap=n.value)
        return kps[:n.value].copy(), desc[:n.value].copy(), mono

    def level_size(self, l):
        w, h = C.c_int(), C.c_int()
        self.L.orc_level_size(self.h, l, C.byref(w), C.byref(h))
        return w.value, h.value

    def _level_img(self, fn, l, border=0):
        w, h = self.level_size(l)
        out = np.zeros((h + 2 * border, w + 2 * border), np.uint8)
        fn(self.h, l, _p(out), out.strides[0])
        return out

    def level_image(self, l):
        return self._level_img(self.L.orc_level_image, l)

    def level_blurred(self, l):
        return self._level_img(self.L.orc_level_blurred, l)

    def level_bordered(self, l):
        return self._level_img(self.L.orc_level_bordered, l, 19)

    def _kps(self, fn, l):
        n = fn(self.h, l, None, 0)
        out = np.zeros(max(n, 1), KP_DTYPE)
        fn(self.h, l, _p(out), n)
        return out[:n]

    def level_candidates(self, l):
        return self._kps(self.L.orc_level_candidates, l)

    def level_keypoints(self, l):
        return self._kps(self.L.orc_level_keypoints, l)


def stereo_matches(ex_left, ex_right, kpl, dl, kpr, dr, mb, mbf):
    """Frame::ComputeStereoMatches on two Extractor objects that just processed the left / right image."""
    kpl = np.ascontiguousarray(kpl, KP_DTYPE)
    kpr = np.ascontiguousarray(kpr, KP_DTYPE)
    dl = np.ascontiguousarray(dl, np.uint8)
    dr = np.ascontiguousarray(dr, np.uint8)
    ur = np.zeros(len(kpl), np.float32)
    dp = np.zeros(len(kpl), np.float32)
    lib().orc_stereo_matches(ex_left.h, ex_right.h, _p(kpl), _p(dl), len(kpl), _p(kpr), _p(dr), len(kpr), float(mb), float(mbf),
                             _p(ur), _p(dp))
    return ur, dp


def stereo_fisheye_matches(desc_left, mono_left, desc_right, mono_right):
    a = np.ascontiguousarray(desc_left, np.uint8).reshape(-1, 32)
    b = np.ascontiguousarray(desc_right, np.uint8).reshape(-1, 32)
    l2r, bd, sd = (np.zeros(len(a), np.int32) for _ in range(3))
    f = lib().orc_stereo_fisheye_matches
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    f(_p(a), len(a), int(mono_left), _p(b), len(b), int(mono_right), _p(l2r), _p(bd), _p(sd))
    return l2r, bd, sd


def resize_linear(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().orc_resize_linear_u8(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dw, dh, dw)
    return dst


def gaussian_blur7(src):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros_like(src)
    lib().orc_gaussian_blur7_u8(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst),
                                dst.strides[0])
    return dst


def fast(img, threshold, nonmax=True):
    img = np.ascontiguousarray(img, np.uint8)
    cap = img.size
    out = np.zeros(cap, KP_DTYPE)
    n = lib().orc_fast(_p(img), img.shape[1], img.shape[0], img.strides[0], threshold, int(nonmax),
                       _p(out), cap)
    return out[:n]


def corner_score(img, x, y, threshold):
    img = np.ascontiguousarray(img, np.uint8)
    return lib().orc_fast_corner_score(_p(img), img.strides[0], x, y, threshold)


def fast_atan2(y, x):
    return lib().orc_fast_atan2(float(y), float(x))


def cv_round(v):
    return lib().orc_cv_round_f(float(v))


def ic_angle(img, x, y):
    img = np.ascontiguousarray(img, np.uint8)
    return lib().orc_ic_angle(_p(img), img.strides[0], x, y)


def brief(blurred, x, y, angle_deg):
    blurred = np.ascontiguousarray(blurred, np.uint8)
    d = np.zeros(32, np.uint8)
    lib().orc_brief(_p(blurred), blurred.strides[0], x, y, float(angle_deg), _p(d))
    return d


def distribute_octree(cand, min_x, max_x, min_y, max_y, n_features):
    cand = np.ascontiguousarray(cand, KP_DTYPE)
    out = np.zeros(max(len(cand), 1), KP_DTYPE)
    n = lib().orc_distribute_octree(_p(cand), len(cand), min_x, max_x, min_y, max_y, n_features,
                                    _p(out), len(out))
    return out[:n]


def cvt_gray(img, mbRGB):
    """cv::cvtColor(COLOR_RGB(A)2GRAY if mbRGB else COLOR_BGR(A)2GRAY) on an H x W x {3,4} u8 image."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w, ch = img.shape
    out = np.zeros((h, w), np.uint8)
    lib().orc_cvt_gray(_p(img), ch, 0 if mbRGB else 1, w, h, img.strides[0], _p(out), out.strides[0])
    return out


def kitti_bin_to_cloud(xyzi):
    pts = np.ascontiguousarray(xyzi, np.float32).reshape(-1, 4)
    out = np.zeros((4, pts.shape[0]), np.float32)
    lib().orc_kitti_bin_to_cloud(_p(pts), pts.shape[0], _p(out))
    return out


def structuring_element(shape, kw, kh):
    out = np.zeros(kw * kh, np.uint8)
    rc = lib().orc_structuring_element(shape, kw, kh, _p(out))
    if rc != 0:
        raise ValueError("bad structuring element")
    return out.reshape(kh, kw)


def projection_matrix(K3x4, Tr4x4):
    K = np.ascontiguousarray(K3x4, np.float32)
    T = np.ascontiguousarray(Tr4x4, np.float32)
    out = np.zeros((3, 4), np.float32)
    lib().orc_projection_matrix(_p(K), _p(T), _p(out))
    return out


def make_depth_params(proj, min_dist=5.0, max_dist=200.0, mbf=100.0, method=3, kernel=None,
                      avg_ksize=5, nn_radius=7.0):
    P = DepthParams()
    proj = np.asarray(proj, np.float32).reshape(12)
    for i in range(12):
        P.proj[i] = float(proj[i])
    P.min_dist, P.max_dist, P.mbf, P.method = min_dist, max_dist, mbf, method
    if kernel is None:
        kernel = structuring_element(3, 5, 5)
    kernel = np.asarray(kernel, np.uint8)
    P.kh, P.kw = kernel.shape
    flat = kernel.reshape(-1)
    for i in range(flat.size):
        P.kernel[i] = int(flat[i])
    P.avg_ksize = avg_ksize
    P.nn_radius = nn_radius
    return P


def depth(P, cloud4xn, w, h, kp_xy, kpun_x, want_maps=True):
    cloud = np.ascontiguousarray(cloud4xn, np.float32)
    n = cloud.shape[1]
    kp_xy = np.ascontiguousarray(kp_xy, np.float32).reshape(-1, 2)
    kpun_x = np.ascontiguousarray(kpun_x, np.float32)
    k = kp_xy.shape[0]
    d = np.zeros(k, np.float32)
    ur = np.zeros(k, np.float32)
    raw = np.zeros((h, w), np.float32) if want_maps else None
    proc = np.zeros((h, w), np.float32) if want_maps else None
    rc = lib().orc_depth(C.byref(P), _p(cloud), n, cloud.strides[0] // 4, w, h, _p(kp_xy), _p(kpun_x), k,
                         _p(d), _p(ur), _p(raw), _p(proc))
    if rc != 0:
        raise RuntimeError("orc_depth failed")
    return d, ur, raw, proc


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib().orc_descriptor_distance(_p(a), _p(b))


def hamming_bf(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    na, nb = len(a), len(b)
    bi = np.zeros(na, np.int32)
    bd = np.zeros(na, np.int32)
    sd = np.zeros(na, np.int32)
    lib().orc_hamming_bf(_p(a), na, _p(b), nb, _p(bi), _p(bd), _p(sd))
    return bi, bd, sd


def fundamental(K1, K2, R12, t12):
    a = [np.ascontiguousarray(v, np.float32) for v in (K1, K2, R12, t12)]
    F = np.zeros(9, np.float32)
    lib().orc_fundamental(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(F))
    return F


def search_by_bow(kf, frame, nnratio=0.7, check_orientation=True, keyframes=False, n_left=-1):
    """ORBmatcher::SearchByBoW(pKF, F, vpMapPointMatches).  kf / frame: dicts as for search_triangulation (kf["has_mp"] = map
    point present and not bad).  n_left = F.Nleft (-1: a single camera; else the two-camera branches, ORBmatcher.cc:298-386).
    Returns (match per frame feature: key-frame feature index or -1, nmatches)."""
    keep = []

    def arr(v, dt):
        a = np.ascontiguousarray(v, dt)
        keep.append(a)
        return a.ctypes.data
    T = TriInput()
    T.n1, T.n2 = len(kf["desc"]), len(frame["desc"])
    for i, d in ((1, kf), (2, frame)):
        setattr(T, "desc%d" % i, arr(d["desc"], np.uint8))
        setattr(T, "kp%d_xy" % i, arr(d["xy"], np.float32))
        setattr(T, "kp%d_octave" % i, arr(d["octave"], np.int32))
        setattr(T, "kp%d_angle" % i, arr(d["angle"], np.float32))
        setattr(T, "uright%d" % i, arr(d["uright"], np.float32))
        setattr(T, "has_mp%d" % i, arr(d["has_mp"], np.uint8))
        setattr(T, "nnodes%d" % i, len(d["node_id"]))
        setattr(T, "node_id%d" % i, arr(d["node_id"], np.int32))
        setattr(T, "node_off%d" % i, arr(d["node_off"], np.int32))
        setattr(T, "node_feat%d" % i, arr(d["node_feat"], np.int32))
    if keyframes:
        m = np.zeros(T.n1, np.int32)
        n = lib().orc_search_by_bow_kf(C.byref(T), C.c_float(nnratio), int(check_orientation), _p(m))
        return m, n
    m = np.zeros(T.n2, np.int32)
    if n_left >= 0:
        n = lib().orc_search_by_bow_rig(C.byref(T), int(n_left), C.c_float(nnratio), int(check_orientation), _p(m))
        return m, n
    n = lib().orc_search_by_bow(C.byref(T), C.c_float(nnratio), int(check_orientation), _p(m))
    return m, n


def search_by_bow_kf(kf1, kf2, nnratio=0.75, check_orientation=True):
    """ORBmatcher::SearchByBoW(pKF1, pKF2, vpMatches12).  Returns (match per kf1 feature: kf2 feature index or -1, nmatches)."""
    return search_by_bow(kf1, kf2, nnratio, check_orientation, keyframes=True)


def search_triangulation(kf1, kf2, F12, ep, scale_factors2, level_sigma2_2, only_stereo=False,
                         coarse=False, check_orientation=False):
    """kf = dict(desc, xy, octave, angle, uright, has_mp, node_id, node_off, node_feat)."""
    keep = []

    def arr(v, dt):
        a = np.ascontiguousarray(v, dt)
        keep.append(a)
        return a.ctypes.data

    T = TriInput()
    T.n1, T.n2 = len(kf1["desc"]), len(kf2["desc"])
    for i, kf in ((1, kf1), (2, kf2)):
        setattr(T, "desc%d" % i, arr(kf["desc"], np.uint8))
        setattr(T, "kp%d_xy" % i, arr(kf["xy"], np.float32))
        setattr(T, "kp%d_octave" % i, arr(kf["octave"], np.int32))
        setattr(T, "kp%d_angle" % i, arr(kf["angle"], np.float32))
        setattr(T, "uright%d" % i, arr(kf["uright"], np.float32))
        setattr(T, "has_mp%d" % i, arr(kf["has_mp"], np.uint8))
        setattr(T, "nnodes%d" % i, len(kf["node_id"]))
        setattr(T, "node_id%d" % i, arr(kf["node_id"], np.int32))
        setattr(T, "node_off%d" % i, arr(kf["node_off"], np.int32))
        setattr(T, "node_feat%d" % i, arr(kf["node_feat"], np.int32))
    for i in range(9):
        T.F12[i] = float(F12[i])
    T.ep[0], T.ep[1] = float(ep[0]), float(ep[1])
    T.scale_factors2 = arr(scale_factors2, np.float32)
    T.level_sigma2_2 = arr(level_sigma2_2, np.float32)
    T.only_stereo, T.coarse, T.check_orientation = int(only_stereo), int(coarse), int(check_orientation)
    m = np.zeros(T.n1, np.int32)
    n = lib().orc_search_triangulation(C.byref(T), _p(m))
    return m, n


def undistort_points(xy, K, dist):
    """cv::undistortPoints(xy, K, dist, R = I, P = K) as Frame::UndistortKeyPoints calls it."""
    a = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    k = np.ascontiguousarray(K, np.float32)
    d = np.ascontiguousarray(dist, np.float32)
    out = np.zeros_like(a)
    lib().orc_undistort_points(_p(a), len(a), _p(k), _p(d), len(d), _p(out))
    return out


def distinctive_descriptors(descriptor_lists):
    """MapPoint::ComputeDistinctiveDescriptors for a batch: BestIdx per list (-1 when empty)."""
    off = np.zeros(len(descriptor_lists) + 1, np.int32)
    for i, d in enumerate(descriptor_lists):
        off[i + 1] = off[i] + len(d)
    desc = (np.concatenate([np.ascontiguousarray(d, np.uint8).reshape(-1, 32) for d in descriptor_lists])
            if off[-1] else np.zeros((1, 32), np.uint8))
    best = np.zeros(len(descriptor_lists), np.int32)
    lib().orc_distinctive_descriptors(_p(desc), _p(off), len(descriptor_lists), _p(best))
    return best


class ProjectSearchInput(C.Structure):
    _fields_ = [("n1", C.c_int), ("valid1", C.c_void_p), ("cam_pos1", C.c_void_p), ("mp_desc1", C.c_void_p),
                ("level1", C.c_void_p), ("n2", C.c_int), ("kp2_xy", C.c_void_p), ("kp2_octave", C.c_void_p),
                ("desc2", C.c_void_p), ("grid", C.c_float * 6), ("K", C.c_float * 4), ("scale_factors", C.c_void_p),
                ("n_levels", C.c_int), ("th", C.c_float), ("proj_form", C.c_int), ("max_dist", C.c_int)]


def project_search(case, th, proj_form, max_dist, matched2=None):
    """The per-point search of Fuse(pKF, Scw, ...) / SearchBySim3 on camera-frame points; returns (best_idx, best_dist)."""
    keep = []

    def arr(v, dt):
        a = np.ascontiguousarray(v, dt)
        keep.append(a)
        return a.ctypes.data
    P = ProjectSearchInput()
    P.n1 = len(case["valid1"])
    P.valid1, P.cam_pos1 = arr(case["valid1"], np.uint8), arr(case["cam_pos1"], np.float32)
    P.mp_desc1, P.level1 = arr(case["mp_desc1"], np.uint8), arr(case["level1"], np.int32)
    P.n2 = len(case["kp2_xy"])
    P.kp2_xy, P.kp2_octave, P.desc2 = arr(case["kp2_xy"], np.float32), arr(case["kp2_octave"], np.int32), arr(case["desc2"], np.uint8)
    for name, n in (("grid", 6), ("K", 4)):
        for i in range(n):
            getattr(P, name)[i] = float(case[name][i])
    P.scale_factors = arr(case["scale_factors"], np.float32)
    P.n_levels = len(case["scale_factors"])
    P.th, P.proj_form, P.max_dist = float(th), int(proj_form), int(max_dist)
    if matched2 is not None:
        m2 = np.ascontiguousarray(matched2, np.uint8)
        match = np.zeros(P.n2, np.int32)
        n = lib().orc_search_by_projection_sim3(C.byref(P), _p(m2), _p(match))
        return match, n
    best = np.zeros(P.n1, np.int32)
    dist = np.zeros(P.n1, np.int32)
    lib().orc_project_search(C.byref(P), _p(best), _p(dist))
    return best, dist


def search_by_projection_sim3(case, matched2, th, proj_form, max_dist):
    """The greedy search of SearchByProjection(pKF, Scw, ...) on camera-frame points; returns (match2, nmatches)."""
    return project_search(case, th, proj_form, max_dist, matched2=matched2)
