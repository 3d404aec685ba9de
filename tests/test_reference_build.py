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
from yaml_cases import PARSE_CASES, yaml_with

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_orbextractor.so")
BUILDS = ["portable", "native"]   # -O2 -ffp-contract=off | the reference's own -O3 -march=native (CMakeLists.txt:10-13)


def ref_build(path, build):
    """Path of the portable or the native build of a reference library; skips when it is not there or (native) was built on
    another CPU (-march=native code must not run elsewhere)."""
    if os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    if build == "native":
        path = path.replace(".so", "_native.so")
        stamp = os.path.join(ROOT, "oracle", "_ref", "native_host.txt")
        if not (os.path.exists(path) and os.path.exists(stamp)):
            pytest.skip("oracle/_ref native build not there (needs /root/reference)")
        here = subprocess.run("g++ -march=native -Q --help=target | grep -E -- '-march=|-mtune=' | tr -s ' \\t' ' '", shell=True,
                              capture_output=True, text=True).stdout
        if here != open(stamp).read():
            pytest.skip("oracle/_ref native build was made on another CPU")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    return path


@pytest.fixture(scope="module", params=BUILDS)
def ref(oracle, request):
    lib = C.CDLL(ref_build(REF_SO, request.param))
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


# ---------------------------------------------------------------------------------------------------------------
# The reference's own src/DepthModule.cc (oracle/_ref/libref_depthmodule.so) versus the restated oracle: YAML
# parsing and its failure modes, K * Tr, projection loop, the three up-sampling chains, keypoint depth / uRight.
REF_DEPTH_SO = os.path.join(ROOT, "oracle", "_ref", "libref_depthmodule.so")
_F = C.c_float


@pytest.fixture(scope="module", params=BUILDS)
def refdepth(oracle, request):
    lib = C.CDLL(ref_build(REF_DEPTH_SO, request.param))
    lib.ref_depth_create.restype = C.c_void_p
    lib.ref_depth_create.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p]
    lib.ref_depth_destroy.argtypes = [C.c_void_p]
    lib.ref_depth_compute.restype = C.c_int
    lib.ref_depth_compute.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return lib


class RefDepth:
    def __init__(self, lib, path):
        self.lib = lib
        a, b = C.c_int(-1), C.c_int(-1)
        self.proj = np.zeros((3, 4), np.float32)
        self.h = lib.ref_depth_create(path.encode(), 5, C.byref(a), C.byref(b), self.proj.ctypes.data)
        self.parsed = (bool(a.value), bool(b.value))

    def __call__(self, cloud, w, h, kp_xy, kpun_x):
        cloud = np.ascontiguousarray(cloud, np.float32)
        k = len(kpun_x)
        d, ur = np.zeros(k, np.float32), np.zeros(k, np.float32)
        raw, proc = np.zeros((h, w), np.float32), np.zeros((h, w), np.float32)
        rc = self.lib.ref_depth_compute(self.h, cloud.ctypes.data, cloud.shape[1], cloud.strides[0] // 4, w, h,
                                        kp_xy.ctypes.data, kpun_x.ctypes.data, k, d.ctypes.data, ur.ctypes.data,
                                        raw.ctypes.data, proc.ctypes.data)
        return rc, d, ur, raw, proc

    def close(self):
        self.lib.ref_depth_destroy(self.h)


def keypoints_for(w, h, k, seed):
    rng = np.random.default_rng(seed)
    xy = np.stack([rng.uniform(0, w - 1, k), rng.uniform(0, h - 1, k)], 1).astype(np.float32)
    xy[: k // 2] = np.floor(xy[: k // 2])          # level-0 keypoints have integer coordinates
    xy[0], xy[1], xy[2] = (0, 0), (w - 1, h - 1), (w - 1, 0)
    un = (xy[:, 0] + rng.uniform(-0.5, 0.5, k)).astype(np.float32)
    return np.ascontiguousarray(xy), un


bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)

METHOD_ID = {"NearestNeighborPixel": 1, "AverageFiltering": 2, "InverseDilation": 3}
SHAPE_ID = {"Rectangle": 0, "Cross": 1, "Ellipse": 2, "Diamond": 3}


@pytest.mark.parametrize("method,ktype,ku,kv,extra", [
    ("InverseDilation", "Diamond", 5, 7, {}),                 # the KITTI default (v is ignored for diamonds)
    ("InverseDilation", "Diamond", 9, 9, {}),
    ("InverseDilation", "Rectangle", 3, 5, {}),
    ("InverseDilation", "Ellipse", 7, 5, {}),
    ("InverseDilation", "Cross", 5, 3, {}),
    ("AverageFiltering", "Diamond", 5, 5, {"LiDAR.MethodAverageFiltering.KernelSize": "3.0"}),
    ("AverageFiltering", "Diamond", 5, 5, {"LiDAR.MethodAverageFiltering.KernelSize": "5.0"}),
    ("NearestNeighborPixel", "Diamond", 5, 5, {}),
    ("NearestNeighborPixel", "Diamond", 5, 5, {"LiDAR.MethodNearestNeighborPixel.SearchDistance": "3.0"}),
])
def test_reference_depthmodule_agrees_with_oracle(refdepth, tmp_path, method, ktype, ku, kv, extra):
    w, h = 1241, 376
    edits = {"LiDAR.Method": '"%s"' % method, "LiDAR.MethodInverseDilation.KernelType": '"%s"' % ktype,
             "LiDAR.MethodInverseDilation.KernelSize_u": "%d.0" % ku, "LiDAR.MethodInverseDilation.KernelSize_v": "%d.0" % kv}
    edits.update(extra)
    R = RefDepth(refdepth, yaml_with(tmp_path, "s.yaml", edits))
    assert R.parsed == (True, True)
    oproj = O.projection_matrix(synth.KITTI_K, synth.KITTI_TR)
    assert np.array_equal(bits(R.proj), bits(oproj)), "K * Tr (DepthModule.cc:434)"
    P = O.make_depth_params(oproj, 5.0, 200.0, 100.0, METHOD_ID[method], O.structuring_element(SHAPE_ID[ktype], ku, ku if ktype == "Diamond" else kv),
                            avg_ksize=int(float(extra.get("LiDAR.MethodAverageFiltering.KernelSize", "5.0"))),
                            nn_radius=float(extra.get("LiDAR.MethodNearestNeighborPixel.SearchDistance", "7.0")))
    for seed in (0, 1):
        cloud = synth.lidar_scan(seed)
        xy, un = keypoints_for(w, h, 1500, seed)
        rc, d, ur, raw, proc = R(cloud, w, h, xy, un)
        od, our, oraw, oproc = O.depth(P, cloud, w, h, xy, un)
        assert np.array_equal(bits(raw), bits(oraw)), "RawDepthMap"
        assert (oraw > 0).sum() > 5000
        if method == "NearestNeighborPixel":
            assert rc in (0, 2)      # this method never writes ProcessedDepthMap
        else:
            assert rc == 0
            assert np.array_equal(bits(proc), bits(oproc)), "ProcessedDepthMap"
        assert np.array_equal(bits(d), bits(od)) and np.array_equal(bits(ur), bits(our))
        assert (d > 0).sum() > 100
    R.close()


def test_reference_depthmodule_second_frame_reuses_buffers(refdepth, tmp_path):
    """`ProcessedDepthMap = S - RawDepthMap` assigns INTO the previous frame's buffer from the second call on."""
    w, h = 620, 188
    R = RefDepth(refdepth, yaml_with(tmp_path, "s.yaml"))
    K = synth.KITTI_K.copy()
    oproj = O.projection_matrix(synth.KITTI_K, synth.KITTI_TR)
    P = O.make_depth_params(oproj, 5.0, 200.0, 100.0, 3, O.structuring_element(3, 5, 5))
    for seed in (3, 4, 5):
        cloud = synth.lidar_scan(seed, n_rings=32, n_az=700)
        xy, un = keypoints_for(w, h, 300, seed)
        rc, d, ur, raw, proc = R(cloud, w, h, xy, un)
        od, our, oraw, oproc = O.depth(P, cloud, w, h, xy, un)
        assert rc == 0 and np.array_equal(bits(proc), bits(oproc)) and np.array_equal(bits(d), bits(od))
    R.close()
    del K


@pytest.mark.parametrize("edits,drop,expect", PARSE_CASES)
def test_reference_depthmodule_parse_failures_match_the_shim_rules(refdepth, tmp_path, edits, drop, expect):
    """The reference's own parser, run on broken settings files; orb_slam3_rgbl_amd/shim/DepthModule.cc encodes the
    same outcomes (tests/test_shim.py exercises that side)."""
    R = RefDepth(refdepth, yaml_with(tmp_path, "bad.yaml", edits, drop))
    lidar_ok, ups_ok = R.parsed
    assert lidar_ok == expect[0]
    if expect[1] is not None:
        assert ups_ok == expect[1]
    if expect[1] is None:
        # the reference runs ParseUpsamplingParameters on an UNINITIALISED SelectedUpsamlingMethod after a failed
        # LiDAR parse (DepthModule.cc:38-39); whatever that yields, the module stays disabled
        pass
    if not (lidar_ok and ups_ok) or edits.get("LiDAR.Method") == '"None"':
        xy, un = keypoints_for(64, 48, 10, 0)
        rc, *_ = R(synth.lidar_scan(0, 4, 50), 64, 48, xy, un)
        assert rc == -1          # disabled module / method None: no keypoint depth is produced
    R.close()


# ---------------------------------------------------------------------------------------------------------------
# The reference's own src/ORBmatcher.cc (oracle/_ref/libref_orbmatcher.so; MapPoint / KeyFrame / Frame / Sophus / DBoW2
# are the plain stand-ins of oracle/cvcompat/orbslam_types.h) versus the restated oracle.
REF_MATCHER_SO = os.path.join(ROOT, "oracle", "_ref", "libref_orbmatcher.so")


class KfArrays(C.Structure):
    _fields_ = [("n", C.c_int), ("desc", C.c_void_p), ("kp_xy", C.c_void_p), ("kp_octave", C.c_void_p), ("kp_angle", C.c_void_p),
                ("uright", C.c_void_p), ("has_mp", C.c_void_p), ("nnodes", C.c_int), ("node_id", C.c_void_p),
                ("node_off", C.c_void_p), ("node_feat", C.c_void_p)]


@pytest.fixture(scope="module")
def refmatcher(oracle):
    if os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    if not os.path.exists(REF_MATCHER_SO):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    lib = C.CDLL(REF_MATCHER_SO)
    lib.ref_descriptor_distance.restype = C.c_int
    lib.ref_descriptor_distance.argtypes = [C.c_void_p, C.c_void_p]
    lib.ref_search_triangulation.restype = C.c_int
    lib.ref_search_triangulation.argtypes = [C.POINTER(KfArrays), C.POINTER(KfArrays)] + [C.c_void_p] * 3 + [C.c_int] + \
        [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_void_p] * 4
    return lib


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


def test_reference_descriptor_distance(refmatcher):
    from orb_slam3_rgbl_amd import synth as S
    a = S.descriptors(300, 5)
    b, _ = S.perturbed_descriptors(a, flip_p=0.2, seed=6)
    for i in range(300):
        assert refmatcher.ref_descriptor_distance(a[i].ctypes.data, b[i].ctypes.data) == O.descriptor_distance(a[i], b[i])
    z, f = np.zeros(32, np.uint8), np.full(32, 255, np.uint8)
    assert refmatcher.ref_descriptor_distance(z.ctypes.data, f.ctypes.data) == 256
    assert refmatcher.ref_descriptor_distance(f.ctypes.data, f.ctypes.data) == 0


@pytest.mark.parametrize("seed,only_stereo,coarse,check_ori", [
    (11, False, False, False),    # LocalMapping::CreateNewMapPoints calls it like this (ORBmatcher(0.6, false))
    (12, False, True, False),     # bCoarse: no epipolar test
    (13, True, False, False),     # bOnlyStereo
    (14, False, False, True),     # orientation histogram on (dead code in the reference's callers, alive in the function)
    (15, True, True, True),
])
def test_reference_search_for_triangulation_agrees_with_oracle(refmatcher, seed, only_stereo, coarse, check_ori):
    import parity_checks as pc
    kf1, kf2, Kc, _, _, _, sf, s2 = pc.make_triangulation_case(1200, seed=seed)
    keep = []
    a1, a2 = kf_arrays(kf1, keep), kf_arrays(kf2, keep)
    # camera 1 at the origin; camera 2 rotated a little and placed so that camera 1 projects into image 2 (epipole guard active)
    ang = 0.03
    q1, t1 = np.array([0, 0, 0, 1], np.float32), np.zeros(3, np.float32)
    q2 = np.array([0, np.sin(ang / 2), 0, np.cos(ang / 2)], np.float32)
    t2 = np.array([-0.3, 0.01, 1.0], np.float32)
    m = np.zeros(a1.n, np.int32)
    R12, t12, ep = np.zeros(9, np.float32), np.zeros(3, np.float32), np.zeros(2, np.float32)
    nm = refmatcher.ref_search_triangulation(C.byref(a1), C.byref(a2), Kc.ctypes.data, sf.ctypes.data, s2.ctypes.data, 8,
                                             q1.ctypes.data, t1.ctypes.data, q2.ctypes.data, t2.ctypes.data, int(only_stereo),
                                             int(coarse), int(check_ori), m.ctypes.data, R12.ctypes.data, t12.ctypes.data,
                                             ep.ctypes.data)
    assert 0 < ep[0] < 1241 and 0 < ep[1] < 376
    F = O.fundamental(Kc, Kc, R12, t12)
    om, onm = O.search_triangulation(kf1, kf2, F, ep, sf, s2, only_stereo, coarse, check_ori)
    assert nm == onm and np.array_equal(m, om)
    assert nm > 5


@pytest.mark.parametrize("seed,motion,th,mono,check_ori", [
    (21, "forward", 7.0, False, True),     # RGB-L / stereo tracking: Tracking.cc:2917-2920 (th = 7)
    (22, "forward", 15.0, False, True),
    (23, "backward", 7.0, False, True),
    (24, "none", 15.0, False, True),
    (25, "forward", 15.0, True, True),     # bMono: level band +-1 regardless of the motion
    (26, "none", 30.0, False, False),      # the retry with a wider window, orientation check off
])
def test_reference_search_by_projection_agrees_with_oracle(refmatcher, seed, motion, th, mono, check_ori):
    import parity_checks as pc
    case = pc.make_projection_case(seed=seed, motion=motion)
    keep = []
    P = O.make_projection_input(case, th, mono, check_ori, keep)
    m = np.zeros(P.n2, np.int32)
    refmatcher.ref_search_by_projection.restype = C.c_int
    refmatcher.ref_search_by_projection.argtypes = [C.c_void_p, C.c_void_p]
    nm = refmatcher.ref_search_by_projection(C.byref(P), m.ctypes.data)
    om, onm = O.search_by_projection(case, th, mono, check_ori)
    assert nm == onm and np.array_equal(m, om)
    assert nm > 300
    # the scenario must exercise the sequential rule: some map point did not get its nearest-descriptor feature
    assert (m >= 0).sum() <= nm


@pytest.mark.parametrize("seed,th,orb_dist,check_ori", [
    (61, 10.0, 100, True),    # Tracking::Relocalization, first widening: ORBmatcher(0.9, true).SearchByProjection(F, pKF, sFound, 10, 100)
    (62, 3.0, 64, True),      # second: th = 3, ORBdist = 64 (Tracking.cc:3723-3752)
    (63, 15.0, 100, False),
    (64, 10.0, 255, True),
])
def test_reference_search_by_projection_keyframe_agrees_with_oracle(refmatcher, seed, th, orb_dist, check_ori):
    """The real ORBmatcher::SearchByProjection(Frame&, KeyFrame*, set, th, ORBdist) on stand-in Frame / KeyFrame / MapPoint
    objects (PredictScale and the invariance range as in MapPoint.cc) against the restatement, and the restated prepass
    against what the reference call actually used."""
    import parity_checks as pc
    case = pc.make_relocalization_case(seed=seed)
    keep = []
    P = O.make_kf_projection_input(case, th, orb_dist, check_ori, keep)
    m = np.zeros(P.n2, np.int32)
    refmatcher.ref_search_by_projection_kf.restype = C.c_int
    refmatcher.ref_search_by_projection_kf.argtypes = [C.c_void_p, C.c_void_p]
    nm = refmatcher.ref_search_by_projection_kf(C.byref(P), m.ctypes.data)
    om, onm = O.search_by_projection_kf(case, th, orb_dist, check_ori)
    assert nm == onm and np.array_equal(m, om)
    assert nm > 150
    # scenario coverage: occupied features stay untouched, found / bad points are never assigned
    assert not np.any((m >= 0) & (case["occupied2"] != 0))
    used = m[m >= 0]
    assert not np.any(case["found1"][used]) and not np.any(case["bad1"][used]) and np.all(case["has_mp1"][used])
    free = dict(case, occupied2=np.zeros_like(case["occupied2"]))
    assert not np.array_equal(O.search_by_projection_kf(free, th, orb_dist, check_ori)[0], om)


@pytest.mark.parametrize("seed,th,nnratio", [(41, 1.0, 0.8), (42, 3.0, 0.8), (43, 5.0, 0.8), (44, 1.0, 0.9), (45, 15.0, 0.7)])
def test_reference_search_local_points_agrees_with_oracle(refmatcher, seed, th, nnratio):
    """ORBmatcher::SearchByProjection(F, vpMapPoints, th) as Tracking::SearchLocalPoints calls it (Tracking.cc:3428-3447:
    th = 1, 3 after a relocalisation, 5 / 6 / 10 / 15 in the IMU cases; ORBmatcher(0.8))."""
    import parity_checks as pc
    case = pc.make_local_points_case(seed=seed)
    keep = []
    P = O.make_local_points_input(case, th, nnratio, keep)
    m = np.zeros(P.n2, np.int32)
    refmatcher.ref_search_local_points.restype = C.c_int
    refmatcher.ref_search_local_points.argtypes = [C.c_void_p, C.c_void_p]
    nm = refmatcher.ref_search_local_points(C.byref(P), m.ctypes.data)
    om, onm = O.search_local_points(case, th, nnratio)
    assert nm == onm and np.array_equal(m, om)
    assert nm > 200
    free = dict(case, blocked2=np.zeros_like(case["blocked2"]), mp_observed1=np.zeros_like(case["mp_observed1"]))
    assert not np.array_equal(O.search_local_points(free, th, nnratio)[0], om)   # the blocking rule matters in this scenario


@pytest.mark.parametrize("seed,window,nnratio,ori", [(61, 100, 0.9, True), (62, 100, 0.9, False), (63, 30, 0.6, True),
                                                     (64, 200, 1.0, True), (65, 100, 1.5, True)])
def test_reference_search_for_initialization_agrees_with_oracle(refmatcher, seed, window, nnratio, ori):
    """ORBmatcher::SearchForInitialization as Tracking::MonocularInitialization calls it (Tracking.cc:2525-2526:
    ORBmatcher(0.9, true), windowSize 100); the scenario takes matches over (vMatchedDistance) several hundred times."""
    import parity_checks as pc
    case = pc.make_initialization_case(5000, seed)
    keep = []
    P = O.make_initialization_input(case, window, nnratio, ori, keep)
    prev = case["prev_matched"].copy()
    m = np.zeros(P.n1, np.int32)
    refmatcher.ref_search_for_initialization.restype = C.c_int
    refmatcher.ref_search_for_initialization.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    nm = refmatcher.ref_search_for_initialization(C.byref(P), prev.ctypes.data, m.ctypes.data)
    om, oprev, onm = O.search_for_initialization(case, window, nnratio, ori)
    assert nm == onm and np.array_equal(m, om) and np.array_equal(prev.view(np.uint32), oprev.view(np.uint32))
    assert nm > 600 and nm == int((m >= 0).sum())
    assert np.all(case["kp1_octave"][m >= 0] == 0) and np.all(case["kp2_octave"][m[m >= 0]] == 0)   # level 0 only
    assert len(np.unique(m[m >= 0])) == nm                                                          # one holder per F2 feature
    # features that lost their match to a later, closer one: unmatched although a partner within TH_LOW exists that a
    # higher-index feature holds at a smaller distance
    if not ori:
        lost = 0
        d1, d2 = case["desc1"], case["desc2"]
        holders = np.nonzero(m >= 0)[0]
        for a in np.nonzero((m < 0) & (case["kp1_octave"] == 0))[0][:400]:
            later = holders[holders > a]
            c = m[later]
            da = np.unpackbits(d1[a][None, :] ^ d2[c], axis=1).sum(1)
            db = np.unpackbits(d1[later] ^ d2[c], axis=1).sum(1)
            near = (np.abs(case["kp2_xy"][c] - case["prev_matched"][a]) < window).all(1)
            lost += bool(((da <= 50) & (db < da) & near).any())
        assert lost > 50


# ---------------------------------------------------------------------------------------------------------------
# The reference's own vendored DBoW2 (oracle/_ref/libref_dbow2.so): vocabulary text loader + transform (Frame::ComputeBoW).
REF_DBOW2_SO = os.path.join(ROOT, "oracle", "_ref", "libref_dbow2.so")


@pytest.fixture(scope="module")
def refdbow(oracle):
    if os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    if not os.path.exists(REF_DBOW2_SO):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    lib = C.CDLL(REF_DBOW2_SO)
    lib.ref_voc_load_text.restype = C.c_void_p
    lib.ref_voc_load_text.argtypes = [C.c_char_p]
    lib.ref_voc_destroy.argtypes = [C.c_void_p]
    lib.ref_voc_transform.restype = C.c_int
    lib.ref_voc_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int),
                                      C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    return lib


def ref_transform(lib, h, desc, levelsup):
    n = len(desc)
    wid, wval = np.zeros(n, np.uint32), np.zeros(n, np.float64)
    nid, noff, nfeat = np.zeros(n, np.uint32), np.zeros(n + 1, np.int32), np.zeros(n, np.uint32)
    nw, nn = C.c_int(0), C.c_int(0)
    rc = lib.ref_voc_transform(h, desc.ctypes.data, n, levelsup, n, wid.ctypes.data, wval.ctypes.data, C.byref(nw), n,
                               nid.ctypes.data, noff.ctypes.data, nfeat.ctypes.data, C.byref(nn))
    assert rc == 0
    return wid[:nw.value], wval[:nw.value], nid[:nn.value], noff[:nn.value + 1], nfeat[:noff[nn.value]]


@pytest.mark.parametrize("k,L,levelsup,seed", [(10, 4, 2, 0), (10, 3, 4, 1), (6, 5, 4, 2), (10, 4, 4, 3)])
def test_reference_dbow2_transform_agrees_with_oracle(refdbow, tmp_path, k, L, levelsup, seed):
    voc = synth.make_vocabulary(k, L, seed)
    path = str(tmp_path / "voc.txt")
    synth.write_vocabulary_text(path, voc)
    h = refdbow.ref_voc_load_text(path.encode())
    assert h
    varr = synth.vocabulary_arrays(voc)
    # features: perturbed copies of leaf descriptors (they land in the neighbourhood of their word) plus random ones
    rng = np.random.default_rng(seed)
    leaves = voc["desc"][voc["is_leaf"] > 0]
    pick = leaves[rng.integers(0, len(leaves), 1500)]
    desc = np.ascontiguousarray(np.concatenate([pick ^ np.packbits(rng.random((1500, 256)) < 0.03, axis=1, bitorder="little"),
                                                synth.descriptors(500, seed)]))
    got = ref_transform(refdbow, h, desc, levelsup)
    want = O.bow_transform(varr, desc, levelsup)
    for g, w, name in zip(got, want, ("word ids", "word values", "node ids", "node offsets", "feature indices")):
        if name == "word values":
            assert np.array_equal(g.view(np.uint64), w.view(np.uint64)), name   # doubles, bit for bit
        else:
            assert np.array_equal(g, w), name
    assert len(got[0]) > 200 and abs(got[1].sum() - 1.0) < 1e-9
    refdbow.ref_voc_destroy(h)


@pytest.mark.parametrize("seed,nnratio,check_ori,nodes", [(81, 0.75, True, 100), (82, 0.9, True, 100), (83, 0.75, False, 30), (84, 0.8, True, 1)])
def test_reference_search_by_bow_keyframes_agrees_with_oracle(refmatcher, seed, nnratio, check_ori, nodes):
    """ORBmatcher::SearchByBoW(pKF1, pKF2, vpMatches12) (LoopClosing: ORBmatcher(0.9, true) / (0.75, true))."""
    import parity_checks as pc
    kf1, kf2, *_ = pc.make_triangulation_case(1500, seed=seed, n_nodes=nodes)
    rng = np.random.default_rng(seed)
    s1 = rng.choice([0, 1, 2], len(kf1["desc"]), p=[0.15, 0.75, 0.1]).astype(np.uint8)   # none / good / bad map point
    s2 = rng.choice([0, 1, 2], len(kf2["desc"]), p=[0.15, 0.75, 0.1]).astype(np.uint8)
    keep = []
    a1 = kf_arrays(dict(kf1, has_mp=s1), keep)
    a2 = kf_arrays(dict(kf2, has_mp=s2), keep)
    m = np.zeros(a1.n, np.int32)
    refmatcher.ref_search_by_bow_kf.restype = C.c_int
    refmatcher.ref_search_by_bow_kf.argtypes = [C.POINTER(KfArrays), C.POINTER(KfArrays), C.c_float, C.c_int, C.c_void_p]
    nm = refmatcher.ref_search_by_bow_kf(C.byref(a1), C.byref(a2), C.c_float(nnratio), int(check_ori), m.ctypes.data)
    om, onm = O.search_by_bow_kf(dict(kf1, has_mp=(s1 == 1).astype(np.uint8)), dict(kf2, has_mp=(s2 == 1).astype(np.uint8)), nnratio, check_ori)
    assert nm == onm and np.array_equal(m, om)
    assert nm > 100
    used = m[m >= 0]
    assert np.all(s2[used] == 1) and len(set(used.tolist())) == len(used)      # only good map points, every kf2 feature at most once


@pytest.mark.parametrize("seed,nnratio,check_ori,nodes", [(51, 0.7, True, 100), (52, 0.7, False, 100), (53, 0.9, True, 30), (54, 0.6, True, 1)])
def test_reference_search_by_bow_agrees_with_oracle(refmatcher, seed, nnratio, check_ori, nodes):
    """ORBmatcher::SearchByBoW(pKF, F, vpMapPointMatches) as Tracking::TrackReferenceKeyFrame calls it (ORBmatcher(0.7, true))."""
    import parity_checks as pc
    kf, fr, *_ = pc.make_triangulation_case(1500, seed=seed, n_nodes=nodes)
    rng = np.random.default_rng(seed)
    state = rng.choice([0, 1, 2], len(kf["desc"]), p=[0.25, 0.65, 0.1]).astype(np.uint8)   # none / good / bad map point
    keep = []
    a1 = kf_arrays(dict(kf, has_mp=state), keep)
    a2 = kf_arrays(fr, keep)
    m = np.zeros(a2.n, np.int32)
    refmatcher.ref_search_by_bow.restype = C.c_int
    refmatcher.ref_search_by_bow.argtypes = [C.POINTER(KfArrays), C.POINTER(KfArrays), C.c_float, C.c_int, C.c_void_p]
    nm = refmatcher.ref_search_by_bow(C.byref(a1), C.byref(a2), C.c_float(nnratio), int(check_ori), m.ctypes.data)
    om, onm = O.search_by_bow(dict(kf, has_mp=(state == 1).astype(np.uint8)), fr, nnratio, check_ori)
    assert nm == onm and np.array_equal(m, om)
    assert nm > 100


@pytest.mark.parametrize("seed,nnratio,check_ori,nodes,kf_rig", [(61, 0.7, True, 100, False), (62, 0.7, False, 100, True), (63, 0.9, True, 30, True),
                                                                 (64, 0.6, True, 1, False), (65, 0.8, True, 5, True)])
def test_reference_search_by_bow_two_camera_frame_agrees_with_oracle(refmatcher, seed, nnratio, check_ori, nodes, kf_rig):
    """ORBmatcher::SearchByBoW(pKF, F, vpMapPointMatches) with F.Nleft != -1 (ORBmatcher.cc:298-326, 357-386): the reference's own
    lines on a stand-in two-camera Frame (and, kf_rig, a two-camera KeyFrame: the angle comes from mvKeys / mvKeysRight then)."""
    from orb_slam3_rgbl_amd.cases import make_bow_rig_case
    kf, fr, n_left = make_bow_rig_case(1500, seed=seed, n_nodes=nodes)
    rng = np.random.default_rng(seed)
    state = rng.choice([0, 1, 2], len(kf["desc"]), p=[0.25, 0.65, 0.1]).astype(np.uint8)   # none / good / bad map point
    keep = []
    a1 = kf_arrays(dict(kf, has_mp=state), keep)
    a2 = kf_arrays(fr, keep)
    m = np.zeros(a2.n, np.int32)
    refmatcher.ref_search_by_bow_rig.restype = C.c_int
    refmatcher.ref_search_by_bow_rig.argtypes = [C.POINTER(KfArrays), C.c_int, C.POINTER(KfArrays), C.c_int, C.c_float, C.c_int, C.c_void_p]
    nm = refmatcher.ref_search_by_bow_rig(C.byref(a1), len(kf["desc"]) // 2 if kf_rig else -1, C.byref(a2), n_left, C.c_float(nnratio), int(check_ori),
                                          m.ctypes.data)
    om, onm = O.search_by_bow(dict(kf, has_mp=(state == 1).astype(np.uint8)), fr, nnratio, check_ori, n_left=n_left)
    assert nm == onm and np.array_equal(m, om)
    left, right = m[:n_left], m[n_left:]
    both = np.intersect1d(left[left >= 0], right[right >= 0])
    assert nm > 150 and len(both) > 30          # map points that went to a left AND a right feature
    # the right camera's features are taken without a ratio test: more of them than a single-camera search of the same features finds
    assert (right >= 0).sum() > 0


@pytest.mark.parametrize("seed,th", [(91, 3.0), (92, 3.0), (93, 4.0), (94, 1.5)])
def test_reference_fuse_agrees_with_oracle(refmatcher, seed, th):
    """The real ORBmatcher::Fuse(pKF, vpMapPoints, th) (LocalMapping::SearchInNeighbors: ORBmatcher().Fuse(pKFi, vpMapPointMatches),
    th = 3) on stand-in objects: which feature every point was fused with is read back from what the function did to them
    (AddObservation / the recorded Replace calls) and must be the restatement's bestIdx; nFused must agree."""
    import parity_checks as pc
    case = pc.make_fuse_case(seed=seed)
    rng = np.random.default_rng(seed)
    state = rng.choice([0, 1, 2, 3], len(case["kp2_xy"]), p=[0.5, 0.2, 0.2, 0.1]).astype(np.uint8)
    keep = []
    P = O.make_fuse_input(case, th, keep)
    best = np.zeros(P.n1, np.int32)
    refmatcher.ref_fuse.restype = C.c_int
    refmatcher.ref_fuse.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    nf = refmatcher.ref_fuse(C.byref(P), state.ctypes.data, best.ctypes.data)
    obest, onf = O.fuse_search(case, th)
    assert nf == onf and nf > 200
    seen = best >= 0
    assert np.array_equal(best[seen], obest[seen])
    # matches the reference counted but left no trace of: the feature held a bad map point (ORBmatcher.cc:1308)
    silent = (~seen) & (obest >= 0)
    assert np.all(state[obest[silent]] == 3) and seen.sum() + silent.sum() == nf
    for s in (0, 1, 2):
        assert np.any(state[obest[seen]] == s)          # AddObservation and both Replace directions were exercised


class Sim3Side(C.Structure):
    _fields_ = [("n", C.c_int), ("kp_xy", C.c_void_p), ("kp_octave", C.c_void_p), ("desc", C.c_void_p), ("mp_state", C.c_void_p),
                ("mp_pos", C.c_void_p), ("mp_normal", C.c_void_p), ("mp_desc", C.c_void_p), ("mp_min_dist", C.c_void_p),
                ("mp_max_dist", C.c_void_p)]


def sim3_side(d, keep):
    a = Sim3Side()
    a.n = len(d["kp_xy"])
    for f, dt in (("kp_xy", np.float32), ("kp_octave", np.int32), ("desc", np.uint8), ("mp_state", np.uint8), ("mp_pos", np.float32),
                  ("mp_normal", np.float32), ("mp_desc", np.uint8), ("mp_min_dist", np.float32), ("mp_max_dist", np.float32)):
        arr = np.ascontiguousarray(d[f], dt)
        keep.append(arr)
        setattr(a, f, arr.ctypes.data)
    return a


@pytest.mark.parametrize("seed,th", [(121, 7.5), (122, 7.5), (123, 3.0), (124, 10.0)])
def test_reference_search_by_sim3_agrees_with_oracle(refmatcher, seed, th):
    """The real ORBmatcher::SearchBySim3 (LoopClosing: matcher.SearchBySim3(mpCurrentKF, pKF, vpMapPointMatches, gScm, 7.5)) with
    identity poses against two oracle per-point searches + the mutual-agreement pass; the prepass written in numpy (invariance
    range, PredictScale on |p3Dc|) is thereby held to what the reference evaluated on the stand-in MapPoint objects."""
    import parity_checks as pc
    case = pc.make_sim3_case(seed=seed)
    keep = []
    a1, a2 = sim3_side(case["a1"], keep), sim3_side(case["a2"], keep)
    K, grid, sf = (np.ascontiguousarray(case[k], np.float32) for k in ("K", "grid", "scale_factors"))
    prior = np.ascontiguousarray(case["prior12"], np.int32)
    m = np.zeros(a1.n, np.int32)
    refmatcher.ref_search_by_sim3.restype = C.c_int
    refmatcher.ref_search_by_sim3.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float,
                                              C.c_void_p, C.c_void_p]
    nf = refmatcher.ref_search_by_sim3(C.byref(a1), C.byref(a2), K.ctypes.data, grid.ctypes.data, sf.ctypes.data, len(sf),
                                       C.c_float(float(case["log_scale_factor"])), C.c_float(th), prior.ctypes.data, m.ctypes.data)
    om, onf = pc.search_by_sim3(case, th, O.project_search)
    assert nf == onf and np.array_equal(m, om)
    assert nf > 300
    new = (m >= 0) & (prior < 0)
    assert np.mean(m[new] == case["inv"][new]) > 0.95          # the agreed matches are the true counterparts


@pytest.mark.parametrize("seed,th", [(131, 4.0), (132, 4.0), (133, 2.5)])
def test_reference_fuse_sim3_agrees_with_oracle(refmatcher, seed, th):
    """The real ORBmatcher::Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) (LoopClosing::SearchAndFuse, th = 4) with an identity Scw."""
    import parity_checks as pc
    case = pc.make_sim3_case(seed=seed)
    rng = np.random.default_rng(seed)
    cand = dict(case["a1"])
    cand["mp_state"] = np.where(cand["mp_state"] == 0, 1, cand["mp_state"]).astype(np.uint8)   # every entry of vpPoints is a point
    in_kf = (rng.random(len(cand["kp_xy"])) < 0.08).astype(np.uint8)
    state2 = rng.choice([0, 1, 3], len(case["a2"]["kp_xy"]), p=[0.5, 0.4, 0.1]).astype(np.uint8)
    keep = []
    c, kfa = sim3_side(cand, keep), sim3_side(case["a2"], keep)
    K, grid, sf = (np.ascontiguousarray(case[k], np.float32) for k in ("K", "grid", "scale_factors"))
    fused = np.zeros(c.n, np.int32)
    refmatcher.ref_fuse_sim3.restype = C.c_int
    refmatcher.ref_fuse_sim3.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                         C.c_float, C.c_float, C.c_void_p]
    nf = refmatcher.ref_fuse_sim3(C.byref(c), in_kf.ctypes.data, C.byref(kfa), state2.ctypes.data, K.ctypes.data, grid.ctypes.data,
                                  sf.ctypes.data, len(sf), C.c_float(float(case["log_scale_factor"])), C.c_float(th), fused.ctypes.data)
    valid, level = pc.camera_prepass(cand["mp_pos"], cand["mp_min_dist"], cand["mp_max_dist"], case["log_scale_factor"], len(sf),
                                     normal=cand["mp_normal"])
    valid &= (cand["mp_state"] == 1) & (in_kf == 0)
    search = dict(valid1=valid.astype(np.uint8), cam_pos1=cand["mp_pos"], mp_desc1=cand["mp_desc"], level1=level,
                  kp2_xy=case["a2"]["kp_xy"], kp2_octave=case["a2"]["kp_octave"], desc2=case["a2"]["desc"], grid=grid, K=K, scale_factors=sf)
    obest, _ = O.project_search(search, th, 0, 50)
    assert nf == int((obest >= 0).sum()) and nf > 300
    seen = fused >= 0
    assert np.array_equal(fused[seen], obest[seen])
    silent = (~seen) & (obest >= 0)                            # the feature held a bad map point: counted, nothing recorded
    holds_bad = state2 == 3
    n2 = len(state2)
    for i in np.nonzero(in_kf)[0]:                             # the glue parks the candidates the key frame "already has" in its last slots
        holds_bad[n2 - 1 - (i % n2)] = cand["mp_state"][i] == 2
    assert np.all(holds_bad[obest[silent]])


@pytest.mark.parametrize("seed,th,ratio,with_kfs", [(161, 8, 1.5, 0), (162, 30, 1.0, 1), (163, 3, 2.5, 1), (164, 8, 1.0, 0)])
def test_reference_search_by_projection_sim3_agrees_with_oracle(refmatcher, seed, th, ratio, with_kfs):
    """The real ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, [vpPointsKFs,] vpMatched, [vpMatchedKF,] th, ratioHamming)
    (LoopClosing::FindMatchesByProjection: th 3 / 8 / 30, ratio 1 / 1.5 / 2.5) with an identity Scw."""
    import parity_checks as pc
    case = pc.make_sim3_case(seed=seed)
    rng = np.random.default_rng(seed)
    cand = dict(case["a1"])
    cand["mp_state"] = np.where(cand["mp_state"] == 0, 1, cand["mp_state"]).astype(np.uint8)
    n1, n2 = len(cand["kp_xy"]), len(case["a2"]["kp_xy"])
    matched2 = (rng.random(n2) < 0.12).astype(np.uint8)
    found = np.zeros(n1, np.uint8)
    found[rng.permutation(n1)[: int(matched2.sum()) // 2]] = 1          # half of the entry matches are candidates of this call
    keep = []
    c, kfa = sim3_side(cand, keep), sim3_side(case["a2"], keep)
    K, grid, sf = (np.ascontiguousarray(case[k], np.float32) for k in ("K", "grid", "scale_factors"))
    m = np.zeros(n2, np.int32)
    refmatcher.ref_search_by_projection_sim3.restype = C.c_int
    refmatcher.ref_search_by_projection_sim3.argtypes = [C.c_void_p] * 7 + [C.c_int, C.c_float, C.c_int, C.c_float, C.c_int, C.c_void_p]
    nm = refmatcher.ref_search_by_projection_sim3(C.byref(c), found.ctypes.data, C.byref(kfa), matched2.ctypes.data, K.ctypes.data,
                                                  grid.ctypes.data, sf.ctypes.data, len(sf), C.c_float(float(case["log_scale_factor"])),
                                                  int(th), C.c_float(ratio), int(with_kfs), m.ctypes.data)
    valid, level = pc.camera_prepass(cand["mp_pos"], cand["mp_min_dist"], cand["mp_max_dist"], case["log_scale_factor"], len(sf),
                                     normal=cand["mp_normal"])
    valid &= (cand["mp_state"] == 1) & (found == 0)
    search = dict(valid1=valid.astype(np.uint8), cam_pos1=cand["mp_pos"], mp_desc1=cand["mp_desc"], level1=level,
                  kp2_xy=case["a2"]["kp_xy"], kp2_octave=case["a2"]["kp_octave"], desc2=case["a2"]["desc"], grid=grid, K=K, scale_factors=sf)
    max_dist = int(np.floor(np.float32(50) * np.float32(ratio)))
    om, onm = O.search_by_projection_sim3(search, matched2, float(th), 2 if with_kfs else 0, max_dist)
    assert nm == onm and np.array_equal(m, om)
    assert nm > 300 and not np.any((m >= 0) & (matched2 != 0))


# ---- Frame::ComputeStereoMatches: the reference's own lines (src/Frame.cc:901-1071, extracted at build time) on the
# reference's own ORBextractor / ORBmatcher, versus the oracle's restatement
REF_FRAME_SO = os.path.join(ROOT, "oracle", "_ref", "libref_frame.so")


@pytest.fixture(scope="module")
def refframe(oracle):
    if os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    if not os.path.exists(REF_FRAME_SO):
        pytest.skip("oracle/_ref/libref_frame.so not built (needs /root/reference)")
    lib = C.CDLL(REF_FRAME_SO)
    lib.ref_stereo_matches.restype = C.c_int
    lib.ref_stereo_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                                       C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                       C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_void_p]
    return lib


@pytest.mark.parametrize("w,h,nf,ini,mn,seq,mb,mbf,dscale", [
    (1241, 376, 2000, 20, 7, 30, 0.54, 386.1448, 1.0),     # KITTI stereo settings (BASELINE cfg 3)
    (1241, 376, 2000, 20, 7, 31, 0.54, 386.1448, 1.5),     # large disparities: more candidates per row band, border windows
    (752, 480, 1200, 20, 7, 32, 0.11, 47.9, 0.6),          # EuRoC shape and baseline
    (640, 480, 1000, 12, 7, 33, 0.08, 40.0, 0.25),         # tiny disparities: the disparity <= 0 clamp and the parabola gate
])
def test_reference_compute_stereo_matches_agrees_with_oracle(refframe, w, h, nf, ini, mn, seq, mb, mbf, dscale):
    import parity_checks as pc
    left, right = pc.stereo_pair(seq, w, h, disparity_scale=dscale)
    cap = nf * 2 + 4096
    kl, kr = np.zeros(cap, O.KP_DTYPE), np.zeros(cap, O.KP_DTYPE)
    dl, dr = np.zeros((cap, 32), np.uint8), np.zeros((cap, 32), np.uint8)
    ur, dp = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
    nl, nr = C.c_int(0), C.c_int(0)
    rc = refframe.ref_stereo_matches(left.ctypes.data, right.ctypes.data, w, h, left.strides[0], nf, 1.2, 8, ini, mn, mb, mbf,
                                     kl.ctypes.data, dl.ctypes.data, kr.ctypes.data, dr.ctypes.data, cap, C.byref(nl), C.byref(nr),
                                     ur.ctypes.data, dp.ctypes.data)
    assert rc == 0
    nl, nr = nl.value, nr.value
    ol, orr = O.Extractor(nf, 1.2, 8, ini, mn), O.Extractor(nf, 1.2, 8, ini, mn)
    okl, odl, _ = ol(left)
    okr, odr, _ = orr(right)
    assert len(okl) == nl and len(okr) == nr
    for f in ("x", "y", "octave"):
        assert np.array_equal(kl[:nl][f], okl[f]) and np.array_equal(kr[:nr][f], okr[f])
    assert np.array_equal(dl[:nl], odl) and np.array_equal(dr[:nr], odr)
    our, odp = O.stereo_matches(ol, orr, okl, odl, okr, odr, mb, mbf)
    assert np.array_equal(ur[:nl].view(np.uint32), our.view(np.uint32)), "mvuRight"
    assert np.array_equal(dp[:nl].view(np.uint32), odp.view(np.uint32)), "mvDepth"
    assert (our >= 0).sum() > 100  # the case must produce matches


def test_reference_compute_stereo_matches_on_a_real_stereo_pair(refframe):
    """Row f1 on real imagery: the rectified Middlebury 'Motorcycle' pair that ships with scikit-image, read
    in place - the reference's own Frame::ComputeStereoMatches lines on the reference's own extractor against the oracle, bit
    for bit, and both against the pair's ground-truth disparity (the function does what it is meant to do on a photograph)."""
    import real_images as R
    lrgb, rrgb, gt = R.stereo_pair()
    left, right = O.cvt_gray(lrgb, True), O.cvt_gray(rrgb, True)
    h, w = left.shape
    nf, ini, mn, mb, mbf = 1500, 20, 7, R.MOTORCYCLE_MB, R.MOTORCYCLE_MBF
    cap = nf * 2 + 4096
    kl, kr = np.zeros(cap, O.KP_DTYPE), np.zeros(cap, O.KP_DTYPE)
    dl, dr = np.zeros((cap, 32), np.uint8), np.zeros((cap, 32), np.uint8)
    ur, dp = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
    nl, nr = C.c_int(0), C.c_int(0)
    rc = refframe.ref_stereo_matches(left.ctypes.data, right.ctypes.data, w, h, left.strides[0], nf, 1.2, 8, ini, mn, mb, mbf,
                                     kl.ctypes.data, dl.ctypes.data, kr.ctypes.data, dr.ctypes.data, cap, C.byref(nl), C.byref(nr),
                                     ur.ctypes.data, dp.ctypes.data)
    assert rc == 0
    nl, nr = nl.value, nr.value
    ol, orr = O.Extractor(nf, 1.2, 8, ini, mn), O.Extractor(nf, 1.2, 8, ini, mn)
    okl, odl, _ = ol(left)
    okr, odr, _ = orr(right)
    assert len(okl) == nl and len(okr) == nr
    assert np.array_equal(dl[:nl], odl) and np.array_equal(dr[:nr], odr)
    our, odp = O.stereo_matches(ol, orr, okl, odl, okr, odr, mb, mbf)
    assert np.array_equal(ur[:nl].view(np.uint32), our.view(np.uint32)), "mvuRight"
    assert np.array_equal(dp[:nl].view(np.uint32), odp.view(np.uint32)), "mvDepth"
    m = our >= 0
    assert m.sum() > 400
    if gt is not None:
        truth = gt[np.round(okl["y"][m]).astype(int), np.round(okl["x"][m]).astype(int)]
        known = np.isfinite(truth)
        err = np.abs((okl["x"][m] - our[m])[known] - truth[known])
        assert np.median(err) < 0.5 and (err < 2.0).mean() > 0.9     # sub-pixel on a photograph: 0.27 px median here


# ---------------------------------------------------------------------------------------------------------------
# The two other RGB-L settings files the reference ships (Examples/RGB-L/KITTI04-12.yaml, KITTIxx-03.yaml: KITTI sequences
# 03 - 12) still carry the stale key LiDAR.MethodInverseDilation.KernelSize; DepthModule.cc:566-582 asks for KernelSize_u /
# KernelSize_v, so ParseUpsamplingParameters fails, CalculateDepthFromPcd returns at once (:52-55) and every keypoint of those
# sequences keeps mvDepth = -1: as shipped, the reference runs them as monocular frames.  Read in place, not copied.
SHIPPED = "/root/reference/Examples/RGB-L"


@pytest.mark.parametrize("name,fx", [("KITTI04-12.yaml", 707.0912), ("KITTIxx-03.yaml", 721.5377)])
def test_reference_disables_upsampling_on_its_other_shipped_settings(refdepth, name, fx):
    path = os.path.join(SHIPPED, name)
    if not os.path.exists(path):
        pytest.skip("the reference's Examples/RGB-L is not on this host")
    R = RefDepth(refdepth, path)
    assert R.parsed == (True, False)
    # K * Tr is still built, with this file's intrinsics, as the restated oracle builds it
    K = synth.KITTI_K.copy()
    cx, cy = {"KITTI04-12.yaml": (601.8873, 183.1104), "KITTIxx-03.yaml": (609.5593, 172.854)}[name]
    K[0, 0] = K[1, 1] = fx
    K[0, 2], K[1, 2] = cx, cy
    assert np.array_equal(bits(R.proj), bits(O.projection_matrix(K, synth.KITTI_TR)))
    # and a frame comes back untouched: rc != 0, no depth
    w, h = 1226, 370
    xy, un = keypoints_for(w, h, 200, 1)
    rc, d, ur, raw, proc = R(synth.lidar_scan(2), w, h, xy, un)
    assert rc != 0 and not d.any() and not raw.any()
    R.close()
    # the settings file the golden fixture was taken from parses
    R = RefDepth(refdepth, os.path.join(SHIPPED, "KITTI00-02.yaml"))
    assert R.parsed == (True, True)
    R.close()
