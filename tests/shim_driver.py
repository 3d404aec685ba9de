"""Builds and runs tests/shim_test.cpp (the C++ drop-in classes) against a given librgbl_frontend variant and
checks everything it dumps against the oracle."""
import os
import struct
import subprocess

import numpy as np

import parity_checks as pc
from oracle import oracle_py as O
from orb_slam3_rgbl_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "orb_slam3_rgbl_amd", "shim")


def build(libdir, libname, exe):
    """Compiles the shim test program; safe under pytest-xdist (file lock, atomic replace, skipped when up to date)."""
    import fcntl
    import glob
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    srcs = [os.path.join(ROOT, "tests", "shim_test.cpp"), os.path.join(SHIM, "ORBextractor.cc"), os.path.join(SHIM, "DepthModule.cc")]
    deps = srcs + glob.glob(os.path.join(SHIM, "*.h")) + [os.path.join(ROOT, "include", "rgbl_frontend.h"), os.path.join(ROOT, "tests", "shim_standins.h"),
                                                           os.path.join(libdir, "lib%s.so" % libname)]
    with open(exe + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if os.path.exists(exe) and all(os.path.getmtime(exe) >= os.path.getmtime(d) for d in deps if os.path.exists(d)):
            return
        tmp = "%s.tmp.%d" % (exe, os.getpid())
        cmd = ["g++", "-O1", "-std=c++17", "-DRGBL_FORCE_CV_COMPAT", "-I" + SHIM] + srcs + ["-o", tmp, "-L" + libdir,
               "-l" + libname, "-Wl,-rpath," + libdir, "-pthread"]
        subprocess.check_call(cmd)
        os.replace(tmp, exe)


def write_kf(f, kf, sf, s2, Tcw, Twc):
    n = len(kf["desc"])
    f.write(struct.pack("<i", n))
    f.write(np.ascontiguousarray(kf["desc"], np.uint8).tobytes())
    for key, dt in (("xy", np.float32), ("octave", np.int32), ("angle", np.float32), ("uright", np.float32), ("has_mp", np.uint8)):
        f.write(np.ascontiguousarray(kf[key], dt).tobytes())
    f.write(struct.pack("<i", len(kf["node_id"])))
    for key in ("node_id", "node_off", "node_feat"):
        f.write(np.ascontiguousarray(kf[key], np.int32).tobytes())
    f.write(sf.astype(np.float32).tobytes())
    f.write(s2.astype(np.float32).tobytes())
    for T in (Tcw, Twc):
        f.write(T[:3, :3].astype(np.float32).tobytes())
        f.write(T[:3, 3].astype(np.float32).tobytes())


def run_and_check(exe, tmp):
    w, h = synth.KITTI_W, synth.KITTI_H
    img = synth.Sequence(21, w, h, 1).frame(0)
    cloud = synth.lidar_scan(21)
    img.tofile(os.path.join(tmp, "img.raw"))
    np.ascontiguousarray(cloud).tofile(os.path.join(tmp, "cloud.raw"))
    kf1, kf2, K, R, t, ep_unused, sf, s2 = pc.make_triangulation_case(1200, seed=31)
    # poses: camera 1 at the origin, camera 2 translated; T12 = T1w * Tw2
    T1w = np.eye(4, dtype=np.float32)
    Tw2 = np.eye(4, dtype=np.float32)
    Tw2[:3, 3] = [0.54, -0.01, 0.9]
    T2w = np.linalg.inv(Tw2).astype(np.float32)
    with open(os.path.join(tmp, "tri.bin"), "wb") as f:
        f.write(K.astype(np.float32).tobytes())
        write_kf(f, kf1, sf, s2, T1w, np.linalg.inv(T1w).astype(np.float32))
        write_kf(f, kf2, sf, s2, T2w, Tw2)
    case = pc.make_projection_case(1500, 1700, seed=33, motion="forward")
    with open(os.path.join(tmp, "proj.bin"), "wb") as f:
        f.write(struct.pack("<iifii", len(case["valid1"]), len(case["kp2_xy"]), 7.0, 0, 1))
        hdr = np.concatenate([case["grid"], case["Tcw_q"], case["Tcw_t"], case["Tlw_q"], case["Tlw_t"], case["K"],
                              [case["mb"], case["mbf"]], case["scale_factors"]]).astype(np.float32)
        assert hdr.size == 34
        f.write(hdr.tobytes())
        for key, dt in (("valid1", np.uint8), ("world_pos1", np.float32), ("mp_desc1", np.uint8), ("mp_observed1", np.uint8),
                        ("octave1", np.int32), ("angle1", np.float32), ("kp2_xy", np.float32), ("kp2_octave", np.int32),
                        ("kp2_angle", np.float32), ("uright2", np.float32), ("desc2", np.uint8)):
            f.write(np.ascontiguousarray(case[key], dt).tobytes())
    lcase = pc.make_local_points_case(2500, 1800, seed=47)
    with open(os.path.join(tmp, "local.bin"), "wb") as f:
        f.write(struct.pack("<iiff", len(lcase["valid1"]), len(lcase["kp2_xy"]), 3.0, 0.8))
        f.write(np.concatenate([lcase["grid"], lcase["scale_factors"]]).astype(np.float32).tobytes())
        for key, dt in (("valid1", np.uint8), ("proj1", np.float32), ("level1", np.int32), ("view_cos1", np.float32),
                        ("mp_desc1", np.uint8), ("mp_observed1", np.uint8), ("kp2_xy", np.float32), ("kp2_octave", np.int32),
                        ("uright2", np.float32), ("desc2", np.uint8), ("blocked2", np.uint8)):
            f.write(np.ascontiguousarray(lcase[key], dt).tobytes())
    rcase = pc.make_relocalization_case(1400, 1600, seed=71)
    with open(os.path.join(tmp, "reloc.bin"), "wb") as f:
        f.write(struct.pack("<iifii", len(rcase["has_mp1"]), len(rcase["kp2_xy"]), 10.0, 100, 1))
        hdr = np.concatenate([rcase["grid"], rcase["Tcw_q"], rcase["Tcw_t"], rcase["K"], rcase["scale_factors"],
                              [rcase["log_scale_factor"]]]).astype(np.float32)
        assert hdr.size == 26
        f.write(hdr.tobytes())
        for key, dt in (("has_mp1", np.uint8), ("bad1", np.uint8), ("found1", np.uint8), ("world_pos1", np.float32),
                        ("mp_desc1", np.uint8), ("min_dist1", np.float32), ("max_dist1", np.float32), ("angle1", np.float32),
                        ("kp2_xy", np.float32), ("kp2_octave", np.int32), ("kp2_angle", np.float32), ("desc2", np.uint8),
                        ("occupied2", np.uint8)):
            f.write(np.ascontiguousarray(rcase[key], dt).tobytes())
    fcase = pc.make_fuse_case(2000, 1600, seed=95)
    fstate = np.random.default_rng(95).choice([0, 1, 2, 3], len(fcase["kp2_xy"]), p=[0.5, 0.2, 0.2, 0.1]).astype(np.uint8)
    with open(os.path.join(tmp, "fuse.bin"), "wb") as f:
        f.write(struct.pack("<iif", len(fcase["has_mp1"]), len(fcase["kp2_xy"]), 3.0))
        hdr = np.concatenate([fcase["grid"], fcase["Tcw_q"], fcase["Tcw_t"], fcase["Ow"], fcase["K"], [fcase["bf"]],
                              fcase["scale_factors"], fcase["inv_level_sigma2"], [fcase["log_scale_factor"]]]).astype(np.float32)
        assert hdr.size == 38
        f.write(hdr.tobytes())
        for key, dt in (("has_mp1", np.uint8), ("bad1", np.uint8), ("in_kf1", np.uint8), ("world_pos1", np.float32),
                        ("normal1", np.float32), ("mp_desc1", np.uint8), ("min_dist1", np.float32), ("max_dist1", np.float32),
                        ("kp2_xy", np.float32), ("kp2_octave", np.int32), ("uright2", np.float32), ("desc2", np.uint8)):
            f.write(np.ascontiguousarray(fcase[key], dt).tobytes())
        f.write(fstate.tobytes())
    scase = pc.make_sim3_case(1200, seed=141)
    with open(os.path.join(tmp, "sim3.bin"), "wb") as f:
        f.write(struct.pack("<iff", len(scase["a1"]["kp_xy"]), 7.5, 4.0))
        hdr = np.concatenate([scase["K"], scase["grid"], scase["scale_factors"], [scase["log_scale_factor"]]]).astype(np.float32)
        assert hdr.size == 19
        f.write(hdr.tobytes())
        for side in ("a1", "a2"):
            for key, dt in (("kp_xy", np.float32), ("kp_octave", np.int32), ("desc", np.uint8), ("mp_state", np.uint8),
                            ("mp_pos", np.float32), ("mp_normal", np.float32), ("mp_desc", np.uint8), ("mp_min_dist", np.float32),
                            ("mp_max_dist", np.float32)):
                f.write(np.ascontiguousarray(scase[side][key], dt).tobytes())
        f.write(np.ascontiguousarray(scase["prior12"], np.int32).tobytes())
    icase = pc.make_initialization_case(3000, seed=151)
    with open(os.path.join(tmp, "init.bin"), "wb") as f:
        f.write(struct.pack("<iii", len(icase["kp1_octave"]), len(icase["kp2_xy"]), 100))
        f.write(np.ascontiguousarray(icase["grid"], np.float32).tobytes())
        for key, dt in (("kp1_octave", np.int32), ("kp1_angle", np.float32), ("desc1", np.uint8), ("prev_matched", np.float32),
                        ("kp2_xy", np.float32), ("kp2_octave", np.int32), ("kp2_angle", np.float32), ("desc2", np.uint8)):
            f.write(np.ascontiguousarray(icase[key], dt).tobytes())
    voc = synth.make_vocabulary(10, 3, seed=5)
    synth.write_vocabulary_text(os.path.join(tmp, "voc.txt"), voc)
    out = os.path.join(tmp, "out.bin")
    res = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "KITTI00-02.yaml"), os.path.join(tmp, "img.raw"), str(w), str(h),
                          os.path.join(tmp, "cloud.raw"), str(cloud.shape[1]), os.path.join(tmp, "tri.bin"), out,
                          os.path.join(tmp, "proj.bin"), os.path.join(tmp, "local.bin"), os.path.join(tmp, "voc.txt"),
                          os.path.join(tmp, "reloc.bin"), os.path.join(tmp, "fuse.bin"), os.path.join(tmp, "sim3.bin"),
                          os.path.join(tmp, "init.bin")],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "Lidar Method: InverseDilation" in res.stdout
    buf = open(out, "rb").read()
    pos = 0

    def take(dtype, count):
        nonlocal pos
        a = np.frombuffer(buf, dtype, count, pos)
        pos += a.nbytes
        return a

    mono, nk = take(np.int32, 2)
    kps = take(O.KP_DTYPE, nk)
    desc = take(np.uint8, nk * 32).reshape(nk, 32)
    sfs = take(np.float32, 8)
    pw, ph = take(np.int32, 2)
    pyr3 = take(np.uint8, (pw + 38) * (ph + 38)).reshape(ph + 38, pw + 38)
    mono_empty = take(np.int32, 1)[0]
    nd = take(np.int32, 1)[0]
    mvDepth, mvuRight = take(np.float32, nd), take(np.float32, nd)
    processed = take(np.float32, w * h).reshape(h, w)
    proj = take(np.float32, 12).reshape(3, 4)
    nb = take(np.int32, 1)[0]
    binDepth, binURight = take(np.float32, nb), take(np.float32, nb)
    mono_col, nc = take(np.int32, 2)
    kcol = take(O.KP_DTYPE, nc)
    dcol = take(np.uint8, nc * 32).reshape(nc, 32)
    gray = take(np.uint8, w * h).reshape(h, w)
    kun = take(np.float32, 2 * nc).reshape(nc, 2)
    nm, npairs = take(np.int32, 2)
    pairs = take(np.int32, 2 * npairs).reshape(npairs, 2)
    dd = take(np.int32, 1)[0]
    nbow, n2b = take(np.int32, 2)
    bow_match = take(np.int32, n2b)
    nrig = take(np.int32, 1)[0]
    rig_match = take(np.int32, n2b)
    nproj, n2p = take(np.int32, 2)
    proj_match = take(np.int32, n2p)
    nloc, n2l = take(np.int32, 2)
    local_match = take(np.int32, n2l)
    nw = take(np.int32, 1)[0]
    bow = np.frombuffer(buf, np.dtype([("id", "<u4"), ("val", "<f8")]), nw, pos)
    pos += bow.nbytes
    nn = take(np.int32, 1)[0]
    fv = []
    for _ in range(nn):
        nid, cnt = take(np.uint32, 1)[0], take(np.int32, 1)[0]
        fv.append((int(nid), take(np.uint32, cnt).copy()))
    nreloc, n2r = take(np.int32, 2)
    reloc_match = take(np.int32, n2r)
    nkk, n1k = take(np.int32, 2)
    kk_match = take(np.int32, n1k)
    nfused, nq = take(np.int32, 2)
    fused_with = take(np.int32, nq)
    best_desc = take(np.int32, 1)[0]
    nsim, n1s = take(np.int32, 2)
    sim_match = take(np.int32, n1s)
    nfs, nqs = take(np.int32, 2)
    fused_sim = take(np.int32, nqs)
    greedy = []
    for _ in range(2):
        nms = take(np.int32, 1)[0]
        greedy.append((nms, take(np.int32, n1s).copy()))
    ninit, n1i = take(np.int32, 2)
    init_match = take(np.int32, n1i)
    init_prev = take(np.float32, 2 * n1i).reshape(n1i, 2)
    hooks_same = take(np.int32, 1)[0]
    assert pos == len(buf)
    assert hooks_same == 1, "ORBextractor::Begin / DepthModule::PrefetchPointcloud changed the results"
    oim, oiprev, oin = O.search_for_initialization(icase, 100, 0.9, True)
    assert ninit == oin and np.array_equal(init_match, oim) and np.array_equal(pc.bits(init_prev), pc.bits(oiprev)) and ninit > 300
    osm, osn = pc.search_by_sim3(scase, 7.5, O.project_search)
    assert nsim == osn and np.array_equal(sim_match, osm) and nsim > 200
    # Fuse(pKF2, Scw = I, points of KF1, 4): candidates = the features of KF1 that hold a point, in index order
    a1, a2 = scase["a1"], scase["a2"]
    sel = np.nonzero(a1["mp_state"])[0]
    fvalid, flevel = pc.camera_prepass(a1["mp_pos"][sel], a1["mp_min_dist"][sel], a1["mp_max_dist"][sel], scase["log_scale_factor"], 8,
                                       normal=a1["mp_normal"][sel])
    fvalid &= a1["mp_state"][sel] == 1
    fbest, _ = O.project_search(dict(valid1=fvalid.astype(np.uint8), cam_pos1=a1["mp_pos"][sel], mp_desc1=a1["mp_desc"][sel], level1=flevel,
                                     kp2_xy=a2["kp_xy"], kp2_octave=a2["kp_octave"], desc2=a2["desc"], grid=scase["grid"], K=scase["K"],
                                     scale_factors=scase["scale_factors"]), 4.0, 0, 50)
    assert nfs == int((fbest >= 0).sum()) and np.array_equal(fused_sim, fbest[fbest >= 0]) and nfs > 200
    matched2 = np.zeros(len(a2["kp_xy"]), np.uint8)
    matched2[::9] = 1
    gcase = dict(valid1=fvalid.astype(np.uint8), cam_pos1=a1["mp_pos"][sel], mp_desc1=a1["mp_desc"][sel], level1=flevel,
                 kp2_xy=a2["kp_xy"], kp2_octave=a2["kp_octave"], desc2=a2["desc"], grid=scase["grid"], K=scase["K"],
                 scale_factors=scase["scale_factors"])
    for (nms, gm), (th_g, ratio_g, form_g) in zip(greedy, ((8, 1.5, 0), (30, 1.0, 2))):
        ogm, ogn = O.search_by_projection_sim3(gcase, matched2, float(th_g), form_g, int(np.floor(np.float32(50) * np.float32(ratio_g))))
        want = np.where(ogm >= 0, sel[np.maximum(ogm, 0)], -1)        # candidate index -> feature index of KF1 (the shim test's pointer base)
        assert nms == ogn and np.array_equal(gm, want) and nms > 150
    rows = [kf1["desc"][i * 3 % len(kf1["desc"])] for i in range(25)]
    assert best_desc == O.distinctive_descriptors([np.stack(rows)])[0]
    ofb, onf = O.fuse_search(fcase, 3.0)
    assert nfused == onf and np.array_equal(fused_with, ofb[ofb >= 0]) and nfused > 150   # in order: one GetMapPoint per fused point
    okk, onkk = O.search_by_bow_kf(kf1, kf2, 0.75, True)
    assert nkk == onkk and np.array_equal(kk_match, okk) and nkk > 50
    oun = O.undistort_points(np.stack([kcol["x"], kcol["y"]], 1), (458.654, 457.296, 367.215, 248.375),
                             (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05))
    assert np.array_equal(pc.bits(kun), pc.bits(oun)) and nc > 500
    orm, orn = O.search_by_projection_kf(rcase, 10.0, 100, True)
    assert nreloc == orn and np.array_equal(reloc_match, orm) and nreloc > 150
    wid, wval, onid, onoff, onfeat = O.bow_transform(synth.vocabulary_arrays(voc), desc, 2)   # desc == the oracle's, checked below
    assert np.array_equal(bow["id"], wid) and np.array_equal(bow["val"].view(np.uint64), wval.view(np.uint64)) and nw > 50
    assert [f[0] for f in fv] == onid.tolist()
    for j, (_, feats) in enumerate(fv):
        assert np.array_equal(feats, onfeat[onoff[j]:onoff[j + 1]])
    olm, oln = O.search_local_points(lcase, 3.0, 0.8)
    assert nloc == oln and np.array_equal(local_match, olm) and nloc > 150
    om, onm = O.search_by_projection(case, 7.0, False, True)
    assert nproj == onm and np.array_equal(proj_match, om) and nproj > 200

    orc = O.Extractor(2000, 1.2, 8, 12, 7)
    okps, odesc, omono = orc(img)
    pc.assert_keypoints_equal(kps, okps, "shim extractor")
    assert np.array_equal(desc, odesc) and mono == omono and mono_empty == -1
    assert np.array_equal(pc.bits(sfs), pc.bits(orc.tables()["scale"]))
    assert np.array_equal(pyr3, orc.level_bordered(3))
    oproj = O.projection_matrix(synth.KITTI_K, synth.KITTI_TR)
    assert np.array_equal(pc.bits(proj), pc.bits(oproj))
    P = O.make_depth_params(oproj)
    d, ur, raw, proc = O.depth(P, cloud, w, h, np.stack([okps["x"], okps["y"]], 1), okps["x"])
    assert nd == nk and np.array_equal(pc.bits(mvDepth), pc.bits(d)) and np.array_equal(pc.bits(mvuRight), pc.bits(ur))
    assert np.array_equal(pc.bits(processed), pc.bits(proc))
    # ingest: the .bin layout gives the same depths; cvtColor + extraction == oracle cvtColor, then the oracle extractor
    assert nb == nk and np.array_equal(pc.bits(binDepth), pc.bits(d)) and np.array_equal(pc.bits(binURight), pc.bits(ur))
    col = np.stack([img, 255 - img, (img // 2 + (np.arange(w) % 50)[None, :]).astype(np.uint8)], 2)
    ogray = O.cvt_gray(col, True)
    assert np.array_equal(gray, ogray)
    ck, cd, cm = orc(ogray)
    pc.assert_keypoints_equal(kcol, ck, "shim ExtractColor")
    assert np.array_equal(dcol, cd) and mono_col == cm and nc > 500
    # triangulation: F12 / epipole as the shim derives them from the poses
    T12 = T1w @ Tw2
    F = O.fundamental(K, K, T12[:3, :3].reshape(9), T12[:3, 3])
    C2 = T2w[:3, :3] @ np.zeros(3, np.float32) + T2w[:3, 3]
    ep = np.array([K[0] * C2[0] / C2[2] + K[2], K[1] * C2[1] / C2[2] + K[3]], np.float32)
    om12, onm = O.search_triangulation(kf1, kf2, F, ep, sf, s2, False, False, False)
    idx1 = np.nonzero(om12 >= 0)[0]
    assert nm == onm == npairs and np.array_equal(pairs[:, 0], idx1) and np.array_equal(pairs[:, 1], om12[idx1])
    assert nm > 20
    assert dd == O.descriptor_distance(kf1["desc"][0], kf2["desc"][0])
    obm, obn = O.search_by_bow(kf1, kf2, 0.7, True)
    assert nbow == obn and np.array_equal(bow_match, obm) and nbow > 50
    orm_, orn_ = O.search_by_bow(kf1, kf2, 0.7, True, n_left=len(kf2["desc"]) // 2)   # the frame's second half as a right camera's features
    assert nrig == orn_ and np.array_equal(rig_match, orm_) and nrig > 50 and not np.array_equal(rig_match, bow_match)
