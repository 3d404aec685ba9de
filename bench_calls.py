"""bench_calls.py - latency of the matcher entry points a drop-in Tracking / LocalMapping calls once per frame / key frame,
one KITTI-size frame per call through the host-pointer C ABI (PCIe, launches and the final synchronisation included), with
the reference's OWN ORBmatcher.cc / DBoW2 / Frame::ComputeStereoMatches (oracle/_ref, one host thread) timed beside each.

bench.py puts the two objects into its JSON line:
  extra.cfg3_triangulation   BASELINE configs[2]: ORBmatcher::SearchForTriangulation (ORBmatcher.cc:907-1146) the way
                             LocalMapping::CreateNewMapPoints calls it (LocalMapping.cc:412, 466: ORBmatcher(0.6, false),
                             bOnlyStereo = false, bCoarse = false), N1 = N2 = 2000 and 8000, <= 100 vocabulary nodes
  extra.tracking_calls       SearchByProjection(CurrentFrame, LastFrame, th = 15) (Tracking.cc:2917-2934), SearchByProjection(F,
                             vpMapPoints, th = 1) (Tracking::SearchLocalPoints), ORBVocabulary::transform (Frame::ComputeBoW),
                             SearchByBoW(pKF, F) (TrackReferenceKeyFrame), Frame::ComputeStereoMatches (stereo mode)
Per call: median / min / max wall time of the C call (the input block is prepared once, a ctypes call adds ~1 us), the kernels'
own time from HIP events (a separate short run with the handle's profiling on), a parity check of the result against the
oracle, and the reference CPU time of the same input.  Never part of `value`.
"""
import os
import tempfile
import time

import numpy as np

CALLS, WARM, REF_CALLS = int(os.environ.get("RGBL_CALL_BENCH_N", "200")), 10, 9


def _stats_us(ts):
    a = np.sort(np.asarray(ts, np.float64)) * 1e6
    return {"median_us": round(float(a[len(a) // 2]), 1), "min_us": round(float(a[0]), 1), "max_us": round(float(a[-1]), 1)}


def _time_call(call, n=CALLS, warm=WARM):
    for _ in range(warm):
        call()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        call()
        ts.append(time.perf_counter() - t0)
    return _stats_us(ts)


def _kernel_us(handle, call, n=20):
    """HIP-event time of the call's kernels (profiling brackets add host work: measured apart from the wall times)."""
    handle.profile(True)
    for _ in range(n):
        call()
    prof = handle.profile_read()
    handle.profile(False)
    return {k: round(v[0] / n * 1e3, 1) for k, v in prof.items()}, round(sum(v[0] for v in prof.values()) / n * 1e3, 1)


def _ref_stats(fn, n=REF_CALLS):
    ts = []
    out = None
    for _ in range(n):
        out = fn()
        ts.append(out[-1])
    s = _stats_us(ts)
    return out, {"median_us": s["median_us"], "min_us": s["min_us"], "calls": n}


def _entry(gpu, kern, kern_total, ref, parity, resident=None, **more):
    e = {"gpu_call": gpu, "kernels_us": kern, "kernels_total_us": kern_total, "parity": parity}
    if resident is not None:
        # the same call with the frame(s) resident on the device (rgbl_device_frame): their descriptors / keypoints are not uploaded
        e["gpu_call_resident"] = resident
    if ref is not None:
        e["cpu_reference"] = ref
        e["gpu_over_cpu_time"] = round(gpu["median_us"] / ref["median_us"], 3) if ref["median_us"] else None
        if resident is not None and ref["median_us"]:
            e["gpu_resident_over_cpu_time"] = round(resident["median_us"] / ref["median_us"], 3)
    e.update(more)
    return e


def real_frames(lib, n=2, seq=7, nfeatures=2000, ini=12, mn=7, w=None, h=None):
    """Keypoints / descriptors / LiDAR uRight of consecutive synthetic KITTI frames from the library's own front end."""
    from orb_slam3_rgbl_amd import frontend as F
    from orb_slam3_rgbl_amd import synth
    w, h = w or synth.KITTI_W, h or synth.KITTI_H
    sq = synth.Sequence(seq, w, h, n_frames=n)
    ex = F.ORBextractor(nfeatures, 1.2, 8, ini, mn, w, h, lib=lib)
    proj = F.projection_matrix(synth.KITTI_K, synth.KITTI_TR, lib)
    dm = F.DepthModule(proj, w, h, max_keypoints=4 * nfeatures + 4096, lib=lib)
    out = []
    for i in range(n):
        kps, desc, _ = ex(sq.frame(i))
        dm.CalculateDepthFromPcd(kps, kps, synth.lidar_scan(seq + i), w, h)
        # the frame's resident copy: straight from the extractor's and the depth module's device results (no PCIe)
        dev = F.DeviceFrame(len(kps) + 64, lib=lib).capture(ex, len(kps), dm)
        dev.set_grid(np.array([0, 0, w, h, np.float32(64) / np.float32(w), np.float32(48) / np.float32(h)], np.float32))   # Frame::AssignFeaturesToGrid, once
        out.append(dict(xy=np.stack([kps["x"], kps["y"]], 1).astype(np.float32), desc=desc.copy(), octave=kps["octave"].astype(np.int32),
                        angle=kps["angle"].astype(np.float32), uright=dm.mvuRight.copy(), resident=dev))
    dm.close()
    ex.close()
    return out


def triangulation_leg(lib, sizes=(2000, 8000)):
    from oracle import oracle_py as O
    from oracle import ref_py as R
    from orb_slam3_rgbl_amd import cases
    from orb_slam3_rgbl_amd import frontend as F
    ref = R.load_matcher()
    res = {"what": "ORBmatcher::SearchForTriangulation (ORBmatcher.cc:907-1146) as LocalMapping.cc:412,466 calls it: ORBmatcher(0.6, false), "
                   "bOnlyStereo false, bCoarse false; synthetic key-frame pairs, <= 100 shared vocabulary nodes; one call = host "
                   "arrays in, pair list out (rgbl_search_triangulation); cpu_reference = the reference's own function in "
                   "oracle/_ref/libref_orbmatcher.so (-O2, 1 thread), timed inside the glue around the call itself"}
    m = F.ORBmatcher(0.6, False, lib=lib)
    q1, t1 = np.array([0, 0, 0, 1], np.float32), np.zeros(3, np.float32)
    ang = 0.03
    q2 = np.array([0, np.sin(ang / 2), 0, np.cos(ang / 2)], np.float32)
    t2 = np.array([-0.3, 0.01, 1.0], np.float32)
    for n in sizes:
        kf1, kf2, K, R12, t12, ep, sf, s2 = cases.make_triangulation_case(n, seed=11 + n, n_nodes=100)
        ref_stat = None
        if ref is not None:
            (rm, rnm, R12, t12, ep, _), ref_stat = _ref_stats(lambda: R.search_triangulation(ref, kf1, kf2, K, sf, s2, q1, t1, q2, t2, False, False, False))
        Fm = m.fundamental(K, K, R12, t12)
        call = m.prepare_SearchForTriangulation(kf1, kf2, Fm, ep, sf, s2, False, False)
        pairs, nm, m12 = call()
        om, onm = O.search_triangulation(kf1, kf2, Fm, ep, sf, s2, False, False, False)
        ok = bool(nm == onm and np.array_equal(m12, om) and (ref is None or (rnm == nm and np.array_equal(rm, m12))))
        gpu = _time_call(call)
        kern, ktot = _kernel_us(m, call)
        # both key frames resident (descriptors, keypoints, uRight, FeatureVector): only the map-point masks and the node pairs go up
        d1 = F.DeviceFrame(n, lib=lib).upload(kf1["desc"], kf1["xy"], kf1["octave"], kf1["uright"]).set_feature_vector(kf1["node_off"], kf1["node_feat"])
        d2 = F.DeviceFrame(n, lib=lib).upload(kf2["desc"], kf2["xy"], kf2["octave"], kf2["uright"]).set_feature_vector(kf2["node_off"], kf2["node_feat"])
        rcall = m.prepare_SearchForTriangulation(dict(kf1, device=d1), dict(kf2, device=d2), Fm, ep, sf, s2, False, False)
        _, rnm2, rm12 = rcall()
        ok = ok and rnm2 == nm and np.array_equal(rm12, om)
        resident = _time_call(rcall)
        d1.close()
        d2.close()
        res["n%d" % n] = _entry(gpu, kern, ktot, ref_stat, "bit-exact vs oracle%s" % (" and the reference's own ORBmatcher.cc" if ref is not None else "") if ok else "MISMATCH",
                                resident=resident, n1=n, n2=n, pairs=int(nm), shared_nodes=int(len(np.intersect1d(kf1["node_id"], kf2["node_id"]))))
    m.close()
    return res


def tracking_leg(lib):
    from oracle import oracle_py as O
    from oracle import ref_py as R
    from orb_slam3_rgbl_amd import cases, synth
    from orb_slam3_rgbl_amd import frontend as F
    ref = R.load_matcher()
    res = {"what": "the matcher calls of one tracked RGB-L / stereo frame through the host-pointer C ABI, one KITTI-size frame (2000 "
                   "features: keypoints / descriptors / uRight from this library's own extraction + LiDAR depth of synthetic frames) per "
                   "call; wall time of the C call incl. PCIe and the final synchronisation; cpu_reference = the reference's own "
                   "source (oracle/_ref, -O2, 1 thread) on the same input, timed around the reference function itself"}
    fr = real_frames(lib, 2)
    prev, cur = fr[0], fr[1]
    # -- SearchByProjection(CurrentFrame, LastFrame, th, bMono): Tracking::TrackWithMotionModel, RGB-L: th = 15, bMono = false
    case = cases.make_projection_case(n1=len(prev["xy"]), seed=21, motion="forward", frame2=cur)
    mt = F.ORBmatcher(0.9, True, lib=lib)
    call = mt.prepare_SearchByProjection(case, 15.0, False)
    m2, nm = call()
    om, onm = O.search_by_projection(case, 15.0, False, True)
    ok = nm == onm and np.array_equal(m2, om)
    ref_stat = None
    if ref is not None:
        keep = []
        P = O.make_projection_input(case, 15.0, False, True, keep)
        (rm, rnm, _), ref_stat = _ref_stats(lambda: R.call_struct(ref, "ref_search_by_projection", P, P.n2))
        ok = ok and rnm == nm and np.array_equal(rm, m2)
    gpu = _time_call(call)
    kern, ktot = _kernel_us(mt, call)
    rcall = mt.prepare_SearchByProjection(dict(case, device2=cur["resident"]), 15.0, False)
    rm2, rnm2 = rcall()
    ok = ok and rnm2 == nm and np.array_equal(rm2, om)
    resident = _time_call(rcall)
    res["search_by_projection"] = _entry(gpu, kern, ktot, ref_stat, "bit-exact" if ok else "MISMATCH", resident=resident, n1=int(len(case["valid1"])),
                                         n2=int(len(cur["xy"])), matches=int(nm), ref_lines="ORBmatcher.cc:1676-1887, Tracking.cc:2917-2934")
    mt.close()
    # -- SearchByProjection(F, vpMapPoints, th): Tracking::SearchLocalPoints, ORBmatcher(0.8), th = 1
    case = cases.make_local_points_case(n1=3000, seed=41, frame2=cur)
    mt = F.ORBmatcher(0.8, True, lib=lib)
    call = mt.prepare_SearchLocalPoints(case, 1.0)
    m2, nm = call()
    om, onm = O.search_local_points(case, 1.0, 0.8)
    ok = nm == onm and np.array_equal(m2, om)
    ref_stat = None
    if ref is not None:
        keep = []
        P = O.make_local_points_input(case, 1.0, 0.8, keep)
        (rm, rnm, _), ref_stat = _ref_stats(lambda: R.call_struct(ref, "ref_search_local_points", P, P.n2))
        ok = ok and rnm == nm and np.array_equal(rm, m2)
    gpu = _time_call(call)
    kern, ktot = _kernel_us(mt, call)
    rcall = mt.prepare_SearchLocalPoints(dict(case, device2=cur["resident"]), 1.0)
    rm2, rnm2 = rcall()
    ok = ok and rnm2 == nm and np.array_equal(rm2, om)
    resident = _time_call(rcall)
    res["search_local_points"] = _entry(gpu, kern, ktot, ref_stat, "bit-exact" if ok else "MISMATCH", resident=resident, n1=3000, n2=int(len(cur["xy"])),
                                        matches=int(nm), ref_lines="ORBmatcher.cc:43-213, Tracking.cc:3370-3450")
    mt.close()
    # -- ORBVocabulary::transform (Frame::ComputeBoW): synthetic ORBvoc-shaped tree, k = 10, L = 5 (ORBvoc.txt: L = 6, absent)
    voc = synth.make_vocabulary(10, 5, 3)
    varr = synth.vocabulary_arrays(voc)
    V = F.ORBVocabulary(lib=lib).from_arrays(varr)
    levelsup = 3      # FeatureVector nodes at tree level 2 (<= 100 nodes), as levelsup = 4 gives on ORBvoc's 6 levels
    tcall = V.prepare_transform(cur["desc"], levelsup)
    got = tcall()
    want = O.bow_transform(varr, cur["desc"], levelsup)
    ok = all(np.array_equal(g.view(np.uint64) if g.dtype == np.float64 else g, w.view(np.uint64) if w.dtype == np.float64 else w)
             for g, w in zip(got, want))
    ref_stat = None
    rd = R.load_dbow2()
    if rd is not None:
        path = os.path.join(tempfile.mkdtemp(prefix="rgbl_voc_"), "voc.txt")
        synth.write_vocabulary_text(path, voc)
        hv = rd.ref_voc_load_text(path.encode())
        os.remove(path)
        if hv:
            (rgot, _), ref_stat = _ref_stats(lambda: R.voc_transform(rd, hv, cur["desc"], levelsup))
            ok = ok and all(np.array_equal(np.asarray(g).view(np.uint64) if g.dtype == np.float64 else g,
                                           np.asarray(w).view(np.uint64) if w.dtype == np.float64 else w) for g, w in zip(rgot, got))
            rd.ref_voc_destroy(hv)
    gpu = _time_call(tcall)
    rcall = V.prepare_transform_frame(cur["resident"], levelsup)
    ok = ok and all(np.array_equal(g.view(np.uint64) if g.dtype == np.float64 else g, w.view(np.uint64) if w.dtype == np.float64 else w)
                    for g, w in zip(rcall(), want))
    resident = _time_call(rcall)
    res["bow_transform"] = _entry(gpu, None, None, ref_stat, "bit-exact" if ok else "MISMATCH", resident=resident, features=int(len(cur["desc"])),
                                  vocabulary="synthetic k=10 L=5 (%d nodes), levelsup 3; ORBvoc.txt (k=10 L=6) is not in the image" % varr["n_nodes"],
                                  ref_lines="Frame.cc:828-835, TemplatedVocabulary.h:1127-1255")
    # -- SearchByBoW(pKF, F, vpMapPointMatches): Tracking::TrackReferenceKeyFrame, ORBmatcher(0.7, true)
    rng = np.random.default_rng(5)

    def with_fv(f, has_mp):
        _, _, nid, noff, nfeat = V.transform(f["desc"], levelsup)
        f = {k: v for k, v in f.items() if k != "resident"}
        return dict(f, has_mp=has_mp, node_id=nid.astype(np.int32), node_off=noff.astype(np.int32), node_feat=nfeat.astype(np.int32))
    kf = with_fv(prev, (rng.random(len(prev["xy"])) < 0.7).astype(np.uint8))
    frm = with_fv(cur, np.zeros(len(cur["xy"]), np.uint8))
    mt = F.ORBmatcher(0.7, True, lib=lib)
    call = mt.prepare_SearchByBoW(kf, frm)
    m2, nm = call()
    om, onm = O.search_by_bow(kf, frm, 0.7, True)
    ok = nm == onm and np.array_equal(m2, om)
    ref_stat = None
    if ref is not None:
        (rm, rnm, _), ref_stat = _ref_stats(lambda: R.search_by_bow(ref, kf, frm, 0.7, True))
        ok = ok and rnm == nm and np.array_equal(rm, m2)
    gpu = _time_call(call)
    kern, ktot = _kernel_us(mt, call)
    prev["resident"].set_feature_vector(kf["node_off"], kf["node_feat"])
    cur["resident"].set_feature_vector(frm["node_off"], frm["node_feat"])
    rcall = mt.prepare_SearchByBoW(dict(kf, device=prev["resident"]), dict(frm, device=cur["resident"]))
    rm2, rnm2 = rcall()
    ok = ok and rnm2 == nm and np.array_equal(rm2, om)
    resident = _time_call(rcall)
    res["search_by_bow"] = _entry(gpu, kern, ktot, ref_stat, "bit-exact" if ok else "MISMATCH", resident=resident, n1=int(len(prev["xy"])), n2=int(len(cur["xy"])),
                                  matches=int(nm), nodes=int(len(kf["node_id"])),
                                  largest_bucket={"key_frame_features_with_a_map_point": int(max(kf["has_mp"][kf["node_feat"][a:b]].sum() for a, b in zip(kf["node_off"][:-1], kf["node_off"][1:]))),
                                                  "key_frame": int(np.diff(kf["node_off"]).max()), "frame": int(np.diff(frm["node_off"]).max())},
                                  ref_lines="ORBmatcher.cc:223-425, Tracking.cc:2798-2810")
    mt.close()
    V.close()
    # -- Frame::ComputeStereoMatches, one stereo pair per call (the extractions are not part of the timed call)
    w, h = synth.KITTI_W, synth.KITTI_H
    left, right = stereo_pair(40, w, h)
    exl, exr = F.ORBextractor(2000, 1.2, 8, 20, 7, w, h, lib=lib), F.ORBextractor(2000, 1.2, 8, 20, 7, w, h, lib=lib)
    kl, dl, _ = exl(left)
    kr, dr, _ = exr(right)
    call = F.prepare_ComputeStereoMatches(exl, exr, kl, dl, kr, dr, 0.54, 386.1448)
    ur, dp = call()
    ol, orr = O.Extractor(2000, 1.2, 8, 20, 7), O.Extractor(2000, 1.2, 8, 20, 7)
    okl, odl, _ = ol(left)
    okr, odr, _ = orr(right)
    our, odp = O.stereo_matches(ol, orr, okl, odl, okr, odr, 0.54, 386.1448)
    ok = np.array_equal(ur.view(np.uint32), our.view(np.uint32)) and np.array_equal(dp.view(np.uint32), odp.view(np.uint32))
    ref_stat = None
    rf = R.load_frame()
    if rf is not None:
        (rur, rdp, _, _, _), ref_stat = _ref_stats(lambda: R.stereo_matches(rf, left, right, 2000, 20, 7, 0.54, 386.1448), n=3)
        ok = ok and np.array_equal(rur.view(np.uint32), ur.view(np.uint32))
    gpu = _time_call(call)
    kern, ktot = _kernel_us(exl, call)
    res["stereo_matches"] = _entry(gpu, {k: v for k, v in kern.items() if k.startswith("k_stereo")},
                                   round(sum(v for k, v in kern.items() if k.startswith("k_stereo")), 1), ref_stat,
                                   "bit-exact" if ok else "MISMATCH", n_left=int(len(kl)), n_right=int(len(kr)), matched=int((ur >= 0).sum()),
                                   ref_lines="Frame.cc:901-1071")
    exl.close()
    exr.close()
    for f in fr:
        f["resident"].close()
    return res


def mapping_leg(lib):
    """The matcher calls of the OTHER threads - LocalMapping (Fuse), relocalisation, loop closing (SearchByBoW between key frames),
    monocular initialisation - measured the same way; they complete the list: every search routine of ORBmatcher has its latency
    in the line.  (The reference's Fuse also does the Replace / AddObservation bookkeeping the drop-in class does on the host;
    its time is the whole function's.)"""
    from oracle import oracle_py as O
    from oracle import ref_py as R
    from orb_slam3_rgbl_amd import cases
    from orb_slam3_rgbl_amd import frontend as F
    ref = R.load_matcher()
    res = {"what": "ORBmatcher calls of LocalMapping / Relocalization / LoopClosing / MonocularInitialization through the host-pointer C ABI, "
                   "synthetic KITTI-size inputs (orb_slam3_rgbl_amd/cases.py), measured like extra.tracking_calls"}
    # -- Fuse(pKF, vpMapPoints, 3.0): LocalMapping::SearchInNeighbors (LocalMapping.cc:737-879), twice per neighbour key frame
    case = cases.make_fuse_case(2500, 2000, 91)
    valid, level = cases.fuse_prepass(case)
    mt = F.ORBmatcher(0.6, True, lib=lib)
    call = mt.prepare_FuseSearch(dict(case, valid1=valid, level1=level), 3.0)
    best, dist = call()
    obest, on = O.fuse_search(case, 3.0)
    ok = np.array_equal(best, obest)
    ref_stat = None
    if ref is not None:
        keep = []
        P = O.make_fuse_input(case, 3.0, keep)
        state = np.random.default_rng(91).choice([0, 1, 2, 3], len(case["kp2_xy"]), p=[0.5, 0.2, 0.2, 0.1]).astype(np.uint8)
        (rb, rnf, _), ref_stat = _ref_stats(lambda: R.fuse(ref, P, state))
        seen = rb >= 0
        ok = ok and rnf == on and np.array_equal(rb[seen], obest[seen])
    gpu = _time_call(call)
    kern, ktot = _kernel_us(mt, call)
    res["fuse_search"] = _entry(gpu, kern, ktot, ref_stat, "bit-exact" if ok else "MISMATCH", n1=2500, n2=2000, fused=int(on),
                                ref_lines="ORBmatcher.cc:1148-1338, LocalMapping.cc:737-879")
    mt.close()
    # -- SearchByProjection(CurrentFrame, pKF, sAlreadyFound, 10, 100): Tracking::Relocalization (Tracking.cc:3723-3752)
    case = cases.make_relocalization_case(1500, 2000, 61)
    valid, level = cases.relocalization_prepass(case)
    mt = F.ORBmatcher(0.9, True, lib=lib)
    call = mt.prepare_SearchByProjectionKeyFrame(dict(case, valid1=valid, level1=level), 10.0, 100)
    m2, nm = call()
    om, onm = O.search_by_projection_kf(case, 10.0, 100, True)
    ok = nm == onm and np.array_equal(m2, om)
    ref_stat = None
    if ref is not None:
        keep = []
        P = O.make_kf_projection_input(case, 10.0, 100, True, keep)
        (rm, rnm, _), ref_stat = _ref_stats(lambda: R.call_struct(ref, "ref_search_by_projection_kf", P, P.n2))
        ok = ok and rnm == nm and np.array_equal(rm, m2)
    gpu = _time_call(call)
    kern, ktot = _kernel_us(mt, call)
    res["search_by_projection_keyframe"] = _entry(gpu, kern, ktot, ref_stat, "bit-exact" if ok else "MISMATCH", n1=1500, n2=2000, matches=int(nm),
                                                  ref_lines="ORBmatcher.cc:1889-2010, Tracking.cc:3723-3752")
    mt.close()
    # -- SearchByBoW(pKF1, pKF2, vpMatches12): LoopClosing (ORBmatcher(0.75 / 0.9, true))
    kf1, kf2, *_ = cases.make_triangulation_case(2000, seed=81, n_nodes=100)
    rng = np.random.default_rng(81)
    kf1 = dict(kf1, has_mp=(rng.random(2000) < 0.8).astype(np.uint8))
    kf2 = dict(kf2, has_mp=(rng.random(2000) < 0.8).astype(np.uint8))
    mt = F.ORBmatcher(0.75, True, lib=lib)
    call = mt.prepare_SearchByBoWKeyFrames(kf1, kf2)
    m12, nm = call()
    om, onm = O.search_by_bow_kf(kf1, kf2, 0.75, True)
    ok = nm == onm and np.array_equal(m12, om)
    ref_stat = None
    if ref is not None:
        (rm, rnm, _), ref_stat = _ref_stats(lambda: R.search_by_bow_kf(ref, kf1, kf2, 0.75, True))
        ok = ok and rnm == nm and np.array_equal(rm, m12)
    gpu = _time_call(call)
    kern, ktot = _kernel_us(mt, call)
    res["search_by_bow_keyframes"] = _entry(gpu, kern, ktot, ref_stat, "bit-exact" if ok else "MISMATCH", n1=2000, n2=2000, matches=int(nm),
                                            ref_lines="ORBmatcher.cc:765-905, LoopClosing.cc")
    mt.close()
    # -- SearchByBoW(pKF, F) on a two-camera frame (F.Nleft != -1: the fisheye rig's TrackReferenceKeyFrame / Relocalization)
    kf, frm, n_left = cases.make_bow_rig_case(2000, 61, 100)
    kf = dict(kf, has_mp=(np.random.default_rng(61).random(2000) < 0.7).astype(np.uint8))
    mt = F.ORBmatcher(0.7, True, lib=lib)
    call = mt.prepare_SearchByBoW(kf, frm, n_left)
    m2, nm = call()
    om, onm = O.search_by_bow(kf, frm, 0.7, True, n_left=n_left)
    ok = nm == onm and np.array_equal(m2, om)
    ref_stat = None
    if ref is not None:
        (rm, rnm, _), ref_stat = _ref_stats(lambda: R.search_by_bow_rig(ref, kf, -1, frm, n_left, 0.7, True))
        ok = ok and rnm == nm and np.array_equal(rm, m2)
    gpu = _time_call(call)
    kern, ktot = _kernel_us(mt, call)
    res["search_by_bow_two_camera_frame"] = _entry(gpu, kern, ktot, ref_stat, "bit-exact" if ok else "MISMATCH", n1=2000, n2=int(len(frm["desc"])), n_left=int(n_left),
                                                   matches=int(nm), right_camera_matches=int((m2[n_left:] >= 0).sum()), ref_lines="ORBmatcher.cc:298-326, 357-386")
    mt.close()
    # -- SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, 100): Tracking::MonocularInitialization (5 x nFeatures)
    case = cases.make_initialization_case(5000, 61)
    mt = F.ORBmatcher(0.9, True, lib=lib)
    call = mt.prepare_SearchForInitialization(case, 100)
    m12, prev, nm = call()
    om, oprev, onm = O.search_for_initialization(case, 100, 0.9, True)
    ok = nm == onm and np.array_equal(m12, om) and np.array_equal(prev.view(np.uint32), oprev.view(np.uint32))
    ref_stat = None
    if ref is not None:
        keep = []
        P = O.make_initialization_input(case, 100, 0.9, True, keep)
        (rm, rprev, rnm, _), ref_stat = _ref_stats(lambda: R.search_for_initialization(ref, P, case["prev_matched"]))
        ok = ok and rnm == nm and np.array_equal(rm, m12)
    gpu = _time_call(call)
    kern, ktot = _kernel_us(mt, call)
    res["search_for_initialization"] = _entry(gpu, kern, ktot, ref_stat, "bit-exact" if ok else "MISMATCH", n1=5000, n2=int(len(case["kp2_xy"])),
                                              matches=int(nm), ref_lines="ORBmatcher.cc:648-763, Tracking.cc:2526")
    mt.close()
    return res


def stereo_pair(seq, w, h, frame=0, disparity_scale=1.0):
    """A rectified synthetic stereo pair (the generator of tests/parity_checks.stereo_pair): the right view is the scene shifted
    horizontally by a disparity that grows towards the bottom of the image, with its own sensor noise."""
    from orb_slam3_rgbl_amd import synth
    sq = synth.Sequence(seq, w + 128, h, n_frames=frame + 1)
    full = sq.frame(frame).astype(np.int32)
    left = full[:, 64:64 + w]
    right = np.empty_like(left)
    for y in range(h):
        d = int(round(disparity_scale * (4 + 36.0 * y / h)))
        right[y] = full[y, 64 + d:64 + d + w]
    rng = np.random.default_rng(seq * 77 + frame)
    right = np.clip(right + np.rint(1.5 * rng.standard_normal(right.shape)).astype(np.int32), 0, 255)
    return np.ascontiguousarray(left.astype(np.uint8)), np.ascontiguousarray(right.astype(np.uint8))


def run(lib):
    return {"cfg3_triangulation": triangulation_leg(lib), "tracking_calls": tracking_leg(lib), "mapping_calls": mapping_leg(lib)}


if __name__ == "__main__":
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from orb_slam3_rgbl_amd import _lib
    print(json.dumps(run(_lib.load())))
