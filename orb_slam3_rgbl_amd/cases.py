"""Synthetic inputs for the matcher entry points: key-frame pairs with FeatureVectors (SearchForTriangulation, SearchByBoW),
a LastFrame / CurrentFrame pair with map points (SearchByProjection), local map points against a frame
(Tracking::SearchLocalPoints).  Shared by the parity tests (tests/parity_checks.py) and bench.py's per-call latency legs,
so that what is timed is what is checked.  numpy only."""
import numpy as np

from . import frontend as F
from . import synth


def make_triangulation_case(n=1500, seed=11, n_nodes=100):
    rng = np.random.default_rng(seed)
    a = synth.descriptors(n, seed)
    b, perm = synth.perturbed_descriptors(a, flip_p=0.04, seed=seed + 1)
    # a few exact duplicates in image 2 to exercise "ties -> later candidate wins"
    dup = rng.integers(0, n, 40)
    b[dup] = b[(dup + 1) % n]
    key1 = (a[:, 0].astype(np.int64) * 131 + a[:, 1] * 31 + a[:, 2]) % n_nodes
    key2 = key1[perm]  # a feature of image 2 falls into the node of the feature it was derived from
    key2[dup] = key2[(dup + 1) % n]

    def csr(key):
        order = np.argsort(key, kind="stable").astype(np.int32)
        ids, cnt = np.unique(key, return_counts=True)
        off = np.zeros(len(ids) + 1, np.int32)
        off[1:] = np.cumsum(cnt)
        return ids.astype(np.int32), off, order

    xy1 = np.stack([rng.uniform(20, 1200, n), rng.uniform(20, 350, n)], 1).astype(np.float32)
    disp = rng.uniform(2, 60, n).astype(np.float32)
    xy2 = xy1[perm].copy()
    xy2[:, 0] -= disp[perm]
    xy2[:, 1] += rng.normal(0, 1.2, n).astype(np.float32)  # some pairs violate the epipolar bound
    oct1 = rng.integers(0, 8, n).astype(np.int32)
    oct2 = rng.integers(0, 8, n).astype(np.int32)
    ang1 = rng.uniform(0, 360, n).astype(np.float32)
    ang2 = (ang1[perm] + rng.normal(0, 25, n)).astype(np.float32) % 360
    ur1 = np.where(rng.random(n) < 0.5, xy1[:, 0] - 5, -1).astype(np.float32)
    ur2 = np.where(rng.random(n) < 0.5, xy2[:, 0] - 5, -1).astype(np.float32)
    mp1 = (rng.random(n) < 0.3).astype(np.uint8)
    mp2 = (rng.random(n) < 0.3).astype(np.uint8)
    id1, off1, f1 = csr(key1)
    id2, off2, f2 = csr(key2)
    # drop a few nodes from each side so the merge walk has to skip
    keep1 = rng.random(len(id1)) < 0.9
    keep2 = rng.random(len(id2)) < 0.9

    def drop(ids, off, feat, keep):
        nid, noff, nfeat = [], [0], []
        for i, k in enumerate(keep):
            if k:
                nid.append(ids[i])
                nfeat.extend(feat[off[i]:off[i + 1]])
                noff.append(len(nfeat))
        return np.array(nid, np.int32), np.array(noff, np.int32), np.array(nfeat, np.int32)

    id1, off1, f1 = drop(id1, off1, f1, keep1)
    id2, off2, f2 = drop(id2, off2, f2, keep2)
    kf1 = dict(desc=a, xy=xy1, octave=oct1, angle=ang1, uright=ur1, has_mp=mp1, node_id=id1, node_off=off1, node_feat=f1)
    kf2 = dict(desc=b, xy=xy2, octave=oct2, angle=ang2, uright=ur2, has_mp=mp2, node_id=id2, node_off=off2, node_feat=f2)
    K = np.array([718.856, 718.856, 607.1928, 185.2157], np.float32)
    R = np.eye(3, dtype=np.float32).reshape(9)
    t = np.array([-0.54, 0.002, 0.001], np.float32)
    ep = np.array([900.0, 185.0], np.float32)  # inside the image so the epipole guard rejects some pairs
    sf = (1.2 ** np.arange(8)).astype(np.float32)
    return kf1, kf2, K, R, t, ep, sf, (sf * sf).astype(np.float32)


def make_bow_rig_case(n=1500, seed=11, n_nodes=100):
    """A key frame and a TWO-CAMERA frame (F.Nleft != -1) for SearchByBoW: the frame of make_triangulation_case as the left
    camera's features and a second, more strongly perturbed and shuffled copy of them as the right camera's; a right feature sits
    in the vocabulary node of the left feature it was derived from.  Returns (kf, frame, Nleft)."""
    kf, left, *_ = make_triangulation_case(n, seed, n_nodes)
    rng = np.random.default_rng(seed + 1000)
    n2 = len(left["desc"])
    right_desc, perm = synth.perturbed_descriptors(left["desc"], flip_p=0.05, seed=seed + 3)
    node_of = np.full(n2, -1, np.int64)   # position of the left feature's node in the frame's FeatureVector, -1: in none
    for k in range(len(left["node_id"])):
        node_of[left["node_feat"][left["node_off"][k]:left["node_off"][k + 1]]] = k
    node_r = node_of[perm]
    feat, off = [], [0]
    for k in range(len(left["node_id"])):
        feat.extend(left["node_feat"][left["node_off"][k]:left["node_off"][k + 1]])
        feat.extend((np.nonzero(node_r == k)[0] + n2).tolist())   # ascending inside a node, as DBoW2 fills it
        off.append(len(feat))
    ang_r = (left["angle"][perm] + rng.normal(0, 10, n2)).astype(np.float32) % 360
    frame = dict(desc=np.concatenate([left["desc"], right_desc]), xy=np.concatenate([left["xy"], left["xy"][perm]]),
                 octave=np.concatenate([left["octave"], left["octave"][perm]]), angle=np.concatenate([left["angle"], ang_r]),
                 uright=np.full(2 * n2, -1, np.float32), has_mp=np.zeros(2 * n2, np.uint8), node_id=left["node_id"],
                 node_off=np.array(off, np.int32), node_feat=np.array(feat, np.int32))
    return kf, frame, n2


def _quat(axis, angle):
    axis = np.asarray(axis, np.float64)
    axis = axis / np.linalg.norm(axis)
    return np.concatenate([axis * np.sin(angle / 2), [np.cos(angle / 2)]]).astype(np.float32)


def _rot(q):
    x, y, z, w = [float(v) for v in q]
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def make_projection_case(n1=1800, n2=2000, seed=21, motion="forward", w=synth.KITTI_W, h=synth.KITTI_H, frame2=None):
    """A LastFrame with map points and a CurrentFrame whose features they should re-find.  Clusters of near-identical
    current features and several map points aiming at the same feature exercise the 'feature already holds an observed
    map point' rule (later points fall back to their second choice); unobserved (temporal) points get overwritten.
    frame2 (optional): dict(xy, desc, octave, angle[, uright]) of a real extraction to use as the CurrentFrame instead of
    the synthetic features (n2 = its size; no clusters are added)."""
    rng = np.random.default_rng(seed)
    K = np.array([718.856, 718.856, 607.1928, 185.2157], np.float32)
    mbf = 386.1448
    if frame2 is not None:
        xy2 = np.ascontiguousarray(frame2["xy"], np.float32)
        n2 = len(xy2)
        desc2 = np.ascontiguousarray(frame2["desc"], np.uint8)
        oct2 = np.ascontiguousarray(frame2["octave"], np.int32)
        ang2 = np.ascontiguousarray(frame2["angle"], np.float32)
        depth2 = rng.uniform(4, 70, n2)
        if frame2.get("uright") is not None:
            uright2 = np.ascontiguousarray(frame2["uright"], np.float32)
            depth2 = np.where(uright2 > 0, mbf / np.maximum(xy2[:, 0] - uright2, 1e-3), depth2)
        else:
            uright2 = np.where(rng.random(n2) < 0.6, xy2[:, 0] - mbf / depth2 + rng.normal(0, 1.0, n2), -1).astype(np.float32)
    else:
        xy2 = np.stack([rng.uniform(5, w - 5, n2), rng.uniform(5, h - 5, n2)], 1).astype(np.float32)
        desc2 = synth.descriptors(n2, seed)
        # clusters: copies of a feature a few pixels away with almost the same descriptor
        ncl = n2 // 10
        src = rng.integers(0, n2, ncl)
        dst = rng.permutation(n2)[:ncl]
        xy2[dst] = xy2[src] + rng.uniform(-4, 4, (ncl, 2)).astype(np.float32)
        flips = rng.random((ncl, 256)) < 0.02
        desc2[dst] = desc2[src] ^ np.packbits(flips, axis=1, bitorder="little")
        xy2[:, 0] = np.clip(xy2[:, 0], 1, w - 2)
        xy2[:, 1] = np.clip(xy2[:, 1], 1, h - 2)
        oct2 = rng.integers(0, 8, n2).astype(np.int32)
        oct2[dst] = oct2[src]
        ang2 = rng.uniform(0, 360, n2).astype(np.float32)
        depth2 = rng.uniform(4, 70, n2)
        uright2 = np.where(rng.random(n2) < 0.6, xy2[:, 0] - mbf / depth2 + rng.normal(0, 1.0, n2), -1).astype(np.float32)
    # poses
    qc = _quat([0.1, 1.0, 0.05], 0.02)
    tc = np.array([0.05, -0.02, 0.3], np.float32)
    dz = {"forward": 0.9, "backward": -0.9, "none": 0.05}[motion]
    ql = _quat([0.0, 1.0, 0.0], 0.01)
    Rc, Rl = _rot(qc), _rot(ql)
    Cc = -Rc.T @ tc.astype(np.float64)                       # current camera centre in the world
    # last camera: dz metres behind (forward motion) along its own optical axis
    tl = (-Rl @ Cc + np.array([0.02, 0.0, dz])).astype(np.float32)
    # map points: most aim at a current feature (several at the same one), the rest are elsewhere
    target = rng.integers(0, n2, n1)
    target[: n1 // 6] = target[n1 // 6: 2 * (n1 // 6)]        # duplicates
    aimed = rng.random(n1) < 0.8
    z = depth2[target] * rng.uniform(0.97, 1.03, n1)
    uv = xy2[target] + rng.normal(0, 1.5, (n1, 2))
    uv[~aimed] = np.stack([rng.uniform(-50, w + 50, (~aimed).sum()), rng.uniform(-50, h + 50, (~aimed).sum())], 1)
    z[rng.random(n1) < 0.03] *= -1                            # behind the camera
    xc = np.stack([(uv[:, 0] - K[2]) / K[0] * z, (uv[:, 1] - K[3]) / K[1] * z, z], 1)
    world = ((xc - tc.astype(np.float64)) @ Rc).astype(np.float32)   # Rc^T (xc - tc)
    flips = rng.random((n1, 256)) < 0.03
    mp_desc = desc2[target] ^ np.packbits(flips, axis=1, bitorder="little")
    mp_desc[~aimed] = synth.descriptors(int((~aimed).sum()), seed + 1)
    oct1 = np.clip(oct2[target] + rng.integers(-1, 2, n1), 0, 7).astype(np.int32)
    ang1 = ((ang2[target] + rng.normal(0, 20, n1)) % 360).astype(np.float32)
    ang1[rng.random(n1) < 0.1] = rng.uniform(0, 360, 1).astype(np.float32)[0]
    gw, gh = np.float32(w), np.float32(h)
    grid = np.array([0, 0, gw, gh, np.float32(64) / (gw - np.float32(0)), np.float32(48) / (gh - np.float32(0))], np.float32)
    return dict(valid1=(rng.random(n1) < 0.9).astype(np.uint8), world_pos1=world, mp_desc1=mp_desc,
                mp_observed1=(rng.random(n1) < 0.75).astype(np.uint8), octave1=oct1, angle1=ang1,
                kp2_xy=xy2, kp2_octave=oct2, kp2_angle=ang2, uright2=uright2, desc2=desc2, grid=grid,
                Tcw_q=qc, Tcw_t=tc, Tlw_q=ql, Tlw_t=tl, K=K, mb=0.54, mbf=mbf,
                scale_factors=(1.2 ** np.arange(8)).astype(np.float32))


def make_local_points_case(n1=3000, n2=2000, seed=41, w=synth.KITTI_W, h=synth.KITTI_H, frame2=None):
    """Local map points with predicted projections (what Frame::isInFrustum leaves in the MapPoint) against a frame in which
    a part of the features already holds tracked points.  Clusters of look-alike features make the ratio test bite,
    several map points aim at the same feature (the later one is blocked and falls back or fails its ratio test).
    frame2 (optional): dict(xy, desc, octave[, uright]) of a real extraction to use as the frame (no clusters added)."""
    rng = np.random.default_rng(seed)
    mbf = 386.1448
    if frame2 is not None:
        xy2 = np.ascontiguousarray(frame2["xy"], np.float32)
        n2 = len(xy2)
        desc2 = np.ascontiguousarray(frame2["desc"], np.uint8)
        oct2 = np.ascontiguousarray(frame2["octave"], np.int32)
        depth2 = rng.uniform(4, 70, n2)
        if frame2.get("uright") is not None:
            uright2 = np.ascontiguousarray(frame2["uright"], np.float32)
            depth2 = np.where(uright2 > 0, mbf / np.maximum(xy2[:, 0] - uright2, 1e-3), depth2)
        else:
            uright2 = np.where(rng.random(n2) < 0.6, xy2[:, 0] - mbf / depth2, -1).astype(np.float32)
    else:
        xy2 = np.stack([rng.uniform(5, w - 5, n2), rng.uniform(5, h - 5, n2)], 1).astype(np.float32)
        desc2 = synth.descriptors(n2, seed)
        ncl = n2 // 6
        src = rng.integers(0, n2, ncl)
        dst = rng.permutation(n2)[:ncl]
        xy2[dst] = xy2[src] + rng.uniform(-3, 3, (ncl, 2)).astype(np.float32)
        flips = rng.random((ncl, 256)) < rng.choice([0.01, 0.05, 0.15], ncl)[:, None]
        desc2[dst] = desc2[src] ^ np.packbits(flips, axis=1, bitorder="little")
        xy2[:, 0] = np.clip(xy2[:, 0], 1, w - 2)
        xy2[:, 1] = np.clip(xy2[:, 1], 1, h - 2)
        oct2 = rng.integers(0, 8, n2).astype(np.int32)
        oct2[dst] = np.where(rng.random(ncl) < 0.7, oct2[src], np.clip(oct2[src] - 1, 0, 7))
        depth2 = rng.uniform(4, 70, n2)
        uright2 = np.where(rng.random(n2) < 0.6, xy2[:, 0] - mbf / depth2, -1).astype(np.float32)
    blocked2 = (rng.random(n2) < 0.3).astype(np.uint8)          # already tracked by the motion model
    target = rng.integers(0, n2, n1)
    target[: n1 // 5] = target[n1 // 5: 2 * (n1 // 5)]
    aimed = rng.random(n1) < 0.8
    uv = xy2[target] + rng.normal(0, 1.0, (n1, 2)).astype(np.float32)
    uv[~aimed] = np.stack([rng.uniform(0, w, (~aimed).sum()), rng.uniform(0, h, (~aimed).sum())], 1)
    xr = (uv[:, 0] - mbf / depth2[target] + rng.normal(0, 1.5, n1)).astype(np.float32)
    proj = np.concatenate([uv, xr[:, None]], 1).astype(np.float32)
    level = np.clip(oct2[target] + rng.integers(0, 2, n1), 0, 7).astype(np.int32)   # window accepts level-1 .. level
    flips = rng.random((n1, 256)) < 0.04
    mp_desc = desc2[target] ^ np.packbits(flips, axis=1, bitorder="little")
    mp_desc[~aimed] = synth.descriptors(int((~aimed).sum()), seed + 1)
    gw, gh = np.float32(w), np.float32(h)
    grid = np.array([0, 0, gw, gh, np.float32(64) / gw, np.float32(48) / gh], np.float32)
    return dict(valid1=(rng.random(n1) < 0.85).astype(np.uint8), proj1=proj, level1=level,
                view_cos1=np.where(rng.random(n1) < 0.5, 0.9995, 0.9).astype(np.float32), mp_desc1=mp_desc,
                mp_observed1=(rng.random(n1) < 0.9).astype(np.uint8), kp2_xy=xy2, kp2_octave=oct2, uright2=uright2, desc2=desc2,
                blocked2=blocked2, grid=grid, scale_factors=(1.2 ** np.arange(8)).astype(np.float32))


def make_relocalization_case(n1=1500, n2=2000, seed=61):
    """A key frame whose map points are searched in a frame with a (PnP-refined) pose: SearchByProjection(CurrentFrame, pKF,
    sAlreadyFound, th, ORBdist).  Built on the Frame-to-Frame case: same clusters / duplicate targets; on top of it points
    without a map point, bad and already-found ones, scale-invariance ranges that exclude some points, features of the
    frame that hold a map point on entry, and points behind the camera (the overload does not test the sign of the depth)."""
    c = make_projection_case(n1, n2, seed, "none")
    rng = np.random.default_rng(seed + 1000)
    qc, tc = c["Tcw_q"], c["Tcw_t"]
    Rc = _rot(qc)
    Ow = (-Rc.T @ tc.astype(np.float64))
    dist = np.linalg.norm(c["world_pos1"].astype(np.float64) - Ow, axis=1)
    # mfMaxDistance = dist * scale^level of the observation that made the point: predicted levels spread over the pyramid
    lvl = c["octave1"].astype(np.int64)                          # near the octave of the feature the point aims at
    max_d = (dist * 1.2 ** lvl * rng.uniform(0.85, 1.0, n1)).astype(np.float32)
    min_d = (max_d / np.float32(1.2 ** 7)).astype(np.float32)
    far = rng.random(n1) < 0.05
    max_d[far] = (dist[far] * 0.5).astype(np.float32)          # outside the invariance range (too far)
    near = rng.random(n1) < 0.03
    min_d[near] = (dist[near] * 2.0).astype(np.float32)        # too close
    return dict(has_mp1=(rng.random(n1) < 0.85).astype(np.uint8), bad1=(rng.random(n1) < 0.05).astype(np.uint8),
                found1=(rng.random(n1) < 0.15).astype(np.uint8), world_pos1=c["world_pos1"], mp_desc1=c["mp_desc1"],
                min_dist1=min_d, max_dist1=max_d, angle1=c["angle1"], kp2_xy=c["kp2_xy"], kp2_octave=c["kp2_octave"],
                kp2_angle=c["kp2_angle"], desc2=c["desc2"], occupied2=(rng.random(n2) < 0.1).astype(np.uint8),
                grid=c["grid"], Tcw_q=qc, Tcw_t=tc, K=c["K"], scale_factors=c["scale_factors"],
                log_scale_factor=np.float32(np.log(np.float32(1.2))))


def relocalization_prepass(case):
    """What the shim evaluates with the MapPoint objects, written with numpy float32 + the C library's logf
    (F.ORBmatcher.PredictScale): must equal the oracle's prepass bit for bit."""
    q, t = np.asarray(case["Tcw_q"], np.float32), np.asarray(case["Tcw_t"], np.float32)

    def rotate(qv, p):  # Eigen QuaternionBase::_transformVector in float32
        u = np.float32(2) * np.cross(qv[:3], p).astype(np.float32)
        return (p + qv[3] * u + np.cross(qv[:3], u).astype(np.float32)).astype(np.float32)
    qinv = np.array([-q[0], -q[1], -q[2], q[3]], np.float32)
    Ow = rotate(qinv, (-t).astype(np.float32))
    PO = (np.asarray(case["world_pos1"], np.float32) - Ow).astype(np.float32)
    sq = (PO * PO).astype(np.float32)
    dist = np.sqrt((sq[:, 0] + (sq[:, 1] + sq[:, 2]).astype(np.float32)).astype(np.float32)).astype(np.float32)
    max_inv = (np.float32(1.2) * case["max_dist1"]).astype(np.float32)
    min_inv = (np.float32(0.8) * case["min_dist1"]).astype(np.float32)
    valid = (case["has_mp1"] != 0) & (case["bad1"] == 0) & (case["found1"] == 0) & ~(dist < min_inv) & ~(dist > max_inv)
    level = F.ORBmatcher.PredictScale(dist, case["max_dist1"], case["log_scale_factor"], len(case["scale_factors"]))
    return valid.astype(np.uint8), np.where(valid, level, 0).astype(np.int32)


def make_fuse_case(n1=2500, n2=2000, seed=91):
    """Map points of neighbouring key frames projected into a key frame (LocalMapping::SearchInNeighbors -> ORBmatcher::Fuse):
    built on the relocalisation case (same clusters, points behind the camera, invariance ranges) plus normals (some seen
    under more than 60 degrees), points already in the key frame, stereo / mono key-frame features (both chi-square gates)."""
    c = make_relocalization_case(n1, n2, seed)
    base = make_projection_case(n1, n2, seed, "none")
    rng = np.random.default_rng(seed + 2000)
    q, t = c["Tcw_q"], c["Tcw_t"]
    Rc = _rot(q)
    Ow = (-Rc.T @ t.astype(np.float64)).astype(np.float32)
    PO = c["world_pos1"].astype(np.float64) - Ow
    dist = np.linalg.norm(PO, axis=1, keepdims=True)
    normal = PO / np.maximum(dist, 1e-6) + rng.normal(0, 0.35, PO.shape)       # roughly towards the camera ...
    normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    flip = rng.random(n1) < 0.08
    normal[flip] *= -1                                                           # ... some from behind
    sf = c["scale_factors"]
    return dict(has_mp1=c["has_mp1"], bad1=c["bad1"], in_kf1=(rng.random(n1) < 0.1).astype(np.uint8),
                world_pos1=c["world_pos1"], normal1=normal.astype(np.float32), mp_desc1=c["mp_desc1"],
                min_dist1=c["min_dist1"], max_dist1=c["max_dist1"], kp2_xy=c["kp2_xy"], kp2_octave=c["kp2_octave"],
                uright2=base["uright2"], desc2=c["desc2"], grid=c["grid"], Tcw_q=q, Tcw_t=t, Ow=Ow, K=c["K"],
                bf=np.float32(base["mbf"]), scale_factors=sf,
                inv_level_sigma2=(np.float32(1.0) / (sf * sf).astype(np.float32)).astype(np.float32),
                log_scale_factor=c["log_scale_factor"])


def fuse_prepass(case):
    """The tests of the Fuse loop that need the MapPoint object, numpy float32 + the C library's logf."""
    Ow = np.asarray(case["Ow"], np.float32)
    PO = (np.asarray(case["world_pos1"], np.float32) - Ow).astype(np.float32)
    sq = (PO * PO).astype(np.float32)
    dist = np.sqrt((sq[:, 0] + (sq[:, 1] + sq[:, 2]).astype(np.float32)).astype(np.float32)).astype(np.float32)
    Pn = np.asarray(case["normal1"], np.float32)
    pr = (PO * Pn).astype(np.float32)
    dot = (pr[:, 0] + (pr[:, 1] + pr[:, 2]).astype(np.float32)).astype(np.float32)
    max_inv = (np.float32(1.2) * case["max_dist1"]).astype(np.float32)
    min_inv = (np.float32(0.8) * case["min_dist1"]).astype(np.float32)
    valid = ((case["has_mp1"] != 0) & (case["bad1"] == 0) & (case["in_kf1"] == 0) & ~(dist < min_inv) & ~(dist > max_inv) &
             ~(dot.astype(np.float64) < 0.5 * dist.astype(np.float64)))
    level = F.ORBmatcher.PredictScale(dist, case["max_dist1"], case["log_scale_factor"], len(case["scale_factors"]))
    return valid.astype(np.uint8), np.where(valid, level, 0).astype(np.int32)


def make_initialization_case(n1=5000, seed=61, w=synth.KITTI_W, h=synth.KITTI_H, motion=(14.0, -6.0)):
    """Two monocular frames: F2's features are F1's moved by `motion` plus noise, descriptors with a few flipped bits;
    vbPrevMatched = F1's positions (Tracking.cc:2493-2495).  Look-alike neighbours make the ratio test bite, and pairs of F1
    features aim at one F2 feature with the later one closer, so that matches are taken over (vMatchedDistance)."""
    rng = np.random.default_rng(seed)
    xy1 = np.stack([rng.uniform(5, w - 5, n1), rng.uniform(5, h - 5, n1)], 1).astype(np.float32)
    oct1 = rng.choice(8, n1, p=[0.45, 0.2, 0.1, 0.08, 0.06, 0.05, 0.03, 0.03]).astype(np.int32)
    desc1 = synth.descriptors(n1, seed)
    ang1 = rng.uniform(0, 360, n1).astype(np.float32)
    n2 = n1
    perm = rng.permutation(n1)                      # F2 feature j comes from F1 feature perm[j]
    xy2 = (xy1[perm] + np.array(motion, np.float32) + rng.normal(0, 1.5, (n2, 2))).astype(np.float32)
    oct2 = oct1[perm].copy()
    oct2[rng.random(n2) < 0.1] = 1
    rate = rng.choice([0.01, 0.04, 0.1, 0.3], n2, p=[0.3, 0.4, 0.2, 0.1])
    desc2 = desc1[perm] ^ np.packbits(rng.random((n2, 256)) < rate[:, None], axis=1, bitorder="little")
    dang = np.where(rng.random(n2) < 0.8, rng.normal(5, 3, n2), rng.uniform(0, 360, n2))
    ang2 = np.mod(ang1[perm] - dang, 360).astype(np.float32)
    # look-alike neighbours in F2 (second-best close to best)
    ncl = n2 // 8
    src = rng.integers(0, n2, ncl)
    dst = rng.permutation(n2)[:ncl]
    xy2[dst] = xy2[src] + rng.uniform(-20, 20, (ncl, 2)).astype(np.float32)
    oct2[dst] = oct2[src]
    desc2[dst] = desc2[src] ^ np.packbits(rng.random((ncl, 256)) < rng.choice([0.01, 0.06], ncl)[:, None], axis=1, bitorder="little")
    # take-overs: F1 feature a (lower index) resembles what F1 feature b (higher index) matches even better
    inv = np.empty(n1, np.int64)
    inv[perm] = np.arange(n1)
    pairs = rng.permutation(n1)[: 2 * (n1 // 10)].reshape(-1, 2)
    a, b = pairs.min(1), pairs.max(1)
    oct1[a] = 0; oct1[b] = 0
    c = inv[b]
    oct2[c] = 0
    xy1[a] = xy1[b] + rng.uniform(-30, 30, (len(a), 2)).astype(np.float32)
    desc2[c] = desc1[b] ^ np.packbits(rng.random((len(a), 256)) < 0.02, axis=1, bitorder="little")
    desc1[a] = desc2[c] ^ np.packbits(rng.random((len(a), 256)) < rng.choice([0.03, 0.08], len(a))[:, None], axis=1, bitorder="little")
    xy2[:, 0] = np.clip(xy2[:, 0], 1, w - 2)
    xy2[:, 1] = np.clip(xy2[:, 1], 1, h - 2)
    prev = xy1.copy()
    prev[rng.random(n1) < 0.01] = np.float32(-500)   # window outside of the grid
    gw, gh = np.float32(w), np.float32(h)
    grid = np.array([0, 0, gw, gh, np.float32(64) / gw, np.float32(48) / gh], np.float32)
    return dict(kp1_octave=oct1, kp1_angle=ang1, desc1=desc1, prev_matched=prev, kp2_xy=xy2, kp2_octave=oct2, kp2_angle=ang2,
                desc2=desc2, grid=grid)
