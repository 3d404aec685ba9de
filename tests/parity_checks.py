"""Parity checks shared by the GPU tests (product library on a real MI355X, `-m gpu`) and by the CPU
tests that run the same kernel sources under the SIMT emulator.  Every check compares the HIP path,
called through the C ABI, with the CPU oracle on identical seeded inputs.  Integer / byte / index results
must be bit-exact; float results are compared bit-for-bit as well unless a tolerance is stated."""
import numpy as np

from oracle import oracle_py as O
from orb_slam3_rgbl_amd import _lib as L
from orb_slam3_rgbl_amd import frontend as F
from orb_slam3_rgbl_amd import synth
from orb_slam3_rgbl_amd.cases import (make_triangulation_case, make_projection_case, make_local_points_case, make_relocalization_case,  # noqa: F401
                                      relocalization_prepass, make_fuse_case, fuse_prepass, make_initialization_case, _quat, _rot)

KP_FIELDS = ("x", "y", "size", "angle", "response", "octave", "class_id")


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def assert_keypoints_equal(kps, okps, what=""):
    assert len(kps) == len(okps), "%s: %d keypoints vs oracle %d" % (what, len(kps), len(okps))
    for f in KP_FIELDS:
        assert np.array_equal(bits(kps[f]), bits(okps[f])), "%s: keypoint field %s differs" % (what, f)


def check_extractor(lib, w, h, nfeatures, frames=(0,), ini=12, mn=7, nlevels=8, seq=0, lapping=(0, 0), stages=False):
    ex = F.ORBextractor(nfeatures, 1.2, nlevels, ini, mn, w, h, lib=lib)
    orc = O.Extractor(nfeatures, 1.2, nlevels, ini, mn)
    t = orc.tables()
    for k, a in (("scale", ex.mvScaleFactor), ("inv_scale", ex.mvInvScaleFactor), ("sigma2", ex.mvLevelSigma2),
                 ("inv_sigma2", ex.mvInvLevelSigma2), ("per_level", ex.mnFeaturesPerLevel), ("umax", ex.umax)):
        assert np.array_equal(bits(t[k]), bits(a)), k
    s = synth.Sequence(seq, w, h, n_frames=max(frames) + 1)
    total = 0
    for fr in frames:
        img = s.frame(fr)
        kps, desc, mono = ex(img, None, lapping)
        okps, odesc, omono = orc(img, lapping)
        assert_keypoints_equal(kps, okps, "frame %d" % fr)
        assert np.array_equal(desc, odesc), "frame %d: descriptors differ" % fr
        assert mono == omono, "monoIndex %d vs %d" % (mono, omono)
        total += len(kps)
        if stages:
            for l in range(nlevels):
                assert np.array_equal(ex.image_pyramid(l), orc.level_image(l)), "pyramid level %d" % l
                assert np.array_equal(ex.image_pyramid(l, with_border=True), orc.level_bordered(l)), "border %d" % l
                if len(orc.level_keypoints(l)):
                    assert np.array_equal(ex.image_pyramid(l, blurred=True), orc.level_blurred(l)), "blur level %d" % l
                c, oc = ex.level_candidates(l), orc.level_candidates(l)
                assert len(c) == len(oc), "level %d: %d candidates vs %d" % (l, len(c), len(oc))
                for f in ("x", "y", "response"):
                    assert np.array_equal(c[f], oc[f]), "level %d candidate %s" % (l, f)
    ex.close()
    return total


def check_extractor_empty_root(lib, w=600, h=160, nfeatures=400):
    """A wide image (round(w / h) = 4 quad-tree roots) whose middle has no corners at all: the reference erases the empty root
    nodes (ORBextractor.cc:597-606), the nodes behind them move up in the list."""
    ex = F.ORBextractor(nfeatures, 1.2, 4, 20, 7, w, h, lib=lib)
    orc = O.Extractor(nfeatures, 1.2, 4, 20, 7)
    img = synth.Sequence(31, w, h, n_frames=1).frame(0).copy()
    img[:, int(0.28 * w):int(0.72 * w)] = 90   # roots 1 and 2 of 4 see a constant image
    kps, desc, mono = ex(img)
    okps, odesc, omono = orc(img)
    assert len(okps) > 50
    assert_keypoints_equal(kps, okps, "empty root")
    assert np.array_equal(desc, odesc) and mono == omono
    left = img.copy()
    left[:, :int(0.55 * w)] = 90               # the first roots empty, the last ones not
    kps, desc, mono = ex(left)
    okps, odesc, omono = orc(left)
    assert len(okps) > 20
    assert_keypoints_equal(kps, okps, "empty leading roots")
    assert np.array_equal(desc, odesc) and mono == omono
    ex.close()


def check_extractor_batch(lib, w, h, nfeatures, batch, ini=12, mn=7, seq=1):
    """The batched entry point must give, frame by frame, what the single-frame entry point gives."""
    ex = F.ORBextractor(nfeatures, 1.2, 8, ini, mn, w, h, max_batch=batch, lib=lib)
    orc = O.Extractor(nfeatures, 1.2, 8, ini, mn)
    s = synth.Sequence(seq, w, h, n_frames=batch)
    imgs = np.stack([s.frame(i) for i in range(batch)])
    res = ex.extract_batch(imgs)
    for i, (kps, desc, mono) in enumerate(res):
        okps, odesc, omono = orc(imgs[i])
        assert_keypoints_equal(kps, okps, "batch frame %d" % i)
        assert np.array_equal(desc, odesc)
        assert mono == omono
    ex.close()


def check_extractor_edge_cases(lib, w=160, h=120):
    ex = F.ORBextractor(300, 1.2, 4, 20, 7, w, h, lib=lib)
    orc = O.Extractor(300, 1.2, 4, 20, 7)
    # empty image -> -1 (ORBextractor.cc:1090-1091)
    kps, desc, mono = ex(np.zeros((0, 0), np.uint8))
    assert mono == -1 and len(kps) == 0
    # constant image -> no corners at all
    flat = np.full((h, w), 77, np.uint8)
    kps, desc, mono = ex(flat)
    assert len(kps) == 0 and mono == 0
    assert len(orc(flat)[0]) == 0
    # low-contrast texture: only the minThFAST fallback fires in most cells
    rng = np.random.default_rng(5)
    low = (100 + 9 * (rng.random((h, w)) > 0.97)).astype(np.uint8)
    kps, desc, mono = ex(low)
    okps, odesc, omono = orc(low)
    assert_keypoints_equal(kps, okps, "low contrast")
    assert np.array_equal(desc, odesc)
    # saturated checkerboard: plateaus of equal scores (strict NMS suppresses both neighbours)
    yy, xx = np.mgrid[0:h, 0:w]
    chk = (((yy // 6) + (xx // 6)) % 2 * 255).astype(np.uint8)
    kps, desc, mono = ex(chk)
    okps, odesc, omono = orc(chk)
    assert_keypoints_equal(kps, okps, "checkerboard")
    assert np.array_equal(desc, odesc)
    # white noise: more pre-screen survivors than the FAST kernel's per-wave lists hold (every pixel is then scored, both
    # polarities) and more corners than its corner lists hold (the NMS then scans all survivors); two-level noise: plateaus
    noise = rng.integers(0, 256, (h, w)).astype(np.uint8)
    salt = (rng.random((h, w)) > 0.5).astype(np.uint8) * 200 + 20
    for name, im in (("white noise", noise), ("two-level noise", salt)):
        kps, desc, mono = ex(im)
        okps, odesc, omono = orc(im)
        assert len(okps) > 100
        assert_keypoints_equal(kps, okps, name)
        assert np.array_equal(desc, odesc)
    # non-contiguous rows (stride > width)
    big = np.zeros((h, w + 37), np.uint8)
    img = synth.Sequence(9, w, h, 1).frame(0)
    big[:, :w] = img
    kps, desc, mono = ex(big[:, :w])
    okps, odesc, omono = orc(img)
    assert_keypoints_equal(kps, okps, "strided")
    assert np.array_equal(desc, odesc)
    ex.close()


def kitti_projection(lib):
    proj = F.projection_matrix(synth.KITTI_K, synth.KITTI_TR, lib)
    assert np.array_equal(bits(proj), bits(O.projection_matrix(synth.KITTI_K, synth.KITTI_TR)))
    return proj


def check_depth(lib, method, w=synth.KITTI_W, h=synth.KITTI_H, seed=0, n_az=1900, kernel=(F.KERNEL_DIAMOND, 5, 7),
                n_kp=1500, min_hits=101):
    proj = kitti_projection(lib)
    if w != synth.KITTI_W:  # rescale the principal point so that the scan still hits the image
        K = synth.KITTI_K.copy()
        K[0, 2], K[1, 2] = w / 2.0, h / 2.0
        K[0, 0] = K[1, 1] = 718.856 * w / synth.KITTI_W
        proj = F.projection_matrix(K, synth.KITTI_TR, lib)
    cloud = synth.lidar_scan(seed, n_az=n_az)
    rng = np.random.default_rng(seed + 17)
    kp_xy = np.stack([rng.uniform(19, w - 20, n_kp), rng.uniform(19, h - 20, n_kp)], 1).astype(np.float32)
    kp_xy[: n_kp // 2] = np.floor(kp_xy[: n_kp // 2])  # level-0 keypoints are integers, higher levels are not
    kpun = kp_xy.copy()
    kpun[:, 0] += rng.uniform(-1, 1, n_kp).astype(np.float32)
    shape, ku, kv = kernel
    dm = F.DepthModule(proj, w, h, method=method, kernel_type=shape, kernel_size_u=ku, kernel_size_v=kv,
                       max_points=max(cloud.shape[1], 1), max_keypoints=max(n_kp, 1), lib=lib)
    dm.CalculateDepthFromPcd(kp_xy, kpun, cloud, w, h)
    ok = O.structuring_element(shape, ku, ku if shape == F.KERNEL_DIAMOND else kv)
    P = O.make_depth_params(proj, method=method, kernel=ok)
    d, ur, raw, proc = O.depth(P, cloud, w, h, kp_xy, kpun[:, 0])
    # projection + ordered scatter: bit-exact (stated tolerance in BASELINE.md is 1e-6 rel; we demand 0)
    assert np.array_equal(bits(dm.RawDepthMap), bits(raw)), "RawDepthMap"
    assert (raw > 0).sum() >= min_hits   # the case must be worth its name (random shapes: a thin image may see a handful of points)
    if dm.ProcessedDepthMap is not None:
        a, b = dm.ProcessedDepthMap, proc
        assert np.array_equal(np.isnan(a), np.isnan(b))
        assert np.array_equal(bits(np.nan_to_num(a)), bits(np.nan_to_num(b))), "ProcessedDepthMap"
    assert np.array_equal(bits(dm.mvDepth), bits(d)), "mvDepth"
    assert np.array_equal(bits(dm.mvuRight), bits(ur)), "mvuRight"
    # the same call without the maps (what Frame needs): the inverse dilation then never materialises the raw map
    dm.CalculateDepthFromPcd(kp_xy, kpun, cloud, w, h, want_maps=False)
    assert np.array_equal(bits(dm.mvDepth), bits(d)) and np.array_equal(bits(dm.mvuRight), bits(ur)), "mvDepth without maps"
    dm.CalculateDepthFromPcd(kp_xy, kpun, cloud, w, h)  # and back: the raw map must come out clean again
    assert np.array_equal(bits(dm.RawDepthMap), bits(raw)), "RawDepthMap after a map-free call"
    n_valid = int((d > 0).sum())
    dm.close()
    return n_valid


def check_depth_edge_cases(lib, w=96, h=64):
    K = np.array([[80, 0, w / 2, 0], [0, 80, h / 2, 0], [0, 0, 1, 0]], np.float32)
    Tr = np.eye(4, dtype=np.float32)  # camera frame == lidar frame: z forward
    proj = F.projection_matrix(K, Tr, lib)
    dm = F.DepthModule(proj, w, h, max_points=64, max_keypoints=16, lib=lib)
    P = O.make_depth_params(proj)
    kp = np.array([[48, 32], [10.7, 5.2], [0.5, 0.5]], np.float32)

    def run(cloud):
        dm.CalculateDepthFromPcd(kp, kp, cloud, w, h)
        d, ur, raw, proc = O.depth(P, cloud, w, h, kp, kp[:, 0])
        assert np.array_equal(bits(dm.RawDepthMap), bits(raw))
        assert np.array_equal(bits(dm.ProcessedDepthMap), bits(proc))
        assert np.array_equal(bits(dm.mvDepth), bits(d)) and np.array_equal(bits(dm.mvuRight), bits(ur))
        return raw, d

    # three points on one pixel: the LAST one wins, not the nearest (DepthModule.cc:123-137)
    c = np.array([[0, 0, 0], [0, 0, 0], [30, 10, 20], [1, 1, 1]], np.float32)
    raw, d = run(c)
    assert raw[32, 48] == 20.0 and d[0] > 0
    # rejected: z <= min_dist, z >= max_dist, behind the camera, u exactly 0
    c = np.array([[0, 0, 0, -6.0], [0, 0, 0, 0], [5.0, 200.0, -10.0, 10.0], [1, 1, 1, 1]], np.float32)
    raw, d = run(c)
    assert (raw > 0).sum() == 0 and (d == -1).all()
    # depth > max_dist - 1 vanishes in the inverse dilation (TOZERO_INV), 199.0 survives
    c = np.array([[0, 20.0], [0, 0], [199.5, 199.0], [1, 1]], np.float32)
    raw, d = run(c)
    assert (raw > 0).sum() == 2
    # empty scan
    raw, d = run(np.zeros((4, 0), np.float32))
    assert (d == -1).all()
    # no keypoints
    dm.CalculateDepthFromPcd(np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32), c, w, h)
    assert len(dm.mvDepth) == 0
    dm.close()


def check_matcher_bf(lib, na=777, nb=700, seed=3):
    m = F.ORBmatcher(0.6, False, lib=lib)
    a = synth.descriptors(na, seed)
    b, _ = synth.perturbed_descriptors(np.resize(a, (max(nb, 1), 32)), seed=seed + 1)
    b = b[:nb]
    bi, bd, sd = m.BruteForce(a, b)
    obi, obd, osd = O.hamming_bf(a, b)
    assert np.array_equal(bi, obi) and np.array_equal(bd, obd) and np.array_equal(sd, osd)
    for i in range(0, min(na, nb), 97):
        assert m.DescriptorDistance(a[i], b[i]) == O.descriptor_distance(a[i], b[i])
    m.close()


def check_stereo_fisheye_matches(lib, n_left=900, n_right=850, mono_left=300, mono_right=260, seed=7):
    """Frame::ComputeStereoFishEyeMatches up to the triangulation: knnMatch(k = 2) + Lowe 0.7 on the
    lapping-area subsets."""
    m = F.ORBmatcher(0.6, False, lib=lib)
    a = synth.descriptors(n_left, seed)
    src = np.resize(a[mono_left:], (max(n_right - mono_right, 1), 32)) if n_left > mono_left else synth.descriptors(max(n_right - mono_right, 1), seed + 9)
    b_tail, _ = synth.perturbed_descriptors(src, seed=seed + 1)
    b = np.concatenate([synth.descriptors(mono_right, seed + 2), b_tail[:n_right - mono_right]]) if n_right else np.zeros((0, 32), np.uint8)
    got = m.StereoFishEyeMatches(a, mono_left, b, mono_right)
    want = O.stereo_fisheye_matches(a, mono_left, b, mono_right)
    for g, w_, name in zip(got, want, ("left_to_right", "best", "second")):
        assert np.array_equal(g, w_), name
    m.close()
    return int((got[0] >= 0).sum())


def check_stereo_fisheye_known_answers(lib):
    m = F.ORBmatcher(0.6, False, lib=lib)
    z = np.zeros((1, 32), np.uint8)
    def flip(k):  # descriptor with k bits set
        d = np.zeros(32, np.uint8)
        for i in range(k):
            d[i // 8] |= 1 << (i % 8)
        return d[None]
    # 7 < 10 * 0.7 is false in double arithmetic (10 * 0.7 == 7.0), 6 < 7.0 is true; indices are into the FULL right array
    right = np.concatenate([flip(200), flip(7), flip(10)])
    l2r, bd, sd = m.StereoFishEyeMatches(z, 0, right, 1)
    assert (l2r[0], bd[0], sd[0]) == (-1, 7, 10)
    right = np.concatenate([flip(200), flip(10), flip(6)])
    l2r, bd, sd = m.StereoFishEyeMatches(z, 0, right, 1)
    assert (l2r[0], bd[0], sd[0]) == (2, 6, 10)
    # the rows before mono are not matched at all; a single train row gives no second neighbour -> never a match
    l2r, bd, sd = m.StereoFishEyeMatches(np.concatenate([z, z]), 1, np.concatenate([z, z, flip(100)]), 0)
    assert list(l2r) == [-1, -1] and list(bd) == [256, 0] and list(sd) == [256, 0]   # tie 0 / 0: 0 < 0 is false
    l2r, bd, sd = m.StereoFishEyeMatches(z, 0, np.concatenate([flip(50), z]), 1)
    assert (l2r[0], bd[0], sd[0]) == (-1, 0, 256)
    l2r, bd, sd = m.StereoFishEyeMatches(z, 0, np.zeros((0, 32), np.uint8), 0)
    assert (l2r[0], bd[0], sd[0]) == (-1, 256, 256)
    m.close()


def check_matcher_known_answers(lib):
    m = F.ORBmatcher(0.6, False, lib=lib)
    z = np.zeros((1, 32), np.uint8)
    o = np.full((1, 32), 255, np.uint8)
    assert m.DescriptorDistance(z[0], o[0]) == 256 and m.DescriptorDistance(z[0], z[0]) == 0
    # duplicates in the train set: the FIRST minimum wins (strict '<'), second-best equals best
    train = np.concatenate([o, z, z, o])
    bi, bd, sd = m.BruteForce(z, train)
    assert (bi[0], bd[0], sd[0]) == (1, 0, 0)
    # single-bit flips
    for bit in (0, 7, 8, 255):
        t = z.copy()
        t[0, bit // 8] = 1 << (bit % 8)
        bi, bd, sd = m.BruteForce(t, np.concatenate([o, z]))
        assert (bi[0], bd[0], sd[0]) == (1, 1, 255)
    # equal distances in different rows of one 32-row tile, in neighbouring tiles, stages and 8192-row sweeps of the
    # matrix-core scan: the lowest index still wins and the second-best distance equals the best
    one = z.copy(); one[0, 5] = 0x10
    for first, other, n in ((3, 4, 9), (4, 3, 40), (31, 32, 70), (63, 64, 130), (40, 8200, 8300), (8191, 8192, 8200), (0, 8299, 8300)):
        train = np.repeat(o, n, 0)
        train[first] = one; train[other] = one
        bi, bd, sd = m.BruteForce(z, train)
        assert (bi[0], bd[0], sd[0]) == (min(first, other), 1, 1), (first, other, bi, bd, sd)
        train[other] = o
        bi, bd, sd = m.BruteForce(z, train)
        assert (bi[0], bd[0], sd[0]) == (first, 1, 256), (first, other, bi, bd, sd)
    # empty train set: nothing found
    bi, bd, sd = m.BruteForce(z, np.zeros((0, 32), np.uint8))
    assert (bi[0], bd[0], sd[0]) == (-1, 256, 256)
    m.close()


def check_triangulation(lib, n=1500, seed=11, n_nodes=100, min_total=101):
    kf1, kf2, K, R, t, ep, sf, s2 = make_triangulation_case(n, seed, n_nodes)
    total = 0
    for check_ori in (False, True):
        m = F.ORBmatcher(0.6, check_ori, lib=lib)
        Fm = m.fundamental(K, K, R, t)
        assert np.array_equal(bits(Fm), bits(O.fundamental(K, K, R, t)))
        for only_stereo in (False, True):
            for coarse in (False, True):
                pairs, nm, m12 = m.SearchForTriangulation(kf1, kf2, Fm, ep, sf, s2, only_stereo, coarse)
                om12, onm = O.search_triangulation(kf1, kf2, Fm, ep, sf, s2, only_stereo, coarse, check_ori)
                assert np.array_equal(m12, om12), "matches differ (ori=%s stereo=%s coarse=%s)" % (check_ori, only_stereo, coarse)
                assert nm == onm == len(pairs)
                total += nm
        m.close()
    assert total >= min_total  # the case must actually produce matches
    return total


def stereo_pair(seq, w, h, frame=0, disparity_scale=1.0):
    """A rectified synthetic stereo pair: the right view is the scene shifted horizontally (constant plus a smooth
    vertical-gradient disparity) with its own sensor noise."""
    sq = synth.Sequence(seq, w + 128, h, n_frames=frame + 1)
    full = sq.frame(frame).astype(np.int32)
    left = full[:, 64:64 + w]
    right = np.empty_like(left)
    for y in range(h):  # disparity grows towards the bottom of the image (near ground), 4 .. ~40 px
        d = int(round(disparity_scale * (4 + 36.0 * y / h)))
        right[y] = full[y, 64 + d:64 + d + w]
    rng = np.random.default_rng(seq * 77 + frame)
    right = np.clip(right + np.rint(1.5 * rng.standard_normal(right.shape)).astype(np.int32), 0, 255)
    return left.astype(np.uint8), right.astype(np.uint8)


def check_stereo_matches(lib, w=synth.KITTI_W, h=synth.KITTI_H, nfeatures=2000, ini=20, mn=7, seq=30, mb=0.54, mbf=386.1448):
    """Frame::ComputeStereoMatches: left/right extraction on the device, then matching with the resident pyramids."""
    left, right = stereo_pair(seq, w, h)
    exl = F.ORBextractor(nfeatures, 1.2, 8, ini, mn, w, h, lib=lib)
    exr = F.ORBextractor(nfeatures, 1.2, 8, ini, mn, w, h, lib=lib)
    ol, orr = O.Extractor(nfeatures, 1.2, 8, ini, mn), O.Extractor(nfeatures, 1.2, 8, ini, mn)
    kl, dl, _ = exl(left)
    kr, dr, _ = exr(right)
    okl, odl, _ = ol(left)
    okr, odr, _ = orr(right)
    assert_keypoints_equal(kl, okl, "left")
    assert_keypoints_equal(kr, okr, "right")
    ur, dp = F.ComputeStereoMatches(exl, exr, kl, dl, kr, dr, mb, mbf)
    our, odp = O.stereo_matches(ol, orr, okl, odl, okr, odr, mb, mbf)
    assert np.array_equal(bits(ur), bits(our)), "mvuRight"
    assert np.array_equal(bits(dp), bits(odp)), "mvDepth"
    n = int((odp > 0).sum())
    # no right keypoints at all / no left keypoints
    ur0, dp0 = F.ComputeStereoMatches(exl, exr, kl, dl, kr[:0], dr[:0], mb, mbf)
    assert (ur0 == -1).all() and (dp0 == -1).all()
    exl.close(); exr.close()
    return n


# ---- ingest either side of the path ------------------------------------------------------------
def color_frame(seed, w, h, channels):
    """A colour image whose channels differ (shifted / inverted copies of a synthetic gray frame plus noise)."""
    g = synth.Sequence(seed, w, h, 1).frame(0).astype(np.int32)
    rng = np.random.default_rng(seed + 99)
    img = np.empty((h, w, channels), np.uint8)
    img[..., 0] = np.clip(g + rng.integers(-20, 21, g.shape), 0, 255)
    img[..., 1] = np.clip(np.roll(g, 3, axis=1) * 3 // 4 + 30, 0, 255)
    img[..., 2] = np.clip(255 - np.roll(g, -2, axis=0) // 2 + rng.integers(-9, 10, g.shape), 0, 255)
    if channels == 4:
        img[..., 3] = rng.integers(0, 256, g.shape)  # alpha must be ignored
    return img


def check_ingest_color(lib, w, h, nfeatures=600, seed=3):
    """cvtColor + operator(): mImGray and the keypoints must equal oracle cvtColor followed by the oracle extractor."""
    ex = F.ORBextractor(nfeatures, 1.2, 8, 12, 7, w, h, lib=lib)
    orc = O.Extractor(nfeatures, 1.2, 8, 12, 7)
    n = 0
    for channels in (3, 4):
        for mbRGB in (True, False):
            img = color_frame(seed + channels, w, h, channels)
            kps, desc, mono, gray = ex.extract_color(img, mbRGB)
            ogray = O.cvt_gray(img, mbRGB)
            assert np.array_equal(gray, ogray), "mImGray (%d channels, RGB=%s)" % (channels, mbRGB)
            okps, odesc, omono = orc(ogray)
            assert_keypoints_equal(kps, okps, "colour %d/%s" % (channels, mbRGB))
            assert np.array_equal(desc, odesc) and mono == omono
            n += len(kps)
    # a strided view (ROI of a wider image) and an already-gray input
    wide = color_frame(seed, w + 13, h, 3)
    roi = wide[:, 5:5 + w]
    kps, desc, mono, gray = ex.extract_color(np.ascontiguousarray(roi), True)
    assert np.array_equal(gray, O.cvt_gray(np.ascontiguousarray(roi), True))
    g1 = synth.Sequence(seed, w, h, 1).frame(0)
    kps, desc, mono, gray = ex.extract_color(g1[..., None], True)
    okps, odesc, omono = orc(g1)
    assert np.array_equal(gray, g1) and np.array_equal(desc, odesc)
    ex.close()
    return n


def check_ingest_color_device_batch(lib, w=1241, h=376, batch=3):
    """rgbl_cvt_gray_batch_device on odd strides / widths: every byte against the oracle (needs torch + a device)."""
    import ctypes as C
    import torch
    dev = torch.device("cuda", 0)
    ex = F.ORBextractor(500, 1.2, 8, 12, 7, w, h, max_batch=batch, lib=lib)
    for channels, blue_first in ((3, 1), (3, 0), (4, 1), (4, 0)):
        imgs = np.stack([color_frame(20 + b, w, h, channels) for b in range(batch)])
        d_src = torch.from_numpy(imgs).to(dev)
        gstride = w + 3  # odd gray stride: the kernel falls back to byte stores
        d_gray = torch.zeros((batch, h, gstride), dtype=torch.uint8, device=dev)
        L.check(lib, lib.rgbl_cvt_gray_batch_device(ex.h, C.c_void_p(d_src.data_ptr()), batch, channels, blue_first, w, h,
                                                    w * channels, w * h * channels, C.c_void_p(d_gray.data_ptr()), gstride,
                                                    h * gstride))
        L.check(lib, lib.rgbl_extractor_sync(ex.h))
        got = d_gray.cpu().numpy()
        for b in range(batch):
            assert np.array_equal(got[b, :, :w], O.cvt_gray(imgs[b], not blue_first))
            assert not got[b, :, w:].any()
    ex.close()


def check_ingest_kitti_bin(lib, method=F.UPS_INVERSE_DILATION, w=620, h=188, n_az=900, n_kp=300, seed=5):
    """The .bin layout (x, y, z, reflectance) must give exactly what the 4 x N path gives on the repacked scan."""
    K = synth.KITTI_K.copy()
    K[0, 2], K[1, 2] = w / 2.0, h / 2.0
    K[0, 0] = K[1, 1] = 718.856 * w / synth.KITTI_W
    proj = F.projection_matrix(K, synth.KITTI_TR, lib)
    cloud = synth.lidar_scan(seed, n_az=n_az)
    rng = np.random.default_rng(seed)
    xyzi = np.ascontiguousarray(np.concatenate([cloud[:3].T, rng.random((cloud.shape[1], 1), np.float32)], 1))  # reflectance != 1
    assert np.array_equal(O.kitti_bin_to_cloud(xyzi), cloud)
    kp = np.stack([rng.uniform(0, w - 1, n_kp), rng.uniform(0, h - 1, n_kp)], 1).astype(np.float32)
    dm = F.DepthModule(proj, w, h, method=method, max_points=cloud.shape[1], max_keypoints=n_kp, lib=lib)
    dm.CalculateDepthFromKittiBin(kp, kp, xyzi, w, h)
    P = O.make_depth_params(proj, method=method)
    d, ur, raw, proc = O.depth(P, O.kitti_bin_to_cloud(xyzi), w, h, kp, kp[:, 0])
    assert np.array_equal(bits(dm.RawDepthMap), bits(raw))
    if dm.ProcessedDepthMap is not None:
        assert np.array_equal(bits(np.nan_to_num(dm.ProcessedDepthMap)), bits(np.nan_to_num(proc)))
    assert np.array_equal(bits(dm.mvDepth), bits(d)) and np.array_equal(bits(dm.mvuRight), bits(ur))
    n_valid = int((d > 0).sum())
    dm.close()
    return n_valid


# ---- ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) -------------------------
def check_search_by_projection(lib, seed=21, motion="forward", th=7.0, mono=False, check_ori=True, n1=1800, n2=2000):
    case = make_projection_case(n1, n2, seed, motion)
    mt = F.ORBmatcher(0.9, check_ori, lib=lib)
    m, n = mt.SearchByProjection(case, th, mono)
    om, on = O.search_by_projection(case, th, mono, check_ori)
    assert n == on and np.array_equal(m, om), "SearchByProjection (seed %d, %s, th %g)" % (seed, motion, th)
    mt.close()
    return n


def check_search_by_projection_keyframe(lib, seed=61, th=15.0, orb_dist=100, check_ori=True, n1=1500, n2=2000):
    case = make_relocalization_case(n1, n2, seed)
    valid, level = relocalization_prepass(case)
    ovalid, olevel = O.kf_projection_prepass(case)
    assert np.array_equal(valid, ovalid) and np.array_equal(level[valid != 0], olevel[valid != 0]), "relocalisation prepass"
    mt = F.ORBmatcher(0.9, check_ori, lib=lib)
    m, n = mt.SearchByProjectionKeyFrame(dict(case, valid1=valid, level1=level), th, orb_dist)
    om, on = O.search_by_projection_kf(case, th, orb_dist, check_ori)
    assert n == on and np.array_equal(m, om), "SearchByProjection(Frame, KeyFrame) (seed %d, th %g, ORBdist %d)" % (seed, th, orb_dist)
    mt.close()
    return n


def check_search_by_projection_edge_cases(lib):
    mt = F.ORBmatcher(0.9, True, lib=lib)
    # (a) nothing to match
    case = make_projection_case(50, 60, 3)
    empty = dict(case, valid1=np.zeros(50, np.uint8))
    m, n = mt.SearchByProjection(empty, 7.0, False)
    assert n == 0 and (m == -1).all()
    none1 = {k: (v[:0] if k in ("valid1", "world_pos1", "mp_desc1", "mp_observed1", "octave1", "angle1") else v) for k, v in case.items()}
    m, n = mt.SearchByProjection(none1, 7.0, False)
    assert n == 0 and (m == -1).all()
    # (b) a long blocking chain: 300 observed map points, all projecting into one 6-px cluster of 40 identical features
    rng = np.random.default_rng(9)
    case = make_projection_case(300, 400, 5, "none")
    c0 = np.array([600.0, 180.0], np.float32)
    case["kp2_xy"][:40] = c0 + rng.uniform(-3, 3, (40, 2)).astype(np.float32)
    case["desc2"][:40] = case["desc2"][0]
    case["kp2_octave"][:40] = 2
    case["uright2"][:40] = -1
    K, qc, tc = case["K"], case["Tcw_q"], case["Tcw_t"]
    z = rng.uniform(10, 20, 300)
    uv = c0 + rng.uniform(-2, 2, (300, 2))
    xc = np.stack([(uv[:, 0] - K[2]) / K[0] * z, (uv[:, 1] - K[3]) / K[1] * z, z], 1)
    case["world_pos1"] = ((xc - tc.astype(np.float64)) @ _rot(qc)).astype(np.float32)
    case["mp_desc1"] = np.repeat(case["desc2"][:1], 300, 0)
    case["octave1"] = np.full(300, 2, np.int32)
    case["valid1"] = np.ones(300, np.uint8)
    case["mp_observed1"] = (rng.random(300) < 0.9).astype(np.uint8)
    for th in (7.0, 30.0):
        m, n = mt.SearchByProjection(case, th, False)
        om, on = O.search_by_projection(case, th, False, True)
        assert n == on and np.array_equal(m, om)
    mt2 = F.ORBmatcher(0.9, False, lib=lib)
    m, n = mt2.SearchByProjection(case, 7.0, False)
    om, on = O.search_by_projection(case, 7.0, False, False)
    assert n == on and np.array_equal(m, om) and (m[:40] >= 0).sum() >= 35   # the cluster fills up one by one
    mt.close()
    mt2.close()


# ---- ORBmatcher::SearchByProjection(F, vpMapPoints, th, ...) = Tracking::SearchLocalPoints -------------------------------
def check_search_local_points(lib, seed=41, th=1.0, nnratio=0.8, n1=3000, n2=2000):
    case = make_local_points_case(n1, n2, seed)
    mt = F.ORBmatcher(nnratio, True, lib=lib)
    m, n = mt.SearchLocalPoints(case, th)
    om, on = O.search_local_points(case, th, nnratio)
    assert n == on and np.array_equal(m, om), "SearchByProjection(F, vpMapPoints) (seed %d, th %g)" % (seed, th)
    # a dense cluster: more candidates per point than the per-point list holds (window re-scan path)
    case = make_local_points_case(400, 500, seed + 1)
    rng = np.random.default_rng(seed)
    case["kp2_xy"][:60] = np.array([400.0, 200.0], np.float32) + rng.uniform(-4, 4, (60, 2)).astype(np.float32)
    case["kp2_octave"][:60] = 1
    case["desc2"][:60] = case["desc2"][0] ^ np.packbits(rng.random((60, 256)) < 0.03, axis=1, bitorder="little")
    case["uright2"][:60] = -1
    case["proj1"][:200, :2] = np.array([400.0, 200.0], np.float32) + rng.uniform(-2, 2, (200, 2)).astype(np.float32)
    case["level1"][:200] = 1
    case["mp_desc1"][:200] = case["desc2"][0]
    case["valid1"][:200] = 1
    m2, n2_ = mt.SearchLocalPoints(case, 3.0)
    om2, on2 = O.search_local_points(case, 3.0, nnratio)
    assert n2_ == on2 and np.array_equal(m2, om2)
    mt.close()
    return n


# ---- ORBmatcher::SearchForInitialization (monocular initialisation) ------------------------------------------------------
def check_search_for_initialization(lib, seed=61, window=100, nnratio=0.9, check_orientation=True, n1=5000):
    case = make_initialization_case(n1, seed)
    mt = F.ORBmatcher(nnratio, check_orientation, lib=lib)
    m, prev, n = mt.SearchForInitialization(case, window)
    om, oprev, on = O.search_for_initialization(case, window, nnratio, check_orientation)
    assert n == on and np.array_equal(m, om), "SearchForInitialization (seed %d, window %d)" % (seed, window)
    assert np.array_equal(prev.view(np.uint32), oprev.view(np.uint32))
    # the second call of the initialiser starts from the updated vbPrevMatched
    case2 = dict(case, prev_matched=prev)
    m2, prev2, n2 = mt.SearchForInitialization(case2, window // 2)
    om2, oprev2, on2 = O.search_for_initialization(case2, window // 2, nnratio, check_orientation)
    assert n2 == on2 and np.array_equal(m2, om2) and np.array_equal(prev2.view(np.uint32), oprev2.view(np.uint32))
    # a dense look-alike cluster: more deciding candidates than the per-feature list holds (window re-scan path)
    rng = np.random.default_rng(seed + 7)
    small = make_initialization_case(600, seed + 1)
    small["kp2_xy"][:80] = np.array([400.0, 200.0], np.float32) + rng.uniform(-30, 30, (80, 2)).astype(np.float32)
    small["kp2_octave"][:80] = 0
    small["desc2"][:80] = small["desc2"][0] ^ np.packbits(rng.random((80, 256)) < 0.03, axis=1, bitorder="little")
    small["prev_matched"][:150] = np.array([400.0, 200.0], np.float32) + rng.uniform(-10, 10, (150, 2)).astype(np.float32)
    small["kp1_octave"][:150] = 0
    small["desc1"][:150] = small["desc2"][0] ^ np.packbits(rng.random((150, 256)) < 0.02, axis=1, bitorder="little")
    for ratio in (nnratio, 1.5):
        mt2 = F.ORBmatcher(ratio, check_orientation, lib=lib)
        m3, prev3, n3 = mt2.SearchForInitialization(small, window)
        om3, oprev3, on3 = O.search_for_initialization(small, window, ratio, check_orientation)
        assert n3 == on3 and np.array_equal(m3, om3) and np.array_equal(prev3.view(np.uint32), oprev3.view(np.uint32))
        mt2.close()
    mt.close()
    return n


# ---- DBoW2 vocabulary transform ----------------------------------------------------------------------
def check_bow_transform(lib, tmp_dir, k=10, L=4, levelsup=2, seed=0, n_feat=2000):
    voc = synth.make_vocabulary(k, L, seed)
    varr = synth.vocabulary_arrays(voc)
    path = str(tmp_dir) + "/voc_%d.txt" % seed
    synth.write_vocabulary_text(path, voc)
    rng = np.random.default_rng(seed)
    leaves = voc["desc"][voc["is_leaf"] > 0]
    pick = leaves[rng.integers(0, len(leaves), n_feat * 3 // 4)]
    desc = np.ascontiguousarray(np.concatenate([pick ^ np.packbits(rng.random((len(pick), 256)) < 0.03, axis=1, bitorder="little"),
                                                synth.descriptors(n_feat - len(pick), seed)]))
    want = O.bow_transform(varr, desc, levelsup)
    for source in ("text", "arrays"):
        V = F.ORBVocabulary(lib=lib)
        if source == "text":
            assert V.loadFromTextFile(path)
        else:
            V.from_arrays(varr)
        info = V.info()
        assert info["L"] == L and info["n_nodes"] == varr["n_nodes"] and info["n_words"] == int(voc["is_leaf"].sum())
        got = V.transform(desc, levelsup)
        for g, w, name in zip(got, want, ("word ids", "word values", "node ids", "node offsets", "feature indices")):
            if name == "word values":
                assert np.array_equal(g.view(np.uint64), w.view(np.uint64)), name
            else:
                assert np.array_equal(g, w), name
        assert len(V.transform(desc[:0], levelsup)[0]) == 0
        V.close()
    V = F.ORBVocabulary(lib=lib)
    assert not V.loadFromTextFile(str(tmp_dir) + "/does_not_exist.txt")
    return len(want[0])


def check_depth_partial_batches(lib, dev=None, w=310, h=94, max_gen=None):
    """A handle sized for 4 scans used with 2, then 3, then 1 ...: every call must see empty maps (regression: the raw
    maps follow max_batch index maps, a single memset sized by the current batch missed them).  The map-free path does not
    clear its index maps but stamps a generation on the entries: max_gen (RGBL_DEPTH_MAX_GEN) makes the generation counter
    wrap within the test, and a host call that wants the maps (plain, cleared index map) sits in the middle.
    dev: torch device for the product library; None = the emulator, whose 'device' pointers are host pointers."""
    import ctypes as C
    import os
    saved = os.environ.get("RGBL_DEPTH_MAX_GEN")
    if max_gen is not None:
        os.environ["RGBL_DEPTH_MAX_GEN"] = str(max_gen)
    K = synth.KITTI_K.copy()
    K[0, 2], K[1, 2] = w / 2.0, h / 2.0
    K[0, 0] = K[1, 1] = 718.856 * w / synth.KITTI_W
    proj = F.projection_matrix(K, synth.KITTI_TR, lib)
    scans = [synth.lidar_scan(30 + i, n_rings=32, n_az=400) for i in range(4)]
    n = scans[0].shape[1]
    cap = 64
    rng = np.random.default_rng(2)
    try:
        dm = F.DepthModule(proj, w, h, max_points=n, max_keypoints=cap, max_batch=4, lib=lib)
    finally:
        if max_gen is not None:
            if saved is None:
                del os.environ["RGBL_DEPTH_MAX_GEN"]
            else:
                os.environ["RGBL_DEPTH_MAX_GEN"] = saved
    P = O.make_depth_params(proj)
    if dev is not None:
        import torch
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        ptr = lambda t: C.c_void_p(t.data_ptr())
        down = lambda t: t.cpu().numpy()
    else:
        up = lambda a: np.ascontiguousarray(a).copy()
        ptr = lambda a: C.c_void_p(a.ctypes.data)
        down = lambda a: a
    for step, (batch, first) in enumerate(((2, 0), (3, 1), (1, 3), (4, 0), (2, 2), (4, 1), (1, 0), (3, 3), (4, 2))):
        if step == 4:  # single-scan host call with the maps, through the same handle
            hk = np.stack([rng.uniform(0, w - 1, cap), rng.uniform(0, h - 1, cap)], 1).astype(np.float32)
            dm.CalculateDepthFromPcd(hk, hk, scans[2], w, h)
            od, our, oraw, oproc = O.depth(P, scans[2], w, h, hk, hk[:, 0])
            assert np.array_equal(bits(dm.RawDepthMap), bits(oraw)) and np.array_equal(bits(dm.ProcessedDepthMap), bits(oproc))
            assert np.array_equal(bits(dm.mvDepth), bits(od))
        cloud = np.stack([scans[(first + b) % 4] for b in range(batch)])
        kps = np.zeros((batch, cap), O.KP_DTYPE)
        kps["x"] = rng.uniform(0, w - 1, (batch, cap)).astype(np.float32)
        kps["y"] = rng.uniform(0, h - 1, (batch, cap)).astype(np.float32)
        cnt = np.full(batch, cap, np.int32)
        d_cloud, d_kp, d_n = up(cloud), up(kps.view(np.float32).reshape(batch, cap, 7)), up(cnt)
        d_depth, d_ur = up(np.zeros((batch, cap), np.float32)), up(np.zeros((batch, cap), np.float32))
        d_proc = up(np.zeros((batch, h, w), np.float32))
        L.check(lib, lib.rgbl_depth_batch_device(dm.h, ptr(d_cloud), batch, n, n, 4 * n, w, h, ptr(d_kp), ptr(d_n), cap, None,
                                                 ptr(d_depth), ptr(d_ur), ptr(d_proc)))
        L.check(lib, lib.rgbl_depth_sync(dm.h))
        for b in range(batch):
            od, our, oraw, oproc = O.depth(P, cloud[b], w, h, np.stack([kps["x"][b], kps["y"][b]], 1), kps["x"][b])
            assert np.array_equal(bits(down(d_proc)[b]), bits(oproc)), "processed map, batch %d frame %d" % (batch, b)
            assert np.array_equal(bits(down(d_depth)[b]), bits(od)) and np.array_equal(bits(down(d_ur)[b]), bits(our))
    # prefetch(A), a batch-device projection of B through the same handle, then compute(A) - the prefetched maps
    # are gone and compute(A) must project A again instead of gathering from B's maps; and a cancelled prefetch is forgotten
    hk = np.stack([rng.uniform(0, w - 1, cap), rng.uniform(0, h - 1, cap)], 1).astype(np.float32)
    A, Bc = np.ascontiguousarray(scans[0]), np.ascontiguousarray(scans[1])
    odA = O.depth(P, A, w, h, hk, hk[:, 0])[0]
    odB = O.depth(P, Bc, w, h, hk, hk[:, 0])[0]
    assert not np.array_equal(bits(odA), bits(odB))
    d_B = up(Bc[None])
    dm.PrefetchPointcloud(A, w, h)
    L.check(lib, lib.rgbl_depth_project_batch_device(dm.h, ptr(d_B), 1, n, n, 4 * n, w, h, None))
    dm.CalculateDepthFromPcd(hk, hk, A, w, h, want_maps=False)
    assert np.array_equal(bits(dm.mvDepth), bits(odA)), "compute(A) after prefetch(A) + project_batch_device(B) gathered from B's maps"
    dm.PrefetchPointcloud(A, w, h)
    dm.CancelPrefetch()
    A[:] = Bc                                   # the same address now holds another scan
    dm.CalculateDepthFromPcd(hk, hk, A, w, h, want_maps=False)
    assert np.array_equal(bits(dm.mvDepth), bits(odB)), "a cancelled prefetch was still used"
    dm.close()


def check_extractor_partial_batches(lib, w=400, h=300, nfeatures=500):
    """One handle sized for 4 frames, called with 2, 3, 1 and 4 frames: every frame must equal the oracle."""
    ex = F.ORBextractor(nfeatures, 1.2, 8, 12, 7, w, h, max_batch=4, lib=lib)
    orc = O.Extractor(nfeatures, 1.2, 8, 12, 7)
    s = synth.Sequence(17, w, h, n_frames=6)
    for batch, first in ((2, 0), (3, 2), (1, 5), (4, 1)):
        imgs = np.stack([s.frame((first + b) % 6) for b in range(batch)])
        for b, (kps, desc, mono) in enumerate(ex.extract_batch(imgs)):
            okps, odesc, omono = orc(imgs[b])
            assert_keypoints_equal(kps, okps, "partial batch %d frame %d" % (batch, b))
            assert np.array_equal(desc, odesc) and mono == omono
    ex.close()


def check_extractor_replay(lib, w=400, h=300, nfeatures=500):
    """The host-pointer path replays a captured launch graph when (batch, stride, lapping area) repeat: different images
    through the same handle, a lapping-area change and a batch change in between, must all equal the oracle."""
    ex = F.ORBextractor(nfeatures, 1.2, 8, 12, 7, w, h, max_batch=2, lib=lib)
    orc = O.Extractor(nfeatures, 1.2, 8, 12, 7)
    s = synth.Sequence(29, w, h, n_frames=8)
    plan = [((0, 0), 1)] * 3 + [((100, 200), 1)] * 2 + [((0, 0), 2)] * 2 + [((0, 0), 1)] * 2
    for i, (lap, batch) in enumerate(plan):
        imgs = np.stack([s.frame((i + b) % 8) for b in range(batch)])
        res = ex.extract_batch(imgs, lap) if batch > 1 else [ex(imgs[0], None, lap)]
        for b, (kps, desc, mono) in enumerate(res):
            okps, odesc, omono = orc(imgs[b], lap)
            assert_keypoints_equal(kps, okps, "replay call %d frame %d" % (i, b))
            assert np.array_equal(desc, odesc) and mono == omono
    ex.close()


def check_extractor_under_load(lib, w=synth.KITTI_W, h=synth.KITTI_H, nfeatures=2000, batch=256, distinct=8, rounds=3):
    """The bench-sized batch (all workgroups of every kernel in flight, both extractor streams busy): every one of the
    `batch` frames - `distinct` different images repeated - must come out identical to the oracle, call after call."""
    ex = F.ORBextractor(nfeatures, 1.2, 8, 12, 7, w, h, max_batch=batch, lib=lib)
    orc = O.Extractor(nfeatures, 1.2, 8, 12, 7)
    s = synth.Sequence(23, w, h, n_frames=distinct)
    base = [s.frame(i) for i in range(distinct)]
    want = [orc(img) for img in base]
    imgs = np.stack([base[i % distinct] for i in range(batch)])
    for r in range(rounds):
        res = ex.extract_batch(imgs)
        for i, (kps, desc, mono) in enumerate(res):
            okps, odesc, omono = want[i % distinct]
            assert_keypoints_equal(kps, okps, "round %d frame %d" % (r, i))
            assert np.array_equal(desc, odesc) and mono == omono
    ex.close()


def check_search_by_bow(lib, seed=51, nnratio=0.7, check_ori=True, n=1500, nodes=100):
    kf, fr, *_ = make_triangulation_case(n, seed=seed, n_nodes=nodes)
    rng = np.random.default_rng(seed)
    kf = dict(kf, has_mp=(rng.random(n) < 0.7).astype(np.uint8))
    mt = F.ORBmatcher(nnratio, check_ori, lib=lib)
    m, nm = mt.SearchByBoW(kf, fr)
    om, onm = O.search_by_bow(kf, fr, nnratio, check_ori)
    assert nm == onm and np.array_equal(m, om), "SearchByBoW (seed %d)" % seed
    mt.close()
    return nm


def check_search_by_bow_rig(lib, seed=61, nnratio=0.7, check_ori=True, n=1500, nodes=100):
    """SearchByBoW on a two-camera frame (F.Nleft != -1, ORBmatcher.cc:298-326, 357-386).  Returns (nmatches, map points that went
    to a left AND a right feature)."""
    from orb_slam3_rgbl_amd.cases import make_bow_rig_case
    kf, fr, n_left = make_bow_rig_case(n, seed=seed, n_nodes=nodes)
    rng = np.random.default_rng(seed)
    kf = dict(kf, has_mp=(rng.random(n) < 0.7).astype(np.uint8))
    mt = F.ORBmatcher(nnratio, check_ori, lib=lib)
    m, nm = mt.SearchByBoW(kf, fr, n_left)
    om, onm = O.search_by_bow(kf, fr, nnratio, check_ori, n_left=n_left)
    assert nm == onm and np.array_equal(m, om), "SearchByBoW, two-camera frame (seed %d)" % seed
    # Nleft = N: every feature is a left one - the single-camera result
    m1, nm1 = mt.SearchByBoW(kf, fr, len(fr["desc"]))
    om1, onm1 = O.search_by_bow(kf, fr, nnratio, check_ori)
    assert nm1 == onm1 and np.array_equal(m1, om1)
    # Nleft = 0: no left feature, and the right camera's match sits inside the left one's test (:315) - nothing is matched
    m0, nm0 = mt.SearchByBoW(kf, fr, 0)
    om0, onm0 = O.search_by_bow(kf, fr, nnratio, check_ori, n_left=0)
    assert nm0 == onm0 == 0 and np.array_equal(m0, om0) and not (m0 >= 0).any()
    mt.close()
    both = np.intersect1d(m[:n_left][m[:n_left] >= 0], m[n_left:][m[n_left:] >= 0])
    return nm, len(both)


def check_search_by_bow_keyframes(lib, seed=81, nnratio=0.75, check_ori=True, n=1500, nodes=100):
    kf1, kf2, *_ = make_triangulation_case(n, seed=seed, n_nodes=nodes)
    rng = np.random.default_rng(seed)
    kf1 = dict(kf1, has_mp=(rng.random(n) < 0.8).astype(np.uint8))
    kf2 = dict(kf2, has_mp=(rng.random(len(kf2["desc"])) < 0.8).astype(np.uint8))
    mt = F.ORBmatcher(nnratio, check_ori, lib=lib)
    m, nm = mt.SearchByBoWKeyFrames(kf1, kf2)
    om, onm = O.search_by_bow_kf(kf1, kf2, nnratio, check_ori)
    assert nm == onm and np.array_equal(m, om), "SearchByBoW(KF, KF) (seed %d)" % seed
    mt.close()
    return nm


def check_fuse_search(lib, seed=91, th=3.0, n1=2500, n2=2000):
    case = make_fuse_case(n1, n2, seed)
    valid, level = fuse_prepass(case)
    ovalid, olevel = O.fuse_prepass(case)
    assert np.array_equal(valid, ovalid) and np.array_equal(level[valid != 0], olevel[valid != 0]), "fuse prepass"
    mt = F.ORBmatcher(0.6, True, lib=lib)
    best, dist = mt.FuseSearch(dict(case, valid1=valid, level1=level), th)
    obest, on = O.fuse_search(case, th)
    assert np.array_equal(best, obest), "Fuse search (seed %d, th %g)" % (seed, th)
    assert int((best >= 0).sum()) == on and np.all(dist[best >= 0] <= 50) and np.all((dist[best < 0] > 50))
    mt.close()
    return on


def check_undistort(lib, dev=None, w=752, h=480):
    """Frame::UndistortKeyPoints with the EuRoC cam0 intrinsics / distortion (Examples/Stereo-Inertial/EuRoC.yaml) and a
    5-coefficient fisheye-ish set: host entry point and the device batch variant against the oracle, bit for bit; and a
    sanity property - distorting the result again lands on the input."""
    import ctypes as C
    rng = np.random.default_rng(7)
    ex = F.ORBextractor(500, 1.2, 8, 20, 7, w, h, max_batch=2, lib=lib)
    for K, dist in (((458.654, 457.296, 367.215, 248.375), (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05)),
                    ((380.0, 379.5, 370.2, 243.1), (-0.15, 0.03, 1e-3, -5e-4, -0.002)),
                    ((718.856, 718.856, 607.19, 185.2), (1e-9, 0.0, 0.0, 0.0))):
        xy = np.stack([rng.uniform(-20, w + 20, 3000), rng.uniform(-20, h + 20, 3000)], 1).astype(np.float32)
        xy[:4] = [[0, 0], [w, 0], [0, h], [w, h]]                  # Frame::ComputeImageBounds' corners
        got = ex.UndistortKeyPoints(xy, K, dist)
        want = O.undistort_points(xy, K, dist)
        assert np.array_equal(bits(got), bits(want)), "undistortPoints"
        # re-distort (forward Brown-Conrady model) and compare with the input where the iteration converged
        fx, fy, cx, cy = K
        k = list(dist) + [0.0] * (5 - len(dist))
        x, y = (want[:, 0].astype(np.float64) - cx) / fx, (want[:, 1].astype(np.float64) - cy) / fy
        r2 = x * x + y * y
        cd = 1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2
        xd = x * cd + 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x)
        yd = y * cd + k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y
        back = np.stack([xd * fx + cx, yd * fy + cy], 1)
        inner = (np.abs(xy[:, 0] - cx) < 0.2 * w) & (np.abs(xy[:, 1] - cy) < 0.2 * h)   # 5 iterations: converged near the centre only
        assert np.abs(back - xy)[inner].max() < 0.01
    # device batch variant on keypoint records
    kp = np.zeros((2, 300), O.KP_DTYPE)
    kp["x"] = rng.uniform(0, w, (2, 300)).astype(np.float32)
    kp["y"] = rng.uniform(0, h, (2, 300)).astype(np.float32)
    cnt = np.array([300, 123], np.int32)
    K = np.array([458.654, 457.296, 367.215, 248.375], np.float32)
    dist = np.array([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05], np.float32)
    if dev is not None:
        import torch
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        ptr = lambda t: C.c_void_p(t.data_ptr())
        down = lambda t: t.cpu().numpy()
    else:
        up = lambda a: np.ascontiguousarray(a).copy()
        ptr = lambda a: C.c_void_p(a.ctypes.data)
        down = lambda a: a
    d_kp, d_n, d_out = up(kp.view(np.float32).reshape(2, 300, 7)), up(cnt), up(np.zeros((2, 300, 2), np.float32))
    L.check(lib, lib.rgbl_undistort_keypoints_batch_device(ex.h, ptr(d_kp), ptr(d_n), 2, 300, L.ptr(K), L.ptr(dist), 4, ptr(d_out)))
    L.check(lib, lib.rgbl_extractor_sync(ex.h))
    out = down(d_out)
    for b in range(2):
        want = O.undistort_points(np.stack([kp["x"][b], kp["y"][b]], 1)[:cnt[b]], K, dist)
        assert np.array_equal(bits(out[b, :cnt[b]]), bits(want)) and not out[b, cnt[b]:].any()
    ex.close()


def check_distinctive_descriptors(lib, seed=101, n_points=400):
    """Batched MapPoint::ComputeDistinctiveDescriptors: observation counts 0 .. 200 (typical 2 .. 30), exact duplicates (ties
    between rows: the first minimum must win), near-duplicates of a common ancestor as real observations are."""
    rng = np.random.default_rng(seed)
    lists = []
    for p in range(n_points):
        n = int(rng.choice([0, 1, 2, 3, 5, 8, 13, 21, 34, 65, 130, 200], p=[.02, .05, .1, .13, .2, .2, .12, .08, .05, .03, .01, .01]))
        base = synth.descriptors(1, seed * 1000 + p)[0]
        d = np.repeat(base[None], n, 0)
        flips = rng.random((n, 256)) < rng.uniform(0.01, 0.2)
        d = d ^ np.packbits(flips, axis=1, bitorder="little")
        if n > 3 and p % 3 == 0:
            d[rng.integers(0, n)] = d[rng.integers(0, n)]          # exact duplicate rows
        if n > 2 and p % 7 == 0:
            d[:] = d[0]                                            # all equal: BestIdx must be 0
        lists.append(d)
    mt = F.ORBmatcher(0.6, True, lib=lib)
    got = mt.ComputeDistinctiveDescriptors(lists)
    want = O.distinctive_descriptors(lists)
    assert np.array_equal(got, want), "ComputeDistinctiveDescriptors (seed %d)" % seed
    assert mt.ComputeDistinctiveDescriptors([]).size == 0
    mt.close()
    return int((want >= 0).sum())


def make_project_search_case(n1=2500, n2=2000, seed=111):
    """Camera-frame map points over a key frame (what Fuse(pKF, Scw, ...) and SearchBySim3 hand to the search): the fuse case with
    the points moved into the camera frame in float64 and rounded once (the product and the oracle both start from these)."""
    c = make_fuse_case(n1, n2, seed)
    valid, level = fuse_prepass(c)
    R = _rot(c["Tcw_q"]).astype(np.float64)
    cam = (c["world_pos1"].astype(np.float64) @ R.T + c["Tcw_t"].astype(np.float64)).astype(np.float32)
    return dict(valid1=valid, cam_pos1=cam, mp_desc1=c["mp_desc1"], level1=level, kp2_xy=c["kp2_xy"], kp2_octave=c["kp2_octave"],
                desc2=c["desc2"], grid=c["grid"], K=c["K"], scale_factors=c["scale_factors"])


def check_project_search(lib, seed=111, th=4.0, proj_form=0, max_dist=50, n1=2500, n2=2000):
    case = make_project_search_case(n1, n2, seed)
    mt = F.ORBmatcher(0.75, True, lib=lib)
    best, dist = mt.ProjectSearch(case, th, proj_form, max_dist)
    obest, odist = O.project_search(case, th, proj_form, max_dist)
    assert np.array_equal(best, obest) and np.array_equal(dist, odist), "project search (seed %d, form %d)" % (seed, proj_form)
    mt.close()
    return int((best >= 0).sum())


def camera_prepass(cam_pos, min_dist, max_dist, log_scale_factor, n_levels, normal=None):
    """dist3D = |p3Dc| (SearchBySim3) resp. |p3Dw - Ow| with Ow = 0, the invariance range, the optional viewing-angle test and
    PredictScale, in numpy float32 + the C library's logf: what the shims evaluate with the MapPoint objects."""
    P = np.asarray(cam_pos, np.float32)
    sq = (P * P).astype(np.float32)
    dist = np.sqrt((sq[:, 0] + (sq[:, 1] + sq[:, 2]).astype(np.float32)).astype(np.float32)).astype(np.float32)
    valid = ~(dist < (np.float32(0.8) * min_dist).astype(np.float32)) & ~(dist > (np.float32(1.2) * max_dist).astype(np.float32))
    if normal is not None:
        pr = (P * np.asarray(normal, np.float32)).astype(np.float32)
        dot = (pr[:, 0] + (pr[:, 1] + pr[:, 2]).astype(np.float32)).astype(np.float32)
        valid &= ~(dot.astype(np.float64) < 0.5 * dist.astype(np.float64))
    level = F.ORBmatcher.PredictScale(dist, max_dist, log_scale_factor, n_levels)
    return valid, np.where(valid, level, 0).astype(np.int32)


def make_sim3_case(n=1500, seed=121, w=synth.KITTI_W, h=synth.KITTI_H):
    """Two key frames that see the same place (all poses identities, so one common camera frame): KF2's features are KF1's,
    permuted, moved by a pixel or two and with a few descriptor bits flipped; every feature's map point sits on the ray of its
    counterpart in the OTHER key frame, so that the two directed searches of SearchBySim3 mostly agree.  Some features have no
    or a bad map point, some points fall outside their invariance range, some lie behind the camera."""
    rng = np.random.default_rng(seed)
    K = np.array([718.856, 718.856, 607.1928, 185.2157], np.float32)
    sf = (1.2 ** np.arange(8)).astype(np.float32)
    xy1 = np.stack([rng.uniform(20, w - 20, n), rng.uniform(20, h - 20, n)], 1).astype(np.float32)
    oct1 = rng.integers(0, 8, n).astype(np.int32)
    desc1 = synth.descriptors(n, seed)
    perm = rng.permutation(n)
    inv = np.argsort(perm)                                   # feature i1 of KF1 <-> feature inv[i1] of KF2
    xy2 = (xy1[perm] + rng.normal(0, 1.0, (n, 2))).astype(np.float32)
    oct2 = oct1[perm].copy()
    desc2 = desc1[perm] ^ np.packbits(rng.random((n, 256)) < 0.03, axis=1, bitorder="little")

    def side(xy_self, octave, desc, xy_other, partner):
        z = rng.uniform(4, 60, n)
        z[rng.random(n) < 0.03] *= -1
        tgt = xy_other[partner] + rng.normal(0, 1.0, (n, 2))
        pos = np.stack([(tgt[:, 0] - K[2]) / K[0] * z, (tgt[:, 1] - K[3]) / K[1] * z, z], 1).astype(np.float32)
        dist = np.linalg.norm(pos.astype(np.float64), axis=1)
        lvl = np.clip(octave + rng.integers(0, 2, n), 0, 7)   # predicted level = octave or octave + 1: band [l - 1, l] holds the octave
        max_d = (dist * 1.2 ** lvl * rng.uniform(0.86, 0.99, n)).astype(np.float32)
        min_d = (max_d / np.float32(1.2 ** 7)).astype(np.float32)
        far = rng.random(n) < 0.04
        max_d[far] = (dist[far] * 0.5).astype(np.float32)
        normal = -pos / np.maximum(np.linalg.norm(pos, axis=1, keepdims=True), 1e-6) + rng.normal(0, 0.3, pos.shape)
        normal /= np.linalg.norm(normal, axis=1, keepdims=True)
        normal = -normal if False else normal
        state = rng.choice([0, 1, 2], n, p=[0.12, 0.8, 0.08]).astype(np.uint8)
        mp_desc = desc ^ np.packbits(rng.random((n, 256)) < 0.02, axis=1, bitorder="little")
        return dict(kp_xy=xy_self, kp_octave=octave, desc=desc, mp_state=state, mp_pos=pos, mp_normal=(-normal).astype(np.float32),
                    mp_desc=mp_desc, mp_min_dist=min_d, mp_max_dist=max_d)
    a1 = side(xy1, oct1, desc1, xy2, inv)
    a2 = side(xy2, oct2, desc2, xy1, perm)
    gw, gh = np.float32(w), np.float32(h)
    grid = np.array([0, 0, gw, gh, np.float32(64) / gw, np.float32(48) / gh], np.float32)
    prior = np.where(rng.random(n) < 0.1, inv, -1).astype(np.int32)      # matches found earlier (by SearchByBoW)
    return dict(a1=a1, a2=a2, K=K, grid=grid, scale_factors=sf, log_scale_factor=np.float32(np.log(np.float32(1.2))), prior12=prior, inv=inv)


def sim3_direction(case, src, dst, already, th):
    """One directed search of SearchBySim3 (ORBmatcher.cc:1497-1566) as input of the C ABI / the oracle."""
    a, b = case[src], case[dst]
    valid, level = camera_prepass(a["mp_pos"], a["mp_min_dist"], a["mp_max_dist"], case["log_scale_factor"], len(case["scale_factors"]))
    valid &= (a["mp_state"] == 1) & ~already
    return dict(valid1=valid.astype(np.uint8), cam_pos1=a["mp_pos"], mp_desc1=a["mp_desc"], level1=level, kp2_xy=b["kp_xy"],
                kp2_octave=b["kp_octave"], desc2=b["desc"], grid=case["grid"], K=case["K"], scale_factors=case["scale_factors"])


def search_by_sim3(case, th, search):
    """SearchBySim3 on top of a per-point search function(case, th, proj_form, max_dist) -> (best_idx, best_dist)."""
    n1, n2 = len(case["a1"]["kp_xy"]), len(case["a2"]["kp_xy"])
    prior = case["prior12"]
    am1 = prior >= 0
    am2 = np.zeros(n2, bool)
    # vbAlreadyMatched2[idx2] for idx2 = GetIndexInKeyFrame(pKF2) of the prior match (the map point of KF2's feature prior[i])
    am2[prior[am1]] = True
    m1, _ = search(sim3_direction(case, "a1", "a2", am1, th), th, 1, 100)
    m2, _ = search(sim3_direction(case, "a2", "a1", am2, th), th, 1, 100)
    out = prior.copy()
    found = 0
    for i1 in range(n1):
        if m1[i1] >= 0 and m2[m1[i1]] == i1:
            out[i1] = m1[i1]
            found += 1
    return out, found


def check_search_by_sim3(lib, seed=121, th=7.5, n=1500):
    case = make_sim3_case(n, seed)
    mt = F.ORBmatcher(0.75, True, lib=lib)
    got, ng = search_by_sim3(case, th, mt.ProjectSearch)
    want, nw = search_by_sim3(case, th, O.project_search)
    assert ng == nw and np.array_equal(got, want), "SearchBySim3 (seed %d)" % seed
    mt.close()
    return nw


def check_search_by_projection_sim3(lib, seed=151, th=8, proj_form=0, ratio=1.5, n1=2500, n2=2000):
    case = make_project_search_case(n1, n2, seed)
    rng = np.random.default_rng(seed)
    matched2 = (rng.random(n2) < 0.12).astype(np.uint8)
    max_dist = int(np.floor(np.float32(50) * np.float32(ratio)))
    mt = F.ORBmatcher(0.75, True, lib=lib)
    m, n = mt.SearchByProjectionSim3(case, matched2, th, proj_form, max_dist)
    om, on = O.search_by_projection_sim3(case, matched2, th, proj_form, max_dist)
    assert n == on and np.array_equal(m, om), "SearchByProjection(KF, Sim3) (seed %d, form %d)" % (seed, proj_form)
    free, _ = O.search_by_projection_sim3(case, np.zeros(n2, np.uint8), th, proj_form, max_dist)
    assert not np.array_equal(free, om)
    mt.close()
    return on


def check_pipeline_gather(lib, mode, dev=None, w=640, h=360, nfeatures=800, batch=16, steps=4, n_az=600, levels=8,
                          sparse_depth=False):
    """The batched step + the gather of its records (orb_slam3_rgbl_amd/pipeline.py) with ONE rank: what the root holds after
    every step must decode to that step's own outputs, and those to the oracle's."""
    import torch

    from orb_slam3_rgbl_amd.pipeline import FrontEndPipeline, unpack_records
    dev = dev or torch.device("cuda", 0)
    K = synth.KITTI_K.copy()
    K[0, 2], K[1, 2] = w / 2.0, h / 2.0
    proj = F.projection_matrix(K, synth.KITTI_TR, lib)
    sq = synth.Sequence(70, w, h, n_frames=batch)
    frames = np.stack([sq.frame(i) for i in range(batch)])
    cloud = np.stack([synth.lidar_scan(700 + i, n_az=n_az) for i in range(batch)])
    pipe = FrontEndPipeline(lib, torch, dev, w, h, nfeatures, proj, cloud.shape[2], batch, levels=levels, ini_th=20, min_th=7, world=1, rank=0,
                            gather=mode, keep_steps=steps, log_steps=steps, sparse_depth=sparse_depth)
    pipe.set_inputs(torch.from_numpy(frames).to(dev), torch.from_numpy(cloud).to(dev))
    for _ in range(steps):
        pipe.step()
    pipe.finish()
    pipe.sync()
    assert len(pipe.received) == steps
    o = pipe.last()
    n = o.n.cpu().numpy()
    assert n.min() > 50
    kp, desc, depth, uright = (t.cpu().numpy() for t in (o.kp, o.desc, o.depth, o.uright))
    for got in pipe.received:             # every step processed the same resident batch
        counts, rec = got[0]
        assert np.array_equal(counts, n)
        fr = unpack_records(rec.cpu().numpy(), counts)
        for f in range(batch):
            m = int(n[f])
            assert np.array_equal(fr[f]["kp"], kp[f, :m].view(np.uint8).reshape(m, 28))
            assert np.array_equal(fr[f]["desc"], desc[f, :m])
            assert np.array_equal(bits(fr[f]["depth"]), bits(depth[f, :m])) and np.array_equal(bits(fr[f]["uright"]), bits(uright[f, :m]))
    orc = O.Extractor(nfeatures, 1.2, levels, 20, 7)
    P = O.make_depth_params(proj)
    fr = unpack_records(pipe.received[-1][0][1].cpu().numpy(), pipe.received[-1][0][0])
    for f in (0, batch // 2, batch - 1):
        okps, odesc, _ = orc(frames[f])
        assert fr[f]["n"] == len(okps) and np.array_equal(fr[f]["kp"], okps.view(np.uint8).reshape(len(okps), 28))
        assert np.array_equal(fr[f]["desc"], odesc)
        od, our, _, _ = O.depth(P, cloud[f], w, h, np.stack([okps["x"], okps["y"]], 1), okps["x"], want_maps=False)
        assert np.array_equal(bits(fr[f]["depth"]), bits(od)) and np.array_equal(bits(fr[f]["uright"]), bits(our))
    pipe.close()


def check_pipeline_idle_steps(lib, mode, dev=None, w=640, h=360, nfeatures=800, batch=8, n_az=600, levels=8):
    """FrontEndPipeline.step(active=False): a rank without frames for a step (BASELINE configs[3]: 11 sequences on 8 ranks) still takes
    part in the step's gather with all-zero counts.  One rank: active, idle, active, idle - the root's records of the active steps equal
    a plain run's, the idle steps arrive empty, and nothing of an idle step leaks into the next active one."""
    import torch

    from orb_slam3_rgbl_amd.pipeline import FrontEndPipeline, unpack_records
    dev = dev or torch.device("cuda", 0)
    K = synth.KITTI_K.copy()
    K[0, 2], K[1, 2] = w / 2.0, h / 2.0
    proj = F.projection_matrix(K, synth.KITTI_TR, lib)
    sq = synth.Sequence(71, w, h, n_frames=batch)
    frames = np.stack([sq.frame(i) for i in range(batch)])
    cloud = np.stack([synth.lidar_scan(710 + i, n_az=n_az) for i in range(batch)])
    pattern = (True, False, True, False, False, True)
    pipe = FrontEndPipeline(lib, torch, dev, w, h, nfeatures, proj, cloud.shape[2], batch, levels=levels, ini_th=20, min_th=7, world=1, rank=0,
                            gather=mode, keep_steps=len(pattern), log_steps=len(pattern))
    pipe.set_inputs(torch.from_numpy(frames).to(dev), torch.from_numpy(cloud).to(dev))
    for a in pattern:
        pipe.step(active=a)
    pipe.finish()
    pipe.sync()
    assert len(pipe.received) == len(pattern)
    orc = O.Extractor(nfeatures, 1.2, levels, 20, 7)
    ref = [orc(frames[f])[:2] for f in range(batch)]
    for a, got in zip(pattern, pipe.received):
        counts, rec = got[0]
        if not a:
            assert int(np.abs(counts).sum()) == 0 and rec.numel() == 0
            continue
        fr = unpack_records(rec.cpu().numpy(), counts)
        for f in range(batch):
            okps, odesc = ref[f]
            assert fr[f]["n"] == len(okps) and np.array_equal(fr[f]["kp"], okps.view(np.uint8).reshape(len(okps), 28))
            assert np.array_equal(fr[f]["desc"], odesc)
    pipe.close()


def check_pipeline_step(lib, w, h, nfeatures, batch, n_az, dev=None, steps=2, seq=90, levels=8, ini=12, mn=7):
    """The batched step exactly as bench.py times it (FrontEndPipeline: extract -> depth -> match on resident inputs, no gather):
    EVERY frame of the last step against the oracle - keypoints, descriptors, depth, uRight and the matches against the
    following frame.  BASELINE configs[4] runs through here at its own size (3840x2160, 8000 features, 262 144 points)."""
    import torch

    from orb_slam3_rgbl_amd.pipeline import FrontEndPipeline
    dev = dev or torch.device("cuda", 0)
    K = synth.KITTI_K.copy()
    K[0, 0] = K[1, 1] = 718.856 * w / synth.KITTI_W
    K[0, 2], K[1, 2] = w / 2.0, h / 2.0
    proj = F.projection_matrix(K, synth.KITTI_TR, lib)
    sq = synth.Sequence(seq, w, h, n_frames=batch)
    frames = np.stack([sq.frame(i) for i in range(batch)])
    cloud = np.stack([synth.lidar_scan(10 * seq + i, n_az=n_az) for i in range(batch)])
    pipe = FrontEndPipeline(lib, torch, dev, w, h, nfeatures, proj, cloud.shape[2], batch, levels=levels, ini_th=ini, min_th=mn, gather="none")
    pipe.set_inputs(torch.from_numpy(frames).to(dev), torch.from_numpy(cloud).to(dev))
    for _ in range(steps):
        pipe.step()
    pipe.finish()
    pipe.sync()
    o = pipe.last()
    n = o.n.cpu().numpy()
    kp, desc, depth, uright, bi, bd, sd = (t.cpu().numpy() for t in (o.kp, o.desc, o.depth, o.uright, o.bi, o.bd, o.sd))
    orc = O.Extractor(nfeatures, 1.2, levels, ini, mn)
    P = O.make_depth_params(proj)
    ref = [orc(frames[f])[:2] for f in range(batch)]
    for f in range(batch):
        okps, odesc = ref[f]
        m = int(n[f])
        assert m == len(okps), "frame %d: %d keypoints, oracle %d" % (f, m, len(okps))
        assert np.array_equal(kp[f, :m].view(np.uint32), okps.view(np.uint32).reshape(m, 7)), "keypoints of frame %d" % f
        assert np.array_equal(desc[f, :m], odesc), "descriptors of frame %d" % f
        od, our, _, _ = O.depth(P, cloud[f], w, h, np.stack([okps["x"], okps["y"]], 1), okps["x"], want_maps=False)
        assert np.array_equal(bits(depth[f, :m]), bits(od)) and np.array_equal(bits(uright[f, :m]), bits(our)), "depth of frame %d" % f
        obi, obd, osd = O.hamming_bf(odesc, ref[(f + 1) % batch][1])
        assert np.array_equal(bi[f, :m], obi) and np.array_equal(bd[f, :m], obd) and np.array_equal(sd[f, :m], osd), "matches of frame %d" % f
    pipe.close()
    return int(n.min())


def check_overlapped_frame(lib, w=synth.KITTI_W, h=synth.KITTI_H, nfeatures=2000, frames=3):
    """The optional latency hooks: rgbl_extract_begin + rgbl_depth_prefetch, then the ordinary calls on the same buffers, must
    give what the ordinary calls alone give (= the oracle's results), whatever is interleaved."""
    ex = F.ORBextractor(nfeatures, 1.2, 8, 12, 7, w, h, lib=lib)
    orc = O.Extractor(nfeatures, 1.2, 8, 12, 7)
    K = synth.KITTI_K.copy()
    K[0, 2], K[1, 2] = w / 2.0, h / 2.0
    proj = F.projection_matrix(K, synth.KITTI_TR, lib)
    P = O.make_depth_params(proj)
    dm = F.DepthModule(proj, w, h, max_keypoints=ex.max_keypoints, lib=lib)
    sq = synth.Sequence(17, w, h, n_frames=frames)
    for i in range(frames):
        img = np.ascontiguousarray(sq.frame(i))
        cloud = np.ascontiguousarray(synth.lidar_scan(170 + i, n_az=600 if w < 1000 else 1900), np.float32)
        ex.Begin(img)
        dm.PrefetchPointcloud(cloud, w, h)
        kps, desc, mono = ex(img)                                   # collects the begun extraction
        dm.CalculateDepthFromPcd(kps, kps, cloud, w, h, want_maps=False)   # gathers on the prefetched maps
        okps, odesc, omono = orc(img)
        assert_keypoints_equal(kps, okps, "overlapped frame %d" % i)
        assert np.array_equal(desc, odesc) and mono == omono
        od, our, _, oproc = O.depth(P, cloud, w, h, np.stack([okps["x"], okps["y"]], 1), okps["x"])
        assert np.array_equal(bits(dm.mvDepth), bits(od)) and np.array_equal(bits(dm.mvuRight), bits(our))
        assert (od > 0).sum() > 20
    # a begun extraction that is overtaken by another image is dropped, a prefetched scan that is not the one computed is ignored
    img0, img1 = np.ascontiguousarray(sq.frame(0)), np.ascontiguousarray(sq.frame(1))
    c0 = np.ascontiguousarray(synth.lidar_scan(170, n_az=600 if w < 1000 else 1900), np.float32)
    c1 = np.ascontiguousarray(synth.lidar_scan(171, n_az=600 if w < 1000 else 1900), np.float32)
    ex.Begin(img0)
    dm.PrefetchPointcloud(c0, w, h)
    kps, desc, mono = ex(img1)
    okps, odesc, _ = orc(img1)
    assert_keypoints_equal(kps, okps, "overtaken")
    assert np.array_equal(desc, odesc)
    dm.CalculateDepthFromPcd(kps, kps, c1, w, h, want_maps=True)     # maps wanted: everything is computed
    od, our, oraw, oproc = O.depth(P, c1, w, h, np.stack([okps["x"], okps["y"]], 1), okps["x"])
    assert np.array_equal(bits(dm.mvDepth), bits(od)) and np.array_equal(bits(dm.ProcessedDepthMap), bits(oproc))
    dm.PrefetchPointcloud(c0, w, h)
    dm.CalculateDepthFromPcd(kps, kps, c1, w, h, want_maps=False)    # another scan than the prefetched one
    assert np.array_equal(bits(dm.mvDepth), bits(od))
    # a cancelled begin is forgotten: the same buffer with new contents is extracted afresh
    buf = img0.copy()
    ex.Begin(buf)
    ex.CancelBegin()
    buf[:] = img1
    kps2, desc2, _ = ex(buf)
    assert_keypoints_equal(kps2, okps, "cancelled begin")
    assert np.array_equal(desc2, odesc)
    ex.close(); dm.close()


def check_extractor_low_contrast(lib, w=614, h=343, ini=20, mn=7, contrast=0.15, seq=77, nfeatures=1500, nlevels=6):
    """Frames whose contrast is scaled down: most detection cells find nothing at iniThFAST and run cv::FAST a second time at
    minThFAST (ORBextractor.cc:832-846) - the second pass of k_fast_cells."""
    img = synth.Sequence(seq, w, h, n_frames=1).frame(0)
    img = np.clip(img.astype(np.float32) * contrast + 90, 0, 255).astype(np.uint8)
    ex = F.ORBextractor(nfeatures, 1.2, nlevels, ini, mn, w, h, lib=lib)
    orc = O.Extractor(nfeatures, 1.2, nlevels, ini, mn)
    kps, desc, mono = ex(img)
    okps, odesc, omono = orc(img)
    assert_keypoints_equal(kps, okps, "low contrast")
    assert np.array_equal(desc, odesc) and mono == omono
    for l in range(nlevels):
        c, oc = ex.level_candidates(l), orc.level_candidates(l)
        assert len(c) == len(oc), "level %d: %d candidates vs %d" % (l, len(c), len(oc))
        for f in ("x", "y", "response"):
            assert np.array_equal(c[f], oc[f]), "level %d candidate %s" % (l, f)
    ex.close()
    return len(kps)


def check_depth_sparse(lib, dev=None, w=310, h=94, kernel=(F.KERNEL_DIAMOND, 5, 7), batch=3, cap=192, seed=5):
    """rgbl_depth_set_sparse: no dense ProcessedDepthMap unless a call asks for it - the gather evaluates the inverse dilation
    at the keypoints' pixels (k_gather_depth_sparse).  mvDepth / mvuRight must be what sampling the oracle's dense map gives,
    keypoints on the image border included (taps outside the image); a call that asks for the map still gets it.
    dev: torch device for the product library; None = the emulator, whose 'device' pointers are host pointers."""
    import ctypes as C
    K = synth.KITTI_K.copy()
    K[0, 2], K[1, 2] = w / 2.0, h / 2.0
    K[0, 0] = K[1, 1] = 718.856 * w / synth.KITTI_W
    proj = F.projection_matrix(K, synth.KITTI_TR, lib)
    scans = [synth.lidar_scan(seed + i, n_rings=32, n_az=400) for i in range(batch)]
    n = scans[0].shape[1]
    shape, ku, kv = kernel
    dm = F.DepthModule(proj, w, h, kernel_type=shape, kernel_size_u=ku, kernel_size_v=kv, max_points=n, max_keypoints=cap,
                       max_batch=batch, lib=lib)
    dm.SetSparseUpsampling(True)
    dm.profile(True)
    P = O.make_depth_params(proj, kernel=O.structuring_element(shape, ku, ku if shape == F.KERNEL_DIAMOND else kv))
    if dev is not None:
        import torch
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        ptr = lambda t: C.c_void_p(t.data_ptr())
        down = lambda t: t.cpu().numpy()
    else:
        up = lambda a: np.ascontiguousarray(a).copy()
        ptr = lambda a: C.c_void_p(a.ctypes.data)
        down = lambda a: a
    rng = np.random.default_rng(seed)
    cloud = np.stack(scans)
    kps = np.zeros((batch, cap), O.KP_DTYPE)
    kps["x"] = rng.uniform(0, w - 1, (batch, cap)).astype(np.float32)
    kps["y"] = rng.uniform(0, h - 1, (batch, cap)).astype(np.float32)
    kps["x"][:, :8] = np.array([0, 0.5, 1, 2, w - 1, w - 1.5, w - 3, w // 2], np.float32)   # columns at the border
    kps["y"][:, 8:16] = np.array([0, 0.5, 1, 2, h - 1, h - 1.5, h - 3, h // 2], np.float32)  # rows at the border
    kps["x"][:, 16], kps["y"][:, 16] = 0, 0
    kps["x"][:, 17], kps["y"][:, 17] = w - 1, h - 1
    cnt = np.array([cap - 7 * b for b in range(batch)], np.int32)
    d_cloud, d_kp, d_n = up(cloud), up(kps.view(np.float32).reshape(batch, cap, 7)), up(cnt)
    expect = [O.depth(P, cloud[b], w, h, np.stack([kps["x"][b], kps["y"][b]], 1), kps["x"][b]) for b in range(batch)]
    assert sum(int((e[0] > 0).sum()) for e in expect) > cap // 4
    for want_map in (False, True, False):
        d_depth, d_ur = up(np.zeros((batch, cap), np.float32)), up(np.zeros((batch, cap), np.float32))
        d_proc = up(np.zeros((batch, h, w), np.float32)) if want_map else None
        L.check(lib, lib.rgbl_depth_batch_device(dm.h, ptr(d_cloud), batch, n, n, 4 * n, w, h, ptr(d_kp), ptr(d_n), cap, None,
                                                 ptr(d_depth), ptr(d_ur), ptr(d_proc) if want_map else None))
        L.check(lib, lib.rgbl_depth_sync(dm.h))
        for b in range(batch):
            od, our, oraw, oproc = expect[b]
            k = int(cnt[b])
            assert np.array_equal(bits(down(d_depth)[b][:k]), bits(od[:k])), "mvDepth, frame %d, map %s" % (b, want_map)
            assert np.array_equal(bits(down(d_ur)[b][:k]), bits(our[:k])), "mvuRight, frame %d, map %s" % (b, want_map)
            if want_map:
                assert np.array_equal(bits(down(d_proc)[b]), bits(oproc)), "processed map, frame %d" % b
    names = {k: v[1] for k, v in dm.profile_read().items()}
    assert names.get("k_gather_depth_sparse") == 2 and names.get("k_gather_depth") == 1 and names.get("k_inverse_dilate") == 1, names
    # the host entry points: without the maps (sparse), with them (dense), and a sparse prefetch whose map is wanted after all
    hk = np.stack([kps["x"][0], kps["y"][0]], 1)
    od, our, oraw, oproc = expect[0]
    dm.CalculateDepthFromPcd(hk, hk, scans[0], w, h, want_maps=False)
    assert np.array_equal(bits(dm.mvDepth), bits(od)) and np.array_equal(bits(dm.mvuRight), bits(our)), "host call, sparse"
    dm.CalculateDepthFromPcd(hk, hk, scans[0], w, h)
    assert np.array_equal(bits(dm.ProcessedDepthMap), bits(oproc)) and np.array_equal(bits(dm.RawDepthMap), bits(oraw))
    assert np.array_equal(bits(dm.mvDepth), bits(od)), "host call with the maps"
    dm.PrefetchPointcloud(scans[0], w, h)
    dep, ur, proc = np.zeros(cap, np.float32), np.zeros(cap, np.float32), np.zeros((h, w), np.float32)
    un = np.ascontiguousarray(hk[:, 0])
    L.check(lib, lib.rgbl_depth_compute(dm.h, L.ptr(scans[0]), n, scans[0].strides[0] // 4, w, h, L.ptr(hk), L.ptr(un), cap,
                                        L.ptr(dep), L.ptr(ur), None, L.ptr(proc)))
    assert np.array_equal(bits(proc), bits(oproc)) and np.array_equal(bits(dep), bits(od)), "prefetch, then the map after all"
    dm.PrefetchPointcloud(scans[0], w, h)
    L.check(lib, lib.rgbl_depth_compute(dm.h, L.ptr(scans[0]), n, scans[0].strides[0] // 4, w, h, L.ptr(hk), L.ptr(un), cap,
                                        L.ptr(dep), L.ptr(ur), None, None))
    assert np.array_equal(bits(dep), bits(od)) and np.array_equal(bits(ur), bits(our)), "prefetch, sparse gather"
    dm.close()


def check_extractor_dense_corners(lib, w=114, h=80):
    """Four out of five pixels are FAST corners at threshold 1 (a sum of two cosines): in the 41 x 48 px cells of this frame
    more corners than k_fast_cells' per-wave lists hold - every pixel is scored, both arcs, and the NMS walks all pixels
    instead of a corner list (the path no natural image takes)."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    rng = np.random.default_rng(3)
    for period, th in ((12, 1), (10, 2)):   # ~1 590 resp. ~1 470 corners in each of the two cells; +-2 of noise against score plateaus
        img = (127 + 60 * (np.cos(2 * np.pi * xx / period) + np.cos(2 * np.pi * yy / period)) + rng.integers(-2, 3, (h, w))).clip(0, 255).astype(np.uint8)
        ex = F.ORBextractor(500, 1.2, 1, th, th, w, h, lib=lib)
        orc = O.Extractor(500, 1.2, 1, th, th)
        kps, desc, mono = ex(img)
        okps, odesc, omono = orc(img)
        assert len(okps) > 50
        assert_keypoints_equal(kps, okps, "dense corners, period %d" % period)
        assert np.array_equal(desc, odesc)
        c, oc = ex.level_candidates(0), orc.level_candidates(0)
        assert len(c) == len(oc) and all(np.array_equal(c[f], oc[f]) for f in ("x", "y", "response"))
        ex.close()


def check_extractor_threshold_extremes(lib, w=200, h=150):
    """FAST thresholds at the ends of the range on saturated images: the pre-screen's bounds v - t / v + t leave 0 .. 255 (they
    are saturated 16-bit halves in k_fast_cells) and scores reach 254."""
    rng = np.random.default_rng(11)
    blocks = (rng.integers(0, 2, (h // 5 + 1, w // 5 + 1)) * 255).astype(np.uint8).repeat(5, 0).repeat(5, 1)[:h, :w]
    salt = np.where(rng.random((h, w)) > 0.93, 255, 0).astype(np.uint8)
    ramp = (np.add.outer(np.arange(h), np.arange(w)) % 256).astype(np.uint8)
    ramp[rng.random((h, w)) > 0.97] = 255
    ramp[rng.random((h, w)) > 0.97] = 0
    for ini, mn in ((255, 255), (255, 1), (254, 200), (128, 127), (1, 1)):
        ex = F.ORBextractor(400, 1.2, 3, ini, mn, w, h, lib=lib)
        orc = O.Extractor(400, 1.2, 3, ini, mn)
        for name, img in (("blocks", blocks), ("salt", salt), ("ramp", ramp)):
            kps, desc, mono = ex(img)
            okps, odesc, omono = orc(img)
            assert_keypoints_equal(kps, okps, "%s at FAST %d / %d" % (name, ini, mn))
            assert np.array_equal(desc, odesc)
            for l in range(3):
                c, oc = ex.level_candidates(l), orc.level_candidates(l)
                assert len(c) == len(oc) and all(np.array_equal(c[f], oc[f]) for f in ("x", "y", "response")), (name, ini, mn, l)
        ex.close()


# ---- frames resident on the device (rgbl_device_frame): every matcher must give the host-array result ---------------------
def check_device_frames(lib, n=900, seed=17, w=640, h=360, nfeatures=700):
    """(i) upload / download round trip; (ii) capture straight from the extractor + depth handles == the arrays the host API
    returned (incl. Frame::UndistortKeyPoints on the device); (iii) SearchForTriangulation, both SearchByBoW overloads, the three
    greedy projection searches, the Fuse / Sim3 per-point searches and ORBVocabulary::transform with resident frames == the
    same calls on host arrays == the oracle."""
    rng = np.random.default_rng(seed)
    # (i)
    kf1, kf2, K, R, t, ep, sf, s2 = make_triangulation_case(n, seed, n_nodes=60)
    d1 = F.DeviceFrame(n, lib=lib).upload(kf1["desc"], kf1["xy"], kf1["octave"], kf1["uright"])
    d2 = F.DeviceFrame(n + 100, lib=lib).upload(kf2["desc"], kf2["xy"], kf2["octave"], kf2["uright"])
    back = d2.download()
    assert len(d2) == n and np.array_equal(back["desc"], kf2["desc"]) and np.array_equal(bits(back["xy"]), bits(kf2["xy"]))
    assert np.array_equal(back["octave"], kf2["octave"]) and np.array_equal(bits(back["uright"]), bits(kf2["uright"]))
    mono = F.DeviceFrame(8, lib=lib).upload(kf1["desc"][:5], kf1["xy"][:5], kf1["octave"][:5], None)
    assert (mono.download()["uright"] == -1).all()
    mono.close()
    try:
        F.DeviceFrame(4, lib=lib).upload(kf1["desc"][:70], kf1["xy"][:70], kf1["octave"][:70])
        raise AssertionError("more features than the capacity must be refused")
    except L.RgblError as e:
        assert e.code == L.ERR_INVALID
    # (iii) key-frame views: triangulation with and without the resident FeatureVector, SearchByBoW x 2
    hollow = lambda kf: dict(kf, desc=np.zeros_like(kf["desc"]), xy=np.zeros_like(kf["xy"]), octave=np.zeros_like(kf["octave"]),
                             uright=np.zeros_like(kf["uright"]))   # host arrays that must NOT be read
    for with_fv in (False, True):
        if with_fv:
            d1.set_feature_vector(kf1["node_off"], kf1["node_feat"])
            d2.set_feature_vector(kf2["node_off"], kf2["node_feat"])
        for check_ori in (False, True):
            m = F.ORBmatcher(0.6, check_ori, lib=lib)
            Fm = m.fundamental(K, K, R, t)
            for only_stereo, coarse in ((False, False), (True, False), (False, True)):
                want = O.search_triangulation(kf1, kf2, Fm, ep, sf, s2, only_stereo, coarse, check_ori)
                for a, b in ((dict(hollow(kf1), device=d1), dict(hollow(kf2), device=d2)), (dict(hollow(kf1), device=d1), kf2), (kf1, dict(hollow(kf2), device=d2))):
                    _, nm, m12 = m.SearchForTriangulation(a, b, Fm, ep, sf, s2, only_stereo, coarse)
                    assert nm == want[1] and np.array_equal(m12, want[0]), "triangulation on resident frames"
            m.close()
    a = dict(kf1, has_mp=(rng.random(n) < 0.7).astype(np.uint8))
    b = dict(kf2, has_mp=(rng.random(n) < 0.8).astype(np.uint8))
    for ori in (False, True):
        mt = F.ORBmatcher(0.7, ori, lib=lib)
        want = O.search_by_bow(a, b, 0.7, ori)
        got = mt.SearchByBoW(dict(hollow(a), device=d1), dict(hollow(b), device=d2))
        assert got[1] == want[1] and np.array_equal(got[0], want[0]), "SearchByBoW on resident frames"
        want = O.search_by_bow_kf(a, b, 0.7, ori)
        got = mt.SearchByBoWKeyFrames(dict(hollow(a), device=d1), dict(hollow(b), device=d2))
        assert got[1] == want[1] and np.array_equal(got[0], want[0]), "SearchByBoW(KF, KF) on resident frames"
        mt.close()
    d1.close()
    d2.close()
    # the projection searches: the CurrentFrame / key frame resident, the map points from the host
    def frame2(c, ur=True, grid=True):
        f = F.DeviceFrame(len(c["kp2_xy"]), lib=lib)
        f.upload(c["desc2"], c["kp2_xy"], c["kp2_octave"], c["uright2"] if ur and "uright2" in c else None)
        return f.set_grid(c["grid"]) if grid else f   # with its own AssignFeaturesToGrid: the calls skip their grid build

    def hollow2(c):
        z = {k: np.zeros_like(c[k]) for k in ("kp2_xy", "kp2_octave", "desc2") if k in c}
        if "uright2" in c:
            z["uright2"] = np.zeros_like(c["uright2"])
        return dict(c, **z)
    case = make_projection_case(n, n + 50, seed + 1, "forward")
    mt = F.ORBmatcher(0.9, True, lib=lib)
    for with_grid in (False, True):
        f2 = frame2(case, grid=with_grid)
        for th in (7.0, 15.0):
            want = O.search_by_projection(case, th, False, True)
            got = mt.SearchByProjection(dict(hollow2(case), device2=f2), th, False)
            assert got[1] == want[1] and np.array_equal(got[0], want[0]), "SearchByProjection on a resident CurrentFrame"
        if with_grid:   # a grid for other image bounds is not used: the call builds its own
            other = dict(case, grid=np.array([0, 0, 1300, 400, 64 / 1300.0, 48 / 400.0], np.float32))
            want = O.search_by_projection(other, 7.0, False, True)
            got = mt.SearchByProjection(dict(hollow2(other), device2=f2), 7.0, False)
            assert got[1] == want[1] and np.array_equal(got[0], want[0]), "SearchByProjection with a grid of other bounds"
        f2.close()
    case = make_relocalization_case(n, n + 50, seed + 2)
    valid, level = relocalization_prepass(case)
    f2 = frame2(case, ur=False)
    want = O.search_by_projection_kf(case, 10.0, 100, True)
    got = mt.SearchByProjectionKeyFrame(dict(hollow2(case), valid1=valid, level1=level, device2=f2), 10.0, 100)
    assert got[1] == want[1] and np.array_equal(got[0], want[0]), "SearchByProjection(F, KF) on a resident frame"
    f2.close()
    mt.close()
    case = make_local_points_case(n + 300, n, seed + 3)
    mt = F.ORBmatcher(0.8, True, lib=lib)
    want = O.search_local_points(case, 1.0, 0.8)
    for with_grid in (False, True):
        f2 = frame2(case, grid=with_grid)
        got = mt.SearchLocalPoints(dict(hollow2(case), device2=f2), 1.0)
        assert got[1] == want[1] and np.array_equal(got[0], want[0]), "SearchLocalPoints on a resident frame"
        f2.close()
    mt.close()
    case = make_fuse_case(n + 200, n, seed + 4)
    valid, level = fuse_prepass(case)
    c = dict(case, valid1=valid, level1=level)
    f2 = frame2(c)
    mt = F.ORBmatcher(0.6, True, lib=lib)
    want = mt.FuseSearch(c, 3.0)
    got = mt.FuseSearch(dict(hollow2(c), device2=f2), 3.0)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and (want[0] >= 0).sum() > 30, "Fuse search on a resident key frame"
    f2.close()
    c = make_project_search_case(n + 200, n, seed + 5)
    f2 = frame2(c, ur=False)
    for form, maxd in ((0, 50), (1, 100)):
        want = mt.ProjectSearch(c, 4.0, form, maxd)
        got = mt.ProjectSearch(dict(hollow2(c), device2=f2), 4.0, form, maxd)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), "project search on a resident key frame"
    matched2 = (rng.random(n) < 0.1).astype(np.uint8)
    want = mt.SearchByProjectionSim3(c, matched2, 8, 0, 75)
    got = mt.SearchByProjectionSim3(dict(hollow2(c), device2=f2), matched2, 8, 0, 75)
    assert got[1] == want[1] and np.array_equal(got[0], want[0]), "SearchByProjection(KF, Sim3) on a resident key frame"
    f2.close()
    mt.close()
    # (ii) capture: extractor (+ depth module) -> resident frame, no host arrays in between
    sq = synth.Sequence(seed, w, h, n_frames=2)
    ex = F.ORBextractor(nfeatures, 1.2, 8, 12, 7, w, h, lib=lib)
    Kc = synth.KITTI_K.copy()
    Kc[0, 2], Kc[1, 2] = w / 2.0, h / 2.0
    dm = F.DepthModule(F.projection_matrix(Kc, synth.KITTI_TR, lib), w, h, max_keypoints=ex.max_keypoints, lib=lib)
    cap = F.DeviceFrame(ex.max_keypoints, lib=lib)
    for i in range(2):
        kps, desc, _ = ex(sq.frame(i))
        dm.CalculateDepthFromPcd(kps, kps, synth.lidar_scan(seed + i, n_az=600), w, h, want_maps=False)
        cap.capture(ex, len(kps), dm)
        got = cap.download()
        assert len(cap) == len(kps) and np.array_equal(got["desc"], desc)
        assert np.array_equal(bits(got["xy"]), bits(np.stack([kps["x"], kps["y"]], 1))) and np.array_equal(got["octave"], kps["octave"])
        assert np.array_equal(bits(got["uright"]), bits(dm.mvuRight)) and (dm.mvuRight > 0).sum() > 10
    cap.capture(ex, len(kps), None)          # monocular: no depth handle
    assert (cap.download()["uright"] == -1).all()
    k4 = np.array([Kc[0, 0], Kc[1, 1], Kc[0, 2], Kc[1, 2]], np.float32)
    dist = np.array([-0.28, 0.07, 1e-4, -2e-4, 0.0], np.float32)
    cap.capture(ex, len(kps), None, K=k4, dist=dist)    # Frame::UndistortKeyPoints on the way
    want_xy = ex.UndistortKeyPoints(np.stack([kps["x"], kps["y"]], 1), k4, dist)
    assert np.array_equal(bits(cap.download()["xy"]), bits(want_xy)) and not np.array_equal(want_xy, np.stack([kps["x"], kps["y"]], 1))
    # a captured frame in a matcher call + ORBVocabulary::transform of its resident descriptors
    cap.capture(ex, len(kps), dm)
    frame = dict(xy=np.stack([kps["x"], kps["y"]], 1), desc=desc, octave=kps["octave"], angle=kps["angle"], uright=dm.mvuRight)
    case = make_local_points_case(1200, seed=seed + 6, w=w, h=h, frame2=frame)
    cap.set_grid(case["grid"])
    mt = F.ORBmatcher(0.8, True, lib=lib)
    want = O.search_local_points(case, 3.0, 0.8)
    got = mt.SearchLocalPoints(dict(hollow2(case), device2=cap), 3.0)
    assert got[1] == want[1] and np.array_equal(got[0], want[0]) and want[1] > 50, "SearchLocalPoints on a captured frame"
    mt.close()
    voc = synth.make_vocabulary(8, 3, seed)
    V = F.ORBVocabulary(lib=lib).from_arrays(synth.vocabulary_arrays(voc))
    want = V.transform(desc, 2)
    got = V.prepare_transform_frame(cap, 2)()
    for g, x in zip(got, want):
        assert np.array_equal(g.view(np.uint64) if g.dtype == np.float64 else g, x.view(np.uint64) if x.dtype == np.float64 else x)
    V.close()
    cap.close()
    dm.close()
    ex.close()
    return True


def check_switches(lib, w=400, h=300, nfeatures=500, batch=8, subset=None):
    """Every RGBL_* tuning switch of include/rgbl_frontend.h that changes a launch path (they are read when a handle is
    created): the batch extraction, the host-pointer extraction and the Hamming scan must stay bit-identical under each
    (the header says so, so it is checked)."""
    import os
    settings = [{"RGBL_SPLIT_PYR": "0"}, {"RGBL_SPLIT_PYR": "3"}, {"RGBL_GAUSS_BS": "256"}, {"RGBL_XCD_MAP": "0"}, {"RGBL_DENSE": "0"},
                {"RGBL_COMPACT": "0"}, {"RGBL_FAST_BS": "128"}, {"RGBL_OCTREE_HIST": "0"}, {"RGBL_OCTREE_WG": "512"}, {"RGBL_GRAPH": "0"},
                {"RGBL_LEVEL_SPLIT": "0"}, {"RGBL_BF_SPLIT": "0"}, {"RGBL_BF_MFMA": "0"}, {"RGBL_BF_MFMA": "i8"}]
    if subset is not None:   # the emulator runs a representative half (the CPU suite's time), the MI355X all of them
        settings = [e for e in settings if list(e.items())[0] in subset]
    for env in settings:
        saved = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            if any(k.startswith("RGBL_BF") for k in env):
                check_matcher_bf(lib, 500, 450, seed=9)
            else:
                check_extractor_batch(lib, w, h, nfeatures, batch)
                check_extractor(lib, w, h, nfeatures, frames=(0,))
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    # the batch path with per-kernel profiling on (one stream, no pyramid split)
    ex = F.ORBextractor(nfeatures, 1.2, 8, 12, 7, w, h, max_batch=batch, lib=lib)
    ex.profile(True)
    s = synth.Sequence(2, w, h, n_frames=batch)
    imgs = np.stack([s.frame(i) for i in range(batch)])
    orc = O.Extractor(nfeatures, 1.2, 8, 12, 7)
    for i, (kps, desc, mono) in enumerate(ex.extract_batch(imgs)):
        okps, odesc, omono = orc(imgs[i])
        assert_keypoints_equal(kps, okps, "profiled batch frame %d" % i)
        assert np.array_equal(desc, odesc) and mono == omono
    ex.close()
    return len(settings)
