"""Known-answer tests that pin the CPU oracle.

The reference ships no tests / golden vectors for this path and cannot be built here (PARITY UNPINNED, see
oracle/oracle.h), so the oracle is pinned by hand-derived answers and by independent numpy restatements of the
documented arithmetic (SURVEY.md §8(c) lists the cases).  The GPU path is then held bit-exact to the oracle."""
import math

import numpy as np

from oracle import oracle_py as O
from orb_slam3_rgbl_amd import synth

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0),
        (-3, 1), (-2, 2), (-1, 3)]


def test_tables_match_survey_appendix_b():
    t = O.Extractor(2000, 1.2, 8, 12, 7).tables()
    assert list(t["per_level"]) == [434, 362, 302, 251, 209, 175, 145, 122]
    assert list(O.Extractor(1000, 1.2, 8, 12, 7).tables()["per_level"]) == [217, 181, 151, 126, 105, 87, 73, 60]
    assert list(O.Extractor(8000, 1.2, 8, 12, 7).tables()["per_level"]) == [1737, 1448, 1207, 1005, 838, 698, 582, 485]
    assert list(t["umax"]) == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert sum(2 * int(u) + 1 for u in t["umax"][1:]) * 2 + 31 == 749  # pixels of the circular patch (SURVEY says 789: a slip)
    ex = O.Extractor(2000, 1.2, 8, 12, 7)
    ex(np.zeros((376, 1241), np.uint8))
    assert [ex.level_size(l) for l in range(8)] == [(1241, 376), (1034, 313), (862, 261), (718, 218), (598, 181),
                                                    (499, 151), (416, 126), (346, 105)]


def test_cv_round_is_half_to_even():
    assert [O.cv_round(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 2.5001)] == [0, 2, 2, 0, -2, 2, 3]


def test_hamming_identities():
    z, o = np.zeros(32, np.uint8), np.full(32, 255, np.uint8)
    assert O.descriptor_distance(z, z) == 0 and O.descriptor_distance(z, o) == 256
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (50, 32), dtype=np.uint8)
    b = rng.integers(0, 256, (50, 32), dtype=np.uint8)
    for i in range(50):
        assert O.descriptor_distance(a[i], b[i]) == int(np.unpackbits(a[i] ^ b[i]).sum())
    t = z.copy()
    t[13] = 0x10
    assert O.descriptor_distance(z, t) == 1


def blur_numpy(img):
    k = np.array([18, 34, 48, 56, 48, 34, 18], np.int64)
    p = np.pad(img.astype(np.int64), 3, mode="reflect")  # numpy 'reflect' == BORDER_REFLECT_101
    h = sum(k[i] * p[:, i:i + img.shape[1]] for i in range(7))
    v = sum(k[i] * h[i:i + img.shape[0], :] for i in range(7))
    return ((v + 32768) >> 16).astype(np.uint8)


def test_gaussian_blur_known_answers():
    k = np.array([18, 34, 48, 56, 48, 34, 18])
    assert k.sum() == 256
    # the 8.8 kernel is what OpenCV's error-diffusing quantiser makes of exp(-x^2/8)
    g = np.exp(-np.arange(-3, 4) ** 2 / 8.0)
    g = g / g.sum() * 256
    err, q = 0.0, []
    for i in range(3):
        v = int(round(g[i] + err))
        err = g[i] + err - v
        q.append(v)
    assert q == [18, 34, 48] and 256 - 2 * sum(q) == 56
    imp = np.zeros((21, 21), np.uint8)
    imp[10, 10] = 255
    out = O.gaussian_blur7(imp)
    assert np.array_equal(out[7:14, 7:14], ((255 * np.outer(k, k) + 32768) >> 16).astype(np.uint8))
    assert np.array_equal(O.gaussian_blur7(np.full((9, 33), 201, np.uint8)), np.full((9, 33), 201, np.uint8))
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (40, 57), dtype=np.uint8)
    assert np.array_equal(O.gaussian_blur7(img), blur_numpy(img))  # includes the reflect-101 border


def test_resize_known_answers():
    assert np.array_equal(O.resize_linear(np.full((24, 36), 77, np.uint8), 30, 20), np.full((20, 30), 77, np.uint8))
    # one row, hand computed: scale 1.2, dx=0: fx=0.1 -> weights 1843/205; rows are identical so the vertical pass
    # only contributes its (x>>4 ... >>16 ... +2 >>2) rounding chain
    row = np.array([0, 10, 20, 30, 40, 50, 60, 70, 80, 90, 100, 110], np.uint8)
    src = np.tile(row, (12, 1))
    out = O.resize_linear(src, 10, 10)
    exp = []
    for dx in range(10):
        fx = np.float32((dx + 0.5) * (1.0 / (10 / 12)) - 0.5)
        sx = int(math.floor(fx))
        f = np.float32(fx - np.float32(sx))
        a0, a1 = int(np.rint((np.float32(1) - f) * 2048)), int(np.rint(f * 2048))
        exp.append(int(row[sx]) * a0 + int(row[min(sx + 1, 11)]) * a1)
    # vertical weights of dy=0: fy=0.1 -> (1843, 205)
    got = out[0]
    want = [((((1843 * (hv >> 4)) >> 16) + ((205 * (hv >> 4)) >> 16) + 2) >> 2) for hv in exp]
    assert list(got) == want
    assert abs(int(got[0]) - 1) <= 1 and abs(int(got[9]) - 109) <= 1  # ~ (dx+0.5)*1.2-0.5 on a 10/px ramp


def ring_image(center, ring_vals):
    img = np.full((9, 9), center, np.uint8)
    for (dx, dy), v in zip(RING, ring_vals):
        img[4 + dy, 4 + dx] = v
    return img


def test_fast_segment_test_and_score():
    t = 12
    # 9 contiguous brighter pixels by exactly t+1 -> corner at t, not at t+1; score == t
    vals = [100 + t + 1] * 9 + [100] * 7
    img = ring_image(100, vals)
    assert len(O.fast(img, t, nonmax=False)) == 1
    assert len(O.fast(img, t + 1, nonmax=False)) == 0
    assert O.corner_score(img, 4, 4, t) == t
    kp = O.fast(img, t)[0]
    assert (kp["x"], kp["y"], kp["size"], kp["angle"], kp["response"]) == (4.0, 4.0, 7.0, -1.0, float(t))
    # only 8 contiguous -> not a corner
    assert len(O.fast(ring_image(100, [150] * 8 + [100] * 8), t, nonmax=False)) == 0
    # wrap-around arc (ring positions 12..15,0..4) darker: corner, score is the min margin - 1
    vals = [60] * 5 + [100] * 7 + [70, 60, 60, 60]
    img = ring_image(100, vals)
    assert len(O.fast(img, 20, nonmax=False)) == 1
    assert O.corner_score(img, 4, 4, 20) == 29  # weakest member differs by 30
    # strictness: ring == v + t is NOT brighter
    assert len(O.fast(ring_image(100, [100 + t] * 16), t, nonmax=False)) == 0


def test_fast_nms_plateau_and_two_threshold_equivalence():
    rng = np.random.default_rng(3)
    img = synth.Sequence(4, 96, 80, 1).frame(0)
    lo, hi = O.fast(img, 7), O.fast(img, 12)
    # (ii) of SURVEY A.3: kept at iniTh == kept at minTh with score >= iniTh
    sel = lo[lo["response"] >= 12]
    assert len(sel) == len(hi) and np.array_equal(sel["x"], hi["x"]) and np.array_equal(sel["y"], hi["y"])
    assert len(lo) > 20
    # the score of a corner does not depend on the threshold it was detected with
    for kp in hi[:20]:
        x, y = int(kp["x"]), int(kp["y"])
        assert O.corner_score(img, x, y, 12) == O.corner_score(img, x, y, 7) == int(kp["response"])
    # emission order is row-major
    order = lo["y"] * 1000 + lo["x"]
    assert np.all(np.diff(order) > 0)
    # plateau: two adjacent corners with equal scores suppress each other (strict '>')
    a = np.full((12, 14), 100, np.uint8)
    for cx in (5, 6):
        for (dx, dy) in RING[:9]:
            a[5 + dy, cx + dx] = 160
    raw = O.fast(a, 20, nonmax=False)
    nms = O.fast(a, 20, nonmax=True)
    pts = {(int(k["x"]), int(k["y"])): O.corner_score(a, int(k["x"]), int(k["y"]), 20) for k in raw}
    for (x, y), s in pts.items():
        nb = [pts.get((x + i, y + j), 0) for i in (-1, 0, 1) for j in (-1, 0, 1) if (i, j) != (0, 0)]
        kept = any(int(k["x"]) == x and int(k["y"]) == y for k in nms)
        assert kept == all(s > v for v in nb)
    del rng


def test_fast_atan2():
    for y, x, deg in ((0, 1, 0), (1, 0, 90), (0, -1, 180), (-1, 0, 270), (1, 1, 45), (1, -1, 135), (-1, -1, 225), (-1, 1, 315)):
        assert abs(O.fast_atan2(y, x) - deg) < 0.3
    assert O.fast_atan2(0, 0) == 0.0
    rng = np.random.default_rng(2)
    for _ in range(2000):
        y, x = rng.integers(-3000000, 3000000, 2)
        true = math.degrees(math.atan2(y, x)) % 360
        got = O.fast_atan2(float(y), float(x))
        assert min(abs(got - true), 360 - abs(got - true)) < 0.3
        assert 0 <= got <= 360


def test_ic_angle_points_to_the_bright_side():
    img = np.zeros((41, 41), np.uint8)
    img[:, 21:] = 200  # bright on +x: centroid angle ~ 0
    assert min(O.ic_angle(img, 20, 20), 360 - O.ic_angle(img, 20, 20)) < 1.0
    img = np.zeros((41, 41), np.uint8)
    img[21:, :] = 200  # bright on +y (down): 90 degrees
    assert abs(O.ic_angle(img, 20, 20) - 90) < 1.0


def test_brief_bit_layout():
    # left half dark / right half bright, angle 0: bit k of byte i is pattern pair 8i+k: I(x0,y0) < I(x1,y1)
    from oracle.oracle_py import lib  # noqa: F401
    img = np.zeros((41, 41), np.uint8)
    img[:, 21:] = 200
    d = O.brief(img, 20, 20, 0.0)
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "orb_slam3_rgbl_amd", "csrc", "brief_pattern.h")).read()
    body = txt[txt.index("kBriefPattern[256 * 4] = {") + len("kBriefPattern[256 * 4] = {"):txt.rindex("};")]
    pat = [int(v) for v in re.findall(r"-?\d+", body)]
    assert len(pat) == 1024
    for p in range(256):
        x0, y0, x1, y1 = pat[4 * p:4 * p + 4]
        assert ((d[p // 8] >> (p % 8)) & 1) == int(img[20 + y0, 20 + x0] < img[20 + y1, 20 + x1])


def test_octree_hand_built_cases():
    def kp(x, y, r):
        a = np.zeros(1, O.KP_DTYPE)
        a["x"], a["y"], a["response"] = x, y, r
        return a

    # region 100x50 -> 2 roots; root 0 holds two keys, root 1 one (a leaf). N=3: root 0 is split, its children are
    # pushed to the FRONT of the list (n1 then n4), so the output order is n4, n1, root1.
    cand = np.concatenate([kp(10, 10, 5), kp(40, 40, 9), kp(60, 10, 3)])
    out = O.distribute_octree(cand, 16, 116, 16, 66, 3)
    assert [(int(k["x"]), int(k["y"])) for k in out] == [(40, 40), (10, 10), (60, 10)]
    # 50x50 -> one root with 3 keys, two of them in the same quadrant; N=2 stops after the first split and the
    # shared node keeps the FIRST of two equally strong keys
    cand = np.concatenate([kp(5, 5, 7), kp(8, 8, 7), kp(40, 40, 1)])
    out = O.distribute_octree(cand, 16, 66, 16, 66, 2)
    assert [(int(k["x"]), int(k["y"])) for k in out] == [(40, 40), (5, 5)]
    cand = np.concatenate([kp(5, 5, 6), kp(8, 8, 7), kp(40, 40, 1)])
    out = O.distribute_octree(cand, 16, 66, 16, 66, 2)
    assert [(int(k["x"]), int(k["y"])) for k in out] == [(40, 40), (8, 8)]
    # no candidates
    assert len(O.distribute_octree(np.zeros(0, O.KP_DTYPE), 16, 116, 16, 66, 5)) == 0
    # never returns more than one key per node and at least min(N, #distinct) keys
    rng = np.random.default_rng(5)
    pts = np.unique(rng.integers(0, [300, 90], (400, 2)), axis=0)
    cand = np.zeros(len(pts), O.KP_DTYPE)
    cand["x"], cand["y"], cand["response"] = pts[:, 0], pts[:, 1], rng.integers(7, 100, len(pts))
    out = O.distribute_octree(cand, 16, 316, 16, 106, 120)
    assert 120 <= len(out) <= 123
    assert len({(k["x"], k["y"]) for k in out}) == len(out)


def test_depth_projection_and_inverse_dilation_known_answers():
    w, h = 40, 30
    K = np.array([[20, 0, 20, 0], [0, 20, 15, 0], [0, 0, 1, 0]], np.float32)
    proj = O.projection_matrix(K, np.eye(4, dtype=np.float32))
    assert np.array_equal(proj, K)
    P = O.make_depth_params(proj)
    kp = np.array([[20, 15], [22, 15], [23, 15], [0, 0]], np.float32)
    # a single point at depth 10 straight ahead lands on (20, 15); Diamond-5 spreads it over 13 pixels
    cloud = np.array([[0.0], [0.0], [10.0], [1.0]], np.float32)
    d, ur, raw, proc = O.depth(P, cloud, w, h, kp, kp[:, 0])
    assert raw[15, 20] == 10.0 and (raw > 0).sum() == 1
    fp = O.structuring_element(3, 5, 5)
    assert fp.sum() == 13
    ys, xs = np.nonzero(proc)
    assert len(ys) == 13 and all(fp[y - 13, x - 18] for y, x in zip(ys, xs))
    val = np.float32(200) - (np.float32(200) - np.float32(10))
    assert np.all(proc[ys, xs] == val)
    assert d[0] == val and d[1] == val and d[2] == -1 and d[3] == -1
    assert ur[0] == np.float32(20) - np.float32(100) / val
    # two overlapping footprints: the NEARER depth wins (max of 200 - d)
    cloud = np.array([[0.0, 1.0], [0.0, 0.0], [10.0, 20.0], [1.0, 1.0]], np.float32)  # second lands on (21, 15)
    d, ur, raw, proc = O.depth(P, cloud, w, h, kp, kp[:, 0])
    assert raw[15, 21] == 20.0
    assert proc[15, 21] == np.float32(200) - np.float32(190)  # the 10 m neighbour dominates
    assert proc[15, 23] == np.float32(200) - np.float32(180)  # only the 20 m point reaches here
    # three points on one pixel: the last in file order wins
    cloud = np.array([[0, 0, 0], [0, 0, 0], [30, 10, 20], [1, 1, 1]], np.float32)
    assert O.depth(P, cloud, w, h, kp, kp[:, 0])[2][15, 20] == 20.0


def test_structuring_elements():
    assert O.structuring_element(0, 3, 2).tolist() == [[1, 1, 1], [1, 1, 1]]
    assert O.structuring_element(1, 3, 3).tolist() == [[0, 1, 0], [1, 1, 1], [0, 1, 0]]
    assert O.structuring_element(3, 3, 3).tolist() == [[0, 1, 0], [1, 1, 1], [0, 1, 0]]
    e = O.structuring_element(2, 5, 5)
    assert e[2].tolist() == [1, 1, 1, 1, 1] and e[0].tolist() == [0, 0, 1, 0, 0]
    assert O.structuring_element(3, 9, 9).sum() == 41


def test_triangulation_tie_goes_to_the_later_candidate():
    d1 = np.zeros((1, 32), np.uint8)
    d2 = np.zeros((3, 32), np.uint8)
    d2[0, 0] = 1  # distance 1
    kf1 = dict(desc=d1, xy=[[100, 100]], octave=[0], angle=[0], uright=[-1], has_mp=[0], node_id=[7], node_off=[0, 1], node_feat=[0])
    kf2 = dict(desc=d2, xy=[[90, 100], [90, 100], [90, 100]], octave=[0, 0, 0], angle=[0, 0, 0], uright=[-1, -1, -1],
               has_mp=[0, 0, 0], node_id=[7], node_off=[0, 3], node_feat=[0, 1, 2])
    F = np.zeros(9, np.float32)
    sf = np.ones(8, np.float32)
    m, n = O.search_triangulation(kf1, kf2, F, [1e6, 1e6], sf, sf, coarse=True)
    assert n == 1 and m[0] == 2  # candidates 1 and 2 tie at distance 0: `dist > bestDist` keeps the later one
    kf2["has_mp"] = [0, 0, 1]
    m, n = O.search_triangulation(kf1, kf2, F, [1e6, 1e6], sf, sf, coarse=True)
    assert m[0] == 1
    # den == 0 (F = 0) rejects every candidate when the epipolar test is on
    m, n = O.search_triangulation(kf1, kf2, F, [1e6, 1e6], sf, sf, coarse=False)
    assert n == 0 and m[0] == -1
    # epipole guard: mono-mono candidates closer than 10*sqrt(scale) px to the epipole are skipped
    m, n = O.search_triangulation(kf1, kf2, F, [92, 100], sf, sf, coarse=True)
    assert n == 0


def test_cvtcolor_gray_known_answers():
    """cv::cvtColor 8-bit RGB/BGR(A) -> gray, OpenCV 4.x: (9798 R + 19235 G + 3735 B + 2^14) >> 15 (weights sum to 2^15)."""
    assert 9798 + 19235 + 3735 == 1 << 15
    px = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 200, 77], [1, 1, 1], [254, 255, 255]]], np.uint8)
    rgb = O.cvt_gray(px, True)[0]
    bgr = O.cvt_gray(px, False)[0]
    assert rgb.tolist() == [255, 0, 76, 150, 29, (10 * 9798 + 200 * 19235 + 77 * 3735 + 16384) >> 15, 1, 255]
    assert bgr.tolist() == [255, 0, 29, 150, 76, (10 * 3735 + 200 * 19235 + 77 * 9798 + 16384) >> 15, 1, 255]
    # the alpha channel is ignored; rows honour the stride
    rng = np.random.default_rng(1)
    img4 = rng.integers(0, 256, (5, 7, 4), dtype=np.uint8)
    ref = ((img4[..., 0].astype(np.int64) * 9798 + img4[..., 1].astype(np.int64) * 19235 + img4[..., 2].astype(np.int64) * 3735 + 16384) >> 15)
    assert np.array_equal(O.cvt_gray(img4, True), ref.astype(np.uint8))
    assert np.array_equal(O.cvt_gray(img4[..., :3].copy(), True), ref.astype(np.uint8))


def test_kitti_bin_repack():
    pts = np.arange(20, dtype=np.float32).reshape(5, 4)
    cloud = O.kitti_bin_to_cloud(pts)
    assert cloud.shape == (4, 5)
    assert np.array_equal(cloud[:3], pts[:, :3].T) and np.array_equal(cloud[3], np.ones(5, np.float32))


def test_distinctive_descriptor_known_answers():
    """MapPoint::ComputeDistinctiveDescriptors: median = sorted row[(N - 1) // 2] with the row's own zero in it; strict '<'."""
    z = np.zeros(32, np.uint8)

    def d(nbits):  # descriptor at Hamming distance nbits from zero
        v = np.zeros(256, np.uint8); v[:nbits] = 1
        return np.packbits(v, bitorder="little")
    # N = 1: itself; N = 2: medians are row[0] = 0 for both -> first wins
    assert O.distinctive_descriptors([[z], [d(7), z]]).tolist() == [0, 0]
    # N = 3 on a line 0 -- 10 -- 30 (nested bit sets): rows sorted {0,10,30}, {0,10,20}, {0,20,30}; median index 1 -> 10, 10, 20
    assert O.distinctive_descriptors([[d(0), d(10), d(30)]]).tolist() == [0]
    # ... and with the middle one first it still wins by being first among equals
    assert O.distinctive_descriptors([[d(10), d(0), d(30)]]).tolist() == [0]
    # N = 4: median index (4-1)//2 = 1.  points 0, 2, 4, 100: rows {0,2,4,100} {0,2,2,98} {0,2,4,96} {0,96,98,100} -> medians 2,2,2,96
    assert O.distinctive_descriptors([[d(100), d(0), d(2), d(4)]]).tolist() == [1]
    # independent numpy restatement on random lists
    rng = np.random.default_rng(5)
    lists = [rng.integers(0, 256, (n, 32), dtype=np.uint8) for n in (5, 6, 17, 40)]
    want = []
    for L_ in lists:
        bits_ = np.unpackbits(L_, axis=1)
        dist = (bits_[:, None, :] != bits_[None, :, :]).sum(2)
        med = np.sort(dist, axis=1)[:, (len(L_) - 1) // 2]
        want.append(int(np.argmin(med)))
    assert O.distinctive_descriptors(lists).tolist() == want
