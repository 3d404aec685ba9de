"""Randomised device-vs-oracle cases (image sizes / feature counts / level counts / thresholds for the extractor, image sizes
that are no multiples of the dilation tile and every structuring element for the depth module, train-set sizes around the
launch-slice and sweep boundaries for the Hamming scan).  One function = one random case drawn from `rng`, checked bit for
bit against the oracle; used by tests/test_fuzz_gpu.py (fixed seeds, `-m gpu`; RGBL_FUZZ_SECONDS adds time-boxed fresh
seeds) and by tools/gpu_random_*_checks.py."""
import numpy as np

import parity_checks as pc
from oracle import oracle_py as O
from orb_slam3_rgbl_amd import frontend as F
from orb_slam3_rgbl_amd import synth


def extractor_case(lib, rng):
    while True:
        w, h = int(rng.integers(200, 1400)), int(rng.integers(120, 700))
        nlevels = int(rng.integers(1, 9))
        # every level needs 1 .. 16 quad-tree roots (ORBextractor.cc:558: nIni = round(width / height) of the level minus its 16-px frame; the
        # reference divides by zero for portrait levels, the library refuses them at create time) - the borders weigh more on the small levels
        sizes = [(round(w / 1.2 ** l), round(h / 1.2 ** l)) for l in range(nlevels)]
        # (the library rounds a level's size in fp32, cvRound(w * mvInvScaleFactor[l]): a pixel either way moves the ratio by up to 2 %)
        if all(1 <= round((wl - 32) / max(hl - 32, 1) - 0.05) and round((wl - 32) / max(hl - 32, 1) + 0.05) <= 16 for wl, hl in sizes) and \
                min(w, h) / 1.2 ** (nlevels - 1) >= 80:
            break
    nf = int(rng.choice([50, 300, 1000, 2000, 3500, 6000]))
    ini = int(rng.choice([12, 20]))
    total = pc.check_extractor(lib, w, h, nf, frames=(0,), ini=ini, mn=7, nlevels=nlevels, seq=int(rng.integers(0, 1000)), stages=True)
    return "%4dx%-4d levels %d nfeatures %5d ini %2d -> %d keypoints" % (w, h, nlevels, nf, ini, total)


def low_contrast_case(lib, rng):
    """most detection cells find nothing at iniThFAST and take the second cv::FAST pass at minThFAST"""
    w, h = int(rng.integers(300, 1300)), int(rng.integers(200, 500))
    ini, mn = int(rng.choice([12, 20, 30])), int(rng.choice([3, 7, 10]))
    contrast = float(rng.choice([0.08, 0.15, 0.3]))
    img = synth.Sequence(int(rng.integers(0, 1000)), w, h, n_frames=1).frame(0)
    img = np.clip(img.astype(np.float32) * contrast + 90, 0, 255).astype(np.uint8)
    ex = F.ORBextractor(1500, 1.2, 6, ini, mn, w, h, lib=lib)
    orc = O.Extractor(1500, 1.2, 6, ini, mn)
    kps, desc, mono = ex(img)
    okps, odesc, omono = orc(img)
    pc.assert_keypoints_equal(kps, okps, "low contrast %dx%d" % (w, h))
    assert np.array_equal(desc, odesc) and mono == omono
    for l in range(6):
        c, oc = ex.level_candidates(l), orc.level_candidates(l)
        assert len(c) == len(oc) and all(np.array_equal(c[f], oc[f]) for f in ("x", "y", "response")), (w, h, l)
    ex.close()
    return "low contrast %4dx%-4d FAST %2d/%2d x%.2f -> %d keypoints" % (w, h, ini, mn, contrast, len(kps))


def depth_case(lib, rng):
    w, h = int(rng.integers(130, 1400)), int(rng.integers(70, 520))
    shape = int(rng.choice([F.KERNEL_DIAMOND, F.KERNEL_DIAMOND, F.KERNEL_RECT, F.KERNEL_CROSS, F.KERNEL_ELLIPSE]))
    ku, kv = int(rng.choice([3, 5, 7, 9])), int(rng.choice([3, 5, 7, 9]))
    method = int(rng.choice([F.UPS_INVERSE_DILATION] * 3 + [F.UPS_AVERAGE_FILTERING, F.UPS_NEAREST_NEIGHBOR_PIXEL]))
    n = pc.check_depth(lib, method, w=w, h=h, seed=int(rng.integers(0, 1000)), n_az=int(rng.integers(300, 2000)), kernel=(shape, ku, kv),
                       n_kp=int(rng.integers(1, 2500)), min_hits=0)
    return "depth %4dx%-4d method %d kernel %d %dx%d -> %d keypoints with depth" % (w, h, method, shape, ku, kv, n)


def hamming_case(lib, rng):
    na = int(rng.choice([1, 63, 64, 255, 256, 257, 700, 2000, 2049, 5000]))
    nb = int(rng.choice([1, 63, 64, 65, 255, 256, 257, 511, 1024, 2000, 4095, 4097, 8191, 8192, 8193, 12000, 20000]))
    pc.check_matcher_bf(lib, na, nb, seed=int(rng.integers(0, 1000)))
    return "hamming %5d x %5d" % (na, nb)


def greedy_search_case(lib, rng):
    """The sequential, blocking searches (SearchByProjection frame-to-frame and key-frame-to-frame, SearchLocalPoints): random
    sizes on both sides of the resolve kernels' LDS limits, every motion / window, dense clusters that make long blocker chains -
    many rounds of the round-based resolve with real concurrency between the work-items of a round."""
    kind = int(rng.integers(0, 3))
    n1 = int(rng.choice([40, 300, 1500, 2000, 3000, 6000, 12500]))
    n2 = int(rng.choice([60, 500, 2000, 3000, 6100, 6200, 9000]))
    seed = int(rng.integers(0, 100000))
    if kind == 0:
        motion, th = str(rng.choice(["forward", "backward", "none"])), float(rng.choice([7.0, 15.0, 30.0]))
        n = pc.check_search_by_projection(lib, seed, motion, th, bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), n1=n1, n2=n2)
        return "SearchByProjection %5d -> %5d %s th %g: %d matches" % (n1, n2, motion, th, n)
    if kind == 1:
        th = float(rng.choice([1.0, 3.0, 5.0, 15.0]))
        n = pc.check_search_local_points(lib, seed, th, float(rng.choice([0.7, 0.8, 0.9])), n1=n1, n2=n2)
        return "SearchLocalPoints %5d -> %5d th %g: %d matches" % (n1, n2, th, n)
    n = pc.check_search_by_projection_keyframe(lib, seed, float(rng.choice([3.0, 10.0, 15.0])), int(rng.choice([64, 100, 255])), True, n1=min(n1, 6000), n2=n2)
    return "SearchByProjection(F, KF) %5d -> %5d: %d matches" % (min(n1, 6000), n2, n)


def node_search_case(lib, rng):
    """The searches that work through the vocabulary nodes two FeatureVectors share (SearchForTriangulation, SearchByBoW x 3): random
    feature counts and node counts, i.e. buckets on both sides of what the kernels keep in LDS (256) and in registers (64 / 256)."""
    kind = int(rng.integers(0, 4))
    n = int(rng.choice([60, 300, 700, 1500, 2000, 4000]))
    nodes = int(rng.choice([1, 2, 5, 12, 30, 100, 400]))
    seed = int(rng.integers(0, 100000))
    if kind == 0:
        m = pc.check_triangulation(lib, n, seed=seed, n_nodes=nodes, min_total=0)
        return "SearchForTriangulation %5d features, %3d nodes: %d matches" % (n, nodes, m)
    if kind == 1:
        m = pc.check_search_by_bow(lib, seed, float(rng.choice([0.6, 0.7, 0.9])), bool(rng.integers(0, 2)), n=n, nodes=nodes)
        return "SearchByBoW(KF, F) %5d features, %3d nodes: %d matches" % (n, nodes, m)
    if kind == 3:
        m, both = pc.check_search_by_bow_rig(lib, seed, float(rng.choice([0.6, 0.7, 0.9])), bool(rng.integers(0, 2)), n=n, nodes=nodes)
        return "SearchByBoW(KF, two-camera F) %5d features, %3d nodes: %d matches, %d map points on both cameras" % (n, nodes, m, both)
    m = pc.check_search_by_bow_keyframes(lib, seed, float(rng.choice([0.75, 0.8, 0.9])), bool(rng.integers(0, 2)), n=n, nodes=nodes)
    return "SearchByBoW(KF, KF) %5d features, %3d nodes: %d matches" % (n, nodes, m)


CASES = {"node_search": node_search_case, "extractor": extractor_case, "low_contrast": low_contrast_case, "depth": depth_case, "hamming": hamming_case,
         "greedy_search": greedy_search_case}
