"""A second, independently written restatement (vectorised numpy, from the published OpenCV 4.x algorithms) of the
OpenCV-internal arithmetic the oracle restates in C++: cv::resize INTER_LINEAR on CV_8UC1, cv::GaussianBlur 7x7 sigma 2,
the FAST-9/16 segment test and cornerScore<16>, cv::fastAtan2 and cv::cvtColor RGB->gray.  None of it can be checked
against a real OpenCV in this image (none installed), so these tests do the next best thing: a typo in one restatement
cannot hide in the other.  Inputs are random images / values; results must agree bit for bit."""
import math

import numpy as np
import pytest

from oracle import oracle_py as O

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0),
        (-3, 1), (-2, 2), (-1, 3)]  # (dx, dy) of cv::FAST's pattern for patternSize 16


def resize_numpy(src, dw, dh):
    """modules/imgproc/src/resize.cpp, INTER_LINEAR, 8-bit: 11-bit coefficients, HResizeLinear / VResizeLinear<uchar,int,short>."""
    sh, sw = src.shape

    def axis(ssize, dsize, clamp):
        scale = 1.0 / (dsize / ssize)                                     # double, inv_scale = dsize / ssize
        d = np.arange(dsize, dtype=np.float64)
        f = ((d + 0.5) * scale - 0.5).astype(np.float32)                  # float fx = (float)((dx + 0.5) * scale_x - 0.5)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        if clamp:                                                         # x only: the row table is clamped when it is used
            lo, hi = s < 0, s >= ssize - 1
            f = np.where(lo | hi, np.float32(0), f)
            s = np.where(lo, 0, np.where(hi, ssize - 1, s))
        a0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)   # saturate_cast<short>(cvRound)
        a1 = np.rint(f * np.float32(2048)).astype(np.int64)
        return s, a0, a1

    sx, ax0, ax1 = axis(sw, dw, True)
    sy, ay0, ay1 = axis(sh, dh, False)
    S = src.astype(np.int64)
    sx1 = np.minimum(sx + 1, sw - 1)
    rows = S[:, sx] * ax0[None, :] + S[:, sx1] * ax1[None, :]             # horizontal pass, int, one value per source row
    y0, y1 = np.clip(sy, 0, sh - 1), np.clip(sy + 1, 0, sh - 1)
    r0, r1 = rows[y0], rows[y1]
    out = (((ay0[:, None] * (r0 >> 4)) >> 16) + ((ay1[:, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


@pytest.mark.parametrize("sw,sh,dw,dh", [(1241, 376, 1034, 313), (1034, 313, 862, 261), (417, 126, 348, 105), (64, 48, 53, 40),
                                         (100, 37, 50, 19), (33, 200, 31, 167)])
def test_resize_two_restatements_agree(sw, sh, dw, dh):
    rng = np.random.default_rng(sw * 7 + dh)
    src = rng.integers(0, 256, (sh, sw), dtype=np.uint8)
    assert np.array_equal(O.resize_linear(src, dw, dh), resize_numpy(src, dw, dh))


def blur_numpy(img):
    """GaussianBlur 7x7 sigma 2 on CV_8U: separable 8.8 fixed-point kernel, rows then columns, one rounding at the end."""
    k = [18, 34, 48, 56, 48, 34, 18]
    h, w = img.shape
    ix = np.arange(-3, w + 3)
    ix = np.where(ix < 0, -ix, np.where(ix >= w, 2 * (w - 1) - ix, ix))   # BORDER_REFLECT_101
    iy = np.arange(-3, h + 3)
    iy = np.where(iy < 0, -iy, np.where(iy >= h, 2 * (h - 1) - iy, iy))
    p = img.astype(np.int64)[iy][:, ix]
    hs = sum(k[i] * p[:, i:i + w] for i in range(7))
    vs = sum(k[i] * hs[i:i + h, :] for i in range(7))
    return ((vs + (1 << 15)) >> 16).astype(np.uint8)


@pytest.mark.parametrize("w,h", [(57, 40), (8, 8), (131, 9), (300, 211)])
def test_gaussian_two_restatements_agree(w, h):
    rng = np.random.default_rng(w + h)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    assert np.array_equal(O.gaussian_blur7(img), blur_numpy(img))


def fast_margins_numpy(img):
    """For every pixel with a 3-px margin: the largest t for which 9 contiguous ring pixels are all brighter than v + t or
    all darker than v - t, i.e. min over the best arc of |ring - v|, minus 1 (-1 when no arc is one-sided)."""
    h, w = img.shape
    v = img[3:h - 3, 3:w - 3].astype(np.int64)
    d = np.stack([img[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx].astype(np.int64) - v for dx, dy in RING])   # ring - v
    best = np.full(v.shape, -1, np.int64)
    for k in range(16):
        arc = d[[(k + i) % 16 for i in range(9)]]
        best = np.maximum(best, np.maximum(arc.min(0), (-arc).min(0)) - 1)
    return best


def test_fast_two_restatements_agree():
    rng = np.random.default_rng(11)
    base = rng.integers(0, 256, (12, 16)).astype(np.float64)
    img = np.kron(base, np.ones((6, 6)))                                  # blocky texture: plenty of real corners
    img = np.clip(img + 9 * rng.standard_normal(img.shape), 0, 255).astype(np.uint8)
    score = fast_margins_numpy(img)
    for t in (7, 12, 20, 40):
        det = O.fast(img, t, nonmax=False)
        got = np.zeros(score.shape, bool)
        got[det["y"].astype(int) - 3, det["x"].astype(int) - 3] = True
        assert np.array_equal(got, score >= t), t                        # the segment test at threshold t
        for kp in det[::17]:                                               # cornerScore<16> (cv::FAST fills it in only with nonmax)
            assert O.corner_score(img, int(kp["x"]), int(kp["y"]), t) == score[int(kp["y"]) - 3, int(kp["x"]) - 3]
    assert (score >= 7).sum() > 200
    # non-maximum suppression: strict maximum of the 8 neighbours' scores (0 where a neighbour is not a corner)
    t = 12
    s = np.where(score >= t, score, 0)
    sp = np.pad(s, 1)
    nb = np.max([sp[1 + j:1 + j + s.shape[0], 1 + i:1 + i + s.shape[1]] for i in (-1, 0, 1) for j in (-1, 0, 1) if (i, j) != (0, 0)], 0)
    keep = (s > 0) & (s > nb)
    nms = O.fast(img, t, nonmax=True)
    got = np.zeros(score.shape, bool)
    got[nms["y"].astype(int) - 3, nms["x"].astype(int) - 3] = True
    assert np.array_equal(got, keep)
    assert np.array_equal(nms["response"].astype(int), score[nms["y"].astype(int) - 3, nms["x"].astype(int) - 3])


def fast_atan2_numpy(y, x):
    """cv::fastAtan2 (modules/core/src/mathfuncs_core.simd.hpp, atan_f32): degree-7 odd polynomial on min/max, in fp32."""
    f = np.float32
    scale = f(180.0 / math.pi)
    p1, p3, p5, p7 = (f(f(c) * scale) for c in (0.9997878412794807, -0.3258083974640975, 0.1555786518463281, -0.04432655554792128))
    y, x = np.asarray(y, np.float32), np.asarray(x, np.float32)
    ax, ay = np.abs(x), np.abs(y)
    eps = f(2.220446049250313e-16)
    with np.errstate(divide="ignore", invalid="ignore"):
        c_lo = ay / (ax + eps)
        c_hi = ax / (ay + eps)
    def poly(c):
        c2 = c * c
        return (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
    with np.errstate(over="ignore", invalid="ignore"):   # the branch that is not selected may overflow
        a = np.where(ax >= ay, poly(c_lo), f(90.0) - poly(c_hi)).astype(np.float32)
    a = np.where(x < 0, f(180.0) - a, a).astype(np.float32)
    a = np.where(y < 0, f(360.0) - a, a).astype(np.float32)
    return a


def test_fast_atan2_two_restatements_agree():
    rng = np.random.default_rng(5)
    ys = rng.integers(-3200000, 3200000, 20000).astype(np.float32)     # the range of the intensity-centroid moments
    xs = rng.integers(-3200000, 3200000, 20000).astype(np.float32)
    ys[:50], xs[:50] = 0, rng.integers(-1000, 1000, 50)                  # axes
    xs[50:100] = 0
    ys[100:150] = xs[100:150]                                             # diagonals
    want = fast_atan2_numpy(ys, xs)
    got = np.array([O.fast_atan2(float(a), float(b)) for a, b in zip(ys, xs)], np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_cvtcolor_two_restatements_agree():
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    r, g, b = (img[..., i].astype(np.int64) for i in range(3))
    want = ((r * 9798 + g * 19235 + b * 3735 + (1 << 14)) >> 15).astype(np.uint8)   # OpenCV 4.x: 15-bit weights
    assert np.array_equal(O.cvt_gray(img, True), want)
    assert np.array_equal(O.cvt_gray(img[..., ::-1].copy(), False), want)
