"""The oracle's OpenCV restatements against a REAL OpenCV, wherever one is installed (`cv2`).

There is none in the build image (no network, no wheel), so there these tests are skipped and the pins of
tests/test_opencv_restatements.py (a second, independent numpy restatement) and tests/test_natural_images.py are what
holds the oracle.  On a machine with OpenCV 4.x (`pip install opencv-python-headless`) they compare, bit for bit, the
primitives the reference's hot path calls: cv::resize INTER_LINEAR (ORBextractor.cc:1170-1195), cv::GaussianBlur 7x7 sigma 2
BORDER_REFLECT_101 (:1133), cv::FAST with non-maximum suppression (:826-846), cv::fastAtan2 (:101), cv::cvtColor (Tracking.cc),
cv::dilate with the module's structuring elements (DepthModule.cc:265-273) and cv::undistortPoints (Frame.cc:775-800).
OpenCV builds dispatch to SIMD code paths; those are required to be bit-identical to the scalar ones for these 8-bit
fixed-point primitives, and the float ones (fastAtan2, undistortPoints) are compared with the tolerance stated in the test.
"""
import numpy as np
import pytest

cv2 = pytest.importorskip("cv2")

from oracle import oracle_py as O  # noqa: E402
from orb_slam3_rgbl_amd import synth  # noqa: E402


def _img(seed, w, h):
    return synth.Sequence(seed, w, h, n_frames=1).frame(0)


@pytest.mark.parametrize("sw,sh,dw,dh", [(1241, 376, 1034, 313), (1034, 313, 862, 261), (417, 126, 348, 105), (64, 48, 53, 40)])
def test_resize_linear_matches_opencv(sw, sh, dw, dh):
    src = _img(3, sw, sh)
    assert np.array_equal(O.resize_linear(src, dw, dh), cv2.resize(src, (dw, dh), interpolation=cv2.INTER_LINEAR))


@pytest.mark.parametrize("w,h", [(57, 40), (131, 9), (640, 480)])
def test_gaussian_blur_matches_opencv(w, h):
    img = _img(4, max(w, 64), max(h, 64))[:h, :w].copy()
    ref = cv2.GaussianBlur(img, (7, 7), 2, 2, borderType=cv2.BORDER_REFLECT_101)
    assert np.array_equal(O.gaussian_blur7(img), ref)


@pytest.mark.parametrize("threshold", [20, 12, 7])
def test_fast_matches_opencv(threshold):
    img = _img(5, 320, 200)
    det = cv2.FastFeatureDetector_create(threshold=threshold, nonmaxSuppression=True, type=cv2.FAST_FEATURE_DETECTOR_TYPE_9_16)
    ref = det.detect(img, None)
    got = O.fast(img, threshold, True)
    assert len(got) == len(ref)
    # cv::FAST emits row-major; the oracle keeps that order
    assert np.array_equal(got["x"], np.array([k.pt[0] for k in ref], np.float32))
    assert np.array_equal(got["y"], np.array([k.pt[1] for k in ref], np.float32))
    assert np.array_equal(got["response"], np.array([k.response for k in ref], np.float32))


def test_fast_atan2_matches_opencv():
    rng = np.random.default_rng(6)
    y = rng.integers(-50000, 50000, 2000).astype(np.float32)
    x = rng.integers(-50000, 50000, 2000).astype(np.float32)
    ref = cv2.phase(x, y, angleInDegrees=True).ravel()  # cv::phase uses the same FastAtan2 kernel as cv::fastAtan2
    got = np.array([O.fast_atan2(float(b), float(a)) for a, b in zip(x, y)], np.float32)
    # the SIMD kernel and the scalar one are the same polynomial; allow one unit in the last place of 360
    assert np.max(np.abs(got - ref)) <= 6e-5


@pytest.mark.parametrize("rgb", [True, False])
def test_cvt_gray_matches_opencv(rgb):
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (61, 83, 3), dtype=np.uint8)
    ref = cv2.cvtColor(img, cv2.COLOR_RGB2GRAY if rgb else cv2.COLOR_BGR2GRAY)
    assert np.array_equal(O.cvt_gray(img, rgb), ref)


def test_undistort_points_matches_opencv():
    K = np.array([458.654, 457.296, 367.215, 248.375], np.float32)  # EuRoC cam0
    dist = np.array([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05], np.float32)
    rng = np.random.default_rng(8)
    xy = np.stack([rng.uniform(0, 752, 500), rng.uniform(0, 480, 500)], 1).astype(np.float32)
    Km = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]], np.float32)
    ref = cv2.undistortPoints(xy.reshape(-1, 1, 2), Km, dist, R=np.eye(3, dtype=np.float32), P=Km).reshape(-1, 2)
    got = O.undistort_points(xy, K, dist)
    assert np.max(np.abs(got - ref)) <= 1e-3  # iterative solve in double, results rounded to float: sub-milli-pixel
