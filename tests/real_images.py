"""Real photographs for rows f1 / f3, read IN PLACE from the scikit-image sample data that ships in this
image (nothing is copied into the repository; tests skip where the files are absent):
  motorcycle_left.png / motorcycle_right.png  a rectified stereo pair (Middlebury 2014 'Motorcycle', 741 x 500 RGB) with its
                                              ground-truth disparity motorcycle_disp.npz - Frame::ComputeStereoMatches (Frame.cc:901-1071)
  astronaut.png, coffee.png                   colour photographs - cvtColor -> extract (Tracking.cc:1567-1580)
"""
import os

import numpy as np
import pytest

DATA = "/opt/conda/lib/python3.9/site-packages/skimage/data"


def _rgb(name):
    path = os.path.join(DATA, name)
    if not os.path.exists(path):
        pytest.skip("%s is not on this host" % path)
    try:
        from PIL import Image
    except ImportError:
        pytest.skip("PIL is not importable here")
    return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"), np.uint8))


def stereo_pair():
    """(left RGB, right RGB, ground-truth disparity of the left view [h, w] float32, inf where unknown)."""
    left, right = _rgb("motorcycle_left.png"), _rgb("motorcycle_right.png")
    path = os.path.join(DATA, "motorcycle_disp.npz")
    disp = np.load(path)["arr_0"].astype(np.float32) if os.path.exists(path) else None
    return left, right, disp


def color_photo(name="astronaut.png"):
    return _rgb(name)


# camera of the skimage copy of the Middlebury pair: focal length 3979.911 px and baseline 193.001 mm at 2964 x 1988, scaled to 741 x 500
MOTORCYCLE_FX = 3979.911 * 741.0 / 2964.0
MOTORCYCLE_MB = 0.193001
MOTORCYCLE_MBF = MOTORCYCLE_FX * MOTORCYCLE_MB
