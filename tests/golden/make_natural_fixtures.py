"""Makes the natural-image fixtures of tests/golden (real texture for corner statistics - every other parity
input comes from the procedural generator).  Source: the sample images that ship with scikit-image in the survey / build
container (/opt/conda/lib/python3.9/site-packages/skimage/data): `camera.png` (CC0, a photographer with a tripod camera)
and `brick.png` (CC0 texture), both 512 x 512 8-bit grey.  The golden vectors are the CPU ORACLE's results on them
(keypoint count per level, SHA-256 of the keypoint records and of the descriptors): they pin the oracle against
regressions on natural texture - they are not a second opinion on OpenCV's arithmetic (skimage's own FAST uses another score).
Run from the repository root:  python tests/golden/make_natural_fixtures.py"""
import hashlib
import json
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O  # noqa: E402

SRC = "/opt/conda/lib/python3.9/site-packages/skimage/data"
OUT = os.path.join(ROOT, "tests", "golden")
CONFIGS = {"kitti": (1000, 12, 7), "stereo": (1500, 20, 7)}  # nfeatures, iniThFAST, minThFAST


def digest(img):
    out = {}
    for name, (nf, ini, mn) in CONFIGS.items():
        ex = O.Extractor(nf, 1.2, 8, ini, mn)
        kps, desc, mono = ex(img)
        out[name] = {"n": int(len(kps)), "mono": int(mono),
                     "per_level": [int((kps["octave"] == l).sum()) for l in range(8)],
                     "keypoints_sha256": hashlib.sha256(np.ascontiguousarray(kps).tobytes()).hexdigest(),
                     "descriptors_sha256": hashlib.sha256(np.ascontiguousarray(desc).tobytes()).hexdigest()}
    return out


def main():
    golden = {}
    for name in ("camera", "brick"):
        img = np.asarray(Image.open(os.path.join(SRC, name + ".png")).convert("L"), np.uint8)
        Image.fromarray(img).save(os.path.join(OUT, "natural_%s.png" % name), optimize=True)
        golden[name] = {"shape": list(img.shape), "pixels_sha256": hashlib.sha256(img.tobytes()).hexdigest(), "oracle": digest(img)}
    json.dump(golden, open(os.path.join(OUT, "natural_golden.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps({k: {c: v["oracle"][c]["per_level"] for c in v["oracle"]} for k, v in golden.items()}))


if __name__ == "__main__":
    main()
