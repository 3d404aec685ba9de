"""Randomised GPU-vs-oracle extractor checks over image sizes / feature counts / level counts (bit-exact keypoints and
descriptors, candidate lists per level).  Usage (on an MI355X): python tools/gpu_random_extractor_checks.py [seed] [cases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch  # noqa: F401  (HIP runtime order)
import parity_checks as pc
from orb_slam3_rgbl_amd import _lib as L

lib = L.load()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 12
done = 0
while done < n_cases:
    w, h = int(rng.integers(200, 1400)), int(rng.integers(120, 700))
    if round((w - 32) / max(h - 32, 1)) < 1 or round((w - 32) / max(h - 32, 1)) > 16:
        continue
    nlevels = int(rng.integers(1, 9))
    if min(w, h) / 1.2 ** (nlevels - 1) < 80:
        continue
    nf = int(rng.choice([50, 300, 1000, 2000, 3500, 6000]))
    ini = int(rng.choice([12, 20]))
    total = pc.check_extractor(lib, w, h, nf, frames=(0,), ini=ini, mn=7, nlevels=nlevels, seq=int(rng.integers(0, 1000)), stages=True)
    print("ok %4dx%-4d levels %d nfeatures %5d ini %2d -> %d keypoints" % (w, h, nlevels, nf, ini, total), flush=True)
    done += 1
print("all", done, "cases bit-exact")

# low-contrast frames: most detection cells find nothing at iniThFAST and take the second cv::FAST pass at minThFAST
from orb_slam3_rgbl_amd import frontend as F, synth
from oracle import oracle_py as O
for case in range(max(n_cases // 3, 4)):
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
    print("ok low contrast %4dx%-4d FAST %2d/%2d x%.2f -> %d keypoints" % (w, h, ini, mn, contrast, len(kps)), flush=True)
    ex.close()
print("low-contrast cases bit-exact")
