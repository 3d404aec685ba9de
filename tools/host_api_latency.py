"""Latency / PCIe-inclusive throughput of the host-pointer (drop-in) entry points, one frame per call."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orb_slam3_rgbl_amd import _lib as L, frontend as F, synth
lib = L.load()
w, h = synth.KITTI_W, synth.KITTI_H
seq = synth.Sequence(0, w, h, n_frames=8)
imgs = [seq.frame(i) for i in range(8)]
cloud = synth.lidar_scan(0)
ex = F.ORBextractor(2000, 1.2, 8, 12, 7, w, h, lib=lib)
dm = F.DepthModule(F.projection_matrix(synth.KITTI_K, synth.KITTI_TR, lib), w, h, max_points=cloud.shape[1], max_keypoints=ex.max_keypoints, lib=lib)
mt = F.ORBmatcher(0.6, False, lib=lib)
prev = None
for it in range(3):
    t = [0.0, 0.0, 0.0]
    n = 0
    for rep in range(5):
        for img in imgs:
            a = time.perf_counter(); kps, desc, _ = ex(img)
            b = time.perf_counter(); dm.CalculateDepthFromPcd(kps, kps, cloud, w, h, want_maps=False)
            c = time.perf_counter()
            if prev is not None: mt.BruteForce(prev, desc)
            d = time.perf_counter(); prev = desc
            t[0] += b - a; t[1] += c - b; t[2] += d - c; n += 1
print("host API, 1 frame per call (H2D + kernels + D2H, synchronous): extract %.3f ms, depth %.3f ms, match %.3f ms -> %.0f frames/s"
      % (t[0] / n * 1e3, t[1] / n * 1e3, t[2] / n * 1e3, n / sum(t)))
cloud32 = np.ascontiguousarray(cloud, np.float32)
for it in range(3):
    a = time.perf_counter(); n = 0
    for rep in range(5):
        for img in imgs:
            ex.Begin(img); dm.PrefetchPointcloud(cloud32, w, h)
            kps, desc, _ = ex(img)
            dm.CalculateDepthFromPcd(kps, kps, cloud32, w, h, want_maps=False)
            mt.BruteForce(prev, desc); prev = desc; n += 1
    tot = (time.perf_counter() - a) / n * 1e3
print("host API, 1 frame per call behind rgbl_extract_begin + rgbl_depth_prefetch: %.3f ms per frame -> %.0f frames/s" % (tot, 1e3 / tot))
res = None
exb = F.ORBextractor(2000, 1.2, 8, 12, 7, w, h, max_batch=64, lib=lib)
big = np.stack([imgs[i % 8] for i in range(64)])
exb.extract_batch(big)
a = time.perf_counter()
for _ in range(5): exb.extract_batch(big)
print("host API, 64 frames per call (rgbl_extract_batch, pageable host memory): %.0f frames/s" % (5 * 64 / (time.perf_counter() - a)))
