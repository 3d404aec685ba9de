"""Times the stereo front end (left + right extraction + Frame::ComputeStereoMatches) on device-resident batches."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import parity_checks as pc
from orb_slam3_rgbl_amd import _lib as L, frontend as F, synth
lib = L.load()
B, w, h = 64, synth.KITTI_W, synth.KITTI_H
pairs = [pc.stereo_pair(40 + i, w, h) for i in range(8)]
left = np.stack([pairs[i % 8][0] for i in range(B)]); right = np.stack([pairs[i % 8][1] for i in range(B)])
dev = torch.device("cuda", 0)
exl = F.ORBextractor(2000, 1.2, 8, 20, 7, w, h, max_batch=B, lib=lib); exr = F.ORBextractor(2000, 1.2, 8, 20, 7, w, h, max_batch=B, lib=lib)
cap = exl.max_keypoints
dl, dr = torch.from_numpy(left).to(dev), torch.from_numpy(right).to(dev)
def outs():
    return (torch.zeros((B, cap, 7), dtype=torch.float32, device=dev), torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev),
            torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev))
kl, ddl, nl, ml = outs(); kr, ddr, nr, mr = outs()
ur = torch.zeros((B, cap), dtype=torch.float32, device=dev); dp = torch.zeros((B, cap), dtype=torch.float32, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
def step():
    L.check(lib, lib.rgbl_extract_batch_device(exl.h, p(dl), B, w, h, w, w * h, 0, 0, p(kl), p(ddl), cap, p(nl), p(ml)))
    L.check(lib, lib.rgbl_extract_batch_device(exr.h, p(dr), B, w, h, w, w * h, 0, 0, p(kr), p(ddr), cap, p(nr), p(mr)))
    L.check(lib, lib.rgbl_stereo_matches_batch_device(exl.h, exr.h, B, p(kl), p(ddl), p(nl), p(kr), p(ddr), p(nr), cap, 0.54, 386.1448, p(ur), p(dp)))
for _ in range(3): step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 10
exl.profile(True); exr.profile(True)
for _ in range(3): step()
torch.cuda.synchronize()
prof = exl.profile_read()
print("stereo front end: %.0f stereo frames/s (%.3f ms per %d pairs); matches/frame %.0f; k_stereo_match %.3f ms k_stereo_filter %.3f ms per step"
      % (B / dt, dt * 1e3, B, float((dp > 0).float().sum() / B), prof["k_stereo_match"][0] / 3, prof["k_stereo_filter"][0] / 3))
