"""Per-kernel time of one extraction step (HIP events, serial) for a given build of the library.
   python tools/extract_kernel_times.py [--lib path/to/librgbl_frontend.so] [--batch 256] [--check]
Used to compare kernel variants (experimental builds of the library) without the rest of bench.py."""
import argparse, ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from orb_slam3_rgbl_amd import _lib as L, frontend as F, synth

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=L.LIB_PATH)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--check", action="store_true", help="compare frame 0 / B-1 with the oracle")
a = ap.parse_args()
lib = L.bind(a.lib)
B, w, h = a.batch, synth.KITTI_W, synth.KITTI_H
seqs = [synth.Sequence(s, w, h, 4) for s in range(4)]
frames = np.stack([seqs[i % 4].frame((i // 4) % 4) for i in range(B)])
dev = torch.device("cuda", 0)
ex = F.ORBextractor(2000, 1.2, 8, 12, 7, w, h, max_batch=B, lib=lib)
cap = ex.max_keypoints
d = torch.from_numpy(frames).to(dev)
kp = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev); ds = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
n = torch.zeros(B, dtype=torch.int32, device=dev); m = torch.zeros(B, dtype=torch.int32, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
def step():
    L.check(lib, lib.rgbl_extract_batch_device(ex.h, p(d), B, w, h, w, w * h, 0, 0, p(kp), p(ds), cap, p(n), p(m)))
for _ in range(3): step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(a.steps): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / a.steps
ex.profile(True)
for _ in range(a.steps): step()
torch.cuda.synchronize()
prof = ex.profile_read()
print("lib=%s B=%d: %.3f ms/step overlapped; serial per kernel: %s" % (
    os.path.basename(a.lib), B, dt * 1e3,
    "  ".join("%s %.3f" % (k, v[0] / a.steps) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0]))))
if a.check:
    from oracle import oracle_py as O
    for f in (0, B - 1):
        okp, odesc, _ = O.Extractor(2000, 1.2, 8, 12, 7)(frames[f])
        nn = int(n[f])
        ok = nn == len(okp) and np.array_equal(ds[f, :nn].cpu().numpy(), odesc) and \
            np.array_equal(kp[f, :nn, 0].cpu().numpy(), okp["x"]) and np.array_equal(kp[f, :nn, 3].cpu().numpy().view(np.uint32), okp["angle"].view(np.uint32))
        print("  parity frame %d: %s (%d keypoints)" % (f, "bit-exact" if ok else "MISMATCH", nn))
        if not ok: sys.exit(1)
