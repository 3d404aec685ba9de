# Phase breakdown of k_level_fused (s_memtime stamps summed over the workgroups of each level's launch) and per-level
# launch durations.  Usage (on the GPU box): python tools/fused_stamps.py [W H NFEATURES BATCH]
import sys, os, time
os.environ["RGBL_FUSED_STAMPS"] = "1"
os.environ["RGBL_FUSED"] = "1"
os.environ["RGBL_FUSED_PER_LEVEL"] = "1"
sys.path.insert(0, '.')
import numpy as np
from orb_slam3_rgbl_amd import _lib as L, frontend as F, synth
lib = L.load()
W, H, NF, B = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (1241, 376, 2000, 256)))
ex = F.ORBextractor(NF, 1.2, 8, 12, 7, W, H, max_batch=B, lib=lib)
s = synth.Sequence(0, W, H, n_frames=8)
imgs = np.stack([s.frame(i % 8) for i in range(B)])
ex.extract_batch(imgs)
st0 = np.zeros(8 * 8, np.uint64)
L.check(lib, lib.rgbl_extractor_debug_stamps(ex.h, st0.ctypes.data, len(st0)))
ex.profile(True)
ex.extract_batch(imgs)
prof = ex.profile_read()
st = np.zeros(8 * 8, np.uint64)
L.check(lib, lib.rgbl_extractor_debug_stamps(ex.h, st.ctypes.data, len(st)))
d = (st.astype(np.int64) - st0.astype(np.int64)).reshape(8, 8)
names = ['stage', 'pre+Hrows', 'score+Vcols', 'nms+resize', 'prefix', 'rank', 'surv', 'wgs']
for l in range(8):
    n = max(d[l, 7], 1)
    ms = prof.get('k_level_fused_%d' % l, (0, 0))[0]
    print('level %d: %6d wgs  %.3f ms  cycles/wg: ' % (l, d[l, 7], ms) + ' '.join('%s=%d' % (nm, d[l, k] / n) for k, nm in enumerate(names) if k != 7) + '  total=%d' % (d[l, :6].sum() / n))
print({k: round(v[0], 3) for k, v in prof.items()})
