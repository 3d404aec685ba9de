import sys, os, ctypes as C
os.environ["RGBL_OCTREE_STAMPS"]="1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orb_slam3_rgbl_amd import _lib as L, frontend as F, synth
lib=L.load()
W,H,NF,B=(int(v) for v in (sys.argv[1:5] if len(sys.argv)>4 else (1241,376,2000,16)))
ex=F.ORBextractor(NF,1.2,8,12,7,W,H,max_batch=B,lib=lib)
s=synth.Sequence(0,W,H,n_frames=B,constant_density=True)
imgs=np.stack([s.frame(i) for i in range(B)])
for it in range(2): res=ex.extract_batch(imgs)
st=np.zeros(B*8*16,np.uint64)
L.check(lib, lib.rgbl_extractor_debug_stamps(ex.h, st.ctypes.data, len(st)))
st=st.reshape(B,8,16)
names=['gather','roots','bfs','careful','final']
for l in range(8):
    d=st[:,l,:]
    seg=[(d[:,k+1].astype(np.int64)-d[:,k].astype(np.int64)).mean() for k in range(5)]
    sort=(d[:,9].astype(np.int64)-d[:,8].astype(np.int64)).mean()
    rr=[(d[:,k].astype(np.int64)-d[:,0].astype(np.int64)).mean()/100.0 for k in range(10,16)]
    print('   rounds (rebuild, pass) since roots:', ' '.join('%.0f'%v for v in rr))
    print('level',l,'C=%d n=%d'%(d[:,6].mean(), d[:,7].mean()), ' '.join('%s=%.0f'%(n,v/100.0) for n,v in zip(names,seg)), 'last_sort=%.0f'%(sort/100.0), 'total=%.0f (x100 ticks)'%((d[:,5].astype(np.int64)-d[:,0].astype(np.int64)).mean()/100.0))
