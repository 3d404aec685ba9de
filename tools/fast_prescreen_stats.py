"""Survivor rates of FAST pre-screens on the bench frames: the 4 opposite pairs at even ring positions
(what k_fast_cells tests), all 8 pairs (cv::FAST), >= 4 consecutive even positions, two adjacent compass points, and the
pixels that really have a 9-arc; per level, at iniThFAST and minThFAST; cells without a corner at iniThFAST.  CPU only (numpy + the oracle)."""
import numpy as np, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam3_rgbl_amd import synth
from oracle import oracle_py as O
seq = synth.Sequence(0, n_frames=512, constant_density=True)
img0 = seq.frame(100)
ring = [(0,3),(1,3),(2,2),(3,1),(3,0),(3,-1),(2,-2),(1,-3),(0,-3),(-1,-3),(-2,-2),(-3,-1),(-3,0),(-3,1),(-2,2),(-1,3)]
def stats(img, name):
    h,w = img.shape
    I = img.astype(np.int16)
    c = I[3:h-3,3:w-3]
    R = [I[3+dy:h-3+dy, 3+dx:w-3+dx] for dx,dy in ring]
    out = {}
    for thr in (12,7):
        dk = [r < c-thr for r in R]; br = [r > c+thr for r in R]
        def pairs(F, ks): 
            m = np.ones_like(F[0])
            for k in ks: m &= (F[k]|F[k+8])
            return m
        p4 = pairs(dk,[0,2,4,6]) | pairs(br,[0,2,4,6])
        p8 = pairs(dk,range(8)) | pairs(br,range(8))
        def run4even(F):
            E=[F[2*i] for i in range(8)]
            m=np.zeros_like(E[0])
            for s in range(8): m |= E[s]&E[(s+1)%8]&E[(s+2)%8]&E[(s+3)%8]
            return m
        def run9(F):
            m=np.zeros_like(F[0])
            for s in range(16):
                a=np.ones_like(F[0])
                for j in range(9): a &= F[(s+j)%16]
                m|=a
            return m
        r4 = run4even(dk)|run4even(br)
        # compass: two adjacent of (0,4,8,12)
        def comp(F): return (F[0]&F[4])|(F[4]&F[8])|(F[8]&F[12])|(F[12]&F[0])
        cp = comp(dk)|comp(br)
        r9 = run9(dk)|run9(br)
        both = (pairs(dk,[0,2,4,6]) & pairs(br,[0,2,4,6]))
        print(name, "thr",thr, "p4 %.3f p8 %.3f run4even %.3f compass2 %.3f p4&compass %.3f corners %.4f  p4both %.4f" % (p4.mean(), p8.mean(), r4.mean(), cp.mean(), (p4&cp).mean(), r9.mean(), both.mean()))
        out[thr]=(p4,r9)
    # cells of 35 px: fraction with no corner at 12
    p4,r9 = out[12]
    H,W = r9.shape
    n=0; empty=0
    for y in range(16, H-16-34, 35):
        for x in range(16, W-16-34, 35):
            n+=1; empty += not r9[y:y+35,x:x+35].any()
    print(name, "cells", n, "no corner at 12:", empty/n)
ex = O.Extractor(2000,1.2,8,12,7)
ex(img0)
for l in (0,1,3,5):
    stats(ex.level_image(l), "L%d"%l)
