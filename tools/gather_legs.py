"""The one-rank gather legs of bench.py (extra.gather_rccl_1rank) on their own, with progress lines and the fault handler on:
   python tools/gather_legs.py [batch] [steps]"""
import faulthandler
import os
import sys

faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

import bench
from orb_slam3_rgbl_amd import _lib as L
from orb_slam3_rgbl_amd import frontend as F
from orb_slam3_rgbl_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
w, h, nfeatures, n_az, _ = bench.WORKLOADS["kitti"]
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
lib = L.load()
seq = synth.Sequence(0, w, h, n_frames=B, constant_density=True)
frames = np.stack([seq.frame(i) for i in range(B)])
scans = [synth.lidar_scan(i, n_az=n_az) for i in range(min(B, 8))]
cloud = np.stack([scans[i % len(scans)] for i in range(B)])
proj = F.projection_matrix(synth.KITTI_K, synth.KITTI_TR, lib)
d_imgs, d_cloud = torch.from_numpy(frames).to(dev), torch.from_numpy(cloud).to(dev)
print("inputs resident", flush=True)
out = bench.gather_legs_one_rank(lib, torch, dist, dev, d_imgs, d_cloud, w, h, nfeatures, proj, scans[0].shape[1], B, steps)
print(out, flush=True)
