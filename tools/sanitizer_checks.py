"""The emulator parity checks against a sanitizer build of the kernel sources (see tools/sanitize_emu.sh)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from orb_slam3_rgbl_amd import _lib as L, frontend as F
import parity_checks as pc
lib = L.bind(os.environ["RGBL_SANITIZED_LIB"])
which = sys.argv[1]
if which == "extract":
    pc.check_extractor(lib, 400, 300, 600, frames=(0,), stages=True)
    pc.check_extractor_partial_batches(lib, 360, 280, 400)
    pc.check_extractor(lib, 159, 152, 200, frames=(0,), nlevels=1, seq=3)   # 43-px cells: the 64-byte tile pitch
    pc.check_extractor(lib, 640, 480, 600, frames=(0,), seq=4)                # 51-px cells on the last level: four waves, 80-byte pitch
    pc.check_extractor_batch(lib, 400, 300, 500, 8)                            # batches: one wave per cell
    pc.check_extractor_low_contrast(lib)                                       # the second FAST pass
    os.environ["RGBL_COMPACT"] = "1"                                           # k_compact_cells for single frames too (round 4)
    pc.check_extractor(lib, 3840, 280, 1500, frames=(0,), nlevels=2, seq=12)   # 424 cells on level 0: two groups, skipped border cells
    pc.check_extractor_empty_root(lib)
    os.environ.pop("RGBL_COMPACT")
elif which == "depth":
    for m in (F.UPS_INVERSE_DILATION, F.UPS_AVERAGE_FILTERING, F.UPS_NEAREST_NEIGHBOR_PIXEL):
        pc.check_depth(lib, m, w=620, h=188, n_az=900, n_kp=400)
    pc.check_depth_edge_cases(lib)
    pc.check_depth_partial_batches(lib)
    pc.check_depth_sparse(lib)
    pc.check_depth_sparse(lib, kernel=(F.KERNEL_RECT, 5, 7))
    pc.check_ingest_kitti_bin(lib)
elif which == "match":
    pc.check_matcher_bf(lib, 300, 280)
    pc.check_matcher_bf(lib, 130, 700, seed=8)   # two launch slices + merge
    pc.check_triangulation(lib, 600, seed=11)
    pc.check_search_by_projection(lib, 21, "forward", 7.0, False, True, n1=700, n2=800)
    pc.check_search_by_projection_edge_cases(lib)
    pc.check_search_local_points(lib, 41, 1.0, 0.8, n1=1200, n2=900)
    pc.check_search_for_initialization(lib, 61, 100, 0.9, True, n1=1500)
    import tempfile
    pc.check_bow_transform(lib, tempfile.mkdtemp(), 10, 3, 2, seed=1, n_feat=600)
    pc.check_search_by_bow(lib, 51, 0.7, True, n=700, nodes=100)
    pc.check_search_by_bow(lib, 54, 0.6, False, n=700, nodes=1)                # one bucket beyond the 256 register positions
    pc.check_search_by_bow_keyframes(lib, 81, 0.75, True, n=800, nodes=30)
    pc.check_search_by_bow_rig(lib, 61, 0.7, True, n=700, nodes=100)           # two-camera frame
    pc.check_search_by_bow_rig(lib, 63, 0.6, True, n=700, nodes=1)
elif which == "misc":
    pc.check_stereo_matches(lib, w=640, h=300, nfeatures=1200)
    pc.check_ingest_color(lib, 402, 300)
    pc.check_extractor_edge_cases(lib)
    pc.check_extractor_threshold_extremes(lib)
    pc.check_extractor_dense_corners(lib)                                      # corner lists overflow: every pixel scored, NMS over all pixels
elif which == "gather":
    # the library's gather (csrc/gather.hip) without a communicator (one rank, device copies) inside the pipeline, both modes,
    # the chunk halo, and the on-device self-test's host side
    pc.check_pipeline_gather(lib, "step", dev=__import__("torch").device("cpu"), w=320, h=200, nfeatures=400, batch=4, steps=3, n_az=300, levels=4)
    pc.check_pipeline_gather(lib, "final", dev=__import__("torch").device("cpu"), w=320, h=200, nfeatures=400, batch=4, steps=3, n_az=300, levels=4)
    L.check(lib, lib.rgbl_selftest_wrappers(0, 1000, 7))
print("asan run", which, "done")
