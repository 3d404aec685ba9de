"""Seeded synthetic stand-ins for KITTI RGB-L input (no dataset is available offline).

Follows SURVEY.md §8(d): a textured grey image with FAST-friendly shapes, a Velodyne-like scan in the
sensor frame (x forward, y left, z up) laid out as the 4xN fp32 matrix that
Examples/RGB-L/rgbl_kitti.cc:151-185 (LoadPointcloudBinaryMat) hands to System::TrackRGBL, and random
256-bit descriptor sets for the matcher.  Pure numpy/scipy, deterministic for a given seed.
"""
import numpy as np
from scipy import ndimage

# Examples/RGB-L/KITTI00-02.yaml:9-12, 44-55
KITTI_K = np.array([[718.856, 0.0, 607.1928, 0.0],
                    [0.0, 718.856, 185.2157, 0.0],
                    [0.0, 0.0, 1.0, 0.0]], np.float32)
KITTI_TR = np.array([[4.276802385584e-04, -9.999672484946e-01, -8.084491683471e-03, -1.198459927713e-02],
                     [-7.210626507497e-03, 8.081198471645e-03, -9.999413164504e-01, -5.403984729748e-02],
                     [9.999738645903e-01, 4.859485810390e-04, -7.206933692422e-03, -2.921968648686e-01],
                     [0.0, 0.0, 0.0, 1.0]], np.float32)
KITTI_W, KITTI_H = 1241, 376


def _scene(rng, w, h, n_shapes):
    base = ndimage.gaussian_filter(rng.standard_normal((h, w)).astype(np.float32), 12.0)
    base = (base - base.min()) / max(float(base.max() - base.min()), 1e-6)
    img = 40.0 + 160.0 * base
    yy, xx = np.mgrid[0:64, 0:64].astype(np.float32)
    for _ in range(n_shapes):
        sw, sh = int(rng.integers(6, 48)), int(rng.integers(6, 48))
        x0, y0 = int(rng.integers(0, max(w - sw, 1))), int(rng.integers(0, max(h - sh, 1)))
        c = float(rng.uniform(15, 90)) * (1 if rng.random() < 0.5 else -1)
        kind = rng.integers(0, 3)
        sub = img[y0:y0 + sh, x0:x0 + sw]
        if kind == 0:  # axis-aligned rectangle
            sub += c
        else:  # triangle / rotated half-plane cut of the box
            ang = rng.uniform(0, 2 * np.pi)
            nx, ny = np.cos(ang), np.sin(ang)
            m = ((xx[:sub.shape[0], :sub.shape[1]] - sw / 2) * nx +
                 (yy[:sub.shape[0], :sub.shape[1]] - sh / 2) * ny) > 0
            sub += c * m
    return img


class Sequence:
    """A synthetic camera sequence: frame t is the scene shifted by (3t, t) px plus fresh sensor noise."""

    def __init__(self, seq_id=0, w=KITTI_W, h=KITTI_H, n_frames=64, n_shapes=None, constant_density=False):
        self.w, self.h, self.seq_id = w, h, seq_id
        self.margin_x, self.margin_y = 3 * n_frames + 8, n_frames + 8
        rng = np.random.default_rng(0xC0FFEE + 7919 * seq_id)
        if n_shapes is None:
            # ~1200 shapes per KITTI-size FRAME give 8-10k FAST candidates on level 0, the density SURVEY.md 8(d) targets.
            # constant_density: the count follows the area of the scene the frames are cut from (a long sequence needs a
            # scene several frames wide); without it the count follows the frame, i.e. long sequences get sparser frames
            # (what the fixtures of the parity tests were generated with).
            area = (w + self.margin_x) * (h + self.margin_y) if constant_density else w * h
            n_shapes = int(round(1200 * area / float(KITTI_W * KITTI_H)))
        self.scene = _scene(rng, w + self.margin_x, h + self.margin_y, n_shapes)

    def frame(self, t):
        rng = np.random.default_rng((0xC0FFEE + 7919 * self.seq_id) * 1000003 + t)
        ox, oy = (3 * t) % self.margin_x, t % self.margin_y
        img = self.scene[oy:oy + self.h, ox:ox + self.w] + 2.0 * rng.standard_normal((self.h, self.w))
        return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def image(seed=0, w=KITTI_W, h=KITTI_H):
    return Sequence(seed, w, h, n_frames=1).frame(0)


def lidar_scan(seed=0, n_rings=64, n_az=1900):
    """Velodyne-like scan -> (4, N) fp32 matrix with rows x, y, z, 1 (N = n_rings*n_az)."""
    rng = np.random.default_rng(0x11DA2 + seed)
    el = np.deg2rad(np.linspace(2.0, -24.8, n_rings)).astype(np.float64)
    az = np.linspace(-np.pi, np.pi, n_az, endpoint=False)
    # piecewise-constant "walls": horizontal distance and height per azimuth sector
    n_sec = 96
    sec = (np.floor((az + np.pi) / (2 * np.pi) * n_sec).astype(int)) % n_sec
    wall_d = rng.uniform(5.0, 80.0, n_sec)[sec]
    wall_h = rng.uniform(0.5, 4.0, n_sec)[sec]
    has_wall = (rng.random(n_sec) < 0.7)[sec]
    EL, AZ = np.meshgrid(el, az, indexing="ij")
    WD, WH, HW = (np.broadcast_to(v, EL.shape) for v in (wall_d, wall_h, has_wall))
    z_at_wall = WD * np.tan(EL)
    hit_wall = HW & (z_at_wall >= -1.73) & (z_at_wall <= -1.73 + WH)
    r_wall = WD / np.cos(EL)
    with np.errstate(divide="ignore", invalid="ignore"):
        r_ground = np.where(EL < 0, 1.73 / np.sin(-EL), np.inf)
    r = np.where(hit_wall, np.minimum(r_wall, r_ground), r_ground)
    r = np.where(r <= 120.0, r, 0.0)
    r = r + (r > 0) * 0.02 * rng.standard_normal(r.shape)
    x = r * np.cos(EL) * np.cos(AZ)
    y = r * np.cos(EL) * np.sin(AZ)
    z = r * np.sin(EL)
    n = x.size
    cloud = np.empty((4, n), np.float32)
    # KITTI .bin order is ring-scan interleaved; any fixed order exercises last-writer-wins
    perm = rng.permutation(n)
    cloud[0] = x.reshape(-1)[perm]
    cloud[1] = y.reshape(-1)[perm]
    cloud[2] = z.reshape(-1)[perm]
    cloud[3] = 1.0
    return cloud


def descriptors(n, seed=0xBEEF):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (n, 32), dtype=np.uint8)


def perturbed_descriptors(a, flip_p=0.08, seed=1):
    """B = A with each bit flipped w.p. flip_p, rows shuffled; returns (B, perm) with B[i] ~ A[perm[i]]."""
    rng = np.random.default_rng(seed)
    bits = np.unpackbits(a, axis=1)
    flips = (rng.random(bits.shape) < flip_p).astype(np.uint8)
    b = np.packbits(bits ^ flips, axis=1)
    perm = rng.permutation(len(a))
    return b[perm].copy(), perm


def feature_vector(desc, n_nodes=100):
    """Stand-in for DBoW2's FeatureVector (ORBvoc.txt is absent): node id = hash of the first descriptor
    bytes folded to <= n_nodes buckets; indices ascend inside a bucket (FeatureVector.cpp:31-45).
    Returns CSR arrays (node_id ascending, node_off, node_feat)."""
    key = (desc[:, 0].astype(np.int64) * 131 + desc[:, 1].astype(np.int64) * 31 + desc[:, 2]) % n_nodes
    order = np.argsort(key, kind="stable")
    ids, counts = np.unique(key, return_counts=True)
    off = np.zeros(len(ids) + 1, np.int32)
    off[1:] = np.cumsum(counts)
    return ids.astype(np.int32), off, order.astype(np.int32)


def make_vocabulary(k=10, L=4, seed=0, stop_fraction=0.02):
    """A synthetic DBoW2 vocabulary tree of the ORBvoc.txt shape (k-ary, L levels below the root; ORBvoc itself is k = 10,
    L = 6): children are their parent's descriptor with ~6 % of the bits flipped, a few inner nodes have fewer than k
    children (k-means clusters can run empty), idf weights are positive with a few stopped (zero) words.
    Returns dict(k, L, parent[int32], is_leaf[uint8], desc[n,32], weight[float64]) in file order (node id = row + 1)."""
    rng = np.random.default_rng(0xB0D + seed)
    parent, leaf, desc, weight = [], [], [], []
    root_desc = np.packbits(rng.integers(0, 2, 256, dtype=np.uint8), bitorder="little")
    level_nodes = [(0, root_desc)]          # (node id, descriptor)
    next_id = 1
    for level in range(1, L + 1):
        nxt = []
        for pid, pdesc in level_nodes:
            nk = k if rng.random() > 0.1 else int(rng.integers(2, k + 1))
            for _ in range(nk):
                flips = rng.random(256) < (0.06 if level > 1 else 0.5)
                d = pdesc ^ np.packbits(flips, bitorder="little")
                parent.append(pid)
                leaf.append(1 if level == L else 0)
                desc.append(d)
                weight.append(0.0 if (level == L and rng.random() < stop_fraction) else float(rng.uniform(0.5, 9.0)) if level == L else 0.0)
                nxt.append((next_id, d))
                next_id += 1
        level_nodes = nxt
    return dict(k=k, L=L, parent=np.array(parent, np.int32), is_leaf=np.array(leaf, np.uint8),
                desc=np.stack(desc).astype(np.uint8), weight=np.array(weight, np.float64))


def write_vocabulary_text(path, voc):
    """The text format TemplatedVocabulary::loadFromTextFile reads (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1425):
    'k L scoring weighting' (0 0 = L1_NORM, TF_IDF as in ORBvoc.txt), then 'parent is_leaf d0 .. d31 weight' per node."""
    lines = ["%d %d 0 0" % (voc["k"], voc["L"])]
    for p, lf, d, w in zip(voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"]):
        lines.append("%d %d %s %r" % (p, lf, " ".join(str(int(b)) for b in d), float(w)))
    with open(path, "w") as f:
        # no newline after the last node: the reference's loader turns a trailing empty line into one more child of the
        # root with an unset descriptor (its `while(!f.eof())` loop, TemplatedVocabulary.h:1378-1420)
        f.write("\n".join(lines))


def vocabulary_arrays(voc):
    """Flat tree arrays (children CSR in file order, node 0 = root) shared by the oracle and the C ABI."""
    n = len(voc["parent"]) + 1
    counts = np.zeros(n, np.int64)
    np.add.at(counts, voc["parent"], 1)
    off = np.zeros(n + 1, np.int32)
    off[1:] = np.cumsum(counts)
    child = np.zeros(n - 1, np.int32)
    fill = off[:-1].copy()
    for i, p in enumerate(voc["parent"]):
        child[fill[p]] = i + 1
        fill[p] += 1
    desc = np.zeros((n, 32), np.uint8)
    desc[1:] = voc["desc"]
    weight = np.zeros(n, np.float64)
    weight[1:] = voc["weight"]
    word_id = np.full(n, -1, np.int32)
    leaves = np.nonzero(voc["is_leaf"])[0] + 1
    word_id[leaves] = np.arange(len(leaves), dtype=np.int32)   # word ids are handed out in file order
    return dict(n_nodes=n, L=voc["L"], child_off=off, child=child, desc=desc, weight=weight, word_id=word_id)
