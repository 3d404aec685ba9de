"""Multi-GPU sharding of the front end: one process per GPU, independent frames / sequences per rank, and the
single exchange step of the path — the gather of the variable-length keypoint / descriptor / depth records to
rank 0 (torch.distributed; backend "nccl" is RCCL over xGMI on MI355X, "gloo" in the CPU tests).

Nothing in the per-frame path depends on another frame, so there is no data-path collective:
 * BASELINE configs[3] (KITTI 00-10, 8 sequences): sequence s -> rank s mod world, consecutive frames stay local.
 * one long sequence: contiguous chunks of frames per rank; Hamming matching of (t, t+1) needs frame t+1, so each
   chunk carries a one-frame halo that the next rank also owns (recomputed, never communicated).
Record layout per frame (padded to `cap` keypoints, little endian):
   int32 n | n... cap x rgbl_keypoint (28 B) | cap x 32 B descriptor | cap x f32 depth | cap x f32 uRight
"""
import torch
import torch.distributed as dist

KP_BYTES, DESC_BYTES = 28, 32


def sequences_of_rank(n_sequences, world, rank):
    return [s for s in range(n_sequences) if s % world == rank]


def frame_chunk(n_frames, world, rank, halo=1):
    """Contiguous chunk [begin, end) of rank plus `halo` extra frames at the end needed for (t, t+1) matching."""
    base, rem = divmod(n_frames, world)
    begin = rank * base + min(rank, rem)
    end = begin + base + (1 if rank < rem else 0)
    return begin, end, min(end + halo, n_frames)


def record_bytes(cap):
    return 4 + cap * (KP_BYTES + DESC_BYTES + 8)


def pack_records(d_n, d_kp, d_desc, d_depth, d_uright, out=None):
    """[B] int32, [B,cap,7] f32-typed keypoint records, [B,cap,32] u8, [B,cap] f32 x2 -> [B, record_bytes] u8."""
    B, cap = d_desc.shape[0], d_desc.shape[1]
    if out is None:
        out = torch.empty((B, record_bytes(cap)), dtype=torch.uint8, device=d_desc.device)
    o = 4
    out[:, :o] = d_n.contiguous().view(torch.uint8).view(B, 4)
    out[:, o:o + cap * KP_BYTES] = d_kp.contiguous().view(torch.uint8).view(B, cap * KP_BYTES)
    o += cap * KP_BYTES
    out[:, o:o + cap * DESC_BYTES] = d_desc.contiguous().view(B, cap * DESC_BYTES)
    o += cap * DESC_BYTES
    out[:, o:o + cap * 4] = d_depth.contiguous().view(torch.uint8).view(B, cap * 4)
    o += cap * 4
    out[:, o:o + cap * 4] = d_uright.contiguous().view(torch.uint8).view(B, cap * 4)
    return out


def unpack_records(buf, cap):
    """Inverse of pack_records for one rank's [B, record_bytes] buffer; returns per-frame dicts of numpy arrays."""
    import numpy as np
    a = buf.cpu().numpy()
    frames = []
    for row in a:
        n = int(row[:4].view(np.int32)[0])
        o = 4
        kp = row[o:o + cap * KP_BYTES].reshape(cap, KP_BYTES)[:n].copy()
        o += cap * KP_BYTES
        desc = row[o:o + cap * DESC_BYTES].reshape(cap, DESC_BYTES)[:n].copy()
        o += cap * DESC_BYTES
        depth = row[o:o + cap * 4].view(np.float32)[:n].copy()
        o += cap * 4
        uright = row[o:o + cap * 4].view(np.float32)[:n].copy()
        frames.append(dict(n=n, kp=kp, desc=desc, depth=depth, uright=uright))
    return frames


def gather_records(send, gather_list=None, dst=0):
    """Gather every rank's packed records on `dst`. Root ingests from all peers concurrently (one xGMI link per
    peer on MI355X); a ring would be per-link bound and is the wrong shape for this pattern."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return [send]
    rank = dist.get_rank()
    if rank == dst and gather_list is None:
        gather_list = [torch.empty_like(send) for _ in range(world)]
    dist.gather(send, gather_list if rank == dst else None, dst=dst)
    return gather_list if rank == dst else None
