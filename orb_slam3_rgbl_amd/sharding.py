"""Multi-GPU sharding of the front end: which frames / sequences a rank owns.  One process per GPU, independent frames /
sequences per rank, no data-path collective; the single exchange of the path - the two-phase gather of the
variable-length keypoint / descriptor / depth records to rank 0 - lives next to the step it follows, in pipeline.py.

 * BASELINE configs[3] (KITTI 00-10, 8 sequences): sequence s -> rank s mod world, consecutive frames stay local.
 * one long sequence: contiguous chunks of frames per rank; Hamming matching of (t, t+1) needs frame t+1, so each
   chunk carries a one-frame halo that the next rank also owns (recomputed, never communicated).
"""


def sequences_of_rank(n_sequences, world, rank):
    return [s for s in range(n_sequences) if s % world == rank]


def frame_chunk(n_frames, world, rank, halo=1):
    """Contiguous chunk [begin, end) of rank plus `halo` extra frames at the end needed for (t, t+1) matching."""
    base, rem = divmod(n_frames, world)
    begin = rank * base + min(rank, rem)
    end = begin + base + (1 if rank < rem else 0)
    return begin, end, min(end + halo, n_frames)
