"""Host-side sharding helpers for the multi-GPU path (one process per GPU).

Mirror of what libtcgpu does internally (csrc/api.hip set_shard / allgather_inplace): particles are
globally Peano-sorted, rank r solves the contiguous index range [r*S, min((r+1)*S, n)) with
S = ceil(n / nranks) (equal particle counts => balanced even with substructure).
"""
import numpy as np


def shard_len(n, nranks):
    return max(1, (n + nranks - 1) // nranks)


def shard_bounds(n, nranks, rank):
    s = shard_len(n, nranks)
    lo = min(rank * s, n)
    hi = min((rank + 1) * s, n)
    return lo, hi


def bootstrap_unique_id(dist, rank, make_id):
    """Rank 0 creates the 128-byte RCCL unique id (tcgpu_comm_unique_id); everyone receives it through
    the already-initialised torch.distributed group (any backend)."""
    box = [make_id().tolist() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return np.array(box[0], dtype=np.uint8)
