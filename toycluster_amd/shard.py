"""Host-side sharding helpers for the multi-GPU path (one process per GPU).

Mirror of what libtcgpu does internally (csrc/api.hip set_shard / allgather_inplace): particles are
globally Peano-sorted, rank r solves the contiguous index range [r*S, min((r+1)*S, n)) with
S = ceil(n / nranks) (equal particle counts => balanced even with substructure), and arrays are padded to
S*nranks so that every rank contributes an equal-size block to the in-place all-gather.
"""
import numpy as np


def shard_len(n, nranks):
    return max(1, (n + nranks - 1) // nranks)


def shard_bounds(n, nranks, rank):
    s = shard_len(n, nranks)
    lo = min(rank * s, n)
    hi = min((rank + 1) * s, n)
    return lo, hi


def padded_len(n, nranks):
    return shard_len(n, nranks) * nranks


def bootstrap_unique_id(dist, rank, make_id):
    """Rank 0 creates the 128-byte RCCL unique id (tcgpu_comm_unique_id); everyone receives it through
    the already-initialised torch.distributed group (any backend)."""
    box = [make_id().tolist() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return np.array(box[0], dtype=np.uint8)


def allgather_shards(dist, local_block, n, nranks):
    """Equal-block all-gather + un-padding: the host-side equivalent of the library's in-place RCCL
    all-gather.  local_block has shard_len(n, nranks) rows (the tail rank's is padded)."""
    import torch
    s = shard_len(n, nranks)
    t = torch.as_tensor(np.ascontiguousarray(local_block))
    assert t.shape[0] == s
    out = [torch.empty_like(t) for _ in range(nranks)]
    dist.all_gather(out, t)
    return torch.cat(out, dim=0)[:n].numpy()
