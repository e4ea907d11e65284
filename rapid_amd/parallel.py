"""Sharding of the simulated population across ranks (one process per GPU).

Receivers (simulated nodes) are independent units: each has its own cut-detector state and its own delivered
alert stream (R/MultiNodeCutDetector.java state is per MembershipService instance).  Rank g of G owns the
contiguous receiver range shard_range(R, g, G); the view (ring tables, configuration id) is rebuilt redundantly
and deterministically on every rank.  The only exchange per round is the sum all-reduce of the positional vote
histogram (rapid_amd/csrc/vote_kernels.h) -- it replaces the N x N unicast fan-out of the fast-round votes
(R/UnicastToAllBroadcaster.java:46-52, R/FastPaxos.java:104).

merge_histograms / decide_from_histogram are the host-side reference of what the engine does with RCCL on the
device; the world_size-2 gloo tests exercise exactly this logic on CPU.
"""
import numpy as np

VOTE_BUCKETS = 1 << 14
_MASK = (1 << 64) - 1


def shard_range(n_units, rank, world):
    """Contiguous, balanced: sizes differ by at most one; concatenation over ranks is range(n_units)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(n_units, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def vote_bucket(fp, salt=0):
    """Same mixing as vote_bucket() in rapid_amd/csrc/vote_kernels.h."""
    x = (int(fp) ^ ((salt * 0xD6E8FEB86659FD93) & _MASK)) & _MASK
    x ^= x >> 32
    x = (x * 0xD6E8FEB86659FD93) & _MASK
    x ^= x >> 32
    return x & (VOTE_BUCKETS - 1)


def local_histogram(fingerprints, prop_counts, salt=0):
    """hist[b] = local voters whose proposal falls into bucket b; hist[VOTE_BUCKETS] = local voters."""
    hist = np.zeros(VOTE_BUCKETS + 2, dtype=np.uint64)
    for fp, c in zip(fingerprints, prop_counts):
        if c > 0:
            hist[vote_bucket(fp, salt)] += np.uint64(1)
            hist[VOTE_BUCKETS] += np.uint64(1)
    return hist


def fast_quorum(membership_size):
    """N - floor((N-1)/4)  (R/FastPaxos.java:145-147)."""
    return membership_size - (membership_size - 1) // 4


def decide_from_histogram(hist, membership_size):
    """(winning bucket, votes, total voters, quorum reached?) from an all-reduced histogram."""
    b = int(np.argmax(hist[:VOTE_BUCKETS]))
    votes = int(hist[b])
    return b, votes, int(hist[VOTE_BUCKETS]), votes >= fast_quorum(membership_size)


def all_reduce_histogram(hist, dist=None):
    """Sum over ranks with torch.distributed (gloo on CPU in tests; the engine itself uses RCCL on the device)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return hist
    import torch
    t = torch.from_numpy(hist.astype(np.int64))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.numpy().astype(np.uint64)
