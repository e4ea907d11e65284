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


def classic_round_sharded(membership_size, emit_batch, fingerprint, local_proposal, dist=None, arrival=None):
    """The classic-Paxos recovery round (consensus.classic_round_population) when the receivers are sharded: every rank
    contributes its receivers' votes (emit_batch >= 0 <=> voted, fingerprint = the vote's identity), all ranks run the
    same O(N) rule on the gathered arrays -- global receiver order is rank order, shards being contiguous -- and the
    rank that owns the chosen receiver hands its proposal (local_proposal(local index) -> list of node indices) to the
    others.  `arrival` indexes the GLOBAL receiver order.  -> (result dict, decided cut or None); identical on all
    ranks.  Two small collectives (gather of 12 bytes per receiver, broadcast of one proposal), host side."""
    from . import consensus as CS
    emit = np.ascontiguousarray(emit_batch, dtype=np.int32)
    fp = np.ascontiguousarray(fingerprint, dtype=np.uint64)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        res, winner = CS.classic_round_from_results(membership_size, emit, fp, arrival)
        return res, (list(local_proposal(winner)) if winner is not None else None)
    world, rank = dist.get_world_size(), dist.get_rank()
    parts = [None] * world
    dist.all_gather_object(parts, (emit, fp))
    sizes = [len(p[0]) for p in parts]
    res, winner = CS.classic_round_from_results(membership_size, np.concatenate([p[0] for p in parts]),
                                                np.concatenate([p[1] for p in parts]), arrival)
    if winner is None:
        return res, None
    starts = np.concatenate([[0], np.cumsum(sizes)])
    owner = int(np.searchsorted(starts, winner, side="right") - 1)
    box = [list(local_proposal(winner - int(starts[owner]))) if rank == owner else None]
    dist.broadcast_object_list(box, src=owner)
    return res, box[0]
