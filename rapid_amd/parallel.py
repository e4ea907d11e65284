"""Sharding of the simulated population across ranks (one process per GPU).

Receivers (simulated nodes) are independent units: each has its own cut-detector state and its own delivered
alert stream (R/MultiNodeCutDetector.java state is per MembershipService instance).  Rank g of G owns the
contiguous receiver range shard_range(R, g, G); the view (ring tables, configuration id) is rebuilt redundantly
and deterministically on every rank.  The only exchange per round is ONE all-gather of the ranks' answer blocks
(candidate proposal, its verified votes, the rank's voters), merged identically on every rank (vote_merge_kernel in
rapid_amd/csrc/vote_kernels.h); the sum all-reduce of the positional vote histogram is the fallback for rounds whose
ranks hold different candidates without a quorum.  It replaces the N x N unicast fan-out of the fast-round votes
(R/UnicastToAllBroadcaster.java:46-52, R/FastPaxos.java:104).

local_candidate / merge_candidates are the host-side statement of the ONE-collective round (every rank's candidate
proposal with its verified votes, one all-gather, the same merge on every rank: vote_merge_kernel in
rapid_amd/csrc/vote_kernels.h); local_histogram / decide_from_histogram of the general count it falls back to when the
ranks' candidates differ or several proposals leave no quorum.  The world_size-2 gloo tests exercise exactly this logic
on CPU.
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


def local_candidate(emit_batch, fingerprint, proposal_of):
    """What a rank contributes to the round's all-gather: its CANDIDATE = the proposal of its lowest voter
    (proposal_of(local receiver) -> list of node indices), how many of its voters hold exactly that proposal (compared
    element for element, as vote_verify_kernel does), and its number of voters.  -> dict(voters, votes, fp, cut)."""
    emit = np.asarray(emit_batch)
    fp = np.asarray(fingerprint, dtype=np.uint64)
    voters = np.flatnonzero(emit >= 0)
    if len(voters) == 0:
        return dict(voters=0, votes=0, fp=0, cut=None)
    rep = int(voters[0])
    cut = list(proposal_of(rep))
    same = [int(r) for r in voters if fp[r] == fp[rep]]
    bad = [r for r in same if list(proposal_of(r)) != cut]
    if bad:
        raise ValueError("fingerprint collision: receivers %s share a fingerprint with %d but not its proposal" % (bad[:3], rep))
    return dict(voters=int(len(voters)), votes=len(same), fp=int(fp[rep]), cut=cut)


def merge_candidates(blocks, membership_size):
    """The merge every rank runs over the gathered blocks (vote_merge_kernel).  -> (settled, votes, voters, cut):
    settled iff every voting rank's candidate is the same proposal and it has a quorum (then no other proposal can have
    one, R/FastPaxos.java:145-150, whatever the remaining voters hold) or every voter of every rank holds it; otherwise
    the caller owes the exact plurality (histogram path).  cut = the decided proposal iff votes >= quorum."""
    voting = [b for b in blocks if b["voters"] > 0]
    voters = sum(b["voters"] for b in blocks)
    if not voting:
        return True, 0, 0, None
    lead = voting[0]
    if any(b["fp"] != lead["fp"] or b["cut"] != lead["cut"] for b in voting[1:]):
        return False, 0, voters, None
    votes = sum(b["votes"] for b in voting)
    quorum = fast_quorum(membership_size)
    if not (votes >= quorum or votes == voters):
        return False, votes, voters, None
    return True, votes, voters, (lead["cut"] if votes >= quorum else None)


def count_votes_sharded(membership_size, emit_batch, fingerprint, proposal_of, dist=None):
    """One collective: all-gather of the ranks' candidate blocks, merged identically everywhere."""
    mine = local_candidate(emit_batch, fingerprint, proposal_of)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return merge_candidates([mine], membership_size)
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, mine)
    return merge_candidates(parts, membership_size)


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
