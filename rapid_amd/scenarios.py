"""Seeded synthetic populations and alert streams.

Everything here is host-side tooling: it produces the *inputs* (endpoints, node ids, per-receiver streams of
packed 20-byte alert records) that are fed, byte for byte, both to the HIP engine and to the CPU oracle.
It needs the monitoring topology (`subj[n][k]` = subject of n on ring k, i.e. the predecessor table of
MembershipView.java:308-322) from whoever built the view.

Record layout (include/rapid_mi355x.h, `rapid_alert_record`): int64 cfg_id; uint32 src; uint32 dst;
uint16 ring_mask; uint8 status (0 = UP, 1 = DOWN); uint8 flags (bit0 = last record of its batch).
One *batch* = one BatchedAlertMessage (rapid.proto:95-99) = all alerts of one sender.
"""
from dataclasses import dataclass, field

import numpy as np

ALERT_DTYPE = np.dtype(
    [("cfg_id", "<i8"), ("src", "<u4"), ("dst", "<u4"), ("ring_mask", "<u2"), ("status", "u1"), ("flags", "u1")]
)
FLAG_LAST_IN_BATCH = 1
UP, DOWN = 0, 1
SEED_IDS = 0x5241504944  # "RAPID"

_M64 = (1 << 64) - 1


def splitmix64(seed, idx):
    """Stateless splitmix64 of (seed + idx * golden) -- used for NodeIds."""
    z = (np.uint64(seed) + (np.asarray(idx, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15))
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


@dataclass
class Population:
    """N endpoints `10.a.b.c:5000` with splitmix64 NodeIds."""

    n: int
    hostnames: list = field(default_factory=list)
    ports: np.ndarray = None
    id_hi: np.ndarray = None
    id_lo: np.ndarray = None

    @staticmethod
    def make(n, seed_ids=SEED_IDS, port=5000):
        with np.errstate(over="ignore"):
            i = np.arange(n, dtype=np.uint64)
            hi = splitmix64(seed_ids, 2 * i).view(np.int64)
            lo = splitmix64(seed_ids, 2 * i + 1).view(np.int64)
        hosts = [b"10.%d.%d.%d" % ((j >> 16) & 255, (j >> 8) & 255, j & 255) for j in range(n)]
        return Population(n, hosts, np.full(n, port, dtype=np.int32), hi.copy(), lo.copy())

    def blob(self):
        """hostnames as one byte blob + offsets (the C-ABI's rapid_view_build input)."""
        off = np.zeros(self.n + 1, dtype=np.int32)
        off[1:] = np.cumsum([len(h) for h in self.hostnames])
        return np.frombuffer(b"".join(self.hostnames), dtype=np.uint8).copy(), off


def pick_faulty(n, f, seed):
    """F distinct nodes by a seeded Fisher-Yates."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return np.sort(rng.permutation(n)[:f]).astype(np.int32)


def close_fault_set(faulty, subj, L):
    """C3b: add every healthy node that has >= L faulty observers until none is left."""
    n, K = subj.shape
    is_f = np.zeros(n, dtype=bool)
    is_f[faulty] = True
    while True:
        # spurious reports about s = number of rings k on which s's observer is faulty.  subj[o][k] == s <=> o observes s on k
        cnt = np.zeros(n, dtype=np.int32)
        for k in range(K):
            np.add.at(cnt, subj[is_f, k], 1)
        add = (~is_f) & (cnt >= L)
        if not add.any():
            return np.flatnonzero(is_f).astype(np.int32)
        is_f |= add


@dataclass
class BatchSet:
    """The round's global alert set as CSR: batch b = recs[off[b]:off[b+1]], sender[b]."""

    recs: np.ndarray
    off: np.ndarray
    sender: np.ndarray

    @property
    def n_batches(self):
        return len(self.off) - 1


def _merge_reports(obs_idx, subj_idx, ring, cfg_id, status=DOWN):
    """(observer, subject, ring) triples -> one record per (observer, subject) with a ring mask, grouped into one
    batch per observer (mirrors addAllRingNumber(getRingNumbers(..)), MembershipService.java:486-492).  `status` is a
    scalar or one value per triple (all triples of one (observer, subject) pair carry the same status)."""
    if len(obs_idx) == 0:
        return BatchSet(np.zeros(0, dtype=ALERT_DTYPE), np.zeros(1, dtype=np.int64), np.zeros(0, dtype=np.int32))
    status = np.broadcast_to(np.asarray(status, dtype=np.uint8), (len(obs_idx),))
    key = obs_idx.astype(np.int64) * (1 << 32) + subj_idx.astype(np.int64)
    order = np.argsort(key, kind="stable")
    key, ring, status = key[order], ring[order], status[order]
    uniq, start = np.unique(key, return_index=True)
    mask = np.bitwise_or.reduceat((1 << ring.astype(np.int64)), start).astype(np.uint16)
    recs = np.zeros(len(uniq), dtype=ALERT_DTYPE)
    recs["cfg_id"] = cfg_id
    recs["src"] = (uniq >> 32).astype(np.uint32)
    recs["dst"] = (uniq & 0xFFFFFFFF).astype(np.uint32)
    recs["ring_mask"] = mask
    recs["status"] = status[start]
    senders, bstart = np.unique(recs["src"], return_index=True)
    off = np.concatenate([bstart, [len(recs)]]).astype(np.int64)
    recs["flags"][off[1:] - 1] = FLAG_LAST_IN_BATCH
    return BatchSet(recs, off, senders.astype(np.int32))


def crash_batches(subj, faulty, cfg_id):
    """C1/C2/C4 crash faults: every NON-crashed observer o of a crashed s reports s DOWN on the rings where it
    observes s; crashed observers report nothing (exercises implicit invalidation)."""
    n, K = subj.shape
    is_f = np.zeros(n, dtype=bool)
    is_f[faulty] = True
    o, k = np.nonzero(is_f[subj] & ~is_f[:, None])  # o healthy, subj[o][k] crashed
    return _merge_reports(o.astype(np.int32), subj[o, k], k.astype(np.int32), cfg_id)


def ingress_loss_batches(subj, faulty, cfg_id):
    """C3 one-way failures: every observer of a faulty f reports f DOWN, and every faulty f (deaf, but able to
    send) reports each of its K subjects DOWN."""
    n, K = subj.shape
    is_f = np.zeros(n, dtype=bool)
    is_f[faulty] = True
    o, k = np.nonzero(is_f[subj] | is_f[:, None])
    return _merge_reports(o.astype(np.int32), subj[o, k], k.astype(np.int32), cfg_id)


def churn_batches(obs, member, crashed, joiners, cfg_id):
    """Churn: members crash and new nodes join in the same configuration.  Every healthy member
    that observes a crashed member reports it DOWN on the rings where it observes it; every healthy member that is an
    EXPECTED observer of a joiner (MembershipView.getExpectedObserversOf, R/MembershipView.java:292-322 -- row `joiner`
    of the observer table for a non-member) reports it UP on those rings, as the gatekeepers of the join protocol do
    (R/MembershipService.java:231-254).  Crashed observers report nothing: a joiner or a crashed node whose observer
    crashed too only completes through the implicit invalidation (R/MultiNodeCutDetector.java:137-164).
    obs: [N][K] observer table of the current view, member: [N] flags."""
    n, K = obs.shape
    dead = np.zeros(n, dtype=bool)
    dead[np.asarray(crashed, dtype=np.int64)] = True
    is_member = np.asarray(member) != 0
    assert is_member[np.asarray(crashed, dtype=np.int64)].all() and not is_member[np.asarray(joiners, dtype=np.int64)].any()
    o_all, s_all, k_all, st_all = [], [], [], []
    for nodes, status in ((np.asarray(crashed, dtype=np.int64), DOWN), (np.asarray(joiners, dtype=np.int64), UP)):
        if len(nodes) == 0:
            continue
        o = obs[nodes]  # [len][K] observers of each node, ring by ring
        ok = (o >= 0) & is_member[np.clip(o, 0, n - 1)] & ~dead[np.clip(o, 0, n - 1)]
        i, k = np.nonzero(ok)
        o_all.append(o[i, k])
        s_all.append(nodes[i])
        k_all.append(k)
        st_all.append(np.full(len(i), status, dtype=np.uint8))
    if not o_all:
        return _merge_reports(np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.int32), cfg_id)
    return _merge_reports(np.concatenate(o_all).astype(np.int32), np.concatenate(s_all).astype(np.int32),
                          np.concatenate(k_all).astype(np.int32), cfg_id, np.concatenate(st_all))


def build_churn_scenario(obs, member, cfg_id, n_crash, n_join, H, L, seed_fault=1, seed_delivery=2, receivers=None,
                         materialise=True, eligible_joiners=None):
    """`n_crash` seeded members crash and `n_join` seeded non-members join; receivers = the surviving members.
    eligible_joiners: the non-members that may join (default: all of them; a node that was a member before cannot come
    back under its old NodeId, R/MembershipView.java:127-131)."""
    n, K = obs.shape
    is_member = np.asarray(member) != 0
    rng = np.random.Generator(np.random.PCG64(seed_fault))
    members = np.flatnonzero(is_member)
    outsiders = np.flatnonzero(~is_member) if eligible_joiners is None else np.asarray(eligible_joiners, dtype=np.int64)
    assert n_crash <= len(members) and n_join <= len(outsiders)
    crashed = np.sort(rng.permutation(members)[:n_crash]).astype(np.int32)
    joiners = np.sort(rng.permutation(outsiders)[:n_join]).astype(np.int32)
    bs = churn_batches(obs, member, crashed, joiners, cfg_id)
    dead = np.zeros(n, dtype=bool)
    dead[crashed] = True
    survivors = np.flatnonzero(is_member & ~dead).astype(np.int32)
    rx = survivors if receivers is None else np.asarray(receivers, dtype=np.int32)
    sc = Scenario("churn", n, K, H, L, np.sort(np.concatenate([crashed, joiners])).astype(np.int32), rx, bs)
    sc.crashed, sc.joiners = crashed, joiners
    if materialise:
        sc.records, sc.rec_off, sc.n_batches_delivered = deliver(bs, rx, seed_delivery)
    return sc


def deliver(batches, receivers, seed_delivery, loss=0.0, stale_cfg=None, stale_rate=0.0):
    """Every receiver gets every batch once, in a receiver-specific seeded order (paper Fig.11 methodology);
    `loss` drops (receiver, batch) pairs i.i.d.  Returns (records, rec_off[R+1], batches_per_receiver[R]).
    With `stale_rate`, every receiver ALSO gets that fraction of the batches (seeded, at random places of its order) as late
    deliveries from the previous configuration: whole BatchedAlertMessages whose alerts carry `stale_cfg` and are dropped by
    the filter of MembershipService.java:653-657 (their batch end still counts: invalidateFailingEdges runs after every
    batch, R/MembershipService.java:330).  Nothing of the current round is lost to them."""
    B = batches.n_batches
    blen = np.diff(batches.off)
    R = len(receivers)
    out, rec_off, nb = [], np.zeros(R + 1, dtype=np.int64), np.zeros(R, dtype=np.int32)
    for i, r in enumerate(receivers):
        rng = np.random.Generator(np.random.PCG64([int(seed_delivery), int(r)]))
        perm = rng.permutation(B)
        if loss > 0:
            perm = perm[rng.random(B) >= loss]
        late = np.zeros(len(perm), dtype=bool)
        if stale_rate > 0 and B:
            n_late = max(1, int(round(stale_rate * B)))
            perm = np.concatenate([perm, rng.integers(0, B, size=n_late)])
            late = np.concatenate([late, np.ones(n_late, dtype=bool)])
            order = rng.permutation(len(perm))
            perm, late = perm[order], late[order]
        lens = blen[perm]
        tot = int(lens.sum())
        # gather indices: for each delivered batch, off[b] + arange(len)
        starts = np.repeat(batches.off[perm] - np.concatenate([[0], np.cumsum(lens)[:-1]]), lens)
        idx = starts + np.arange(tot, dtype=np.int64)
        recs = batches.recs[idx]
        if stale_rate > 0 and tot:
            recs["cfg_id"][np.repeat(late, lens)] = stale_cfg
        out.append(recs)
        rec_off[i + 1] = rec_off[i] + tot
        nb[i] = len(perm)
    records = np.concatenate(out) if out else np.zeros(0, dtype=ALERT_DTYPE)
    return records, rec_off, nb


def mix64(x):
    """splitmix64 finaliser on uint64 arrays (== rapid::mix64 / gen_mix64 on the device)."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def _perm_keys(seed, receiver):
    with np.errstate(over="ignore"):
        key = mix64(np.uint64(int(seed) & _M64) + np.uint64(int(receiver) & 0xFFFFFFFF))
        rk = [np.uint32(int(mix64(key + np.uint64(i + 1))) >> 32) for i in range(3)]
        keepk = mix64(key ^ np.uint64(0xD1B54A32D192ED03))
    return rk, keepk


def _f16(x, k):
    """csrc/index_kernels.h: gen_f16 -- sixteen pseudo-random bits of a half line number under a round key (24-bit multiplies)."""
    m24 = np.uint32(0xFFFFFF)
    h = ((x ^ k) & m24) * np.uint32(0x9E3779)
    h = h ^ (h >> np.uint32(15))
    h = (h & m24) * np.uint32(0x85EBCB)
    return h >> np.uint32(16)


GEN_LINE_BITS = 3  # (csrc/index_kernels.h: kGenLineBits -- eight batches per line of the delivery order)


def hashed_order(seed, receiver, n_batches):
    """The order in which rapid_sim_generate delivers the round's batches to `receiver` (its node index), csrc/index_kernels.h:
    gen_perm_at -- a permutation with two levels: the batch list in lines of eight consecutive batches, the LINES in a
    receiver-specific pseudo-random order (a two-round alternating Feistel network on [0, b) x [0, a), a = the power of two at or
    above sqrt(lines), b = ceil(lines / a), keyed by mix64(seed + receiver), walked until it lands below `lines`), the eight batches
    of a line one after the other under an affine map of their places drawn from the line and the receiver; the whole walked until
    it lands below n_batches."""
    n = int(n_batches)
    if n <= 1:
        return np.zeros(n, dtype=np.int64)
    rk, _ = _perm_keys(seed, receiver)
    lb = np.uint32(GEN_LINE_BITS)
    lm = np.uint32((1 << GEN_LINE_BITS) - 1)
    lines = (n + (1 << GEN_LINE_BITS) - 1) >> GEN_LINE_BITS
    s = 1
    while s < 16 and (1 << (2 * s)) < lines:
        s += 1
    mask_r = np.uint32((1 << s) - 1)
    b = np.uint32(max(1, (lines + (1 << s) - 1) >> s))

    def line_at(q):
        out = np.zeros(len(q), dtype=np.uint32)
        todo = np.arange(len(q))
        x = q.copy()
        while len(todo):
            r, l = x & mask_r, x >> np.uint32(s)
            l = l + ((_f16(r, rk[0]) * b) >> np.uint32(16))
            l = np.where(l >= b, l - b, l)
            r = (r + _f16(l, rk[1])) & mask_r
            x = (l << np.uint32(s)) | r
            done = x < lines
            out[todo[done]] = x[done]
            todo, x = todo[~done], x[~done]
        return out

    x = np.arange(n, dtype=np.uint32)
    out = np.zeros(n, dtype=np.int64)
    todo = np.arange(n)
    with np.errstate(over="ignore"):
        while len(todo):
            Q = line_at(x >> lb)
            h = _f16(Q & np.uint32(0xFFFF), rk[2] ^ (Q >> np.uint32(16)))
            T = ((((x & lm) ^ (h >> np.uint32(8))) * ((h & np.uint32(6)) | np.uint32(1))) + (h >> np.uint32(3))) & lm
            x = (Q << lb) | T
            done = x < n
            out[todo[done]] = x[done]
            todo, x = todo[~done], x[~done]
    return out


def delivered_mask(seed, receiver, keep):
    """Which batches reach `receiver` under the per-batch thresholds `keep` (uint32; None: all) -- gen_delivered on the device."""
    if keep is None:
        return None
    keep = np.asarray(keep, dtype=np.uint32)
    _, keepk = _perm_keys(seed, receiver)
    with np.errstate(over="ignore"):
        draw = (mix64(keepk + np.arange(len(keep), dtype=np.uint64)) >> np.uint64(32)).astype(np.uint32)
    return draw <= keep


def deliver_hashed(batches, receivers, seed, keep=None):
    """The host statement of rapid_sim_generate: every receiver gets every batch once, in hashed_order; the batch end is on
    the last record of every batch.  keep: per-batch delivery thresholds -- the places of a batch that does not reach a
    receiver hold EMPTY records (all zero: no ring, no batch end), so every stream has the same length.
    Returns (records, rec_off[R+1], batches_per_receiver[R]) like deliver()."""
    B = batches.n_batches
    blen = np.diff(batches.off)
    assert (blen > 0).all(), "a BatchedAlertMessage is never empty"
    R = len(receivers)
    A = int(batches.off[-1])
    base = batches.recs.copy()
    base["flags"] = 0
    base["flags"][batches.off[1:] - 1] = FLAG_LAST_IN_BATCH  # a batch ends with its last alert, whatever the set's flags say
    out, nb = [], np.full(R, B, dtype=np.int32)
    for i, r in enumerate(receivers):
        perm = hashed_order(seed, int(r), B)
        lens = blen[perm]
        starts = np.repeat(batches.off[perm] - np.concatenate([[0], np.cumsum(lens)[:-1]]), lens)
        recs = base[starts + np.arange(A, dtype=np.int64)]
        got = delivered_mask(seed, int(r), keep)
        if got is not None:
            recs[np.repeat(~got[perm], lens)] = np.zeros(1, dtype=ALERT_DTYPE)
            nb[i] = int(got.sum())
        out.append(recs)
    records = np.concatenate(out) if out else np.zeros(0, dtype=ALERT_DTYPE)
    return records, np.arange(R + 1, dtype=np.int64) * A, nb


@dataclass
class Scenario:
    name: str
    n: int
    K: int
    H: int
    L: int
    faulty: np.ndarray
    receivers: np.ndarray
    batches: BatchSet
    records: np.ndarray = None
    rec_off: np.ndarray = None
    n_batches_delivered: np.ndarray = None


CONFIGS = {
    # name: (N, K, H, L, fault fraction / count, kind)
    "C1": dict(n=50, K=3, H=3, L=1, f=1, kind="crash"),
    "C2": dict(n=2000, K=10, H=9, L=4, f=20, kind="crash"),
    "C3a": dict(n=10000, K=10, H=9, L=4, f=500, kind="ingress"),
    "C3b": dict(n=10000, K=10, H=9, L=4, f=500, kind="ingress_closed"),
    "C4": dict(n=100000, K=10, H=9, L=4, f=1000, kind="crash"),
    # C5 = BASELINE configs[4]: continuous churn, a fresh 1 % of the members crashes in every round (plus joins), 1 % of the
    # delivered records still carry the previous configuration id; see StreamingChurn
    "C5": dict(n=1000000, K=10, H=9, L=4, f=10000, kind="stream"),
}


def build_scenario(name, subj, cfg_id, seed_fault=1, seed_delivery=2, receivers=None, loss=0.0, n=None, f=None,
                   H=None, L=None, kind=None, materialise=True):
    """Builds one of the BASELINE.json configurations on the topology `subj` ([N][K] predecessor table)."""
    c = dict(CONFIGS.get(name, {}))
    for k_, v in (("n", n), ("f", f), ("H", H), ("L", L), ("kind", kind)):
        if v is not None:
            c[k_] = v
    N, K = subj.shape
    c.setdefault("n", N)
    c.setdefault("K", K)
    assert c["n"] == N, "topology size does not match scenario"
    faulty = pick_faulty(N, c["f"], seed_fault)
    if c["kind"] == "ingress_closed":
        faulty = close_fault_set(faulty, subj, c["L"])
    if c["kind"] == "crash":
        bs = crash_batches(subj, faulty, cfg_id)
    else:
        bs = ingress_loss_batches(subj, faulty, cfg_id)
    is_f = np.zeros(N, dtype=bool)
    is_f[faulty] = True
    healthy = np.flatnonzero(~is_f).astype(np.int32)
    rx = healthy if receivers is None else np.asarray(receivers, dtype=np.int32)
    sc = Scenario(name, N, K, c["H"], c["L"], faulty, rx, bs)
    if materialise:
        sc.records, sc.rec_off, sc.n_batches_delivered = deliver(bs, rx, seed_delivery, loss)
    return sc


def with_late_batches(batches, prev_cfg, rate, seed):
    """The round's batch set plus `rate` x its batches as LATE DELIVERIES of the previous configuration ("stale-cfg
    alerts from round r-1 are interleaved at rate 1 %"): seeded copies of whole BatchedAlertMessages whose alerts carry `prev_cfg`
    and are dropped per delivery (R/MembershipService.java:653-657; their batch end still counts, :330).  They are additional
    batches -- nothing of the current round is lost to them -- and, as members of the batch set, they fall at receiver-specific
    places of every receiver's delivery order (rapid_sim_generate / deliver_hashed)."""
    B = batches.n_batches
    if prev_cfg is None or rate <= 0 or B == 0:
        return batches
    rng = np.random.Generator(np.random.PCG64([int(seed) & _M64, 4242]))
    pick = np.sort(rng.integers(0, B, size=max(1, int(round(rate * B)))))
    lens = np.diff(batches.off)[pick]
    starts = np.repeat(batches.off[pick] - np.concatenate([[0], np.cumsum(lens)[:-1]]), lens)
    late = batches.recs[starts + np.arange(int(lens.sum()), dtype=np.int64)].copy()
    late["cfg_id"] = prev_cfg
    off = np.concatenate([batches.off, batches.off[-1] + np.cumsum(lens)]).astype(np.int64)
    return BatchSet(np.concatenate([batches.recs, late]), off, np.concatenate([batches.sender, batches.sender[pick]]))


class StreamingChurn:
    """C5: rounds of continuous churn over one population (BASELINE configs[4]).  Every round takes the CURRENT
    view (observer table + membership, after the previous round's cut was applied), crashes `crash_frac` of the members
    and lets `join_frac` x members outsiders join, delivers the round's batches to a seeded sample of the surviving
    members, and interleaves late deliveries from the PREVIOUS configuration (`stale_rate` x the round's batches, carrying the
    previous configuration id: filtered by R/MembershipService.java:653-657).  The registry must hold the
    joiners of all rounds as non-members: `Population.make(n_members + spare)`."""

    def __init__(self, H, L, crash_frac=0.01, join_frac=0.005, stale_rate=0.01, receivers_per_round=None, seed=5):
        self.H, self.L = H, L
        self.crash_frac, self.join_frac, self.stale_rate = crash_frac, join_frac, stale_rate
        self.receivers_per_round = receivers_per_round
        self.seed = seed
        self.round = 0
        self.prev_cfg = None
        self.ever_member = None  # identifiers are never pruned (quirk Q5): a node joins once

    def next_round_batches(self, obs, member, cfg_id):
        """The round WITHOUT materialised deliveries -- what rapid_sim_generate / rapid_sim_round_tiled take: the scenario
        (receivers = every surviving member, batches = the round's own alerts) and the batch set to deliver (+ the late
        deliveries of the previous configuration as additional batches).  -> (scenario, batch set to deliver)."""
        is_member = np.asarray(member) != 0
        self.ever_member = is_member.copy() if self.ever_member is None else (self.ever_member | is_member)
        fresh = np.flatnonzero(~self.ever_member)
        n_members = int(is_member.sum())
        n_crash = max(1, int(round(self.crash_frac * n_members)))
        n_join = min(len(fresh), int(round(self.join_frac * n_members)))
        sc = build_churn_scenario(obs, member, cfg_id, n_crash, n_join, self.H, self.L, seed_fault=self.seed + 1000 * self.round,
                                  materialise=False, eligible_joiners=fresh)
        deliver_set = with_late_batches(sc.batches, self.prev_cfg, self.stale_rate, self.seed + 31 * self.round)
        self.prev_cfg = cfg_id
        self.round += 1
        return sc, deliver_set

    def next_round(self, obs, member, cfg_id, receivers=None):
        is_member = np.asarray(member) != 0
        self.ever_member = is_member.copy() if self.ever_member is None else (self.ever_member | is_member)
        fresh = np.flatnonzero(~self.ever_member)
        n_members = int(is_member.sum())
        n_out = len(fresh)
        n_crash = max(1, int(round(self.crash_frac * n_members)))
        n_join = min(n_out, int(round(self.join_frac * n_members)))
        sc = build_churn_scenario(obs, member, cfg_id, n_crash, n_join, self.H, self.L, seed_fault=self.seed + 1000 * self.round,
                                  materialise=False, eligible_joiners=fresh)
        rx = sc.receivers
        if receivers is not None:
            rx = np.asarray(receivers, dtype=np.int32)
        elif self.receivers_per_round is not None and self.receivers_per_round < len(rx):
            rng = np.random.Generator(np.random.PCG64([self.seed, self.round, 77]))
            rx = np.sort(rng.permutation(rx)[: self.receivers_per_round]).astype(np.int32)
        sc.receivers = rx
        stale = self.prev_cfg is not None and self.stale_rate > 0
        sc.records, sc.rec_off, sc.n_batches_delivered = deliver(sc.batches, rx, self.seed + 31 * self.round,
                                                                 stale_cfg=self.prev_cfg if stale else None,
                                                                 stale_rate=self.stale_rate if stale else 0.0)
        self.prev_cfg = cfg_id
        self.round += 1
        return sc
