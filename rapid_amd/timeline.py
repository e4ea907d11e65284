"""Alert producer model: WHEN the alerts of a round exist and in which order they reach a receiver.

scenarios.py feeds the engine every batch in a seeded random order per receiver.  This module derives the batches and
their order from the reference's own timers instead, so that a round has a protocol time line:

  crash of s at t  ->  each observer's PingPongFailureDetector counts FAILURE_THRESHOLD failed probes, one probe per
  detector tick (R/monitoring/impl/PingPongFailureDetector.java:41,70-85,122-125; one detector per entry of
  getSubjectsOf, scheduleAtFixedRate(0, failureDetectorInterval), R/MembershipService.java:697-706)
  ->  edgeFailureNotification enqueues one AlertMessage with all ring numbers (:472-495, :572-581)
  ->  the AlertBatcher, ticking every batchingWindow, sends the queue as one BatchedAlertMessage at its first tick
  more than one window after the LAST enqueue (:147-148, :613-637)
  ->  the batch reaches receiver r `latency(sender, r)` later; r handles batches in arrival order.

Closed forms, vectorised; the literal event-by-event restatement is oracle/timeline_oracle.py and the two are compared
in tests/test_timeline.py.  Host-side workload tooling like scenarios.py (numpy): the engine consumes what it produces.
Time is integer milliseconds; events with equal timestamps are ordered probe callback < detector tick < batcher tick.
"""
from dataclasses import dataclass

import numpy as np

from .scenarios import ALERT_DTYPE, DOWN, FLAG_LAST_IN_BATCH, BatchSet, splitmix64

NEVER = np.iinfo(np.int64).max


@dataclass
class ProducerModel:
    fd_interval_ms: int = 1000     # MembershipService.DEFAULT_FAILURE_DETECTOR_INTERVAL_IN_MS, :77
    failure_threshold: int = 10    # PingPongFailureDetector.FAILURE_THRESHOLD, :41
    batching_window_ms: int = 100  # MembershipService.BATCHING_WINDOW_IN_MS, :75
    probe_fail_ms: int = 1         # how long a probe to a dead node takes to fail (<= DEFAULT_GRPC_PROBE_TIMEOUT = 1000)


@dataclass
class LatencyModel:
    """One-way delay of a unicast, fixed per (sender, receiver) pair: base + a seeded value in [0, jitter]."""
    base_ms: int = 1
    jitter_ms: int = 4
    seed: int = 11

    def delay(self, sender, receiver, n):
        pair = np.asarray(sender, dtype=np.uint64) * np.uint64(n) + np.asarray(receiver, dtype=np.uint64)
        return self.base_ms + (splitmix64(self.seed, pair) % np.uint64(self.jitter_ms + 1)).astype(np.int64)


def notifications(model, subj, crash_ms, start_ms):
    """-> (observer, ring, subject, t_notify) of every detector instance that notifies, sorted by (observer, t, ring).
    crash_ms[n] = NEVER for nodes that stay up; start_ms[n] = phase of n's timers."""
    subj = np.asarray(subj)
    crash_ms = np.asarray(crash_ms, dtype=np.int64)
    start_ms = np.asarray(start_ms, dtype=np.int64)
    I, T, d = model.fd_interval_ms, model.failure_threshold, model.probe_fail_ms
    o, k = np.nonzero(crash_ms[subj] != NEVER)
    s = subj[o, k]
    # first tick at or after the crash: probes sent from then on fail
    j0 = np.maximum(0, -((start_ms[o] - crash_ms[s]) // I))
    # a failure is counted by tick m if it came back by then: the probe of tick j counts iff (m - j) * I >= d, j < m
    q = max(1, -(-d // I))
    m = j0 + T + q - 1
    t = start_ms[o] + m * I
    # the observer must still run then; its callbacks before that all ran, because it was alive even later
    ok = t < crash_ms[o]
    o, k, s, t = o[ok], k[ok], s[ok], t[ok]
    order = np.lexsort((k, t, o))
    return o[order].astype(np.int32), k[order].astype(np.int32), s[order].astype(np.int32), t[order]


def _first_tick_after(start, window, t):
    """first batcher tick b = start + i * window (i >= 0) with b - t > window"""
    i = np.maximum(0, (t + window - start) // window + 1)
    return start + i * window


def batches(model, subj, crash_ms, start_ms, cfg_id):
    """-> (BatchSet, send_ms[batch]): the BatchedAlertMessages of the run in (send time, sender) order.  One record
    per notifying detector instance -- an observer that watches s on m rings runs m detectors for it and sends the
    same alert (all m ring numbers) m times, as the reference does."""
    subj = np.asarray(subj)
    crash_ms = np.asarray(crash_ms, dtype=np.int64)
    start_ms = np.asarray(start_ms, dtype=np.int64)
    W = model.batching_window_ms
    o, k, s, t = notifications(model, subj, crash_ms, start_ms)
    if len(o) == 0:
        return BatchSet(np.zeros(0, dtype=ALERT_DTYPE), np.zeros(1, dtype=np.int64), np.zeros(0, dtype=np.int32)), np.zeros(0, dtype=np.int64)
    # ring mask of (observer, subject): every ring on which the observer watches that subject
    same = subj[o] == s[:, None]
    mask = (same.astype(np.int64) << np.arange(subj.shape[1], dtype=np.int64)).sum(axis=1).astype(np.uint16)
    # group the enqueue events of each observer into batches: a batch closes at the first tick more than W after its
    # last enqueue; an enqueue up to and including that tick's time joins (and moves the tick)
    batch_of = np.empty(len(o), dtype=np.int64)
    send, sender = [], []
    first = np.flatnonzero(np.r_[True, o[1:] != o[:-1]])
    bounds = np.r_[first, len(o)]
    for a, b in zip(bounds[:-1], bounds[1:]):
        ob, st = int(o[a]), int(start_ms[o[a]])
        i = a
        while i < b:
            last, j = int(t[i]), i + 1
            while True:
                f = int(_first_tick_after(st, W, last))
                if j < b and (int(t[j]) <= f or last <= 0):  # lastEnqueueTimestamp > 0 is part of the Java's test (:620)
                    last, j = int(t[j]), j + 1
                    continue
                break
            if last <= 0 or f >= int(crash_ms[ob]):
                batch_of[i:j] = -1  # never sent (the sender stopped first)
            else:
                batch_of[i:j] = len(send)
                send.append(f)
                sender.append(ob)
            i = j
    send = np.asarray(send, dtype=np.int64)
    sender = np.asarray(sender, dtype=np.int32)
    keep = batch_of >= 0
    o, s, mask, batch_of = o[keep], s[keep], mask[keep], batch_of[keep]
    # global (send time, sender) order; records keep their enqueue order inside a batch
    border = np.lexsort((sender, send))
    rank = np.empty(len(border), dtype=np.int64)
    rank[border] = np.arange(len(border))
    rorder = np.argsort(rank[batch_of], kind="stable")
    recs = np.zeros(len(rorder), dtype=ALERT_DTYPE)
    recs["cfg_id"] = cfg_id
    recs["src"], recs["dst"], recs["ring_mask"], recs["status"] = o[rorder], s[rorder], mask[rorder], DOWN
    counts = np.bincount(rank[batch_of], minlength=len(border))
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    recs["flags"][off[1:] - 1] = FLAG_LAST_IN_BATCH
    return BatchSet(recs, off, sender[border]), send[border]


def deliver_timed(bs, send_ms, receivers, n, latency):
    """Every receiver gets every batch once, in arrival order (ties: lower sender index first).
    -> (records, rec_off[R+1], arrival_ms (concatenated per receiver), arr_off[R+1])"""
    B = bs.n_batches
    blen = np.diff(bs.off)
    R = len(receivers)
    out = []
    rec_off = np.zeros(R + 1, dtype=np.int64)
    arrival = np.empty(R * B, dtype=np.int32)  # milliseconds since the start of the run
    for i, r in enumerate(receivers):
        arr = send_ms + latency.delay(bs.sender, np.full(B, r), n)
        perm = np.lexsort((bs.sender, arr))
        lens = blen[perm]
        tot = int(lens.sum())
        starts = np.repeat(bs.off[perm] - np.concatenate([[0], np.cumsum(lens)[:-1]]), lens)
        out.append(bs.recs[starts + np.arange(tot, dtype=np.int64)])
        rec_off[i + 1] = rec_off[i] + tot
        arrival[i * B:(i + 1) * B] = arr[perm]
    records = np.concatenate(out) if out else np.zeros(0, dtype=ALERT_DTYPE)
    return records, rec_off, arrival, np.arange(R + 1, dtype=np.int64) * B


def proposal_times(emit_batch, arrival, arr_off):
    """When each receiver announced its proposal: the arrival time of the batch that made it (NEVER if it did not)."""
    emit_batch = np.asarray(emit_batch)
    t = np.full(len(emit_batch), NEVER, dtype=np.int64)
    has = emit_batch >= 0
    t[has] = arrival[arr_off[:-1][has] + emit_batch[has]]
    return t


def fast_round_decision_times(proposal_ms, receivers, vote_key, n, latency, chunk=256):
    """When each receiver learns the fast-round decision: every receiver broadcasts its vote when it announces
    (R/MembershipService.java:341-349 -> FastPaxos.propose), node r decides on the arrival of the vote that completes
    a quorum N - floor((N-1)/4) of identical votes out of at least that many votes (R/FastPaxos.java:142-150).
    -> decision_ms[R] (NEVER where no quorum forms)."""
    receivers = np.asarray(receivers)
    proposal_ms = np.asarray(proposal_ms, dtype=np.int64)
    vote_key = np.asarray(vote_key)
    R = len(receivers)
    quorum = n - (n - 1) // 4
    voted = proposal_ms != NEVER
    out = np.full(R, NEVER, dtype=np.int64)
    if voted.sum() < quorum:
        return out
    keys, counts = np.unique(vote_key[voted], return_counts=True)
    if counts.max() < quorum:
        return out
    win = voted & (vote_key == keys[np.argmax(counts)])
    src, t0 = receivers[win], proposal_ms[win]
    for a in range(0, R, chunk):
        rx = receivers[a:a + chunk]
        arr = t0[None, :] + latency.delay(src[None, :], rx[:, None], n)  # the winning votes at each receiver
        # the vote that brings the winner's count to the quorum also finds the total at or above it (:146-147)
        out[a:a + chunk] = np.partition(arr, quorum - 1, axis=1)[:, quorum - 1]
    return out


def classic_round_times(proposal_ms, receivers, vote_key, n, latency, base_delay_ms, u):
    """The recovery round on the time line, for a population whose fast round found no quorum.  Every node arms its
    recovery timer when it proposes: proposal time + base delay + an exponential jitter with mean N seconds
    (R/FastPaxos.java:76,107-109,201-204; u[i] = the node's uniform draw) -- so the FIRST timer of N fires about a second
    after the proposals.  That node is the coordinator: Phase1a out, Phase1b back (their arrival order at the coordinator
    is what the coordinator rule sees), Phase2a out, Phase2b all-to-all; a node decides on the arrival of the
    (floor(N/2)+1)-th Phase2b.  Nodes that never proposed arm no timer (R/MembershipService.java:347-349) but answer.
    -> dict(coordinator = index into receivers, start_ms, arrival = Phase1b arrival order (indices into receivers),
            result = consensus.classic_round_population(...), phase2a_ms, decision_ms[R] (NEVER if undecided))"""
    from . import consensus as CS
    receivers = np.asarray(receivers)
    proposal_ms = np.asarray(proposal_ms, dtype=np.int64)
    R = len(receivers)
    voted = proposal_ms != NEVER
    out = dict(coordinator=-1, start_ms=NEVER, arrival=None, result=None, phase2a_ms=NEVER, decision_ms=np.full(R, NEVER, dtype=np.int64))
    if not voted.any():
        return out
    u = np.asarray(u, dtype=np.float64)
    jitter = (-1000.0 * np.log(1.0 - u) * n).astype(np.int64)  # (long)(-1000 ln(1-u) / (1/N))
    fire = np.where(voted, proposal_ms + base_delay_ms + jitter, NEVER)
    c = int(np.lexsort((receivers, fire))[0])
    t0 = int(fire[c])
    node_c = np.full(R, receivers[c])
    t_1a = t0 + latency.delay(node_c, receivers, n)      # Phase1a reaches acceptor i
    t_1b = t_1a + latency.delay(receivers, node_c, n)    # its Phase1b reaches the coordinator
    arrival = np.lexsort((receivers, t_1a, t_1b))        # ties: the acceptor that answered first, then the lower index
    res = CS.classic_round_population(n, vote_key, voted, arrival)
    out.update(coordinator=c, start_ms=t0, arrival=arrival, result=res)
    if not res["decided"]:
        return out
    t_2a = int(t_1b[arrival[res["promises_used"] - 1]])  # the promise that completes the coordinator's choice
    t_acc = t_2a + latency.delay(node_c, receivers, n)   # Phase2a reaches acceptor i, which broadcasts its Phase2b at once
    need = n // 2 + 1
    dec = np.empty(R, dtype=np.int64)
    for a in range(0, R, 256):
        rx = receivers[a:a + 256]
        arr = t_acc[None, :] + latency.delay(receivers[None, :], rx[:, None], n)
        dec[a:a + 256] = np.partition(arr, need - 1, axis=1)[:, need - 1]
    out.update(phase2a_ms=t_2a, decision_ms=dec)
    return out


def engine_round_on_the_time_line(sim, subj, crash_ms, start_ms, cfg_id, n, model=None, latency=None, receivers=None):
    """One round of the engine driven by the reference's timers instead of a seeded shuffle: the alert producer model turns
    crash times into BatchedAlertMessages with send times, every receiver gets them in arrival order, the GPU tallies the
    streams, and the announcing batch of every receiver is mapped back onto the protocol time line -- proposal time, fast-round
    decision time (R/FastPaxos.java:142-150).  `sim`: a rapid_amd.engine.ClusterSimulation of an engine whose view has the
    topology `subj`.  -> dict(emit_batch, num_proposals, prop_count, fingerprint, proposal_ms, decision_ms, receivers,
    batches, send_ms, records, rec_off, arrival, arr_off, time_to_stable_cut_ms = last decision - first crash, or None)."""
    model = model or ProducerModel()
    latency = latency or LatencyModel()
    crash_ms = np.asarray(crash_ms, dtype=np.int64)
    bs, send = batches(model, subj, crash_ms, start_ms, cfg_id)
    rx = np.flatnonzero(crash_ms == NEVER).astype(np.int32) if receivers is None else np.asarray(receivers, dtype=np.int32)
    records, rec_off, arrival, arr_off = deliver_timed(bs, send, rx, n, latency)
    sim.load_streams(records, rec_off)
    sim.set_alert_set(bs.recs, trust_copies=True)  # the deliveries are copies of these batches by construction
    sim.tally()
    emit, nprop, pcount, fp = sim.results()
    t_prop = proposal_times(emit, arrival, arr_off)
    t_dec = fast_round_decision_times(t_prop, rx, fp, n, latency)
    crashed = crash_ms[crash_ms != NEVER]
    ttsc = None
    if len(crashed) and np.all(t_dec != NEVER) and len(t_dec):
        ttsc = int(t_dec.max() - crashed.min())
    return dict(emit_batch=emit, num_proposals=nprop, prop_count=pcount, fingerprint=fp, proposal_ms=t_prop, decision_ms=t_dec,
                receivers=rx, batches=bs, send_ms=send, records=records, rec_off=rec_off, arrival=arrival, arr_off=arr_off,
                time_to_stable_cut_ms=ttsc)
