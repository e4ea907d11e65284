"""Host mirror of the consensus entry points of include/rapid_mi355x.h: one node's FastPaxos object (fast round +
classic-Paxos recovery, R/FastPaxos.java + R/Paxos.java), the coordinator rule as a function, the closed form of one
classic round over a whole population, and the wire forms of the five consensus messages.  Pure host code: works
without a GPU (the library itself still has to be built)."""
import ctypes as C
from dataclasses import dataclass
from typing import Tuple

import numpy as np

from . import _native as N
from .engine import _addr

FAST_ROUND_PHASE2B, PHASE1A, PHASE1B, PHASE2A, PHASE2B = 5, 6, 7, 8, 9  # RapidRequest.content field numbers
BROADCAST = -1


def _check(rc, what):
    if rc != N.OK:
        N.raise_for(rc, what)


@dataclass(frozen=True)
class Message:
    """rapid_consensus_msg + its endpoint list.  rnd is Phase1a.rank / Phase1b, 2a, 2b.rnd; vrnd is Phase1b's."""
    kind: int
    sender: int
    configurationId: int
    rnd: Tuple[int, int] = (0, 0)
    vrnd: Tuple[int, int] = (0, 0)
    endpoints: Tuple[int, ...] = ()
    dest: int = BROADCAST

    def _c(self):
        head = N.ConsensusMsg(self.kind, self.sender, self.configurationId, N.Rank(*self.rnd), N.Rank(*self.vrnd),
                              len(self.endpoints), self.dest)
        eps = np.asarray(self.endpoints, dtype=np.int32)
        return head, eps

    @staticmethod
    def _from_c(head, eps):
        return Message(head.kind, head.sender, head.config_id, (head.rnd.round, head.rnd.node_index),
                       (head.vrnd.round, head.vrnd.node_index), tuple(int(e) for e in eps[: head.n_endpoints]), head.dest)


class FastPaxos:
    """One node's consensus instance for one configuration (R/FastPaxos.java:62-87).  `rank_index` stands in for
    myAddr.hashCode() (R/Paxos.java:102).  Outgoing messages are collected: poll() returns them oldest first."""

    def __init__(self, my_addr, configuration_id, membership_size, rank_index=None):
        self._lib = N.lib()
        self._h = C.c_void_p()
        rank_index = my_addr + 2 if rank_index is None else rank_index
        _check(self._lib.rapid_consensus_create(my_addr, rank_index, configuration_id, membership_size, C.byref(self._h)),
               "rapid_consensus_create")
        self.membership_size = membership_size

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                self._lib.rapid_consensus_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def propose(self, proposal):  # :95-110
        e = np.ascontiguousarray(proposal, dtype=np.int32)
        _check(self._lib.rapid_consensus_propose(self._h, _addr(e) if len(e) else None, len(e)), "rapid_consensus_propose")

    def handleMessages(self, msg: Message):  # :163-185
        head, eps = msg._c()
        _check(self._lib.rapid_consensus_handle(self._h, C.byref(head), _addr(eps) if len(eps) else None), "rapid_consensus_handle")

    def startClassicPaxosRound(self):  # :190-196
        _check(self._lib.rapid_consensus_start_classic_round(self._h), "rapid_consensus_start_classic_round")

    def startPhase1a(self, round_):  # R/Paxos.java:98-111
        _check(self._lib.rapid_consensus_start_phase1a(self._h, round_), "rapid_consensus_start_phase1a")

    def poll(self, cap=64):
        """All queued outgoing messages, oldest first."""
        out = []
        while True:
            head = N.ConsensusMsg()
            eps = np.empty(max(cap, 1), dtype=np.int32)
            got = C.c_int32(0)
            rc = self._lib.rapid_consensus_poll(self._h, C.byref(head), _addr(eps), cap, C.byref(got))
            if rc == N.ECAPACITY:
                cap = head.n_endpoints
                continue
            _check(rc, "rapid_consensus_poll")
            if not got.value:
                return out
            out.append(Message._from_c(head, eps))

    def decision(self):
        """The decided value, or None."""
        cap = 1024
        while True:
            out = np.empty(cap, dtype=np.int32)
            n = C.c_int32(0)
            rc = self._lib.rapid_consensus_decision(self._h, _addr(out), cap, C.byref(n))
            if rc == N.ESTATE:
                return None
            if rc == N.ECAPACITY:
                cap = n.value
                continue
            _check(rc, "rapid_consensus_decision")
            return tuple(out[: n.value].tolist())

    def getRandomDelayMs(self, u, base_delay_ms):  # :201-204
        return fallback_delay_ms(self.membership_size, base_delay_ms, u)


def fallback_delay_ms(membership_size, base_delay_ms, u):
    out = C.c_int64(0)
    _check(N.lib().rapid_consensus_fallback_delay_ms(membership_size, base_delay_ms, u, C.byref(out)),
           "rapid_consensus_fallback_delay_ms")
    return out.value


def select_proposal(membership_size, messages):
    """Paxos.selectProposalUsingCoordinatorRule (R/Paxos.java:271-328).  messages: [(vrnd, vval)] in arrival order.
    Returns the chosen value (the empty tuple if no message carries one)."""
    vrnd = (N.Rank * max(len(messages), 1))(*[N.Rank(*m[0]) for m in messages])
    off = np.zeros(len(messages) + 1, dtype=np.int32)
    off[1:] = np.cumsum([len(m[1]) for m in messages])
    vals = np.asarray([e for m in messages for e in m[1]], dtype=np.int32)
    chosen = C.c_int32(-2)
    rc = N.lib().rapid_paxos_select_proposal(membership_size, C.cast(vrnd, C.c_void_p), _addr(off), _addr(vals) if len(vals) else None,
                                             len(messages), C.byref(chosen))
    _check(rc, "rapid_paxos_select_proposal")
    return tuple(messages[chosen.value][1]) if chosen.value >= 0 else ()


def classic_round_population(membership_size, vote_key, voted, arrival=None):
    """One classic round over a whole population (rapid_classic_round_population).  -> dict of the result fields."""
    vote_key = np.ascontiguousarray(vote_key, dtype=np.uint64)
    voted = np.ascontiguousarray(voted, dtype=np.uint8)
    assert len(vote_key) == len(voted)
    arr = None if arrival is None else np.ascontiguousarray(arrival, dtype=np.int32)
    assert arr is None or len(arr) == len(voted)
    res = N.ClassicRoundResult()
    rc = N.lib().rapid_classic_round_population(membership_size, len(voted), _addr(vote_key) if len(voted) else None,
                                                _addr(voted) if len(voted) else None, None if arr is None else _addr(arr), C.byref(res))
    _check(rc, "rapid_classic_round_population")
    return {"decided": bool(res.decided), "chosen_acceptor": res.chosen_acceptor, "promises_used": res.promises_used,
            "rule": res.rule, "messages": res.messages}


def classic_rounds_population(membership_size, vote_key, voted, starts, rank_index=None, schedule=None, drop=None, n_steps=0, seed=0,
                              loss=0.0):
    """The recovery with CONCURRENT coordinators and message loss over a population of whole consensus instances
    (rapid_classic_rounds_population).  starts: [(step, acceptor, round)] ascending by step; schedule / drop: explicit delivery
    (node whose oldest queued message is handled at each step / lost instead), or None for a seeded run with loss probability
    `loss`.  -> dict of the result fields + "decided_vote_of": per acceptor, an acceptor whose vote it decided (-1: undecided)."""
    vote_key = np.ascontiguousarray(vote_key, dtype=np.uint64)
    voted = np.ascontiguousarray(voted, dtype=np.uint8)
    n = len(voted)
    assert len(vote_key) == n
    st = (N.ClassicStart * max(len(starts), 1))()
    for j, (step, a, rnd) in enumerate(starts):
        st[j].step, st[j].acceptor, st[j].round = int(step), int(a), int(rnd)
    ri = None if rank_index is None else np.ascontiguousarray(rank_index, dtype=np.int32)
    sch = None if schedule is None else np.ascontiguousarray(schedule, dtype=np.int32)
    dr = None if drop is None else np.ascontiguousarray(drop, dtype=np.uint8)
    if sch is not None:
        n_steps = len(sch)
        assert dr is None or len(dr) == n_steps
    res = N.ClassicRoundsResult()
    dec = np.full(max(n, 1), -2, dtype=np.int32)
    rc = N.lib().rapid_classic_rounds_population(membership_size, n, _addr(vote_key) if n else None, _addr(voted) if n else None,
                                                 None if ri is None else _addr(ri), C.cast(st, C.c_void_p), len(starts),
                                                 None if sch is None or not len(sch) else _addr(sch), None if dr is None or not len(dr) else _addr(dr),
                                                 int(n_steps), int(seed), float(loss), C.byref(res), _addr(dec))
    _check(rc, "rapid_classic_rounds_population")
    return {"decided_nodes": res.decided_nodes, "agreed": bool(res.agreed), "chosen_acceptor": res.chosen_acceptor, "lost": res.lost,
            "steps": res.steps, "undelivered": res.undelivered, "sent": list(res.sent), "delivered": list(res.delivered),
            "decided_vote_of": dec[:n].tolist()}


def classic_round_from_results(membership_size, emit_batch, fingerprint, arrival=None):
    """The recovery of a simulated population whose fast round found no quorum, from ClusterSimulation.results():
    receiver i voted iff emit_batch[i] >= 0, for the proposal with fingerprint[i].  -> (result dict, index of a
    receiver whose proposal is the decided cut, or None)."""
    voted = (np.asarray(emit_batch) >= 0).astype(np.uint8)
    res = classic_round_population(membership_size, fingerprint, voted, arrival)
    return res, (res["chosen_acceptor"] if res["decided"] else None)


def decode_message(endpoint_map, kind, payload, cap=4096):
    """payload of a RapidRequest of content case `kind` (5..9) -> Message"""
    arr = np.frombuffer(payload, dtype=np.uint8) if len(payload) else np.zeros(1, dtype=np.uint8)
    head = N.ConsensusMsg()
    eps = np.empty(max(cap, 1), dtype=np.int32)
    rc = N.lib().rapid_decode_consensus_message(endpoint_map._h, kind, _addr(arr), len(payload), C.byref(head), _addr(eps), cap)
    _check(rc, "rapid_decode_consensus_message")
    return Message._from_c(head, eps)


def encode_request(endpoint_map, msg: Message):
    """Message -> bytes of the serialized RapidRequest that carries it"""
    head, eps = msg._c()
    cap = 256 + 64 * len(eps)
    while True:
        out = np.empty(cap, dtype=np.uint8)
        n = C.c_int64(0)
        rc = N.lib().rapid_encode_consensus_request(endpoint_map._h, C.byref(head), _addr(eps) if len(eps) else None, _addr(out), cap,
                                                    C.byref(n))
        if rc == N.ECAPACITY:
            cap = n.value
            continue
        _check(rc, "rapid_encode_consensus_request")
        return out[: n.value].tobytes()
