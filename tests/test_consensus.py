"""Consensus at one node: rapid_amd/csrc/consensus.h through the C ABI, against the restatement of
R/Paxos.java + R/FastPaxos.java in oracle/paxos_oracle.py (itself pinned to the reference's PaxosTests by
tests/test_paxos_oracle.py).  Host only: runs without a GPU."""
import random

import numpy as np
import pytest

from oracle import paxos_oracle as PX
from rapid_amd import _native as N
from rapid_amd import consensus as CS
from rapid_amd import scenarios as S
from rapid_amd import wire as W
from tests import proto_rapid as P
from tests.test_paxos_oracle import (COORDINATOR_RULE, COORDINATOR_RULE_SAME_RANK, MIXED, N_VALUES, P1, P2, check_rule,
                                     oracle_select)


def product_select(n, msgs):
    return CS.select_proposal(n, [(m.vrnd, m.vval) for m in msgs])


# ------------------------------------------------------------------------------------------- the coordinator rule
@pytest.mark.parametrize("case", COORDINATOR_RULE, ids=lambda c: f"N{c[0]}-{c[1]}-{c[2]}-{sorted(c[4])}")
def test_coordinator_rule_reference_table(case):
    """PaxosTests.coordinatorRuleTests (:200-296) on the product."""
    check_rule(case, False, product_select)


@pytest.mark.parametrize("case", COORDINATOR_RULE_SAME_RANK, ids=lambda c: f"N{c[0]}-{c[1]}-{c[2]}-{sorted(c[4])}")
def test_coordinator_rule_same_rank_reference_table(case):
    """PaxosTests.coordinatorRuleTestsSameRank (:307-397) on the product."""
    check_rule(case, True, product_select)


def test_coordinator_rule_matches_the_oracle_on_random_promises():
    rng = random.Random(7)
    values = [(), (3,), (3, 4), (4, 3), (9, 1, 2), (5,)]
    for _ in range(3000):
        n = rng.randrange(1, 30)
        m = rng.randrange(1, n + 3)
        msgs = []
        for _ in range(m):
            vrnd = rng.choice([(0, 0), (1, 1), (1, 1), (2, 5), (2, -7), (3, 0)])
            msgs.append(PX.Phase1bMessage(sender=0, configurationId=1, rnd=(2, 1), vrnd=vrnd, vval=rng.choice(values)))
        assert product_select(n, msgs) == oracle_select(n, msgs), (n, msgs)
    with pytest.raises(N.IllegalArgumentException):
        CS.select_proposal(5, [])


# ------------------------------------------------------------------------------- message-by-message, in lock step
def to_product(m):
    if m.kind == PX.FAST_ROUND_PHASE2B:
        return CS.Message(m.kind, m.sender, m.configurationId, endpoints=m.endpoints)
    if m.kind == PX.PHASE1A:
        return CS.Message(m.kind, m.sender, m.configurationId, rnd=m.rank)
    if m.kind == PX.PHASE1B:
        return CS.Message(m.kind, m.sender, m.configurationId, rnd=m.rnd, vrnd=m.vrnd, endpoints=m.vval)
    if m.kind == PX.PHASE2A:
        return CS.Message(m.kind, m.sender, m.configurationId, rnd=m.rnd, endpoints=m.vval)
    return CS.Message(m.kind, m.sender, m.configurationId, rnd=m.rnd, endpoints=m.endpoints)


class LockStep(PX.Network):
    """The oracle's network with one product object next to every oracle node.  Whatever is done to an oracle node is
    done to its twin, and the twin must queue exactly the messages the oracle node sent, in the same order."""

    def __post_init__(self):
        super().__post_init__()
        self.emitted = []
        hc = self.hash_codes or [1000 + 7 * i for i in range(self.n)]
        self.twins = {i: CS.FastPaxos(i, self.configurationId, self.membershipSize, hc[i]) for i in self.live}
        self.double_decisions = 0

    def _send(self, dest, m):
        self.emitted.append(CS.Message(**{**to_product(m).__dict__, "dest": dest}))
        super()._send(dest, m)

    def _broadcast(self, m):
        self.emitted.append(to_product(m))  # emitted even when the test network then drops it
        super()._broadcast(m)

    def _compare(self, i):
        got = self.twins[i].poll()
        assert got == self.emitted, (i, got, self.emitted)
        self.emitted = []
        want = self.nodes[i].decision
        assert self.twins[i].decision() == want

    def propose(self, i, value):
        self.nodes[i].propose(value)
        self.twins[i].propose(value)
        self._compare(i)

    def start_classic(self, i):
        self.nodes[i].startClassicPaxosRound()
        self.twins[i].startClassicPaxosRound()
        self._compare(i)

    def deliver_one(self, dest):
        m = self.queues[dest][0]
        try:
            super().deliver_one(dest)
        except AssertionError:
            # R/FastPaxos.java:79: a second decision (classic majority after a fast-round decision, or the reverse).
            # The Java only asserts; the product keeps the first decision.
            self.double_decisions += 1
        self.twins[dest].handleMessages(to_product(m))
        self._compare(dest)
        return m

    def some(self, steps):
        for _ in range(steps):
            p = self.pending()
            if not p:
                return
            self.deliver_one(self.rng.choice(p))


def agreement(net, expected):
    decisions = {i: net.twins[i].decision() for i in net.live}
    decided = {i: v for i, v in decisions.items() if v is not None}
    assert len(decided) == expected
    assert len(set(decided.values())) <= 1
    return next(iter(decided.values())) if decided else None


@pytest.mark.parametrize("n", N_VALUES)
def test_recovery_for_single_propose(n):
    """PaxosTests.testRecoveryForSinglePropose (:69-83)"""
    for seed in range(5):
        net = LockStep(n, seed=seed)
        proposer = net.rng.randrange(n)
        net.propose(proposer, (31234,))
        net.run()
        assert agreement(net, 0) is None
        net.start_classic(proposer)
        net.run()
        assert agreement(net, n) == (31234,)


@pytest.mark.parametrize("n", N_VALUES)
def test_recovery_from_fast_round_with_different_proposals(n):
    """PaxosTests.testRecoveryFromFastRoundWithDifferentProposals (:87-103): n concurrent coordinators"""
    for seed in range(8):
        net = LockStep(n, seed=seed, hash_codes=[((i * 2654435761) % 2**32) - 2**31 for i in range(n)])  # hashCode() is any int
        for i in range(n):
            net.propose(i, (i,))
        starters = list(range(n))
        net.rng.shuffle(starters)
        for i in starters:
            net.some(net.rng.randrange(0, 2 * n))
            net.start_classic(i)
        net.run()
        value = agreement(net, n)
        assert len(value) == 1 and 0 <= value[0] < n


@pytest.mark.parametrize("n", N_VALUES)
def test_classic_round_after_successful_fast_round(n):
    """PaxosTests.testClassicRoundAfterSuccessfulFastRound (:109-123)"""
    for seed in range(5):
        net = LockStep(n, seed=seed, drop={PX.FAST_ROUND_PHASE2B})
        for i in range(n):
            net.propose(i, (1234,))
        net.run()
        assert agreement(net, 0) is None
        for i in range(n):
            net.start_classic(i)
        net.run()
        assert agreement(net, n) == (1234,)


@pytest.mark.parametrize("case", MIXED, ids=lambda c: f"N{c[0]}-p2votes{c[1]}")
def test_classic_round_after_successful_fast_round_mixed_values(case):
    """PaxosTests.testClassicRoundAfterSuccessfulFastRoundMixedValues (:140-192)"""
    n, p2votes, choices = case
    for seed in range(10):
        net = LockStep(n, seed=seed, drop={PX.FAST_ROUND_PHASE2B})
        order = list(range(n))
        net.rng.shuffle(order)
        for idx, i in enumerate(order):
            net.propose(i, P1 if idx < n - p2votes else P2)
        net.run()
        for i in range(n):
            net.start_classic(i)
        net.run()
        assert agreement(net, n) in choices


def test_fast_round_decides_while_classic_rounds_run():
    """Nothing is dropped and timers fire early: some nodes decide through the fast quorum, some through a classic
    majority, possibly both -- one decision per node, the same everywhere."""
    both = 0
    for seed in range(40):
        n = 5 + seed % 7
        net = LockStep(n, seed=seed)
        for i in range(n):
            net.propose(i, (7, 8))
            net.some(net.rng.randrange(0, n))
            if net.rng.random() < 0.5:
                net.start_classic(i)
        net.run()
        assert agreement(net, n) == (7, 8)
        both += net.double_decisions
    assert both > 0  # the case the Java only asserts on did occur


def test_crashed_minority_and_lost_quorum():
    """Acceptors that never answer: with more than N/2 live nodes the classic round decides, without it nobody does."""
    for seed in range(10):
        n = 9
        live = list(range(5 + seed % 4))  # 5..8 live of 9
        net = LockStep(n, seed=seed, live=live)
        for i in live:
            net.propose(i, (i % 2,))
        net.run()
        net.start_classic(live[-1])
        net.run()
        assert agreement(net, len(live)) in ((0,), (1,))
    net = LockStep(9, seed=1, live=[0, 1, 2, 3])  # 4 of 9: no majority
    for i in net.live:
        net.propose(i, (5,))
    for i in net.live:
        net.start_classic(i)
    net.run()
    assert agreement(net, 0) is None


def test_messages_of_other_configurations_and_kinds():
    fp = CS.FastPaxos(0, 42, 5, rank_index=77)
    for kind in (PX.FAST_ROUND_PHASE2B, PX.PHASE1A, PX.PHASE1B, PX.PHASE2A, PX.PHASE2B):
        fp.handleMessages(CS.Message(kind, 1, 41, rnd=(2, 9), endpoints=(3,) if kind != PX.PHASE1A else ()))
    assert fp.poll() == [] and fp.decision() is None
    for kind in (0, 3, 4, 10):
        with pytest.raises(N.IllegalArgumentException):  # FastPaxos.handleMessages default case, :181
            fp.handleMessages(CS.Message(kind, 1, 42))
    # a decided node no longer starts a classic round (:191)
    for s in range(4):
        fp.handleMessages(CS.Message(PX.FAST_ROUND_PHASE2B, s, 42, endpoints=(3, 1)))
    assert fp.decision() == (3, 1)
    fp.startClassicPaxosRound()
    assert fp.poll() == []
    fp.startPhase1a(2)  # the Paxos object itself would (R/Paxos.java:98-111)
    assert fp.poll() == [CS.Message(PX.PHASE1A, 0, 42, rnd=(2, 77))]
    with pytest.raises(N.IllegalArgumentException):
        CS.FastPaxos(0, 1, 0)


def test_outbox_capacity():
    fp = CS.FastPaxos(2, 1, 3)
    big = tuple(range(500))
    fp.propose(big)
    assert fp.poll(cap=4) == [CS.Message(PX.FAST_ROUND_PHASE2B, 2, 1, endpoints=big)]


def test_fallback_delay():
    o = PX.FastPaxos(0, 1, 1000, None, None, None, 0)
    rng = random.Random(3)
    for _ in range(2000):
        u, base = rng.random(), rng.randrange(0, 5000)
        assert CS.fallback_delay_ms(1000, base, u) == o.getRandomDelayMs(u, base)
    with pytest.raises(N.IllegalArgumentException):
        CS.fallback_delay_ms(10, 0, 1.0)


# ------------------------------------------------------------------------------------- one round, whole population
def oracle_population_round(n, live, votes, arrival, coordinator_pos, seed):
    """Message by message on the oracle: fast round without quorum, then ONE classic round whose Phase1a reaches the
    acceptors in the order `arrival` -- which is the order their Phase1b reach the coordinator (one FIFO per node)."""
    net = PX.Network(n, seed=seed, live=live)
    for a, i in enumerate(live):
        if votes[a] is not None:
            net.nodes[i].propose(votes[a])
    net.run()
    if net.decisions:
        return None  # the fast round decided: not the case under test
    c = live[coordinator_pos]
    net.nodes[c].startClassicPaxosRound()
    for a in arrival:
        i = live[a]
        m = net.deliver_one(i)
        assert m.kind == PX.PHASE1A
    net.run()
    decided = {i: v for i, v in net.decisions}
    assert len(set(decided.values())) <= 1
    n_msgs = {k: sum(1 for _, m in net.log if m.kind == k) for k in (PX.PHASE1A, PX.PHASE1B, PX.PHASE2A, PX.PHASE2B)}
    return {"decided": len(decided) == len(live) and len(live) > 0, "value": next(iter(decided.values()), None),
            "partial": 0 < len(decided) < len(live), "delivered": n_msgs}


def test_population_round_matches_the_message_level_run():
    rng = random.Random(11)
    palette = [(1, 2), (2, 1), (3,), (4, 5, 6)]
    checked = rules = 0
    seen_rules = set()
    for trial in range(400):
        n = rng.randrange(3, 40)
        n_live = rng.randrange(1, n + 1)
        live = sorted(rng.sample(range(n), n_live))
        p_vote = rng.choice([0.2, 0.6, 0.9, 1.0])
        k = rng.randrange(1, len(palette) + 1)
        weights = [rng.random() ** 2 for _ in range(k)]
        votes = [rng.choices(palette[:k], weights)[0] if rng.random() < p_vote else None for _ in range(n_live)]
        arrival = list(range(n_live))
        rng.shuffle(arrival)
        want = oracle_population_round(n, live, votes, arrival, rng.randrange(n_live), trial)
        if want is None:
            continue
        key = {v: np.uint64(1000 + i) for i, v in enumerate(palette)}
        vote_key = [key[v] if v is not None else np.uint64(0) for v in votes]
        voted = [v is not None for v in votes]
        got = CS.classic_round_population(n, vote_key, voted, arrival)
        assert not want["partial"]
        assert got["decided"] == want["decided"], (n, live, votes, arrival)
        if got["decided"]:
            assert votes[got["chosen_acceptor"]] == want["value"]
            seen_rules.add(got["rule"])
            # all messages of the round were delivered to live nodes only: scale the broadcasts
            d = want["delivered"]
            assert d[PX.PHASE1B] == n_live and d[PX.PHASE1A] == n_live and d[PX.PHASE2A] == n_live and d[PX.PHASE2B] == n_live * n_live
            assert got["messages"] == n + n_live + n + n_live * n
            assert n // 2 < got["promises_used"] <= n_live
        else:
            assert got["chosen_acceptor"] == -1
        checked += 1
    assert checked > 200 and seen_rules == {1, 2, 3}


def test_population_rounds_with_concurrent_coordinators_and_loss_match_the_message_level_run():
    """rapid_classic_rounds_population against the oracle's network (PaxosTests.java:403-476 restated), SAME schedule: several
    coordinators start classic rounds (rounds 2, 3, ...; their ranks ordered by round, then by the hash that stands in for the node
    index, R/Paxos.java:98-111, 333-339) while earlier rounds' messages are still in flight, messages are lost one by one, and
    every step handles the oldest message queued at a chosen node.  Per node: decided or not, and what; per kind: messages handled."""
    rng = random.Random(23)
    palette = [(1, 2), (2, 1), (3,), (4, 5, 6)]
    saw_concurrent = saw_loss_undecided = saw_decided = 0
    for trial in range(250):
        n = rng.randrange(3, 24)
        n_live = rng.randrange(n // 2 + 1, n + 1)
        k = rng.randrange(1, len(palette) + 1)
        votes = [rng.choice(palette[:k]) if rng.random() < rng.choice([0.3, 0.8, 1.0]) else None for _ in range(n_live)]
        hc = rng.sample(range(5, 5000), n_live)
        net = PX.Network(n_live, seed=trial, membershipSize=n, hash_codes=hc, drop={PX.FAST_ROUND_PHASE2B})
        for a in range(n_live):
            if votes[a] is not None:
                net.nodes[a].propose(votes[a])  # (the fast round's own messages are swallowed: nobody decides in it)
        assert not net.decisions and not net.pending()
        n_coord = rng.randrange(1, 4)
        p_loss = rng.choice([0.0, 0.0, 0.05, 0.3])
        starts, schedule, drop = [], [], []
        to_start = sorted((rng.randrange(0, 6 * n_live), rng.randrange(n_live), 2 + j) for j in range(n_coord))
        step = 0
        while True:
            while to_start and to_start[0][0] <= step:
                _, a, rnd = to_start.pop(0)
                starts.append((step, a, rnd))
                net.nodes[a].paxos.startPhase1a(rnd)
            pend = net.pending()
            if not pend:
                if not to_start:
                    break
                # nothing in flight: the next coordinator starts now
                t, a, rnd = to_start.pop(0)
                to_start.insert(0, (step, a, rnd))
                continue
            d = rng.choice(pend)
            lose = rng.random() < p_loss
            schedule.append(d)
            drop.append(1 if lose else 0)
            if lose:
                net.queues[d].pop(0)
            else:
                net.deliver_one(d)
            step += 1
        decided = dict(net.decisions)
        key = {v: np.uint64(7000 + i) for i, v in enumerate(palette)}
        vote_key = [key[v] if v is not None else np.uint64(0) for v in votes]
        got = CS.classic_rounds_population(n, vote_key, [v is not None for v in votes], starts, rank_index=hc, schedule=schedule, drop=drop)
        assert got["agreed"] and got["undelivered"] == 0 and got["steps"] == len(schedule) and got["lost"] == sum(drop)
        assert got["decided_nodes"] == len(decided)
        for a in range(n_live):
            w = got["decided_vote_of"][a]
            assert (w >= 0) == (a in decided), (trial, a)
            if w >= 0:
                assert votes[w] == decided[a]
        handled = {kk: sum(1 for _, m in net.log if m.kind == kk) for kk in (PX.PHASE1A, PX.PHASE1B, PX.PHASE2A, PX.PHASE2B)}
        assert got["delivered"] == [handled[PX.PHASE1A], handled[PX.PHASE1B], handled[PX.PHASE2A], handled[PX.PHASE2B]], trial
        assert len(set(decided.values())) <= 1  # (safety, on the oracle's side too)
        saw_concurrent += n_coord > 1 and len(starts) > 1 and starts[1][0] < len(schedule)
        saw_decided += bool(decided)
        saw_loss_undecided += p_loss > 0 and not decided
    assert saw_concurrent > 40 and saw_decided > 80 and saw_loss_undecided > 3
    # the seeded form: no schedule, a loss probability -- safety holds, and without loss every live node of a majority decides
    votes = [palette[i % 3] for i in range(60)]
    vote_key = [np.uint64(hash(v) & 0xFFFF) for v in votes]
    for seed in range(20):
        got = CS.classic_rounds_population(100, vote_key, [True] * 60, [(0, 5, 2), (40, 17, 3), (300, 33, 4)], seed=seed, loss=0.0)
        assert got["agreed"] and got["decided_nodes"] == 60 and got["undelivered"] == 0 and got["lost"] == 0
        got = CS.classic_rounds_population(100, vote_key, [True] * 60, [(0, 5, 2), (40, 17, 3), (300, 33, 4)], seed=seed, loss=0.2)
        assert got["agreed"] and got["lost"] > 0
    with pytest.raises(N.IllegalArgumentException):
        CS.classic_rounds_population(10, [1, 2], [True, True], [(3, 0, 2)], schedule=[1], drop=[0])  # nothing is queued at node 1 before step 3
    with pytest.raises(N.IllegalArgumentException):
        CS.classic_rounds_population(10, [1, 2], [True, True], [(0, 5, 2)])  # no such acceptor


def test_population_round_from_tally_results():
    """The glue ClusterSimulation.classic_round uses: receivers without a proposal (emit_batch -1) did not vote."""
    n = 12
    emit = np.array([3, 3, -1, 5, 2, 6, 4, 4, 4, -1], dtype=np.int32)  # 10 live receivers of 12 members
    fp = np.array([11, 11, 0, 22, 11, 11, 22, 22, 11, 0], dtype=np.uint64)
    res, winner = CS.classic_round_from_results(n, emit, fp)
    # index order: the first 7 promises (N/2 + 1) hold 11, 11, 22, 11, 11, 22 -> 11 is seen more than N/4 = 3 times at index 5
    assert res["decided"] and res["promises_used"] == 7 and res["rule"] == 2 and winner == 5
    res, winner = CS.classic_round_from_results(n, emit, fp, arrival=[2, 9, 3, 6, 7, 0, 1, 4, 5, 8])
    assert res["rule"] == 3 and winner == 3  # 22, 22, 22, 11, 11 among the first seven: nothing above N/4, the first value
    res, winner = CS.classic_round_from_results(n, emit[:6], fp[:6])
    assert not res["decided"] and winner is None  # 6 of 12 is not a majority
    with pytest.raises(N.IllegalArgumentException):
        CS.classic_round_population(n, fp, emit >= 0, arrival=[0] * 10)


# ------------------------------------------------------------------------------------------------------- wire forms
def test_wire_forms_match_the_protobuf_runtime():
    pop = S.Population.make(40)
    emap = W.EndpointMap(pop.hostnames + [b""], list(pop.ports) + [0])  # node 40: empty hostname, port 0 (all defaults)

    def ep(i):
        return P.Endpoint(hostname=(pop.hostnames + [b""])[i], port=int((list(pop.ports) + [0])[i]))

    def rank(r):
        return P.Rank(round=r[0], nodeIndex=r[1])

    def reference_bytes(m):
        eps = [ep(e) for e in m.endpoints]
        if m.kind == PX.FAST_ROUND_PHASE2B:
            return P.RapidRequest(fastRoundPhase2bMessage=P.FastRoundPhase2bMessage(sender=ep(m.sender), configurationId=m.configurationId,
                                                                                    endpoints=eps)).SerializeToString()
        if m.kind == PX.PHASE1A:
            return P.RapidRequest(phase1aMessage=P.Phase1aMessage(sender=ep(m.sender), configurationId=m.configurationId,
                                                                  rank=rank(m.rnd))).SerializeToString()
        if m.kind == PX.PHASE1B:
            return P.RapidRequest(phase1bMessage=P.Phase1bMessage(sender=ep(m.sender), configurationId=m.configurationId, rnd=rank(m.rnd),
                                                                  vrnd=rank(m.vrnd), vval=eps)).SerializeToString()
        if m.kind == PX.PHASE2A:
            return P.RapidRequest(phase2aMessage=P.Phase2aMessage(sender=ep(m.sender), configurationId=m.configurationId, rnd=rank(m.rnd),
                                                                  vval=eps)).SerializeToString()
        return P.RapidRequest(phase2bMessage=P.Phase2bMessage(sender=ep(m.sender), configurationId=m.configurationId, rnd=rank(m.rnd),
                                                              endpoints=eps)).SerializeToString()

    rng = random.Random(5)
    for trial in range(400):
        kind = rng.choice([5, 6, 7, 8, 9])
        eps = () if kind == PX.PHASE1A else tuple(rng.choice([rng.randrange(41), 40]) for _ in range(rng.choice([0, 1, 3, 17])))
        rnd = (0, 0) if kind == 5 else (rng.choice([0, 1, 2, 300]), rng.choice([0, 1, -1, 2**31 - 1, -2**31]))
        vrnd = (rng.choice([0, 1, 2]), rng.choice([0, 1, -99])) if kind == PX.PHASE1B else (0, 0)
        m = CS.Message(kind, rng.randrange(41), rng.choice([0, 1, -1, -2**63, 2**63 - 1, 123456789]), rnd, vrnd, eps)
        want = reference_bytes(m)
        assert CS.encode_request(emap, m) == want
        k, payload = W.decode_request(want)
        assert k == kind
        assert CS.decode_message(emap, k, payload) == m
    # unknown fields are skipped, unknown endpoints and malformed bytes are reported
    m = CS.Message(PX.PHASE2B, 1, 9, (2, 5), (0, 0), (3, 4))
    _, payload = W.decode_request(reference_bytes(m))
    assert CS.decode_message(emap, PX.PHASE2B, payload + b"\x78\x05") == m
    with pytest.raises(N.NodeNotInRingException):
        stranger = P.Phase2bMessage(sender=P.Endpoint(hostname=b"who", port=1), configurationId=9)
        CS.decode_message(emap, PX.PHASE2B, stranger.SerializeToString())
    with pytest.raises(N.IllegalArgumentException):
        CS.decode_message(emap, PX.PHASE2B, payload[:-1])
    with pytest.raises(N.IllegalArgumentException):
        CS.decode_message(emap, 4, payload)
    with pytest.raises(N.NodeNotInRingException):
        CS.encode_request(emap, CS.Message(PX.PHASE2A, 1, 9, (2, 5), (0, 0), (41,)))


def test_a_round_over_the_wire():
    """Five product nodes that only exchange serialized RapidRequests."""
    n = 5
    pop = S.Population.make(n)
    emap = W.EndpointMap(pop.hostnames, pop.ports)
    nodes = [CS.FastPaxos(i, 77, n) for i in range(n)]
    queues = [[] for _ in range(n)]
    rng = random.Random(2)

    def flush(i):
        for m in nodes[i].poll():
            data = CS.encode_request(emap, m)
            for d in (range(n) if m.dest == CS.BROADCAST else [m.dest]):
                queues[d].append(data)

    for i in range(n):
        nodes[i].propose((i % 2, 4))
        flush(i)
    nodes[3].startClassicPaxosRound()
    flush(3)
    while any(queues):
        i = rng.choice([q for q in range(n) if queues[q]])
        kind, payload = W.decode_request(queues[i].pop(0))
        nodes[i].handleMessages(CS.decode_message(emap, kind, payload))
        flush(i)
    decisions = {nodes[i].decision() for i in range(n)}
    assert len(decisions) == 1 and decisions.pop() in ((0, 4), (1, 4))
