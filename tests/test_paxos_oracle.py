"""Pins oracle/paxos_oracle.py (the restatement of R/Paxos.java + R/FastPaxos.java) to the cases of the reference's own
test, rapid/src/test/java/com/vrg/rapid/PaxosTests.java: the two coordinator-rule tables (:251-296 and :356-397 there: 19 + 17 rows,
100 random quorums each), the mixed-value recovery table (:174-192) and the three recovery scenarios (:69-123) for
N in {5, 6, 10, 11, 20}.  Threads and sleeps of the Java test become seeded interleavings."""
import random

import pytest

from oracle import paxos_oracle as PX

P1 = (5891, 5821)  # "127.0.0.1:5891", "127.0.0.1:5821"
P2 = (5821, 5872)
NOISE = (1, 2)
PROPOSALS = [P1, P2, NOISE]
SWAPPED = [P2, P1, NOISE]

# PaxosTests.java:251-296 -- (N, p1N, p2N, proposals, valid proposal indices)
COORDINATOR_RULE = [
    (6, 4, 2, PROPOSALS, {0}), (6, 5, 1, PROPOSALS, {0}), (6, 6, 0, PROPOSALS, {0}), (9, 6, 3, PROPOSALS, {0, 1}),
    (9, 7, 2, PROPOSALS, {0}), (9, 8, 1, PROPOSALS, {0}),
    (6, 1, 5, PROPOSALS, {0, 1}),
    (6, 2, 4, PROPOSALS, {0, 1}),
    (6, 3, 3, PROPOSALS, {0}), (6, 3, 3, SWAPPED, {0}),
    (6, 4, 1, PROPOSALS, {0}), (6, 5, 1, PROPOSALS, {0}), (9, 6, 1, PROPOSALS, {0, 1, 2}), (9, 7, 1, PROPOSALS, {0}),
    (9, 8, 1, PROPOSALS, {0}),
    (6, 1, 2, PROPOSALS, {0, 1, 2}),
    (6, 2, 1, PROPOSALS, {0, 1, 2}),
    (6, 3, 0, PROPOSALS, {0}), (6, 3, 0, SWAPPED, {0}),
]
# PaxosTests.java:356-397
COORDINATOR_RULE_SAME_RANK = [
    (6, 4, 2, PROPOSALS, {0, 1}), (6, 5, 1, PROPOSALS, {0}), (6, 6, 0, PROPOSALS, {0}), (9, 6, 3, PROPOSALS, {0, 1}),
    (9, 7, 2, PROPOSALS, {0}), (9, 8, 1, PROPOSALS, {0}),
    (6, 3, 3, PROPOSALS, {0, 1}), (6, 3, 3, SWAPPED, {0, 1}),
    (6, 4, 1, PROPOSALS, {0, 1}), (6, 5, 0, PROPOSALS, {0}), (9, 6, 1, PROPOSALS, {0, 1, 2}), (9, 7, 1, PROPOSALS, {0}),
    (9, 8, 1, PROPOSALS, {0}),
    (6, 1, 2, PROPOSALS, {0, 1, 2}),
    (6, 2, 1, PROPOSALS, {0, 1, 2}),
    (6, 3, 0, PROPOSALS, {0}), (6, 3, 0, SWAPPED, {0}),
]
INT_MAX = 2**31 - 1


def rule_messages(N, p1N, p2N, proposals, same_rank):
    """The Phase1b messages of PaxosTests.coordinatorRuleTests (:200-249) / ...SameRank (:307-354)."""
    msgs = []
    for _ in range(p1N):  # highest rank, proposals[0]
        msgs.append(PX.Phase1bMessage(sender=0, configurationId=1, rnd=(0, 0), vrnd=(1, 1), vval=proposals[0]))
    for _ in range(p2N):
        vrnd = (1, 1) if same_rank else (0, INT_MAX)
        msgs.append(PX.Phase1bMessage(sender=0, configurationId=1, rnd=(0, 0), vrnd=vrnd, vval=proposals[1]))
    for i in range(p1N + p2N, N):  # lower ranks
        msgs.append(PX.Phase1bMessage(sender=0, configurationId=1, rnd=(0, 0), vrnd=(0, i), vval=proposals[2]))
    return msgs


def check_rule(case, same_rank, select):
    N, p1N, p2N, proposals, valid = case
    valid_values = [proposals[i] for i in valid]
    rng = random.Random(N * 1000 + p1N * 10 + p2N)
    seen = set()
    for _ in range(100):
        msgs = rule_messages(N, p1N, p2N, proposals, same_rank)
        rng.shuffle(msgs)
        chosen = select(N, msgs[: N // 2 + 1])
        assert chosen in valid_values, (case, chosen)
        seen.add(chosen)
    return seen


def oracle_select(N, msgs):
    paxos = PX.Paxos(1234, 1, N, lambda d, m: None, lambda m: None, lambda v: None, 99)
    return paxos.selectProposalUsingCoordinatorRule(msgs)


@pytest.mark.parametrize("case", COORDINATOR_RULE, ids=lambda c: f"N{c[0]}-{c[1]}-{c[2]}-{sorted(c[4])}")
def test_coordinator_rule(case):
    check_rule(case, False, oracle_select)


@pytest.mark.parametrize("case", COORDINATOR_RULE_SAME_RANK, ids=lambda c: f"N{c[0]}-{c[1]}-{c[2]}-{sorted(c[4])}")
def test_coordinator_rule_same_rank(case):
    check_rule(case, True, oracle_select)


def test_coordinator_rule_rejects_an_empty_list():
    with pytest.raises(ValueError):
        oracle_select(5, [])


N_VALUES = [5, 6, 10, 11, 20]  # PaxosTests.java:125-134


def assert_agreement(net, expected_size):
    """waitAndVerifyAgreement (:513-530): that many decisions, all equal."""
    assert len(net.decisions) == expected_size
    assert len({v for _, v in net.decisions}) <= 1
    assert len({i for i, _ in net.decisions}) == expected_size  # each node decides once


@pytest.mark.parametrize("n", N_VALUES)
def test_recovery_for_single_propose(n):
    """:69-83 -- one node proposes and, 50 ms later, starts a classic round alone."""
    for seed in range(10):
        net = PX.Network(n, seed=seed)
        proposer = net.rng.randrange(n)
        net.nodes[proposer].propose((31234,))
        net.run()
        assert_agreement(net, 0)
        net.nodes[proposer].startClassicPaxosRound()
        net.run()
        assert_agreement(net, n)
        assert net.decisions[0][1] == (31234,)


@pytest.mark.parametrize("n", N_VALUES)
def test_recovery_from_fast_round_with_different_proposals(n):
    """:87-103 -- every node proposes itself; all classic rounds fire (here at random points of the run)."""
    for seed in range(20):
        net = PX.Network(n, seed=seed)
        for i in range(n):
            net.nodes[i].propose((i,))
        starters = list(range(n))
        net.rng.shuffle(starters)
        for i in starters:  # the timers fire while messages are still in flight
            for _ in range(net.rng.randrange(0, 2 * n)):
                p = net.pending()
                if p:
                    net.deliver_one(net.rng.choice(p))
            net.nodes[i].startClassicPaxosRound()
        net.run()
        assert_agreement(net, n)
        value = net.decisions[0][1]
        assert len(value) == 1 and 0 <= value[0] < n


@pytest.mark.parametrize("n", N_VALUES)
def test_classic_round_after_successful_fast_round(n):
    """:109-123 -- every node voted for the same value but no vote was delivered; the classic rounds recover it."""
    for seed in range(10):
        net = PX.Network(n, seed=seed, drop={PX.FAST_ROUND_PHASE2B})
        for i in range(n):
            net.nodes[i].propose((1234,))
        net.run()
        assert_agreement(net, 0)
        for i in range(n):
            net.nodes[i].startClassicPaxosRound()
        net.run()
        assert_agreement(net, n)
        assert net.decisions[0][1] == (1234,)


# PaxosTests.java:174-192 -- (N, p2 votes, decision choices); p1 gets the other N - p2 votes
MIXED = [(6, 5, [P2]), (6, 1, [P1]), (6, 4, [P1, P2]), (6, 2, [P1, P2]), (5, 4, [P2]), (5, 1, [P1]), (10, 4, [P1, P2]),
         (10, 1, [P1, P2])]


@pytest.mark.parametrize("case", MIXED, ids=lambda c: f"N{c[0]}-p2votes{c[1]}")
def test_classic_round_after_successful_fast_round_mixed_values(case):
    """:140-172.  (The Java's last row lists both values as admissible although only p1 can win; kept as written.)"""
    n, p2votes, choices = case
    for seed in range(25):
        net = PX.Network(n, seed=seed, drop={PX.FAST_ROUND_PHASE2B})
        order = list(range(n))
        net.rng.shuffle(order)  # ConcurrentHashMap iteration order
        for idx, i in enumerate(order):
            net.nodes[i].propose(P1 if idx < n - p2votes else P2)
        net.run()
        assert_agreement(net, 0)
        for i in range(n):
            net.nodes[i].startClassicPaxosRound()
        net.run()
        assert_agreement(net, n)
        assert net.decisions[0][1] in choices


def test_fallback_delay_formula():
    """FastPaxos.java:201-204 with jitterRate = 1/N (:76)."""
    fp = PX.FastPaxos(0, 1, 100, None, None, None, 0)
    assert fp.getRandomDelayMs(0.0, 1000) == 1000
    assert fp.getRandomDelayMs(1 - 2.718281828459045 ** -1, 1000) in (1000 + 99999, 1000 + 100000)  # -ln(1/e) * 1000 * N
    assert fp.getRandomDelayMs(0.5, 0) == int(1000 * 100 * 0.6931471805599453)
