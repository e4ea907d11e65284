"""Alert producer model: the closed forms of rapid_amd/timeline.py against the literal event-by-event
restatement of the reference's timers in oracle/timeline_oracle.py (PingPongFailureDetector, AlertBatcher,
edgeFailureNotification), then a whole round on the time line: detection -> batches -> per-receiver arrival order ->
cut -> fast-round decision.  Host only."""
import numpy as np
import pytest

from oracle import paxos_oracle as PX
from oracle import pyoracle as O
from oracle import timeline_oracle as TO
from rapid_amd import scenarios as S
from rapid_amd import timeline as T
from tests.helpers import oracle_view


def oracle_batches(subj, crash, start, t_end, model):
    sim = TO.ProducerSimulation(subj, [None if c == T.NEVER else int(c) for c in crash], [int(x) for x in start], t_end,
                                fd_interval_ms=model.fd_interval_ms, batching_window_ms=model.batching_window_ms,
                                probe_fail_ms=model.probe_fail_ms)
    return sim.run()


def as_tuples(bs, send):
    out = []
    for b in range(bs.n_batches):
        recs = bs.recs[bs.off[b]:bs.off[b + 1]]
        msgs = [(int(r["src"]), int(r["dst"]), int(r["status"]), tuple(k for k in range(16) if (int(r["ring_mask"]) >> k) & 1))
                for r in recs]
        assert recs["flags"][-1] == S.FLAG_LAST_IN_BATCH and not recs["flags"][:-1].any()
        out.append((int(send[b]), int(bs.sender[b]), msgs))
    return out


@pytest.mark.parametrize("seed", range(40))
def test_batches_equal_the_event_simulation(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(8, 60))
    K = int(rng.integers(3, 11))
    if seed % 3 == 0:  # the real topology of a view
        pop = S.Population.make(n)
        _, view = oracle_view(pop, K)
        subj = view.tables(n)[1]
    else:  # arbitrary tables, duplicates within a row likely
        subj = rng.integers(0, n, size=(n, K)).astype(np.int32)
    model = T.ProducerModel(fd_interval_ms=int(rng.choice([1000, 250, 70])), failure_threshold=10,
                            batching_window_ms=int(rng.choice([100, 30, 400])), probe_fail_ms=int(rng.choice([0, 1, 999, 1000, 1001, 2500])))
    if model.fd_interval_ms == 70:
        model.probe_fail_ms = int(rng.choice([0, 1, 69, 70, 71, 200]))
    crash = np.full(n, T.NEVER, dtype=np.int64)
    f = int(rng.integers(1, max(2, n // 3)))
    crash[rng.choice(n, f, replace=False)] = rng.integers(0, 3 * model.fd_interval_ms, f)
    if seed % 2:  # some observers die later, in the middle of detecting or batching
        late = rng.choice(n, max(1, n // 6), replace=False)
        horizon = (12 + model.probe_fail_ms // model.fd_interval_ms) * model.fd_interval_ms
        crash[late] = np.minimum(crash[late], rng.integers(0, 2 * horizon, len(late)))
    start = rng.integers(0, model.fd_interval_ms, n).astype(np.int64)
    t_end = int(crash[crash != T.NEVER].max() + (14 + 2 * (model.probe_fail_ms // model.fd_interval_ms)) * model.fd_interval_ms
                + 10 * model.batching_window_ms)
    want = oracle_batches(subj, crash, start, t_end, model)
    bs, send = T.batches(model, subj, crash, start, cfg_id=7)
    assert as_tuples(bs, send) == want
    assert len(want) > 0 and (bs.recs["cfg_id"] == 7).all()


def test_default_timing_of_one_crash():
    """Defaults of the reference: a crash is reported 10 detector ticks after the first failed probe and leaves the
    observer between one and two batching windows later."""
    n, K = 30, 10
    pop = S.Population.make(n)
    _, view = oracle_view(pop, K)
    subj = view.tables(n)[1]
    model = T.ProducerModel()
    crash = np.full(n, T.NEVER, dtype=np.int64)
    crash[4] = 12_345
    start = (np.arange(n) * 37) % 1000
    o, k, s, t = T.notifications(model, subj, crash, start)
    assert set(o.tolist()) == set(np.nonzero((subj == 4).any(axis=1))[0].tolist()) - {4} or 4 in o
    first_failed = start[o] + -(-(12_345 - start[o]) // 1000) * 1000
    assert np.array_equal(t, first_failed + 10_000)
    bs, send = T.batches(model, subj, crash, start, 1)
    t_of = {int(a): int(b) for a, b in zip(o, t)}
    for b in range(bs.n_batches):
        lag = int(send[b]) - t_of[int(bs.sender[b])]
        assert 100 < lag <= 200 and (int(send[b]) - int(start[bs.sender[b]])) % 100 == 0


def test_duplicate_rings_send_duplicate_alerts():
    """One detector per ENTRY of getSubjectsOf (R/MembershipService.java:697-706): an observer that watches s on two
    rings reports s twice, each alert carrying both ring numbers."""
    subj = np.array([[1, 1, 2], [0, 2, 2], [0, 1, 0]], dtype=np.int32)
    crash = np.array([T.NEVER, 500, T.NEVER], dtype=np.int64)
    bs, send = T.batches(T.ProducerModel(), subj, crash, np.zeros(3, dtype=np.int64), 9)
    got = as_tuples(bs, send)
    assert got == [(11_200, 0, [(0, 1, 1, (0, 1)), (0, 1, 1, (0, 1))]), (11_200, 2, [(2, 1, 1, (1,))])]
    assert got == oracle_batches(subj, crash, np.zeros(3, dtype=np.int64), 20_000, T.ProducerModel())


def test_a_round_on_the_time_line():
    """Crashes at different moments -> detection -> batches -> each receiver's arrival order -> the cut detector ->
    fast round.  The cut is the crashed set at every receiver; the time line orders what happened."""
    n, K, H, L = 300, 10, 9, 4
    pop = S.Population.make(n)
    reg, view = oracle_view(pop, K)
    obs, subj, member = view.tables(n)
    cfg = view.getCurrentConfigurationId()
    rng = np.random.default_rng(3)
    faulty = np.sort(rng.choice(n, 12, replace=False))
    crash = np.full(n, T.NEVER, dtype=np.int64)
    crash[faulty] = rng.integers(5_000, 5_900, len(faulty))  # within one detector interval of each other
    start = rng.integers(0, 1000, n)
    model, lat = T.ProducerModel(), T.LatencyModel(base_ms=1, jitter_ms=4, seed=5)
    bs, send = T.batches(model, subj, crash, start, cfg)
    receivers = np.flatnonzero(crash == T.NEVER).astype(np.int32)
    records, rec_off, arrival, arr_off = T.deliver_timed(bs, send, receivers, n, lat)
    # every receiver sees every batch, in non-decreasing arrival order, each batch intact
    assert np.all(np.diff(rec_off) == len(bs.recs)) and np.all(np.diff(arr_off) == bs.n_batches)
    for i in (0, 7, len(receivers) - 1):
        arr = arrival[arr_off[i]:arr_off[i + 1]]
        assert np.all(np.diff(arr) >= 0)
        want = np.sort(send + lat.delay(bs.sender, np.full(bs.n_batches, receivers[i]), n))
        assert np.array_equal(arr, want)
        ends = np.flatnonzero(records[rec_off[i]:rec_off[i + 1]]["flags"] & 1)
        assert len(ends) == bs.n_batches
    fe, fn, fo, fp = O.fast_sim_run(n, K, H, L, cfg, obs, subj, member, records, rec_off, nthreads=4)
    for i in range(len(receivers)):
        assert fe[i] >= 0 and sorted(fp[fo[i]:fo[i + 1]].tolist()) == faulty.tolist()
    t_prop = T.proposal_times(fe, arrival, arr_off)
    # detection needs ten failed probes: nothing can be proposed earlier than 10 s after the first crash
    assert t_prop.min() > crash[faulty].min() + 10_000 and t_prop.max() < crash[faulty].max() + 11_000 + 200 + 5
    t_dec = T.fast_round_decision_times(t_prop, receivers, np.ones(len(receivers), dtype=np.uint64), n, lat)
    quorum = n - (n - 1) // 4
    assert np.all(t_dec >= np.sort(t_prop)[quorum - 1] + 1) and np.all(t_dec <= t_prop.max() + 5)
    # the same decision times from the restated FastPaxos, vote by vote in arrival order, at a few nodes
    for i in (0, 11, 200):
        r = int(receivers[i])
        arr = t_prop + lat.delay(receivers, np.full(len(receivers), r), n)
        fpx = PX.FastPaxos(r, cfg, n, None, None, lambda v: None, 0)
        when = None
        for j in np.lexsort((receivers, arr)):
            fpx.handleFastRoundProposal(PX.FastRoundPhase2bMessage(int(receivers[j]), cfg, tuple(faulty.tolist())))
            if fpx.decided:
                when = int(arr[j])
                break
        assert when == int(t_dec[i])
    # too few voters: no decision
    few = np.where(np.arange(len(receivers)) < quorum - 1, t_prop, T.NEVER)
    assert np.all(T.fast_round_decision_times(few, receivers, np.ones(len(receivers), dtype=np.uint64), n, lat) == T.NEVER)


def test_timed_streams_through_the_emulated_tally_kernel():
    """The time-ordered streams (with the reference's duplicate alerts for subjects watched on several rings) through
    rapid_amd/csrc/tally_kernel.h under the SIMT emulator: same announcing batch and cut as the faithful oracle."""
    from tests.test_kernel_emulated import _check
    n, K, H, L = 40, 10, 8, 3
    pop = S.Population.make(n)
    reg, view = oracle_view(pop, K)
    obs, subj, member = view.tables(n)
    cfg = view.getCurrentConfigurationId()
    crash = np.full(n, T.NEVER, dtype=np.int64)
    crash[[3, 17, 18, 30]] = [100, 150, 1200, 2050]  # three detection waves, one second apart
    start = (np.arange(n) * 61) % 1000
    bs, send = T.batches(T.ProducerModel(), subj, crash, start, cfg)
    pairs = np.stack([bs.recs["src"], bs.recs["dst"]], axis=1)
    assert len(np.unique(pairs, axis=0)) < len(pairs)  # small rings: some observer watches a subject on two rings
    receivers = np.flatnonzero(crash == T.NEVER).astype(np.int32)[::3]
    records, rec_off, arrival, arr_off = T.deliver_timed(bs, send, receivers, n, T.LatencyModel())
    _check(records, rec_off, n, K, H, L, cfg, obs, subj, member)


def test_classic_round_on_the_time_line():
    """No fast quorum -> the first recovery timer -> one classic round, with the message delays of the latency model:
    coordinator, Phase1b arrival order, decided value and every node's decision time equal the restated Paxos.java run
    message by message on the same time line."""
    rng = np.random.default_rng(8)
    compared = 0
    for trial in range(80):
        n = int(rng.integers(9, 60))
        n_live = int(rng.integers(n // 2 + 1, n + 1))
        receivers = np.sort(rng.choice(n, n_live, replace=False)).astype(np.int32)
        lat = T.LatencyModel(base_ms=int(rng.integers(1, 4)), jitter_ms=int(rng.integers(0, 9)), seed=trial)
        values = [(1, 2), (3,), (2, 1)]
        which = rng.integers(0, len(values), n_live)
        proposal_ms = np.where(rng.random(n_live) < 0.85, rng.integers(10_000, 10_400, n_live), T.NEVER)
        if (proposal_ms != T.NEVER).sum() == 0:
            proposal_ms[0] = 10_000
        vote_key = (which + 100).astype(np.uint64)
        u = rng.random(n_live)
        base = 1000
        got = T.classic_round_times(proposal_ms, receivers, vote_key, n, lat, base, u)
        # the same on the oracle: proposals at their times (fast-round votes travel too), every proposer's timer armed
        delay = lambda s, r: int(lat.delay(np.array([s]), np.array([r]), n)[0])  # noqa: E731
        net = PX.TimedNetwork(n, [int(r) for r in receivers], 7, delay)
        fire = {}
        for i in range(n_live):
            if proposal_ms[i] != T.NEVER:
                node = net.nodes[int(receivers[i])]
                net.at(int(proposal_ms[i]), (lambda node=node, v=values[which[i]]: node.propose(v)))
                fire[i] = int(proposal_ms[i]) + node.getRandomDelayMs(float(u[i]), base)
        first = min(fire, key=lambda i: (fire[i], int(receivers[i])))
        assert first == got["coordinator"] and fire[first] == got["start_ms"]
        if sorted(fire.values())[1:2] and sorted(fire.values())[1] < got["start_ms"] + 200:
            continue  # a second timer would fire before the round is over: not the single-coordinator case
        net.at(fire[first], net.nodes[int(receivers[first])].startClassicPaxosRound)
        net.run()
        fast_decided = any(t < fire[first] for t in net.decision_ms.values())
        if fast_decided:
            continue  # the fast round had a quorum after all: nothing to recover
        if not got["result"]["decided"]:
            assert not net.decisions
            continue
        assert len(net.decisions) == n_live and len(set(net.decisions.values())) == 1
        assert values[which[got["result"]["chosen_acceptor"]]] == next(iter(net.decisions.values()))
        want = np.array([net.decision_ms[int(r)] for r in receivers])
        assert np.array_equal(got["decision_ms"], want), (trial, got["decision_ms"][:5], want[:5])
        compared += 1
    assert compared >= 40
