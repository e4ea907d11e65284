"""Parity tests proper (-m gpu): the HIP path through the C ABI (rapid_amd.engine -> librapid_mi355x.so) against the
CPU oracle on the same seeded inputs.  Bit-exact everywhere: ring keys, ring orders, observer/subject tables,
configuration ids, per-receiver announcing batch / getNumProposals() / proposal list, decided cut, post-cut
configuration id.  Nothing here reads /root/reference."""
import numpy as np
import pytest

from oracle import pyoracle as O
from rapid_amd import scenarios as S
from tests.helpers import oracle_view, proposal_fingerprints, props_list, random_stream

pytestmark = pytest.mark.gpu

K10 = 10


@pytest.fixture(scope="module")
def E():
    from rapid_amd import engine
    if engine.device_count() < 1:
        pytest.fail("no gfx950 device visible: the product has no CPU fallback")
    return engine


@pytest.fixture()
def test_build():
    """The tests that use rapid_debug_* (reading delivered records back, one GPU standing in for several ranks) run on
    librapid_mi355x_test.so -- the same sources plus those entry points; every other test runs on the product library."""
    from rapid_amd import _native as N
    prev = N.use_test_build()
    yield
    N.use_library(prev)


def make_engine(E, pop, K, H, L, members=None, **kw):
    eng = E.Engine(n_max=pop.n, K=K, H=H, L=L, **kw)
    view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo, members=members)
    return eng, view


# ------------------------------------------------------------------------------------------------ view
@pytest.mark.parametrize("n,K", [(1, 10), (2, 10), (3, 10), (50, 3), (1000, 10), (4099, 7)])
def test_view_matches_oracle(E, n, K):
    pop = S.Population.make(n)
    eng, view = make_engine(E, pop, K, K, 1)
    reg, oview = oracle_view(pop, K)
    assert view.getMembershipSize() == n
    for k in range(K):
        for node in range(0, n, max(1, n // 50)):
            assert view.ringKey(k, node) == oview.ringKey(k, node)
        assert np.array_equal(view.getRing(k), oview.getRing(k))
    obs, subj, member = view.tables()
    oobs, osubj, omember = oview.tables(n)
    assert np.array_equal(obs, oobs) and np.array_equal(subj, osubj) and np.array_equal(member, omember)
    assert view.getCurrentConfigurationId() == oview.getCurrentConfigurationId()
    for node in range(0, n, max(1, n // 20)):
        assert view.getObserversOf(node) == oview.getObserversOf(node)
        assert view.getSubjectsOf(node) == oview.getSubjectsOf(node)
        assert view.getExpectedObserversOf(node) == oview.getExpectedObserversOf(node)
        for o in set(oview.getObserversOf(node)):
            assert view.getRingNumbers(o, node) == oview.getRingNumbers(o, node)


def test_view_long_hostnames_and_ports(E):
    """XXH64 stripe loop (>= 32 bytes), all tail lengths, negative / large ports."""
    rng = np.random.default_rng(11)
    n = 200
    pop = S.Population.make(n)
    pop.hostnames = [bytes(rng.integers(33, 127, size=int(l)).astype(np.uint8)) + b"-%d" % i
                     for i, l in enumerate(rng.integers(0, 120, size=n))]
    pop.ports = rng.integers(-2**31, 2**31 - 1, size=n).astype(np.int32)
    eng, view = make_engine(E, pop, 10, 9, 4)
    reg, oview = oracle_view(pop, 10)
    for k in range(10):
        assert np.array_equal(view.getRing(k), oview.getRing(k))
    assert view.getCurrentConfigurationId() == oview.getCurrentConfigurationId()


def test_view_add_delete_sequences_and_errors(E):
    """MembershipViewTest.java: ringReAdditions, ringDeletionsOnly, nodeUniqueId*, monitoringRelationship*."""
    n, K = 64, 10
    pop = S.Population.make(n)
    pop.id_hi[40], pop.id_lo[40] = pop.id_hi[3], pop.id_lo[3]  # node 40 reuses node 3's identifier
    members = list(range(0, 32))
    eng, view = make_engine(E, pop, K, 8, 2, members=members)
    reg, oview = oracle_view(pop, K, members)
    nid = lambda i: (int(pop.id_hi[i]), int(pop.id_lo[i]))

    def same():
        assert view.getMembershipSize() == oview.getMembershipSize()
        assert view.getCurrentConfigurationId() == oview.getCurrentConfigurationId()
        for k in range(K):
            assert np.array_equal(view.getRing(k), oview.getRing(k))
        for node in range(n):
            assert view.isHostPresent(node) == oview.isHostPresent(node)
            assert view.getExpectedObserversOf(node) == oview.getExpectedObserversOf(node)
            if oview.isHostPresent(node):
                assert view.getObserversOf(node) == oview.computeObserversOf(node)
                assert view.getSubjectsOf(node) == oview.getSubjectsOf(node)
            assert view.isSafeToJoin(node, nid(node)) == oview.isSafeToJoin(node, nid(node))

    same()
    view.ringAdd(33, nid(33))
    oview.ringAdd(33, nid(33))
    with pytest.raises(E.UUIDAlreadySeenException):
        view.ringAdd(33, nid(33))  # same host, same id (MembershipViewTest.java:360-366)
    with pytest.raises(E.NodeAlreadyInRingException):
        view.ringAdd(33, (123, 456))  # same host, different id (:369-374)
    with pytest.raises(E.UUIDAlreadySeenException):
        view.ringAdd(34, nid(33))  # different host, same id (:377-383)
    with pytest.raises(E.NodeNotInRingException):
        view.ringDelete(50)
    with pytest.raises(E.NodeNotInRingException):
        view.getObserversOf(50)
    with pytest.raises(E.NodeNotInRingException):
        view.getSubjectsOf(50)
    with pytest.raises(E.UUIDAlreadySeenException):
        view.ringAdd(40, nid(40))  # identifier of node 3 already seen
    same()
    for node in (5, 3, 17):
        view.ringDelete(node)
        oview.ringDelete(node)
    same()
    with pytest.raises(E.UUIDAlreadySeenException):
        view.ringAdd(3, nid(3))  # identifiers are never pruned (R/MembershipView.java:167-201): rejoin with the old id fails
    with pytest.raises(E.UUIDAlreadySeenException):
        view.ringAdd(40, nid(40))
    view.ringAdd(3, (77, 88))  # re-attempt with a new id succeeds (MembershipViewTest.java:430-433)
    oview.ringAdd(3, (77, 88))
    same()
    for node in (45, 46, 60):
        view.ringAdd(node, nid(node))
        oview.ringAdd(node, nid(node))
        same()
    # down to one member, then empty
    for node in [m for m in range(n) if oview.isHostPresent(m)][1:]:
        view.ringDelete(node)
        oview.ringDelete(node)
    same()
    last = [m for m in range(n) if oview.isHostPresent(m)][0]
    assert view.getObserversOf(last) == [] and view.getSubjectsOf(last) == []
    view.ringDelete(last)
    oview.ringDelete(last)
    same()
    assert view.getCurrentConfigurationId() == oview.getCurrentConfigurationId()


def test_bulk_removal_compacts_rings_like_the_oracle(E):
    """A decided cut that only removes members takes the ring-compaction path (no re-sort); a later join the sort path.
    Every ring, table and configuration id must equal the oracle's after each (R/MembershipView.java:123-201)."""
    n, K = 1500, 10
    pop = S.Population.make(n)
    members = list(range(0, n - 20))  # the last 20 endpoints stay outside as future joiners
    eng, view = make_engine(E, pop, K, 9, 4, members=members)
    sim = E.ClusterSimulation(eng)  # apply_cut lives on the simulation facade
    reg, oview = oracle_view(pop, K, members)
    rng = np.random.default_rng(7)

    def same():
        assert view.getMembershipSize() == oview.getMembershipSize()
        for k in range(K):
            assert np.array_equal(view.getRing(k), oview.getRing(k))
        obs, subj, member = view.tables()
        oobs, osubj, omember = oview.tables(n)
        assert np.array_equal(obs, oobs) and np.array_equal(subj, osubj) and np.array_equal(member, omember)
        assert view.getCurrentConfigurationId() == oview.getCurrentConfigurationId()

    same()
    for round_size in (1, 37, 300, 2):
        alive = np.array([m for m in range(n) if oview.isHostPresent(m)])
        cut = sorted(rng.choice(alive, size=round_size, replace=False).tolist())
        sim.apply_cut(cut)
        for node in cut:
            oview.ringDelete(node)
        same()
    joiner = n - 5
    view.ringAdd(joiner, (int(pop.id_hi[joiner]), int(pop.id_lo[joiner])))
    oview.ringAdd(joiner, (int(pop.id_hi[joiner]), int(pop.id_lo[joiner])))
    same()
    cut = [int(x) for x in rng.choice(np.array([m for m in range(n) if oview.isHostPresent(m)]), size=50, replace=False)]
    sim.apply_cut(sorted(cut))
    for node in sorted(cut):
        oview.ringDelete(node)
    same()


def test_config_id_properties(E):
    """MembershipViewTest.java:441-499: N adds -> N distinct ids; same final set in any order -> same id."""
    n, K = 300, 10
    pop = S.Population.make(n)
    eng1, v1 = make_engine(E, pop, K, 9, 4, members=[])
    eng2, v2 = make_engine(E, pop, K, 9, 4, members=[])
    l1, l2 = [], []
    for i in range(n):
        v1.ringAdd(i, (int(pop.id_hi[i]), int(pop.id_lo[i])))
        l1.append(v1.getCurrentConfigurationId())
    for i in range(n - 1, -1, -1):
        v2.ringAdd(i, (int(pop.id_hi[i]), int(pop.id_lo[i])))
        l2.append(v2.getCurrentConfigurationId())
    assert len(set(l1)) == n
    assert all(a != b for a, b in zip(l1[:-1], l2[:-1])) and l1[-1] == l2[-1]


# ------------------------------------------------------------------------------- single cut detector (KATs)
class TestCutDetectionOnDevice:
    """CutDetectionTest.java:42-301 through rapid_cd_* (K=10, H=8, L=2)."""
    K, H, L = 10, 8, 2

    @pytest.fixture()
    def setup(self, E):
        # nodes 0..29 = 127.0.0.2:2..31 (the KAT-7 view); 30..49 = sources 127.0.0.1:1..20; 50.. = other dsts
        pop = S.Population.make(60)
        pop.hostnames = [b"127.0.0.2"] * 30 + [b"127.0.0.1"] * 20 + [b"127.0.0.3", b"127.0.0.4"] + [b"10.9.9.9"] * 8
        pop.ports = np.array(list(range(2, 32)) + list(range(1, 21)) + [2, 2] + list(range(8)), dtype=np.int32)
        eng, view = make_engine(E, pop, self.K, self.H, self.L, members=list(range(30)))
        return E, eng, view, pop

    def test_kat1_to_6(self, setup):
        E, eng, view, pop = setup
        K, H, L = self.K, self.H, self.L
        src = lambda i: 30 + i - 1  # 127.0.0.1:i
        d1, d2, d3 = 0, 50, 51
        a = lambda s, d, r: E.alert(s, d, E.UP, -1, r)
        # KAT-1 (:42-59)
        wb = E.MultiNodeCutDetector(eng, K, H, L)
        for i in range(H - 1):
            assert wb.aggregateForProposal(a(src(i + 1), d1, i)) == [] and wb.getNumProposals() == 0
        assert wb.aggregateForProposal(a(src(H), d1, H - 1)) == [d1] and wb.getNumProposals() == 1
        # KAT-2 (:61-91)
        wb = E.MultiNodeCutDetector(eng, K, H, L)
        for d in (d1, d2):
            for i in range(H - 1):
                assert wb.aggregateForProposal(a(src(i + 1), d, i)) == []
        assert wb.aggregateForProposal(a(src(H), d1, H - 1)) == [] and wb.getNumProposals() == 0
        assert sorted(wb.aggregateForProposal(a(src(H), d2, H - 1))) == [d1, d2] and wb.getNumProposals() == 1
        # KAT-3/4 (:94-189)
        for past_h in (False, True):
            wb = E.MultiNodeCutDetector(eng, K, H, L)
            for d in (d1, d2, d3):
                for i in range(H - 1):
                    assert wb.aggregateForProposal(a(src(i + 1), d, i)) == []
            for d in (d1, d3):
                assert wb.aggregateForProposal(a(src(H), d, H - 1)) == []
                if past_h:
                    assert wb.aggregateForProposal(a(src(H + 1), d, H - 1)) == []
                assert wb.getNumProposals() == 0
            assert sorted(wb.aggregateForProposal(a(src(H), d2, H - 1))) == [d1, d2, d3] and wb.getNumProposals() == 1
        # KAT-5 (:191-230)
        wb = E.MultiNodeCutDetector(eng, K, H, L)
        for i in range(H - 1):
            wb.aggregateForProposal(a(src(i + 1), d1, i))
        for i in range(L - 1):
            wb.aggregateForProposal(a(src(i + 1), d2, i))
        for i in range(H - 1):
            wb.aggregateForProposal(a(src(i + 1), d3, i))
        assert wb.aggregateForProposal(a(src(H), d1, H - 1)) == [] and wb.getNumProposals() == 0
        assert sorted(wb.aggregateForProposal(a(src(H), d3, H - 1))) == [d1, d3] and wb.getNumProposals() == 1
        # KAT-6 (:233-252)
        wb = E.MultiNodeCutDetector(eng, K, H, L)
        proposal = []
        for d in (0, 1, 2):
            for ring in range(K):
                proposal += wb.aggregateForProposal(a(src(1), d, ring))
        assert len(proposal) == 3
        # clear() (:169-178)
        wb.clear()
        assert wb.getNumProposals() == 0
        for i in range(H - 1):
            assert wb.aggregateForProposal(a(src(i + 1), d1, i)) == []
        assert wb.aggregateForProposal(a(src(H), d1, H - 1)) == [d1]

    def test_kat7_link_invalidation(self, setup):  # :254-301
        E, eng, view, pop = setup
        K, H, L = self.K, self.H, self.L
        wb = E.MultiNodeCutDetector(eng, K, H, L)
        dst = 0
        observers = view.getObserversOf(dst)
        assert len(observers) == K
        assert [int(pop.ports[o]) for o in observers] == [4, 7, 3, 16, 20, 30, 29, 19, 15, 28]
        for i in range(H - 1):
            assert wb.aggregateForProposal(E.alert(observers[i], dst, E.DOWN, -1, i)) == []
            assert wb.getNumProposals() == 0
        failed = set()
        for i in range(H - 1, K):
            oo = view.getObserversOf(observers[i])
            failed.add(observers[i])
            for j in range(K):
                assert wb.aggregateForProposal(E.alert(oo[j], observers[i], E.DOWN, -1, j)) == []
                assert wb.getNumProposals() == 0
        ret = wb.invalidateFailingEdges()
        assert len(ret) == 4 and wb.getNumProposals() == 1
        assert set(ret) == failed | {dst}

    def test_ctor_validation(self, setup):
        E, eng, view, pop = setup
        for bad in [(10, 11, 2), (10, 8, 9), (2, 2, 1), (10, 8, 0), (10, 0, 0)]:
            with pytest.raises(E.IllegalArgumentException):
                E.MultiNodeCutDetector(eng, *bad)


# ------------------------------------------------------------------------------------------ population
def run_population(E, eng, records, rec_off, force_exact=False, alert_set=None, trust=True):
    sim = E.ClusterSimulation(eng)
    sim.set_force_exact(force_exact)
    sim.load_streams(records, rec_off)
    if alert_set is not None:
        sim.set_alert_set(alert_set, trust_copies=trust)  # the generators' streams are copies of the round's alerts
    sim.tally()
    return sim, sim.results()


def assert_matches(sim, res, oracle_out, key0, sample=None):
    emit, nprop, pcount, fp = res
    oe, on, oo, op = oracle_out
    bad = np.flatnonzero(emit != oe)
    assert len(bad) == 0, (bad[:8], emit[bad[:8]], oe[bad[:8]])
    assert np.array_equal(nprop, on)
    assert np.array_equal(pcount, np.diff(oo))
    assert np.all((fp != 0) == (oe >= 0))
    R = len(oe)
    idx = range(R) if sample is None else sample
    for r in idx:
        want = sorted(op[oo[r]:oo[r + 1]].tolist(), key=lambda x: key0[x])  # ring-0 comparator order (:346-348)
        assert sim.proposal(r) == want, r
    # equal fingerprints <=> equal proposals
    groups = {}
    for r in range(R):
        if oe[r] >= 0:
            groups.setdefault(tuple(op[oo[r]:oo[r + 1]].tolist()), set()).add(int(fp[r]))
    assert all(len(v) == 1 for v in groups.values())
    assert len({next(iter(v)) for v in groups.values()}) == len(groups)


@pytest.mark.parametrize("seed", range(6))
def test_population_random_adversarial_vs_faithful_oracle(E, seed):
    rng = np.random.default_rng(4000 + seed)
    n_nodes = int(rng.integers(8, 64))
    K = int(rng.integers(3, 12))
    H = int(rng.integers(1, K + 1))
    L = int(rng.integers(1, H + 1))
    pop = S.Population.make(n_nodes)
    n_members = int(rng.integers(max(2, n_nodes // 2), n_nodes + 1))
    members = sorted(rng.permutation(n_nodes)[:n_members].tolist())
    eng, view = make_engine(E, pop, K, H, L, members=members)
    reg, oview = oracle_view(pop, K, members)
    cfg = view.getCurrentConfigurationId()
    assert cfg == oview.getCurrentConfigurationId()
    member = np.zeros(n_nodes, dtype=np.uint8)
    member[members] = 1
    recs, off = [], [0]
    for r in range(300):
        hot = rng.permutation(n_nodes)[: int(rng.integers(1, min(n_nodes, 16) + 1))]
        n_rec = int(rng.integers(0, 700))
        recs.append(random_stream(rng, n_nodes, K, member, cfg, n_rec, hot, p_eob=float(rng.choice([0.02, 0.1, 0.4, 1.0])),
                                  p_multi=float(rng.choice([0.0, 0.15, 0.5]))))
        off.append(off[-1] + n_rec)
    records, rec_off = np.concatenate(recs), np.array(off)
    oracle_out = O.sim_run(oview, K, H, L, pop.id_hi, pop.id_lo, records, rec_off, nthreads=4)
    key0 = [oview.ringKey(0, i) for i in range(n_nodes)]
    for force in (False, True):
        sim, res = run_population(E, eng, records, rec_off, force_exact=force)
        assert_matches(sim, res, oracle_out, key0, sample=range(0, 300, 7))


@pytest.mark.parametrize("name,n,f,K,H,L", [("C1", 50, 1, 3, 3, 1), ("C2", 2000, 20, 10, 9, 4), ("C2", 500, 25, 10, 8, 2),
                                            ("C3a", 1500, 75, 10, 9, 4), ("C3b", 1500, 40, 10, 9, 4)])
def test_population_scenarios_vs_faithful_oracle(E, name, n, f, K, H, L):
    """BASELINE configs 1-2 at full size, config 3 at reduced N (the faithful oracle re-scans preProposal after
    every batch, as the Java does, which is quadratic); config 3 at full size is covered by the fast oracle below."""
    pop = S.Population.make(n)
    eng, view = make_engine(E, pop, K, H, L)
    reg, oview = oracle_view(pop, K)
    obs, subj, member = view.tables()
    cfg = view.getCurrentConfigurationId()
    sc = S.build_scenario(name, subj, cfg, n=n, f=f, H=H, L=L)
    rx = np.arange(0, len(sc.receivers), max(1, len(sc.receivers) // 200))  # oracle on a sample of receivers
    sub_off = np.zeros(len(rx) + 1, dtype=np.int64)
    parts = []
    for i, r in enumerate(rx):
        parts.append(sc.records[sc.rec_off[r]:sc.rec_off[r + 1]])
        sub_off[i + 1] = sub_off[i] + len(parts[-1])
    oracle_out = O.sim_run(oview, K, H, L, pop.id_hi, pop.id_lo, np.concatenate(parts), sub_off, nthreads=8)
    key0 = [oview.ringKey(0, i) for i in range(n)]
    sim, res = run_population(E, eng, sc.records, sc.rec_off)
    st = sim.stats()  # the engine's counters: read before the next run on this engine resets them
    assert st["records_consumed"] <= len(sc.records)
    if np.diff(sc.rec_off).max() >= 3 * 256:  # streams of several windows: the cold / fast windows were exercised
        assert st["lean_windows"] > 0
    # same round with the distinct alert set declared (index + validation from the set, kTrusted tally) and with the
    # per-delivery filter forced on (knob 64): identical results
    for kw in (dict(alert_set=sc.batches.recs), dict(force_exact=64), dict(force_exact=8)):
        sim2, res2 = run_population(E, eng, sc.records, sc.rec_off, **kw)
        assert all(np.array_equal(a, b) for a, b in zip(res, res2)), kw
    sub = tuple(a[rx] for a in res)

    class Sub:
        def proposal(self, i):
            return sim.proposal(int(rx[i]))
    assert_matches(Sub(), sub, oracle_out, key0, sample=range(0, len(rx), 5))
    # every receiver against the optimised CPU formulation
    fe, fn, fo, fpp = O.fast_sim_run(n, K, H, L, cfg, obs, subj, member, sc.records, sc.rec_off, nthreads=8)
    assert np.array_equal(res[0], fe) and np.array_equal(res[1], fn) and np.array_equal(res[2], np.diff(fo))


def test_round_decides_and_applies_cut(E):
    """Tally -> vote count (quorum N - floor((N-1)/4), R/FastPaxos.java:145-150) -> decideViewChange
    (R/MembershipService.java:385-430): decided cut and the next configuration id equal the oracle's."""
    n, K, H, L = 2000, 10, 9, 4
    pop = S.Population.make(n)
    eng, view = make_engine(E, pop, K, H, L)
    reg, oview = oracle_view(pop, K)
    obs, subj, member = view.tables()
    cfg = view.getCurrentConfigurationId()
    sc = S.build_scenario("C2", subj, cfg, n=n, f=20, H=H, L=L)
    sim = E.ClusterSimulation(eng)
    sim.load_streams(sc.records, sc.rec_off)
    sim.tally()
    emit, nprop, pcount, fp = sim.results()
    rr = sim.count_votes()
    # expected vote table from the per-receiver proposals themselves
    fe, fn, fo, fpp = O.fast_sim_run(n, K, H, L, cfg, obs, subj, member, sc.records, sc.rec_off, nthreads=8)
    votes = {}
    for r in range(len(fe)):
        if fe[r] >= 0:
            t = tuple(fpp[fo[r]:fo[r + 1]].tolist())
            votes[t] = votes.get(t, 0) + 1
    best = max(votes.items(), key=lambda kv: kv[1])
    quorum = n - (n - 1) // 4
    assert rr.membership_size == n and rr.quorum == quorum
    assert rr.votes_total == sum(votes.values()) and rr.votes_winner == best[1]
    assert rr.decided == (1 if best[1] >= quorum else 0) == 1
    key0 = [oview.ringKey(0, i) for i in range(n)]
    cut = sim.decided_cut()
    assert cut == sorted(best[0], key=lambda x: key0[x]) and sorted(cut) == sc.faulty.tolist()
    new_cfg = sim.apply_cut(cut)
    svc = O.AlertBatchService(oview, K, H, L, pop.id_hi, pop.id_lo)
    svc.decideViewChange(cut)
    assert new_cfg == oview.getCurrentConfigurationId() and view.getMembershipSize() == n - 20
    o2, s2, m2 = view.tables()
    oo2, os2, om2 = oview.tables(n)
    assert np.array_equal(m2, om2) and np.array_equal(s2, os2)
    assert np.array_equal(o2[om2 == 1], oo2[om2 == 1])
    # second round in the new configuration: alerts stamped with the OLD configuration id are all filtered
    sim.load_streams(sc.records, sc.rec_off)
    sim.tally()
    e2, n2, p2, f2 = sim.results()
    assert np.all(e2 == -1) and np.all(p2 == 0)
    # and a fresh crash burst in the new configuration decides again
    sc2 = S.build_scenario("C2", s2.copy(), new_cfg, n=n, f=10, H=H, L=L, seed_fault=9, kind="crash",
                           receivers=np.flatnonzero(m2 == 1)[::3])
    alive = m2[sc2.faulty] == 1
    if alive.all():
        sim.load_streams(sc2.records, sc2.rec_off)
        rr2, cfg3 = sim.round(apply=False)
        fe2 = O.fast_sim_run(n, K, H, L, new_cfg, o2, s2, m2, sc2.records, sc2.rec_off, nthreads=8)
        assert np.array_equal(sim.results()[0], fe2[0])


def test_churn_round_joins_and_crashes_in_one_cut(E):
    """UP alerts about joiners (non-members, expected observers) and DOWN alerts about crashed members
    in the same configuration: every surviving member announces crashed + joiners, the fast round decides, and
    rapid_apply_cut removes / adds them like decideViewChange (R/MembershipService.java:385-430) -- same rings, tables
    and configuration id as the oracle."""
    n, n_out, n_crash, n_join, K, H, L = 1200, 60, 25, 30, 10, 9, 4
    pop = S.Population.make(n)
    members = list(range(0, n - n_out))
    eng, view = make_engine(E, pop, K, H, L, members=members)
    reg, oview = oracle_view(pop, K, members)
    obs, subj, member = view.tables()
    oobs, osubj, omember = oview.tables(n)
    assert np.array_equal(obs, oobs) and np.array_equal(member, omember)  # incl. expected observers of non-members (Q3)
    cfg = view.getCurrentConfigurationId()
    sc = S.build_churn_scenario(obs, member, cfg, n_crash, n_join, H, L)
    sim = E.ClusterSimulation(eng)
    sim.load_streams(sc.records, sc.rec_off)
    sim.set_alert_set(sc.batches.recs, trust_copies=True)
    sim.tally()
    emit, nprop, pcount, fp = sim.results()
    fe, fn, fo, fpp = O.fast_sim_run(n, K, H, L, cfg, obs, subj, member, sc.records, sc.rec_off, nthreads=8)
    assert np.array_equal(emit, fe) and np.array_equal(nprop, fn) and np.array_equal(pcount, np.diff(fo))
    for r in range(0, len(fe), 37):
        assert sorted(sim.proposal(r)) == sorted(fpp[fo[r]:fo[r + 1]].tolist()) == sc.faulty.tolist()
    rr = sim.count_votes()
    assert rr.decided == 1 and rr.membership_size == n - n_out and rr.votes_winner == len(sc.receivers)
    cut = sim.decided_cut()
    key0 = [oview.ringKey(0, i) for i in range(n)]
    assert cut == sorted(sc.faulty.tolist(), key=lambda x: key0[x])
    new_cfg = sim.apply_cut(cut)
    # the oracle learns the joiners' NodeIds from the UP alerts of one receiver's stream, then decides
    svc = O.AlertBatchService(oview, K, H, L, pop.id_hi, pop.id_lo)
    r0 = sc.records[sc.rec_off[0]:sc.rec_off[1]]
    beg = 0
    for e in np.flatnonzero(r0["flags"] & S.FLAG_LAST_IN_BATCH) + 1:
        svc.handleBatchedAlertMessage(r0[beg:e])
        beg = int(e)
    svc.decideViewChange(cut)
    assert new_cfg == oview.getCurrentConfigurationId()
    assert view.getMembershipSize() == oview.getMembershipSize() == (n - n_out) - n_crash + n_join
    for k in range(K):
        assert np.array_equal(view.getRing(k), oview.getRing(k))
    o2, s2, m2 = view.tables()
    oo2, os2, om2 = oview.tables(n)
    assert np.array_equal(m2, om2) and np.array_equal(s2, os2) and np.array_equal(o2, oo2)
    for j in sc.joiners[:5]:
        assert view.isHostPresent(int(j)) and view.getObserversOf(int(j)) == oview.getObserversOf(int(j))


def test_no_quorum_when_receivers_disagree(E):
    n, K, H, L = 40, 10, 8, 2
    pop = S.Population.make(n)
    eng, view = make_engine(E, pop, K, H, L)
    cfg = view.getCurrentConfigurationId()
    recs, off = [], [0]
    for r in range(30):  # every receiver hears about a different single subject
        a = np.zeros(1, dtype=S.ALERT_DTYPE)
        a["dst"], a["ring_mask"], a["status"], a["cfg_id"], a["flags"] = r, (1 << H) - 1, S.DOWN, cfg, 1
        recs.append(a)
        off.append(off[-1] + 1)
    sim = E.ClusterSimulation(eng)
    sim.load_streams(np.concatenate(recs), np.array(off))
    rr, _ = sim.round(apply=True)
    assert rr.decided == 0 and rr.votes_total == 30 and rr.votes_winner == 1
    assert view.getMembershipSize() == n
    with pytest.raises(E.RapidError):
        sim.decided_cut()


def test_empty_population_and_misuse(E):
    n = 30
    pop = S.Population.make(n)
    eng = E.Engine(n_max=n, K=10, H=9, L=4)
    sim = E.ClusterSimulation(eng)
    with pytest.raises(E.RapidError) as ei:
        sim.tally()  # no view
    assert ei.value.code == -7
    view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)
    with pytest.raises(E.RapidError):
        sim.tally()  # no streams
    sim.load_streams(np.zeros(0, dtype=S.ALERT_DTYPE), np.zeros(1, dtype=np.int64))
    sim.tally()
    rr = sim.count_votes()
    assert rr.decided == 0 and rr.votes_total == 0
    sim.load_streams(np.zeros(0, dtype=S.ALERT_DTYPE), np.zeros(4, dtype=np.int64))  # three receivers, no alerts
    sim.tally()
    emit, nprop, pcount, fp = sim.results()
    assert emit.tolist() == [-1, -1, -1] and pcount.tolist() == [0, 0, 0]
    with pytest.raises(E.IllegalArgumentException):
        E.MembershipView(E.Engine(n_max=4, K=10, H=9, L=4)).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)


def test_proposal_capacity_is_reported(E):
    n, K, H, L = 64, 10, 4, 2
    pop = S.Population.make(n)
    eng = E.Engine(n_max=n, K=K, H=H, L=L, max_cut=8)
    view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)
    cfg = view.getCurrentConfigurationId()
    a = np.zeros(20, dtype=S.ALERT_DTYPE)
    a["dst"], a["ring_mask"], a["status"], a["cfg_id"] = np.arange(20), (1 << H) - 1, S.DOWN, cfg
    a["flags"][-1] = 1
    sim = E.ClusterSimulation(eng)
    sim.load_streams(a, np.array([0, 20]))
    sim.tally()
    emit, nprop, pcount, fp = sim.results()
    assert emit[0] == 0 and pcount[0] == -1
    with pytest.raises(E.RapidError) as ei:
        sim.proposal(0)
    assert ei.value.code == -5


def test_full_size_c3_against_fast_oracle(E):
    """BASELINE config 3 (N=10,000, K=10, 5% one-way failures) at full size: every one of the ~9,500 receivers
    against the optimised CPU formulation, for the raw (blocked) and the closed (stable cut) variant."""
    n, K, H, L = 10000, 10, 9, 4
    pop = S.Population.make(n)
    eng, view = make_engine(E, pop, K, H, L)
    obs, subj, member = view.tables()
    cfg = view.getCurrentConfigurationId()
    for name in ("C3a", "C3b"):
        sc = S.build_scenario(name, subj, cfg)
        fe, fn, fo, fpp = O.fast_sim_run(n, K, H, L, cfg, obs, subj, member, sc.records, sc.rec_off, nthreads=8)
        sim, (emit, nprop, pcount, fp) = run_population(E, eng, sc.records, sc.rec_off)
        assert np.array_equal(emit, fe) and np.array_equal(nprop, fn) and np.array_equal(pcount, np.diff(fo))
        # the proposal CONTENTS of every receiver: the kernel's fingerprint restated on the host over the oracle's lists
        assert np.array_equal(fp, proposal_fingerprints(fo, fpp, fe >= 0))
        for r in range(0, len(fe), 97):
            assert sorted(sim.proposal(r)) == fpp[fo[r]:fo[r + 1]].tolist()
        # bench.py's own form of the round: the distinct alerts declared (index built from the set, not from the streams), the
        # deliveries vouched for as copies, the 20-byte records tallied where they lie with the tables in LDS -- the
        # instantiation `value` is measured with
        sim_b, res_b = run_population(E, eng, sc.records, sc.rec_off, alert_set=sc.batches.recs, trust=True)
        info_b = sim_b.index_info()
        assert info_b["alerts_prevalidated"] == 1 and info_b["alert_set_declared"] == 1 and info_b["dict_mode"] == 1
        assert all(np.array_equal(a_, b_) for a_, b_ in zip((emit, nprop, pcount, fp), res_b))
        for r in range(0, len(fe), 997):
            assert sorted(sim_b.proposal(r)) == fpp[fo[r]:fo[r + 1]].tolist()
        sim_b, res_b = run_population(E, eng, sc.records, sc.rec_off, alert_set=sc.batches.recs, trust=False)
        assert sim_b.index_info()["alerts_prevalidated"] == 0
        assert all(np.array_equal(a_, b_) for a_, b_ in zip((emit, nprop, pcount, fp), res_b))
        # ... and the faithful restatement of the Java (quadratic: a sample of >= 200 receivers) on the same streams
        rx = np.arange(0, len(fe), len(fe) // 210)
        sub_off = np.zeros(len(rx) + 1, dtype=np.int64)
        parts = []
        for i, r in enumerate(rx):
            parts.append(sc.records[sc.rec_off[r]:sc.rec_off[r + 1]])
            sub_off[i + 1] = sub_off[i] + len(parts[-1])
        reg, oview = oracle_view(pop, K)
        oe, on, oo, op = O.sim_run(oview, K, H, L, pop.id_hi, pop.id_lo, np.concatenate(parts), sub_off, nthreads=32)
        assert len(rx) >= 200 and np.array_equal(emit[rx], oe) and np.array_equal(nprop[rx], on)
        assert np.array_equal(pcount[rx], np.diff(oo)) and np.array_equal(fp[rx], proposal_fingerprints(oo, op, oe >= 0))
        rr = sim.count_votes()
        if name == "C3a":
            assert rr.decided == 0  # healthy subjects stuck in [L, H) block every proposal
        else:
            assert rr.decided == 1 and sorted(sim.decided_cut()) == sc.faulty.tolist()
            # size-independent property: re-delivering the same batches in another order decides the same cut
            recs2, off2, _ = S.deliver(sc.batches, sc.receivers[::4], 12345)
            sim.load_streams(recs2, off2)
            rr2, _ = sim.round(apply=False)
            assert rr2.votes_winner == len(off2) - 1  # a quarter of the receivers cannot reach the quorum, but they all agree
            emit2, nprop2, pcount2, fp2 = sim.results()
            want_fp = proposal_fingerprints(np.array([0, len(sc.faulty)]), sc.faulty, [True])[0]
            assert np.all(emit2 >= 0) and np.all(pcount2 == len(sc.faulty)) and np.all(fp2 == want_fp)  # every receiver proposes exactly the fault set
            assert sorted(sim.proposal(0)) == sc.faulty.tolist() and sorted(sim.proposal(len(emit2) - 1)) == sc.faulty.tolist()


def test_classic_round_recovers_a_population_without_fast_quorum(E):
    """the receivers disagree (no fast quorum), one classic round decides a cut every acceptor voted
    for under the coordinator rule (R/Paxos.java:271-328), and the view change is applied like the oracle's."""
    from oracle import paxos_oracle as PX
    n, K, H, L = 40, 10, 8, 2
    pop = S.Population.make(n)
    eng, view = make_engine(E, pop, K, H, L)
    reg, oview = oracle_view(pop, K)
    cfg = view.getCurrentConfigurationId()
    # receivers 0..11 hear about subject 3, 12..19 about subject 5, 20..29 about subject 7: 12 + 8 + 10 votes, quorum is 31
    subject = [3] * 12 + [5] * 8 + [7] * 10
    recs, off = [], [0]
    for r in range(30):
        a = np.zeros(1, dtype=S.ALERT_DTYPE)
        a["dst"], a["ring_mask"], a["status"], a["cfg_id"], a["flags"] = subject[r], (1 << H) - 1, S.DOWN, cfg, 1
        recs.append(a)
        off.append(off[-1] + 1)
    sim = E.ClusterSimulation(eng)
    sim.load_streams(np.concatenate(recs), np.array(off))
    rr, _ = sim.round(apply=True)
    assert rr.decided == 0 and rr.votes_total == 30 and rr.votes_winner == 12
    arrival = list(np.random.default_rng(4).permutation(30))
    # the same round, message by message, on the restated Paxos.java: 30 live acceptors of 40 members
    net = PX.Network(n, configurationId=cfg, live=list(range(30)), seed=1)
    for r in range(30):
        net.nodes[r].propose((subject[r],))
    net.run()
    assert not net.decisions
    net.nodes[0].startClassicPaxosRound()
    for a in arrival:
        assert net.deliver_one(int(a)).kind == PX.PHASE1A
    net.run()
    want = {v for _, v in net.decisions}
    assert len(net.decisions) == 30 and len(want) == 1
    res, cut, new_cfg = sim.classic_round(arrival=arrival, apply=True)
    assert res["decided"] and tuple(cut) == want.pop()
    svc = O.AlertBatchService(oview, K, H, L, pop.id_hi, pop.id_lo)
    svc.decideViewChange(cut)
    assert new_cfg == oview.getCurrentConfigurationId() == view.getCurrentConfigurationId()
    assert view.getMembershipSize() == n - 1 and not view.isHostPresent(cut[0])


# ------------------------------------------------------------------------- RCCL path on one GPU
def test_vote_count_through_a_one_rank_communicator(E):
    """The sharded vote count (R/FastPaxos.java:104, 125-156 with the N x N unicast fan-out of
    R/UnicastToAllBroadcaster.java:46-52 replaced by collectives) executed on ONE GPU: an engine with a 1-rank RCCL
    communicator runs the collectives of rapid_sim_count_votes -- by default the ONE all-gather of the ranks' local
    answers + the device-side merge (a round whose voters disagree then falls through to the general count), and under
    knob 512 always the general count: the histogram all-reduce, the max-reduce that elects the rank holding the
    representative (rank-tag trick), the max-reduce that hands its list to everybody, the sum-reduce of the verification
    counters -- and must decide exactly what the engine without a communicator decides: a crash burst with a quorum, a
    churn round (joins + crashes), and a population without a fast quorum."""
    n, K, H, L = 2000, 10, 9, 4
    pop = S.Population.make(n)
    members = list(range(0, n - 40))

    def run(with_comm, case, knob=0):
        eng, view = make_engine(E, pop, K, H, L, members=members)
        if with_comm:
            eng.comm_init(E.comm_unique_id(), 0, 1)
            assert eng.comm_info() == (0, 1)
        else:
            assert eng.comm_info() == (0, 1)
        obs, subj, member = view.tables()
        cfg = view.getCurrentConfigurationId()
        if case == "crash":
            sc = S.build_churn_scenario(obs, member, cfg, 20, 0, H, L)
        elif case == "churn":
            sc = S.build_churn_scenario(obs, member, cfg, 12, 15, H, L)
        elif case == "noquorum":  # 3 % of the (receiver, batch) deliveries are lost: too few receivers propose at all
            sc = S.build_churn_scenario(obs, member, cfg, 25, 0, H, L, materialise=False)
            recs, off, nb = S.deliver(sc.batches, sc.receivers, 77, loss=0.03)
            sc.records, sc.rec_off = recs, off
        elif case == "dissent":  # ten receivers saw a different fault set: their votes do not stop the quorum of the others
            sc = S.build_churn_scenario(obs, member, cfg, 20, 0, H, L)
            sc3 = S.build_churn_scenario(obs, member, cfg, 20, 0, H, L, seed_fault=5)
            b = int(sc3.rec_off[10])
            sc.rec_off = np.concatenate([sc.rec_off, sc3.rec_off[1:11] + sc.rec_off[-1]])
            sc.records = np.concatenate([sc.records, sc3.records[:b]])
        else:  # two groups of receivers see two different fault sets: two proposals, the merge must hand over to the general count
            sc = S.build_churn_scenario(obs, member, cfg, 20, 0, H, L)
            sc3 = S.build_churn_scenario(obs, member, cfg, 20, 0, H, L, seed_fault=5)
            a, b = int(sc.rec_off[700]), int(sc3.rec_off[400])
            sc.records = np.concatenate([sc.records[:a], sc3.records[:b]])
            sc.rec_off = np.concatenate([sc.rec_off[:701], sc3.rec_off[1:401] + a])
        sim = E.ClusterSimulation(eng)
        sim.load_streams(sc.records, sc.rec_off)
        sim.set_force_exact(knob)
        sim.tally()
        rr = sim.count_votes()
        out = dict(decided=rr.decided, cut_size=rr.cut_size, quorum=rr.quorum, votes_total=rr.votes_total,
                   votes_winner=rr.votes_winner, membership=rr.membership_size,
                   cut=sim.decided_cut() if rr.decided else None, results=[a.copy() for a in sim.results()])
        if rr.decided:
            out["new_cfg"] = sim.apply_cut(out["cut"])
        eng.close()
        return out, sc

    for case in ("crash", "churn", "noquorum", "conflict", "dissent"):
        a, sc = run(False, case)
        # 2048: count without the statistics the tally kernel gathers; 262144: verification on the node lists instead of the slot bitmaps
        for with_comm, knob in ((True, 0), (True, 512), (False, 2048), (False, 262144), (True, 262144)):
            b, _ = run(with_comm, case, knob)
            for k in ("decided", "cut_size", "quorum", "votes_total", "votes_winner", "membership", "cut", "new_cfg"):
                assert a.get(k) == b.get(k), (case, knob, k, a.get(k), b.get(k))
            assert all(np.array_equal(x, y) for x, y in zip(a["results"], b["results"])), (case, knob)
        assert a["decided"] == (0 if case in ("noquorum", "conflict") else 1)
        if case == "conflict":
            assert (a["votes_total"], a["votes_winner"]) == (1100, 700)
        if case == "dissent":
            assert a["votes_total"] == a["votes_winner"] + 10 == len(sc.rec_off) - 1


def test_sharded_vote_count_merges_the_ranks_local_answers(E, test_build):
    """The merge every rank runs after the all-gather of rapid_sim_count_votes (vote_merge_kernel), with one GPU standing in
    for three ranks: the receivers of a round are cut into three shards, each shard is tallied and counted on its own
    (rapid_debug_vote_segment = what the rank would contribute), and the merged answer must be the answer of the whole
    population counted at once (R/FastPaxos.java:141-150: votes for the one proposal = sum over the ranks).  A shard
    nobody proposes on contributes nothing; dissenters do not matter once the common candidate has a quorum; a round
    whose shards hold different candidates, or several proposals without a quorum, is recognised (status 2: the general
    histogram count would run) instead of being merged.  (Deliveries lost on the way do
    not make voters disagree: a receiver that misses a batch never reaches H for its subjects and does not vote at all.)"""
    n, K, H, L = 2000, 10, 9, 4
    pop = S.Population.make(n)
    eng, view = make_engine(E, pop, K, H, L)
    obs, subj, member = view.tables()
    cfg = view.getCurrentConfigurationId()
    sim = E.ClusterSimulation(eng)

    def shard(records, rec_off, a, b):
        return records[rec_off[a]: rec_off[b]], (rec_off[a: b + 1] - rec_off[a]).astype(np.int64)

    def segments_of(sc, cuts, declared=None):
        segs = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            recs, off = shard(sc.records, sc.rec_off, a, b)
            sim.load_streams(recs, off)
            sim.set_alert_set(sc.batches.recs if declared is None else declared)
            sim.tally()
            segs.append(sim.vote_segment())
        return segs

    def whole(sc):
        sim.load_streams(sc.records, sc.rec_off)
        sim.set_alert_set(sc.batches.recs)
        sim.tally()
        rr = sim.count_votes()
        return rr, (sim.decided_cut() if rr.decided else None)

    sc = S.build_churn_scenario(obs, member, cfg, 20, 0, H, L)
    R = len(sc.receivers)
    rr0, cut0 = whole(sc)
    assert rr0.decided == 1 and rr0.votes_winner == R
    segs = segments_of(sc, [0, R // 3, R // 3 + 700, R])
    status, rr = sim.merge_vote_segments(segs)
    assert status == 1
    assert (rr.decided, rr.cut_size, rr.votes_total, rr.votes_winner, rr.quorum) == (1, rr0.cut_size, R, R, rr0.quorum)
    assert sim.decided_cut() == cut0
    # a rank whose receivers got nothing delivered (nobody proposes there) adds no votes: 2 of 3 shards < quorum here
    empty = np.zeros(0, dtype=S.ALERT_DTYPE)
    sim.load_streams(empty, np.zeros(R // 3 + 1, dtype=np.int64))
    sim.set_alert_set(sc.batches.recs)
    sim.tally()
    silent = sim.vote_segment()
    status, rr = sim.merge_vote_segments([segs[0], silent, segs[2]])
    assert status == 1 and rr.votes_winner == R - 700 == rr.votes_total and rr.cut_size == rr0.cut_size
    assert rr.decided == (1 if R - 700 >= rr0.quorum else 0)
    status, rr = sim.merge_vote_segments([silent, silent])
    assert status == 1 and rr.votes_winner == 0 and rr.decided == 0
    # two proposals in one round: 600 receivers that saw this round's crashes and 500 that saw a different fault set (same
    # size, different nodes).  Unanimous shards that disagree with each other, and a shard that disagrees within itself.
    sc3 = S.build_churn_scenario(obs, member, cfg, 20, 0, H, L, seed_fault=5)
    assert len(sc3.faulty) == len(sc.faulty) and not np.array_equal(sc3.faulty, sc.faulty)
    ra, oa = shard(sc.records, sc.rec_off, 0, 600)
    rb, ob = shard(sc3.records, sc3.rec_off, 0, 500)

    class Mixed:
        records = np.concatenate([ra, rb])
        rec_off = np.concatenate([oa, ob[1:] + oa[-1]])

        class batches:
            recs = np.concatenate([sc.batches.recs, sc3.batches.recs])

    both = segments_of(Mixed, [0, 600, 1100])
    status, _ = sim.merge_vote_segments(both)
    assert status == 2
    status, rr = sim.merge_vote_segments([both[1]])
    assert status == 1 and rr.votes_winner == 500 and rr.decided == 0 and rr.cut_size == 20
    status, _ = sim.merge_vote_segments(segments_of(Mixed, [0, 1100]))
    assert status == 2
    status, _ = sim.merge_vote_segments([segs[0], segments_of(Mixed, [0, 1100])[0]])
    assert status == 2
    rr, _ = whole(Mixed)  # (what the general count says about that round: the larger group wins, without a quorum)
    assert (rr.decided, rr.votes_total, rr.votes_winner) == (0, 1100, 600)
    # dissenters do not stop a merge once the candidate has a quorum: ten receivers that saw the other fault set sit at the
    # END of the last shard (the shard's candidate -- the proposal its lowest voter or its winning bucket holds -- is the
    # cluster's); at the FRONT of a shard they become that shard's candidate when the statistics come from the tally
    # kernel, which the merge reports as a disagreement (status 2) -- with the counting kernel the shard's plurality wins
    rc, oc = shard(sc3.records, sc3.rec_off, 0, 10)

    class Tail:
        records = np.concatenate([sc.records, rc])
        rec_off = np.concatenate([sc.rec_off, oc[1:] + sc.rec_off[-1]])
        batches = Mixed.batches

    segs_t = segments_of(Tail, [0, R // 3, R // 3 + 700, R + 10])
    status, rr = sim.merge_vote_segments(segs_t)
    assert status == 1 and (rr.decided, rr.votes_winner, rr.votes_total, rr.cut_size) == (1, R, R + 10, rr0.cut_size)
    assert sim.decided_cut() == cut0
    rr, cut = whole(Tail)  # the single-population count of the same round agrees
    assert (rr.decided, rr.votes_winner, rr.votes_total) == (1, R, R + 10) and cut == cut0
    eng.close()


def test_streams_handed_over_in_device_memory(E):
    """rapid_sim_load_streams_device: the 20-byte records already sit in device memory (as a producer on the same GPU
    would leave them; here allocated through the HIP runtime the library itself uses) -- copied into the engine's buffer,
    with the same results as the host form; the record buffer may be released right after the call, the offsets stay
    borrowed.  rapid_sim_attach_streams_device: the same records tallied IN PLACE, nothing copied (both buffers borrowed)."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes, hip.hipMemcpy.argtypes, hip.hipFree.argtypes = [C.POINTER(C.c_void_p), C.c_size_t], [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int], [C.c_void_p]

    def to_device(a):
        a = np.ascontiguousarray(a)
        ptr = C.c_void_p()
        assert hip.hipMalloc(C.byref(ptr), max(a.nbytes, 16)) == 0
        assert hip.hipMemcpy(ptr, a.ctypes.data_as(C.c_void_p), a.nbytes, 1) == 0  # hipMemcpyHostToDevice, synchronous
        return ptr

    n, K, H, L = 1500, 10, 9, 4
    pop = S.Population.make(n)
    eng, view = make_engine(E, pop, K, H, L)
    obs, subj, member = view.tables()
    cfg = view.getCurrentConfigurationId()
    sc = S.build_churn_scenario(obs, member, cfg, 15, 0, H, L)
    sim = E.ClusterSimulation(eng)
    sim.load_streams(sc.records, sc.rec_off)
    sim.tally()
    want = [a.copy() for a in sim.results()]
    rr0 = sim.count_votes()
    raw = np.ascontiguousarray(sc.records).view(np.uint8).reshape(-1)
    d_rec, d_off = to_device(raw), to_device(np.ascontiguousarray(sc.rec_off, dtype=np.int64))
    sim.load_streams_device(d_rec.value, raw.nbytes, d_off.value, len(sc.rec_off) - 1)
    assert hip.hipFree(d_rec) == 0  # borrowed for the call only
    sim.set_alert_set(sc.batches.recs)
    sim.tally()
    got = sim.results()
    assert all(np.array_equal(a, b) for a, b in zip(want, got))
    rr = sim.count_votes()
    assert (rr.decided, rr.votes_winner, rr.cut_size) == (rr0.decided, rr0.votes_winner, rr0.cut_size) and rr.decided == 1
    # in place: the engine reads the caller's buffer (at a 4-byte aligned, not 16-byte aligned, address)
    pad = np.zeros(12, dtype=np.uint8)
    d_rec2 = to_device(np.concatenate([pad, raw]))
    sim.attach_streams_device(d_rec2.value + 12, raw.nbytes, d_off.value, len(sc.rec_off) - 1)
    for declared in (False, True):
        if declared:
            sim.set_alert_set(sc.batches.recs, trust_copies=True)
        sim.new_round()
        sim.tally()
        assert all(np.array_equal(a, b) for a, b in zip(want, sim.results())), declared
        rr = sim.count_votes()
        assert (rr.decided, rr.votes_winner, rr.cut_size) == (rr0.decided, rr0.votes_winner, rr0.cut_size)
    # ... and the round's distinct alerts declared where they lie, too (rapid_sim_set_alert_set_device)
    d_al = to_device(np.ascontiguousarray(sc.batches.recs).view(np.uint8).reshape(-1))
    sim.attach_streams_device(d_rec2.value + 12, raw.nbytes, d_off.value, len(sc.rec_off) - 1)
    sim.set_alert_set_device(d_al.value, len(sc.batches.recs), trust_copies=True)
    sim.tally()
    info = sim.index_info()
    assert info["alert_set_declared"] == 1 and info["alerts_prevalidated"] == 1
    assert all(np.array_equal(a, b) for a, b in zip(want, sim.results()))
    with pytest.raises(E.IllegalArgumentException):
        sim.set_alert_set_device(d_al.value + 2, len(sc.batches.recs))
    with pytest.raises(E.IllegalArgumentException):  # more alerts than the buffer holds: refused on the host, not a fault in the index kernel
        sim.set_alert_set_device(d_al.value, len(sc.batches.recs) + 1, alerts_bytes=20 * len(sc.batches.recs))
    with pytest.raises(E.IllegalArgumentException):
        sim.attach_streams_device(d_rec2.value + 13, raw.nbytes, d_off.value, len(sc.rec_off) - 1)  # not 4-byte aligned
    # offsets that run past the records, or are not ascending: checked on the device, ahead of the round's kernels, without
    # a copy to the host -- no stream is read, and the round's results come back as IllegalArgumentException
    bad_off = np.ascontiguousarray(sc.rec_off, dtype=np.int64).copy()
    bad_off[3], bad_off[4] = bad_off[4], bad_off[3]
    d_bad = to_device(bad_off)
    for args in ((d_rec2.value + 12, raw.nbytes - 20, d_off.value), (d_rec2.value + 12, raw.nbytes, d_bad.value)):
        sim.attach_streams_device(*args, len(sc.rec_off) - 1)
        with pytest.raises(E.IllegalArgumentException):  # nothing declared: the pass over the delivered records asks for their
            sim.tally()                                  # number (a count past the records: here); else the kernel's check
            sim.results()
        sim.attach_streams_device(*args, len(sc.rec_off) - 1)
        sim.set_alert_set(sc.batches.recs, trust_copies=True)  # declared: nothing on the host looks at the offsets ...
        sim.tally()
        with pytest.raises(E.IllegalArgumentException):        # ... the kernels did
            sim.results()
        with pytest.raises(E.IllegalArgumentException):
            sim.count_votes()
    sim.attach_streams_device(d_rec2.value + 12, raw.nbytes, d_off.value, len(sc.rec_off) - 1)  # ... and a good set afterwards is fine
    sim.tally()
    assert all(np.array_equal(a, b) for a, b in zip(want, sim.results()))
    assert hip.hipFree(d_bad) == 0
    eng.close()
    assert hip.hipFree(d_off) == 0 and hip.hipFree(d_rec2) == 0 and hip.hipFree(d_al) == 0



def test_declared_alert_set_that_does_not_cover_the_streams_is_rejected(E):
    """rapid_sim_set_alert_set is a promise the engine checks: a delivered report about a subject (or ring of a cold
    subject) missing from the declared set -> RAPID_EINVAL at the next read of results, not silently different cuts; a set
    from another configuration is simply not trusted (per-delivery filter), and an honest set changes nothing."""
    n, K, H, L = 2000, 10, 9, 4
    pop = S.Population.make(n)
    eng, view = make_engine(E, pop, K, H, L)
    obs, subj, member = view.tables()
    cfg = view.getCurrentConfigurationId()
    sc = S.build_scenario("C2", subj, cfg, n=n, f=20, H=H, L=L)
    sim, want = run_population(E, eng, sc.records, sc.rec_off)
    # honest declaration
    sim, res = run_population(E, eng, sc.records, sc.rec_off, alert_set=sc.batches.recs)
    assert sim.index_info()["alerts_prevalidated"] == 1 and all(np.array_equal(a, b) for a, b in zip(want, res))
    # every alert about one faulty subject withheld from the set
    short = sc.batches.recs[sc.batches.recs["dst"] != sc.faulty[3]]
    sim = E.ClusterSimulation(eng)
    sim.load_streams(sc.records, sc.rec_off)
    sim.set_alert_set(short, trust_copies=True)
    sim.tally()
    with pytest.raises(E.IllegalArgumentException):
        sim.results()
    with pytest.raises(E.IllegalArgumentException):
        sim.count_votes()
    # ... with or without the caller vouching for the deliveries: the index simply was not built for that subject
    sim = E.ClusterSimulation(eng)
    sim.load_streams(sc.records, sc.rec_off)
    sim.set_alert_set(short)
    sim.tally()
    with pytest.raises(E.IllegalArgumentException):
        sim.results()
    # a delivered record with the wrong status for its subject (an UP alert about a member), set otherwise honest
    bad = sc.records.copy()
    k = int(np.flatnonzero(np.isin(bad["dst"], sc.faulty))[5])
    bad["status"][k] = S.UP
    sim = E.ClusterSimulation(eng)
    sim.load_streams(bad, sc.rec_off)
    sim.set_alert_set(sc.batches.recs, trust_copies=True)
    sim.tally()
    with pytest.raises(E.IllegalArgumentException):
        sim.results()
    # the same streams without the promise (index from the set, filter per delivery) and without a declaration: the filter
    # drops that record, as the reference does
    sim, res2b = run_population(E, eng, bad, sc.rec_off, alert_set=sc.batches.recs, trust=False)
    sim, res2 = run_population(E, eng, bad, sc.rec_off)
    assert all(np.array_equal(a, b) for a, b in zip(res2, res2b))
    fe, fn, fo, fpp = O.fast_sim_run(n, K, H, L, cfg, obs, subj, member, bad, sc.rec_off, nthreads=8)
    assert np.array_equal(res2[0], fe) and np.array_equal(res2[2], np.diff(fo))
    # Late deliveries are no breach of the promise: the kernel compares the configuration id of EVERY delivered record on its
    # way through the registers, whatever the instantiation, and drops a record of another configuration as the reference
    # drops it (R/MembershipService.java:653-657) -- same results with the request, without it, without any declaration, and
    # as the oracle fed the same bytes
    late = sc.records.copy()
    k = int(np.flatnonzero(np.isin(late["dst"], sc.faulty))[11])
    late["cfg_id"][k] = cfg + 5
    sim, res_t = run_population(E, eng, late, sc.rec_off, alert_set=sc.batches.recs, trust=True)
    assert sim.index_info()["alerts_prevalidated"] == 1
    sim, res_u = run_population(E, eng, late, sc.rec_off, alert_set=sc.batches.recs, trust=False)
    sim, res_n = run_population(E, eng, late, sc.rec_off)
    fe, fn, fo, fpp = O.fast_sim_run(n, K, H, L, cfg, obs, subj, member, late, sc.rec_off, nthreads=8)
    for res_x in (res_t, res_u, res_n):
        assert np.array_equal(res_x[0], fe) and np.array_equal(res_x[1], fn) and np.array_equal(res_x[2], np.diff(fo))
        assert np.array_equal(res_x[3], proposal_fingerprints(fo, fpp, fe >= 0))
    # clean streams: the request is honoured (and, unrequested, it is not)
    sim, res_c = run_population(E, eng, sc.records, sc.rec_off, alert_set=sc.batches.recs, trust=True)
    assert sim.index_info()["alerts_prevalidated"] == 1 and all(np.array_equal(a, b) for a, b in zip(want, res_c))
    sim, res_c = run_population(E, eng, sc.records, sc.rec_off, alert_set=sc.batches.recs, trust=False)
    assert sim.index_info()["alerts_prevalidated"] == 0 and all(np.array_equal(a, b) for a, b in zip(want, res_c))
    # a declared set stamped with another configuration id: its alerts are not validated (copies of them would be dropped whole per
    # delivery, whatever they say -- late deliveries among a round's batches keep the pre-validated instantiation, round 5); the
    # deliveries here carry the CURRENT id, so they are no copies of it -- the promise is broken, the results are not: what the
    # filter would have dropped is an error in that instantiation, and nothing here is
    stale = sc.batches.recs.copy()
    stale["cfg_id"] = cfg + 1
    sim, res3 = run_population(E, eng, sc.records, sc.rec_off, alert_set=stale)
    assert sim.index_info()["alerts_prevalidated"] == 1 and all(np.array_equal(a, b) for a, b in zip(want, res3))
    sim, res3 = run_population(E, eng, sc.records, sc.rec_off, alert_set=stale, trust=False)
    assert sim.index_info()["alerts_prevalidated"] == 0 and all(np.array_equal(a, b) for a, b in zip(want, res3))


# ------------------------------------------------------------- the dictionary-in-memory instantiations (C4's mode)
@pytest.mark.parametrize("name,n,f,K,H,L", [("C2", 2000, 20, 10, 9, 4), ("C3b", 1500, 40, 10, 9, 4)])
def test_tables_in_memory_mode_vs_faithful_oracle(E, name, n, f, K, H, L):
    """Populations whose plain node -> slot tables do not fit the LDS (N >~ 30,000) run the tally kernel with the
    compressed tables (bitmap + rank) in LDS, and beyond that with the dictionary in memory; from 40,000 nodes on their
    round index is built by several workgroups.  Knobs 128 / 256 / 4096 force these forms at a size the faithful oracle can check: both instantiations (deliveries vouched for / per-delivery filter) of
    both modes must give the oracle's results."""
    pop = S.Population.make(n)
    eng, view = make_engine(E, pop, K, H, L)
    reg, oview = oracle_view(pop, K)
    obs, subj, member = view.tables()
    cfg = view.getCurrentConfigurationId()
    sc = S.build_scenario(name, subj, cfg, n=n, f=f, H=H, L=L)
    rx = np.arange(0, len(sc.receivers), max(1, len(sc.receivers) // 150))
    sub_off = np.zeros(len(rx) + 1, dtype=np.int64)
    parts = []
    for i, r in enumerate(rx):
        parts.append(sc.records[sc.rec_off[r]:sc.rec_off[r + 1]])
        sub_off[i + 1] = sub_off[i] + len(parts[-1])
    oe, on, oo, op = O.sim_run(oview, K, H, L, pop.id_hi, pop.id_lo, np.concatenate(parts), sub_off, nthreads=8)
    want_fp = proposal_fingerprints(oo, op, oe >= 0)
    sim0, ref = run_population(E, eng, sc.records, sc.rec_off)
    assert sim0.index_info()["dict_mode"] == 1  # the product at this size: the 20-byte records looked up in direct tables in LDS
    # where the tally looks a record's subject up -- nothing: the tables where they fit best (direct, in LDS), 128: compressed
    # tables in LDS, 256: tables in memory (through L2); alone: the pre-validated instantiation (every delivered alert passes
    # the filter); | 64: per-delivery filter; | 1: exact path
    # 4096: the round index built by several workgroups (count / assign / adjacency), the form of populations >= 40,000 nodes
    # (such a round never has direct tables)
    # 131072: a declared alert set indexed by the two-kernel form (touch + one workgroup) instead of the one-launch form
    for mode_knob, mode in ((0, 1), (128, 2), (256, 0), (4096, 2), (4096 | 128, 2), (4096 | 256, 0), (131072, 1), (131072 | 128, 2)):
      for kw in (dict(force_exact=mode_knob), dict(force_exact=mode_knob | 64), dict(force_exact=mode_knob, alert_set=sc.batches.recs),
                 dict(force_exact=mode_knob | 64, alert_set=sc.batches.recs), dict(force_exact=mode_knob | 1)):
        sim, res = run_population(E, eng, sc.records, sc.rec_off, **kw)
        info = sim.index_info()
        assert info["dict_mode"] == mode and info["alert_set_declared"] == (1 if "alert_set" in kw else 0)
        assert all(np.array_equal(a, b) for a, b in zip(ref, res)), kw
        assert np.array_equal(res[0][rx], oe) and np.array_equal(res[1][rx], on) and np.array_equal(res[3][rx], want_fp)


def test_c4_shaped_shard_against_fast_oracle(E):
    """BASELINE configs[3] as one rank sees it: N = 100,000, K = 10, 1,000 crashed nodes (1 % churn), a shard of the
    receivers.  The dictionary (2 x 200 KB) stays in memory; results against the optimised CPU formulation, proposal
    contents through the fingerprints, and the vote count over the shard."""
    n, K, H, L = 100000, 10, 9, 4
    pop = S.Population.make(n)
    eng, view = make_engine(E, pop, K, H, L)
    obs, subj, member = view.tables()
    cfg = view.getCurrentConfigurationId()
    sc0 = S.build_scenario("C4", subj, cfg, materialise=False)
    rx = sc0.receivers[:: len(sc0.receivers) // 640][:640]
    sc = S.build_scenario("C4", subj, cfg, receivers=rx)
    fe, fn, fo, fpp = O.fast_sim_run(n, K, H, L, cfg, obs, subj, member, sc.records, sc.rec_off, nthreads=16)
    for kw in (dict(), dict(alert_set=sc.batches.recs)):
        sim, (emit, nprop, pcount, fp) = run_population(E, eng, sc.records, sc.rec_off, **kw)
        assert sim.index_info()["dict_mode"] == 2  # 400 KB of plain tables do not fit the LDS, the compressed form does
        assert np.array_equal(emit, fe) and np.array_equal(nprop, fn) and np.array_equal(pcount, np.diff(fo))
        assert np.array_equal(fp, proposal_fingerprints(fo, fpp, fe >= 0))
    sim, (emit, nprop, pcount, fp) = run_population(E, eng, sc.records, sc.rec_off, force_exact=256)  # ... and from memory
    assert sim.index_info()["dict_mode"] == 0 and np.array_equal(emit, fe) and np.array_equal(fp, proposal_fingerprints(fo, fpp, fe >= 0))
    assert np.all(fe >= 0) and sorted(sim.proposal(0)) == sc.faulty.tolist()
    rr = sim.count_votes()
    assert rr.votes_winner == len(rx) and rr.decided == 0 and rr.quorum == n - (n - 1) // 4  # a shard alone has no quorum


def test_c4_full_shard_against_fast_oracle(E):
    """BASELINE configs[3] exactly as `bench.py --config C4 --gpus 1` runs it: shard 0 of the eight (all 12,375 receivers
    of P.shard_range(99,000, 0, 8), ~122 M delivered records), the round's alerts declared and the per-delivery
    configuration-id check waived on the load pass's verdict.  Every receiver against the optimised CPU formulation
    (announcing batch, getNumProposals, proposal size, proposal contents through the fingerprints), 64 of them also against
    the faithful restatement of the Java; the same streams through the per-delivery filter give the same results; a shard alone
    has no quorum."""
    from rapid_amd import parallel as P
    n, K, H, L = 100000, 10, 9, 4
    pop = S.Population.make(n)
    eng, view = make_engine(E, pop, K, H, L)
    obs, subj, member = view.tables()
    cfg = view.getCurrentConfigurationId()
    sc0 = S.build_scenario("C4", subj, cfg, materialise=False)
    lo, hi = P.shard_range(len(sc0.receivers), 0, 8)
    rx = sc0.receivers[lo:hi]
    assert len(rx) == 12375
    records, rec_off, nb = S.deliver(sc0.batches, rx, seed_delivery=2)  # bench.py's delivery seed
    fe, fn, fo, fpp = O.fast_sim_run(n, K, H, L, cfg, obs, subj, member, records, rec_off, nthreads=64)
    want_fp = proposal_fingerprints(fo, fpp, fe >= 0)
    sim, (emit, nprop, pcount, fp) = run_population(E, eng, records, rec_off, alert_set=sc0.batches.recs, trust=True)
    info = sim.index_info()
    assert info["alerts_prevalidated"] == 1 and info["dict_mode"] == 2
    assert np.array_equal(emit, fe) and np.array_equal(nprop, fn) and np.array_equal(pcount, np.diff(fo))
    assert np.array_equal(fp, want_fp)
    assert np.all(fe >= 0) and np.all(pcount == len(sc0.faulty))
    for r in (0, 6000, len(rx) - 1):
        assert sorted(sim.proposal(r)) == sc0.faulty.tolist()
    # ... and the faithful restatement of the Java at this size, on the same streams: 64 receivers spread over the shard (the
    # faithful detector walks every subject in flux at every batch end, R/MultiNodeCutDetector.java:137-164 -- ~6 s per receiver
    # here, which is why the other 12,311 are held against the optimised formulation only)
    sample = np.linspace(0, len(rx) - 1, 64).astype(np.int64)
    sub_off = np.zeros(len(sample) + 1, dtype=np.int64)
    parts = []
    for i, r in enumerate(sample):
        parts.append(records[rec_off[r]:rec_off[r + 1]])
        sub_off[i + 1] = sub_off[i] + len(parts[-1])
    reg, oview = oracle_view(pop, K)
    assert oview.getCurrentConfigurationId() == cfg
    oe, on, oo, op = O.sim_run(oview, K, H, L, pop.id_hi, pop.id_lo, np.concatenate(parts), sub_off, nthreads=32)
    assert np.array_equal(emit[sample], oe) and np.array_equal(nprop[sample], on) and np.array_equal(pcount[sample], np.diff(oo))
    assert np.array_equal(fp[sample], proposal_fingerprints(oo, op, oe >= 0))
    del reg, oview
    rr = sim.count_votes()
    assert rr.votes_winner == len(rx) and rr.decided == 0 and rr.quorum == n - (n - 1) // 4
    sim, res_f = run_population(E, eng, records, rec_off, alert_set=sc0.batches.recs, trust=False)
    assert sim.index_info()["alerts_prevalidated"] == 0
    assert all(np.array_equal(a_, b_) for a_, b_ in zip((emit, nprop, pcount, fp), res_f))
    eng.close()


def test_configuration_ids_left_in_the_cache_lines_only_when_known_current(E, test_build):
    """Level 2 of rapid_sim_trust_alert_copies and the instantiation behind it (kCurrent: one load per boundary record, its
    configuration id never read).  Over the four dictionary forms (direct / compressed tables in LDS, tables in memory, packed
    state): level 2 gives level 1's results and the oracle's on streams of one configuration, and reports itself in the index
    info; deliveries the library lays down itself take that path without anybody's word; a declared alert of another configuration
    switches it off; and level 1 -- verified facts only -- drops a late delivery of the previous configuration per delivery
    (R/MembershipService.java:653-657), where level 2 would have been the caller's mistake."""
    K, H, L = 10, 9, 4
    n = 3000
    pop = S.Population.make(n)
    eng, view = make_engine(E, pop, K, H, L)
    obs, subj, member = view.tables()
    cfg = view.getCurrentConfigurationId()
    sc = S.build_churn_scenario(obs, member, cfg, 30, 0, H, L, materialise=False)
    rx = sc.receivers[:600]
    records, rec_off, _ = S.deliver(sc.batches, rx, 5)
    fe, fn, fo, fpp = O.fast_sim_run(n, K, H, L, cfg, obs, subj, member, records, rec_off, nthreads=8)
    want = (fe, fn, np.diff(fo), proposal_fingerprints(fo, fpp, fe >= 0))
    sim = E.ClusterSimulation(eng)
    for knob, mode in ((0, 1), (128, 2), (256, 0), (8192, 0)):  # direct tables, compressed tables, tables in memory, packed state
        for level in (1, 2):
            sim.set_force_exact(knob)
            sim.load_streams(records, rec_off)
            sim.set_alert_set(sc.batches.recs, trust_copies=level)
            sim.tally()
            info = sim.index_info()
            assert (info["alerts_prevalidated"], info["dict_mode"], info["configuration_ids_known_current"]) == (1, mode, int(level == 2)), (knob, level, info)
            assert all(np.array_equal(a, b) for a, b in zip(want, sim.results())), (knob, level)
    sim.set_force_exact(4194304)  # (knob: level 2 asked for, ids compared all the same)
    sim.load_streams(records, rec_off)
    sim.set_alert_set(sc.batches.recs, trust_copies=2)
    sim.tally()
    assert sim.index_info()["configuration_ids_known_current"] == 0 and all(np.array_equal(a, b) for a, b in zip(want, sim.results()))
    sim.set_force_exact(0)
    # the library's own deliveries (boundary records): known current without anybody saying so
    sim.generate(sc.batches, rx, 11, boundary=True, trust_copies=True)
    sim.tally()
    assert sim.index_info()["configuration_ids_known_current"] == 1
    got = [a.copy() for a in sim.results()]
    recs_g, off_g, _ = S.deliver_hashed(sc.batches, rx, 11)
    ge, gn, go, gpp = O.fast_sim_run(n, K, H, L, cfg, obs, subj, member, recs_g, off_g, nthreads=8)
    assert np.array_equal(got[0], ge) and np.array_equal(got[1], gn) and np.array_equal(got[2], np.diff(go))
    # ... unless the set it copies from holds alerts of another configuration (late batches): ids compared, copies dropped
    both = S.with_late_batches(sc.batches, cfg - 1, 0.2, 5)
    sim.generate(both, rx, 11, boundary=True, trust_copies=True)
    sim.tally()
    assert sim.index_info()["configuration_ids_known_current"] == 0
    recs_l, off_l, _ = S.deliver_hashed(both, rx, 11)
    le, ln, lo, lpp = O.fast_sim_run(n, K, H, L, cfg, obs, subj, member, recs_l, off_l, nthreads=8)
    res = sim.results()
    assert np.array_equal(res[0], le) and np.array_equal(res[1], ln) and np.array_equal(res[2], np.diff(lo))
    # ... and what the library knew about its own deliveries it knows about the set it made them FROM: declaring another set over the
    # same generated streams -- the current alerts only -- at level 1 must not let the late copies pass as current (ADVICE round 5:
    # streams_generated used to survive the declaration, and the ids were then not read at all)
    sim.set_alert_set(sc.batches.recs, trust_copies=1)
    sim.new_round()
    sim.tally()
    assert sim.index_info()["configuration_ids_known_current"] == 0
    res = sim.results()
    assert np.array_equal(res[0], le) and np.array_equal(res[1], ln) and np.array_equal(res[2], np.diff(lo))
    # a late delivery among host-delivered records: level 1 drops it, as the oracle (and the reference) does
    late = records.copy()
    first = int(rec_off[3])  # the first record of receiver 3: a DOWN report, now of the previous configuration
    late["cfg_id"][first] = cfg - 1
    oe, on, oo, opp = O.fast_sim_run(n, K, H, L, cfg, obs, subj, member, late, rec_off, nthreads=8)
    sim.load_streams(late, rec_off)
    sim.set_alert_set(sc.batches.recs, trust_copies=1)
    sim.tally()
    res = sim.results()
    assert sim.index_info()["alerts_prevalidated"] == 1 and sim.index_info()["configuration_ids_known_current"] == 0
    assert np.array_equal(res[0], oe) and np.array_equal(res[1], on) and np.array_equal(res[2], np.diff(oo))
    with pytest.raises(E.IllegalArgumentException):
        eng._check(eng._lib.rapid_sim_trust_alert_copies(eng._h, 3))
    eng.close()


# ------------------------------------------------------------------ C5: streaming rounds, stale records, quirk Q4
def _oracle_decide(oview, K, H, L, pop, sc, cut):
    """decideViewChange on the oracle: it learns the joiners' NodeIds from the UP alerts (only the batches that carry one
    are replayed: the faithful detector is quadratic in the number of subjects in flux)."""
    svc = O.AlertBatchService(oview, K, H, L, pop.id_hi, pop.id_lo)
    r0 = sc.records[sc.rec_off[0]:sc.rec_off[1]]
    beg = 0
    for e in np.flatnonzero(r0["flags"] & S.FLAG_LAST_IN_BATCH) + 1:
        b = r0[beg:int(e)]
        if np.any((b["status"] == S.UP) & (b["cfg_id"] == oview.getCurrentConfigurationId())):
            svc.handleBatchedAlertMessage(b)
        beg = int(e)
    svc.decideViewChange(cut)


def test_streaming_rounds_with_stale_records_and_the_observer_cache(E):
    """BASELINE configs[4] (continuous churn) at a size the faithful oracle can follow: six consecutive rounds over one
    population -- a fresh 1 % of the members crashes and 0.5 % joins in every round, 1 % of the delivered records still carry
    the previous configuration id (R/MembershipService.java:653-657), the decided cut is applied
    (R/MembershipService.java:385-430) and the next round runs in the new configuration.  Every round: the engine's
    per-receiver results equal the optimised oracle on all receivers and the FAITHFUL oracle -- whose view keeps its
    observer cache across the rounds, quirk Q4 -- on a sample; the new configuration id and the observer table equal the
    oracle's; and the Q4 guard shows that no receiver can hold a stale cache entry for a subject in flux."""
    K, H, L = 10, 9, 4
    n_mem, spare, rounds = 10000, 400, 6
    pop = S.Population.make(n_mem + spare)
    members = list(range(n_mem))
    eng, view = make_engine(E, pop, K, H, L, members=members)
    reg, oview = oracle_view(pop, K, members)
    st = S.StreamingChurn(H, L, receivers_per_round=300)
    guard = E.ObserverCacheGuard()
    sim = E.ClusterSimulation(eng)
    n = pop.n
    for rnd in range(rounds):
        obs, subj, member = view.tables()
        cfg = view.getCurrentConfigurationId()
        assert cfg == oview.getCurrentConfigurationId()
        sc = st.next_round(obs, member, cfg)
        if rnd > 0:
            assert 0 < int((sc.records["cfg_id"] != cfg).sum()) < len(sc.records) // 50  # the stale records are there
        # Q4: no member that is in flux now was in flux (= queried, cached) in an earlier configuration with other observers
        assert guard.check_round(view, sc.faulty) == []
        for s_ in sc.crashed[:: max(1, len(sc.crashed) // 25)]:
            assert oview.getObserversOf(int(s_)) == oview.computeObserversOf(int(s_)) == view.getObserversOf(int(s_))
        sim.load_streams(sc.records, sc.rec_off)
        sim.set_alert_set(sc.batches.recs)  # the index from the round's alerts; the deliveries are NOT vouched for (stale records)
        sim.tally()
        emit, nprop, pcount, fp = sim.results()
        fe, fn, fo, fpp = O.fast_sim_run(n, K, H, L, cfg, obs, subj, member, sc.records, sc.rec_off, nthreads=16)
        assert np.array_equal(emit, fe) and np.array_equal(nprop, fn) and np.array_equal(pcount, np.diff(fo))
        assert np.array_equal(fp, proposal_fingerprints(fo, fpp, fe >= 0))
        m = 40  # the faithful restatement, observer cache carried over from the earlier rounds (single-threaded)
        oe, on, oo, op = O.sim_run(oview, K, H, L, pop.id_hi, pop.id_lo, sc.records[: sc.rec_off[m]], sc.rec_off[: m + 1],
                                   prewarm_observers=False)
        assert np.array_equal(emit[:m], oe) and np.array_equal(nprop[:m], on)
        assert np.array_equal(fp[:m], proposal_fingerprints(oo, op, oe >= 0))
        first = int(np.flatnonzero(fe >= 0)[0])
        cut = sorted(fpp[fo[first]:fo[first + 1]].tolist())
        assert cut == sc.faulty.tolist()
        new_cfg = sim.apply_cut(cut)
        _oracle_decide(oview, K, H, L, pop, sc, cut)
        guard.on_view_change(sc.crashed)
        assert new_cfg == oview.getCurrentConfigurationId()
        assert view.getMembershipSize() == oview.getMembershipSize()
    o2, s2, m2 = view.tables()
    oo2, os2, om2 = oview.tables(n)
    assert np.array_equal(m2, om2) and np.array_equal(s2, os2) and np.array_equal(o2, oo2)


def test_streams_generated_on_the_device(E, test_build):
    """rapid_sim_generate: the round's deliveries made on the device -- every receiver gets every batch once, in a
    seeded permutation of its own evaluated in place (no keys, no sort).  Against the host statement scenarios.deliver_hashed:
    the 20-byte boundary records are equal byte for byte; the resolved 8-byte records hold the subjects' dictionary entries and
    the core words; the tally over either equals the tally over the host's records loaded through the boundary, and the
    oracle fed those records; the round decides the same cut.  Per-batch delivery thresholds: late batches of the previous
    configuration reach some receivers only, their places are empty records elsewhere."""
    from tests.emu import pyemu
    K, H, L = 10, 9, 4
    for n, n_crash, n_join, seed in ((2000, 20, 0, 2), (1500, 30, 12, 977)):
        pop = S.Population.make(n + 40)
        eng, view = make_engine(E, pop, K, H, L, members=list(range(n)))
        obs, subj, member = view.tables()
        cfg = view.getCurrentConfigurationId()
        sc = S.build_churn_scenario(obs, member, cfg, n_crash, n_join, H, L, materialise=False)
        rx = sc.receivers
        want, want_off, nb = S.deliver_hashed(sc.batches, rx, seed)
        A = int(sc.batches.off[-1])
        fe, fn, fo, fpp = O.fast_sim_run(pop.n, K, H, L, cfg, obs, subj, member, want, want_off, nthreads=16)
        sim2, ref = run_population(E, eng, want, want_off, alert_set=sc.batches.recs)
        rr2 = sim2.count_votes()
        assert np.array_equal(ref[0], fe) and np.array_equal(ref[1], fn) and np.array_equal(ref[2], np.diff(fo))
        assert np.array_equal(ref[3], proposal_fingerprints(fo, fpp, fe >= 0))
        ix = pyemu.build_round_index(sc.batches.recs, pop.n, K, L, np.asarray(obs), member)
        entries = pyemu.dict_entries(ix, pop.n)
        sim = E.ClusterSimulation(eng)
        for boundary in (True, False):
            sim.generate(sc.batches, rx, seed, boundary=boundary)
            info = sim.index_info()
            assert info["dict_mode"] == (1 if boundary else 3) and info["alert_set_declared"] == 1
            for first in (0, A - 7, (len(rx) // 2) * A + 3, len(want) - 1000):
                first = max(0, first)
                m = min(1000, len(want) - first)
                dst, words = sim.read_records(first, m)
                seg = want[first:first + m]
                assert np.array_equal(dst, seg["dst"] if boundary else entries[seg["dst"]]), boundary
                assert np.array_equal(words, (seg["ring_mask"].astype(np.uint32) & 0x3FFF) | np.where(seg["status"] != 0, 1 << 14, 1 << 15).astype(np.uint32)
                                      | ((seg["flags"].astype(np.uint32) & 1) << 16))
            for knob in (0, 64):  # the per-delivery filter, too
                sim.set_force_exact(knob)
                sim.new_round()
                sim.tally()
                assert all(np.array_equal(a, b) for a, b in zip(ref, sim.results())), (boundary, knob)
            sim.set_force_exact(0)
            rr = sim.count_votes()
            assert (rr.decided, rr.votes_winner, rr.cut_size) == (rr2.decided, rr2.votes_winner, rr2.cut_size) and rr.decided == 1
            assert sim.decided_cut() == sim2.decided_cut() and sorted(sim.decided_cut()) == sc.faulty.tolist()
        with pytest.raises(E.RapidError):
            sim.set_alert_set(sc.batches.recs)  # the set the deliveries were generated from IS the declared one
        # late deliveries: a copy of a fifth of the batches, carrying the previous configuration id, each reaching about a tenth
        # of the receivers; nothing of the round itself is lost, so every receiver's outcome is the one above -- except the
        # number of batches it had seen when it announced
        nb_ = sc.batches.n_batches
        late = np.arange(0, nb_, 5)
        lrecs = np.concatenate([sc.batches.recs[sc.batches.off[b]:sc.batches.off[b + 1]] for b in late])
        lrecs["cfg_id"] = cfg - 1
        loff = np.cumsum([0] + [int(sc.batches.off[b + 1] - sc.batches.off[b]) for b in late])
        both = S.BatchSet(np.concatenate([sc.batches.recs, lrecs]), np.concatenate([sc.batches.off, sc.batches.off[-1] + loff[1:]]).astype(np.int64),
                          np.concatenate([sc.batches.sender, sc.batches.sender[late]]))
        keep = np.concatenate([np.full(nb_, 0xFFFFFFFF, dtype=np.uint32), np.full(len(late), 0xFFFFFFFF // 10, dtype=np.uint32)])
        want_l, want_l_off, nb_l = S.deliver_hashed(both, rx, seed + 1, keep=keep)
        assert nb_ <= nb_l.min() and nb_l.max() < nb_ + len(late) and abs(nb_l.mean() - nb_ - len(late) / 10) < 0.03 * len(late)
        fe_l, fn_l, fo_l, fpp_l = O.fast_sim_run(pop.n, K, H, L, cfg, obs, subj, member, want_l, want_l_off, nthreads=16)
        assert np.array_equal(np.diff(fo_l), np.diff(fo)) and np.array_equal(fpp_l, fpp)
        for boundary in (True, False):
            sim.generate(both, rx, seed + 1, keep=keep, boundary=boundary)
            sim.tally()
            got = sim.results()
            assert np.array_equal(got[0], fe_l) and np.array_equal(got[1], fn_l) and np.array_equal(got[2], np.diff(fo_l))
            assert np.array_equal(got[3], proposal_fingerprints(fo_l, fpp_l, fe_l >= 0))
            if boundary:
                dst, words = sim.read_records(0, 3 * int(both.off[-1]))
                assert np.array_equal(dst, want_l["dst"][: len(dst)])
        # a view change: resolved deliveries belong to the view they were generated in
        rr = sim.count_votes()
        sim.apply_cut(sim.decided_cut())
        with pytest.raises(E.RapidError):
            sim.tally()
        with pytest.raises(E.IllegalArgumentException):  # a BatchedAlertMessage is never empty
            sim.generate(S.BatchSet(sc.batches.recs, np.concatenate([sc.batches.off[:1], sc.batches.off]), np.concatenate([[0], sc.batches.sender])), rx, 1)
        eng.close()


def test_generated_streams_at_full_size(E, test_build):
    """rapid_sim_generate at BASELINE configs[2]'s full size (N = 10,000, C3b: 9,487 receivers x 4,343 batches = 93.7 M
    records): EVERY receiver's outcome against the optimised CPU formulation fed scenarios.deliver_hashed's records, for the
    resolved and the boundary form; the round decides the closed fault set."""
    n, K, H, L = 10000, 10, 9, 4
    pop = S.Population.make(n)
    eng, view = make_engine(E, pop, K, H, L)
    obs, subj, member = view.tables()
    cfg = view.getCurrentConfigurationId()
    sc = S.build_scenario("C3b", subj, cfg, materialise=False)
    want, want_off, nb = S.deliver_hashed(sc.batches, sc.receivers, 2)
    fe, fn, fo, fpp = O.fast_sim_run(n, K, H, L, cfg, obs, subj, member, want, want_off, nthreads=64)
    want_fp = proposal_fingerprints(fo, fpp, fe >= 0)
    sim = E.ClusterSimulation(eng)
    for boundary in (False, True):
        sim.generate(sc.batches, sc.receivers, 2, trust_copies=True, boundary=boundary)
        assert sim.index_info()["alerts_prevalidated"] == 1
        sim.tally()
        emit, nprop, pcount, fp = sim.results()
        assert np.array_equal(emit, fe) and np.array_equal(nprop, fn) and np.array_equal(pcount, np.diff(fo)) and np.array_equal(fp, want_fp)
        rr = sim.count_votes()
        assert rr.decided == 1 and sorted(sim.decided_cut()) == sc.faulty.tolist()
        for r in (0, len(fe) // 2, len(fe) - 1):
            A = int(sc.batches.off[-1])
            dst, words = sim.read_records(r * A, 2000)
            if boundary:
                assert np.array_equal(dst, want["dst"][r * A: r * A + 2000])
    eng.close()


def test_q4_stale_observer_cache_is_reproduced_and_reported(E):
    """The reference's memoised getObserversOf, kept on the device, and rapid_view_q4_at_risk against the oracle's faithful cachedObservers (R/MembershipView.java:143-152, 181-195, 210-224).
    A subject that was hot once (its observers memoised) and stays in the view: (1) an unrelated removal changes nothing;
    (2) removing its successor on a ring drops its entry in the Java as well (lower(successor) == subject): no risk, and the
    entry is re-created from today's observers; (3) removing the ring MINIMUM while the subject is that ring's MAXIMUM changes
    the subject's observer on that ring but not its memoised list -- the quirk: the Java now reads a stale list, and the
    call names the subject."""
    K, H, L = 10, 9, 4
    n = 300
    pop = S.Population.make(n)
    eng, view = make_engine(E, pop, K, H, L)
    reg, oview = oracle_view(pop, K)
    guard = E.ObserverCacheGuard()
    sim = E.ClusterSimulation(eng)
    ring0 = view.getRing(0)
    x = int(ring0[-1])                         # the maximum of ring 0: its ring-0 observer is the minimum (wrap-around)
    assert view.getObserversOf(x)[0] == int(ring0[0])
    assert guard.check_round(view, [x]) == [] and oview.getObserversOf(x) == view.getObserversOf(x)   # memoised on both sides

    def cut(nodes):
        sim.apply_cut(nodes)
        for v in nodes:
            oview.ringDelete(int(v))

    # (1) somebody nowhere near x
    obs_x = set(view.getObserversOf(x)) | set(view.getSubjectsOf(x)) | {x, int(ring0[0])}
    far = next(int(v) for v in ring0[5:] if int(v) not in obs_x and x not in view.getObserversOf(int(v)) + view.getSubjectsOf(int(v)))
    cut([far])
    assert oview.getObserversOf(x) == oview.computeObserversOf(x) == view.getObserversOf(x)
    assert guard.check_round(view, [x]) == []
    # (2) x's successor on ring 1 (x is not the maximum of ring 1, or pick another ring where it is not)
    k = next(k for k in range(1, K) if int(view.getRing(k)[-1]) != x)
    succ = view.getObserversOf(x)[k]
    before = view.getObserversOf(x)
    cut([succ])
    assert view.getObserversOf(x) != before
    assert oview.getObserversOf(x) == oview.computeObserversOf(x) == view.getObserversOf(x)    # the Java dropped the entry, too
    assert guard.check_round(view, [x]) == []
    # (3) the minimum of ring 0 goes; x stays the maximum
    ring0 = view.getRing(0)
    assert int(ring0[-1]) == x
    m0 = int(ring0[0])
    if x in [int(view.getRing(kk)[max(0, list(view.getRing(kk)).index(m0) - 1)]) for kk in range(1, K) if list(view.getRing(kk)).index(m0) > 0]:
        pytest.skip("the ring-0 minimum is also x's direct successor elsewhere: the entry is dropped legitimately")
    cut([m0])
    assert view.getObserversOf(x)[0] == int(view.getRing(0)[0]) != m0
    assert oview.getObserversOf(x) != oview.computeObserversOf(x) == view.getObserversOf(x)    # the reference reads the stale list
    assert guard.check_round(view, [x, int(ring0[1])]) == [x]
    # ... and the quirk DECIDES a round, on the device as in the reference: m0 (gone) is reported UP on every ring, x DOWN on
    # eight rings other than ring 0; invalidateFailingEdges finds m0 as x's ring-0 observer in the stale memo, credits x with
    # the implicit report, and every receiver proposes {x, m0}.  Against the faithful oracle carrying its cache; with the memo
    # switched off (today's observers) x stays one report short of H and nobody proposes.
    from tests.test_kernel_emulated import _q4_round
    cfg = view.getCurrentConfigurationId()
    assert cfg == oview.getCurrentConfigurationId()
    alerts, records, rec_off, _, _, _ = _q4_round(oview, x, m0, n, K, cfg, n_receivers=40)
    oe, on, oo, op = O.sim_run(oview, K, H, L, pop.id_hi, pop.id_lo, records, rec_off, prewarm_observers=False)
    assert np.all(oe >= 0)
    for declared in (False, True):
        sim.load_streams(records, rec_off)
        if declared:
            sim.set_alert_set(alerts, trust_copies=True)
        sim.tally()
        emit, nprop, pcount, fp = sim.results()
        assert sim.index_info()["q4_live"] == 1
        assert np.array_equal(emit, oe) and np.array_equal(nprop, on) and np.array_equal(pcount, np.diff(oo))
        assert np.array_equal(fp, proposal_fingerprints(oo, op, oe >= 0)) and sorted(sim.proposal(0)) == sorted([x, m0])
    view.setObserverCacheEmulation(False)
    sim.new_round()
    sim.tally()
    emit, nprop, pcount, fp = sim.results()
    # (x, one report short of H, is never proposed and blocks the proposal -- except at a receiver that has m0 complete before
    # x reaches L, which proposes m0 alone with or without the memo)
    assert sim.index_info()["q4_live"] == 0 and np.all(pcount <= 1) and np.array_equal(emit >= 0, np.diff(oo) == 1)
    assert int((emit >= 0).sum()) < len(emit) // 4 and all(sim.proposal(int(r)) == [m0] for r in np.flatnonzero(emit >= 0))
    view.setObserverCacheEmulation(True)
    sim.new_round()
    sim.tally()
    assert np.array_equal(sim.results()[0], oe)
    # ... until x itself leaves the view: its entry dies with it
    cut([x])
    assert guard.check_round(view, [x]) == []


def test_streaming_rounds_at_100k_nodes(E):
    """The same stream at N = 100,000 (a single-GPU-sized slice of BASELINE configs[4]): three rounds, 1,000 crashes + 500 joins
    each, against the optimised oracle on every simulated receiver -- and, in the second round (late deliveries, a view change
    behind it), three receivers against the faithful restatement of the Java; configuration ids against the oracle's view."""
    K, H, L = 10, 9, 4
    n_mem, spare, rounds = 100000, 2000, 3
    pop = S.Population.make(n_mem + spare)
    members = list(range(n_mem))
    eng, view = make_engine(E, pop, K, H, L, members=members)
    reg, oview = oracle_view(pop, K, members)
    st = S.StreamingChurn(H, L, receivers_per_round=192)
    sim = E.ClusterSimulation(eng)
    for rnd in range(rounds):
        obs, subj, member = view.tables()
        cfg = view.getCurrentConfigurationId()
        assert cfg == oview.getCurrentConfigurationId()
        sc = st.next_round(obs, member, cfg)
        sim.load_streams(sc.records, sc.rec_off)
        sim.set_alert_set(sc.batches.recs)  # the index from the round's alerts; the deliveries are NOT vouched for (stale records)
        sim.tally()
        emit, nprop, pcount, fp = sim.results()
        fe, fn, fo, fpp = O.fast_sim_run(pop.n, K, H, L, cfg, obs, subj, member, sc.records, sc.rec_off, nthreads=16)
        assert np.array_equal(emit, fe) and np.array_equal(nprop, fn) and np.array_equal(pcount, np.diff(fo))
        assert np.array_equal(fp, proposal_fingerprints(fo, fpp, fe >= 0))
        # the late deliveries of the previous configuration are dropped, nothing of this round is lost to them: every receiver
        # announces, and announces the whole fault set
        assert (fe >= 0).all()
        for r_ in np.flatnonzero(fe >= 0)[:5]:
            assert sorted(fpp[fo[r_]:fo[r_ + 1]].tolist()) == sc.faulty.tolist()
        if rnd == 1:
            # the second round -- late deliveries among the records, a view change behind it -- also through the FAITHFUL
            # restatement of the Java, on the oracle's own view (its observer cache as the first view change left it): three
            # receivers, ~50 s each (the faithful detector walks the ~1,500 subjects in flux at each of ~14,000 batch ends; at
            # 10^6 nodes that is ~1.5 h per receiver, which is why the rounds at 10^6 are held against the optimised one only)
            m = 3
            oe, on, oo, op = O.sim_run(oview, K, H, L, pop.id_hi, pop.id_lo, sc.records[: sc.rec_off[m]], sc.rec_off[: m + 1],
                                       prewarm_observers=False)
            assert np.array_equal(emit[:m], oe) and np.array_equal(nprop[:m], on) and np.array_equal(pcount[:m], np.diff(oo))
            assert np.array_equal(fp[:m], proposal_fingerprints(oo, op, oe >= 0))
        cut = sc.faulty.tolist()
        new_cfg = sim.apply_cut(cut)
        _oracle_decide(oview, K, H, L, pop, sc, cut)
        assert new_cfg == oview.getCurrentConfigurationId() and view.getMembershipSize() == oview.getMembershipSize()
        for s_ in (int(sc.joiners[0]), int(sc.receivers[0]), int(sc.receivers[-1])):
            assert view.getObserversOf(s_) == oview.computeObserversOf(s_)


def test_streaming_rounds_at_one_million_nodes(E):
    """BASELINE configs[4] at its own population: N = 1,000,000 members, K = 10, continuous churn -- every round 10,000
    members crash and 5,000 nodes join (15,000 subjects in flux), 1 % of the delivered records are late deliveries of the
    previous configuration (R/MembershipService.java:653-657), the round's cut is applied (R/MembershipService.java:385-430)
    and the next round runs in the new configuration.  Two consecutive rounds, 256 simulated receivers per round (one GPU's
    sample of the population: ~150,000 delivered records each): EVERY receiver against the optimised oracle (announcing
    batch, getNumProposals, proposal size and contents), and after each view change the configuration id, the membership
    size and sampled observer lists against the oracle's MembershipView, whose rings went through the same
    ringDelete / ringAdd sequence (R/MembershipView.java:123-201)."""
    K, H, L = 10, 9, 4
    n_mem, spare, rounds, n_rx = 1000000, 12000, 2, 256
    pop = S.Population.make(n_mem + spare)
    members = list(range(n_mem))
    eng, view = make_engine(E, pop, K, H, L, members=members, max_cut=16384)
    reg, oview = oracle_view(pop, K, members)
    st = S.StreamingChurn(H, L, receivers_per_round=n_rx)
    sim = E.ClusterSimulation(eng)
    guard = E.ObserverCacheGuard()
    for rnd in range(rounds):
        obs, subj, member = view.tables()
        cfg = view.getCurrentConfigurationId()
        assert cfg == oview.getCurrentConfigurationId()
        sc = st.next_round(obs, member, cfg)
        assert len(sc.receivers) == n_rx and len(sc.crashed) >= 9900 and len(sc.joiners) >= 4900
        if rnd > 0:
            assert 0 < int((sc.records["cfg_id"] != cfg).sum()) < len(sc.records) // 50  # the late deliveries are there
        assert guard.check_round(view, sc.faulty[:: 50]) == []  # Q4 (sampled: the guard is host-side bookkeeping)
        sim.load_streams(sc.records, sc.rec_off)
        sim.set_alert_set(sc.batches.recs, trust_copies=True)  # (late deliveries of the previous configuration are dropped per delivery either way)
        sim.tally()
        assert sim.index_info()["alerts_prevalidated"] == 1
        emit, nprop, pcount, fp = sim.results()
        fe, fn, fo, fpp = O.fast_sim_run(pop.n, K, H, L, cfg, obs, subj, member, sc.records, sc.rec_off, nthreads=64)
        assert np.array_equal(emit, fe) and np.array_equal(nprop, fn) and np.array_equal(pcount, np.diff(fo))
        assert np.array_equal(fp, proposal_fingerprints(fo, fpp, fe >= 0))
        if rnd == 0:  # nothing is lost in the first round: every receiver announces the whole fault set, and they agree
            assert np.all(fe >= 0) and np.all(pcount == len(sc.faulty))
            assert sorted(sim.proposal(0)) == sc.faulty.tolist()
            rr = sim.count_votes()
            assert rr.votes_winner == n_rx and rr.cut_size == len(sc.faulty)
        for r_ in np.flatnonzero(fe >= 0)[:3]:
            assert sorted(fpp[fo[r_]:fo[r_ + 1]].tolist()) == sc.faulty.tolist()
        # the cut the round settles on, applied in the order FastPaxos hands it over (ring-0 order of the decided list is
        # irrelevant to the resulting view: deletes and adds commute on the sorted sets)
        cut = sc.faulty.tolist()
        new_cfg = sim.apply_cut(cut)
        is_member = member != 0
        for node in cut:  # decideViewChange on the oracle's view (R/MembershipService.java:394-413), joiner ids from the registry
            if is_member[node]:
                oview.ringDelete(int(node))
            else:
                oview.ringAdd(int(node), (int(pop.id_hi[node]), int(pop.id_lo[node])))
        guard.on_view_change(sc.crashed)
        assert new_cfg == oview.getCurrentConfigurationId() == view.getCurrentConfigurationId()
        assert view.getMembershipSize() == oview.getMembershipSize() == int(is_member.sum()) - len(sc.crashed) + len(sc.joiners)
        for s_ in (int(sc.joiners[0]), int(sc.joiners[-1]), int(sc.receivers[0]), int(sc.receivers[-1])):
            assert view.getObserversOf(s_) == oview.computeObserversOf(s_)
            assert view.getSubjectsOf(s_) == oview.getSubjectsOf(s_)
    eng.close()


def test_hashed_dictionary_of_packed_rounds(E):
    """Rounds with thousands of hot subjects keep the node -> slot dictionary in LDS as hashed buckets of one-byte remainders
    (tally_kernel.h: kDictHashed; index_hash_kernel renumbers the hot subjects in hash order) next to the packed detector state.
    Forced at a small population (testing knobs 8192: packed whatever the size, 1048576: the hashed dictionary): every receiver's outcome, the proposals in
    ring-0 order, the decision and the cut equal the default instantiation's and the oracle's -- vouched-for copies and the
    per-delivery filter, late deliveries of another configuration among the records, joiners and crashed members in one cut; the
    same with the packed state and the dictionary in memory (the default of such rounds); a report of the current configuration
    about a node the alert set does not name is an error of the stream, not a silently different tally."""
    K, H, L = 10, 9, 4
    n, n_out = 3000, 60
    pop = S.Population.make(n + n_out)
    eng, view = make_engine(E, pop, K, H, L, members=list(range(n)))
    obs, subj, member = view.tables()
    cfg = view.getCurrentConfigurationId()
    sc = S.build_churn_scenario(obs, member, cfg, 40, 25, H, L)
    sc = S.build_churn_scenario(obs, member, cfg, 40, 25, H, L, receivers=sc.receivers[::3])
    late = sc.records.copy()
    late["cfg_id"][::9] = cfg - 7  # late deliveries: dropped per delivery, whoever they name
    late["dst"][::27] = np.arange(len(late["dst"][::27])) % pop.n
    for records in (sc.records, late):
        fe, fn, fo, fpp = O.fast_sim_run(pop.n, K, H, L, cfg, obs, subj, member, records, sc.rec_off, nthreads=8)
        sim0, ref = run_population(E, eng, records, sc.rec_off, alert_set=sc.batches.recs)
        assert sim0.index_info()["dict_mode"] == 1
        assert np.array_equal(ref[0], fe) and np.array_equal(ref[1], fn) and np.array_equal(ref[2], np.diff(fo))
        assert np.array_equal(ref[3], proposal_fingerprints(fo, fpp, fe >= 0))
        rr0 = sim0.count_votes()
        cut0 = sim0.decided_cut() if rr0.decided else None
        props0 = [sim0.proposal(r) for r in range(0, len(fe), 37)]
        HK = 8192 | 1048576
        for knob, mode, trust in ((HK, 4, True), (HK | 64, 4, True), (HK, 4, False), (8192, 0, True)):
            sim, res = run_population(E, eng, records, sc.rec_off, force_exact=knob, alert_set=sc.batches.recs, trust=trust)
            info = sim.index_info()
            assert info["dict_mode"] == mode and info["alerts_prevalidated"] == (1 if trust and not (knob & 64) else 0), (knob, info)
            assert all(np.array_equal(a, b) for a, b in zip(ref, res)), (knob, trust)
            assert [sim.proposal(r) for r in range(0, len(fe), 37)] == props0
            rr = sim.count_votes()
            assert (rr.decided, rr.votes_winner, rr.votes_total, rr.cut_size) == (rr0.decided, rr0.votes_winner, rr0.votes_total, rr0.cut_size)
            assert (sim.decided_cut() if rr.decided else None) == cut0
            sim.set_force_exact(0)
    stray = sc.records.copy()
    stray["dst"][11] = int(np.setdiff1d(np.flatnonzero(member != 0), sc.batches.recs["dst"])[3])
    stray["status"][11] = 1
    for trust in (True, False):
        sim, _ = None, None
        sim = E.ClusterSimulation(eng)
        sim.set_force_exact(8192 | 1048576)
        sim.load_streams(stray, sc.rec_off)
        sim.set_alert_set(sc.batches.recs, trust_copies=trust)
        sim.tally()
        with pytest.raises(E.IllegalArgumentException):
            sim.results()
        sim.set_force_exact(0)
    eng.close()


def test_full_population_rounds_at_one_million_nodes(E):
    """BASELINE configs[4] as complete, DECIDED rounds: N = 1,000,000 members, every one of the ~985,000 surviving members a
    simulated receiver of ~150,000 deliveries (1.5 x 10^11 delivered alerts per round: they never exist at once --
    rapid_sim_round_tiled makes a tile's deliveries on the device, tallies them and accumulates the fast-round votes across the
    tiles' launches).  Three consecutive rounds of continuous churn (10,000 crash + 5,000 join per round, 1 % late deliveries of
    the previous configuration from the second round on), each decided by the accumulated votes of the WHOLE population
    (quorum N - floor((N-1)/4) = 750,001 in the first round), the decided cut applied, the next round in the new configuration:
    * every receiver of one whole tile (1,024 receivers; another tile in every round) against the optimised oracle fed the host
      statement of the same deliveries (scenarios.deliver_hashed): announce batch, getNumProposals, size, fingerprint of the
      oracle's list;
    * every receiver of the population: it announces, and its fingerprint is the checked tile's;
    * the decided cut = the round's fault set; configuration id, membership size and sampled observer lists after every view
      change against the oracle's MembershipView put through the same ringDelete / ringAdd sequence."""
    K, H, L = 10, 9, 4
    n_mem, spare, rounds, tile = 1000000, 16000, 3, 1024
    pop = S.Population.make(n_mem + spare)
    members = list(range(n_mem))
    eng, view = make_engine(E, pop, K, H, L, members=members, max_cut=16384)
    reg, oview = oracle_view(pop, K, members)
    st = S.StreamingChurn(H, L)
    sim = E.ClusterSimulation(eng)
    for rnd in range(rounds):
        obs, subj, member = view.tables()
        cfg = view.getCurrentConfigurationId()
        assert cfg == oview.getCurrentConfigurationId()
        sc, deliver_set = st.next_round_batches(obs, member, cfg)
        rx = sc.receivers
        n_members = int((member != 0).sum())
        assert len(rx) == n_members - len(sc.crashed) and len(sc.crashed) >= 9800 and len(sc.joiners) >= 4900
        assert (deliver_set.n_batches > sc.batches.n_batches) == (rnd > 0)  # the late deliveries are there from the second round on
        seed = 1000 + rnd
        boundary = rnd == 1  # (one round on the 20-byte boundary records, the others on resolved 8-byte records)
        rr = sim.round_tiled(deliver_set, rx, seed, tile_receivers=tile, boundary=boundary)
        info = sim.round_tiled_info()
        assert info["passes"] == 1 and info["tiles"] == -(-len(rx) // tile) and info["records_delivered"] == len(rx) * len(deliver_set.recs)
        quorum = n_members - (n_members - 1) // 4
        assert (rr.decided, rr.quorum, rr.membership_size, rr.cut_size) == (1, quorum, n_members, len(sc.faulty))
        assert rr.votes_winner == rr.votes_total == len(rx) >= quorum  # unanimous: nothing of the round is lost to the late deliveries
        cut = sim.decided_cut()
        assert sorted(cut) == sc.faulty.tolist()
        emit, nprop, pcount, fp = sim.results()
        assert np.all(emit >= 0) and np.all(pcount == len(sc.faulty)) and np.all(nprop >= 1) and len(np.unique(fp)) == 1
        # one whole tile against the oracle, 256 receivers at a time (a tile's 1.5 x 10^8 records are 3 GB on the host)
        k = (3 + 411 * rnd) % (len(rx) // tile)
        for lo in range(k * tile, (k + 1) * tile, 256):
            part = rx[lo:lo + 256]
            recs, off, _ = S.deliver_hashed(deliver_set, part, seed)
            fe, fn, fo, fpp = O.fast_sim_run(pop.n, K, H, L, cfg, obs, subj, member, recs, off, nthreads=64)
            sl = slice(lo, lo + 256)
            assert np.array_equal(emit[sl], fe) and np.array_equal(nprop[sl], fn) and np.array_equal(pcount[sl], np.diff(fo))
            assert np.array_equal(fp[sl], proposal_fingerprints(fo, fpp, fe >= 0))
            assert sorted(fpp[fo[0]:fo[1]].tolist()) == sc.faulty.tolist()
            del recs
        new_cfg = sim.apply_cut(np.asarray(cut, dtype=np.int32))
        is_member = member != 0
        for node in cut:  # decideViewChange on the oracle's view (R/MembershipService.java:394-413)
            if is_member[node]:
                oview.ringDelete(int(node))
            else:
                oview.ringAdd(int(node), (int(pop.id_hi[node]), int(pop.id_lo[node])))
        assert new_cfg == oview.getCurrentConfigurationId() == view.getCurrentConfigurationId()
        assert view.getMembershipSize() == oview.getMembershipSize() == n_members - len(sc.crashed) + len(sc.joiners)
        for s_ in (int(sc.joiners[0]), int(sc.joiners[-1]), int(rx[0]), int(rx[-1])):
            assert view.getObserversOf(s_) == oview.computeObserversOf(s_)
    eng.close()


# ------------------------------------------------------------------ f3: serialized rapid.proto bytes -> device tally
def test_wire_bytes_to_device_tally_with_joiners_unknown_at_start(E):
    """SURVEY 8f rank 3 end to end: a churn round as the reference puts it on the wire -- one serialized
    RapidRequest{BatchedAlertMessage} per sender (rapid.proto:95-129), delivered to every receiver in its own order -- goes
    through rapid_decode_request / rapid_decode_batched_alerts_ex into packed records, is loaded and tallied on the GPU, and
    must give what the oracle gives when it is fed the SAME messages decoded by the Python protobuf runtime.  The joiners'
    endpoints are in neither the endpoint map nor the engine's registry when the round starts (a receiver first hears of
    them through the UP alerts, R/MembershipService.java:677-685): the facade registers them on both sides
    (rapid_endpoint_map_add_wire, rapid_view_register_endpoints) and decodes again; then the fast round decides and the
    cut (crashed members out, joiners in) gives the oracle's next configuration."""
    from rapid_amd import wire as W
    from tests import proto_rapid as P
    n_all, n_mem, K, H, L = 760, 700, 10, 9, 4
    pop = S.Population.make(n_all)
    members = list(range(n_mem))
    reg, oview = oracle_view(pop, K, members)  # the oracle's registry holds everybody (handle == index into pop)
    obs_o, subj_o, member_o = oview.tables(n_all)
    cfg = oview.getCurrentConfigurationId()
    sc = S.build_churn_scenario(obs_o, member_o, cfg, 14, 9, H, L)
    rx = sc.receivers[:: max(1, len(sc.receivers) // 48)][:48]
    sc = S.build_churn_scenario(obs_o, member_o, cfg, 14, 9, H, L, receivers=rx)

    def ep(i):
        return P.Endpoint(hostname=pop.hostnames[i], port=int(pop.ports[i]))

    # one serialized request per batch of the round
    bs = sc.batches
    wire_msgs = []
    for b in range(bs.n_batches):
        msg = P.BatchedAlertMessage(sender=ep(int(bs.sender[b])))
        for rec in bs.recs[bs.off[b]:bs.off[b + 1]]:
            a = P.AlertMessage(edgeSrc=ep(int(rec["src"])), edgeDst=ep(int(rec["dst"])), edgeStatus=int(rec["status"]),
                               configurationId=int(rec["cfg_id"]))
            a.ringNumber.extend([k for k in range(K) if (int(rec["ring_mask"]) >> k) & 1])
            if int(rec["status"]) == S.UP:
                a.nodeId.high, a.nodeId.low = int(pop.id_hi[int(rec["dst"])]), int(pop.id_lo[int(rec["dst"])])
            msg.messages.append(a)
        wire_msgs.append(P.RapidRequest(batchedAlertMessage=msg).SerializeToString())
    # the engine knows the members only
    eng = E.Engine(n_max=n_all, K=K, H=H, L=L)
    view = E.MembershipView(eng).build(pop.hostnames[:n_mem], pop.ports[:n_mem], pop.id_hi[:n_mem], pop.id_lo[:n_mem])
    assert view.getCurrentConfigurationId() == cfg
    emap = W.EndpointMap(pop.hostnames[:n_mem], pop.ports[:n_mem])
    # pass 1: decode every distinct message; register what is unknown, in the order it is met
    new_hosts, new_ports, new_hi, new_lo = [], [], [], []
    for req in wire_msgs:
        kind, payload = W.decode_request(req)
        assert kind == W.MSG_BATCHED_ALERT
        recs_, ids_, status, unknown, sender = emap.decode_batched_alerts_ex(payload, K)
        for i, st_ in enumerate(status):
            if st_ == E.N.ENODE_MISSING:
                before = emap.size()
                idx = emap.add_wire(unknown[i])
                if idx == before:  # first time: the engine gets the same index
                    h_, p_ = emap.get(idx)
                    new_hosts.append(h_); new_ports.append(p_); new_hi.append(ids_[i][0]); new_lo.append(ids_[i][1])
            else:
                assert st_ == E.N.OK
    assert len(new_hosts) == len(sc.joiners)
    assert view.registerEndpoints(new_hosts, new_ports, new_hi, new_lo) == n_mem
    assert view.getCurrentConfigurationId() == cfg  # the membership did not change
    to_pop = {i: i for i in range(n_mem)}
    for j, h_ in enumerate(new_hosts):
        to_pop[n_mem + j] = pop.hostnames.index(h_)
    # pass 2: every receiver's deliveries, message by message, in its own order (S.deliver's permutation)
    streams, off = [], [0]
    for r in sc.receivers:
        rng = np.random.Generator(np.random.PCG64([2, int(r)]))
        parts = []
        for b in rng.permutation(bs.n_batches):
            kind, payload = W.decode_request(wire_msgs[int(b)])
            recs_, ids_, status, unknown, sender = emap.decode_batched_alerts_ex(payload, K)
            assert all(s_ == E.N.OK for s_ in status)
            parts.append(recs_)
        streams.append(np.concatenate(parts))
        off.append(off[-1] + len(streams[-1]))
    records = np.concatenate(streams)
    # the same deliveries through the Python protobuf runtime, in the oracle's numbering
    idx_of = {(pop.hostnames[i], int(pop.ports[i])): i for i in range(n_all)}
    o_streams = []
    for r in sc.receivers:
        rng = np.random.Generator(np.random.PCG64([2, int(r)]))
        parts = []
        for b in rng.permutation(bs.n_batches):
            m_ = P.RapidRequest.FromString(wire_msgs[int(b)]).batchedAlertMessage
            recs_ = np.zeros(len(m_.messages), dtype=S.ALERT_DTYPE)
            for i, a in enumerate(m_.messages):
                recs_[i] = (a.configurationId, idx_of[(a.edgeSrc.hostname, a.edgeSrc.port)], idx_of[(a.edgeDst.hostname, a.edgeDst.port)],
                            sum(1 << k for k in a.ringNumber), a.edgeStatus, 0)
            recs_["flags"][-1] = S.FLAG_LAST_IN_BATCH
            parts.append(recs_)
        o_streams.append(np.concatenate(parts))
    o_records = np.concatenate(o_streams)
    assert np.array_equal(o_records, sc.records)  # (and both equal what the generator delivers directly)
    oe, on, oo, op = O.sim_run(oview, K, H, L, pop.id_hi, pop.id_lo, o_records, np.array(off), nthreads=8)
    sim = E.ClusterSimulation(eng)
    sim.load_streams(records, np.array(off))
    sim.tally()
    emit, nprop, pcount, fp = sim.results()
    assert np.array_equal(emit, oe) and np.array_equal(nprop, on) and np.array_equal(pcount, np.diff(oo))
    for r in range(len(oe)):
        assert sorted(to_pop[x] for x in sim.proposal(r)) == op[oo[r]:oo[r + 1]].tolist() == sc.faulty.tolist()
    # the 48 receivers agree but are no quorum of 700: the cut they all announce is applied, joiners enter with their NodeIds
    new_cfg = sim.apply_cut(sim.proposal(0))
    _oracle_decide(oview, K, H, L, pop, sc, sc.faulty.tolist())
    assert new_cfg == oview.getCurrentConfigurationId() and view.getMembershipSize() == n_mem - 14 + 9
    for node in (n_mem, n_mem + 3):
        assert sorted(to_pop[x] for x in view.getObserversOf(node)) == sorted(oview.getObserversOf(to_pop[node]))


# ------------------------------------------------------------------ f4: the round on the reference's own time line
def test_round_driven_by_the_failure_detector_and_batching_timers(E):
    """SURVEY 8f rank 4 wired to the engine: crashes at given instants -> PingPongFailureDetector notifications
    (R/monitoring/impl/PingPongFailureDetector.java:41-85) -> AlertBatcher flushes (R/MembershipService.java:613-637) ->
    arrival order at every receiver -> tally ON THE GPU -> proposal and fast-round decision times.  Results against the
    faithful oracle on the same timed streams, the producer side against its literal event simulation, and the protocol's
    time-to-stable-cut (~10 s of failure detection + batching + network) next to the engine's compute time."""
    from rapid_amd import timeline as T
    n, K, H, L = 2000, 10, 9, 4
    pop = S.Population.make(n)
    eng, view = make_engine(E, pop, K, H, L)
    reg, oview = oracle_view(pop, K)
    obs, subj, member = view.tables()
    cfg = view.getCurrentConfigurationId()
    rng = np.random.default_rng(8)
    faulty = np.sort(rng.choice(n, 20, replace=False))
    crash = np.full(n, T.NEVER, dtype=np.int64)
    crash[faulty] = rng.integers(2_000, 2_800, len(faulty))
    start = rng.integers(0, 1000, n)
    model, lat = T.ProducerModel(), T.LatencyModel(base_ms=1, jitter_ms=6, seed=9)
    sim = E.ClusterSimulation(eng)
    out = T.engine_round_on_the_time_line(sim, subj, crash, start, cfg, n, model, lat)
    rx = out["receivers"]
    # the producer side equals the literal event simulation (one PingPongFailureDetector object per detector, one AlertBatcher per node)
    from tests.test_timeline import as_tuples, oracle_batches
    t_end = int(crash[faulty].max() + 14 * model.fd_interval_ms + 10 * model.batching_window_ms)
    assert as_tuples(out["batches"], out["send_ms"]) == oracle_batches(subj, crash, start, t_end, model)
    # the tally on the device == the faithful oracle on the same timed streams
    m = np.arange(0, len(rx), max(1, len(rx) // 120))
    sub_off = np.zeros(len(m) + 1, dtype=np.int64)
    parts = []
    for i, r in enumerate(m):
        parts.append(out["records"][out["rec_off"][r]:out["rec_off"][r + 1]])
        sub_off[i + 1] = sub_off[i] + len(parts[-1])
    oe, on, oo, op = O.sim_run(oview, K, H, L, pop.id_hi, pop.id_lo, np.concatenate(parts), sub_off, nthreads=8)
    assert np.array_equal(out["emit_batch"][m], oe) and np.array_equal(out["num_proposals"][m], on)
    assert np.array_equal(out["fingerprint"][m], proposal_fingerprints(oo, op, oe >= 0))
    assert np.all(out["emit_batch"] >= 0) and sorted(sim.proposal(0)) == faulty.tolist()
    # on the time line: ten failed probes, one per second, then the batching window and the network
    t_prop, t_dec = out["proposal_ms"], out["decision_ms"]
    assert t_prop.min() > crash[faulty].min() + 10_000 and t_prop.max() < crash[faulty].max() + 11_000 + 2 * model.batching_window_ms + 10
    quorum = n - (n - 1) // 4
    assert np.all(t_dec >= np.sort(t_prop)[quorum - 1] + 1) and np.all(t_dec <= t_prop.max() + 7)
    assert 10_000 < out["time_to_stable_cut_ms"] < 12_500
    rr = sim.count_votes()
    assert rr.decided == 1 and sorted(sim.decided_cut()) == faulty.tolist()


def test_round_taken_tile_by_tile_accumulates_the_votes(E, test_build):
    """rapid_sim_round_tiled: a population taken tile by tile -- the deliveries of a tile made on the device, tallied, the fast-round
    votes accumulated across the tiles' launches -- equals the same round held in ONE launch (rapid_sim_generate + tally +
    count_votes, itself checked against the oracle above): every receiver's announce batch / getNumProposals() / size /
    fingerprint, the decision, the cut, the votes; for tiles that do not divide the population, both record formats, late
    deliveries among the batches.  And a round in which the LOWEST voter dissents while a quorum agrees: the first candidate ends
    without a quorum, the fingerprints' histogram names the proposal that can have one, the tiles are taken once more for it."""
    K, H, L = 10, 9, 4
    n, n_crash, n_join, seed = 1500, 30, 12, 977
    pop = S.Population.make(n + 40)
    eng, view = make_engine(E, pop, K, H, L, members=list(range(n)))
    obs, subj, member = view.tables()
    cfg = view.getCurrentConfigurationId()
    sc = S.build_churn_scenario(obs, member, cfg, n_crash, n_join, H, L, materialise=False)
    rx = sc.receivers
    both = S.with_late_batches(sc.batches, cfg - 1, 0.2, 5)
    sim = E.ClusterSimulation(eng)
    for batches in (sc.batches, both):
        for boundary in (False, True):
            sim.generate(batches, rx, seed, boundary=boundary, trust_copies=True)
            sim.tally()
            want = [a.copy() for a in sim.results()]
            rr1 = sim.count_votes()
            cut1 = sim.decided_cut()
            assert rr1.decided == 1 and sorted(cut1) == sc.faulty.tolist()
            for tile in (0, 256, 1000, 97, -97):
                # (a tile's deliveries are made on a second stream while the tile before it is tallied; -97: knob bit 21, one
                # stream and one buffer, the tiles strictly one after the other)
                sim.set_force_exact(2097152 if tile < 0 else 0)
                tile = abs(tile)
                rr = sim.round_tiled(batches, rx, seed, tile_receivers=tile, boundary=boundary)
                sim.set_force_exact(0)
                got = sim.results()
                assert all(np.array_equal(a, b) for a, b in zip(want, got)), (boundary, tile)
                assert (rr.decided, rr.votes_winner, rr.votes_total, rr.cut_size, rr.quorum) == (1, rr1.votes_winner, rr1.votes_total, rr1.cut_size, rr1.quorum)
                assert sim.decided_cut() == cut1
                info = sim.round_tiled_info()
                t = tile if tile else 16 * 256
                assert info["passes"] == 1 and info["tiles"] == -(-len(rx) // min(t, len(rx))) and info["records_delivered"] == len(rx) * len(batches.recs)
                last = len(rx) - 1
                assert sorted(sim.proposal(last)) == sc.faulty.tolist()  # the last tile's node lists are resident ...
                if tile and tile < len(rx):
                    with pytest.raises(E.RapidError):                    # ... the others went with their tiles
                        sim.proposal(0)
                with pytest.raises(E.RapidError):  # the streams of a tiled round are gone: nothing to tally or count again
                    sim.tally()
                with pytest.raises(E.RapidError):
                    sim.count_votes()
    # ---- the round sharded over three ranks: every rank takes ITS receivers tile by tile and contributes the block its last pass
    # left for the all-gather; the merge every rank then runs (vote_merge_kernel) must decide what the whole population decides ----
    sim.generate(sc.batches, rx, seed, trust_copies=True)
    sim.tally()
    rr1 = sim.count_votes()
    cut1 = sim.decided_cut()
    cuts = [0, len(rx) // 3, len(rx) // 3 + 500, len(rx)]
    segs = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        rr = sim.round_tiled(sc.batches, rx[a:b], seed, tile_receivers=200)
        assert rr.decided == 0 and rr.votes_winner == rr.votes_total == b - a  # (a shard alone: unanimous, no quorum)
        segs.append(sim.vote_segment())
    status, rr = sim.merge_vote_segments(segs)
    assert status == 1 and (rr.decided, rr.cut_size, rr.votes_total, rr.votes_winner, rr.quorum) == (1, rr1.cut_size, len(rx), len(rx), rr1.quorum)
    assert sim.decided_cut() == cut1
    # ... and through a real (one-rank) RCCL communicator: the all-gather + merge of the tiled round, and -- for the round below, in
    # which the first candidate has no quorum -- the histogram all-reduce and the max-reduce of the exact plurality
    eng_c, _ = make_engine(E, pop, K, H, L, members=list(range(n)))
    eng_c.comm_init(E.comm_unique_id(), 0, 1)
    sim_c = E.ClusterSimulation(eng_c)
    rr = sim_c.round_tiled(sc.batches, rx, seed, tile_receivers=256)
    assert (rr.decided, rr.votes_winner, rr.votes_total, rr.cut_size, rr.quorum) == (1, rr1.votes_winner, rr1.votes_total, rr1.cut_size, rr1.quorum)
    assert sim_c.decided_cut() == cut1 and sim_c.round_tiled_info()["passes"] == 1

    # ---- the lowest voter dissents, a quorum agrees ----
    # one more "batch": a single alert naming a healthy member on all K rings (a legal AlertMessage with K ring numbers), which
    # reaches nine receivers in ten; the others propose the cut without it.  The seed is chosen so that receiver 0 is one of them.
    healthy = np.setdiff1d(np.flatnonzero(member != 0), sc.faulty)
    extra = E.alert(int(healthy[5]), int(healthy[7]), E.DOWN, cfg, list(range(K)))
    extra["flags"] = 1
    bs = S.BatchSet(np.concatenate([sc.batches.recs, extra]), np.concatenate([sc.batches.off, [sc.batches.off[-1] + 1]]).astype(np.int64),
                    np.concatenate([sc.batches.sender, [int(healthy[5])]]).astype(np.int32))
    keep = np.full(bs.n_batches, 0xFFFFFFFF, dtype=np.uint32)
    keep[-1] = int(0.9 * 0xFFFFFFFF)
    s2 = next(s for s in range(100, 400) if not S.delivered_mask(s, int(rx[0]), keep)[-1])
    minority = sum(1 for r in rx if not S.delivered_mask(s2, int(r), keep)[-1])
    assert 0 < minority < len(rx) // 5
    for boundary in (False, True):
        sim.generate(bs, rx, s2, keep=keep, boundary=boundary)
        sim.tally()
        want = [a.copy() for a in sim.results()]
        rr1 = sim.count_votes()
        # (a receiver that hears of the extra subject only after it has announced proposes the plain cut, too: the majority is
        # smaller than the receivers the batch reaches -- R/MembershipService.java:318-319 ignores what comes after the announcement)
        assert rr1.decided == 1 and rr1.quorum <= rr1.votes_winner <= len(rx) - minority and sorted(sim.decided_cut()) == sorted(sc.faulty.tolist() + [int(healthy[7])])
        cut1 = sim.decided_cut()
        for tile in (0, 300):
            rr = sim.round_tiled(bs, rx, s2, tile_receivers=tile, keep=keep, boundary=boundary)
            assert all(np.array_equal(a, b) for a, b in zip(want, sim.results()))
            assert (rr.decided, rr.votes_winner, rr.votes_total, rr.cut_size) == (1, rr1.votes_winner, rr1.votes_total, len(cut1)) and sim.decided_cut() == cut1
            assert sim.round_tiled_info()["passes"] == 2
        rr = sim_c.round_tiled(bs, rx, s2, tile_receivers=300, keep=keep, boundary=boundary)  # (the one-rank communicator)
        assert all(np.array_equal(a, b) for a, b in zip(want, sim_c.results()))
        assert (rr.decided, rr.votes_winner, rr.votes_total, rr.cut_size) == (1, rr1.votes_winner, rr1.votes_total, len(cut1)) and sim_c.decided_cut() == cut1
        assert sim_c.round_tiled_info()["passes"] == 2
    eng_c.close()
    # ... and nobody has a quorum when four receivers in ten dissent: undecided, after ONE pass over the population
    keep[-1] = int(0.5 * 0xFFFFFFFF)
    rr = sim.round_tiled(bs, rx, s2, tile_receivers=300, keep=keep)
    assert rr.decided == 0 and 0 < rr.votes_winner < rr.quorum and rr.votes_total == len(rx) and sim.round_tiled_info()["passes"] == 1
    with pytest.raises(E.RapidError):
        sim.decided_cut()
    # an empty population, and misuse
    rr = sim.round_tiled(sc.batches, rx[:0], seed)
    assert rr.decided == 0 and rr.votes_total == 0
    with pytest.raises(E.IllegalArgumentException):
        sim.round_tiled(S.BatchSet(sc.batches.recs, np.concatenate([sc.batches.off[:1], sc.batches.off]), np.concatenate([[0], sc.batches.sender])), rx, 1)
    eng.close()
