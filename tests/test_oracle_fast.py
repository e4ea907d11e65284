"""oracle/fast_cut.hpp (dense masks + incremental invalidation) must agree with the faithful restatement
(oracle/rapid_oracle.hpp) on every output the batch path defines: which batch announced, getNumProposals(),
and the proposal set -- for adversarial random streams and for the BASELINE scenario generators."""
import numpy as np
import pytest

from oracle import pyoracle as O
from rapid_amd import scenarios as S
from tests.helpers import oracle_view, props_list, random_stream


def _compare(view, pop, K, H, L, records, rec_off, n_nodes, orders=(0, 1, 2)):
    cfg = view.getCurrentConfigurationId()
    obs, subj, member = view.tables(n_nodes)
    fe, fn, fo, fp = O.fast_sim_run(n_nodes, K, H, L, cfg, obs, subj, member, records, rec_off)
    for order in orders:
        e, n, o, p = O.sim_run(view, K, H, L, pop.id_hi, pop.id_lo, records, rec_off, snapshot_order=order)
        assert np.array_equal(e, fe), (order, np.flatnonzero(e != fe)[:5])
        assert np.array_equal(n, fn), (order, np.flatnonzero(n != fn)[:5])
        assert np.array_equal(o, fo)
        assert np.array_equal(p, fp)
    return fe, fn, fo, fp


@pytest.mark.parametrize("seed", range(12))
def test_random_adversarial_streams(seed):
    rng = np.random.default_rng(seed)
    n_nodes = int(rng.integers(8, 48))
    K = int(rng.integers(3, 11))
    H = int(rng.integers(1, K + 1))
    L = int(rng.integers(1, H + 1))
    pop = S.Population.make(n_nodes)
    n_members = int(rng.integers(max(2, n_nodes // 2), n_nodes + 1))
    members = sorted(rng.permutation(n_nodes)[:n_members].tolist())
    reg, view = oracle_view(pop, K, members)
    member = np.zeros(n_nodes, dtype=np.uint8)
    member[members] = 1
    cfg = view.getCurrentConfigurationId()
    R = 60
    recs, off = [], [0]
    for r in range(R):
        hot = rng.permutation(n_nodes)[: int(rng.integers(1, min(n_nodes, 12) + 1))]
        n_rec = int(rng.integers(0, 160))
        recs.append(random_stream(rng, n_nodes, K, member, cfg, n_rec, hot, p_eob=float(rng.choice([0.1, 0.5, 1.0]))))
        off.append(off[-1] + n_rec)
    records = np.concatenate(recs)
    e, n, o, p = _compare(view, pop, K, H, L, records, np.array(off), n_nodes)
    assert (e >= 0).any() or H > 6  # the generator must actually produce proposals in most settings


@pytest.mark.parametrize("name,n,f,K,H,L", [("C1", 50, 1, 3, 3, 1), ("C2", 300, 12, 10, 9, 4), ("C2", 200, 20, 10, 8, 2),
                                            ("C3a", 400, 20, 10, 9, 4), ("C3b", 400, 20, 10, 9, 4),
                                            ("C3b", 400, 10, 10, 7, 3)])
def test_scenarios_small(name, n, f, K, H, L):
    pop = S.Population.make(n)
    reg, view = oracle_view(pop, K)
    obs, subj, member = view.tables(n)
    cfg = view.getCurrentConfigurationId()
    sc = S.build_scenario(name, subj, cfg, n=n, f=f, H=H, L=L)
    e, npr, o, p = _compare(view, pop, K, H, L, sc.records, sc.rec_off, n, orders=(0, 2))
    is_f = np.zeros(n, dtype=bool)
    is_f[sc.faulty] = True
    if name in ("C1", "C2", "C3b"):
        # a stable cut exists: a clear majority of receivers proposes exactly the fault set
        exact = sum(props_list(o, p, r) == sc.faulty.tolist() for r in range(len(sc.receivers)))
        assert exact > 0.6 * len(sc.receivers)


def test_loss_and_stale_cfg():
    n, K, H, L = 300, 10, 9, 4
    pop = S.Population.make(n)
    reg, view = oracle_view(pop, K)
    obs, subj, member = view.tables(n)
    cfg = view.getCurrentConfigurationId()
    sc = S.build_scenario("C2", subj, cfg, n=n, f=15, H=H, L=L, materialise=False)
    records, rec_off, nb = S.deliver(sc.batches, sc.receivers[:80], 7, loss=0.1, stale_cfg=cfg ^ 1, stale_rate=0.05)
    _compare(view, pop, K, H, L, records, rec_off, n, orders=(0,))


def test_empty_and_ragged():
    n, K, H, L = 20, 5, 4, 2
    pop = S.Population.make(n)
    reg, view = oracle_view(pop, K)
    cfg = view.getCurrentConfigurationId()
    member = np.ones(n, dtype=np.uint8)
    rng = np.random.default_rng(5)
    parts = [np.zeros(0, dtype=S.ALERT_DTYPE), random_stream(rng, n, K, member, cfg, 1, [3]),
             np.zeros(0, dtype=S.ALERT_DTYPE), random_stream(rng, n, K, member, cfg, 300, [1, 2, 3])]
    off = np.cumsum([0] + [len(x) for x in parts])
    e, npr, o, p = _compare(view, pop, K, H, L, np.concatenate(parts), off, n)
    assert e[0] == -1 and e[2] == -1 and npr[0] == 0


@pytest.mark.parametrize("n,n_out,n_crash,n_join,K,H,L", [(300, 30, 10, 12, 10, 9, 4), (120, 20, 3, 8, 10, 8, 2),
                                                           (200, 10, 0, 10, 7, 6, 3)])
def test_churn_crashes_and_joins_in_one_configuration(n, n_out, n_crash, n_join, K, H, L):
    """UP alerts from the expected observers of joiners (non-members, Q3) mixed with DOWN alerts about
    crashed members; a joiner or a crashed node whose observer crashed too completes through the implicit invalidation.
    Every surviving member must announce exactly crashed + joiners, and decideViewChange must add / remove them
    (R/MembershipService.java:385-430)."""
    pop = S.Population.make(n)
    members = list(range(0, n - n_out))
    reg, view = oracle_view(pop, K, members)
    obs, subj, member = view.tables(n)
    cfg = view.getCurrentConfigurationId()
    sc = S.build_churn_scenario(obs, member, cfg, n_crash, n_join, H, L)
    rx = sc.receivers[:: max(1, len(sc.receivers) // 25)]
    sc = S.build_churn_scenario(obs, member, cfg, n_crash, n_join, H, L, receivers=rx)
    assert set(np.unique(sc.batches.recs["status"]).tolist()) <= {S.UP, S.DOWN}
    fe, fn, fo, fp = _compare(view, pop, K, H, L, sc.records, sc.rec_off, n, orders=(0, 1))
    assert np.all(fe >= 0)
    for r in range(len(fe)):
        assert sorted(fp[fo[r]:fo[r + 1]].tolist()) == sc.faulty.tolist()
    # the decided cut applied the way the Java does: members leave, joiners enter with their NodeId
    svc = O.AlertBatchService(view, K, H, L, pop.id_hi, pop.id_lo)
    r0 = sc.records[sc.rec_off[0]:sc.rec_off[1]]  # one receiver's stream, batch by batch: records the joiners' NodeIds
    ends = np.flatnonzero(r0["flags"] & S.FLAG_LAST_IN_BATCH) + 1
    announced, beg = [], 0
    for e in ends:
        announced += svc.handleBatchedAlertMessage(r0[beg:e])
        beg = int(e)
    assert sorted(announced) == sc.faulty.tolist() and svc.announcedProposal()
    svc.decideViewChange(announced)
    assert view.getMembershipSize() == (n - n_out) - n_crash + n_join
    for j in sc.joiners:
        assert view.isHostPresent(int(j))
    for c in sc.crashed:
        assert not view.isHostPresent(int(c))
