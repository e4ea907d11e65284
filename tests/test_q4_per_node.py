"""Quirk Q4 with ONE cachedObservers PER NODE, as a deployment has them (R/MembershipView.java:49, 210-224: every node holds its own
MembershipView object) -- decided by a test, as the round-4 review asked: does a receiver's proposal ever differ from what the
engine's ONE memo per simulated population yields?

It does, and this file pins exactly when.  A node's cache gets an entry for subject x only when THAT node asks for x's observers
-- invalidateFailingEdges at a batch end with x in ITS preProposal (R/MultiNodeCutDetector.java:140-149) -- and a node that
announced its proposal earlier in the round ignores every later batch (R/MembershipService.java:318-319) and asks nothing any
more.  Scene: x is the maximum of ring 0, m0 its minimum (x's ring-0 observer by wrap-around).
  c1: m0 crashes (ten DOWN reports), x is reported DOWN on five rings (spurious: stays in preProposal).  Receivers of group A hear
      about x first: x sits in their preProposal at batch ends, they cache x's observers [m0, ...] and are blocked by x; receivers
      of group B hear all of m0's reports first, announce {m0} and ignore the rest: no entry for x.
  c2: m0 has left.  ringDelete drops the entries of m0's ring predecessors WITHOUT wrap-around: x, the maximum, keeps its stale
      entry -- at the nodes that have one.
  round in c2: m0 on its way back (UP on every ring), x reported DOWN on eight rings other than ring 0.  A node with the stale
      entry finds m0 as x's ring-0 observer, credits x with the implicit report and proposes {x, m0}; a node without an entry
      computes today's observers, x stays one short of H and blocks (or m0 goes alone).
The engine's memo (and the oracle's shared cache, its model) remembers x from c1 because the ROUND's alert set made x hot -- for
everybody: it reproduces group A at every receiver.  Equal to the per-node reference exactly when every receiver that takes part in
the later round had the subject in its preProposal at a batch end of the round that filled the memo; the engine flags the rounds in
which the difference can exist at all (rapid_sim_index_info: a hot member's memoised observers are stale), and on the benchmark
streams it never does (tests/test_gpu_parity.py)."""
import numpy as np

from oracle import pyoracle as O
from rapid_amd import scenarios as S
from tests.helpers import oracle_view

K, H, L = 10, 9, 4


def _alert(src, dst, status, cfg, ring):
    a = np.zeros(1, dtype=S.ALERT_DTYPE)
    a["src"], a["dst"], a["status"], a["cfg_id"], a["ring_mask"], a["flags"] = src, dst, status, cfg, 1 << ring, 1  # its own batch
    return a


def _scene():
    for n in range(300, 360):
        pop = S.Population.make(n)
        reg, view = oracle_view(pop, K)
        ring0 = view.getRing(0)
        x, m0 = int(ring0[-1]), int(ring0[0])
        obs_x = view.computeObserversOf(x)
        if m0 in obs_x[1:] or x in view.computeObserversOf(m0):
            continue
        return pop, x, m0
    raise AssertionError("no population with the scene found")


def test_per_node_observer_caches_against_the_shared_memo():
    pop, x, m0 = _scene()
    n = pop.n
    views = {}
    for mode in ("per_node", "shared"):
        reg, views[mode] = oracle_view(pop, K)
    v = views["per_node"]
    cfg1 = v.getCurrentConfigurationId()
    obs_m0, obs_x = v.computeObserversOf(m0), v.computeObserversOf(x)
    assert obs_x[0] == m0
    about_m0 = [_alert(obs_m0[k], m0, S.DOWN, cfg1, k) for k in range(K)]
    about_x = [_alert(obs_x[k], x, S.DOWN, cfg1, k) for k in range(1, 6)]  # five rings: L <= 5 < H, never complete
    busy = set(obs_m0) | set(obs_x) | {x, m0}
    nodes = [i for i in range(n) if i not in busy][:10]
    A, B = nodes[:4], nodes[4:]
    streams = [np.concatenate(about_x + about_m0) for _ in A] + [np.concatenate(about_m0 + about_x) for _ in B]
    rec_off = np.arange(len(nodes) + 1, dtype=np.int64) * len(streams[0])
    records = np.concatenate(streams)
    # ---- c1, one cache per node
    oe, on, oo, op = O.sim_run(v, K, H, L, pop.id_hi, pop.id_lo, records, rec_off, receiver_nodes=nodes)
    for i, node in enumerate(nodes):
        in_a = node in A
        assert v.nodeHasCached(node, x) == in_a          # only who had x in preProposal at a batch end asked for its observers
        assert (oe[i] < 0) == in_a                       # ... and is blocked by it; the others announced {m0} and stopped listening
        if not in_a:
            assert op[oo[i]:oo[i + 1]].tolist() == [m0]
    # ---- c1, the shared cache (the engine's model): filled by whoever asks first, read by everybody
    se, sn, so, sp = O.sim_run(views["shared"], K, H, L, pop.id_hi, pop.id_lo, records, rec_off, prewarm_observers=False)
    assert np.array_equal(se, oe) and np.array_equal(sp, op)  # (nothing is stale yet: the caches cannot disagree in c1)
    # ---- the view change the cluster decided: m0 leaves; x's entry survives where there is one (quirk Q4)
    for view in views.values():
        view.ringDelete(m0)
    cfg2 = v.getCurrentConfigurationId()
    assert cfg2 == views["shared"].getCurrentConfigurationId() != cfg1
    fresh_x = v.computeObserversOf(x)
    assert fresh_x[0] != m0 and fresh_x[1:] == obs_x[1:]
    assert all(v.nodeHasCached(a, x) for a in A) and not any(v.nodeHasCached(b, x) for b in B)
    # ---- the round in c2 that the stale entry decides
    exp_m0 = v.getExpectedObserversOf(m0)
    alerts = [_alert(exp_m0[k], m0, S.UP, cfg2, k) for k in range(K)] + [_alert(fresh_x[k], x, S.DOWN, cfg2, k) for k in range(1, 9)]
    rng = np.random.default_rng(5)
    order = [rng.permutation(len(alerts)) for _ in nodes]
    records2 = np.concatenate([np.concatenate([alerts[j] for j in o]) for o in order])
    rec_off2 = np.arange(len(nodes) + 1, dtype=np.int64) * len(alerts)
    pe, pn, po, pp = O.sim_run(v, K, H, L, pop.id_hi, pop.id_lo, records2, rec_off2, receiver_nodes=nodes)
    se, sn, so, sp = O.sim_run(views["shared"], K, H, L, pop.id_hi, pop.id_lo, records2, rec_off2, prewarm_observers=False)
    per_node = [sorted(pp[po[i]:po[i + 1]].tolist()) for i in range(len(nodes))]
    shared = [sorted(sp[so[i]:so[i + 1]].tolist()) for i in range(len(nodes))]
    both = sorted([x, m0])
    assert all(p == both for p in shared)                       # the one memo: the stale row at every receiver
    assert all(per_node[i] == both for i in range(len(A)))      # the nodes that cached x in c1: the same
    differing = [i for i in range(len(nodes)) if per_node[i] != shared[i]]
    assert differing and all(i >= len(A) for i in differing)    # ... and ONLY nodes that never asked about x differ:
    assert all(per_node[i] in ([], [m0]) for i in differing)    # x, one short of H with today's observers, blocks them (or m0 goes alone)
    # what decides it is the cache alone: the same nodes, given the entry, agree with the shared memo
    for view in (v,):
        pass
    reg3, v3 = oracle_view(pop, K)
    O.sim_run(v3, K, H, L, pop.id_hi, pop.id_lo, np.concatenate([np.concatenate(about_x + about_m0) for _ in nodes]), rec_off, receiver_nodes=nodes)
    v3.ringDelete(m0)
    qe, qn, qo, qp = O.sim_run(v3, K, H, L, pop.id_hi, pop.id_lo, records2, rec_off2, receiver_nodes=nodes)
    assert all(sorted(qp[qo[i]:qo[i + 1]].tolist()) == both for i in range(len(nodes)))
