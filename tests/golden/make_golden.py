"""Generates the golden fixtures in this directory from the CPU oracle (the reference itself cannot run here: no
JDK).  The oracle is pinned by the reference's own known-answer tests (tests/test_oracle_kat.py), so these vectors are
"reference semantics as restated", frozen: any later change of oracle OR engine that alters them is flagged.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyoracle as O  # noqa: E402
from rapid_amd import scenarios as S  # noqa: E402
from tests.helpers import oracle_view  # noqa: E402

CASES = [("c1_n50_k3", "C1", 50, 1, 3, 3, 1), ("c2_n300_k10", "C2", 300, 12, 10, 9, 4), ("c3b_n400_k10", "C3b", 400, 16, 10, 9, 4)]

for fname, name, n, f, K, H, L in CASES:
    pop = S.Population.make(n)
    reg, view = oracle_view(pop, K)
    obs, subj, member = view.tables(n)
    cfg = view.getCurrentConfigurationId()
    sc = S.build_scenario(name, subj, cfg, n=n, f=f, H=H, L=L)
    emit, nprop, poff, props = O.sim_run(view, K, H, L, pop.id_hi, pop.id_lo, sc.records, sc.rec_off)
    rings = np.stack([view.getRing(k) for k in range(K)])
    keys0 = np.array([view.ringKey(0, i) for i in range(n)], dtype=np.int64)
    # the cut the population decides on and the configuration that follows
    votes = {}
    for r in range(len(emit)):
        if emit[r] >= 0:
            t = tuple(props[poff[r]:poff[r + 1]].tolist())
            votes[t] = votes.get(t, 0) + 1
    best = max(votes.items(), key=lambda kv: kv[1])
    cut = sorted(best[0], key=lambda x: keys0[x])
    svc = O.AlertBatchService(view, K, H, L, pop.id_hi, pop.id_lo)
    svc.decideViewChange(cut)
    np.savez_compressed(os.path.join(HERE, fname + ".npz"), n=n, K=K, H=H, L=L, f=f, config_id=cfg, rings=rings, obs=obs,
                        subj=subj, keys0=keys0, records=sc.records, rec_off=sc.rec_off, alert_set=sc.batches.recs,
                        emit_batch=emit, num_proposals=nprop, prop_off=poff, props=props, faulty=sc.faulty,
                        cut=np.array(cut, dtype=np.int32), votes_winner=best[1], next_config_id=view.getCurrentConfigurationId())
    print(fname, "records", len(sc.records), "receivers", len(emit), "proposers", int((emit >= 0).sum()), "cut", len(cut))
