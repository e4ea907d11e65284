"""TEST INFRASTRUCTURE ONLY -- seeded random sweep of the emulated tally kernel against the oracle, beyond the fixed
cases of tests/test_kernel_emulated.py:  python tests/emu/sweep.py <first_seed> <last_seed>
Every seed draws a population (8..63 nodes, random membership), K, H, L, ten receiver streams of 0..699 records with
random batch lengths, multi-ring alerts and (two seeds out of three) invalid alerts, a random number of waves per
workgroup and of workgroups, and the three dictionary placements (memory / direct tables / compressed tables in LDS).  Prints the failing seeds; `done, failures: 0` otherwise."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_kernel_emulated import _check
from rapid_amd import scenarios as S
from tests.helpers import oracle_view, random_stream
bad=0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    n_nodes = int(rng.integers(8, 64)); K = int(rng.integers(3, 12)); H = int(rng.integers(1, K + 1)); L = int(rng.integers(1, H + 1))
    pop = S.Population.make(n_nodes)
    n_members = int(rng.integers(max(2, n_nodes // 2), n_nodes + 1))
    members = sorted(rng.permutation(n_nodes)[:n_members].tolist())
    reg, view = oracle_view(pop, K, members)
    obs, subj, member = view.tables(n_nodes)
    cfg = view.getCurrentConfigurationId()
    recs, off = [], [0]
    pb = 0.05 if seed % 3 else 0.0
    for r in range(10):
        hot = rng.permutation(n_nodes)[: int(rng.integers(1, min(n_nodes, 20) + 1))]
        n_rec = int(rng.integers(0, 700))
        recs.append(random_stream(rng, n_nodes, K, member, cfg, n_rec, hot, p_eob=float(rng.choice([0.02, 0.1, 0.3, 0.6, 1.0])), p_multi=float(rng.choice([0.0,0.15,0.5])), p_bad_cfg=pb, p_bad_status=pb))
        off.append(off[-1] + n_rec)
    try:
        _check(np.concatenate(recs), np.array(off), n_nodes, K, H, L, cfg, obs, subj, member, seed=seed, waves=int(rng.integers(1,5)), grid=int(rng.integers(1,4)), tables_in_lds=int(seed%4), pool=bool((seed >> 2) & 1))
    except AssertionError as e:
        bad+=1; print("FAIL seed", seed, n_nodes,K,H,L, str(e)[:200])
print("done, failures:", bad)
