"""apply_cut at N members: where a view change (10,000 out, 5,000 in at 10^6) spends its time.
    RAPID_TIME_VIEW=1 python scripts/time_apply.py [members=1000000]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rapid_amd import engine as E, scenarios as S  # noqa: E402
from rapid_amd import _native as _N  # noqa: E402

_N.use_test_build()  # measurement aids: environment knobs, probes and rapid_debug_* exist in the test build only

n_mem = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
spare = n_mem // 50 + 64
K, H, L = 10, 9, 4
pop = S.Population.make(n_mem + spare)
t = time.perf_counter()
eng = E.Engine(n_max=pop.n, K=K, H=H, L=L, max_cut=max(4096, n_mem // 50))
view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo, members=list(range(n_mem)))
print("build %.1f ms" % (1e3 * (time.perf_counter() - t)), flush=True)
sim = E.ClusterSimulation(eng)
rng = np.random.default_rng(3)
members = np.arange(n_mem)
nxt = n_mem
for rnd in range(3):
    out = rng.choice(members, size=n_mem // 100, replace=False)
    join = np.arange(nxt, nxt + n_mem // 200)
    nxt += len(join)
    cut = np.concatenate([out, join]).astype(np.int32)
    eng.sync()
    t = time.perf_counter()
    cfg = sim.apply_cut(cut)
    eng.sync()
    print("round %d: apply_cut(%d out, %d in) %.3f ms -> config %d" % (rnd, len(out), len(join), 1e3 * (time.perf_counter() - t), cfg), flush=True)
    members = np.setdiff1d(np.concatenate([members, join]), out)
