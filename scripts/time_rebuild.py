"""Measurement aid: cost of one view change (rebuild of the K rings, tables and configuration id) at N = 10,000."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rapid_amd import engine as E, scenarios as S
from rapid_amd import _native as _N  # noqa: E402

_N.use_test_build()  # measurement aids: environment knobs, probes and rapid_debug_* exist in the test build only
n, K, H, L = 10000, 10, 9, 4
pop = S.Population.make(n)
eng = E.Engine(n_max=n, K=K, H=H, L=L)
view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)
ts = []
for i in range(12):
    eng.sync()
    t = time.perf_counter()
    view.ringDelete(100 + i)
    eng.sync()
    ts.append(1e3 * (time.perf_counter() - t))
print("ringDelete + rebuild, ms:", " ".join("%.3f" % x for x in ts), " median %.3f" % sorted(ts)[len(ts) // 2])
