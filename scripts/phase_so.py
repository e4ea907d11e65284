"""Measurement aid (profiling build): phase timers of the tally kernel in stream-only mode."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rapid_amd import engine as E, scenarios as S
n, K, H, L = 10000, 10, 9, 4
pop = S.Population.make(n)
eng = E.Engine(n_max=n, K=K, H=H, L=L)
view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)
obs, subj, member = view.tables()
sc = S.build_scenario("C3b", subj, view.getCurrentConfigurationId())
sim = E.ClusterSimulation(eng)
sim.load_streams(sc.records, sc.rec_off)
sim.set_alert_set(sc.batches.recs, trust_copies=True)
sim.set_force_exact(32)
ms = sim.time_tally(3)
s = np.zeros(8, dtype=np.uint64)
eng._check(eng._lib.rapid_sim_stats(eng._h, E._addr(s)))
tot = float(s[0])
print("stream_only tally_ms", round(ms, 4))
for i, nm in enumerate(["total", "dma_wait", "lean_window", "careful", "output+init", "flush"]):
    print("%-12s %6.1f %%   %.0f cycles/receiver" % (nm, 100.0 * float(s[i]) / tot, float(s[i]) / max(1.0, float(s[6]))))
