"""Profiling aid: runs the tally kernel of the -DRAPID_PHASE_TIMERS build (scripts/phase_timers.sh) and prints where
the wave cycles go.  The numbers replace the usual stats counters in that build."""
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np  # noqa: E402
from rapid_amd import engine as E, scenarios as S  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3b"
spec = S.CONFIGS[name]
n, K, H, L = spec["n"], spec["K"], spec["H"], spec["L"]
pop = S.Population.make(n)
eng = E.Engine(n_max=n, K=K, H=H, L=L)
view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)
obs, subj, member = view.tables()
sc = S.build_scenario(name, subj, view.getCurrentConfigurationId())
sim = E.ClusterSimulation(eng)
sim.load_streams(sc.records, sc.rec_off)
sim.set_alert_set(sc.batches.recs)
reps = 3
ms = sim.time_tally(reps)
s = np.zeros(8, dtype=np.uint64)
eng._check(eng._lib.rapid_sim_stats(eng._h, E._addr(s)))
runs = reps + 1
tot = float(s[0])
info = sim.index_info()
print("tally_ms", round(ms, 4), info)
names = ["total", "dma_wait", "lean_window", "careful", "output+init", "flush"]
for i, nm in enumerate(names):
    print("%-12s %6.1f %%   %.0f cycles/receiver" % (nm, 100.0 * float(s[i]) / tot, float(s[i]) / max(1.0, float(s[6]))))
print("receivers", int(s[6]) // runs, "lean windows/receiver", float(s[7]) / max(1.0, float(s[6])))
