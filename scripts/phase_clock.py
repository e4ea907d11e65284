"""Measurement aid (profiling build with -DRAPID_PHASE_TIMERS -DRAPID_TIMER_REALTIME): shader cycles vs constant-rate
time per receiver -- does the clock change over the life of one tally launch, and when was each receiver processed?"""
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
print("tally_ms (back-to-back launches)", round(sim.time_tally(5), 4))
sim.tally()
emit, nprop, pcount, fpw = sim.results()
cyc = (fpw & np.uint64(0xFFFFFFFF)).astype(np.float64)
real = (fpw >> np.uint64(32)).astype(np.float64) * 10.0  # ns
start = emit.astype(np.int64)
start = (start - start.min()) * 10.0 / 1000.0  # us since the first receiver started
print("kernel span by receiver start+duration: %.1f us" % ((start + real / 1000.0).max()))
order = np.argsort(start)
for lo in range(0, len(order), 1000):
    sel = order[lo:lo + 1000]
    print("started %7.1f..%7.1f us: duration %7.1f us  %8.0f cycles  -> %.0f MHz" % (
        start[sel].min(), start[sel].max(), real[sel].mean() / 1000.0, cyc[sel].mean(), cyc[sel].sum() / real[sel].sum() * 1000.0))
