"""Profiling aid: runs the tally kernel of the -DRAPID_PHASE_TIMERS build (scripts/phase_timers.sh) and prints where
the wave cycles go.  The numbers replace the usual stats counters in that build."""
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np  # noqa: E402
from rapid_amd import engine as E, scenarios as S  # noqa: E402
from rapid_amd import _native as _N  # noqa: E402

_N.use_test_build()  # measurement aids: environment knobs, probes and rapid_debug_* exist in the test build only

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
sim.set_alert_set(sc.batches.recs, trust_copies=True)
reps = 3
ms = sim.time_tally(reps)
s = np.zeros(8, dtype=np.uint64)
eng._check(eng._lib.rapid_sim_stats(eng._h, E._addr(s)))
runs = reps + 1
tot = float(s[0])
info = sim.index_info()
print("tally_ms", round(ms, 4), info)
names = ["total", "fast windows, steady-state loop", "cold + fast windows, general iteration", "slow windows", "output+init", "flush+sweeps"]
for i, nm in enumerate(names):
    print("%-42s %6.1f %%   %.0f cycles/receiver" % (nm, 100.0 * float(s[i]) / tot, float(s[i]) / max(1.0, float(s[6]))))
print("receivers", int(s[6]) // runs, "windows taken by the steady-state loop per receiver", float(s[7]) / max(1.0, float(s[6])))

sim.tally()
emit, nprop, pcount, fpw = sim.results()
cyc = (fpw & np.uint64(0xFFFFFFFF)).astype(np.float64)
dma = (fpw >> np.uint64(32)).astype(np.float64)  # flush + sweeps
careful = nprop.astype(np.float64) * 16
lean = pcount.astype(np.float64) * 16
order = np.argsort(cyc)
print("per-receiver cycles: min %.0f  median %.0f  p90 %.0f  p99 %.0f  max %.0f  sum/1e6 %.1f" % (
    cyc.min(), np.median(cyc), np.percentile(cyc, 90), np.percentile(cyc, 99), cyc.max(), cyc.sum() / 1e6))
for name, sel in (("slowest 1%", order[-100:]), ("median 1%", order[4950:5050]), ("fastest 1%", order[:100])):
    print("%-11s total %8.0f  flush+sweep %8.0f  windows %8.0f  slow %8.0f  other %8.0f   mean receiver index %.0f" % (
        name, cyc[sel].mean(), dma[sel].mean(), lean[sel].mean(), careful[sel].mean(),
        (cyc[sel] - dma[sel] - lean[sel] - careful[sel]).mean(), sel.mean()))
# does the time depend on when (which round) a receiver was processed?
for lo in range(0, len(cyc), 1000):
    sel = np.arange(lo, min(lo + 1000, len(cyc)))
    print("receivers %5d.. total %8.0f flush+sweep %8.0f windows %8.0f slow %8.0f" % (lo, cyc[sel].mean(), dma[sel].mean(), lean[sel].mean(), careful[sel].mean()))
