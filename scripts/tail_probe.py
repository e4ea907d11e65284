"""Where the tally kernel's time goes beyond its steady state: kernel time against the number of receivers of C3b it is given
(a prefix of the population), next to the stream probe (same access pattern, no processing) over the same prefix.  The slope is
the steady-state rate, the intercept what a launch pays once (table staging at its head, the ramp-down at its tail)."""
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from rapid_amd import engine as E, scenarios as S  # noqa: E402
from rapid_amd import _native as _N  # noqa: E402

_N.use_test_build()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
spec = S.CONFIGS["C3b"]
n, K, H, L = spec["n"], spec["K"], spec["H"], spec["L"]
pop = S.Population.make(n)
eng = E.Engine(n_max=n, K=K, H=H, L=L)
view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)
obs, subj, member = view.tables()
sc = S.build_scenario("C3b", subj, view.getCurrentConfigurationId())
sim = E.ClusterSimulation(eng)
R = len(sc.rec_off) - 1
print("receivers", R, "records", len(sc.records))
rows = []
for r in [256, 512, 1024, 1920, 3840, 5760, 7680, 8704, 9492][::-1]:
    r = min(r, R)
    nrec = int(sc.rec_off[r])
    sim.load_streams(sc.records[:nrec], sc.rec_off[:r + 1])
    sim.set_alert_set(sc.batches.recs, trust_copies=True)
    ms = min(sim.time_tally(reps) for _ in range(3))
    info = sim.index_info(timed=False)
    probe = min(sim.stream_probe(0, 16, reps) for _ in range(3))
    rows.append((r, nrec, ms, probe))
    print("receivers %5d records %9d  tally %.4f ms (%.0f GB/s)  stream probe %.4f ms (%.0f GB/s)  waves/wg %d wgs %d" %
          (r, nrec, ms, 20 * nrec / ms / 1e6, probe, 20 * nrec / probe / 1e6, info["waves_per_workgroup"], info["workgroups"]), flush=True)
a = np.array(rows, dtype=np.float64)
for name, col in (("tally", 2), ("probe", 3)):
    big = a[a[:, 0] >= 3840]
    slope, icpt = np.polyfit(big[:, 1] * 20, big[:, col], 1)
    print("%s: fit over >= 3840 receivers: %.0f GB/s steady + %.4f ms per launch" % (name, 1e-6 / slope, icpt))
