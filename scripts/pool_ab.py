"""Common pool for the tail of a launch: tally kernel time with everything dealt statically (knob 1024) and with the last
1/8 .. 3/8 of the receivers in the pool (RAPID_POOL_EIGHTHS), vouched and per-delivery-filter kernels, interleaved.
    python scripts/pool_ab.py [config] [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rapid_amd import engine as E  # noqa: E402
from rapid_amd import scenarios as S  # noqa: E402
from rapid_amd import _native as _N  # noqa: E402

_N.use_test_build()  # measurement aids: environment knobs, probes and rapid_debug_* exist in the test build only

name = sys.argv[1] if len(sys.argv) > 1 else "C3b"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
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
nbytes = 20 * len(sc.records)
for rnd in range(3):
    row = []
    for label, knob, eighths in (("static", 1024, 1), ("pool 1/8", 0, 1), ("pool 2/8", 0, 2), ("pool 3/8", 0, 3)):
        os.environ["RAPID_POOL_EIGHTHS"] = str(eighths)
        sim.set_force_exact(knob)
        a = min(sim.time_tally(reps) for _ in range(2))
        sim.set_force_exact(knob | 64)
        b = min(sim.time_tally(reps) for _ in range(2))
        row.append("%s: %.4f (%.1f %%) / filter %.4f" % (label, a, 100 * nbytes / a / 1e6 / 8000, b))
    print("round %d  " % rnd + "   ".join(row), flush=True)
sim.set_force_exact(0)
os.environ["RAPID_POOL_EIGHTHS"] = "1"
sim.tally()
rr = sim.count_votes()
print("decided", rr.decided, "cut", rr.cut_size, "votes", rr.votes_winner, "of", len(sc.receivers))
