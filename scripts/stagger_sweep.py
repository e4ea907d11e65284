"""Tally kernel time as a function of the late start of every other wave (RAPID_TALLY_STAGGER x 8,128 cycles):
    python scripts/stagger_sweep.py [config] [reps] [s1,s2,...]"""
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
vals = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 2, 4, 6, 8, 10, 12, 0]
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
for v in vals:
    os.environ["RAPID_TALLY_STAGGER"] = str(v)
    ms = min(sim.time_tally(reps) for _ in range(3))
    print("stagger %2d x 8128 cycles: tally %.4f ms  (%.1f %% of 8 TB/s on 8 B per record)" % (v, ms, 100 * 8 * len(sc.records) / ms / 1e6 / 8000), flush=True)
