"""Tally kernel time as a function of the waves per workgroup (RAPID_TALLY_WAVES caps the host's choice):
    python scripts/waves_sweep.py [config] [reps] [w1,w2,...]"""
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
waves = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [8, 10, 12, 13, 14, 16]
spec = S.CONFIGS[name]
n, K, H, L = spec["n"], spec["K"], spec["H"], spec["L"]
pop = S.Population.make(n)
eng = E.Engine(n_max=n, K=K, H=H, L=L)
view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)
obs, subj, member = view.tables()
sc = S.build_scenario(name, subj, view.getCurrentConfigurationId())
nbytes = 20 * len(sc.records)
for w in waves:
    os.environ["RAPID_TALLY_WAVES"] = str(w)
    os.environ["RAPID_TALLY_WAVES_EXACT"] = str(w)  # (exactly that many, not the cost model's choice below the cap)
    sim = E.ClusterSimulation(eng)
    sim.load_streams(sc.records, sc.rec_off)
    sim.set_alert_set(sc.batches.recs, trust_copies=True)
    ms = min(sim.time_tally(reps) for _ in range(3))
    info = sim.index_info()
    sim.set_force_exact(32)
    so = min(sim.time_tally(reps) for _ in range(2))
    sim.set_force_exact(16)
    stag = min(sim.time_tally(reps) for _ in range(2))
    sim.set_force_exact(0)
    print("waves/workgroup %2d (asked %2d)  tally %.4f ms  %.0f GB/s  %.1f %% of 8 TB/s   stream only %.4f ms %.0f GB/s  staggered start %.4f ms" % (
        info["waves_per_workgroup"], w, ms, nbytes / ms / 1e6, 100 * nbytes / ms / 1e6 / 8000, so, nbytes / so / 1e6, stag), flush=True)
