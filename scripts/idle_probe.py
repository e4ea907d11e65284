"""Does idle time between two receivers cost kernel time?  Stream-only mode with and without ~35 k idle cycles per receiver,
for several wave counts:  python scripts/idle_probe.py [config] [reps]"""
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
for w in (8, 10, 13, 16):
    os.environ["RAPID_TALLY_WAVES"] = str(w)
    sim = E.ClusterSimulation(eng)
    sim.load_streams(sc.records, sc.rec_off)
    sim.set_alert_set(sc.batches.recs, trust_copies=True)
    out = []
    for knob in (32, 32 | 4, 32 | 16, 0, 4):
        sim.set_force_exact(knob)
        out.append(min(sim.time_tally(reps) for _ in range(2)))
    print("waves %2d: stream only %.4f  + idle/receiver %.4f  + staggered %.4f | tally %.4f  tally + idle/receiver %.4f ms" % (
        sim.index_info()["waves_per_workgroup"], *out), flush=True)
