"""Two workgroups per CU (more waves than one workgroup can hold): tally kernel time by (waves per workgroup, workgroups per CU).
    python scripts/blocks_per_cu.py [config] [reps]"""
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
for w, b in ((15, 1), (16, 1), (12, 2), (11, 2), (10, 2), (8, 3), (8, 2)):
    os.environ["RAPID_TALLY_WAVES"] = str(w)
    os.environ["RAPID_TALLY_BLOCKS_PER_CU"] = str(b)
    sim = E.ClusterSimulation(eng)
    sim.load_streams(sc.records, sc.rec_off)
    sim.set_alert_set(sc.batches.recs, trust_copies=True)
    ms = min(sim.time_tally(reps) for _ in range(3))
    info = sim.index_info()
    sim.set_force_exact(64)
    msf = min(sim.time_tally(reps) for _ in range(2))
    sim.set_force_exact(0)
    print("waves %2d x %d workgroups/CU (grid %d, %d B LDS each): tally %.4f ms, filter per delivery %.4f ms" % (
        info["waves_per_workgroup"], b, info["workgroups"], info["lds_bytes_per_workgroup"], ms, msf), flush=True)
