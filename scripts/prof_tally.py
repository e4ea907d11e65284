"""Loads one BASELINE workload and runs the alert-tally kernel a few times -- the command rocprofv3 wraps for the
per-kernel statistics and PMC passes committed under profiles/.  Also runs the streaming probe (same access pattern,
no processing, known byte count) that calibrates FETCH_SIZE."""
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from rapid_amd import engine as E, scenarios as S  # noqa: E402
from rapid_amd import _native as _N  # noqa: E402

_N.use_test_build()  # measurement aids: environment knobs, probes and rapid_debug_* exist in the test build only

name = sys.argv[1] if len(sys.argv) > 1 else "C3b"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
spec = S.CONFIGS[name]
n, K, H, L = spec["n"], spec["K"], spec["H"], spec["L"]
pop = S.Population.make(n)
eng = E.Engine(n_max=n, K=K, H=H, L=L)
view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)
obs, subj, member = view.tables()
cfg = view.getCurrentConfigurationId()
sc = S.build_scenario(name, subj, cfg)
sim = E.ClusterSimulation(eng)
sim.load_streams(sc.records, sc.rec_off)
sim.set_alert_set(sc.batches.recs, trust_copies=True)
ms = sim.time_tally(reps)
st = sim.stats()
probe = sim.stream_probe(0, 16, reps)
print("workload", name, "records", len(sc.records), "stream_bytes", 20 * len(sc.records), "tally_ms", round(ms, 4), "GB/s",
      round(20 * len(sc.records) / ms / 1e6, 1), "probe_ms", round(probe, 4), st, sim.index_info())
