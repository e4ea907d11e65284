"""One round = new_round + tally + count_votes (bench.py's step), timed in one process under different knobs of
rapid_sim_set_force_exact (0 = product; 2048 = vote count with its own counting pass; 1024 = no common pool; 131072 = round
index by the two-kernel form; 262144 = vote verification on node lists instead of slot bitmaps):
    python scripts/step_ab.py [config] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rapid_amd import engine as E  # noqa: E402
from rapid_amd import scenarios as S  # noqa: E402
from rapid_amd import _native as _N  # noqa: E402

_N.use_test_build()  # measurement aids: environment knobs, probes and rapid_debug_* exist in the test build only

name = sys.argv[1] if len(sys.argv) > 1 else "C3b"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
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
for rnd in range(3):
    for knob in (0, 262144, 131072, 2048, 1024, 0):
        sim.set_force_exact(knob)
        for _ in range(5):
            sim.new_round(); sim.tally(); rr = sim.count_votes()
        eng.sync()
        ts = []
        for _ in range(steps):
            t = time.perf_counter()
            sim.new_round(); sim.tally(); rr = sim.count_votes()
            ts.append(time.perf_counter() - t)
        ts.sort()
        print("round %d knob %6d: step min %.4f median %.4f mean %.4f ms  decided %d votes %d" % (
            rnd, knob, 1e3 * ts[0], 1e3 * ts[len(ts) // 2], 1e3 * sum(ts) / len(ts), rr.decided, rr.votes_winner), flush=True)
