import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rapid_amd import engine as E, scenarios as S, _native as N
N.use_test_build()
spec = S.CONFIGS["C3b"]
n, K, H, L = spec["n"], spec["K"], spec["H"], spec["L"]
pop = S.Population.make(n)
eng = E.Engine(n_max=n, K=K, H=H, L=L)
view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)
obs, subj, member = view.tables()
sc = S.build_scenario("C3b", subj, view.getCurrentConfigurationId())
sim = E.ClusterSimulation(eng)
sim.load_streams(sc.records, sc.rec_off)
sim.set_alert_set(sc.batches.recs, trust_copies=True)
for rep in range(2):
    for skip in (0, 1, 2, 3, 4, 7):
        os.environ["RAPID_SETTLE_SKIP"] = str(skip)
        ms = min(sim.time_tally(10) for _ in range(3))
        print("skip %d (1 = no per-receiver note, 2 = no workgroup-end comparison, 4 = no deferred list): tally %.4f ms" % (skip, ms), flush=True)
    os.environ.pop("RAPID_SETTLE_SKIP")
    sim.set_force_exact(8388608)
    print("round-5 form: tally %.4f ms" % min(sim.time_tally(10) for _ in range(3)), flush=True)
    sim.set_force_exact(0)
