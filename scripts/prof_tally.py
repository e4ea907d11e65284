"""Loads the C3b workload once and runs the tally kernel a few times (for rocprofv3 runs)."""
import sys
import numpy as np
sys.path.insert(0, '/root/repo')
from rapid_amd import engine as E, scenarios as S
name = sys.argv[1] if len(sys.argv) > 1 else "C3b"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n, K, H, L = 10000, 10, 9, 4
pop = S.Population.make(n)
eng = E.Engine(n_max=n, K=K, H=H, L=L)
view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)
obs, subj, member = view.tables()
cfg = view.getCurrentConfigurationId()
sc = S.build_scenario(name, subj, cfg)
sim = E.ClusterSimulation(eng)
sim.load_streams(sc.records, sc.rec_off)
ms = sim.time_tally(reps)
print(name, "records", len(sc.records), "tally_ms", round(ms, 4), "GB/s", round(20 * len(sc.records) / ms / 1e6, 1), sim.stats())
