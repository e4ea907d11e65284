"""How much of the tally kernel's time is ramp-up, tail and uneven last rounds?  C3b's alert set delivered (on the device,
20-byte boundary records) to populations of different sizes -- exact multiples of waves x CUs and the real 9,492 -- and the
kernel time per receiver-round:  python scripts/rounds_probe.py [reps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rapid_amd import engine as E  # noqa: E402
from rapid_amd import scenarios as S  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
spec = S.CONFIGS["C3b"]
n, K, H, L = spec["n"], spec["K"], spec["H"], spec["L"]
pop = S.Population.make(n)
eng = E.Engine(n_max=n, K=K, H=H, L=L)
view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)
obs, subj, member = view.tables()
sc = S.build_scenario("C3b", subj, view.getCurrentConfigurationId(), materialise=False)
A = int(sc.batches.off[-1])
sim = E.ClusterSimulation(eng)
for R in (3840, 7680, 9492, 11520, 15360, 18984):
    rx = np.resize(sc.receivers, R)  # (receivers repeated: equal streams, equal work)
    sim.generate(sc.batches, rx, seed=2, trust_copies=True, boundary=True)
    ms = min(sim.time_tally(reps) for _ in range(3))
    info = sim.index_info()
    w = info["waves_per_workgroup"]
    print("receivers %6d = %.2f rounds of %d waves x %d CUs : tally %.4f ms, %.1f %% of 8 TB/s, %.2f us per receiver and CU-wave" % (
        R, R / (w * info["workgroups"]), w, info["workgroups"], ms, 100 * 20.0 * R * A / ms / 1e6 / 8000, 1e3 * ms / (R / (w * info["workgroups"]))), flush=True)
