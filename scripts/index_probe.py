"""Device time of the round index build (touch + build kernels) at a BASELINE configuration, with and without the observer memo
(quirk Q4): python scripts/index_probe.py [config]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rapid_amd import engine as E  # noqa: E402
from rapid_amd import scenarios as S  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3b"
spec = S.CONFIGS[name]
n, K, H, L = spec["n"], spec["K"], spec["H"], spec["L"]
pop = S.Population.make(n)
eng = E.Engine(n_max=n, K=K, H=H, L=L)
view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)
obs, subj, member = view.tables()
sc = S.build_scenario(name, subj, view.getCurrentConfigurationId(), receivers=list(range(64)) if n > 20000 else None, materialise=False)
sim = E.ClusterSimulation(eng)
sim.generate(sc.batches, sc.receivers[:256], seed=1, boundary=True)
for q4 in (True, False, True):
    view.setObserverCacheEmulation(q4)
    ts = []
    for _ in range(8):
        sim.new_round()
        ts.append(sim.index_info()["index_build_ms"])
    print("%s q4 memo %s: index build (touch + build kernels) %s ms" % (name, "on " if q4 else "off", " ".join("%.4f" % t for t in ts)), flush=True)
