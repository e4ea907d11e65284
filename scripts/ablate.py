import sys, time, json
import numpy as np
sys.path.insert(0, '/root/repo')
from rapid_amd import engine as E, scenarios as S
from rapid_amd import _native as _N  # noqa: E402

_N.use_test_build()  # measurement aids: environment knobs, probes and rapid_debug_* exist in the test build only
n,K,H,L=10000,10,9,4
pop=S.Population.make(n)
eng=E.Engine(n_max=n,K=K,H=H,L=L)
view=E.MembershipView(eng).build(pop.hostnames,pop.ports,pop.id_hi,pop.id_lo)
obs,subj,member=view.tables(); cfg=view.getCurrentConfigurationId()
sc=S.build_scenario("C3b",subj,cfg)
sim=E.ClusterSimulation(eng); sim.load_streams(sc.records, sc.rec_off); sim.set_alert_set(sc.batches.recs, trust_copies=True)
for flag,name in [(0,"full"),(64,"full_untrusted"),(32,"stream_only")]:
    sim.set_force_exact(flag)
    ms=sim.time_tally(10)
    print(name, round(ms,3),"ms", round(20*len(sc.records)/ms/1e6,1),"GB/s", sim.stats())

sim.set_force_exact(0)
gb = 20*len(sc.records)/1e6
for variant,name in [(0,"2KiBx8"),(3,"2KiBx4"),(1,"4KiBx4"),(2,"8KiBx2")]:
    for waves in (4, 8, 16):
        ms = sim.stream_probe(variant, waves, 10)
        print("probe", name, "waves/block", waves, round(ms,3), "ms", round(gb/ms,1), "GB/s")
