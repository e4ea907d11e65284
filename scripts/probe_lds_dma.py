import sys, os
sys.path.insert(0, os.getcwd())
from rapid_amd import engine as E, scenarios as S
from rapid_amd import _native as _N  # noqa: E402

_N.use_test_build()  # measurement aids: environment knobs, probes and rapid_debug_* exist in the test build only
n,K,H,L=10000,10,9,4
pop=S.Population.make(n)
eng=E.Engine(n_max=n,K=K,H=H,L=L)
view=E.MembershipView(eng).build(pop.hostnames,pop.ports,pop.id_hi,pop.id_lo)
obs,subj,member=view.tables(); cfg=view.getCurrentConfigurationId()
sc=S.build_scenario("C3b",subj,cfg)
sim=E.ClusterSimulation(eng); sim.load_streams(sc.records, sc.rec_off)
gb = 20*len(sc.records)/1e6
w=int(sys.argv[1])
ms = sim.stream_probe(9, w, 5)
print("LDS-DMA 1KiBx6, waves/block", w, "waves/CU", os.environ.get("RAPID_PROBE_WAVES_PER_CU","16"), "pad", os.environ.get("RAPID_PROBE_LDS_PAD","0"), round(ms,3), "ms", round(gb/ms,1), "GB/s")
