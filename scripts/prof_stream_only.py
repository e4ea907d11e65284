"""Measurement aid: the tally kernel in stream-only mode (flag 32) under rocprofv3."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rapid_amd import engine as E, scenarios as S
n,K,H,L=10000,10,9,4
pop=S.Population.make(n)
eng=E.Engine(n_max=n,K=K,H=H,L=L)
view=E.MembershipView(eng).build(pop.hostnames,pop.ports,pop.id_hi,pop.id_lo)
obs,subj,member=view.tables(); cfg=view.getCurrentConfigurationId()
sc=S.build_scenario("C3b",subj,cfg)
sim=E.ClusterSimulation(eng); sim.load_streams(sc.records, sc.rec_off); sim.set_alert_set(sc.batches.recs, trust_copies=True)
sim.set_force_exact(32)
print("stream_only ms", sim.time_tally(3), sim.index_info())
