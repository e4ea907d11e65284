import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from rapid_amd import engine as E, scenarios as S
n, K, H, L = 2000, 10, 9, 4
pop = S.Population.make(n)
eng = E.Engine(n_max=n, K=K, H=H, L=L)
view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)
obs, subj, member = view.tables()
sc = S.build_scenario("C2", subj, view.getCurrentConfigurationId(), n=n, f=20, H=H, L=L)
raw = torch.from_numpy(np.ascontiguousarray(sc.records).view(np.uint8).reshape(-1)).cuda()
off = torch.from_numpy(np.ascontiguousarray(sc.rec_off, dtype=np.int64)).cuda()
al = torch.from_numpy(np.ascontiguousarray(sc.batches.recs).view(np.uint8).reshape(-1).copy()).cuda()
sim = E.ClusterSimulation(eng)
R = len(sc.rec_off) - 1
for name, fn in (("attach", lambda: sim.attach_streams_device(raw.data_ptr(), raw.numel(), off.data_ptr(), R)),
                 ("declare", lambda: sim.set_alert_set_device(al.data_ptr(), len(sc.batches.recs), trust_copies=True))):
    fn()
    t0 = time.perf_counter()
    for _ in range(20000):
        fn()
    print(name, "%.2f us per call" % (1e6 * (time.perf_counter() - t0) / 20000))
