"""C3b step and tally kernel with the fast round settled inside the tally launch (opt-in, knob bit 23) against the default (vote
statistics in the tally, verification in a launch of its own), interleaved.
    python scripts/step_ab_settle.py [steps]"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch  # noqa: E402

from rapid_amd import engine as E, scenarios as S  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
spec = S.CONFIGS["C3b"]
n, K, H, L = spec["n"], spec["K"], spec["H"], spec["L"]
pop = S.Population.make(n)
eng = E.Engine(n_max=n, K=K, H=H, L=L)
view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)
obs, subj, member = view.tables()
sc = S.build_scenario("C3b", subj, view.getCurrentConfigurationId(), materialise=False)
alert_set = np.ascontiguousarray(sc.batches.recs)
d_al = torch.from_numpy(alert_set.view(np.uint8).reshape(-1).copy()).cuda()
recs, off, nb = S.deliver(sc.batches, sc.receivers, seed_delivery=2)
d_rec = torch.from_numpy(recs.view(np.uint8).reshape(-1)).cuda()
d_off = torch.from_numpy(np.ascontiguousarray(off, dtype=np.int64)).cuda()
sim = E.ClusterSimulation(eng)
R = len(off) - 1


def one(knob, one_call):
    sim.set_force_exact(knob)

    def step():
        if one_call:
            rr, _ = sim.round_device(d_rec.data_ptr(), d_rec.numel(), d_off.data_ptr(), R, d_al.data_ptr(), len(alert_set), trust=1, keepalive=(d_rec, d_off, d_al))
            return rr
        sim.attach_streams_device(d_rec.data_ptr(), d_rec.numel(), d_off.data_ptr(), R, keepalive=(d_rec, d_off))
        sim.set_alert_set_device(d_al.data_ptr(), len(alert_set), trust_copies=True, keepalive=d_al)
        sim.tally()
        return sim.count_votes()

    for _ in range(5):
        rr = step()
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        rr = step()
    eng.sync()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    sim.attach_streams_device(d_rec.data_ptr(), d_rec.numel(), d_off.data_ptr(), R, keepalive=(d_rec, d_off))
    sim.set_alert_set_device(d_al.data_ptr(), len(alert_set), trust_copies=True, keepalive=d_al)
    k = sim.time_tally(10)
    return ms, k, rr


for rep in range(3):
    a = one(8388608, True)
    b = one(8388608, False)
    c = one(0, False)
    assert (a[2].decided, a[2].votes_winner, a[2].cut_size) == (c[2].decided, c[2].votes_winner, c[2].cut_size) == (b[2].decided, b[2].votes_winner, b[2].cut_size)
    print("rep %d  settled in the tally, one call: step %.4f ms kernel %.4f | five calls: step %.4f kernel %.4f | default (own launches), five calls: step %.4f kernel %.4f  (decided %d, %d votes)" %
          (rep, a[0], a[1], b[0], b[1], c[0], c[1], a[2].decided, a[2].votes_winner), flush=True)
