"""What a round's collective costs before any second GPU exists: the C3b step (fresh round: attach, declare, index, tally, vote
count) on ONE GPU without a communicator, with a 1-rank RCCL communicator (ONE ncclAllGather of the answer block + the merge
kernel), and with the general count forced (knob 512: the all-reduce of the 128 KiB vote histogram + the small reductions) --
the per-round floor RCCL's launch path adds, whatever the link.  Also: the step at 1/2, 1/4, 1/8 of the receivers (what a
rank of a 2-, 4-, 8-GPU run carries), so that the 1 -> 8 curve of DESIGN.md section 4 is a sum of measured terms.
    python scripts/collective_latency.py [steps]   ->  profiles/rNN_collective_latency.txt"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch  # noqa: E402

from rapid_amd import engine as E, scenarios as S, parallel as P  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
spec = S.CONFIGS["C3b"]
n, K, H, L = spec["n"], spec["K"], spec["H"], spec["L"]
pop = S.Population.make(n)


def make(comm):
    eng = E.Engine(n_max=n, K=K, H=H, L=L)
    if comm:
        eng.comm_init(E.comm_unique_id(), 0, 1)
    view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)
    return eng, view


eng, view = make(False)
obs, subj, member = view.tables()
cfg = view.getCurrentConfigurationId()
sc = S.build_scenario("C3b", subj, cfg, materialise=False)
alert_set = np.ascontiguousarray(sc.batches.recs)
d_al = torch.from_numpy(alert_set.view(np.uint8).reshape(-1).copy()).cuda()


def resident(shards):
    lo, hi = P.shard_range(len(sc.receivers), 0, shards)
    recs, off, nb = S.deliver(sc.batches, sc.receivers[lo:hi], seed_delivery=2)
    d_rec = torch.from_numpy(recs.view(np.uint8).reshape(-1)).cuda()
    d_off = torch.from_numpy(np.ascontiguousarray(off, dtype=np.int64)).cuda()
    return d_rec, d_off, len(off) - 1, len(recs)


def time_steps(eng, data, knob=0):
    d_rec, d_off, n_rx, n_rec = data
    sim = E.ClusterSimulation(eng)
    sim.set_force_exact(knob)

    def step():
        sim.attach_streams_device(d_rec.data_ptr(), d_rec.numel(), d_off.data_ptr(), n_rx, keepalive=data)
        sim.set_alert_set_device(d_al.data_ptr(), len(alert_set), trust_copies=True, keepalive=d_al)
        sim.tally()
        return sim.count_votes()

    for _ in range(5):
        rr = step()
    eng.sync()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            rr = step()
        eng.sync()
        best = min(best, (time.perf_counter() - t0) / steps)
    sim.set_force_exact(0)
    return 1e3 * best, rr


full = resident(1)
ms0, rr0 = time_steps(eng, full)
print("C3b step, no communicator:                          %.4f ms  (decided %d, %d votes)" % (ms0, rr0.decided, rr0.votes_winner))
eng_c, view_c = make(True)
ms1, rr1 = time_steps(eng_c, full)
print("C3b step, 1-rank communicator (all-gather + merge): %.4f ms  (+%.1f us)" % (ms1, 1e3 * (ms1 - ms0)))
ms2, rr2 = time_steps(eng_c, full, knob=512)
print("C3b step, 1-rank communicator, general count:       %.4f ms  (+%.1f us)" % (ms2, 1e3 * (ms2 - ms0)))
assert (rr1.decided, rr1.votes_winner) == (rr0.decided, rr0.votes_winner) == (rr2.decided, rr2.votes_winner)
for shards in (2, 4, 8):
    data = resident(shards)
    ms, rr = time_steps(eng_c, data)
    sim = E.ClusterSimulation(eng_c)
    sim.attach_streams_device(data[0].data_ptr(), data[0].numel(), data[1].data_ptr(), data[2], keepalive=data)
    sim.set_alert_set_device(d_al.data_ptr(), len(alert_set), trust_copies=True, keepalive=d_al)
    kms = min(sim.time_tally(10) for _ in range(3))
    info = sim.index_info(timed=False)
    print("1/%d of the receivers (%5d, %8d records), 1-rank communicator: %.4f ms per step; tally kernel %.4f ms, %d waves x %d workgroups" %
          (shards, data[2], data[3], ms, kms, info["waves_per_workgroup"], info["workgroups"]))
    del data
