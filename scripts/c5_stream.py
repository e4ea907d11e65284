"""BASELINE configs[4] on one MI355X at a single-GPU-sized N: consecutive rounds of continuous churn (1 % crashes + 0.5 % joins
per round, 1 % of the delivered records stale), the DECIDED cut applied after every round (the script stops if a round has no
fast-round decision).  Prints per round: records, the tally kernel time, the whole round from the boundary (the round's 20-byte
records attached in place, alert set declared, index + tally + votes) and the view change.
    python scripts/c5_stream.py [members=100000] [rounds=5] [receivers_per_round=4000]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rapid_amd import engine as E  # noqa: E402
from rapid_amd import scenarios as S  # noqa: E402

n_mem = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n_rx = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
K, H, L = 10, 9, 4
pop = S.Population.make(n_mem + int(0.006 * n_mem * rounds) + 64)
eng = E.Engine(n_max=pop.n, K=K, H=H, L=L, max_cut=max(4096, int(0.02 * n_mem)))  # a round's cut: ~1.5 % of the members
view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo, members=list(range(n_mem)))
st = S.StreamingChurn(H, L, receivers_per_round=n_rx)
guard = E.ObserverCacheGuard()
sim = E.ClusterSimulation(eng)
for rnd in range(rounds):
    obs, subj, member = view.tables()
    cfg = view.getCurrentConfigurationId()
    sc = st.next_round(obs, member, cfg)
    at_risk = guard.check_round(view, sc.faulty)
    d_rec = torch.from_numpy(np.ascontiguousarray(sc.records).view(np.uint8).reshape(-1)).cuda()
    d_off = torch.from_numpy(np.ascontiguousarray(sc.rec_off, dtype=np.int64)).cuda()
    torch.cuda.synchronize()
    sim.attach_streams_device(d_rec.data_ptr(), d_rec.numel(), d_off.data_ptr(), len(sc.rec_off) - 1, keepalive=(d_rec, d_off))
    sim.set_alert_set(sc.batches.recs, trust_copies=True)  # (late deliveries of the previous configuration are dropped per delivery)
    kern_ms = sim.time_tally(5)
    eng.sync()
    t = time.perf_counter()
    sim.attach_streams_device(d_rec.data_ptr(), d_rec.numel(), d_off.data_ptr(), len(sc.rec_off) - 1, keepalive=(d_rec, d_off))
    sim.set_alert_set(sc.batches.recs, trust_copies=True)
    sim.tally()
    rr = sim.count_votes()
    round_ms = 1e3 * (time.perf_counter() - t)
    emit, nprop, pcount, fp = sim.results()
    voters = np.flatnonzero(emit >= 0)
    if rr.decided:
        cut, how = sim.decided_cut(), "fast-round quorum"
    elif len(voters) and len(np.unique(fp[voters])) == 1:
        # N - F votes need the whole population; the simulated receivers are a sample of it.  The sample is unanimous: its
        # proposal is applied (what the rest of the population, holding the same alerts, would vote for).
        cut, how = sim.proposal(int(voters[0])), "unanimous sample (%d of %d receivers propose; quorum %d needs the whole population)" % (
            len(voters), len(emit), int(rr.quorum))
    else:
        print(json.dumps({"round": rnd, "decided": 0, "proposing": int(len(voters)), "distinct_proposals": int(len(np.unique(fp[voters]))),
                          "note": "no decision and no unanimous sample: stopping"}), flush=True)
        break
    assert sorted(cut) == sc.faulty.tolist(), "the cut is not the round's fault set"
    cut_arr = np.ascontiguousarray(cut, dtype=np.int32)  # (what the C ABI takes: a Python list of 15,000 ints costs 0.4 ms to convert)
    t = time.perf_counter()
    new_cfg = sim.apply_cut(cut_arr)
    eng.sync()
    apply_ms = 1e3 * (time.perf_counter() - t)
    guard.on_view_change(sc.crashed)
    info = sim.index_info()
    print(json.dumps({"round": rnd, "members": int((member != 0).sum()), "receivers": len(sc.receivers), "records": int(len(sc.records)),
                      "stale_records": int((sc.records["cfg_id"] != cfg).sum()), "cut": len(cut), "crashed": len(sc.crashed),
                      "joined": len(sc.joiners), "proposing": int((emit >= 0).sum()), "votes_winner": int(rr.votes_winner),
                      "kernel_ms": round(kern_ms, 4), "kernel_records_per_s": round(len(sc.records) / kern_ms * 1e3, 1),
                      "kernel_frac_of_8TBps": round(20 * len(sc.records) / kern_ms / 1e6 / 8000, 4),  # 20 B per delivered record
                      "round_from_boundary_ms": round(round_ms, 3), "waves_per_workgroup": info["waves_per_workgroup"],
                      "round_records_per_s": round(len(sc.records) / round_ms * 1e3, 1), "decided": int(rr.decided), "cut_from": how,
                      "apply_cut_ms": round(apply_ms, 3), "dict_mode": info["dict_mode"], "hot_subjects": info["hot_subjects"],
                      "q4_at_risk": at_risk, "config_id": int(new_cfg)}), flush=True)
