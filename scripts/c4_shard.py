"""BASELINE configs[3] (N = 100,000, K = 10, 1 % crashed) as ONE of eight ranks sees it: its shard of the receivers on one
MI355X, the 20-byte records tallied where they lie (compressed tables in LDS).  Prints the tally kernel time of both
instantiations, the roofline fraction on the 20 B per record, and a whole round.
    python scripts/c4_shard.py [ranks=8] [reps=10]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rapid_amd import engine as E  # noqa: E402
from rapid_amd import parallel as P  # noqa: E402
from rapid_amd import scenarios as S  # noqa: E402

ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
spec = S.CONFIGS["C4"]
n, K, H, L = spec["n"], spec["K"], spec["H"], spec["L"]
t0 = time.time()
pop = S.Population.make(n)
eng = E.Engine(n_max=n, K=K, H=H, L=L)
view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)
obs, subj, member = view.tables()
cfg = view.getCurrentConfigurationId()
sc = S.build_scenario("C4", subj, cfg, materialise=False)
lo, hi = P.shard_range(len(sc.receivers), 0, ranks)
records, rec_off, nb = S.deliver(sc.batches, sc.receivers[lo:hi], seed_delivery=2)
sim = E.ClusterSimulation(eng)
sim.load_streams(records, rec_off)
out = {"workload": "C4 shard 0 of %d: N=%d K=%d, %d crashed, %d receivers, %d records" % (ranks, n, K, len(sc.faulty), hi - lo, len(records)),
       "setup_s": round(time.time() - t0, 1)}
nbytes = 20 * len(records)
for tag, declare in (("filter_per_delivery", False), ("alert_set_declared", True)):
    if declare:
        sim.set_alert_set(sc.batches.recs, trust_copies=True)
    sim.index_info()  # (builds the index, and asks for the device time of the next build)
    sim.new_round()
    ms = min(sim.time_tally(reps) for _ in range(2))
    info = sim.index_info(timed=False)
    out[tag] = {"kernel_ms": round(ms, 4), "GBps": round(nbytes / ms / 1e6, 1), "frac_of_8TBps": round(nbytes / ms / 1e6 / 8000, 4),
                "dict_mode": info["dict_mode"], "waves_per_workgroup": info["waves_per_workgroup"],
                "index_build_ms": round(info["index_build_ms"], 4)}
t = time.perf_counter()
sim.new_round()
sim.tally()
rr = sim.count_votes()
out["round_ms"] = round(1e3 * (time.perf_counter() - t), 3)
out["votes_winner"] = int(rr.votes_winner)
out["stats"] = sim.stats()
print(json.dumps(out))
