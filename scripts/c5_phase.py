"""Where a receiver's cycles go at BASELINE configs[4] (10^6 members, ~15,000 hot subjects): the tally kernel's event counters per
receiver (test build) and the phase timers of the -DRAPID_PHASE_TIMERS build (scripts/build_variants.sh timers "-DRAPID_PHASE_TIMERS").
    python scripts/c5_phase.py [members=1000000] [receivers=1024] [boundary|resolved]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rapid_amd import _native as N  # noqa: E402
from rapid_amd import engine as E  # noqa: E402
from rapid_amd import scenarios as S  # noqa: E402

n_mem = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
n_rx = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
boundary = not (len(sys.argv) > 3 and sys.argv[3] == "resolved")
K, H, L = 10, 9, 4
pop = S.Population.make(n_mem + int(0.006 * n_mem) + 64)
sc = deliver_set = rx = None
for tag, path in (("counters", N.TEST_LIB_PATH), ("timers", os.path.join(ROOT, "rapid_amd", "librapid_mi355x_timers.so"))):
    if not os.path.exists(path):
        continue
    N._lib = None
    N.LIB_PATH = path
    eng = E.Engine(n_max=pop.n, K=K, H=H, L=L, max_cut=max(4096, int(0.02 * n_mem)))
    view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo, members=np.arange(n_mem, dtype=np.int32))
    if sc is None:
        obs, subj, member = view.tables()
        cfg = view.getCurrentConfigurationId()
        st = S.StreamingChurn(H, L)
        st.prev_cfg = cfg ^ 0x5A5A
        sc, deliver_set = st.next_round_batches(obs, member, cfg)
        rx = np.sort(np.random.Generator(np.random.PCG64(11)).permutation(sc.receivers)[:n_rx]).astype(np.int32)
    sim = E.ClusterSimulation(eng)
    sim.generate(deliver_set, rx, seed=7, trust_copies=True, boundary=boundary)
    reps = 3
    ms = sim.time_tally(reps)
    s = np.zeros(8, dtype=np.uint64)
    eng._check(eng._lib.rapid_sim_stats(eng._h, E._addr(s)))
    runs = reps + 1
    info = sim.index_info(timed=False)
    nwin = (len(deliver_set.recs) + 255) // 256
    print(tag, "boundary" if boundary else "resolved", "tally_ms", round(ms, 4), "windows per receiver", nwin, info, flush=True)
    if tag == "counters":
        names = ("exact_subchunks", "lean_windows", "full_sweeps", "restarts", "implicit_reports", "records_consumed", "lean_give_ups", "careful_subchunks")
        print({k: round(float(v) / runs / len(rx), 2) for k, v in zip(names, s)}, flush=True)
    else:
        tot = float(s[0])
        names = ["total", "fast windows, steady-state loop", "cold + fast windows, general iteration", "slow windows", "output+init", "flush+sweeps"]
        for i, nm in enumerate(names):
            print("%-42s %6.1f %%   %.0f cycles/receiver" % (nm, 100.0 * float(s[i]) / tot, float(s[i]) / max(1.0, float(s[6]))))
        print("windows taken by the steady-state loop per receiver", float(s[7]) / max(1.0, float(s[6])))
    del sim
    eng.close()
