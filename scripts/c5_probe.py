"""The tally at BASELINE configs[4] (10^6 members, ~15,000 hot subjects per round) on one tile: one churn round, its deliveries made on
the device for a sample of receivers (late deliveries of another configuration among them) as 20-byte boundary records and as
resolved 8-byte records, and the tally kernel timed in every prepared build of the library (scripts/build_variants.sh; RAPID_AB_ONLY
picks).  (Round 5 also ran it over timing-only probe builds -- no dictionary gather, no OR, streaming only: profiles/
r05_c5_probe_builds.txt; those probes are gone from the kernel source.)  RAPID_C5_KNOB=<bits>: rapid_sim_set_force_exact for the runs.
    python scripts/c5_probe.py [members=1000000] [receivers=1024] [reps=5]"""
import glob
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
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
K, H, L = 10, 9, 4

libs = {"default": N.TEST_LIB_PATH}
for path in sorted(glob.glob(os.path.join(ROOT, "rapid_amd", "librapid_mi355x_*.so"))):
    tag = os.path.basename(path)[len("librapid_mi355x_"):-3]
    if tag not in ("test",) and not tag.startswith("timers"):
        libs[tag] = path
if os.environ.get("RAPID_AB_ONLY"):
    libs = {k: v for k, v in libs.items() if k in os.environ["RAPID_AB_ONLY"].split(",")}

t0 = time.time()
pop = S.Population.make(n_mem + int(0.006 * n_mem) + 64)
sc = deliver_set = None
rx = None
for tag, path in libs.items():
    N._lib = None
    N.LIB_PATH = path
    eng = E.Engine(n_max=pop.n, K=K, H=H, L=L, max_cut=max(4096, int(0.02 * n_mem)))
    view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo, members=np.arange(n_mem, dtype=np.int32))
    if sc is None:
        obs, subj, member = view.tables()
        cfg = view.getCurrentConfigurationId()
        st = S.StreamingChurn(H, L)
        st.prev_cfg = cfg ^ 0x5A5A  # (a first round has no predecessor: any other id stands in for "the previous configuration")
        sc, deliver_set = st.next_round_batches(obs, member, cfg)
        rng = np.random.Generator(np.random.PCG64(11))
        rx = np.sort(rng.permutation(sc.receivers)[:n_rx]).astype(np.int32)
        print("setup %.1f s: %d members, %d alerts in %d batches (+ %d late), %d receivers" % (
            time.time() - t0, n_mem, len(sc.batches.recs), sc.batches.n_batches, deliver_set.n_batches - sc.batches.n_batches, len(rx)), flush=True)
    sim = E.ClusterSimulation(eng)
    n_rec = len(rx) * len(deliver_set.recs)
    for boundary in (True, False):
        sim.generate(deliver_set, rx, seed=7, trust_copies=True, boundary=boundary)
        rec_b = 20 if boundary else 8
        base = int(os.environ.get("RAPID_C5_KNOB", "0"))
        for knob, what in ((base, "as chosen"), (base | 64, "filter per delivery")):
            sim.set_force_exact(knob)
            ms = sim.time_tally(reps)
            info = sim.index_info(timed=False)
            print("%-8s %-9s %-20s tally %.4f ms  %.3e records/s  %.1f %% of 8 TB/s on %d B  (generated in %.3f ms; dict mode %d, %d waves x %d workgroups, %d hot, prevalidated %d)" % (
                tag, "boundary" if boundary else "resolved", what, ms, n_rec / ms * 1e3, 100 * rec_b * n_rec / ms / 1e6 / 8000, rec_b, info["generate_ms"],
                info["dict_mode"], info["waves_per_workgroup"], info["workgroups"], info["hot_subjects"], info["alerts_prevalidated"]), flush=True)
        sim.set_force_exact(0)
    sim.generate(deliver_set, rx, seed=7, trust_copies=True, boundary=True)
    if tag == "default":
        sim.tally()
        rr = sim.count_votes()
        emit, nprop, pcount, fp = sim.results()
        print("default: %d of %d receivers propose, %d distinct proposals, cut %s the round's fault set" % (
            int((emit >= 0).sum()), len(emit), len(np.unique(fp[emit >= 0])),
            "==" if sorted(sim.proposal(int(np.flatnonzero(emit >= 0)[0]))) == sc.faulty.tolist() else "!="), flush=True)
    del sim
    eng.close()
