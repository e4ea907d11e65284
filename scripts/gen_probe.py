"""Measurement aid: the generator's lay-down time at N = 10^6 (resolved records, 1,024 and 4,096 receivers) in every prepared build
(RAPID_AB_ONLY picks).  Only rapid_sim_generate runs -- probe builds write void records, nothing may tally them."""
import ctypes as C
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rapid_amd import _native as N  # noqa: E402
from rapid_amd import engine as E  # noqa: E402
from rapid_amd import scenarios as S  # noqa: E402

n_mem = 1000000
K, H, L = 10, 9, 4
libs = {"default": N.TEST_LIB_PATH}
for path in sorted(glob.glob(os.path.join(ROOT, "rapid_amd", "librapid_mi355x_gp*.so"))):
    libs[os.path.basename(path)[len("librapid_mi355x_"):-3]] = path
pop = S.Population.make(n_mem + 6064)
sc = None
for tag, path in libs.items():
    N._lib = None
    N.LIB_PATH = path
    eng = E.Engine(n_max=pop.n, K=K, H=H, L=L, max_cut=20000)
    view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo, members=np.arange(n_mem, dtype=np.int32))
    if sc is None:
        obs, subj, member = view.tables()
        cfg = view.getCurrentConfigurationId()
        st = S.StreamingChurn(H, L)
        st.prev_cfg = cfg ^ 0x5A5A
        sc, deliver_set = st.next_round_batches(obs, member, cfg)
    sim = E.ClusterSimulation(eng)
    for n_rx in (1024, 4096):
        rx = sc.receivers[:n_rx].astype(np.int32)
        best = 1e9
        for rep in range(4):
            sim.generate(deliver_set, rx, seed=7 + rep, trust_copies=True, boundary=False)
            t = np.zeros(4, dtype=np.float32)
            eng._check(eng._lib.rapid_sim_pass_times(eng._h, t.ctypes.data_as(C.c_void_p)))
            best = min(best, float(t[2]))
        print("%-8s %5d receivers: lay-down %.3f ms  (%.3e records/s)" % (tag, n_rx, best, n_rx * len(deliver_set.recs) / best * 1e3), flush=True)
    del sim
    eng.close()
