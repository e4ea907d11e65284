"""Profiling build only (as scripts/phase_timers.sh builds it, plus -DRAPID_BLOCK_STAMPS): when does every workgroup of ONE tally launch start and finish?
    RAPID_MI355X_LIB=.../librapid_mi355x_timers.so python scripts/block_times.py [config]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rapid_amd import engine as E  # noqa: E402
from rapid_amd import scenarios as S  # noqa: E402
from rapid_amd import _native as _N  # noqa: E402

_N.use_test_build()  # measurement aids: environment knobs, probes and rapid_debug_* exist in the test build only

name = sys.argv[1] if len(sys.argv) > 1 else "C3b"
spec = S.CONFIGS[name]
n, K, H, L = spec["n"], spec["K"], spec["H"], spec["L"]
pop = S.Population.make(n)
eng = E.Engine(n_max=n, K=K, H=H, L=L)
view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)
obs, subj, member = view.tables()
sc = S.build_scenario(name, subj, view.getCurrentConfigurationId())
sim = E.ClusterSimulation(eng)
sim.load_streams(sc.records, sc.rec_off)
sim.set_alert_set(sc.batches.recs, trust_copies=True)
sim.tally()
eng.sync()
for trial in range(2):
    ms = sim.time_tally(1)
    sim.tally()  # zeroes the counters, one launch
    eng.sync()
    out = np.zeros((1024, 8), dtype=np.uint64)
    rows = C.c_int32(0)
    eng._check(eng._lib.rapid_debug_block_stats(eng._h, out.ctypes.data, 1024, C.byref(rows)))
    b = out[: rows.value].astype(np.float64)
    start, end = b[:, 7] / 100.0, b[:, 1] / 100.0  # microseconds (100 MHz counter)
    t0 = start.min()
    dur = end - start
    print("trial %d: tally %.4f ms; workgroups %d; first start 0, last start %.1f us; end: min %.1f median %.1f p90 %.1f max %.1f us; "
          "duration: min %.1f median %.1f max %.1f us" % (trial, ms, rows.value, start.max() - t0, (end - t0).min(), np.median(end - t0),
                                                         np.percentile(end - t0, 90), (end - t0).max(), dur.min(), np.median(dur), dur.max()))
    xcd = np.arange(rows.value) % 8
    print("   end time by XCD (block %% 8): " + " ".join("%.0f" % (end[xcd == x] - t0).mean() for x in range(8)),
          "| receivers/block min %d max %d" % (b[:, 6].min(), b[:, 6].max()))
