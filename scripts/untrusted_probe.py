"""Tally kernel time with the alert set declared (pre-validated instantiation) and with the per-delivery filter forced on
(knob 64), for every build present:  python scripts/untrusted_probe.py [config] [reps]"""
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rapid_amd import _native as N  # noqa: E402
from rapid_amd import engine as E  # noqa: E402
from rapid_amd import scenarios as S  # noqa: E402
from rapid_amd import _native as _N  # noqa: E402

_N.use_test_build()  # measurement aids: environment knobs, probes and rapid_debug_* exist in the test build only

name = sys.argv[1] if len(sys.argv) > 1 else "C3b"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
spec = S.CONFIGS[name]
n, K, H, L = spec["n"], spec["K"], spec["H"], spec["L"]
pop = S.Population.make(n)
libs = {"default": os.path.join(ROOT, "rapid_amd", "librapid_mi355x.so")}
for path in sorted(glob.glob(os.path.join(ROOT, "rapid_amd", "librapid_mi355x_*.so"))):
    tag = os.path.basename(path)[len("librapid_mi355x_"):-3]
    if not tag.startswith("timers"):
        libs[tag] = path
sc = None
for rnd in range(2):
    for tag, path in libs.items():
        N._lib = None
        N.LIB_PATH = path
        eng = E.Engine(n_max=n, K=K, H=H, L=L)
        view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)
        if sc is None:
            obs, subj, member = view.tables()
            sc = S.build_scenario(name, subj, view.getCurrentConfigurationId())
        sim = E.ClusterSimulation(eng)
        sim.load_streams(sc.records, sc.rec_off)
        sim.set_alert_set(sc.batches.recs, trust_copies=True)
        out = []
        for knob in (0, 64, 32, 32 | 64):
            sim.set_force_exact(knob)
            out.append(min(sim.time_tally(reps) for _ in range(2)))
        print("%-8s waves %2d: tally %.4f  filter per delivery %.4f | stream only %.4f  with the configuration id %.4f ms" % (
            tag, sim.index_info()["waves_per_workgroup"], *out), flush=True)
        eng.close()
