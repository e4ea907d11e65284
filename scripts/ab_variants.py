"""One process, one scenario, every prepared build of the library: tally kernel time of each variant, interleaved over
several rounds (box noise is ~5 %), plus the kernel's event counters.
    python scripts/ab_variants.py [config] [rounds] [reps] -- variants are the rapid_amd/librapid_mi355x_<name>.so present."""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rapid_amd import _native as N  # noqa: E402
from rapid_amd import engine as E  # noqa: E402
from rapid_amd import scenarios as S  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3b"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
spec = S.CONFIGS[name]
n, K, H, L = spec["n"], spec["K"], spec["H"], spec["L"]

libs = {"default": os.path.join(ROOT, "rapid_amd", "librapid_mi355x.so")}
for path in sorted(glob.glob(os.path.join(ROOT, "rapid_amd", "librapid_mi355x_*.so"))):
    tag = os.path.basename(path)[len("librapid_mi355x_"):-3]
    if not tag.startswith("timers"):
        libs[tag] = path
if os.environ.get("RAPID_AB_ONLY"):  # comma-separated subset of the variants
    keep = os.environ["RAPID_AB_ONLY"].split(",")
    libs = {k: v for k, v in libs.items() if k in keep}


# environment knobs that belong to a variant (read by the test build when the round's launch geometry is chosen)
VARIANT_ENV = {"occ5": {"RAPID_TALLY_WAVES": "10", "RAPID_TALLY_BLOCKS_PER_CU": "2"}}


def _load_lenient(path, with_test_symbols):
    """A build of another round lacks the entry points added since: bind what it has (this script calls the old ones only)."""
    import ctypes as C
    L = C.CDLL(path)
    for name_, (res, args) in dict(N.SIGNATURES, **N.TEST_SIGNATURES).items():
        f = getattr(L, name_, None)
        if f is not None:
            f.restype, f.argtypes = res, args
    return L


N._load = _load_lenient


def use(path):
    """Point the loader at another build: engines created from now on come from it (each .so carries its own kernels)."""
    N._lib = None
    N.LIB_PATH = path


use(libs["default"])
pop = S.Population.make(n)
eng = E.Engine(n_max=n, K=K, H=H, L=L)
view = E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)
obs, subj, member = view.tables()
cfg = view.getCurrentConfigurationId()
sc = S.build_scenario(name, subj, cfg)
eng.close()
nbytes = 20 * len(sc.records)
print("workload", name, "records", len(sc.records), "bytes", nbytes, "variants", list(libs))

results = {k: [] for k in libs}
reference = None
for rnd in range(rounds):
    for tag, path in libs.items():
        use(path)
        for k in ("RAPID_TALLY_WAVES", "RAPID_TALLY_BLOCKS_PER_CU"):
            os.environ.pop(k, None)
        os.environ.update(VARIANT_ENV.get(tag.split("-")[0], {}))
        eng = E.Engine(n_max=n, K=K, H=H, L=L)
        E.MembershipView(eng).build(pop.hostnames, pop.ports, pop.id_hi, pop.id_lo)
        sim = E.ClusterSimulation(eng)
        sim.load_streams(sc.records, sc.rec_off)
        sim.set_alert_set(sc.batches.recs, trust_copies=True)
        ms = sim.time_tally(reps)
        sim.set_force_exact(64)
        ms_filter = sim.time_tally(reps)
        ms_ids = ms
        try:  # (round 5: level 2 of the trust -- no late deliveries among the records: their configuration ids are not read)
            sim.set_alert_set(sc.batches.recs, trust_copies=2)
            ms = sim.time_tally(reps)
        except Exception:
            pass
        sim.set_force_exact(0)
        sim.tally()
        emit, nprop, pcount, fp = sim.results()
        if reference is None:
            reference = (emit.copy(), nprop.copy(), pcount.copy(), fp.copy())
        same = all(np.array_equal(a, b) for a, b in zip(reference, (emit, nprop, pcount, fp)))
        st = sim.stats()
        launches = reps + 1
        results[tag].append(ms)
        print("round %d %-8s tally %.4f ms with the ids known current (ids compared per delivery %.4f, filter per delivery %.4f)  %.0f GB/s  results==default: %s  per receiver: rolled back %.2f careful %.2f windows %.1f"
              % (rnd, tag, ms, ms_ids, ms_filter, nbytes / ms / 1e6, same, st["lean_give_ups"] / launches / len(emit),
                 st["careful_subchunks"] / launches / len(emit), st["lean_windows"] / launches / len(emit)), flush=True)
        eng.close()
print("best of %d rounds:" % rounds)
for tag, v in results.items():
    print("  %-8s %.4f ms  %5.1f %% of 8 TB/s" % (tag, min(v), 100 * nbytes / min(v) / 1e6 / 8000))
