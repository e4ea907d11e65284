"""Event counts of the tally kernel on the REAL streams of a BASELINE configuration, without a GPU: a sample of receivers
at full size through the SIMT emulator (tests/emu), checked against the oracle.  What it prints per receiver -- cold / fast
windows applied, windows that went the slow way, sweeps, careful sub-chunks, exact replays, implicit reports -- is what the kernel does on the
device for the same streams (the counters of rapid_sim_stats), so changes to the certificates or to the careful path
can be evaluated before a GPU session.   python scripts/emu_counts.py [config] [receivers]   RAPID_EMU_VARIANT=all ...
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as O  # noqa: E402
from rapid_amd import scenarios as S  # noqa: E402
from tests.emu import pyemu  # noqa: E402
from tests.helpers import oracle_view  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3b"
nrx = int(sys.argv[2]) if len(sys.argv) > 2 else 64
spec = S.CONFIGS[name]
n, K, H, L = spec["n"], spec["K"], spec["H"], spec["L"]
pop = S.Population.make(n)
reg, view = oracle_view(pop, K)
obs, subj, member = view.tables(n)
cfg = view.getCurrentConfigurationId()
sc0 = S.build_scenario(name, subj, cfg, materialise=False)
rx = sc0.receivers[:: max(1, len(sc0.receivers) // nrx)][:nrx]
sc = S.build_scenario(name, subj, cfg, receivers=rx)
t = time.time()
in_lds = 1 if n <= 20000 else 0  # C4: a 200 KB dictionary does not fit the LDS; the kernel then reads its tables from memory
emit, nprop, pcount, fpr, props, stats = pyemu.tally(sc.records, sc.rec_off, n, K, H, L, cfg, obs, subj, member, force_exact=0,
                                                    trusted=True, waves=2, grid=1, tables_in_lds=in_lds)
fe, fn, fo, fp = O.fast_sim_run(n, K, H, L, cfg, obs, subj, member, sc.records, sc.rec_off, nthreads=4)
assert np.array_equal(emit, fe) and np.array_equal(pcount, np.diff(fo)), "emulated kernel and oracle disagree"
keys = ["exact sub-chunks", "cold + fast windows", "sweeps of the hot slots", "restarts", "implicit reports", "records consumed",
        "windows that went slow", "careful sub-chunks"]
print("%s, variant %r: %d receivers, %d records, emulated in %.1f s, results equal to the oracle" %
      (name, os.environ.get("RAPID_EMU_VARIANT", ""), len(rx), len(sc.records), time.time() - t))
for k, v in zip(keys, stats):
    print("  %-26s %10d   %8.2f per receiver" % (k, int(v), int(v) / len(rx)))
