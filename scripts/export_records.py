"""Writes a BASELINE workload as the record file tools/java/CutDetectorBench.java reads (format: its header comment), so that
whoever has a JDK can time the REAL MultiNodeCutDetector on the streams bench.py replays and compare its cuts with the
engine's.  No GPU needed (the topology comes from the oracle).   python scripts/export_records.py [config] [receivers] [out.bin]"""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rapid_amd import scenarios as S  # noqa: E402
from tests.helpers import oracle_view  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3b"
n_rx = int(sys.argv[2]) if len(sys.argv) > 2 else 64
out = sys.argv[3] if len(sys.argv) > 3 else "records_%s_%drx.bin" % (name, n_rx)
spec = S.CONFIGS[name]
n, K, H, L = spec["n"], spec["K"], spec["H"], spec["L"]
pop = S.Population.make(n)
reg, view = oracle_view(pop, K)
obs, subj, member = view.tables(n)
cfg = view.getCurrentConfigurationId()
sc0 = S.build_scenario(name, subj, cfg, materialise=False)
rx = sc0.receivers[:: max(1, len(sc0.receivers) // n_rx)][:n_rx]
sc = S.build_scenario(name, subj, cfg, receivers=rx)
with open(out, "wb") as f:
    f.write(b"RAPIDREC")
    f.write(struct.pack("<iiiiq", n, K, H, L, cfg))
    members = np.flatnonzero(member).astype("<i4")
    f.write(struct.pack("<i", len(members)))
    f.write(members.tobytes())
    for i in range(n):
        h = pop.hostnames[i]
        f.write(struct.pack("<i", len(h)) + h + struct.pack("<iqq", int(pop.ports[i]), int(pop.id_hi[i]), int(pop.id_lo[i])))
    f.write(struct.pack("<i", len(rx)))
    f.write(np.asarray(sc.rec_off, dtype="<i8").tobytes())
    f.write(np.ascontiguousarray(sc.records).tobytes())
print("wrote", out, os.path.getsize(out), "bytes:", len(rx), "receivers,", len(sc.records), "records, configuration id", cfg)
