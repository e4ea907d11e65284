"""TEST INFRASTRUCTURE ONLY -- the other half of tools/java/: compares what the REAL reference printed (on a machine with a JDK)
with this repository's oracle, closing the "parity unpinned" items of DESIGN.md section 2.
    python oracle/check_java_dump.py java_view_dump.txt                   ring orders, configuration ids, observers (DumpView)
    python oracle/check_java_dump.py --cuts java_cuts.txt records.bin     per-receiver cuts of CutDetectorBench vs oracle
Exit code 0 = everything equal; every difference is printed."""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402
from rapid_amd import scenarios as S  # noqa: E402


def populations():
    out = {}
    a = [(b"127.0.0.1", p, (p - 1234, p - 1234)) for p in range(1234, 1240)]
    out["appendix_c_six_ports"] = a
    out["appendix_c_single"] = a[:1]
    out["kat7_thirty_nodes"] = [(b"127.0.0.2", 2 + i, (i + 1, i + 1)) for i in range(30)]
    for n, k in ((50, 3), (300, 10), (400, 10)):
        pop = S.Population.make(n)
        out["golden_n%d_k%d" % (n, k)] = [(pop.hostnames[i], int(pop.ports[i]), (int(pop.id_hi[i]), int(pop.id_lo[i]))) for i in range(n)]
    return out


def check_views(path):
    pops = populations()
    bad = 0
    cur = None
    for line in open(path):
        t = line.split()
        if not t:
            continue
        if t[0] == "view":
            name, K = t[1], int(t[3])
            nodes = pops[name]
            reg = O.Registry()
            view = O.MembershipView(reg, K)
            handles = [reg.intern(h, p) for h, p, _ in nodes]
            for hd, (_, _, nid) in zip(handles, nodes):
                view.ringAdd(hd, nid)
            label = {hd: "%s:%d" % (h.decode(), p) for hd, (h, p, _) in zip(handles, nodes)}
            cur = (name, view, label, handles)
        elif t[0] == "config_id":
            if int(t[1]) != cur[1].getCurrentConfigurationId():
                bad += 1
                print("DIFF %s: configuration id java %s oracle %d" % (cur[0], t[1], cur[1].getCurrentConfigurationId()))
        elif t[0] == "ring":
            mine = [cur[2][x] for x in cur[1].getRing(int(t[1]))]
            if mine != t[2:]:
                bad += 1
                print("DIFF %s: ring %s" % (cur[0], t[1]))
        elif t[0] == "observers_of_first":
            mine = [cur[2][x] for x in cur[1].getObserversOf(cur[3][0])]
            if mine != t[1:]:
                bad += 1
                print("DIFF %s: observers of the first endpoint" % cur[0])
    print("views:", "all equal" if bad == 0 else "%d differences" % bad)
    return bad


def check_cuts(cuts_path, rec_path):
    f = open(rec_path, "rb").read()
    assert f[:8] == b"RAPIDREC"
    n, K, H, L, cfg = struct.unpack_from("<iiiiq", f, 8)
    p = 8 + 24
    (nm,) = struct.unpack_from("<i", f, p)
    p += 4
    members = np.frombuffer(f, dtype="<i4", count=nm, offset=p).tolist()
    p += 4 * nm
    hosts, ports, hi, lo = [], [], [], []
    for _ in range(n):
        (ln,) = struct.unpack_from("<i", f, p)
        hosts.append(f[p + 4: p + 4 + ln])
        port, a, b = struct.unpack_from("<iqq", f, p + 4 + ln)
        ports.append(port); hi.append(a); lo.append(b)
        p += 4 + ln + 20
    (nr,) = struct.unpack_from("<i", f, p)
    p += 4
    rec_off = np.frombuffer(f, dtype="<i8", count=nr + 1, offset=p).copy()
    p += 8 * (nr + 1)
    records = np.frombuffer(f, dtype=S.ALERT_DTYPE, count=int(rec_off[-1]), offset=p).copy()
    reg = O.Registry()
    for h, q in zip(hosts, ports):
        reg.intern(h, q)
    view = O.MembershipView(reg, K, [(hi[m], lo[m]) for m in members], members)
    oe, on, oo, op = O.sim_run(view, K, H, L, np.array(hi), np.array(lo), records, rec_off, nthreads=8)
    bad = 0
    for line in open(cuts_path):
        t = line.split()
        if t and t[0] == "receiver":
            r = int(t[1])
            got = (int(t[3]), int(t[5]), int(t[7]))
            want = (int(oe[r]), int(on[r]), int(oo[r + 1] - oo[r]))
            if got != want:
                bad += 1
                print("DIFF receiver %d: java (emit_batch, num_proposals, cut_size) %s oracle %s" % (r, got, want))
        elif t and t[0] == "alert_batches":
            print("reference MultiNodeCutDetector:", line.strip())
    print("cuts:", "all equal" if bad == 0 else "%d differences" % bad)
    return bad


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "--cuts":
        sys.exit(1 if check_cuts(sys.argv[2], sys.argv[3]) else 0)
    sys.exit(1 if check_views(sys.argv[1]) else 0)
