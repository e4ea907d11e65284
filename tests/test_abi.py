"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol that
include/rapid_mi355x.h declares, the record layout is the documented 20 bytes, and the host-only control
object (FastPaxos fast round, R/FastPaxos.java:125-156) passes the reference's quorum tables.  No device
compute is attempted here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from rapid_amd import _native as N
from rapid_amd import scenarios as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    """-> (the entry points of the product library, the ones only the test build exports: the header's RAPID_TEST_BUILD section)"""
    text = open(os.path.join(ROOT, "include", "rapid_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    a, b = text.index("#ifdef RAPID_TEST_BUILD"), text.index("#endif", text.index("#ifdef RAPID_TEST_BUILD"))
    find = lambda t: sorted(set(re.findall(r"\b(rapid_[a-z0-9_]+)\s*\(", t)))
    return find(text[:a] + text[b:]), find(text[a:b])


def exported(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return {ln.split()[-1] for ln in out.splitlines() if " T " in ln and ln.split()[-1].startswith("rapid_")}


def test_library_exports_every_declared_symbol():
    """The product library exports exactly what the header declares outside its RAPID_TEST_BUILD section -- no testing aid, no
    probe -- and holds no probe kernel and no getenv; the test build exports both sections."""
    N.build()
    product, test_only = declared_symbols()
    assert len(product) >= 40 and len(test_only) >= 5
    assert set(product) == set(N.SIGNATURES), set(product) ^ set(N.SIGNATURES)
    assert set(test_only) == set(N.TEST_SIGNATURES), set(test_only) ^ set(N.TEST_SIGNATURES)
    assert exported(N.LIB_PATH) == set(product), exported(N.LIB_PATH) ^ set(product)
    assert exported(N.TEST_LIB_PATH) == set(product) | set(test_only)
    blob = open(N.LIB_PATH, "rb").read()
    for needle in (b"stream_probe_kernel", b"dma_probe_kernel", b"rapid_debug_", b"RAPID_TALLY_WAVES", b"RAPID_POOL_EIGHTHS", b"RAPID_TIME_VIEW"):
        assert needle not in blob, needle
    assert b"stream_probe_kernel" in open(N.TEST_LIB_PATH, "rb").read()


def test_record_layout():
    assert S.ALERT_DTYPE.itemsize == 20
    assert [S.ALERT_DTYPE.fields[f][1] for f in ("cfg_id", "src", "dst", "ring_mask", "status", "flags")] == [0, 8, 12, 16, 18, 19]
    assert C.sizeof(N.RoundResult) == 48 and C.sizeof(N.EngineConfig) == 24


def test_engine_create_validates_khl_before_touching_the_device():
    """R/MultiNodeCutDetector.java:52-55: H > K || L > H || K < 3 || L <= 0 || H <= 0 -> IllegalArgumentException."""
    L = N.lib()
    for K, H, Lw in [(10, 11, 2), (10, 8, 9), (2, 2, 1), (10, 8, 0), (10, 0, 0), (15, 9, 4)]:
        h = C.c_void_p()
        cfg = N.EngineConfig(100, K, H, Lw, 0, 0)
        assert L.rapid_engine_create(C.byref(cfg), C.byref(h)) == N.EINVAL


def test_no_device_is_an_error_not_a_fallback():
    L = N.lib()
    if L.rapid_device_count() > 0:
        pytest.skip("a gfx950 device is present")
    h = C.c_void_p()
    cfg = N.EngineConfig(100, 10, 9, 4, 0, 0)
    assert L.rapid_engine_create(C.byref(cfg), C.byref(h)) == N.EDEVICE
    from rapid_amd import engine as E
    with pytest.raises(N.RapidError):
        E.Engine(100)


# ---- FastPaxosWithoutFallbackTests.java:61-148 through rapid_fast_round_* (host control object) ----
NO_CONFLICT = [(6, 5), (48, 37), (50, 38), (100, 76), (102, 77), (5, 4), (51, 39), (49, 37), (99, 75), (101, 76)]
CONFLICTS = ([(6, 5, 1, True), (48, 37, 1, True), (50, 38, 1, True), (100, 76, 1, True), (102, 77, 1, True)]
             + [(48, 37, 11, True), (50, 38, 12, True), (100, 76, 24, True), (102, 77, 25, True)]
             + [(6, 5, 2, False), (48, 37, 14, False), (50, 38, 13, False), (100, 76, 25, False), (102, 77, 26, False)])


@pytest.mark.parametrize("n,quorum", NO_CONFLICT)
def test_fastQuorumTestNoConflicts(n, quorum):
    from rapid_amd.engine import FastPaxos
    fp = FastPaxos(configuration_id=-7, membership_size=n)
    for i in range(quorum - 1):
        assert not fp.handleFastRoundProposal(1000 + i, -7, [1])
        assert fp.decision() is None
    assert fp.handleFastRoundProposal(1000 + quorum - 1, -7, [1])
    assert fp.decision() == [1]


@pytest.mark.parametrize("n,quorum,num_conflicts,change", CONFLICTS)
def test_fastQuorumTestWithConflicts(n, quorum, num_conflicts, change):
    from rapid_amd.engine import FastPaxos
    fp = FastPaxos(configuration_id=5, membership_size=n)
    for i in range(num_conflicts):
        assert not fp.handleFastRoundProposal(i, 5, [2])
    non_conflict = min(num_conflicts + quorum - 1, n - 1)
    for i in range(num_conflicts, non_conflict):
        assert not fp.handleFastRoundProposal(i, 5, [1])
    assert fp.handleFastRoundProposal(non_conflict, 5, [1]) == change
    assert fp.decision() == ([1] if change else None)


def test_fast_round_filters():
    from rapid_amd.engine import FastPaxos
    fp = FastPaxos(7, 5)
    assert not fp.handleFastRoundProposal(0, 8, [1])  # other configuration: dropped (R/FastPaxos.java:126-132)
    for s in range(3):
        assert not fp.handleFastRoundProposal(s, 7, [1])
        assert not fp.handleFastRoundProposal(s, 7, [1])  # same sender twice: second ignored (:134-136)
    assert fp.handleFastRoundProposal(3, 7, [1])
    assert fp.handleFastRoundProposal(4, 7, [2])  # decided: ignored, still decided on [1] (:138-140)
    assert fp.decision() == [1]
    # ordered-list identity: [1, 2] and [2, 1] are different proposals (List<Endpoint> equality)
    fp = FastPaxos(1, 4)
    for s, p in enumerate(([1, 2], [2, 1], [1, 2])):
        assert not fp.handleFastRoundProposal(s, 1, p)
    assert fp.handleFastRoundProposal(3, 1, [1, 2]) is False  # 3 votes for [1,2] < N - F = 4


def test_self_test_constants_are_the_oracles():
    """rapid_engine_self_test (csrc/engine.hip) compares the device's answers for a fixed 5-node cluster with constants compiled into
    the library; they are the CPU oracle's values for the same endpoints -- recomputed here, so they cannot drift unnoticed.  (The
    product never calls the oracle: the constants are how it knows the answers.)"""
    from oracle import pyoracle as O
    src = open(os.path.join(ROOT, "rapid_amd", "csrc", "engine.hip")).read()
    body = src[src.index("int rapid_engine_self_test(rapid_engine* h) {"):]
    body = body[:body.index("\n}\n")]
    cfg, after = [int(x) for x in re.search(r"kCfg = (-?\d+)ll, kCfgAfter = (-?\d+)ll", body).groups()]
    ring0 = [int(x) for x in re.search(r"kRing0\[5\] = \{([^}]*)\}", body).group(1).split(",")]
    ring0_after = [int(x) for x in re.search(r"kRing0After\[4\] = \{([^}]*)\}", body).group(1).split(",")]
    reg = O.Registry()
    for i in range(5):
        assert reg.intern(b"10.0.0.%d" % i, 5000) == i
    view = O.MembershipView(reg, 3, [(i + 1, i + 101) for i in range(5)], list(range(5)))
    assert view.getCurrentConfigurationId() == cfg and view.getRing(0).tolist() == ring0
    assert len(view.getObserversOf(4)) == 3
    view.ringDelete(4)
    assert view.getCurrentConfigurationId() == after and view.getRing(0).tolist() == ring0_after
