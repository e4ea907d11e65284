"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of the CPU oracle (oracle/_build/liboracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; nothing under
rapid_amd/ does.  Class and method names follow the reference's Java classes so the known-answer tests
read like /root/reference/rapid/src/test/java/com/vrg/rapid/*Test.java.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")

ALERT_DTYPE = np.dtype(
    [("cfg_id", "<i8"), ("src", "<u4"), ("dst", "<u4"), ("ring_mask", "<u2"), ("status", "u1"), ("flags", "u1")]
)
assert ALERT_DTYPE.itemsize == 20

OK, EINVAL, ENODE_EXISTS, ENODE_MISSING, EUUID_SEEN, ECAPACITY = 0, -1, -2, -3, -4, -5
UP, DOWN = 0, 1


class NodeAlreadyInRingException(RuntimeError):
    pass


class NodeNotInRingException(RuntimeError):
    pass


class UUIDAlreadySeenException(RuntimeError):
    pass


def build(force=False):
    """Compile the oracle with g++ (a few seconds)."""
    srcs = [os.path.join(_HERE, f) for f in ("oracle_capi.cpp", "rapid_oracle.hpp", "fast_cut.hpp", "xxh64.hpp")]
    if not force and os.path.exists(_LIB_PATH):
        if os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(s) for s in srcs if os.path.exists(s)):
            return _LIB_PATH
    if not all(os.path.exists(s) for s in srcs):
        if os.path.exists(_LIB_PATH):
            return _LIB_PATH
        raise FileNotFoundError("oracle sources missing")
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    vp, i32, i64, u64 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64
    pi32, pi64 = C.POINTER(C.c_int32), C.POINTER(C.c_int64)

    def sig(name, res, args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args

    sig("orc_xxh64", u64, [C.c_char_p, u64, u64])
    sig("orc_registry_new", vp, [])
    sig("orc_registry_free", None, [vp])
    sig("orc_registry_intern", i32, [vp, C.c_char_p, i32, i32])
    sig("orc_view_new", vp, [vp, i32])
    sig("orc_view_new_from", vp, [vp, i32, pi64, pi64, i32, pi32, i32])
    sig("orc_view_free", None, [vp])
    sig("orc_view_ring_add", i32, [vp, i32, i64, i64])
    sig("orc_view_ring_delete", i32, [vp, i32])
    sig("orc_view_is_safe_to_join", i32, [vp, i32, i64, i64])
    for n in ("observers", "observers_fresh", "subjects", "expected_observers"):
        sig("orc_view_" + n, i32, [vp, i32, pi32, i32])
    sig("orc_view_ring_numbers", i32, [vp, i32, i32, pi32, i32])
    sig("orc_view_ring", i32, [vp, i32, pi32, i32])
    sig("orc_view_ring_key", i64, [vp, i32, i32])
    sig("orc_view_is_host_present", i32, [vp, i32])
    sig("orc_view_is_identifier_present", i32, [vp, i64, i64])
    sig("orc_view_size", i32, [vp])
    sig("orc_view_config_id", i64, [vp])
    sig("orc_view_configuration", i32, [vp, pi64, pi64, i32, pi32, i32, C.POINTER(i32), C.POINTER(i32)])
    sig("orc_cd_new", vp, [i32, i32, i32])
    sig("orc_cd_free", None, [vp])
    sig("orc_cd_aggregate", i32, [vp, i32, i32, i32, pi32, i32, pi32, i32])
    sig("orc_cd_invalidate", i32, [vp, vp, pi32, i32])
    sig("orc_cd_num_proposals", i32, [vp])
    sig("orc_cd_report_count", i32, [vp, i32])
    sig("orc_cd_clear", None, [vp])
    sig("orc_cd_set_snapshot_order", None, [vp, i32])
    sig("orc_svc_new", vp, [vp, i32, i32, i32, pi64, pi64, i32])
    sig("orc_svc_free", None, [vp])
    sig("orc_svc_set_snapshot_order", None, [vp, i32])
    sig("orc_svc_handle_batch", i32, [vp, vp, i32, pi32, i32])
    sig("orc_svc_announced", i32, [vp])
    sig("orc_svc_num_proposals", i32, [vp])
    sig("orc_svc_report_count", i32, [vp, i32])
    sig("orc_svc_decide", i32, [vp, pi32, i32])
    sig("orc_fr_new", vp, [i64, i32])
    sig("orc_fr_free", None, [vp])
    sig("orc_fr_vote", i32, [vp, i32, i64, pi32, i32])
    sig("orc_fr_decided", i32, [vp, pi32, i32])
    sig("orc_sim_set_prewarm", None, [i32])
    sig("orc_sim_set_receiver_nodes", None, [vp, vp])
    sig("orc_view_per_node_caches", None, [vp, i32])
    sig("orc_view_node_has_cached", i32, [vp, i32, i32])
    sig("orc_sim_run", i32, [vp, i32, i32, i32, pi64, pi64, i32, vp, pi64, i32, i32, i32, pi32, pi32, pi64, pi32, i64])
    sig("orc_fast_sim_run", i32,
        [i32, i32, i32, i32, i64, pi32, pi32, C.POINTER(C.c_uint8), vp, pi64, i32, i32, pi32, pi32, pi64, pi32, i64])
    _lib = L
    return L


def _p32(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _p64(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def xxh64(data: bytes, seed: int = 0) -> int:
    return lib().orc_xxh64(data, len(data), seed & 0xFFFFFFFFFFFFFFFF)


class Registry:
    """Interns (hostname bytes, port) endpoints into dense node handles."""

    def __init__(self):
        self._h = lib().orc_registry_new()
        self.endpoints = []

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_registry_free(self._h)
            self._h = None

    def intern(self, hostname, port):
        if isinstance(hostname, str):
            hostname = hostname.encode()
        h = lib().orc_registry_intern(self._h, hostname, len(hostname), port)
        if h == len(self.endpoints):
            self.endpoints.append((hostname, port))
        return h


def _raise(rc):
    if rc == ENODE_EXISTS:
        raise NodeAlreadyInRingException()
    if rc == ENODE_MISSING:
        raise NodeNotInRingException()
    if rc == EUUID_SEEN:
        raise UUIDAlreadySeenException()
    if rc == EINVAL:
        raise ValueError("EINVAL")
    if rc == ECAPACITY:
        raise OverflowError("capacity")
    raise RuntimeError("oracle error %d" % rc)


class MembershipView:
    """MembershipView.java restated (oracle/rapid_oracle.hpp)."""

    CAP = 64

    def nodeHasCached(self, node, subject):
        """Per-node cache mode (sim_run(receiver_nodes=...)): does cluster node `node` hold a cachedObservers entry for `subject`?"""
        return bool(lib().orc_view_node_has_cached(self._h, int(node), int(subject)))

    def __init__(self, registry, K, node_ids=None, endpoints=None):
        self.registry = registry
        self.K = K
        if node_ids is None:
            self._h = lib().orc_view_new(registry._h, K)
        else:
            hi = np.ascontiguousarray([i[0] for i in node_ids], dtype=np.int64)
            lo = np.ascontiguousarray([i[1] for i in node_ids], dtype=np.int64)
            eps = np.ascontiguousarray(endpoints, dtype=np.int32)
            self._h = lib().orc_view_new_from(registry._h, K, _p64(hi), _p64(lo), len(hi), _p32(eps), len(eps))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_view_free(self._h)
            self._h = None

    def ringAdd(self, node, node_id):
        rc = lib().orc_view_ring_add(self._h, node, node_id[0], node_id[1])
        if rc != OK:
            _raise(rc)

    def ringDelete(self, node):
        rc = lib().orc_view_ring_delete(self._h, node)
        if rc != OK:
            _raise(rc)

    def isSafeToJoin(self, node, node_id):
        return lib().orc_view_is_safe_to_join(self._h, node, node_id[0], node_id[1])

    def _list(self, fn, node):
        out = np.empty(self.CAP, dtype=np.int32)
        n = fn(self._h, node, _p32(out), self.CAP)
        if n < 0:
            _raise(n)
        return out[:n].tolist()

    def getObserversOf(self, node):
        return self._list(lib().orc_view_observers, node)

    def computeObserversOf(self, node):
        return self._list(lib().orc_view_observers_fresh, node)

    def getSubjectsOf(self, node):
        return self._list(lib().orc_view_subjects, node)

    def getExpectedObserversOf(self, node):
        return self._list(lib().orc_view_expected_observers, node)

    def getRingNumbers(self, observer, subject):
        out = np.empty(self.CAP, dtype=np.int32)
        n = lib().orc_view_ring_numbers(self._h, observer, subject, _p32(out), self.CAP)
        if n < 0:
            _raise(n)
        return out[:n].tolist()

    def getRing(self, k):
        n = self.getMembershipSize()
        out = np.empty(max(n, 1), dtype=np.int32)
        m = lib().orc_view_ring(self._h, k, _p32(out), len(out))
        if m < 0:
            _raise(m)
        return out[:m].copy()

    def ringKey(self, k, node):
        return lib().orc_view_ring_key(self._h, k, node)

    def isHostPresent(self, node):
        return bool(lib().orc_view_is_host_present(self._h, node))

    def isIdentifierPresent(self, node_id):
        return bool(lib().orc_view_is_identifier_present(self._h, node_id[0], node_id[1]))

    def getMembershipSize(self):
        return lib().orc_view_size(self._h)

    def getCurrentConfigurationId(self):
        return lib().orc_view_config_id(self._h)

    def getConfiguration(self):
        cap_e = max(self.getMembershipSize(), 1)
        cap_i = cap_e
        while True:
            hi = np.empty(cap_i, dtype=np.int64)
            lo = np.empty(cap_i, dtype=np.int64)
            eps = np.empty(cap_e, dtype=np.int32)
            ni, ne = C.c_int(0), C.c_int(0)
            rc = lib().orc_view_configuration(self._h, _p64(hi), _p64(lo), cap_i, _p32(eps), cap_e, C.byref(ni), C.byref(ne))
            if rc == OK:
                return list(zip(hi[: ni.value].tolist(), lo[: ni.value].tolist())), eps[: ne.value].copy()
            cap_i, cap_e = max(ni.value, 1), max(ne.value, 1)

    def tables(self, n_nodes):
        """Dense [n_nodes][K] observer / subject tables + member flags for the index-based engines.
        Members: observers = successors, subjects = predecessors.  Non-members: observers row = expected
        observers (MembershipView.java:292-303), subjects row = -1."""
        K = self.K
        obs = np.full((n_nodes, K), -1, dtype=np.int32)
        subj = np.full((n_nodes, K), -1, dtype=np.int32)
        member = np.zeros(n_nodes, dtype=np.uint8)
        for n in range(n_nodes):
            if self.isHostPresent(n):
                member[n] = 1
                o = self.getObserversOf(n)
                s = self.getSubjectsOf(n)
                if o:
                    obs[n] = o
                    subj[n] = s
            else:
                e = self.getExpectedObserversOf(n)
                if e:
                    obs[n] = e
        return obs, subj, member


class MultiNodeCutDetector:
    """MultiNodeCutDetector.java restated."""

    ASCENDING, DESCENDING, SHUFFLED = 0, 1, 2

    def __init__(self, K, H, L):
        self._h = lib().orc_cd_new(K, H, L)
        if not self._h:
            raise ValueError("Arguments do not satisfy K > H >= L >= 0")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_cd_free(self._h)
            self._h = None

    def aggregateForProposal(self, src, dst, status, ring_numbers):
        if isinstance(ring_numbers, int):
            ring_numbers = [ring_numbers]
        rings = np.ascontiguousarray(ring_numbers, dtype=np.int32)
        out = np.empty(4096, dtype=np.int32)
        n = lib().orc_cd_aggregate(self._h, src, dst, status, _p32(rings), len(rings), _p32(out), len(out))
        if n < 0:
            _raise(n)
        return out[:n].tolist()

    def invalidateFailingEdges(self, view):
        out = np.empty(4096, dtype=np.int32)
        n = lib().orc_cd_invalidate(self._h, view._h, _p32(out), len(out))
        if n < 0:
            _raise(n)
        return out[:n].tolist()

    def getNumProposals(self):
        return lib().orc_cd_num_proposals(self._h)

    def reportCount(self, dst):
        return lib().orc_cd_report_count(self._h, dst)

    def clear(self):
        lib().orc_cd_clear(self._h)

    def setSnapshotOrder(self, order):
        lib().orc_cd_set_snapshot_order(self._h, order)


class AlertBatchService:
    """MembershipService.java:300-354 / 385-430 / 644-685 at one receiver."""

    def __init__(self, view, K, H, L, id_hi, id_lo):
        self.view = view
        self._hi = np.ascontiguousarray(id_hi, dtype=np.int64)
        self._lo = np.ascontiguousarray(id_lo, dtype=np.int64)
        self._h = lib().orc_svc_new(view._h, K, H, L, _p64(self._hi), _p64(self._lo), len(self._hi))
        if not self._h:
            raise ValueError("bad K/H/L")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_svc_free(self._h)
            self._h = None

    def handleBatchedAlertMessage(self, records):
        records = np.ascontiguousarray(records, dtype=ALERT_DTYPE)
        out = np.empty(max(self.view.registry and len(self.view.registry.endpoints), 16), dtype=np.int32)
        n = lib().orc_svc_handle_batch(self._h, records.ctypes.data, len(records), _p32(out), len(out))
        if n < 0:
            _raise(n)
        return out[:n].tolist()

    def announcedProposal(self):
        return bool(lib().orc_svc_announced(self._h))

    def getNumProposals(self):
        return lib().orc_svc_num_proposals(self._h)

    def reportCount(self, dst):
        return lib().orc_svc_report_count(self._h, dst)

    def setSnapshotOrder(self, order):
        lib().orc_svc_set_snapshot_order(self._h, order)

    def decideViewChange(self, proposal):
        p = np.ascontiguousarray(proposal, dtype=np.int32)
        rc = lib().orc_svc_decide(self._h, _p32(p), len(p))
        if rc != OK:
            _raise(rc)


class FastRound:
    """FastPaxos.java:125-156 (fast round vote counting)."""

    def __init__(self, configuration_id, membership_size):
        self._h = lib().orc_fr_new(configuration_id, membership_size)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_fr_free(self._h)
            self._h = None

    def handleFastRoundProposal(self, sender, configuration_id, endpoints):
        e = np.ascontiguousarray(endpoints, dtype=np.int32)
        return bool(lib().orc_fr_vote(self._h, sender, configuration_id, _p32(e), len(e)))

    def decided(self):
        out = np.empty(1 << 16, dtype=np.int32)
        n = lib().orc_fr_decided(self._h, _p32(out), len(out))
        return None if n < 0 else out[:n].tolist()


def _csr(off, vals, R):
    return [vals[off[r]: off[r + 1]].copy() for r in range(R)]


def sim_run(view, K, H, L, id_hi, id_lo, records, rec_off, snapshot_order=0, nthreads=1, prewarm_observers=True, receiver_nodes=None):
    """Faithful whole-population run: one AlertBatchService per receiver over a shared view.  prewarm_observers=False:
    the view's observer cache (quirk Q4) is only filled by what these receivers query, and survives into the next call
    with whatever the view changes in between left of it (single-threaded).  receiver_nodes (one cluster node per receiver):
    every receiver reads and fills ITS OWN cachedObservers, as the nodes of a deployment do -- each holds its own MembershipView
    object (R/MembershipView.java:49, 210-224); the caches persist in the view, per node, across calls and view changes."""
    lib().orc_sim_set_prewarm(1 if prewarm_observers else 0)
    rx_nodes = None if receiver_nodes is None else np.ascontiguousarray(receiver_nodes, dtype=np.int32)
    assert rx_nodes is None or len(rx_nodes) == len(rec_off) - 1
    lib().orc_sim_set_receiver_nodes(view._h, None if rx_nodes is None else rx_nodes.ctypes.data)
    records = np.ascontiguousarray(records, dtype=ALERT_DTYPE)
    rec_off = np.ascontiguousarray(rec_off, dtype=np.int64)
    hi = np.ascontiguousarray(id_hi, dtype=np.int64)
    lo = np.ascontiguousarray(id_lo, dtype=np.int64)
    R = len(rec_off) - 1
    emit = np.empty(R, dtype=np.int32)
    nprop = np.empty(R, dtype=np.int32)
    poff = np.empty(R + 1, dtype=np.int64)
    cap = max(int(R) * 64, 1 << 16)
    while True:
        props = np.empty(cap, dtype=np.int32)
        rc = lib().orc_sim_run(view._h, K, H, L, _p64(hi), _p64(lo), len(hi), records.ctypes.data, _p64(rec_off), R,
                               snapshot_order, nthreads, _p32(emit), _p32(nprop), _p64(poff), _p32(props), cap)
        if rc == OK:
            break
        if rc == ECAPACITY:
            cap = int(poff[R]) + 1
            continue
        lib().orc_sim_set_receiver_nodes(view._h, None)
        _raise(rc)
    lib().orc_sim_set_receiver_nodes(view._h, None)
    return emit, nprop, poff, props[: poff[R]].copy()


def fast_sim_run(n_nodes, K, H, L, cfg_id, obs, subj, member, records, rec_off, nthreads=1):
    """Optimised CPU formulation (oracle/fast_cut.hpp) over dense tables."""
    records = np.ascontiguousarray(records, dtype=ALERT_DTYPE)
    rec_off = np.ascontiguousarray(rec_off, dtype=np.int64)
    obs = np.ascontiguousarray(obs, dtype=np.int32)
    subj = np.ascontiguousarray(subj, dtype=np.int32)
    member = np.ascontiguousarray(member, dtype=np.uint8)
    R = len(rec_off) - 1
    emit = np.empty(R, dtype=np.int32)
    nprop = np.empty(R, dtype=np.int32)
    poff = np.empty(R + 1, dtype=np.int64)
    cap = max(int(R) * 64, 1 << 16)
    while True:
        props = np.empty(cap, dtype=np.int32)
        rc = lib().orc_fast_sim_run(n_nodes, K, H, L, cfg_id, _p32(obs), _p32(subj),
                                    member.ctypes.data_as(C.POINTER(C.c_uint8)), records.ctypes.data, _p64(rec_off), R,
                                    nthreads, _p32(emit), _p32(nprop), _p64(poff), _p32(props), cap)
        if rc == OK:
            break
        if rc == ECAPACITY:
            cap = int(poff[R]) + 1
            continue
        _raise(rc)
    return emit, nprop, poff, props[: poff[R]].copy()
