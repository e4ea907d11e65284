"""Host-side mirror of the reference's interface for the hot path, over the C ABI (include/rapid_mi355x.h).

Class and method names follow the Java classes they stand in for, so the parity tests read like the reference's
own tests (R/ = /root/reference/rapid/src/main/java/com/vrg/rapid/):

    Engine                  one simulated cluster on one MI355X (lifecycle + registry of endpoints)
    MembershipView          R/MembershipView.java          (ringAdd / ringDelete / getObserversOf / ...)
    MultiNodeCutDetector    R/MultiNodeCutDetector.java    (aggregateForProposal / invalidateFailingEdges / ...)
    FastPaxos               R/FastPaxos.java:125-156       (handleFastRoundProposal; fast round only)
    ClusterSimulation       R/MembershipService.java:300-354, 385-430 replayed at every receiver at once

Everything executes in librapid_mi355x.so on the GPU.  There is no CPU path: constructing an Engine without
the built library or without a gfx950 device raises.
"""
import ctypes as C

import numpy as np

from . import _native as N
from ._native import (IllegalArgumentException, NodeAlreadyInRingException, NodeNotInRingException,  # noqa: F401
                      RapidError, RoundResult, UUIDAlreadySeenException)
from .scenarios import ALERT_DTYPE

UP, DOWN = 0, 1
HOSTNAME_ALREADY_IN_RING, UUID_ALREADY_IN_RING, SAFE_TO_JOIN = 0, 1, 2


def _addr(a):
    return None if a is None else a.ctypes.data


def device_count():
    return N.lib().rapid_device_count()


class Engine:
    def __init__(self, n_max, K=10, H=9, L=4, device_id=0, max_cut=0):
        self._lib = N.lib()
        self._h = C.c_void_p()
        cfg = N.EngineConfig(n_max, K, H, L, device_id, max_cut)
        rc = self._lib.rapid_engine_create(C.byref(cfg), C.byref(self._h))
        if rc != N.OK:
            self._h = C.c_void_p()
            msg = {N.EINVAL: "Arguments do not satisfy K > H >= L >= 0 (or K > %d)" % 14,
                   N.EDEVICE: "no usable gfx950 device (there is no CPU fallback)"}.get(rc, "")
            N.raise_for(rc, msg)
        self.n_max, self.K, self.H, self.L = n_max, K, H, L
        self.max_cut = max_cut if max_cut > 0 else min(n_max, 4096)
        self.n_nodes = 0

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.rapid_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != N.OK:
            msg = self._lib.rapid_last_error(self._h)
            N.raise_for(rc, msg.decode() if msg else "")

    def sync(self):
        self._check(self._lib.rapid_engine_sync(self._h))

    def self_test(self):
        """rapid_engine_self_test: a known-answer view + round on a private engine on this device; raises RapidError(EDEVICE)."""
        self._check(self._lib.rapid_engine_self_test(self._h))

    @property
    def stream(self):
        return self._lib.rapid_engine_stream(self._h)

    def comm_init(self, unique_id: bytes, rank: int, n_ranks: int):
        buf = np.frombuffer(unique_id, dtype=np.uint8).copy()
        self._check(self._lib.rapid_engine_comm_init(self._h, _addr(buf), rank, n_ranks))

    def comm_info(self):
        """(rank, n_ranks) as the RCCL communicator reports them; (0, 1) without one."""
        r, n = C.c_int32(0), C.c_int32(0)
        self._check(self._lib.rapid_engine_comm_info(self._h, C.byref(r), C.byref(n)))
        return int(r.value), int(n.value)


def comm_unique_id() -> bytes:
    buf = np.zeros(128, dtype=np.uint8)
    rc = N.lib().rapid_comm_unique_id(_addr(buf))
    if rc != N.OK:
        N.raise_for(rc, "ncclGetUniqueId failed")
    return buf.tobytes()


class MembershipView:
    """R/MembershipView.java on the device.  Endpoints are dense node indices into the registry given to build()."""

    def __init__(self, engine):
        self.e = engine
        self.K = engine.K

    def build(self, hostnames, ports, id_hi, id_lo, members=None, extra_ids=None):
        """MembershipView(K, nodeIds, endpoints) (:74-89).  hostnames: list of bytes."""
        n = len(hostnames)
        off = np.zeros(n + 1, dtype=np.int32)
        off[1:] = np.cumsum([len(x) for x in hostnames])
        blob = np.frombuffer(b"".join(hostnames) or b"\0", dtype=np.uint8).copy()
        ports = np.ascontiguousarray(ports, dtype=np.int32)
        hi = np.ascontiguousarray(id_hi, dtype=np.int64)
        lo = np.ascontiguousarray(id_lo, dtype=np.int64)
        mem = np.arange(n, dtype=np.int32) if members is None else np.ascontiguousarray(members, dtype=np.int32)
        if extra_ids:
            xh = np.ascontiguousarray([i[0] for i in extra_ids], dtype=np.int64)
            xl = np.ascontiguousarray([i[1] for i in extra_ids], dtype=np.int64)
        else:
            xh = xl = None
        self.e._check(self.e._lib.rapid_view_build(self.e._h, _addr(blob), _addr(off), _addr(ports), _addr(hi), _addr(lo),
                                                   n, _addr(mem) if len(mem) else None, len(mem), _addr(xh), _addr(xl),
                                                   0 if xh is None else len(xh)))
        self.e.n_nodes = n
        return self

    def _list(self, fn, *args, cap=None):
        cap = cap or max(self.K, 1)
        out = np.empty(cap, dtype=np.int32)
        n = C.c_int32(0)
        self.e._check(fn(self.e._h, *args, _addr(out), cap, C.byref(n)))
        return out[: n.value].tolist()

    def registerEndpoints(self, hostnames, ports, id_hi, id_lo):
        """Appends endpoints to the registry as non-members (rapid_view_register_endpoints); -> index of the first one."""
        blob = np.frombuffer(b"".join(hostnames), dtype=np.uint8).copy() if len(hostnames) else np.zeros(1, dtype=np.uint8)
        off = np.zeros(len(hostnames) + 1, dtype=np.int32)
        off[1:] = np.cumsum([len(h) for h in hostnames])
        ports = np.ascontiguousarray(ports, dtype=np.int32)
        hi = np.ascontiguousarray(id_hi, dtype=np.int64)
        lo = np.ascontiguousarray(id_lo, dtype=np.int64)
        first = C.c_int32(-1)
        self.e._check(self.e._lib.rapid_view_register_endpoints(self.e._h, _addr(blob), _addr(off), _addr(ports), _addr(hi), _addr(lo),
                                                                len(hostnames), C.byref(first)))
        self.e.n_nodes += len(hostnames)
        return first.value

    def setObserverCacheEmulation(self, on):
        """Quirk Q4 (stale cachedObservers, R/MembershipView.java:143-152, 181-195): reproduced by default; False = the round
        index always reads today's observers."""
        self.e._check(self.e._lib.rapid_view_q4_emulation(self.e._h, 1 if on else 0))

    def isSafeToJoin(self, node, node_id):
        s = C.c_int32(0)
        self.e._check(self.e._lib.rapid_view_is_safe_to_join(self.e._h, node, int(node_id[0]), int(node_id[1]), C.byref(s)))
        return s.value

    def ringAdd(self, node, node_id):
        """ringAdd(Endpoint, NodeId) (:123-160); node_id = (high, low)."""
        self.e._check(self.e._lib.rapid_view_ring_add(self.e._h, node, int(node_id[0]), int(node_id[1])))

    def ringDelete(self, node):
        self.e._check(self.e._lib.rapid_view_ring_delete(self.e._h, node))

    def getObserversOf(self, node):
        return self._list(self.e._lib.rapid_view_observers, node)

    def getSubjectsOf(self, node):
        return self._list(self.e._lib.rapid_view_subjects, node)

    def getExpectedObserversOf(self, node):
        return self._list(self.e._lib.rapid_view_expected_observers, node)

    def getRingNumbers(self, observer, subject):
        return self._list(self.e._lib.rapid_view_ring_numbers, observer, subject)

    def getRing(self, k):
        return np.asarray(self._list(self.e._lib.rapid_view_ring, k, cap=max(self.getMembershipSize(), 1)), dtype=np.int32)

    def ringKey(self, k, node):
        v = C.c_int64(0)
        self.e._check(self.e._lib.rapid_view_ring_key(self.e._h, k, node, C.byref(v)))
        return v.value

    def isHostPresent(self, node):
        v = C.c_int32(0)
        self.e._check(self.e._lib.rapid_view_is_host_present(self.e._h, node, C.byref(v)))
        return bool(v.value)

    def getMembershipSize(self):
        v = C.c_int32(0)
        self.e._check(self.e._lib.rapid_view_size(self.e._h, C.byref(v)))
        return v.value

    def getCurrentConfigurationId(self):
        v = C.c_int64(0)
        self.e._check(self.e._lib.rapid_view_config_id(self.e._h, C.byref(v)))
        return v.value

    def tables(self):
        n, K = self.e.n_nodes, self.K
        obs = np.empty((n, K), dtype=np.int32)
        subj = np.empty((n, K), dtype=np.int32)
        member = np.empty(n, dtype=np.uint8)
        self.e._check(self.e._lib.rapid_view_tables(self.e._h, _addr(obs), _addr(subj), _addr(member), n))
        return obs, subj, member


def alert(src, dst, status, configuration_id, ring_numbers):
    """AlertMessage (rapid.proto:102-111) as one packed record."""
    if isinstance(ring_numbers, int):
        ring_numbers = [ring_numbers]
    a = np.zeros(1, dtype=ALERT_DTYPE)
    a["src"], a["dst"], a["status"], a["cfg_id"] = src, dst, status, configuration_id
    m = 0
    for r in ring_numbers:
        m |= 1 << r
    a["ring_mask"] = m
    return a


class MultiNodeCutDetector:
    """R/MultiNodeCutDetector.java, state on the device (one wavefront runs the exact sequential kernel)."""

    def __init__(self, engine, K, H, L):
        self.e = engine
        self._h = C.c_void_p()
        self.e._check(self.e._lib.rapid_cd_create(self.e._h, K, H, L, C.byref(self._h)))
        self._cap = engine.n_max

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value and self.e._h.value:
                self.e._lib.rapid_cd_destroy(self._h)
        except Exception:
            pass

    def aggregateForProposal(self, alerts):
        """One AlertMessage (or an array of them, applied in order).  Returns the list of proposed endpoints."""
        alerts = np.ascontiguousarray(alerts, dtype=ALERT_DTYPE)
        out = np.empty(self._cap, dtype=np.int32)
        counts = np.empty(max(len(alerts), 1), dtype=np.int32)
        n = C.c_int32(0)
        self.e._check(self.e._lib.rapid_cd_aggregate(self._h, _addr(alerts), len(alerts), _addr(out), self._cap,
                                                     _addr(counts), C.byref(n)))
        return out[: n.value].tolist()

    def invalidateFailingEdges(self):
        out = np.empty(self._cap, dtype=np.int32)
        n = C.c_int32(0)
        self.e._check(self.e._lib.rapid_cd_invalidate(self._h, _addr(out), self._cap, C.byref(n)))
        return out[: n.value].tolist()

    def getNumProposals(self):
        n = C.c_int32(0)
        self.e._check(self.e._lib.rapid_cd_num_proposals(self._h, C.byref(n)))
        return n.value

    def clear(self):
        self.e._check(self.e._lib.rapid_cd_clear(self._h))


class FastPaxos:
    """Fast round of R/FastPaxos.java (:125-156): count identical proposals, decide at N - floor((N-1)/4)."""

    def __init__(self, configuration_id, membership_size):
        self._lib = N.lib()
        self._h = C.c_void_p()
        rc = self._lib.rapid_fast_round_create(configuration_id, membership_size, C.byref(self._h))
        if rc != N.OK:
            N.raise_for(rc)

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                self._lib.rapid_fast_round_destroy(self._h)
        except Exception:
            pass

    def handleFastRoundProposal(self, sender, configuration_id, endpoints):
        e = np.ascontiguousarray(endpoints, dtype=np.int32)
        d = C.c_int32(0)
        rc = self._lib.rapid_fast_round_vote(self._h, sender, configuration_id, _addr(e) if len(e) else None, len(e),
                                             C.byref(d))
        if rc != N.OK:
            N.raise_for(rc)
        return bool(d.value)

    def decision(self):
        out = np.empty(1 << 16, dtype=np.int32)
        n = C.c_int32(0)
        rc = self._lib.rapid_fast_round_decision(self._h, _addr(out), len(out), C.byref(n))
        if rc == N.ESTATE:
            return None
        if rc != N.OK:
            N.raise_for(rc)
        return out[: n.value].tolist()


class ClusterSimulation:
    """Every receiver's MembershipService alert path at once (R/MembershipService.java:300-354), the fast-round
    vote count over the population (R/FastPaxos.java:125-156) and decideViewChange (:385-430)."""

    def __init__(self, engine):
        self.e = engine
        self.n_receivers = 0
        self._keep = None

    def load_streams(self, records, rec_off):
        records = np.ascontiguousarray(records, dtype=ALERT_DTYPE)
        rec_off = np.ascontiguousarray(rec_off, dtype=np.int64)
        self.n_receivers = len(rec_off) - 1
        self.e._check(self.e._lib.rapid_sim_load_streams(self.e._h, _addr(records) if len(records) else None, _addr(rec_off),
                                                         self.n_receivers))

    def load_streams_device(self, d_records_ptr, records_bytes, d_rec_off_ptr, n_receivers, keepalive=None):
        """20-byte records already in device memory, copied into the engine's buffer (the offsets stay borrowed)."""
        self.e._check(self.e._lib.rapid_sim_load_streams_device(self.e._h, d_records_ptr, records_bytes, d_rec_off_ptr,
                                                                n_receivers))
        self._keep = keepalive  # (a refused call changes nothing, here either)
        self.n_receivers = n_receivers

    def attach_streams_device(self, d_records_ptr, records_bytes, d_rec_off_ptr, n_receivers, keepalive=None):
        """20-byte records in device memory, tallied IN PLACE (rapid_sim_attach_streams_device): nothing is copied; both
        buffers stay borrowed until the next load / attach / generate -- `keepalive` holds them."""
        self.e._check(self.e._lib.rapid_sim_attach_streams_device(self.e._h, d_records_ptr, records_bytes, d_rec_off_ptr,
                                                                  n_receivers))
        self._keep = keepalive
        self.n_receivers = n_receivers

    def generate(self, batches, receivers, seed, trust_copies=False, keep=None, boundary=False):
        """The round's deliveries made on the device (rapid_sim_generate): every receiver gets every batch of `batches`
        (scenarios.BatchSet) once, in the order scenarios.deliver_hashed states on the host; `batches.recs` is the declared
        alert set of the round.  keep: per-batch 32-bit delivery thresholds (None: every batch reaches every receiver).
        boundary: write the 20-byte boundary records instead of resolved 8-byte ones."""
        recs = np.ascontiguousarray(batches.recs, dtype=ALERT_DTYPE)
        off = np.ascontiguousarray(batches.off, dtype=np.int64)
        rx = np.ascontiguousarray(receivers, dtype=np.int32)
        kp = None if keep is None else np.ascontiguousarray(keep, dtype=np.uint32)
        assert kp is None or len(kp) == len(off) - 1
        self.n_receivers = len(rx)
        self.e._check(self.e._lib.rapid_sim_generate(self.e._h, _addr(recs) if len(recs) else None, _addr(off), len(off) - 1, _addr(kp),
                                                     _addr(rx) if len(rx) else None, len(rx), C.c_uint64(int(seed) & ((1 << 64) - 1)),
                                                     1 if boundary else 0))
        if trust_copies:
            self.e._check(self.e._lib.rapid_sim_trust_alert_copies(self.e._h, 1))

    def round_tiled(self, batches, receivers, seed, tile_receivers=0, keep=None, boundary=False):
        """One round over a population that does not fit one launch (rapid_sim_round_tiled): the deliveries made tile by tile on
        the device, every tile tallied, the fast-round votes accumulated across the tiles' launches.  -> RoundResult; afterwards
        results() covers every receiver, decided_cut() the decision, proposal(r) the receivers of the last tile."""
        recs = np.ascontiguousarray(batches.recs, dtype=ALERT_DTYPE)
        off = np.ascontiguousarray(batches.off, dtype=np.int64)
        rx = np.ascontiguousarray(receivers, dtype=np.int32)
        kp = None if keep is None else np.ascontiguousarray(keep, dtype=np.uint32)
        assert kp is None or len(kp) == len(off) - 1
        rr = RoundResult()
        self.e._check(self.e._lib.rapid_sim_round_tiled(self.e._h, _addr(recs) if len(recs) else None, _addr(off), len(off) - 1, _addr(kp),
                                                        _addr(rx) if len(rx) else None, len(rx), int(tile_receivers),
                                                        C.c_uint64(int(seed) & ((1 << 64) - 1)), 1 if boundary else 0, C.byref(rr)))
        self.n_receivers = len(rx)
        return rr

    def round_tiled_info(self):
        t = np.zeros(4, dtype=np.float64)
        self.e._check(self.e._lib.rapid_sim_round_tiled_info(self.e._h, _addr(t)))
        return dict(wall_ms=float(t[0]), tiles=int(t[1]), passes=int(t[2]), records_delivered=int(round(t[3] * 1e6)))

    def read_records(self, first, n):
        """Testing aid: (subjects -- or, of generated resolved records, their dictionary entries --, core words) of n records."""
        subj = np.zeros(max(n, 1), dtype=np.uint32)
        words = np.zeros(max(n, 1), dtype=np.uint32)
        self.e._check(self.e._lib.rapid_debug_read_records(self.e._h, C.c_int64(int(first)), int(n), _addr(subj), _addr(words)))
        return subj[:n], words[:n]

    def set_alert_set(self, alerts, trust_copies=False):
        """Declares the round's distinct alerts (see rapid_sim_set_alert_set).  trust_copies (rapid_sim_trust_alert_copies): True / 1 =
        asks for the pre-validated instantiation (granted on verified facts; late deliveries of another configuration are still
        dropped per delivery); 2 = the caller also vouches that NO delivered record carries another configuration id -- the records'
        configuration ids are then not read at all (this one rests on the caller's word)."""
        alerts = np.ascontiguousarray(alerts, dtype=ALERT_DTYPE)
        self.e._check(self.e._lib.rapid_sim_set_alert_set(self.e._h, _addr(alerts) if len(alerts) else None, len(alerts)))
        self.e._check(self.e._lib.rapid_sim_trust_alert_copies(self.e._h, int(trust_copies)))

    def set_alert_set_device(self, d_alerts_ptr, n_alerts, trust_copies=False, keepalive=None, alerts_bytes=None):
        """The round's distinct alerts, already in device memory (rapid_sim_set_alert_set_device): read in place.
        alerts_bytes: readable bytes at d_alerts_ptr (default: exactly the n_alerts records)."""
        self._keep_alerts = keepalive
        nbytes = 20 * int(n_alerts) if alerts_bytes is None else int(alerts_bytes)
        self.e._check(self.e._lib.rapid_sim_set_alert_set_device(self.e._h, d_alerts_ptr, nbytes, n_alerts))
        self.e._check(self.e._lib.rapid_sim_trust_alert_copies(self.e._h, int(trust_copies)))

    def tally(self):
        self.e._check(self.e._lib.rapid_sim_tally(self.e._h))

    def results(self):
        R = self.n_receivers
        emit = np.empty(R, dtype=np.int32)
        nprop = np.empty(R, dtype=np.int32)
        pcount = np.empty(R, dtype=np.int32)
        fp = np.empty(R, dtype=np.uint64)
        self.e._check(self.e._lib.rapid_sim_results(self.e._h, _addr(emit), _addr(nprop), _addr(pcount), _addr(fp), R))
        return emit, nprop, pcount, fp

    def proposal(self, receiver):
        out = np.empty(self.e.max_cut, dtype=np.int32)
        n = C.c_int32(0)
        self.e._check(self.e._lib.rapid_sim_proposal(self.e._h, receiver, _addr(out), len(out), C.byref(n)))
        return out[: n.value].tolist()

    def count_votes(self):
        rr = RoundResult()
        self.e._check(self.e._lib.rapid_sim_count_votes(self.e._h, C.byref(rr)))
        return rr

    def vote_segment(self):
        """Testing aid: the answer block this engine's voters contribute to the sharded count's all-gather (bytes)."""
        n = C.c_int64(0)
        self.e._check(self.e._lib.rapid_debug_vote_segment(self.e._h, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=np.uint8)
        self.e._check(self.e._lib.rapid_debug_vote_segment(self.e._h, _addr(out), n.value, C.byref(n)))
        return out

    def merge_vote_segments(self, segments):
        """Testing aid: the device-side merge of the ranks' answer blocks -> (status, RoundResult); status 1 = merged,
        2 = the voters disagree somewhere (rapid_sim_count_votes would then count through the histogram all-reduces)."""
        blob = np.ascontiguousarray(np.concatenate(segments))
        rr = RoundResult()
        status = C.c_int32(0)
        self.e._check(self.e._lib.rapid_debug_vote_merge(self.e._h, _addr(blob), len(segments), C.byref(rr), C.byref(status)))
        return status.value, rr

    def decided_cut(self):
        out = np.empty(self.e.max_cut, dtype=np.int32)
        n = C.c_int32(0)
        self.e._check(self.e._lib.rapid_sim_decided_cut(self.e._h, _addr(out), len(out), C.byref(n)))
        return out[: n.value].tolist()

    def round(self, apply=True):
        rr = RoundResult()
        cfg = C.c_int64(0)
        self.e._check(self.e._lib.rapid_sim_round(self.e._h, 1 if apply else 0, C.byref(rr), C.byref(cfg)))
        return rr, cfg.value

    def round_device(self, d_records_ptr, records_bytes, d_rec_off_ptr, n_receivers, d_alerts_ptr=None, n_alerts=0, trust=1,
                     apply=False, keepalive=None, alerts_bytes=None):
        """A round whose deliveries (and distinct alerts) already lie in device memory, in ONE library call
        (rapid_sim_round_device): attach in place + declare in place + trust level + index + tally + vote count (+ apply)."""
        rr = RoundResult()
        cfg = C.c_int64(0)
        nbytes = 20 * int(n_alerts) if alerts_bytes is None else int(alerts_bytes)
        keep_old = self._keep
        self._keep = (keep_old, keepalive)  # (both sets of buffers stay alive across the call: which one the engine holds depends on how far it gets)
        rc = self.e._lib.rapid_sim_round_device(self.e._h, d_records_ptr, records_bytes, d_rec_off_ptr, n_receivers, d_alerts_ptr, nbytes,
                                                n_alerts, int(trust), 1 if apply else 0, C.byref(rr), C.byref(cfg))
        self.e._check(rc)
        self._keep = keepalive
        self.n_receivers = n_receivers
        return rr, cfg.value

    def classic_round(self, arrival=None, apply=True):
        """The recovery the reference runs when the fast round finds no quorum (R/FastPaxos.java:107-109, 190-196 ->
        R/Paxos.java): ONE classic round over the receivers of the last tally, each an acceptor holding its fast-round
        vote; `arrival` = the order in which their Phase1b messages reach the coordinator (default: receiver order).
        Single-engine populations only.  -> (result dict of consensus.classic_round_population, decided cut or None,
        new configuration id or None)."""
        from . import consensus as CS
        n = C.c_int32(0)
        self.e._check(self.e._lib.rapid_view_size(self.e._h, C.byref(n)))
        emit, _, _, fp = self.results()
        res, winner = CS.classic_round_from_results(n.value, emit, fp, arrival)
        if winner is None:
            return res, None, None
        cut = self.proposal(winner)
        return res, cut, (self.apply_cut(cut) if apply else None)

    def apply_cut(self, cut):
        cut = np.ascontiguousarray(cut, dtype=np.int32)
        cfg = C.c_int64(0)
        self.e._check(self.e._lib.rapid_apply_cut(self.e._h, _addr(cut) if len(cut) else None, len(cut), C.byref(cfg)))
        return cfg.value

    def stats(self):
        s = np.zeros(8, dtype=np.uint64)
        self.e._check(self.e._lib.rapid_sim_stats(self.e._h, _addr(s)))
        return dict(exact_subchunks=int(s[0]), lean_windows=int(s[1]), full_sweeps=int(s[2]), restarts=int(s[3]),
                    implicit_reports=int(s[4]), records_consumed=int(s[5]), lean_give_ups=int(s[6]),
                    careful_subchunks=int(s[7]))

    def time_tally(self, reps):
        ms = C.c_float(0)
        self.e._check(self.e._lib.rapid_sim_time_tally(self.e._h, reps, C.byref(ms)))
        return ms.value

    def index_info(self, timed=True):
        """The round index of the loaded streams (built if stale).  timed: also ask for the build's device time -- the NEXT build
        (this call's own if the index is stale) is bracketed by timing events and reported as index_build_ms by the call after it."""
        info = np.zeros(8, dtype=np.int32)
        ms = C.c_float(0)
        self.e._check(self.e._lib.rapid_sim_index_info(self.e._h, _addr(info), C.byref(ms) if timed else None))
        keys = ("hot_subjects", "adjacency_entries", "waves_per_workgroup", "workgroups", "lds_bytes_per_workgroup",
                "alerts_prevalidated", "dict_mode", "alert_set_declared")
        out = {k: int(v) for k, v in zip(keys, info)}
        out["q4_live"] = (out["alert_set_declared"] >> 1) & 1  # a hot member's memoised observers are stale in this round (quirk Q4)
        out["configuration_ids_known_current"] = (out["alert_set_declared"] >> 2) & 1  # the tally loads {dst, word} only (kCurrent)
        out["alert_set_declared"] &= 1
        # dict_mode -- where the tally maps a boundary record's subject to its slot: 1 = direct tables in LDS, 2 = compressed
        # tables in LDS, 0 = tables in memory (through L2); 3 = nowhere: generated records carry their subjects' entries
        out["tables_in_lds"] = int(out["dict_mode"] in (1, 2))
        t = np.zeros(4, dtype=np.float32)
        self.e._check(self.e._lib.rapid_sim_pass_times(self.e._h, _addr(t)))
        out["index_build_ms"] = round(float(t[0]), 4)  # (the last build that was timed)
        out["generate_ms"] = round(float(t[2]), 4)
        return out

    def stream_probe(self, variant, waves, reps=10):
        ms = C.c_float(0)
        self.e._check(self.e._lib.rapid_debug_stream_probe(self.e._h, variant, waves, reps, C.byref(ms)))
        return ms.value

    def new_round(self):
        """Another round over the loaded streams: the per-round index is rebuilt by the next tally."""
        self.e._check(self.e._lib.rapid_sim_new_round(self.e._h))

    def set_force_exact(self, on):
        self.e._check(self.e._lib.rapid_sim_set_force_exact(self.e._h, int(on)))


class ObserverCacheGuard:
    """Quirk Q4 of the reference (R/MembershipView.java:143-152, 181-195, 210-224): getObserversOf is memoised per node of the
    cluster and an entry survives a view change that moves the ring minimum.  The only reader on the hot path is
    invalidateFailingEdges (R/MultiNodeCutDetector.java:147-149), which asks for the observers of the subjects in
    preProposal -- subjects the round's alerts name on >= L rings ("hot").  The engine always works from the fresh tables,
    so its results equal the reference's as long as no receiver can hold a STALE entry for a subject that is hot now.
    The bookkeeping lives behind the C ABI (rapid_view_q4_at_risk; updated by rapid_apply_cut / rapid_view_ring_delete);
    this class is the host-side handle tests and scripts use: `check_round` returns the members for which the condition
    fails (empty: Q4 cannot fire at any receiver in this round; non-empty: bit-exactness against the Java is not guaranteed
    for this round and the caller must say so)."""

    def check_round(self, view, hot_nodes):
        e = view.e
        hot = np.ascontiguousarray(hot_nodes, dtype=np.int32)
        out = np.empty(max(len(hot), 1), dtype=np.int32)
        n = C.c_int32(0)
        e._check(e._lib.rapid_view_q4_at_risk(e._h, _addr(hot) if len(hot) else None, len(hot), _addr(out), len(out), C.byref(n)))
        return out[: n.value].tolist()

    def on_view_change(self, removed):
        """(kept for callers of the round-2 interface: the engine forgets a node when it leaves the view)"""
        return None
