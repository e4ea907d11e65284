"""Host mirror of the wire-ingest entry points of include/rapid_mi355x.h: serialized protobuf messages of
rapid.proto -> packed alert records / votes (what MembershipService.handleMessage dispatches on,
R/MembershipService.java:174-196).  Pure host code: works without a GPU."""
import ctypes as C

import numpy as np

from . import _native as N
from .engine import _addr
from .scenarios import ALERT_DTYPE


def _raise_for(rc, what):
    if rc != N.OK:
        N.raise_for(rc, what)

MSG_OTHER, MSG_BATCHED_ALERT, MSG_FAST_ROUND_2B = 0, 3, 5


def _bytes_addr(b):
    arr = np.frombuffer(b, dtype=np.uint8) if len(b) else np.zeros(1, dtype=np.uint8)
    return arr, _addr(arr)


class EndpointMap:
    """(hostname bytes, port) -> node index, built from the registry that MembershipView.build receives."""

    def __init__(self, hostnames, ports):
        self._lib = N.lib()
        blob = np.frombuffer(b"".join(hostnames), dtype=np.uint8).copy() if len(hostnames) else np.zeros(1, dtype=np.uint8)
        off = np.zeros(len(hostnames) + 1, dtype=np.int32)
        off[1:] = np.cumsum([len(h) for h in hostnames])
        ports = np.ascontiguousarray(ports, dtype=np.int32)
        h = C.c_void_p()
        rc = self._lib.rapid_endpoint_map_create(_addr(blob), _addr(off), _addr(ports) if len(ports) else None, len(hostnames), C.byref(h))
        _raise_for(rc, "rapid_endpoint_map_create")
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.rapid_endpoint_map_destroy(self._h)
            self._h = None

    def lookup(self, hostname, port):
        arr, a = _bytes_addr(hostname)
        out = C.c_int32(-1)
        rc = self._lib.rapid_endpoint_map_lookup(self._h, a, len(hostname), port, C.byref(out))
        _raise_for(rc, "rapid_endpoint_map_lookup")
        return out.value

    def decode_batched_alerts(self, msg, K, cap=4096):
        """-> (records[ALERT_DTYPE], node ids [(hi, lo)], sender index)"""
        arr, a = _bytes_addr(msg)
        out = np.zeros(cap, dtype=ALERT_DTYPE)
        hi = np.zeros(cap, dtype=np.int64)
        lo = np.zeros(cap, dtype=np.int64)
        n, sender = C.c_int32(0), C.c_int32(-1)
        rc = self._lib.rapid_decode_batched_alerts(self._h, a, len(msg), K, _addr(out), _addr(hi), _addr(lo), cap, C.byref(n),
                                                   C.byref(sender))
        _raise_for(rc, "rapid_decode_batched_alerts")
        return out[: n.value].copy(), list(zip(hi[: n.value].tolist(), lo[: n.value].tolist())), sender.value

    def decode_batched_alerts_ex(self, msg, K, cap=4096):
        """Every alert of the message on its own (rapid_decode_batched_alerts_ex): -> (records, node ids, status per alert,
        serialized Endpoint of the first unknown endpoint per alert or None, sender index)."""
        arr, a = _bytes_addr(msg)
        out = np.zeros(cap, dtype=ALERT_DTYPE)
        hi = np.zeros(cap, dtype=np.int64)
        lo = np.zeros(cap, dtype=np.int64)
        status = np.zeros(cap, dtype=np.int32)
        unresolved = np.full(2 * cap, -1, dtype=np.int64)
        n, sender = C.c_int32(0), C.c_int32(-1)
        rc = self._lib.rapid_decode_batched_alerts_ex(self._h, a, len(msg), K, _addr(out), _addr(hi), _addr(lo), _addr(status),
                                                      _addr(unresolved), cap, C.byref(n), C.byref(sender))
        _raise_for(rc, "rapid_decode_batched_alerts_ex")
        k = n.value
        unknown = [bytes(msg[unresolved[2 * i]: unresolved[2 * i] + unresolved[2 * i + 1]]) if status[i] == N.ENODE_MISSING else None
                   for i in range(k)]
        return out[:k].copy(), list(zip(hi[:k].tolist(), lo[:k].tolist())), status[:k].tolist(), unknown, sender.value

    def add(self, hostname, port):
        """Registers an endpoint (next free index) unless it is known; -> its index."""
        arr, a = _bytes_addr(hostname)
        out = C.c_int32(-1)
        _raise_for(self._lib.rapid_endpoint_map_add(self._h, a, len(hostname), port, C.byref(out)), "rapid_endpoint_map_add")
        return out.value

    def add_wire(self, endpoint_msg):
        """The same from a serialized remoting.Endpoint."""
        arr, a = _bytes_addr(endpoint_msg)
        out = C.c_int32(-1)
        _raise_for(self._lib.rapid_endpoint_map_add_wire(self._h, a, len(endpoint_msg), C.byref(out)), "rapid_endpoint_map_add_wire")
        return out.value

    def size(self):
        return int(self._lib.rapid_endpoint_map_size(self._h))

    def get(self, node):
        buf = np.zeros(512, dtype=np.uint8)
        ln, port = C.c_int32(0), C.c_int32(0)
        _raise_for(self._lib.rapid_endpoint_map_get(self._h, node, _addr(buf), len(buf), C.byref(ln), C.byref(port)), "rapid_endpoint_map_get")
        return bytes(buf[: ln.value]), int(port.value)

    def decode_fast_round_vote(self, msg, cap=4096):
        """-> (sender index, configuration id, endpoint indices in message order)"""
        arr, a = _bytes_addr(msg)
        out = np.zeros(max(cap, 1), dtype=np.int32)
        n, sender, cfg = C.c_int32(0), C.c_int32(-1), C.c_int64(0)
        rc = self._lib.rapid_decode_fast_round_vote(self._h, a, len(msg), C.byref(sender), C.byref(cfg), _addr(out), cap, C.byref(n))
        _raise_for(rc, "rapid_decode_fast_round_vote")
        return sender.value, cfg.value, out[: n.value].tolist()


def decode_request(msg):
    """-> (content case = field number of the RapidRequest oneof, payload bytes)"""
    arr, a = _bytes_addr(msg)
    kind, off, ln = C.c_int32(0), C.c_int64(0), C.c_int64(0)
    rc = N.lib().rapid_decode_request(a, len(msg), C.byref(kind), C.byref(off), C.byref(ln))
    _raise_for(rc, "rapid_decode_request")
    return kind.value, bytes(msg[off.value: off.value + ln.value])
