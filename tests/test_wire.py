"""Wire ingest: rapid_amd/csrc/wire.h decodes serialized rapid.proto messages into packed records.
The checker is the Python protobuf runtime (an independent implementation of the proto3 wire format) on messages
built from the reference's field numbers (tests/proto_rapid.py).  Host only: runs without a GPU."""
import numpy as np
import pytest

from oracle import pyoracle as O
from rapid_amd import _native as N
from rapid_amd import scenarios as S
from rapid_amd import wire as W
from tests import proto_rapid as P
from tests.helpers import oracle_view


def ep(pop, i):
    return P.Endpoint(hostname=pop.hostnames[i], port=int(pop.ports[i]))


def alert_msg(pop, rec, node_id=None):
    a = P.AlertMessage(edgeSrc=ep(pop, int(rec["src"])), edgeDst=ep(pop, int(rec["dst"])), edgeStatus=int(rec["status"]),
                       configurationId=int(rec["cfg_id"]))
    a.ringNumber.extend([k for k in range(16) if (int(rec["ring_mask"]) >> k) & 1])
    if node_id is not None:
        a.nodeId.high, a.nodeId.low = node_id
    return a


def test_endpoint_map_lookup():
    pop = S.Population.make(300)
    m = W.EndpointMap(pop.hostnames, pop.ports)
    for i in (0, 1, 17, 299):
        assert m.lookup(pop.hostnames[i], int(pop.ports[i])) == i
    with pytest.raises(N.NodeNotInRingException):
        m.lookup(b"no.such.host", 1)
    with pytest.raises(N.NodeNotInRingException):
        m.lookup(pop.hostnames[3], int(pop.ports[3]) + 1)


def test_scenario_batches_round_trip_through_the_wire_format():
    """Every batch of a churn round (DOWN alerts, UP alerts with NodeIds, multi-ring masks) serialized the way the
    reference sends it (UnicastToAllBroadcaster -> RapidRequest{batchedAlertMessage}) and decoded back: same records."""
    n, K, H, L = 400, 10, 9, 4
    pop = S.Population.make(n)
    reg, view = oracle_view(pop, K, list(range(0, n - 40)))
    obs, subj, member = view.tables(n)
    cfg = view.getCurrentConfigurationId()
    sc = S.build_churn_scenario(obs, member, cfg, 12, 25, H, L, materialise=False)
    m = W.EndpointMap(pop.hostnames, pop.ports)
    bs = sc.batches
    assert bs.n_batches > 50 and set(np.unique(bs.recs["status"]).tolist()) == {S.UP, S.DOWN}
    for b in range(bs.n_batches):
        recs = bs.recs[bs.off[b]:bs.off[b + 1]]
        msg = P.BatchedAlertMessage(sender=ep(pop, int(bs.sender[b])))
        want_ids = []
        for rec in recs:
            joiner = int(rec["status"]) == S.UP
            nid = (int(pop.id_hi[int(rec["dst"])]), int(pop.id_lo[int(rec["dst"])])) if joiner else None
            msg.messages.append(alert_msg(pop, rec, nid))
            want_ids.append(nid if joiner else (0, 0))
        req = P.RapidRequest(batchedAlertMessage=msg).SerializeToString()
        kind, payload = W.decode_request(req)
        assert kind == W.MSG_BATCHED_ALERT and payload == msg.SerializeToString()
        got, ids, sender = m.decode_batched_alerts(payload, K)
        assert sender == int(bs.sender[b]) and ids == want_ids
        assert got.tobytes() == recs.tobytes()  # field for field, including the end-of-batch flag on the last record


def test_proto3_defaults_unpacked_rings_and_unknown_fields():
    pop = S.Population.make(8)
    m = W.EndpointMap(pop.hostnames + [b"zero.port"], list(pop.ports) + [0])
    # defaults are not on the wire: status UP (0), configuration id 0, port 0, no rings
    a = P.AlertMessage(edgeSrc=ep(pop, 1), edgeDst=P.Endpoint(hostname=b"zero.port"))
    got, ids, sender = m.decode_batched_alerts(P.BatchedAlertMessage(messages=[a]).SerializeToString(), 10)
    assert sender == -1 and len(got) == 1
    r = got[0]
    assert (r["src"], r["dst"], r["status"], r["cfg_id"], r["ring_mask"], r["flags"]) == (1, 8, S.UP, 0, 0, S.FLAG_LAST_IN_BATCH)
    # hand-written bytes: ring numbers UNPACKED (tag 0x28 per element), an unknown varint field 15 and an unknown
    # length-delimited field 16 inside the alert, an unknown fixed32 field 9 in the batch
    src, dst = ep(pop, 2).SerializeToString(), ep(pop, 5).SerializeToString()
    alert = (b"\x0a" + bytes([len(src)]) + src + b"\x12" + bytes([len(dst)]) + dst + b"\x18\x01" + b"\x20\x2a" +
             b"\x28\x00" + b"\x28\x07" + b"\x78\x05" + b"\x82\x01\x03abc" + b"\x28\x03")
    batch = b"\x4d\x01\x02\x03\x04" + b"\x1a" + bytes([len(alert)]) + alert
    got, ids, sender = m.decode_batched_alerts(batch, 10)
    r = got[0]
    assert (r["src"], r["dst"], r["status"], r["cfg_id"], r["ring_mask"]) == (2, 5, S.DOWN, 42, (1 << 0) | (1 << 7) | (1 << 3))
    # the runtime parses the same bytes the same way
    ref = P.BatchedAlertMessage.FromString(batch).messages[0]
    assert list(ref.ringNumber) == [0, 7, 3] and ref.configurationId == 42 and ref.edgeStatus == P.DOWN
    # negative 64-bit values take ten bytes
    a = alert_msg(pop, np.array([(-(1 << 63), 1, 2, 1, S.DOWN, 0)], dtype=S.ALERT_DTYPE)[0], node_id=(-5, -(1 << 62)))
    got, ids, _ = m.decode_batched_alerts(P.BatchedAlertMessage(messages=[a]).SerializeToString(), 10)
    assert got[0]["cfg_id"] == -(1 << 63) and ids == [(-5, -(1 << 62))]


def test_errors():
    pop = S.Population.make(8)
    m = W.EndpointMap(pop.hostnames, pop.ports)
    good = P.BatchedAlertMessage(sender=ep(pop, 0), messages=[alert_msg(pop, np.array([(7, 1, 2, 0b101, S.DOWN, 0)],
                                                                            dtype=S.ALERT_DTYPE)[0])]).SerializeToString()
    assert len(m.decode_batched_alerts(good, 10)[0]) == 1
    for cut in range(1, len(good)):  # every truncation is either rejected or a shorter valid message, never a crash
        try:
            m.decode_batched_alerts(good[:cut], 10)
        except (N.IllegalArgumentException, N.NodeNotInRingException):
            pass
    with pytest.raises(N.IllegalArgumentException):
        m.decode_batched_alerts(good, 2)  # ring number 2 >= K
    stranger = P.AlertMessage(edgeSrc=ep(pop, 1), edgeDst=P.Endpoint(hostname=b"stranger", port=9))
    with pytest.raises(N.NodeNotInRingException):
        m.decode_batched_alerts(P.BatchedAlertMessage(messages=[stranger]).SerializeToString(), 10)
    with pytest.raises(N.RapidError):
        m.decode_batched_alerts(P.BatchedAlertMessage(messages=[alert_msg(pop, np.array([(7, 1, 2, 1, 1, 0)], dtype=S.ALERT_DTYPE)[0])] * 5
                                                      ).SerializeToString(), 10, cap=3)
    with pytest.raises(N.IllegalArgumentException):
        W.decode_request(b"\x1a\x7f\x00")  # length runs past the end
    assert W.decode_request(b"") == (W.MSG_OTHER, b"")
    kind, payload = W.decode_request(P.RapidRequest(probeMessage=P.ProbeMessage(sender=ep(pop, 1))).SerializeToString())
    assert kind == 4  # not a message of this path: the caller keeps it in Java


def test_fast_round_vote_feeds_the_fast_round_object():
    """FastRoundPhase2bMessage bytes -> (sender, configuration id, endpoints) -> the same decision as the oracle's fast
    round (R/FastPaxos.java:125-156)."""
    n = 9
    pop = S.Population.make(n)
    m = W.EndpointMap(pop.hostnames, pop.ports)
    cfg = -123456789012345
    proposal = [4, 2, 7]
    fr = O.FastRound(cfg, n)
    decided = None
    for sender in range(n):
        vote = P.FastRoundPhase2bMessage(sender=ep(pop, sender), configurationId=cfg, endpoints=[ep(pop, e) for e in proposal])
        req = P.RapidRequest(fastRoundPhase2bMessage=vote).SerializeToString()
        kind, payload = W.decode_request(req)
        assert kind == W.MSG_FAST_ROUND_2B
        s, c, eps = m.decode_fast_round_vote(payload)
        assert (s, c, eps) == (sender, cfg, proposal)
        if fr.handleFastRoundProposal(s, c, eps) and decided is None:
            decided = (sender, fr.decided())
    quorum = n - (n - 1) // 4
    assert decided is not None and decided[0] == quorum - 1 and decided[1] == proposal
    with pytest.raises(N.NodeNotInRingException):
        m.decode_fast_round_vote(P.FastRoundPhase2bMessage(sender=ep(pop, 1), configurationId=1,
                                                           endpoints=[P.Endpoint(hostname=b"x", port=1)]).SerializeToString())


def test_up_alert_about_an_endpoint_not_yet_in_the_map():
    """The join path of a live service (R/MembershipService.java:677-685): most receivers first hear of a joiner through an UP
    alert.  The map built at start-up does not know its endpoint: rapid_decode_batched_alerts_ex still decodes every other
    alert of the batch, reports the unknown one with the bytes of its Endpoint and its NodeId, the facade registers it
    (rapid_endpoint_map_add_wire) and the second decode yields the whole batch."""
    n, K = 60, 10
    pop = S.Population.make(n)
    known = 50  # endpoints 50..59 are strangers to the map
    m = W.EndpointMap(pop.hostnames[:known], pop.ports[:known])
    recs = np.zeros(4, dtype=S.ALERT_DTYPE)
    recs["src"], recs["dst"] = [3, 3, 4, 55], [7, 55, 56, 8]
    recs["ring_mask"], recs["status"], recs["cfg_id"] = [1, 6, 8, 2], [S.DOWN, S.UP, S.UP, S.DOWN], 99
    msg = P.BatchedAlertMessage(sender=ep(pop, 3))
    for i, rec in enumerate(recs):
        msg.messages.append(alert_msg(pop, rec, (100 + i, 200 + i) if rec["status"] == S.UP else None))
    payload = msg.SerializeToString()
    with pytest.raises(N.NodeNotInRingException):
        m.decode_batched_alerts(payload, K)  # the all-or-nothing entry point still says so
    got, ids, status, unknown, sender = m.decode_batched_alerts_ex(payload, K)
    assert sender == 3 and status == [N.OK, N.ENODE_MISSING, N.ENODE_MISSING, N.ENODE_MISSING]
    assert got[0].tobytes()[:19] == recs[0].tobytes()[:19] and got[0]["flags"] == S.FLAG_LAST_IN_BATCH  # the last usable record closes the batch
    assert ids[1] == (101, 201) and ids[2] == (102, 202)
    assert unknown[0] is None and P.Endpoint.FromString(unknown[1]).hostname == pop.hostnames[55]
    assert P.Endpoint.FromString(unknown[2]).hostname == pop.hostnames[56]
    assert P.Endpoint.FromString(unknown[3]).hostname == pop.hostnames[55]  # edgeSrc is looked at first
    first = m.size()
    assert first == known
    assert m.add_wire(unknown[1]) == known and m.add_wire(unknown[2]) == known + 1 and m.add_wire(unknown[3]) == known  # idempotent
    assert m.size() == known + 2 and m.get(known) == (pop.hostnames[55], int(pop.ports[55]))
    got, ids, status, unknown, sender = m.decode_batched_alerts_ex(payload, K)
    assert status == [N.OK] * 4 and unknown == [None] * 4
    want = recs.copy()
    want["src"], want["dst"] = [3, 3, 4, known], [7, known, known + 1, 8]  # the map's own numbering of the newcomers
    want["flags"][-1] = S.FLAG_LAST_IN_BATCH
    assert got.tobytes() == want.tobytes()
    got2, ids2, sender2 = m.decode_batched_alerts(payload, K)
    assert got2.tobytes() == want.tobytes() and ids2 == ids
    # malformed alert inside an otherwise good batch: reported for that alert only
    bad = P.BatchedAlertMessage(sender=ep(pop, 3))
    bad.messages.append(alert_msg(pop, recs[0]))
    a = alert_msg(pop, recs[0])
    a.ringNumber.append(77)  # not a ring of this cluster
    bad.messages.append(a)
    got, ids, status, unknown, sender = m.decode_batched_alerts_ex(bad.SerializeToString(), K)
    assert status == [N.OK, N.EINVAL] and got[0]["flags"] == S.FLAG_LAST_IN_BATCH
