"""TEST INFRASTRUCTURE ONLY -- the message shapes of rapid/src/main/proto/rapid.proto that the hot path ingests, built as
dynamic descriptors for the Python protobuf runtime (no protoc in this image): the independent implementation of the
proto3 wire format that tests/test_wire.py checks rapid_amd/csrc/wire.h against.  Field names and numbers follow
rapid.proto:13-17 (Endpoint), :20-34 (RapidRequest), :50-54 (NodeId), :95-115 (alerts), :124-129 (fast-round vote), :133-169 (classic Paxos);
messages that the path does not read are declared empty."""
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

_T = descriptor_pb2.FieldDescriptorProto


def _field(m, name, number, ftype, repeated=False, type_name=None, oneof=None):
    f = m.field.add()
    f.name, f.number, f.type = name, number, ftype
    f.label = _T.LABEL_REPEATED if repeated else _T.LABEL_OPTIONAL
    if type_name:
        f.type_name = ".remoting." + type_name
    if oneof is not None:
        f.oneof_index = oneof
    return f


def _build():
    fdp = descriptor_pb2.FileDescriptorProto()
    fdp.name, fdp.package, fdp.syntax = "rapid_subset.proto", "remoting", "proto3"
    e = fdp.enum_type.add()
    e.name = "EdgeStatus"
    for name, num in (("UP", 0), ("DOWN", 1)):
        v = e.value.add()
        v.name, v.number = name, num
    m = fdp.message_type.add(); m.name = "Endpoint"
    _field(m, "hostname", 1, _T.TYPE_BYTES)
    _field(m, "port", 2, _T.TYPE_INT32)
    m = fdp.message_type.add(); m.name = "NodeId"
    _field(m, "high", 1, _T.TYPE_INT64)
    _field(m, "low", 2, _T.TYPE_INT64)
    m = fdp.message_type.add(); m.name = "Metadata"
    m = fdp.message_type.add(); m.name = "AlertMessage"
    _field(m, "edgeSrc", 1, _T.TYPE_MESSAGE, type_name="Endpoint")
    _field(m, "edgeDst", 2, _T.TYPE_MESSAGE, type_name="Endpoint")
    _field(m, "edgeStatus", 3, _T.TYPE_ENUM, type_name="EdgeStatus")
    _field(m, "configurationId", 4, _T.TYPE_INT64)
    _field(m, "ringNumber", 5, _T.TYPE_INT32, repeated=True)
    _field(m, "nodeId", 6, _T.TYPE_MESSAGE, type_name="NodeId")
    _field(m, "metadata", 7, _T.TYPE_MESSAGE, type_name="Metadata")
    m = fdp.message_type.add(); m.name = "BatchedAlertMessage"
    _field(m, "sender", 1, _T.TYPE_MESSAGE, type_name="Endpoint")
    _field(m, "messages", 3, _T.TYPE_MESSAGE, repeated=True, type_name="AlertMessage")
    m = fdp.message_type.add(); m.name = "FastRoundPhase2bMessage"
    _field(m, "sender", 1, _T.TYPE_MESSAGE, type_name="Endpoint")
    _field(m, "configurationId", 2, _T.TYPE_INT64)
    _field(m, "endpoints", 3, _T.TYPE_MESSAGE, repeated=True, type_name="Endpoint")
    m = fdp.message_type.add(); m.name = "Rank"  # rapid.proto:133-137
    _field(m, "round", 1, _T.TYPE_INT32)
    _field(m, "nodeIndex", 2, _T.TYPE_INT32)
    m = fdp.message_type.add(); m.name = "Phase1aMessage"  # :139-144
    _field(m, "sender", 1, _T.TYPE_MESSAGE, type_name="Endpoint")
    _field(m, "configurationId", 2, _T.TYPE_INT64)
    _field(m, "rank", 3, _T.TYPE_MESSAGE, type_name="Rank")
    m = fdp.message_type.add(); m.name = "Phase1bMessage"  # :146-153
    _field(m, "sender", 1, _T.TYPE_MESSAGE, type_name="Endpoint")
    _field(m, "configurationId", 2, _T.TYPE_INT64)
    _field(m, "rnd", 3, _T.TYPE_MESSAGE, type_name="Rank")
    _field(m, "vrnd", 4, _T.TYPE_MESSAGE, type_name="Rank")
    _field(m, "vval", 5, _T.TYPE_MESSAGE, repeated=True, type_name="Endpoint")
    m = fdp.message_type.add(); m.name = "Phase2aMessage"  # :155-161
    _field(m, "sender", 1, _T.TYPE_MESSAGE, type_name="Endpoint")
    _field(m, "configurationId", 2, _T.TYPE_INT64)
    _field(m, "rnd", 3, _T.TYPE_MESSAGE, type_name="Rank")
    _field(m, "vval", 5, _T.TYPE_MESSAGE, repeated=True, type_name="Endpoint")
    m = fdp.message_type.add(); m.name = "Phase2bMessage"  # :163-169
    _field(m, "sender", 1, _T.TYPE_MESSAGE, type_name="Endpoint")
    _field(m, "configurationId", 2, _T.TYPE_INT64)
    _field(m, "rnd", 3, _T.TYPE_MESSAGE, type_name="Rank")
    _field(m, "endpoints", 4, _T.TYPE_MESSAGE, repeated=True, type_name="Endpoint")
    m = fdp.message_type.add(); m.name = "ProbeMessage"
    _field(m, "sender", 1, _T.TYPE_MESSAGE, type_name="Endpoint")
    m = fdp.message_type.add(); m.name = "RapidRequest"
    m.oneof_decl.add().name = "content"
    _field(m, "batchedAlertMessage", 3, _T.TYPE_MESSAGE, type_name="BatchedAlertMessage", oneof=0)
    _field(m, "probeMessage", 4, _T.TYPE_MESSAGE, type_name="ProbeMessage", oneof=0)
    _field(m, "fastRoundPhase2bMessage", 5, _T.TYPE_MESSAGE, type_name="FastRoundPhase2bMessage", oneof=0)
    _field(m, "phase1aMessage", 6, _T.TYPE_MESSAGE, type_name="Phase1aMessage", oneof=0)
    _field(m, "phase1bMessage", 7, _T.TYPE_MESSAGE, type_name="Phase1bMessage", oneof=0)
    _field(m, "phase2aMessage", 8, _T.TYPE_MESSAGE, type_name="Phase2aMessage", oneof=0)
    _field(m, "phase2bMessage", 9, _T.TYPE_MESSAGE, type_name="Phase2bMessage", oneof=0)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fdp)
    out = {}
    for name in ("Endpoint", "NodeId", "AlertMessage", "BatchedAlertMessage", "FastRoundPhase2bMessage", "ProbeMessage",
                 "RapidRequest", "Rank", "Phase1aMessage", "Phase1bMessage", "Phase2aMessage", "Phase2bMessage"):
        out[name] = message_factory.GetMessageClass(pool.FindMessageTypeByName("remoting." + name))
    return out


globals().update(_build())
UP, DOWN = 0, 1
