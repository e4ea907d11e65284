// Wire ingest: serialized protobuf messages of rapid.proto -> the packed records of this library,
// so that a live MembershipService can hand the bytes it received (R/MembershipService.java:174-196 dispatches a
// RapidRequest by its `content` case) to the engine without building Java objects per alert.  Host code only.
//
// The reference reads these messages through protobuf-java generated classes (com.google.protobuf 3.x, rapid/pom.xml,
// not vendored); what is restated here is the published proto3 wire format for the message shapes of
// rapid/src/main/proto/rapid.proto:13-17 (Endpoint), :20-34 (RapidRequest), :50-54 (NodeId), :95-115 (BatchedAlertMessage,
// AlertMessage, EdgeStatus), :124-129 (FastRoundPhase2bMessage).  tests/test_wire.py checks it against the Python protobuf
// runtime on messages built from the same field numbers.
#pragma once
#include <stdint.h>

#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/rapid_mi355x.h"

struct rapid_endpoint_map {
    std::unordered_map<std::string, int32_t> index;  // hostname bytes + '\0' + 4 port bytes -> node index
    std::vector<std::string> hostname;               // node index -> hostname bytes (for the encoders)
    std::vector<int32_t> port;
    static std::string key(const uint8_t* host, size_t host_len, int32_t port) {
        std::string k(reinterpret_cast<const char*>(host), host_len);
        k.push_back('\0');
        k.append(reinterpret_cast<const char*>(&port), 4);
        return k;
    }
};

namespace rapid_wire {

// A cursor over one length-delimited region.  Every read is bounds-checked; `ok` turns false on malformed input.
struct Reader {
    const uint8_t* p;
    const uint8_t* end;
    bool ok = true;
    Reader(const uint8_t* b, int64_t n) : p(b), end(b + (n > 0 ? n : 0)) {}
    bool done() const { return !ok || p >= end; }
    uint64_t varint() {
        uint64_t v = 0;
        for (int shift = 0; shift < 64; shift += 7) {
            if (p >= end) {
                ok = false;
                return 0;
            }
            const uint8_t b = *p++;
            v |= (uint64_t)(b & 0x7F) << shift;
            if (!(b & 0x80)) return v;
        }
        ok = false;  // more than 10 bytes
        return 0;
    }
    // reads a tag; returns the field number, sets the wire type
    uint32_t tag(int* wire_type) {
        const uint64_t t = varint();
        *wire_type = (int)(t & 7u);
        return (uint32_t)(t >> 3);
    }
    Reader sub() {  // length-delimited payload
        const uint64_t n = varint();
        if (!ok || n > (uint64_t)(end - p)) {
            ok = false;
            return Reader(p, 0);
        }
        Reader r(p, (int64_t)n);
        p += n;
        return r;
    }
    void skip(int wire_type) {
        switch (wire_type) {
            case 0: (void)varint(); break;
            case 1: if (end - p < 8) ok = false; else p += 8; break;
            case 2: (void)sub(); break;
            case 5: if (end - p < 4) ok = false; else p += 4; break;
            default: ok = false;  // groups are not used by rapid.proto
        }
    }
};

// remoting.Endpoint { bytes hostname = 1; int32 port = 2; } -> node index, or -1 if it is not in the map
inline int32_t read_endpoint(Reader r, const rapid_endpoint_map& m, bool* ok) {
    const uint8_t* host = nullptr;
    size_t host_len = 0;
    int32_t port = 0;  // proto3: a zero port is not on the wire
    while (!r.done()) {
        int wt;
        const uint32_t f = r.tag(&wt);
        if (f == 1 && wt == 2) {
            Reader h = r.sub();
            host = h.p;
            host_len = (size_t)(h.end - h.p);
        } else if (f == 2 && wt == 0) {
            port = (int32_t)r.varint();
        } else {
            r.skip(wt);
        }
    }
    if (!r.ok) {
        *ok = false;
        return -1;
    }
    static const uint8_t empty = 0;
    const auto it = m.index.find(rapid_endpoint_map::key(host ? host : &empty, host_len, port));
    return it == m.index.end() ? -1 : it->second;
}

// Parses a serialized remoting.Endpoint into (hostname bytes, port).
inline bool parse_endpoint(Reader r, const uint8_t** host, size_t* host_len, int32_t* port) {
    static const uint8_t empty = 0;
    *host = &empty;
    *host_len = 0;
    *port = 0;
    while (!r.done()) {
        int wt;
        const uint32_t f = r.tag(&wt);
        if (f == 1 && wt == 2) {
            Reader h = r.sub();
            *host = h.p;
            *host_len = (size_t)(h.end - h.p);
        } else if (f == 2 && wt == 0) {
            *port = (int32_t)r.varint();
        } else {
            r.skip(wt);
        }
    }
    return r.ok;
}

// remoting.NodeId { int64 high = 1; int64 low = 2; }
inline void read_node_id(Reader r, int64_t* hi, int64_t* lo, bool* ok) {
    *hi = 0;
    *lo = 0;
    while (!r.done()) {
        int wt;
        const uint32_t f = r.tag(&wt);
        if (f == 1 && wt == 0)
            *hi = (int64_t)r.varint();
        else if (f == 2 && wt == 0)
            *lo = (int64_t)r.varint();
        else
            r.skip(wt);
    }
    if (!r.ok) *ok = false;
}

// remoting.AlertMessage -> one packed record.  Returns RAPID_OK, RAPID_EINVAL (malformed, ring number out of range) or
// RAPID_ENODE_MISSING (an endpoint that is not in the map).
// unresolved (optional): [0] / [1] = offset (from `base`) and length of the serialized Endpoint of the FIRST endpoint of the
// alert that the map does not know (edgeSrc before edgeDst), untouched otherwise.
inline int read_alert(Reader r, const rapid_endpoint_map& m, int K, rapid_alert_record* rec, int64_t* id_hi, int64_t* id_lo,
                      const uint8_t* base = nullptr, int64_t* unresolved = nullptr) {
    bool ok = true;
    int32_t src = -1, dst = -1;
    const uint8_t* src_p = nullptr;
    const uint8_t* dst_p = nullptr;
    int64_t src_n = 0, dst_n = 0;
    bool have_src = false, have_dst = false;
    uint32_t mask = 0;
    int64_t cfg = 0;
    uint32_t status = 0;  // EdgeStatus.UP = 0 is the proto3 default and is not on the wire
    int64_t hi = 0, lo = 0;
    auto add_ring = [&](uint64_t v) {
        const int32_t ring = (int32_t)v;
        if (ring < 0 || ring >= K || ring > 13)
            ok = false;  // the Java only asserts it (Q2); the packed mask has room for 14 rings
        else
            mask |= 1u << ring;
    };
    while (!r.done()) {
        int wt;
        const uint32_t f = r.tag(&wt);
        if (f == 1 && wt == 2) {
            const Reader e = r.sub();
            src_p = e.p;
            src_n = (int64_t)(e.end - e.p);
            src = read_endpoint(e, m, &ok);
            have_src = true;
        } else if (f == 2 && wt == 2) {
            const Reader e = r.sub();
            dst_p = e.p;
            dst_n = (int64_t)(e.end - e.p);
            dst = read_endpoint(e, m, &ok);
            have_dst = true;
        } else if (f == 3 && wt == 0) {
            status = (uint32_t)r.varint();
        } else if (f == 4 && wt == 0) {
            cfg = (int64_t)r.varint();
        } else if (f == 5 && wt == 2) {  // packed repeated int32 (the proto3 default)
            Reader pk = r.sub();
            while (!pk.done()) add_ring(pk.varint());
            if (!pk.ok) ok = false;
        } else if (f == 5 && wt == 0) {  // unpacked form: parsers must accept both
            add_ring(r.varint());
        } else if (f == 6 && wt == 2) {
            read_node_id(r.sub(), &hi, &lo, &ok);
        } else {
            r.skip(wt);  // metadata (7) and anything newer
        }
    }
    if (!r.ok || !ok || status > 1u) return RAPID_EINVAL;
    if (id_hi) *id_hi = hi;  // the NodeId of a joiner is what the facade needs to register it
    if (id_lo) *id_lo = lo;
    if (!have_src || !have_dst || src < 0 || dst < 0) {
        if (unresolved && base) {
            if (have_src && src < 0) {
                unresolved[0] = (int64_t)(src_p - base);
                unresolved[1] = src_n;
            } else if (have_dst && dst < 0) {
                unresolved[0] = (int64_t)(dst_p - base);
                unresolved[1] = dst_n;
            }
        }
        return RAPID_ENODE_MISSING;
    }
    rec->cfg_id = cfg;
    rec->src = (uint32_t)src;
    rec->dst = (uint32_t)dst;
    rec->ring_mask = (uint16_t)mask;
    rec->status = (uint8_t)status;
    rec->flags = 0;
    if (id_hi) *id_hi = hi;
    if (id_lo) *id_lo = lo;
    return RAPID_OK;
}

// remoting.Rank { int32 round = 1; int32 nodeIndex = 2; } (rapid.proto:133-137)
inline rapid_rank read_rank(Reader r, bool* ok) {
    rapid_rank k{0, 0};
    while (!r.done()) {
        int wt;
        const uint32_t f = r.tag(&wt);
        if (f == 1 && wt == 0)
            k.round = (int32_t)r.varint();
        else if (f == 2 && wt == 0)
            k.node_index = (int32_t)r.varint();
        else
            r.skip(wt);
    }
    if (!r.ok) *ok = false;
    return k;
}

// Serializer for the handful of message shapes the consensus path sends.  Appends to a byte vector.
struct Writer {
    std::vector<uint8_t> b;
    void varint(uint64_t v) {
        while (v >= 0x80) {
            b.push_back((uint8_t)(v | 0x80));
            v >>= 7;
        }
        b.push_back((uint8_t)v);
    }
    void tag(uint32_t field, int wire_type) { varint(((uint64_t)field << 3) | (uint64_t)wire_type); }
    void int_field(uint32_t field, int64_t v) {  // int32 / int64: zero is not written, negatives take ten bytes
        if (v == 0) return;
        tag(field, 0);
        varint((uint64_t)v);
    }
    void bytes_field(uint32_t field, const uint8_t* p, size_t n, bool even_if_empty) {
        if (n == 0 && !even_if_empty) return;
        tag(field, 2);
        varint(n);
        b.insert(b.end(), p, p + n);
    }
    void message_field(uint32_t field, const Writer& sub) { bytes_field(field, sub.b.data(), sub.b.size(), true); }
};

inline Writer write_endpoint(const rapid_endpoint_map& m, int32_t node) {
    Writer w;
    const std::string& h = m.hostname[(size_t)node];
    w.bytes_field(1, reinterpret_cast<const uint8_t*>(h.data()), h.size(), false);
    w.int_field(2, m.port[(size_t)node]);
    return w;
}

inline Writer write_rank(rapid_rank k) {
    Writer w;
    w.int_field(1, k.round);
    w.int_field(2, k.node_index);
    return w;
}

}  // namespace rapid_wire
