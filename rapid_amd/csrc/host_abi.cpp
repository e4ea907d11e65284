// Host-only part of the C ABI (include/rapid_mi355x.h): the FastPaxos vote counter of one node, the wire ingest of
// serialized rapid.proto messages and the consensus state machine.  No device code and no HIP calls: these are the
// control-plane objects a MembershipService holds per configuration; the data-parallel work is in engine.hip.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <unordered_map>
#include <deque>
#include <vector>

#include "../../include/rapid_mi355x.h"
#include "consensus.h"
#include "wire.h"

struct rapid_fast_round {
    int64_t config_id = 0;
    long membership_size = 0;
    std::map<std::vector<int32_t>, int> votes_per_proposal;
    std::set<int32_t> votes_received;
    bool decided = false;
    std::vector<int32_t> decision;
};

extern "C" {

// -------------------------------------------------------------------------------------- fast round object
int rapid_fast_round_create(int64_t config_id, int32_t membership_size, rapid_fast_round** out) {
    if (!out || membership_size < 0) return RAPID_EINVAL;
    rapid_fast_round* f = new rapid_fast_round();
    f->config_id = config_id;
    f->membership_size = membership_size;
    *out = f;
    return RAPID_OK;
}

void rapid_fast_round_destroy(rapid_fast_round* f) { delete f; }

int rapid_fast_round_vote(rapid_fast_round* f, int32_t sender, int64_t config_id, const int32_t* endpoints, int32_t n,
                          int32_t* decided_out) {
    if (!f || n < 0 || (n > 0 && !endpoints)) return RAPID_EINVAL;
    if (decided_out) *decided_out = f->decided ? 1 : 0;
    if (config_id != f->config_id) return RAPID_OK;       // R/FastPaxos.java:126-132
    if (f->votes_received.count(sender)) return RAPID_OK;  // :134-136
    if (f->decided) return RAPID_OK;                       // :138-140
    f->votes_received.insert(sender);
    std::vector<int32_t> key(endpoints, endpoints + n);
    const int count = ++f->votes_per_proposal[key];
    const long F = (long)std::floor((double)(f->membership_size - 1) / 4.0);  // :145
    if ((long)f->votes_received.size() >= f->membership_size - F && count >= f->membership_size - F) {
        f->decided = true;
        f->decision = key;
    }
    if (decided_out) *decided_out = f->decided ? 1 : 0;
    return RAPID_OK;
}

int rapid_fast_round_decision(rapid_fast_round* f, int32_t* out, int32_t cap, int32_t* n_out) {
    if (!f || cap < 0 || (cap > 0 && !out)) return RAPID_EINVAL;
    if (!f->decided) return RAPID_ESTATE;
    if (n_out) *n_out = (int32_t)f->decision.size();
    if ((int32_t)f->decision.size() > cap) return RAPID_ECAPACITY;
    for (size_t i = 0; i < f->decision.size(); ++i) out[i] = f->decision[i];
    return RAPID_OK;
}

// -------------------------------------------------------------------------------------- wire ingest (host only)
int rapid_endpoint_map_create(const uint8_t* hostnames, const int32_t* host_off, const int32_t* ports, int32_t n,
                              rapid_endpoint_map** out) {
    if (!out || n < 0 || (n > 0 && (!hostnames || !host_off || !ports)) || (n > 0 && host_off[0] < 0)) return RAPID_EINVAL;
    rapid_endpoint_map* m = new rapid_endpoint_map();
    m->index.reserve((size_t)n * 2);
    m->hostname.reserve((size_t)n);
    m->port.assign(ports, ports + n);
    for (int32_t i = 0; i < n; ++i) {
        if (host_off[i + 1] < host_off[i]) {
            delete m;
            return RAPID_EINVAL;
        }
        m->hostname.emplace_back(reinterpret_cast<const char*>(hostnames + host_off[i]), (size_t)(host_off[i + 1] - host_off[i]));
        // the first registration of an endpoint wins, as in the Java-side Endpoint -> int map of the facade
        m->index.emplace(rapid_endpoint_map::key(hostnames + host_off[i], (size_t)(host_off[i + 1] - host_off[i]), ports[i]), i);
    }
    *out = m;
    return RAPID_OK;
}

void rapid_endpoint_map_destroy(rapid_endpoint_map* m) { delete m; }

int rapid_endpoint_map_lookup(const rapid_endpoint_map* m, const uint8_t* hostname, int32_t hostname_len, int32_t port,
                              int32_t* node_out) {
    if (!m || !node_out || hostname_len < 0 || (hostname_len > 0 && !hostname)) return RAPID_EINVAL;
    static const uint8_t empty = 0;
    const auto it = m->index.find(rapid_endpoint_map::key(hostname ? hostname : &empty, (size_t)hostname_len, port));
    if (it == m->index.end()) return RAPID_ENODE_MISSING;
    *node_out = it->second;
    return RAPID_OK;
}

int rapid_decode_request(const uint8_t* msg, int64_t len, int32_t* kind_out, int64_t* payload_off, int64_t* payload_len) {
    if ((!msg && len > 0) || len < 0 || !kind_out) return RAPID_EINVAL;
    rapid_wire::Reader r(msg, len);
    int32_t kind = RAPID_MSG_OTHER;
    int64_t off = 0, plen = 0;
    while (!r.done()) {  // a oneof: the LAST member on the wire wins, as in every protobuf parser
        int wt;
        const uint32_t f = r.tag(&wt);
        if (f >= 1 && f <= 10 && wt == 2) {
            rapid_wire::Reader p = r.sub();
            kind = (int32_t)f;
            off = (int64_t)(p.p - msg);
            plen = (int64_t)(p.end - p.p);
        } else {
            r.skip(wt);
        }
    }
    if (!r.ok) return RAPID_EINVAL;
    *kind_out = kind;
    if (payload_off) *payload_off = off;
    if (payload_len) *payload_len = plen;
    return RAPID_OK;
}

int rapid_decode_batched_alerts(const rapid_endpoint_map* m, const uint8_t* msg, int64_t len, int32_t K,
                                rapid_alert_record* out, int64_t* id_hi, int64_t* id_lo, int32_t cap, int32_t* n_out,
                                int32_t* sender_out) {
    if (!m || (!msg && len > 0) || len < 0 || cap < 0 || (cap > 0 && !out) || !n_out || K < 1 || K > RAPID_MAX_K)
        return RAPID_EINVAL;
    rapid_wire::Reader r(msg, len);
    int32_t n = 0, sender = -1;
    int rc_first = RAPID_OK;
    bool ok = true;
    while (!r.done()) {
        int wt;
        const uint32_t f = r.tag(&wt);
        if (f == 1 && wt == 2) {
            sender = rapid_wire::read_endpoint(r.sub(), *m, &ok);
        } else if (f == 3 && wt == 2) {
            rapid_wire::Reader a = r.sub();
            if (n < cap) {
                const int rc = rapid_wire::read_alert(a, *m, K, out + n, id_hi ? id_hi + n : nullptr, id_lo ? id_lo + n : nullptr);
                if (rc != RAPID_OK && rc_first == RAPID_OK) rc_first = rc;
            }
            ++n;
        } else {
            r.skip(wt);
        }
    }
    if (!r.ok || !ok) return RAPID_EINVAL;
    *n_out = n;
    if (sender_out) *sender_out = sender;
    if (n > cap) return RAPID_ECAPACITY;
    if (rc_first != RAPID_OK) return rc_first;
    if (n > 0) out[n - 1].flags |= RAPID_ALERT_LAST_IN_BATCH;
    return RAPID_OK;
}

int rapid_endpoint_map_add(rapid_endpoint_map* m, const uint8_t* hostname, int32_t hostname_len, int32_t port, int32_t* node_out) {
    if (!m || !node_out || hostname_len < 0 || (hostname_len > 0 && !hostname)) return RAPID_EINVAL;
    static const uint8_t empty = 0;
    const uint8_t* hp = hostname ? hostname : &empty;
    const std::string key = rapid_endpoint_map::key(hp, (size_t)hostname_len, port);
    const auto it = m->index.find(key);
    if (it != m->index.end()) {  // already registered: the first registration wins
        *node_out = it->second;
        return RAPID_OK;
    }
    const int32_t idx = (int32_t)m->hostname.size();
    m->hostname.emplace_back(reinterpret_cast<const char*>(hp), (size_t)hostname_len);
    m->port.push_back(port);
    m->index.emplace(key, idx);
    *node_out = idx;
    return RAPID_OK;
}

int rapid_endpoint_map_add_wire(rapid_endpoint_map* m, const uint8_t* endpoint_msg, int64_t len, int32_t* node_out) {
    if (!m || !node_out || len < 0 || (len > 0 && !endpoint_msg)) return RAPID_EINVAL;
    const uint8_t* host;
    size_t host_len;
    int32_t port;
    if (!rapid_wire::parse_endpoint(rapid_wire::Reader(endpoint_msg, len), &host, &host_len, &port)) return RAPID_EINVAL;
    return rapid_endpoint_map_add(m, host, (int32_t)host_len, port, node_out);
}

int32_t rapid_endpoint_map_size(const rapid_endpoint_map* m) { return m ? (int32_t)m->hostname.size() : 0; }

int rapid_endpoint_map_get(const rapid_endpoint_map* m, int32_t node, uint8_t* hostname_out, int32_t cap, int32_t* hostname_len,
                           int32_t* port_out) {
    if (!m || node < 0 || node >= (int32_t)m->hostname.size() || cap < 0 || (cap > 0 && !hostname_out)) return RAPID_EINVAL;
    const std::string& hn = m->hostname[(size_t)node];
    if (hostname_len) *hostname_len = (int32_t)hn.size();
    if (port_out) *port_out = m->port[(size_t)node];
    if ((int32_t)hn.size() > cap) return RAPID_ECAPACITY;
    std::memcpy(hostname_out, hn.data(), hn.size());
    return RAPID_OK;
}

int rapid_decode_batched_alerts_ex(const rapid_endpoint_map* m, const uint8_t* msg, int64_t len, int32_t K,
                                   rapid_alert_record* out, int64_t* id_hi, int64_t* id_lo, int32_t* status, int64_t* unresolved,
                                   int32_t cap, int32_t* n_out, int32_t* sender_out) {
    if (!m || (!msg && len > 0) || len < 0 || cap < 0 || (cap > 0 && (!out || !status)) || !n_out || K < 1 || K > RAPID_MAX_K)
        return RAPID_EINVAL;
    rapid_wire::Reader r(msg, len);
    int32_t n = 0, sender = -1;
    bool ok = true;
    while (!r.done()) {
        int wt;
        const uint32_t f = r.tag(&wt);
        if (f == 1 && wt == 2) {
            sender = rapid_wire::read_endpoint(r.sub(), *m, &ok);
        } else if (f == 3 && wt == 2) {
            rapid_wire::Reader a = r.sub();
            if (n < cap) {
                std::memset(out + n, 0, sizeof(rapid_alert_record));
                if (unresolved) unresolved[2 * n] = unresolved[2 * n + 1] = -1;
                status[n] = rapid_wire::read_alert(a, *m, K, out + n, id_hi ? id_hi + n : nullptr, id_lo ? id_lo + n : nullptr, msg,
                                                   unresolved ? unresolved + 2 * n : nullptr);
            }
            ++n;
        } else {
            r.skip(wt);
        }
    }
    if (!r.ok || !ok) return RAPID_EINVAL;
    *n_out = n;
    if (sender_out) *sender_out = sender;
    if (n > cap) return RAPID_ECAPACITY;
    // the batch ends at its last alert, decodable or not: the flag goes on the last record the caller can use
    for (int32_t i = n - 1; i >= 0; --i)
        if (status[i] == RAPID_OK) {
            out[i].flags |= RAPID_ALERT_LAST_IN_BATCH;
            break;
        }
    return RAPID_OK;
}

int rapid_decode_fast_round_vote(const rapid_endpoint_map* m, const uint8_t* msg, int64_t len, int32_t* sender_out,
                                 int64_t* config_id_out, int32_t* endpoints_out, int32_t cap, int32_t* n_out) {
    if (!m || (!msg && len > 0) || len < 0 || cap < 0 || (cap > 0 && !endpoints_out) || !n_out) return RAPID_EINVAL;
    rapid_wire::Reader r(msg, len);
    int32_t n = 0, sender = -1;
    int64_t cfg = 0;
    bool ok = true, missing = false;
    while (!r.done()) {
        int wt;
        const uint32_t f = r.tag(&wt);
        if (f == 1 && wt == 2) {
            sender = rapid_wire::read_endpoint(r.sub(), *m, &ok);
        } else if (f == 2 && wt == 0) {
            cfg = (int64_t)r.varint();
        } else if (f == 3 && wt == 2) {
            const int32_t e = rapid_wire::read_endpoint(r.sub(), *m, &ok);
            if (e < 0) missing = true;
            if (n < cap) endpoints_out[n] = e;
            ++n;
        } else {
            r.skip(wt);
        }
    }
    if (!r.ok || !ok) return RAPID_EINVAL;
    *n_out = n;
    if (sender_out) *sender_out = sender;
    if (config_id_out) *config_id_out = cfg;
    if (n > cap) return RAPID_ECAPACITY;
    if (missing || sender < 0) return RAPID_ENODE_MISSING;
    return RAPID_OK;
}

// ------------------------------------------------------------- consensus at one node (host only): consensus.h
int rapid_consensus_create(int32_t my_index, int32_t rank_index, int64_t config_id, int32_t membership_size,
                           rapid_consensus** out) {
    if (!out || membership_size < 1 || my_index < 0) return RAPID_EINVAL;
    rapid_consensus* c = new rapid_consensus();
    c->me = my_index;
    c->rank_index = rank_index;
    c->config_id = config_id;
    c->N = membership_size;
    *out = c;
    return RAPID_OK;
}

void rapid_consensus_destroy(rapid_consensus* c) { delete c; }

int rapid_consensus_propose(rapid_consensus* c, const int32_t* proposal, int32_t n) {
    if (!c || n < 0 || (n > 0 && !proposal)) return RAPID_EINVAL;
    c->propose(rapid_px::Value(proposal, proposal + n));
    return RAPID_OK;
}

int rapid_consensus_handle(rapid_consensus* c, const rapid_consensus_msg* msg, const int32_t* endpoints) {
    if (!c || !msg || msg->n_endpoints < 0 || (msg->n_endpoints > 0 && !endpoints)) return RAPID_EINVAL;
    if (msg->kind < RAPID_MSG_FAST_ROUND_2B || msg->kind > RAPID_MSG_PHASE2B) return RAPID_EINVAL;
    if (msg->config_id != c->config_id) return RAPID_OK;  // every handler starts with this test
    const rapid_px::Value eps(endpoints, endpoints + msg->n_endpoints);
    switch (msg->kind) {
        case RAPID_MSG_FAST_ROUND_2B: c->on_fast_vote(msg->sender, eps); break;
        case RAPID_MSG_PHASE1A: c->on_phase1a(msg->sender, msg->rnd); break;
        case RAPID_MSG_PHASE1B: c->on_phase1b(msg->rnd, msg->vrnd, eps); break;
        case RAPID_MSG_PHASE2A: c->on_phase2a(msg->rnd, eps); break;
        default: c->on_phase2b(msg->sender, msg->rnd, eps); break;
    }
    return RAPID_OK;
}

int rapid_consensus_start_classic_round(rapid_consensus* c) {
    if (!c) return RAPID_EINVAL;
    if (!c->decided) c->start_phase1a(2);  // R/FastPaxos.java:190-196
    return RAPID_OK;
}

int rapid_consensus_start_phase1a(rapid_consensus* c, int32_t round) {
    if (!c) return RAPID_EINVAL;
    c->start_phase1a(round);
    return RAPID_OK;
}

int rapid_consensus_poll(rapid_consensus* c, rapid_consensus_msg* msg_out, int32_t* endpoints_out, int32_t cap,
                         int32_t* got) {
    if (!c || !msg_out || !got || cap < 0 || (cap > 0 && !endpoints_out)) return RAPID_EINVAL;
    *got = 0;
    if (c->outbox.empty()) return RAPID_OK;
    const rapid_px::Outgoing& o = c->outbox.front();
    *msg_out = o.head;
    if ((int32_t)o.endpoints.size() > cap) return RAPID_ECAPACITY;
    std::copy(o.endpoints.begin(), o.endpoints.end(), endpoints_out);
    *got = 1;
    c->outbox.pop_front();
    return RAPID_OK;
}

int rapid_consensus_decision(rapid_consensus* c, int32_t* out, int32_t cap, int32_t* n_out) {
    if (!c || cap < 0 || (cap > 0 && !out)) return RAPID_EINVAL;
    if (!c->decided) return RAPID_ESTATE;
    if (n_out) *n_out = (int32_t)c->decision.size();
    if ((int32_t)c->decision.size() > cap) return RAPID_ECAPACITY;
    std::copy(c->decision.begin(), c->decision.end(), out);
    return RAPID_OK;
}

int rapid_consensus_fallback_delay_ms(int32_t membership_size, int64_t base_delay_ms, double u, int64_t* delay_out) {
    if (!delay_out || membership_size < 1 || !(u >= 0.0 && u < 1.0)) return RAPID_EINVAL;
    const double jitter_rate = 1.0 / (double)membership_size;  // R/FastPaxos.java:76
    *delay_out = (int64_t)(-1000.0 * std::log(1.0 - u) / jitter_rate) + base_delay_ms;
    return RAPID_OK;
}

int rapid_paxos_select_proposal(int32_t membership_size, const rapid_rank* vrnd, const int32_t* vval_off,
                                const int32_t* vvals, int32_t n_msgs, int32_t* chosen_out) {
    if (!chosen_out || n_msgs < 1 || !vrnd || !vval_off || membership_size < 1) return RAPID_EINVAL;
    std::vector<rapid_px::Promise> promises((size_t)n_msgs);
    for (int32_t i = 0; i < n_msgs; ++i) {
        if (vval_off[i + 1] < vval_off[i] || (vval_off[i + 1] > vval_off[i] && !vvals)) return RAPID_EINVAL;
        promises[i].vrnd = vrnd[i];
        if (vval_off[i + 1] > vval_off[i]) promises[i].vval.assign(vvals + vval_off[i], vvals + vval_off[i + 1]);
    }
    *chosen_out = rapid_px::select_promise(membership_size, promises);
    return RAPID_OK;
}

int rapid_classic_round_population(int32_t membership_size, int32_t n_acceptors, const uint64_t* vote_key,
                                   const uint8_t* voted, const int32_t* arrival, rapid_classic_round_result* out) {
    const int32_t N = membership_size;
    if (!out || N < 1 || n_acceptors < 0 || n_acceptors > N || (n_acceptors > 0 && (!vote_key || !voted))) return RAPID_EINVAL;
    *out = rapid_classic_round_result{0, -1, 0, 0, 0};
    // Phase1a to every member, one Phase1b per live acceptor, Phase2a to every member, one Phase2b broadcast per acceptor
    // that accepts (all of them: each promised crnd and last voted in round 1, R/Paxos.java:199)
    out->messages = (int64_t)N + n_acceptors;
    if (arrival) {
        std::vector<uint8_t> seen((size_t)n_acceptors, 0);
        for (int32_t j = 0; j < n_acceptors; ++j) {
            if (arrival[j] < 0 || arrival[j] >= n_acceptors || seen[(size_t)arrival[j]]) return RAPID_EINVAL;
            seen[(size_t)arrival[j]] = 1;
        }
    }
    auto at = [&](int32_t j) { return arrival ? arrival[j] : j; };
    // The coordinator runs its rule on every Phase1b from the (N/2+1)-th on, until a value comes out (R/Paxos.java:172-187),
    // i.e. on the shortest prefix that is a majority AND holds a voter.
    int32_t first_voter = -1;
    for (int32_t j = 0; j < n_acceptors && first_voter < 0; ++j)
        if (voted[at(j)]) first_voter = j;
    const int32_t need = std::max(N / 2 + 1, first_voter + 1);
    if (first_voter < 0 || need > n_acceptors) return RAPID_OK;  // the coordinator never gets to send a Phase2a
    // Voters answered with vrnd = (1, 1), the others with (0, 0) and no value: V = the voters' values in arrival order.
    std::unordered_map<uint64_t, int32_t> count;
    int32_t chosen = -1, rule = 0;
    for (int32_t j = 0; j < need && chosen < 0; ++j) {
        const int32_t a = at(j);
        if (!voted[a]) continue;
        if (++count[vote_key[a]] > N / 4) {
            chosen = a;
            rule = 2;
        }
    }
    if (chosen < 0 || count.size() == 1) {
        // one distinct value (checked on the WHOLE prefix, R/Paxos.java:287), or nothing above N/4: the first voter's value
        bool single = true;
        const uint64_t k0 = vote_key[at(first_voter)];
        for (int32_t j = first_voter + 1; j < need && single; ++j)
            if (voted[at(j)] && vote_key[at(j)] != k0) single = false;
        if (single) {
            chosen = at(first_voter);
            rule = 1;
        } else if (chosen < 0) {
            chosen = at(first_voter);
            rule = 3;
        }
    }
    out->chosen_acceptor = chosen;
    out->promises_used = need;
    out->rule = rule;
    out->messages += (int64_t)N + (int64_t)n_acceptors * N;
    out->decided = 1;  // n_acceptors >= need > N/2 Phase2b messages for the round reach every live node (R/Paxos.java:231)
    return RAPID_OK;
}

// The population form of the recovery with CONCURRENT coordinators and message loss: every live acceptor is a whole rapid_consensus
// object holding its fast-round vote (values travel as one-element lists {index of the first acceptor that holds the key}), behind
// the reference's test network -- ONE FIFO per destination, the single-threaded executor per node of PaxosTests.java:403-476: a
// broadcast is appended to every live node's queue, the sender's included (R/UnicastToAllBroadcaster.java:46-53).  The per-node code is
// the one rapid_consensus_* runs message by message; what this call adds is the network around n of them.
int rapid_classic_rounds_population(int32_t membership_size, int32_t n_acceptors, const uint64_t* vote_key, const uint8_t* voted,
                                    const int32_t* rank_index, const rapid_classic_start* starts, int32_t n_starts,
                                    const int32_t* schedule, const uint8_t* drop, int64_t n_steps, uint64_t seed, double loss,
                                    rapid_classic_rounds_result* out, int32_t* decided_vote_of) {
    const int32_t N = membership_size, n = n_acceptors;
    if (!out || N < 1 || n < 0 || n > N || n > RAPID_CLASSIC_ROUNDS_MAX_ACCEPTORS || (n > 0 && (!vote_key || !voted)) || n_starts < 0 ||
        (n_starts > 0 && !starts) || n_steps < 0 || (n_steps > 0 && schedule == nullptr && drop != nullptr) || !(loss >= 0.0 && loss < 1.0))
        return RAPID_EINVAL;
    for (int32_t j = 0; j < n_starts; ++j)
        if (starts[j].acceptor < 0 || starts[j].acceptor >= n || starts[j].step < 0 || (j > 0 && starts[j].step < starts[j - 1].step)) return RAPID_EINVAL;
    *out = rapid_classic_rounds_result{};
    out->chosen_acceptor = -1;
    // values: the first acceptor holding a key stands for it
    std::unordered_map<uint64_t, int32_t> first_of;
    std::vector<rapid_consensus> nodes((size_t)n);
    for (int32_t a = 0; a < n; ++a) {
        rapid_consensus& c = nodes[(size_t)a];
        c.me = a;
        c.rank_index = rank_index ? rank_index[a] : a + 2;
        c.config_id = 1;
        c.N = N;
        if (voted[a]) {  // registerFastRoundVote (R/Paxos.java:246-259); the fast round's own messages are not part of this call
            const int32_t id = first_of.emplace(vote_key[a], a).first->second;
            c.rnd = rapid_rank{1, 1};
            c.vrnd = c.rnd;
            c.vval = rapid_px::Value{id};
        }
    }
    struct Msg {
        rapid_consensus_msg head;
        int32_t value;  // -1: no value
    };
    std::vector<Msg> msgs;
    std::vector<std::deque<int32_t>> queues((size_t)n);
    std::vector<int32_t> pending;          // nodes with queued messages (seeded mode), with their places
    std::vector<int32_t> place((size_t)n, -1);
    auto mark = [&](int32_t d) {
        if (place[(size_t)d] < 0) {
            place[(size_t)d] = (int32_t)pending.size();
            pending.push_back(d);
        }
    };
    auto unmark = [&](int32_t d) {
        const int32_t at = place[(size_t)d];
        if (at < 0) return;
        const int32_t last = pending.back();
        pending[(size_t)at] = last;
        place[(size_t)last] = at;
        pending.pop_back();
        place[(size_t)d] = -1;
    };
    auto drain = [&](int32_t a) {  // what node a wants sent, in its order
        rapid_consensus& c = nodes[(size_t)a];
        while (!c.outbox.empty()) {
            const rapid_px::Outgoing& o = c.outbox.front();
            const int32_t id = (int32_t)msgs.size();
            msgs.push_back(Msg{o.head, o.endpoints.empty() ? -1 : o.endpoints[0]});
            const int k = o.head.kind - RAPID_MSG_PHASE1A;
            if (o.head.dest == RAPID_DEST_BROADCAST) {
                for (int32_t d = 0; d < n; ++d) {
                    queues[(size_t)d].push_back(id);
                    mark(d);
                }
                if (k >= 0 && k < 4) out->sent[k] += n;
            } else if (o.head.dest >= 0 && o.head.dest < n) {
                queues[(size_t)o.head.dest].push_back(id);
                mark(o.head.dest);
                if (k >= 0 && k < 4) out->sent[k] += 1;
            }
            c.outbox.pop_front();
        }
    };
    uint64_t rng = seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
    auto next = [&]() {  // splitmix64
        uint64_t z = (rng += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    int32_t next_start = 0;
    int64_t step = 0;
    for (;; ++step) {
        while (next_start < n_starts && starts[next_start].step <= step) {
            nodes[(size_t)starts[next_start].acceptor].start_phase1a(starts[next_start].round);
            drain(starts[next_start].acceptor);
            ++next_start;
        }
        int32_t d;
        bool lose;
        if (schedule) {
            if (step >= n_steps) break;
            d = schedule[step];
            if (d < 0 || d >= n || queues[(size_t)d].empty()) return RAPID_EINVAL;  // (the schedule names a node with nothing queued)
            lose = drop != nullptr && drop[step] != 0;
        } else {
            if (pending.empty()) {
                if (next_start >= n_starts) break;
                step = starts[next_start].step - 1;  // (nothing in flight: on to the next coordinator's start)
                continue;
            }
            if (n_steps > 0 && step >= n_steps) break;
            d = pending[(size_t)(next() % (uint64_t)pending.size())];
            lose = loss > 0.0 && (double)(next() >> 11) * (1.0 / 9007199254740992.0) < loss;
        }
        const int32_t id = queues[(size_t)d].front();
        queues[(size_t)d].pop_front();
        if (queues[(size_t)d].empty()) unmark(d);
        const Msg m = msgs[(size_t)id];
        const int k = m.head.kind - RAPID_MSG_PHASE1A;
        if (lose) {
            out->lost += 1;
            continue;
        }
        if (k >= 0 && k < 4) out->delivered[k] += 1;
        const int32_t v = m.value;
        const rapid_px::Value eps = v < 0 ? rapid_px::Value{} : rapid_px::Value{v};
        rapid_consensus& c = nodes[(size_t)d];
        switch (m.head.kind) {
            case RAPID_MSG_PHASE1A: c.on_phase1a(m.head.sender, m.head.rnd); break;
            case RAPID_MSG_PHASE1B: c.on_phase1b(m.head.rnd, m.head.vrnd, eps); break;
            case RAPID_MSG_PHASE2A: c.on_phase2a(m.head.rnd, eps); break;
            case RAPID_MSG_PHASE2B: c.on_phase2b(m.head.sender, m.head.rnd, eps); break;
            default: break;
        }
        drain(d);
    }
    out->steps = step;
    int32_t agreed = 1;
    for (int32_t a = 0; a < n; ++a) {
        const rapid_consensus& c = nodes[(size_t)a];
        const int32_t v = c.decided && !c.decision.empty() ? c.decision[0] : -1;
        if (decided_vote_of) decided_vote_of[a] = v;
        if (c.decided) {
            out->decided_nodes += 1;
            if (out->chosen_acceptor < 0) out->chosen_acceptor = v;
            else if (out->chosen_acceptor != v) agreed = 0;  // (Paxos never lets this happen: reported, not assumed)
        }
    }
    out->agreed = agreed;
    for (int32_t a = 0; a < n; ++a) out->undelivered += (int64_t)queues[(size_t)a].size();
    return RAPID_OK;
}

int rapid_decode_consensus_message(const rapid_endpoint_map* m, int32_t kind, const uint8_t* msg, int64_t len,
                                   rapid_consensus_msg* out, int32_t* endpoints_out, int32_t cap) {
    if (!m || !out || (!msg && len > 0) || len < 0 || cap < 0 || (cap > 0 && !endpoints_out)) return RAPID_EINVAL;
    if (kind < RAPID_MSG_FAST_ROUND_2B || kind > RAPID_MSG_PHASE2B) return RAPID_EINVAL;
    // field numbers: sender 1, configurationId 2 everywhere; the ranks and the endpoint list per message (rapid.proto:124-169)
    const uint32_t f_rnd = kind == RAPID_MSG_FAST_ROUND_2B ? 0u : 3u;
    const uint32_t f_vrnd = kind == RAPID_MSG_PHASE1B ? 4u : 0u;
    const uint32_t f_list = kind == RAPID_MSG_FAST_ROUND_2B ? 3u
                            : kind == RAPID_MSG_PHASE2B     ? 4u
                            : kind == RAPID_MSG_PHASE1A     ? 0u
                                                            : 5u;
    rapid_wire::Reader r(msg, len);
    rapid_consensus_msg h{};
    h.kind = kind;
    h.sender = -1;
    h.dest = RAPID_DEST_BROADCAST;
    bool ok = true, missing = false;
    int32_t n = 0;
    while (!r.done()) {
        int wt;
        const uint32_t f = r.tag(&wt);
        if (f == 1 && wt == 2) {
            h.sender = rapid_wire::read_endpoint(r.sub(), *m, &ok);
        } else if (f == 2 && wt == 0) {
            h.config_id = (int64_t)r.varint();
        } else if (f_rnd && f == f_rnd && wt == 2) {
            h.rnd = rapid_wire::read_rank(r.sub(), &ok);
        } else if (f_vrnd && f == f_vrnd && wt == 2) {
            h.vrnd = rapid_wire::read_rank(r.sub(), &ok);
        } else if (f_list && f == f_list && wt == 2) {
            const int32_t e = rapid_wire::read_endpoint(r.sub(), *m, &ok);
            if (e < 0) missing = true;
            if (n < cap) endpoints_out[n] = e;
            ++n;
        } else {
            r.skip(wt);
        }
    }
    if (!r.ok || !ok) return RAPID_EINVAL;
    h.n_endpoints = n;
    *out = h;
    if (n > cap) return RAPID_ECAPACITY;
    if (missing || h.sender < 0) return RAPID_ENODE_MISSING;
    return RAPID_OK;
}

int rapid_encode_consensus_request(const rapid_endpoint_map* m, const rapid_consensus_msg* msg, const int32_t* endpoints,
                                   uint8_t* out, int64_t cap, int64_t* len_out) {
    if (!m || !msg || !len_out || cap < 0 || (cap > 0 && !out) || msg->n_endpoints < 0 || (msg->n_endpoints > 0 && !endpoints))
        return RAPID_EINVAL;
    const int32_t kind = msg->kind;
    if (kind < RAPID_MSG_FAST_ROUND_2B || kind > RAPID_MSG_PHASE2B) return RAPID_EINVAL;
    const int32_t n_nodes = (int32_t)m->hostname.size();
    if (msg->sender < 0 || msg->sender >= n_nodes) return RAPID_ENODE_MISSING;
    for (int32_t i = 0; i < msg->n_endpoints; ++i)
        if (endpoints[i] < 0 || endpoints[i] >= n_nodes) return RAPID_ENODE_MISSING;
    if (kind == RAPID_MSG_PHASE1A && msg->n_endpoints != 0) return RAPID_EINVAL;
    rapid_wire::Writer body;
    body.message_field(1, rapid_wire::write_endpoint(*m, msg->sender));
    body.int_field(2, msg->config_id);
    if (kind != RAPID_MSG_FAST_ROUND_2B) body.message_field(3, rapid_wire::write_rank(msg->rnd));
    if (kind == RAPID_MSG_PHASE1B) body.message_field(4, rapid_wire::write_rank(msg->vrnd));
    const uint32_t f_list = kind == RAPID_MSG_FAST_ROUND_2B ? 3u : kind == RAPID_MSG_PHASE2B ? 4u : 5u;
    for (int32_t i = 0; i < msg->n_endpoints; ++i) body.message_field(f_list, rapid_wire::write_endpoint(*m, endpoints[i]));
    rapid_wire::Writer req;
    req.message_field((uint32_t)kind, body);
    *len_out = (int64_t)req.b.size();
    if ((int64_t)req.b.size() > cap) return RAPID_ECAPACITY;
    std::memcpy(out, req.b.data(), req.b.size());
    return RAPID_OK;
}

}  // extern "C"
