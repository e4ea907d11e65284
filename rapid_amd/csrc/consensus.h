// Consensus for one configuration change at ONE node: the fast round of R/FastPaxos.java:62-204 and
// the classic-Paxos recovery of R/Paxos.java:57-328, as a host-side state machine.  Control plane: a handful of small
// messages per view change, branchy, no data parallelism -- host C++ by design (the population-wide vote count is the
// GPU's, vote_kernels.h).
//
// Instead of an IBroadcaster / IMessagingClient the object has an outbox: every handler appends the messages the Java
// would have sent, in the order it would have sent them; the embedding service drains it (rapid_consensus_poll) and
// owns the network and the fallback timer (R/FastPaxos.java:107-109).
#pragma once
#include <stdint.h>

#include <algorithm>
#include <deque>
#include <map>
#include <unordered_set>
#include <utility>
#include <vector>

#include "../../include/rapid_mi355x.h"

namespace rapid_px {

using Value = std::vector<int32_t>;  // List<Endpoint>: ordered, compared element by element

// Paxos.compareRanks, R/Paxos.java:333-339: by round, then by the index of the node that started the round
inline int compare(const rapid_rank& a, const rapid_rank& b) {
    if (a.round != b.round) return a.round < b.round ? -1 : 1;
    if (a.node_index != b.node_index) return a.node_index < b.node_index ? -1 : 1;
    return 0;
}
inline bool same(const rapid_rank& a, const rapid_rank& b) { return a.round == b.round && a.node_index == b.node_index; }

struct Promise {  // what the coordinator keeps of a Phase1bMessage
    rapid_rank vrnd;
    Value vval;
};

// The coordinator's choice among the promises it holds (Figure 2 of the Fast Paxos report, as the reference applies
// it: R/Paxos.java:271-328).  Returns the index of the promise whose value is chosen, or -1 when no promise carries a
// value (the Java returns the empty list then, and the coordinator waits for more promises).
inline int select_promise(int32_t N, const std::vector<Promise>& promises) {
    // k = the largest vrnd; V = the non-empty values promised at k, in arrival order
    size_t top = 0;
    for (size_t i = 1; i < promises.size(); ++i)
        if (compare(promises[i].vrnd, promises[top].vrnd) > 0) top = i;
    std::vector<int> at_top;
    for (size_t i = 0; i < promises.size(); ++i)
        if (same(promises[i].vrnd, promises[top].vrnd) && !promises[i].vval.empty()) at_top.push_back((int)i);
    bool one_value = !at_top.empty();
    for (size_t j = 1; j < at_top.size() && one_value; ++j)
        one_value = promises[at_top[j]].vval == promises[at_top[0]].vval;
    if (one_value) return at_top[0];  // :287-289
    if (at_top.size() > 1) {          // :293-307: the first value, in arrival order, seen more than N/4 times
        std::map<Value, int32_t> seen;
        for (int i : at_top)
            if (++seen[promises[i].vval] > N / 4) return i;
    }
    for (size_t i = 0; i < promises.size(); ++i)  // :319-326: anything proposed at all, whatever its round
        if (!promises[i].vval.empty()) return (int)i;
    return -1;
}

struct Outgoing {
    rapid_consensus_msg head;
    Value endpoints;
};

}  // namespace rapid_px

struct rapid_consensus {
    // identity
    int32_t me = 0;
    int32_t rank_index = 0;  // stands in for myAddr.hashCode() of R/Paxos.java:102 (see include/rapid_mi355x.h)
    int64_t config_id = 0;
    int32_t N = 0;
    // fast round (R/FastPaxos.java:125-156)
    std::unordered_set<int32_t> fast_voters;
    std::map<rapid_px::Value, int32_t> fast_tally;
    // acceptor
    rapid_rank rnd{0, 0}, vrnd{0, 0};
    rapid_px::Value vval;
    // coordinator
    rapid_rank crnd{0, 0};
    rapid_px::Value cval;
    std::vector<rapid_px::Promise> promises;
    // learner: who accepted in which round
    std::map<std::pair<int32_t, int32_t>, std::unordered_set<int32_t>> accepted_by;
    bool classic_decided = false;  // Paxos.decided
    bool decided = false;          // FastPaxos.decided
    rapid_px::Value decision;
    std::deque<rapid_px::Outgoing> outbox;

    void emit(int32_t kind, int32_t dest, rapid_rank r, rapid_rank vr, const rapid_px::Value& eps) {
        rapid_px::Outgoing o;
        o.head.kind = kind;
        o.head.sender = me;
        o.head.config_id = config_id;
        o.head.rnd = r;
        o.head.vrnd = vr;
        o.head.n_endpoints = (int32_t)eps.size();
        o.head.dest = dest;
        o.endpoints = eps;
        outbox.push_back(std::move(o));
    }

    // onDecidedWrapped, R/FastPaxos.java:78-85.  The Java asserts that this runs once; without -ea a classic-round
    // majority arriving after a fast-round decision would announce the view change twice.  Here the first decision stands.
    void decide(const rapid_px::Value& v) {
        if (decided) return;
        decided = true;
        decision = v;
    }

    // R/FastPaxos.java:95-110 (the timer is the caller's) + R/Paxos.java:246-259
    void propose(const rapid_px::Value& proposal) {
        if (rnd.round <= 1) {  // registerFastRoundVote: not once a classic round has been joined
            rnd = rapid_rank{1, 1};
            vrnd = rnd;
            vval = proposal;
        }
        emit(RAPID_MSG_FAST_ROUND_2B, RAPID_DEST_BROADCAST, rapid_rank{0, 0}, rapid_rank{0, 0}, proposal);
    }

    // R/FastPaxos.java:125-156
    void on_fast_vote(int32_t sender, const rapid_px::Value& eps) {
        if (fast_voters.count(sender) || decided) return;
        fast_voters.insert(sender);
        const int32_t count = ++fast_tally[eps];
        const int32_t quorum = N - (N - 1) / 4;  // N - floor((N-1)/4.0), :145
        if ((int32_t)fast_voters.size() >= quorum && count >= quorum) decide(eps);
    }

    // R/Paxos.java:98-111
    void start_phase1a(int32_t round) {
        if (crnd.round > round) return;
        crnd = rapid_rank{round, rank_index};
        emit(RAPID_MSG_PHASE1A, RAPID_DEST_BROADCAST, crnd, rapid_rank{0, 0}, {});
    }

    // acceptor, R/Paxos.java:118-148: promise, and report the last vote
    void on_phase1a(int32_t sender, rapid_rank rank) {
        if (rapid_px::compare(rnd, rank) >= 0) return;
        rnd = rank;
        emit(RAPID_MSG_PHASE1B, sender, rnd, vrnd, vval);
    }

    // coordinator, R/Paxos.java:156-188
    void on_phase1b(rapid_rank r, rapid_rank vr, const rapid_px::Value& vv) {
        if (!rapid_px::same(crnd, r)) return;
        promises.push_back(rapid_px::Promise{vr, vv});  // a List in the Java: no de-duplication by sender
        if ((int32_t)promises.size() <= N / 2 || !cval.empty()) return;
        const int chosen = rapid_px::select_promise(N, promises);
        if (chosen < 0) return;
        cval = promises[chosen].vval;
        emit(RAPID_MSG_PHASE2A, RAPID_DEST_BROADCAST, crnd, rapid_rank{0, 0}, cval);
    }

    // acceptor, R/Paxos.java:195-216
    void on_phase2a(rapid_rank r, const rapid_px::Value& vv) {
        if (rapid_px::compare(rnd, r) > 0 || rapid_px::same(vrnd, r)) return;
        rnd = r;
        vrnd = r;
        vval = vv;
        emit(RAPID_MSG_PHASE2B, RAPID_DEST_BROADCAST, r, rapid_rank{0, 0}, vval);
    }

    // learner, R/Paxos.java:223-238
    void on_phase2b(int32_t sender, rapid_rank r, const rapid_px::Value& eps) {
        auto& who = accepted_by[{r.round, r.node_index}];
        who.insert(sender);
        if ((int32_t)who.size() > N / 2 && !classic_decided) {
            classic_decided = true;
            decide(eps);
        }
    }
};
