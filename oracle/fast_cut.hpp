// TEST INFRASTRUCTURE ONLY (see oracle/README.md) -- never linked into the product library.
//
// Optimised CPU formulation of ONE receiver's alert path (MultiNodeCutDetector.java:84-164 +
// MembershipService.java:300-354, 644-675) on dense node indices: a K-bit ring mask per subject instead
// of HashMap<Endpoint, Map<Integer, Endpoint>>, and an *incremental* implicit-edge invalidation that only
// visits the nodes that crossed the L watermark since the last pass instead of re-scanning the whole
// preProposal set after every batch (the Java does |preProposal| x K lookups per batch).
//
// Equivalence argument (DESIGN.md "Incremental invalidation"): an implicit report (o -> s, ring k) is
// applicable iff s is in preProposal and o is in proposal U preProposal.  Implicit reports never move a
// node across L, so membership of "proposal U preProposal" grows only through explicit reports; a pair
// becomes applicable at the first batch end after BOTH endpoints crossed L, i.e. when the later of the
// two is among the nodes that crossed L since the previous pass.  Pairs whose endpoints both crossed L
// earlier were applied by an earlier pass and are duplicates (no-ops) now.  The same holds for a joiner in flux and its
// expected observers: the pair is looked at when the later of the two crosses L.
//
// This file is validated against the faithful restatement (rapid_oracle.hpp) on randomised streams in
// tests/test_oracle_fast.py, and is then used (a) as the full-size checker for the GPU kernels and
// (b) as the "optimised CPU" leg of bench.py.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

namespace oracle {

class FastCutReceiver {
   public:
#pragma pack(push, 1)
    struct Rec {
        int64_t cfg_id;
        uint32_t src;
        uint32_t dst;
        uint16_t ring_mask;
        uint8_t status;
        uint8_t flags;
    };
#pragma pack(pop)

    FastCutReceiver(int n_nodes, int K, int H, int L, int64_t cfg_id, const int32_t* obs, const int32_t* subj,
                    const uint8_t* member)
        : n_(n_nodes), K_(K), H_(H), L_(L), cfg_(cfg_id), obs_(obs), subj_(subj), member_(member),
          mask_((size_t)n_nodes, 0), flushed_((size_t)n_nodes, 0) {
        // who is an EXPECTED observer of which registered non-member (row j of the observer table for a non-member j holds its
        // expected observers, R/MembershipView.java:292-322): sorted by observer, so that a node that has just entered
        // proposal U preProposal finds the joiners it vouches for without a walk over all joiners (10^6-node rounds have 5,000)
        for (int j = 0; j < n_nodes; ++j) {
            if (member[j]) continue;
            for (int k = 0; k < K; ++k) {
                const int o = obs[(size_t)j * K + k];
                if (o >= 0) rev_.push_back({o, j, k});
            }
        }
        std::sort(rev_.begin(), rev_.end(), [](const Rev& a, const Rev& b) { return a.o != b.o ? a.o < b.o : (a.j != b.j ? a.j < b.j : a.k < b.k); });
    }

    void reset() {
        for (int d : touched_) {
            mask_[(size_t)d] = 0;
            flushed_[(size_t)d] = 0;
        }
        touched_.clear();
        pending_.clear();
        unflushedH_.clear();
        proposal_.clear();
        inprog_ = 0;
        proposalCount_ = 0;
        seenDown_ = false;
        announced_ = false;
        emitBatch_ = -1;
    }

    // Feed one receiver's whole delivered stream (batches delimited by flags bit0 / end of stream).
    void run(const Rec* recs, int64_t n) {
        int batch = 0;
        bool batchEmitted = false;
        for (int64_t i = 0; i < n; ++i) {
            const Rec& r = recs[i];
            if (!announced_) {
                if (passesFilter(r)) batchEmitted |= applyRecord((int)r.dst, r.ring_mask, r.status != 0);
            }
            const bool last = (r.flags & 1) || i + 1 == n;
            if (last) {
                if (!announced_) {
                    batchEmitted |= invalidate();
                    if (batchEmitted) {  // MembershipService.java:333-335
                        announced_ = true;
                        emitBatch_ = batch;
                    }
                }
                batchEmitted = false;
                ++batch;
            }
        }
    }

    int emitBatch() const { return emitBatch_; }
    int numProposals() const { return proposalCount_; }
    std::vector<int> proposalSorted() const {
        std::vector<int> p = proposal_;
        std::sort(p.begin(), p.end());
        return p;
    }
    int reportCount(int d) const { return __builtin_popcount(mask_[(size_t)d]); }

   private:
    int count(int d) const { return __builtin_popcount(mask_[(size_t)d]); }
    bool inPre(int d) const {
        const int c = count(d);
        return c >= L_ && c < H_;
    }
    bool inAct(int d) const { return count(d) >= L_ && !flushed_[(size_t)d]; }

    // MembershipService.java:644-675
    bool passesFilter(const Rec& r) const {
        if (r.cfg_id != cfg_) return false;
        const bool present = member_[r.dst] != 0;
        if (r.status == 0 && present) return false;   // UP about a member
        if (r.status != 0 && !present) return false;  // DOWN about a non-member
        return true;
    }

    // MultiNodeCutDetector.java:76-128 for every ring of one alert (ascending ring order)
    bool applyRecord(int dst, uint16_t bits, bool down) {
        bool emitted = false;
        for (int k = 0; k < 16; ++k) {
            if (!(bits & (1u << k))) continue;
            if (down) seenDown_ = true;
            emitted |= applyBit(dst, k);
        }
        return emitted;
    }

    bool applyBit(int dst, int k) {
        uint16_t& m = mask_[(size_t)dst];
        if (m & (1u << k)) return false;
        if (m == 0) touched_.push_back(dst);
        m = (uint16_t)(m | (1u << k));
        const int c = __builtin_popcount(m);
        if (c == L_) {
            ++inprog_;
            pending_.push_back(dst);
        }
        if (c == H_) {
            --inprog_;
            unflushedH_.push_back(dst);
            if (inprog_ == 0) {
                ++proposalCount_;
                for (int d : unflushedH_) {
                    flushed_[(size_t)d] = 1;
                    proposal_.push_back(d);
                }
                unflushedH_.clear();
                return true;
            }
        }
        return false;
    }

    // MultiNodeCutDetector.java:137-164, incremental form
    bool invalidate() {
        if (!seenDown_) return false;
        if (pending_.empty()) return false;
        struct Cand { int s, k, o; };
        std::vector<Cand> cands;
        for (int n : pending_) {
            if (inPre(n)) {  // n as the node in flux
                for (int k = 0; k < K_; ++k) {
                    const int o = obs_[(size_t)n * K_ + k];
                    if (o >= 0 && inAct(o)) cands.push_back({n, k, o});
                }
            }
            if (inAct(n) && member_[n]) {  // n as an observer of its (member) subjects ...
                for (int k = 0; k < K_; ++k) {
                    const int s = subj_[(size_t)n * K_ + k];
                    if (s >= 0 && inPre(s)) cands.push_back({s, k, n});
                }
                // ... and as an expected observer of joiners in flux (a joiner that crosses L itself is "the node in flux" above:
                // its row of the observer table holds its expected observers)
                auto it = std::lower_bound(rev_.begin(), rev_.end(), n, [](const Rev& a, int o) { return a.o < o; });
                for (; it != rev_.end() && it->o == n; ++it)
                    if (inPre(it->j)) cands.push_back({it->j, it->k, n});
            }
        }
        pending_.clear();
        bool emitted = false;
        for (const Cand& c : cands)
            if (inAct(c.o)) emitted |= applyBit(c.s, c.k);
        return emitted;
    }

    const int n_, K_, H_, L_;
    const int64_t cfg_;
    const int32_t* obs_;
    const int32_t* subj_;
    const uint8_t* member_;
    std::vector<uint16_t> mask_;
    std::vector<uint8_t> flushed_;
    struct Rev { int o, j, k; };
    std::vector<Rev> rev_;
    std::vector<int> touched_, pending_, unflushedH_, proposal_;
    int inprog_ = 0, proposalCount_ = 0, emitBatch_ = -1;
    bool seenDown_ = false, announced_ = false;
};

}  // namespace oracle
