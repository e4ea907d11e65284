// TEST INFRASTRUCTURE ONLY (see oracle/README.md) -- never linked into the product library.
//
// CPU restatement of the reference's hot-path algorithms (lalithsuresh/rapid, Java), statement by
// statement, for use as the parity checker in tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg.  Citations are relative to /root/reference/rapid/src/main/java/com/vrg/rapid/.
//
//   MembershipView          <- MembershipView.java:58-587
//   MultiNodeCutDetector    <- MultiNodeCutDetector.java:38-179
//   AlertBatchService       <- MembershipService.java:300-354 (batch), 385-430 (decideViewChange),
//                              644-685 (filter + joiner bookkeeping)
//   FastRound               <- FastPaxos.java:125-156 (handleFastRoundProposal)
//
// Endpoints are interned: a node handle (int) stands for one (hostname bytes, port) pair held in an
// EndpointRegistry, so handle equality <=> protobuf Endpoint equality.  Everything keyed by Endpoint in
// the Java is keyed by handle here; all hashing is done on the real hostname bytes / port.
//
// PARITY STATUS: tally, watermark, invalidation, quorum and view-structure behaviour are pinned by the
// reference's own known-answer tests (tests/test_oracle_kat.py ports T/CutDetectionTest.java:42-301,
// T/FastPaxosWithoutFallbackTests.java:85-148, T/MembershipViewTest.java).  Literal ring orders and
// configuration-ID values are "parity unpinned": no reference test asserts one and no JVM is available
// here; they rest on the XXH64 specification (oracle/xxh64.hpp).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <functional>
#include <map>
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "xxh64.hpp"

namespace oracle {

// ---- wire types (rapid/src/main/proto/rapid.proto:13-17, 50-54, 114-118) --------------------------
struct Endpoint {
    std::string hostname;  // bytes
    int32_t port = 0;
    bool operator==(const Endpoint& o) const { return port == o.port && hostname == o.hostname; }
};
struct NodeId {
    int64_t high = 0, low = 0;
    bool operator==(const NodeId& o) const { return high == o.high && low == o.low; }
};
enum EdgeStatus : int { UP = 0, DOWN = 1 };

struct AlertMessage {  // rapid.proto:102-111
    int src = -1;
    int dst = -1;
    EdgeStatus status = UP;
    int64_t configurationId = 0;
    std::vector<int> ringNumbers;
    NodeId nodeId;  // join protocol only
};

class EndpointRegistry {
   public:
    int intern(const std::string& hostname, int32_t port) {
        const std::string key = hostname + '\0' + std::to_string(port);
        auto it = index_.find(key);
        if (it != index_.end()) return it->second;
        const int h = (int)eps_.size();
        eps_.push_back(Endpoint{hostname, port});
        index_.emplace(key, h);
        return h;
    }
    const Endpoint& get(int h) const { return eps_.at((size_t)h); }
    int size() const { return (int)eps_.size(); }

   private:
    std::vector<Endpoint> eps_;
    std::unordered_map<std::string, int> index_;
};

// ---- exceptions (MembershipView.java:502-519) ------------------------------------------------------
struct NodeAlreadyInRingException : std::runtime_error { using std::runtime_error::runtime_error; };
struct NodeNotInRingException : std::runtime_error { using std::runtime_error::runtime_error; };
struct UUIDAlreadySeenException : std::runtime_error { using std::runtime_error::runtime_error; };

enum JoinStatusCode : int {  // rapid.proto:84-90
    HOSTNAME_ALREADY_IN_RING = 0,
    UUID_ALREADY_IN_RING = 1,
    SAFE_TO_JOIN = 2,
    CONFIG_CHANGED = 3,
    MEMBERSHIP_REJECTED = 4
};

// MembershipView.java:474-500
struct NodeIdComparator {
    bool operator()(const NodeId& a, const NodeId& b) const {
        if (a.high < b.high) return true;
        if (a.high > b.high) return false;
        return a.low < b.low;
    }
};

// MembershipView.java:562-587.  compare() orders by the cached signed 64-bit hash only; endpoints whose
// hashes collide compare equal (TreeSet then treats them as the same element).
class AddressComparator {
   public:
    AddressComparator(const EndpointRegistry* reg, int seed) : reg_(reg), seed_(seed) {}
    // :579-582  hashBytes(hostname) * 31 + hashInt(port), Java long wraparound
    int64_t computeHash(int node) const {
        const Endpoint& e = reg_->get(node);
        const uint64_t hh = xxh64(e.hostname.data(), e.hostname.size(), (uint64_t)(int64_t)seed_);
        const uint64_t hp = xxh64_int(e.port, (uint64_t)(int64_t)seed_);
        return (int64_t)(hh * 31ULL + hp);
    }
    int64_t hashOf(int node) const {
        auto it = cache_.find(node);
        if (it != cache_.end()) return it->second;
        const int64_t h = computeHash(node);
        cache_.emplace(node, h);
        return h;
    }
    bool operator()(int a, int b) const { return hashOf(a) < hashOf(b); }  // Long.compare(...) < 0
    void removeEndpoint(int node) const { cache_.erase(node); }            // :584-586

   private:
    const EndpointRegistry* reg_;
    int seed_;
    mutable std::unordered_map<int, int64_t> cache_;
};

// MembershipView.java:526-557
struct Configuration {
    std::vector<NodeId> nodeIds;
    std::vector<int> endpoints;
};

class MembershipView {
   public:
    using Ring = std::set<int, std::reference_wrapper<const AddressComparator>>;

    // MembershipView.java:58-69
    MembershipView(const EndpointRegistry* reg, int K) : reg_(reg), K_(K) { init(); }
    MembershipView(const MembershipView&) = delete;
    MembershipView& operator=(const MembershipView&) = delete;

    // MembershipView.java:74-89
    MembershipView(const EndpointRegistry* reg, int K, const std::vector<NodeId>& nodeIds,
                   const std::vector<int>& endpoints)
        : reg_(reg), K_(K) {
        init();
        for (int k = 0; k < K_; ++k) {
            for (int e : endpoints) {
                rings_[k].insert(e);
                allNodes_.insert(e);
            }
        }
        for (const NodeId& id : nodeIds) identifiersSeen_.insert(id);
    }

    // :100-115
    JoinStatusCode isSafeToJoin(int node, const NodeId& uuid) const {
        if (allNodes_.count(node)) return HOSTNAME_ALREADY_IN_RING;
        if (identifiersSeen_.count(uuid)) return UUID_ALREADY_IN_RING;
        return SAFE_TO_JOIN;
    }

    // :123-160
    void ringAdd(int node, const NodeId& nodeId) {
        if (isIdentifierPresent(nodeId)) throw UUIDAlreadySeenException("uuid already seen");
        if (rings_[0].count(node)) throw NodeAlreadyInRingException("node already in ring");
        std::unordered_set<int> affectedSubjects;
        for (int k = 0; k < K_; ++k) {
            Ring& endpoints = rings_[k];
            endpoints.insert(node);
            auto it = endpoints.find(node);  // lower(node): greatest element strictly less than node
            if (it != endpoints.end() && it != endpoints.begin()) affectedSubjects.insert(*std::prev(it));
        }
        allNodes_.insert(node);
        for (int s : affectedSubjects) eraseCached(s);
        identifiersSeen_.insert(nodeId);
        shouldUpdateConfigurationId_ = true;
    }

    // :167-201
    void ringDelete(int node) {
        if (!rings_[0].count(node)) throw NodeNotInRingException("node not in ring");
        std::unordered_set<int> affectedSubjects;
        for (int k = 0; k < K_; ++k) {
            Ring& endpoints = rings_[k];
            auto it = endpoints.find(node);
            if (it != endpoints.end() && it != endpoints.begin()) affectedSubjects.insert(*std::prev(it));
            if (it != endpoints.end()) endpoints.erase(it);
            comparators_[k].removeEndpoint(node);
            eraseCached(node);
        }
        allNodes_.erase(node);
        for (int s : affectedSubjects) eraseCached(s);
        shouldUpdateConfigurationId_ = true;
    }

    // :210-224 (memoised, with the stale-entry behaviour of :143-152/:181-195 -- SURVEY quirk Q4)
    const std::vector<int>& getObserversOf(int node) const {
        if (!allNodes_.count(node)) throw NodeNotInRingException("node not in ring");
        auto& cache = activeCache();
        auto it = cache.find(node);
        if (it == cache.end()) it = cache.emplace(node, computeObserversOf(node)).first;
        return it->second;
    }

    // ---- one cachedObservers PER NODE OF THE CLUSTER (test infrastructure for quirk Q4) ----
    // In a deployment every node holds its own MembershipView object, hence its own cachedObservers (:49), filled by what THAT
    // node asked for (MultiNodeCutDetector.java:147-149: the subjects in its preProposal at a batch end) and invalidated by the
    // ringAdd / ringDelete every node applies alike.  The rings are the same at every node, so ONE ring structure stands for all
    // of them here; with per-node caches enabled, selectNode(n) makes getObserversOf read and fill node n's cache, and a view
    // change drops the affected entries from every node's cache.  selectNode(-1): back to the view's own (shared) cache.
    void enablePerNodeCaches(bool on) {
        perNode_ = on;
        if (!on) {
            nodeCaches_.clear();
            active_ = -1;
        }
    }
    void selectNode(int node) const { active_ = perNode_ ? node : -1; }
    bool nodeHasCached(int node, int subject) const {
        auto it = nodeCaches_.find(node);
        return it != nodeCaches_.end() && it->second.count(subject) != 0;
    }

    // :234-257 (fresh computation, bypassing the cache; public here so tests can detect Q4 staleness)
    std::vector<int> computeObserversOf(int node) const {
        if (!rings_[0].count(node)) throw NodeNotInRingException("node not in ring");
        if (rings_[0].size() <= 1) return {};
        std::vector<int> observers;
        for (int k = 0; k < K_; ++k) {
            const Ring& list = rings_[k];
            auto succ = list.upper_bound(node);  // higher(node)
            observers.push_back(succ == list.end() ? *list.begin() : *succ);
        }
        return observers;
    }

    // :267-282
    std::vector<int> getSubjectsOf(int node) const {
        if (!allNodes_.count(node)) throw NodeNotInRingException("node not in ring");
        if (rings_[0].size() <= 1) return {};
        return getPredecessorsOf(node);
    }

    // :292-303
    std::vector<int> getExpectedObserversOf(int node) const {
        if (rings_[0].empty()) return {};
        return getPredecessorsOf(node);
    }

    bool isHostPresent(int node) const { return allNodes_.count(node) != 0; }                    // :330-337
    bool isIdentifierPresent(const NodeId& id) const { return identifiersSeen_.count(id) != 0; }  // :345-352

    // :360-372
    int64_t getCurrentConfigurationId() const {
        if (shouldUpdateConfigurationId_) {
            updateCurrentConfigurationId();
            shouldUpdateConfigurationId_ = false;
        }
        return currentConfigurationId_;
    }

    std::vector<int> getRing(int k) const { return std::vector<int>(rings_[k].begin(), rings_[k].end()); }  // :380-388

    // :397-418
    std::vector<int> getRingNumbers(int observer, int subject) const {
        const std::vector<int> subjects = getSubjectsOf(observer);
        std::vector<int> ringIndexes;
        int ringNumber = 0;
        for (int node : subjects) {
            if (node == subject) ringIndexes.push_back(ringNumber);
            ringNumber++;
        }
        return ringIndexes;
    }

    int getMembershipSize() const { return (int)rings_[0].size(); }  // :425-432

    // :450-462
    const Configuration& getConfiguration() const {
        if (shouldUpdateConfigurationId_) {
            updateCurrentConfigurationId();
            shouldUpdateConfigurationId_ = false;
        }
        return currentConfiguration_;
    }

    const AddressComparator& getRingZeroComparator() const { return comparators_[0]; }  // :470-472
    int64_t ringKey(int k, int node) const { return comparators_[k].hashOf(node); }
    int K() const { return K_; }
    const EndpointRegistry* registry() const { return reg_; }

    // :544-556
    static int64_t configurationIdOf(const EndpointRegistry& reg, const std::vector<NodeId>& identifiers,
                                     const std::vector<int>& endpoints) {
        uint64_t hash = 1;
        for (const NodeId& id : identifiers) {
            hash = hash * 37 + xxh64_long(id.high, 0);
            hash = hash * 37 + xxh64_long(id.low, 0);
        }
        for (int e : endpoints) {
            const Endpoint& ep = reg.get(e);
            hash = hash * 37 + xxh64(ep.hostname.data(), ep.hostname.size(), 0);
            hash = hash * 37 + xxh64_int(ep.port, 0);
        }
        return (int64_t)hash;
    }

   private:
    void init() {
        comparators_.reserve((size_t)K_);
        for (int k = 0; k < K_; ++k) comparators_.emplace_back(reg_, k);
        for (int k = 0; k < K_; ++k) rings_.emplace_back(std::cref(comparators_[k]));
    }
    // :308-322
    std::vector<int> getPredecessorsOf(int node) const {
        std::vector<int> subjects;
        for (int k = 0; k < K_; ++k) {
            const Ring& list = rings_[k];
            auto it = list.lower_bound(node);  // first element >= node; lower(node) is the one before it
            subjects.push_back(it == list.begin() ? *list.rbegin() : *std::prev(it));
        }
        return subjects;
    }
    // :438-441
    void updateCurrentConfigurationId() const {
        currentConfiguration_.nodeIds.assign(identifiersSeen_.begin(), identifiersSeen_.end());
        currentConfiguration_.endpoints.assign(rings_[0].begin(), rings_[0].end());
        currentConfigurationId_ =
            configurationIdOf(*reg_, currentConfiguration_.nodeIds, currentConfiguration_.endpoints);
    }

    const EndpointRegistry* reg_;
    int K_;
    std::vector<AddressComparator> comparators_;
    std::vector<Ring> rings_;
    std::set<NodeId, NodeIdComparator> identifiersSeen_;
    std::unordered_map<int, std::vector<int>>& activeCache() const { return (perNode_ && active_ >= 0) ? nodeCaches_[active_] : cachedObservers_; }
    void eraseCached(int subject) {
        cachedObservers_.erase(subject);
        for (auto& kv : nodeCaches_) kv.second.erase(subject);
    }
    mutable std::unordered_map<int, std::vector<int>> cachedObservers_;
    mutable std::unordered_map<int, std::unordered_map<int, std::vector<int>>> nodeCaches_;  // per-node caches (Q4 test mode)
    bool perNode_ = false;
    mutable int active_ = -1;
    std::unordered_set<int> allNodes_;
    mutable int64_t currentConfigurationId_ = -1;
    mutable Configuration currentConfiguration_;
    mutable bool shouldUpdateConfigurationId_ = true;
};

// ---- MultiNodeCutDetector.java ---------------------------------------------------------------------
class MultiNodeCutDetector {
   public:
    // Iteration order of the Java HashSet snapshot in invalidateFailingEdges (:145-146) is unspecified;
    // the result set does not depend on it.  Tests exercise all three orders.
    enum SnapshotOrder { ASCENDING = 0, DESCENDING = 1, SHUFFLED = 2 };

    // :51-60
    MultiNodeCutDetector(int K, int H, int L) : K_(K), H_(H), L_(L) {
        if (H > K || L > H || K < K_MIN || L <= 0 || H <= 0)
            throw std::invalid_argument("Arguments do not satisfy K > H >= L >= 0");
    }

    int getNumProposals() const { return proposalCount_; }  // :62-66

    // :76-82
    std::vector<int> aggregateForProposal(const AlertMessage& msg) {
        std::vector<int> proposals;
        for (int ringNumber : msg.ringNumbers) {
            std::vector<int> r = aggregateForProposal(msg.src, msg.dst, msg.status, ringNumber);
            proposals.insert(proposals.end(), r.begin(), r.end());
        }
        return proposals;
    }

    // :137-164
    std::vector<int> invalidateFailingEdges(const MembershipView& view) {
        if (!seenLinkDownEvents_) return {};
        std::vector<int> proposalsToReturn;
        std::vector<int> preProposalCopy = snapshot(preProposal_);
        for (int nodeInFlux : preProposalCopy) {
            const bool present = view.isHostPresent(nodeInFlux);
            const std::vector<int> observers =
                present ? view.getObserversOf(nodeInFlux) : view.getExpectedObserversOf(nodeInFlux);
            int ringNumber = 0;
            for (int observer : observers) {
                if (contains(proposal_, observer) || contains(preProposal_, observer)) {
                    const EdgeStatus st = present ? DOWN : UP;
                    std::vector<int> r = aggregateForProposal(observer, nodeInFlux, st, ringNumber);
                    proposalsToReturn.insert(proposalsToReturn.end(), r.begin(), r.end());
                }
                ringNumber++;
            }
        }
        return proposalsToReturn;
    }

    // :169-178
    void clear() {
        reportsPerHost_.clear();
        proposal_.clear();
        updatesInProgress_ = 0;
        proposalCount_ = 0;
        preProposal_.clear();
        seenLinkDownEvents_ = false;
    }

    void setSnapshotOrder(SnapshotOrder o) { order_ = o; }
    // introspection for tests
    int reportCount(int dst) const {
        auto it = reportsPerHost_.find(dst);
        return it == reportsPerHost_.end() ? 0 : (int)it->second.size();
    }
    int updatesInProgress() const { return updatesInProgress_; }
    bool seenLinkDownEvents() const { return seenLinkDownEvents_; }

   private:
    static bool contains(const std::set<int>& v, int x) { return v.count(x) != 0; }
    static void removeFrom(std::set<int>& v, int x) { v.erase(x); }
    static void addTo(std::set<int>& v, int x) { v.insert(x); }
    std::vector<int> snapshot(const std::set<int>& s) const {
        std::vector<int> c(s.begin(), s.end());
        if (order_ == DESCENDING) std::reverse(c.begin(), c.end());
        if (order_ == SHUFFLED) {  // deterministic Fisher-Yates on an LCG
            uint64_t st = 0x9E3779B97F4A7C15ULL ^ (uint64_t)c.size();
            for (size_t i = c.size(); i > 1; --i) {
                st = st * 6364136223846793005ULL + 1442695040888963407ULL;
                std::swap(c[i - 1], c[(size_t)((st >> 33) % i)]);
            }
        }
        return c;
    }

    // :84-128
    std::vector<int> aggregateForProposal(int linkSrc, int linkDst, EdgeStatus edgeStatus, int ringNumber) {
        if (edgeStatus == DOWN) seenLinkDownEvents_ = true;
        std::map<int, int>& reportsForHost = reportsPerHost_[linkDst];
        if (reportsForHost.count(ringNumber)) return {};  // duplicate announcement, ignore.
        reportsForHost[ringNumber] = linkSrc;
        const int numReportsForHost = (int)reportsForHost.size();
        if (numReportsForHost == L_) {
            updatesInProgress_++;
            addTo(preProposal_, linkDst);
        }
        if (numReportsForHost == H_) {
            removeFrom(preProposal_, linkDst);
            addTo(proposal_, linkDst);
            updatesInProgress_--;
            if (updatesInProgress_ == 0) {
                proposalCount_++;
                std::vector<int> ret(proposal_.begin(), proposal_.end());
                proposal_.clear();
                return ret;
            }
        }
        return {};
    }

    static constexpr int K_MIN = 3;
    const int K_, H_, L_;
    int proposalCount_ = 0;
    int updatesInProgress_ = 0;
    std::unordered_map<int, std::map<int, int>> reportsPerHost_;
    std::set<int> proposal_;     // Set<Endpoint> (HashSet in the Java; iteration order unspecified there)
    std::set<int> preProposal_;  // Set<Endpoint>
    bool seenLinkDownEvents_ = false;
    SnapshotOrder order_ = ASCENDING;
};

// ---- FastPaxos.java:125-156 (fast round only) ------------------------------------------------------
struct FastRoundVote {
    int sender = -1;
    int64_t configurationId = 0;
    std::vector<int> endpoints;
};

class FastRound {
   public:
    FastRound(int64_t configurationId, int membershipSize, std::function<void(const std::vector<int>&)> onDecide)
        : configurationId_(configurationId), membershipSize_(membershipSize), onDecide_(std::move(onDecide)) {}

    void handleFastRoundProposal(const FastRoundVote& m) {
        if (m.configurationId != configurationId_) return;  // :126-132
        if (votesReceived_.count(m.sender)) return;         // :134-136
        if (decided_) return;                               // :138-140
        votesReceived_.insert(m.sender);
        const int count = ++votesPerProposal_[m.endpoints];
        const int F = (int)std::floor((double)(membershipSize_ - 1) / 4.0);  // :145
        if ((long)votesReceived_.size() >= membershipSize_ - F) {
            if (count >= membershipSize_ - F) {
                decided_ = true;  // onDecidedWrapped, :78-85
                if (onDecide_) onDecide_(m.endpoints);
            }
        }
    }
    bool decided() const { return decided_; }

   private:
    const int64_t configurationId_;
    const long membershipSize_;
    std::function<void(const std::vector<int>&)> onDecide_;
    std::map<std::vector<int>, int> votesPerProposal_;
    std::unordered_set<int> votesReceived_;
    bool decided_ = false;
};

// ---- MembershipService.java: the alert-batch path at ONE receiver -----------------------------------
class AlertBatchService {
   public:
    AlertBatchService(MembershipView* view, MultiNodeCutDetector* cd) : view_(view), cd_(cd) {}

    // MembershipService.java:300-354.  Returns the proposal handed to fastPaxos.propose (sorted by the
    // ring-0 comparator, :346-348); empty if this batch announced nothing.
    std::vector<int> handleBatchedAlertMessage(const std::vector<AlertMessage>& batch) {
        const int64_t currentConfigurationId = view_->getCurrentConfigurationId();
        if (announcedProposal_) return {};  // :318-319 -- lazy stream never runs (quirk Q6)
        std::vector<int> proposal;          // Set<Endpoint>
        for (const AlertMessage& msg : batch) {
            if (!filterAlertMessage(msg, currentConfigurationId)) continue;  // :311
            extractJoinerUuid(msg);                                         // :314
            for (int e : cd_->aggregateForProposal(msg)) addUnique(proposal, e);
        }
        for (int e : cd_->invalidateFailingEdges(*view_)) addUnique(proposal, e);  // :330
        if (proposal.empty()) return {};
        announcedProposal_ = true;  // :335
        const AddressComparator& c0 = view_->getRingZeroComparator();
        std::stable_sort(proposal.begin(), proposal.end(), [&](int a, int b) { return c0(a, b); });
        return proposal;
    }

    // MembershipService.java:385-430
    void decideViewChange(const std::vector<int>& proposal) {
        for (int node : proposal) {
            if (view_->isHostPresent(node)) {
                view_->ringDelete(node);
            } else {
                auto it = joinerUuid_.find(node);
                if (it == joinerUuid_.end()) throw std::logic_error("joinerUuid missing (assert :405)");
                const NodeId id = it->second;
                joinerUuid_.erase(it);
                view_->ringAdd(node, id);
            }
        }
        (void)view_->getCurrentConfigurationId();
        cd_->clear();
        announcedProposal_ = false;
    }

    bool announcedProposal() const { return announcedProposal_; }

   private:
    static void addUnique(std::vector<int>& v, int x) {
        if (std::find(v.begin(), v.end(), x) == v.end()) v.push_back(x);
    }
    // :644-675
    bool filterAlertMessage(const AlertMessage& m, int64_t currentConfigurationId) const {
        if (currentConfigurationId != m.configurationId) return false;
        if (m.status == UP && view_->isHostPresent(m.dst)) return false;
        if (m.status == DOWN && !view_->isHostPresent(m.dst)) return false;
        return true;
    }
    // :677-685
    void extractJoinerUuid(const AlertMessage& m) {
        if (m.status == UP) joinerUuid_[m.dst] = m.nodeId;
    }

    MembershipView* view_;
    MultiNodeCutDetector* cd_;
    bool announcedProposal_ = false;
    std::unordered_map<int, NodeId> joinerUuid_;
};

}  // namespace oracle
