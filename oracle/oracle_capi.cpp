// TEST INFRASTRUCTURE ONLY (see oracle/README.md) -- never linked into the product library.
//
// C entry points over oracle/rapid_oracle.hpp so that tests/, smoke() and bench.py's cpu_baseline leg can
// drive the CPU restatement through ctypes.  Nothing under rapid_amd/ may load this library.
#include <cstdio>
#include <cstring>
#include <memory>
#include <thread>

#include "fast_cut.hpp"
#include "rapid_oracle.hpp"

using namespace oracle;

namespace {
// Same 20-byte packed alert record as include/rapid_mi355x.h (rapid_alert_record).
#pragma pack(push, 1)
struct PackedAlert {
    int64_t cfg_id;
    uint32_t src;
    uint32_t dst;
    uint16_t ring_mask;
    uint8_t status;  // 0 = UP, 1 = DOWN (rapid.proto:114-117)
    uint8_t flags;   // bit0: last record of its BatchedAlertMessage
};
#pragma pack(pop)
static_assert(sizeof(PackedAlert) == 20, "packed alert record must be 20 bytes");

constexpr int ORC_OK = 0;
constexpr int ORC_EINVAL = -1;
constexpr int ORC_ENODE_EXISTS = -2;
constexpr int ORC_ENODE_MISSING = -3;
constexpr int ORC_EUUID_SEEN = -4;
constexpr int ORC_ECAPACITY = -5;

struct IdTable {
    std::vector<NodeId> ids;  // NodeId of every interned node handle (joiners included)
};

AlertMessage unpack(const PackedAlert& r, const IdTable* ids) {
    AlertMessage m;
    m.src = (int)r.src;
    m.dst = (int)r.dst;
    m.status = r.status ? DOWN : UP;
    m.configurationId = r.cfg_id;
    for (int k = 0; k < 16; ++k)
        if (r.ring_mask & (1u << k)) m.ringNumbers.push_back(k);  // ascending, as getRingNumbers (:397-418)
    if (ids && (size_t)m.dst < ids->ids.size()) m.nodeId = ids->ids[(size_t)m.dst];
    return m;
}

int copy_out(const std::vector<int>& v, int32_t* out, int cap) {
    if ((int)v.size() > cap) return ORC_ECAPACITY;
    for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
    return (int)v.size();
}
}  // namespace

extern "C" {

uint64_t orc_xxh64(const void* data, uint64_t len, uint64_t seed) { return xxh64(data, (size_t)len, seed); }

// ---- registry ----------------------------------------------------------------------------------------
void* orc_registry_new() { return new EndpointRegistry(); }
void orc_registry_free(void* r) { delete static_cast<EndpointRegistry*>(r); }
int orc_registry_intern(void* r, const char* host, int hostlen, int port) {
    return static_cast<EndpointRegistry*>(r)->intern(std::string(host, (size_t)hostlen), port);
}

// ---- view ----------------------------------------------------------------------------------------------
void* orc_view_new(void* reg, int K) { return new MembershipView(static_cast<EndpointRegistry*>(reg), K); }
void* orc_view_new_from(void* reg, int K, const int64_t* id_hi, const int64_t* id_lo, int n_ids,
                        const int32_t* endpoints, int n_eps) {
    std::vector<NodeId> ids((size_t)n_ids);
    for (int i = 0; i < n_ids; ++i) ids[(size_t)i] = NodeId{id_hi[i], id_lo[i]};
    std::vector<int> eps(endpoints, endpoints + n_eps);
    return new MembershipView(static_cast<EndpointRegistry*>(reg), K, ids, eps);
}
void orc_view_free(void* v) { delete static_cast<MembershipView*>(v); }
int orc_view_ring_add(void* v, int node, int64_t hi, int64_t lo) {
    try {
        static_cast<MembershipView*>(v)->ringAdd(node, NodeId{hi, lo});
        return ORC_OK;
    } catch (const UUIDAlreadySeenException&) {
        return ORC_EUUID_SEEN;
    } catch (const NodeAlreadyInRingException&) {
        return ORC_ENODE_EXISTS;
    }
}
int orc_view_ring_delete(void* v, int node) {
    try {
        static_cast<MembershipView*>(v)->ringDelete(node);
        return ORC_OK;
    } catch (const NodeNotInRingException&) {
        return ORC_ENODE_MISSING;
    }
}
int orc_view_is_safe_to_join(void* v, int node, int64_t hi, int64_t lo) {
    return (int)static_cast<MembershipView*>(v)->isSafeToJoin(node, NodeId{hi, lo});
}
int orc_view_observers(void* v, int node, int32_t* out, int cap) {
    try {
        return copy_out(static_cast<MembershipView*>(v)->getObserversOf(node), out, cap);
    } catch (const NodeNotInRingException&) {
        return ORC_ENODE_MISSING;
    }
}
int orc_view_observers_fresh(void* v, int node, int32_t* out, int cap) {
    try {
        return copy_out(static_cast<MembershipView*>(v)->computeObserversOf(node), out, cap);
    } catch (const NodeNotInRingException&) {
        return ORC_ENODE_MISSING;
    }
}
int orc_view_subjects(void* v, int node, int32_t* out, int cap) {
    try {
        return copy_out(static_cast<MembershipView*>(v)->getSubjectsOf(node), out, cap);
    } catch (const NodeNotInRingException&) {
        return ORC_ENODE_MISSING;
    }
}
int orc_view_expected_observers(void* v, int node, int32_t* out, int cap) {
    return copy_out(static_cast<MembershipView*>(v)->getExpectedObserversOf(node), out, cap);
}
int orc_view_ring_numbers(void* v, int observer, int subject, int32_t* out, int cap) {
    try {
        return copy_out(static_cast<MembershipView*>(v)->getRingNumbers(observer, subject), out, cap);
    } catch (const NodeNotInRingException&) {
        return ORC_ENODE_MISSING;
    }
}
int orc_view_ring(void* v, int k, int32_t* out, int cap) {
    return copy_out(static_cast<MembershipView*>(v)->getRing(k), out, cap);
}
int64_t orc_view_ring_key(void* v, int k, int node) { return static_cast<MembershipView*>(v)->ringKey(k, node); }
int orc_view_is_host_present(void* v, int node) { return static_cast<MembershipView*>(v)->isHostPresent(node); }
int orc_view_is_identifier_present(void* v, int64_t hi, int64_t lo) {
    return static_cast<MembershipView*>(v)->isIdentifierPresent(NodeId{hi, lo});
}
int orc_view_size(void* v) { return static_cast<MembershipView*>(v)->getMembershipSize(); }
int64_t orc_view_config_id(void* v) { return static_cast<MembershipView*>(v)->getCurrentConfigurationId(); }
// Configuration snapshot: identifiers (sorted) then ring-0 endpoints.
int orc_view_configuration(void* v, int64_t* id_hi, int64_t* id_lo, int id_cap, int32_t* eps, int ep_cap,
                           int* n_ids, int* n_eps) {
    const Configuration& c = static_cast<MembershipView*>(v)->getConfiguration();
    *n_ids = (int)c.nodeIds.size();
    *n_eps = (int)c.endpoints.size();
    if (*n_ids > id_cap || *n_eps > ep_cap) return ORC_ECAPACITY;
    for (int i = 0; i < *n_ids; ++i) {
        id_hi[i] = c.nodeIds[(size_t)i].high;
        id_lo[i] = c.nodeIds[(size_t)i].low;
    }
    for (int i = 0; i < *n_eps; ++i) eps[i] = c.endpoints[(size_t)i];
    return ORC_OK;
}

// ---- cut detector ----------------------------------------------------------------------------------------
void* orc_cd_new(int K, int H, int L) {
    try {
        return new MultiNodeCutDetector(K, H, L);
    } catch (const std::invalid_argument&) {
        return nullptr;
    }
}
void orc_cd_free(void* cd) { delete static_cast<MultiNodeCutDetector*>(cd); }
int orc_cd_aggregate(void* cd, int src, int dst, int status, const int32_t* rings, int nrings, int32_t* out,
                     int cap) {
    AlertMessage m;
    m.src = src;
    m.dst = dst;
    m.status = status ? DOWN : UP;
    m.ringNumbers.assign(rings, rings + nrings);
    return copy_out(static_cast<MultiNodeCutDetector*>(cd)->aggregateForProposal(m), out, cap);
}
int orc_cd_invalidate(void* cd, void* view, int32_t* out, int cap) {
    return copy_out(static_cast<MultiNodeCutDetector*>(cd)->invalidateFailingEdges(*static_cast<MembershipView*>(view)),
                    out, cap);
}
int orc_cd_num_proposals(void* cd) { return static_cast<MultiNodeCutDetector*>(cd)->getNumProposals(); }
int orc_cd_report_count(void* cd, int dst) { return static_cast<MultiNodeCutDetector*>(cd)->reportCount(dst); }
void orc_cd_clear(void* cd) { static_cast<MultiNodeCutDetector*>(cd)->clear(); }
void orc_cd_set_snapshot_order(void* cd, int order) {
    static_cast<MultiNodeCutDetector*>(cd)->setSnapshotOrder((MultiNodeCutDetector::SnapshotOrder)order);
}

// ---- one receiver's alert-batch service --------------------------------------------------------------
struct OrcService {
    std::unique_ptr<MultiNodeCutDetector> cd;
    std::unique_ptr<AlertBatchService> svc;
    IdTable ids;
};
void* orc_svc_new(void* view, int K, int H, int L, const int64_t* id_hi, const int64_t* id_lo, int n_ids) {
    auto* s = new OrcService();
    try {
        s->cd.reset(new MultiNodeCutDetector(K, H, L));
    } catch (const std::invalid_argument&) {
        delete s;
        return nullptr;
    }
    s->svc.reset(new AlertBatchService(static_cast<MembershipView*>(view), s->cd.get()));
    s->ids.ids.resize((size_t)n_ids);
    for (int i = 0; i < n_ids; ++i) s->ids.ids[(size_t)i] = NodeId{id_hi[i], id_lo[i]};
    return s;
}
void orc_svc_free(void* s) { delete static_cast<OrcService*>(s); }
void orc_svc_set_snapshot_order(void* s, int order) {
    static_cast<OrcService*>(s)->cd->setSnapshotOrder((MultiNodeCutDetector::SnapshotOrder)order);
}
// One BatchedAlertMessage (n packed records; flags ignored).  Returns the proposal size (0 = none).
int orc_svc_handle_batch(void* sv, const void* records, int n, int32_t* out, int cap) {
    auto* s = static_cast<OrcService*>(sv);
    const PackedAlert* r = static_cast<const PackedAlert*>(records);
    std::vector<AlertMessage> batch;
    batch.reserve((size_t)n);
    for (int i = 0; i < n; ++i) batch.push_back(unpack(r[i], &s->ids));
    return copy_out(s->svc->handleBatchedAlertMessage(batch), out, cap);
}
int orc_svc_announced(void* sv) { return static_cast<OrcService*>(sv)->svc->announcedProposal(); }
int orc_svc_num_proposals(void* sv) { return static_cast<OrcService*>(sv)->cd->getNumProposals(); }
int orc_svc_report_count(void* sv, int dst) { return static_cast<OrcService*>(sv)->cd->reportCount(dst); }
int orc_svc_decide(void* sv, const int32_t* proposal, int n) {
    try {
        static_cast<OrcService*>(sv)->svc->decideViewChange(std::vector<int>(proposal, proposal + n));
        return ORC_OK;
    } catch (const UUIDAlreadySeenException&) {
        return ORC_EUUID_SEEN;
    } catch (const NodeAlreadyInRingException&) {
        return ORC_ENODE_EXISTS;
    } catch (const NodeNotInRingException&) {
        return ORC_ENODE_MISSING;
    } catch (const std::logic_error&) {
        return ORC_EINVAL;
    }
}

// ---- fast round ------------------------------------------------------------------------------------------
struct OrcFastRound {
    std::unique_ptr<FastRound> fr;
    std::vector<int> decided;
};
void* orc_fr_new(int64_t cfg, int membership_size) {
    auto* f = new OrcFastRound();
    f->fr.reset(new FastRound(cfg, membership_size, [f](const std::vector<int>& d) { f->decided = d; }));
    return f;
}
void orc_fr_free(void* f) { delete static_cast<OrcFastRound*>(f); }
// returns 1 if a decision exists after this vote, else 0
int orc_fr_vote(void* f, int sender, int64_t cfg, const int32_t* eps, int n) {
    FastRoundVote v;
    v.sender = sender;
    v.configurationId = cfg;
    v.endpoints.assign(eps, eps + n);
    static_cast<OrcFastRound*>(f)->fr->handleFastRoundProposal(v);
    return static_cast<OrcFastRound*>(f)->fr->decided() ? 1 : 0;
}
int orc_fr_decided(void* f, int32_t* out, int cap) {
    auto* p = static_cast<OrcFastRound*>(f);
    if (!p->fr->decided()) return -1;
    return copy_out(p->decided, out, cap);
}

// ---- whole population, faithful: every receiver is an AlertBatchService over the SAME (read-only) view
// records: concatenated per-receiver streams; rec_off[R+1] in records; batches delimited by flags bit0
// (and by the end of the receiver's stream).  Outputs per receiver: index of the batch that announced a
// proposal (-1 if none), getNumProposals(), number of batches consumed, and the proposal (ascending node
// index) in a CSR list.  Returns 0 or ORC_ECAPACITY.
// 1 (default): orc_sim_run queries getObserversOf of EVERY member before its (possibly multi-threaded) run, i.e. it models
// nodes whose observer cache (R/MembershipView.java:210-224, quirk Q4) holds every subject.  0: the cache is filled only
// by what the receivers of the run actually ask for (MultiNodeCutDetector.java:147-149: the subjects in preProposal) --
// the union of the per-node caches of the simulated receivers, which is what a multi-round run must carry from one
// configuration to the next; the run is then single-threaded (the cache is written while it is read).
static int g_prewarm_observers = 1;
void orc_sim_set_prewarm(int on) { g_prewarm_observers = on; }
// Per-node observer caches (MembershipView::enablePerNodeCaches): receiver r of the next orc_sim_run calls is cluster node
// receiver_nodes[r] and reads / fills THAT node's cache; nullptr: the view's one shared cache again.  Implies no prewarming.
static const int32_t* g_receiver_nodes = nullptr;
void orc_sim_set_receiver_nodes(void* view_p, const int32_t* receiver_nodes) {
    g_receiver_nodes = receiver_nodes;
    if (receiver_nodes != nullptr) static_cast<MembershipView*>(view_p)->enablePerNodeCaches(true);
}
void orc_view_per_node_caches(void* view_p, int on) { static_cast<MembershipView*>(view_p)->enablePerNodeCaches(on != 0); }
int orc_view_node_has_cached(void* view_p, int node, int subject) { return static_cast<MembershipView*>(view_p)->nodeHasCached(node, subject) ? 1 : 0; }

int orc_sim_run(void* view_p, int K, int H, int L, const int64_t* id_hi, const int64_t* id_lo, int n_ids,
                const void* records, const int64_t* rec_off, int R, int snapshot_order, int nthreads,
                int32_t* out_emit_batch, int32_t* out_num_proposals, int64_t* out_prop_off, int32_t* out_props,
                int64_t props_cap) {
    MembershipView* view = static_cast<MembershipView*>(view_p);
    const PackedAlert* recs = static_cast<const PackedAlert*>(records);
    // prewarm every lazily-filled cache so the concurrent phase only reads
    (void)view->getCurrentConfigurationId();
    for (int k = 0; k < K; ++k)
        for (int n = 0; n < n_ids; ++n) (void)view->ringKey(k, n);
    view->selectNode(-1);
    if (g_prewarm_observers && g_receiver_nodes == nullptr) {
        for (int n = 0; n < n_ids; ++n)
            if (view->isHostPresent(n)) (void)view->getObserversOf(n);
    } else {
        nthreads = 1;
    }
    IdTable ids;
    ids.ids.resize((size_t)n_ids);
    for (int i = 0; i < n_ids; ++i) ids.ids[(size_t)i] = NodeId{id_hi[i], id_lo[i]};

    std::vector<std::vector<int>> props((size_t)R);
    auto work = [&](int r0, int r1) {
        for (int r = r0; r < r1; ++r) {
            if (g_receiver_nodes != nullptr) view->selectNode(g_receiver_nodes[r]);  // this receiver's own cachedObservers
            MultiNodeCutDetector cd(K, H, L);
            cd.setSnapshotOrder((MultiNodeCutDetector::SnapshotOrder)snapshot_order);
            AlertBatchService svc(view, &cd);
            int emit = -1, b = 0;
            std::vector<AlertMessage> batch;
            for (int64_t i = rec_off[r]; i < rec_off[r + 1]; ++i) {
                batch.push_back(unpack(recs[i], &ids));
                const bool last = (recs[i].flags & 1) || i + 1 == rec_off[r + 1];
                if (!last) continue;
                if (emit < 0) {
                    std::vector<int> p = svc.handleBatchedAlertMessage(batch);
                    if (!p.empty()) {
                        emit = b;
                        std::sort(p.begin(), p.end());
                        props[(size_t)r] = std::move(p);
                    }
                }
                batch.clear();
                ++b;
            }
            out_emit_batch[r] = emit;
            out_num_proposals[r] = cd.getNumProposals();
        }
    };
    if (nthreads <= 1) {
        work(0, R);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; ++t)
            th.emplace_back(work, (int)((int64_t)R * t / nthreads), (int)((int64_t)R * (t + 1) / nthreads));
        for (auto& t : th) t.join();
    }
    view->selectNode(-1);
    int64_t off = 0;
    for (int r = 0; r < R; ++r) {
        out_prop_off[r] = off;
        off += (int64_t)props[(size_t)r].size();
    }
    out_prop_off[R] = off;
    if (off > props_cap) return ORC_ECAPACITY;
    for (int r = 0; r < R; ++r)
        std::copy(props[(size_t)r].begin(), props[(size_t)r].end(), out_props + out_prop_off[r]);
    return ORC_OK;
}

// ---- whole population, optimised CPU formulation (oracle/fast_cut.hpp) ----------------------------------
// obs/subj: [n_nodes][K] int32 tables (observers of members / expected observers of non-members; subjects
// of members, -1 rows for non-members); member[n_nodes] bytes.
int orc_fast_sim_run(int n_nodes, int K, int H, int L, int64_t cfg_id, const int32_t* obs, const int32_t* subj,
                     const uint8_t* member, const void* records, const int64_t* rec_off, int R, int nthreads,
                     int32_t* out_emit_batch, int32_t* out_num_proposals, int64_t* out_prop_off,
                     int32_t* out_props, int64_t props_cap) {
    const PackedAlert* recs = static_cast<const PackedAlert*>(records);
    std::vector<std::vector<int>> props((size_t)R);
    auto work = [&](int r0, int r1) {
        FastCutReceiver rx(n_nodes, K, H, L, cfg_id, obs, subj, member);
        for (int r = r0; r < r1; ++r) {
            rx.reset();
            rx.run(reinterpret_cast<const FastCutReceiver::Rec*>(recs + rec_off[r]), rec_off[r + 1] - rec_off[r]);
            out_emit_batch[r] = rx.emitBatch();
            out_num_proposals[r] = rx.numProposals();
            props[(size_t)r] = rx.proposalSorted();
        }
    };
    if (nthreads <= 1) {
        work(0, R);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; ++t)
            th.emplace_back(work, (int)((int64_t)R * t / nthreads), (int)((int64_t)R * (t + 1) / nthreads));
        for (auto& t : th) t.join();
    }
    int64_t off = 0;
    for (int r = 0; r < R; ++r) {
        out_prop_off[r] = off;
        off += (int64_t)props[(size_t)r].size();
    }
    out_prop_off[R] = off;
    if (off > props_cap) return ORC_ECAPACITY;
    for (int r = 0; r < R; ++r)
        std::copy(props[(size_t)r].begin(), props[(size_t)r].end(), out_props + out_prop_off[r]);
    return ORC_OK;
}

}  // extern "C"
